"""`python bench.py --gpus N` must start its own ranks when nothing else did (the driver's command form on an 8-GPU node).
CPU: the parent spawns torch.distributed.run before touching the GPU and propagates the children's failure (there is no GPU
here, so every rank refuses to run -- there is no CPU fallback for the product path).  GPU: the two-rank REHEARSAL form
(both ranks on cuda:0, collective over gloo) runs end to end and prints one JSON line with the collective's fields."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT, has_gpu


def _run(extra_env, *argv, timeout=600):
    env = dict(os.environ, **extra_env)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True,
                          timeout=timeout)


@pytest.mark.skipif(has_gpu(), reason="the no-GPU refusal path")
def test_self_spawn_propagates_rank_failure_without_a_gpu():
    r = _run({}, "--gpus", "2", "--steps", "1", "--warmup", "0")
    assert r.returncode != 0
    assert "starting 2 ranks" in r.stderr and "torch.distributed.run" in r.stderr
    assert "needs an MI355X" in r.stderr                      # each child refused: no CPU fallback
    assert "2-rank child exited with code" in r.stderr
    assert not r.stdout.strip()                                # no result line on failure


def test_explicit_world_size_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr


@pytest.mark.gpu
def test_two_rank_rehearsal_end_to_end():
    if not has_gpu():
        pytest.skip("needs a GPU")
    r = _run({"IGN_BENCH_REHEARSAL": "1"}, "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "8", "--alt-steps", "0",
             "--iso-steps", "1", "--cpu-sample", "0")
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["config"]["parallelism"] == "dp2" and res["config"]["global_batch"] == 16
    col = res["config"]["collective"]
    assert col["ranks"] == 2 and col["bucket_bytes"] > 4_000_000 and col["allreduce_ms_per_step"] > 0
    assert "rehearsal" in res["config"] and res["scaling"] == "weak"
