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


@pytest.mark.gpu
def test_one_rank_rccl_rendezvous_emits_the_collective_fields():
    """`python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1`: an explicit WORLD_SIZE=1 rendezvous runs the gradient
    all-reduce on a one-rank RCCL communicator and reports config.collective (backend nccl, bucket bytes, ms per step)."""
    if not has_gpu():
        pytest.skip("needs a GPU")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                        "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                        "--batch", "8", "--alt-steps", "0", "--iso-steps", "1", "--cpu-sample", "0", "--baseline-steps", "0",
                        "--harness-epochs", "0"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    col = res["config"]["collective"]
    assert res["n_gpus"] == 1 and col["ranks"] == 1 and col["backend"].startswith("nccl") and col["bucket_bytes"] > 300_000
    assert col["allreduce_ms_per_step"] > 0


def test_committed_bench_line_carries_the_contract_fields():
    """The driver-facing JSON line (profiles/r3_bench_line.json = the default `python bench.py` run of the round-end measurement
    pass, profiles/collect_round.sh): metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling /
    vs_baseline / dtype / data / config.workload, the roofline object of the dominant kernel, the CPU baseline, configs 3 / 4 --
    and the round-3 additions: config.data says the inputs are resident, the `harness` leg (Experiment.train_one_epoch with
    per-step H2D copies), the launch count, and GEMM groups priced by the products their arithmetic executes."""
    line = json.load(open(os.path.join(ROOT, "profiles", "r3_bench_line.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "baselines", "harness"):
        assert k in line, k
    assert line["unit"] == "epochs/s" and line["n_gpus"] == 1 and line["scaling"] == "weak" and line["vs_baseline"] is None
    assert line["dtype"] == "f32" and line["data"] == "synthetic" and "workload" in line["config"] and "model" not in line["config"]
    assert "resident in HBM" in line["config"]["data"]
    assert abs(line["value"] - line["steps"] * 256 / 8192 / (line["ms_per_step"] * 1e-3 * line["steps"])) < 1e-9 * line["value"] + 1e-12
    roof = line["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"} <= set(roof)
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12 and 0 < roof["frac"] < 1
    assert roof["traffic"] and "NOT measured in this run" in roof["traffic_source"]
    for k, v in roof["isolated"].items():
        if k.startswith("clconv_"):
            assert v["products_per_fp32_product"] == 3.0 and abs(v["frac"] - 3.0 * v["achieved"] / 2500.0) < 1e-9
    cpu = line["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample", "cpu"} <= set(cpu) and cpu["kind"] == "port" and "3 timed" in cpu["sample"]
    har = line["harness"]
    assert har["unit"] == "epochs/s" and 0 < har["value"] <= line["value"] * 1.05 and "H2D" in har["what"] and har["steps_per_epoch"] == 32
    assert line["config"]["launches_per_step"]["per_step"] <= 42
    for name in ("eegcnn", "transformer"):
        b = line["baselines"][name]
        assert b["roofline"]["peak"] == 2500.0 and 0 < b["roofline"]["frac"] < 1          # executed 16-bit flops / dense bf16 peak
        assert all(0 < g["frac"] < 0.6 and g["products_per_fp32_product"] == 3.0 for g in b["gemm_kernels"].values())
        assert b["cpu_baseline"]["kind"] == "port" and b["cpu_baseline"]["value"] > 0
