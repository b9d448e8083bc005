"""GPU end-to-end checks of the drop-in driver (BASELINE config 0: UEA BasicMotions shape through run_uea.sh) on a
generated BasicMotions-format fixture (the real archive cannot be fetched): the harness on the HIP models against the
same harness on the CPU oracle, and the shell launcher itself."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu
PKG = os.path.join(ROOT, "speech-imagery-eeg_amd")


def _write_bm(tmp, n=24, C=6, T=100):
    import speech_imagery_eeg_amd  # noqa
    from data_provider.ts_reader import write_ts
    d = os.path.join(tmp, "BasicMotions")
    os.makedirs(d, exist_ok=True)
    classes = ["badminton", "running", "standing", "walking"]
    tt = np.arange(T)
    for split, seed in (("TRAIN", 1), ("TEST", 2)):
        rng = np.random.RandomState(seed)
        X, y = [], []
        for i in range(n):
            k = i % 4
            X.append(rng.randn(C, T) * 0.5 + np.sin(2 * np.pi * (k + 1) * tt / T)[None, :] * (1 + 0.2 * k))
            y.append(classes[k])
        write_ts(os.path.join(d, f"BasicMotions_{split}.ts"), X, y, "BasicMotions", classes)
    return d


def _args(tmp, model):
    import run
    return run.get_args(["--model", model, "--dnn_type", "FCN", "--data", "UEA", "--data_root", tmp, "--dataset",
                         "BasicMotions", "--train_epochs", "3", "--batch_size", "8", "--seed", "0", "--amp",
                         "--log_interval", "1", "--num_shapelet", "3", "--gating_value", "1"])


@pytest.mark.parametrize("model", ["SBM", "InterpGN"])
def test_harness_hip_vs_oracle(tmp_path, monkeypatch, capsys, model):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import speech_imagery_eeg_amd  # noqa
    import run
    from exp.experiment_classification import Experiment
    from oracle import ign_oracle as O
    _write_bm(str(tmp_path))
    monkeypatch.chdir(tmp_path)
    run.set_seed(0)
    e = Experiment(_args(str(tmp_path), model))
    assert e.device.type == "cuda"
    sd0 = {k: v.detach().cpu().clone() for k, v in e.model.state_dict().items()}

    # same harness, CPU oracle models, same initial weights and the same (seeded) batch order
    reg = dict(Experiment.model_dict)
    reg.update(InterpGN=lambda cfg: O.OracleIGN(cfg),
               SBM=lambda configs, num_shapelet, shapelet_len: O.OracleSBM(configs, num_shapelet, shapelet_len))
    monkeypatch.setattr(Experiment, "model_dict", reg)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: False)
    os.makedirs(tmp_path / "cpu", exist_ok=True)
    monkeypatch.chdir(tmp_path / "cpu")
    run.set_seed(0)
    ref = Experiment(_args(str(tmp_path), model))
    assert ref.device.type == "cpu"
    ref.model.load_state_dict(sd0)
    monkeypatch.undo()

    def first_epoch_loss(exp):
        torch.manual_seed(123)                      # DataLoader shuffle order
        exp.args.train_epochs = 1
        exp.train()
        return exp.validation()

    monkeypatch.chdir(tmp_path)
    lg, ag = first_epoch_loss(e)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: False)
    monkeypatch.chdir(tmp_path / "cpu")
    lr, ar = first_epoch_loss(ref)
    monkeypatch.undo()
    assert abs(lg - lr) < 5e-3 * max(1.0, abs(lr)), (lg, lr)
    loss, res, _ = e.test(result_dir=str(tmp_path / "result"))
    assert res.preds.shape == (24,) and np.isfinite(loss)
    # interpretability outputs: where each shapelet matched (window index -> sample range inside the series)
    assert res.t.shape == res.p.shape and res.t.dtype == torch.int32
    assert res.match_start.shape == res.p.shape and res.match_len.shape == (res.p.shape[1],)
    assert int((res.match_start + res.match_len.unsqueeze(0)).max()) <= 100 and int(res.match_start.min()) >= 0
    assert os.path.exists(os.path.join(tmp_path, e.checkpoint_dir, "checkpoint.pth"))


def test_run_uea_sh_launcher(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    data = tmp_path / "data"
    _write_bm(str(data))
    env = dict(os.environ, EPOCHS="2", SEEDS="0", DATA_ROOT=str(data))
    work = tmp_path / "work"
    os.makedirs(work)
    for f in ("run.py", "run_uea.sh"):
        os.symlink(os.path.join(PKG, f), work / f)
    r = subprocess.run(["bash", "run_uea.sh", "BasicMotions"], cwd=work, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "accuracy:" in r.stdout
    ck = work / "checkpoints" / "InterpGN" / "BasicMotions"
    assert any("checkpoint.pth" in fs for _, _, fs in os.walk(ck)), r.stdout[-1000:]


def test_harness_accumulation_and_clipping_on_the_flat_path(tmp_path, monkeypatch):
    """Gradient accumulation (gradients stay per-parameter tensors between backward passes, gathered once per optimizer
    step) and clipping (acts on the gathered flat-buffer views) through the GPU harness: equal to the same run on torch's
    own Adam / per-tensor gradients."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import speech_imagery_eeg_amd  # noqa
    import run
    from exp import experiment_classification as ec
    _write_bm(str(tmp_path))
    monkeypatch.chdir(tmp_path)
    finals = {}
    for flat in (True, False):
        a = _args(str(tmp_path), "InterpGN")
        a.gradient_accumulation_steps, a.gradient_clip, a.train_epochs, a.min_epochs = 2, 0.5, 2, 10
        run.set_seed(0)
        if not flat:           # reference-style optimizer on the same HIP models
            monkeypatch.setattr(ec, "FlatAdam", lambda bucket, lr: torch.optim.Adam(bucket.params, lr=lr))
        e = ec.Experiment(a)
        e.train()
        finals[flat] = {k: v.detach().float().cpu().clone() for k, v in e.model.state_dict().items()}
        monkeypatch.undo()
        monkeypatch.chdir(tmp_path)
    for k, v in finals[True].items():
        w = finals[False][k]
        assert float((v - w).abs().max()) <= 2e-4 * max(1.0, float(w.abs().max())), k


@pytest.mark.parametrize("model,dnn", [("DNN", "ResNet"), ("InterpGN", "ResNet"), ("DNN", "PatchTST"), ("InterpGN", "PatchTST"),
                                      ("DNN", "TimesNet"), ("InterpGN", "TimesNet")])
def test_harness_trains_the_other_deep_experts(tmp_path, monkeypatch, model, dnn):
    """`--dnn_type ResNet|PatchTST|TimesNet` through the unmodified harness (standalone and as the deep expert of the gated
    mixture): three epochs on the generated BasicMotions fixture run on the HIP kernels, the loss falls, a checkpoint with the
    reference's key names is written and reloads, and test() returns the result object."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import speech_imagery_eeg_amd  # noqa
    import run
    from exp.experiment_classification import Experiment
    _write_bm(str(tmp_path))
    monkeypatch.chdir(tmp_path)
    run.set_seed(0)
    a = run.get_args(["--model", model, "--dnn_type", dnn, "--data", "UEA", "--data_root", str(tmp_path), "--dataset",
                      "BasicMotions", "--train_epochs", "3", "--batch_size", "8", "--seed", "0", "--amp", "--log_interval", "1",
                      "--d_model", "32", "--d_ff", "32", "--n_heads", "2", "--e_layers", "1", "--top_k", "2", "--lr", "0.002"])
    e = Experiment(a)
    assert e.device.type == "cuda"
    x = torch.randn(8, 100, 6, device=e.device)
    with torch.no_grad():
        first = e.model(x, torch.ones(8, 100, device=e.device), None, None)
    e.train()
    ckpt = torch.load(os.path.join(e.checkpoint_dir, "checkpoint.pth"), map_location="cpu", weights_only=True)
    assert set(ckpt) == set(e.model.state_dict())
    e.model.load_state_dict(ckpt)
    res = e.test()
    assert res is not None
    with torch.no_grad():
        after = e.model(x, torch.ones(8, 100, device=e.device), None, None)
    first = first[0] if isinstance(first, tuple) else first
    after = after[0] if isinstance(after, tuple) else after
    assert torch.isfinite(after).all() and not torch.allclose(first, after)


def test_harness_default_bf16_autocast_mode_trains(tmp_path, monkeypatch):
    """The reference's DEFAULT mode: `--amp` NOT passed = bf16 autocast on (run.py:100, exp:319).  IGN + FCN through the
    harness inside the autocast region: the shapelet path stays on the fp32 kernels, the FCN body runs the single-product bf16
    kernels (fcn.CONV_MATH "bf16"), training makes progress and the checkpoint reloads."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import speech_imagery_eeg_amd  # noqa
    import run
    from exp.experiment_classification import Experiment
    from ign_hip import fcn
    _write_bm(str(tmp_path))
    monkeypatch.chdir(tmp_path)
    run.set_seed(0)
    a = run.get_args(["--model", "InterpGN", "--dnn_type", "FCN", "--data", "UEA", "--data_root", str(tmp_path), "--dataset",
                      "BasicMotions", "--train_epochs", "3", "--batch_size", "8", "--seed", "0", "--log_interval", "1"])
    assert a.amp is True
    seen = []
    orig = fcn.FcnBodyFn.forward

    def spy(ctx, x, states_math, *params):
        seen.append(states_math[1])
        return orig(ctx, x, states_math, *params)

    monkeypatch.setattr(fcn.FcnBodyFn, "forward", staticmethod(spy))
    e = Experiment(a)
    e.train()
    assert seen and set(seen) == {"bf16"}
    ckpt = torch.load(os.path.join(e.checkpoint_dir, "checkpoint.pth"), map_location="cpu", weights_only=True)
    e.model.load_state_dict(ckpt)
    assert e.test() is not None


def test_hipgraph_harness_run_equals_the_eager_run(tmp_path, monkeypatch):
    """`run.py --hipgraph`: Experiment.train replays the step as one captured hipGraph (first batch of every (beta, lr) eager, then
    capture, then replays; ragged last batch eager).  Same data, same seeds: the per-epoch validation numbers and the final weights
    must equal the eager run's (1e-5: the captured step is the same launch sequence)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import speech_imagery_eeg_amd  # noqa
    import run
    from exp.experiment_classification import Experiment
    monkeypatch.chdir(tmp_path)
    outs = {}
    for mode in ("eager", "graph"):
        argv = ["--model", "InterpGN", "--dnn_type", "FCN", "--data", "SYNTH", "--synthetic", "104,6,100,4", "--dataset", "g" + mode,
                "--batch_size", "32", "--amp", "--train_epochs", "3", "--num_workers", "0", "--seed", "0", "--beta_schedule", "cosine",
                "--lr_decay", "--patience", "10"] + (["--hipgraph"] if mode == "graph" else [])
        a = run.get_args(argv)
        run.set_seed(0)
        e = Experiment(a)
        vals, orig = [], e.validation

        def rec(orig=orig, vals=vals):
            r = orig()
            vals.append(r)
            return r
        e.validation = rec
        torch.manual_seed(123)
        e.train()
        outs[mode] = (vals, {k: v.detach().float().cpu().clone() for k, v in e.model.state_dict().items()})
        if mode == "graph":
            assert getattr(e, "_graphed", None) is not None and e.optimizer.capturable      # the graph path really ran
    for (la, aa), (lb, ab) in zip(*[o[0] for o in outs.values()]):
        assert abs(la - lb) <= 1e-5 * max(1.0, abs(la)) and aa == ab
    for k, v in outs["eager"][1].items():
        w = outs["graph"][1][k]
        assert float((v - w).abs().max()) <= 1e-5 * max(1.0, float(v.abs().max())), k
