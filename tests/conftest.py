import os
import sys
from argparse import Namespace

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    path = os.path.join(GOLDEN, name + ".npz")
    if not os.path.exists(path):
        pytest.skip(f"fixture {name}.npz not generated")
    return dict(np.load(path, allow_pickle=False))


def make_cfg(**kw):
    """The argparse-namespace fields the reference's model ctors read."""
    base = dict(enc_in=6, seq_len=100, num_class=4, epsilon=1.0, distance_func='euclidean',
                memory_efficient=False, sbm_cls='linear', dropout=0.0, lambda_reg=0.1, lambda_div=0.1,
                dnn_type='FCN', task_name='classification', pred_len=0, label_len=0, output_attention=False,
                d_model=64, embed='timeF', freq='h', factor=1, n_heads=4, d_ff=128, activation='gelu',
                e_layers=2, c_out=4, dec_in=6, d_layers=1,
                eegcnn_layers=2, eegcnn_pooling='mean', eegcnn_dropout1=0.0, eegcnn_dropout2=0.0,
                eegcnn_n_heads=8, eegcnn_d_ff=256)
    base.update(kw)
    return Namespace(**base)


def sd_from(g, prefix="sd."):
    import torch
    return {k[len(prefix):]: torch.from_numpy(v) for k, v in g.items() if k.startswith(prefix)}


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(autouse=True)
def _reset_torch_matmul_precision():
    """Experiment.train() sets float32 matmul precision to 'medium' like the reference (exp:297); parity tests that
    run later in the same process must not inherit it."""
    import torch
    torch.set_float32_matmul_precision('highest')
    yield
    torch.set_float32_matmul_precision('highest')


# ---------------------------------------------------------------------------------------------------------------------
# Parity ledger.  Every golden / oracle comparison of a `-m gpu` test goes through `parity()`, which records the error
# actually observed (max abs error, the scale it is judged against, and the error as a multiple of north_star's 1e-4
# budget) before asserting.  At session end the ledger is written to gpurun_out/parity.json; the copy judged is
# committed as profiles/parity_r2.json.  IGN_PARITY_RECORD_ONLY=1 records without asserting (used once to survey).
_LEDGER = {}
NORTH_STAR_TOL = 1e-4


def _np64(a):
    import torch
    if torch.is_tensor(a):
        a = a.detach().double().cpu().numpy()
    return np.asarray(a, dtype=np.float64)


def parity(label, got, ref, tol=NORTH_STAR_TOL, kind="scale", floor=0.0, f64=None, ref_is="reference fp32"):
    """kind="elem":  |got-ref| <= tol*(1+|ref|) element-wise (numpy allclose with rtol=atol=tol) -- for O(1) outputs.
    kind="scale": max|got-ref| <= tol*max(max|ref|, floor) -- for gradients / statistics whose scale is not O(1).
    `f64` (optional) is a float64 run of the same reference module: when the fp32 fixture itself is further than `tol`
    from it, the bound widens to the reference's own distance from float64 (the HIP result must be no further from
    the exact answer than the reference's fp32 result is) -- both distances are recorded."""
    g, r = _np64(got), _np64(ref)
    assert g.shape == r.shape, f"{label}: shape {g.shape} vs {r.shape}"
    diff = np.abs(g - r)
    if kind == "elem":
        budget = float((diff / (1.0 + np.abs(r))).max()) if diff.size else 0.0
        scale = float(np.abs(r).max()) if r.size else 0.0
    else:
        scale = max(float(np.abs(r).max()) if r.size else 0.0, floor, 1e-30)
        budget = float(diff.max()) / scale if diff.size else 0.0
    rec = dict(max_abs_err=float(diff.max()) if diff.size else 0.0, scale=scale, kind=kind, tol=tol,
               err_over_1e4=budget / NORTH_STAR_TOL, against=ref_is)
    ok = budget <= tol
    rec["passed_by"] = "fp32 reference within tol" if ok else "FAILED"
    if f64 is not None:
        t = _np64(f64)
        sc = max(float(np.abs(t).max()), floor, 1e-30)
        e_hip, e_ref = float(np.abs(g - t).max()) / sc, float(np.abs(r - t).max()) / sc
        rec.update(hip_vs_f64=e_hip, ref_fp32_vs_f64=e_ref)
        if not ok and e_hip <= max(tol, e_ref):
            # the widened clause: recorded as such so the ledger shows every comparison that needed it
            ok = True
            rec["passed_by"] = "float64 clause: no further from the float64 run than the reference's own fp32 run"
    test = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0].split("::", 1)[-1]
    _LEDGER.setdefault(test, {})[label] = rec
    if os.environ.get("IGN_PARITY_RECORD_ONLY") != "1":
        assert ok, f"{label}: {rec}"
    return rec


def pytest_sessionfinish(session, exitstatus):
    if not _LEDGER:
        return
    import json
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    recs = [(t, l, r) for t, d in _LEDGER.items() for l, r in d.items()]
    strict = [x for x in recs if x[2]["tol"] <= NORTH_STAR_TOL]
    relaxed = [x for x in recs if x[2]["tol"] > NORTH_STAR_TOL]
    worst = max(strict, key=lambda x: x[2]["err_over_1e4"], default=None)
    with open(os.path.join(out, "parity.json"), "w") as f:
        json.dump(dict(north_star_tol=NORTH_STAR_TOL, n_records=len(recs), n_records_at_1e4=len(strict),
                       worst_at_1e4=None if worst is None else dict(test=worst[0], tensor=worst[1], **worst[2]),
                       records_passed_by_the_float64_clause=[dict(test=t, tensor=l) for t, l, r in recs
                                                             if str(r.get("passed_by", "")).startswith("float64")],
                       records_above_1e4_by_design=[dict(test=t, tensor=l, tol=r["tol"], err_over_1e4=r["err_over_1e4"],
                                                         why=r["against"]) for t, l, r in relaxed],
                       tests=_LEDGER), f, indent=1, sort_keys=True)
