"""Pin oracle/baselines_oracle.py (CPU restatements of the Transformer-encoder and EEG-CNN baselines, BASELINE.json configs
3 / 4) and the oracle's strided / CHISCO-row shapelet paths against outputs of the reference itself (tests/golden/*.npz)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN, golden, make_cfg, sd_from
from oracle import baselines_oracle as B
from oracle import ign_oracle as O

sys.path.insert(0, GOLDEN)
from fill import seeded_fill  # noqa: E402

TOL = dict(rtol=2e-5, atol=2e-5)


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _grad_check(params, g, tol=1e-4):
    gmax = max(float(np.abs(g[k]).max()) for k in g if k.startswith(("grad.", "gradsample.")))
    n = 0
    for k, p in params.items():
        if "grad." + k in g:
            ref, got = g["grad." + k], p.grad.numpy()
        elif "gradsample." + k in g:
            idx = torch.linspace(0, p.numel() - 1, min(2048, p.numel())).long()
            ref, got = g["gradsample." + k], p.grad.flatten()[idx].numpy()
            assert abs(float(p.grad.double().norm()) - float(g["gradnorm." + k])) <= tol * float(g["gradnorm." + k])
        else:
            continue
        if float(np.abs(ref).max()) < 1e-5 * gmax:      # true-zero gradient (BN1 affine under BN2, key bias): rounding noise on both sides
            assert float(np.abs(got).max()) < 1e-5 * gmax, k
            continue
        assert float(np.abs(got - ref).max()) <= tol * max(float(np.abs(ref).max()), 1e-4 * gmax), k
        n += 1
    return n


def test_transformer_oracle_bm():
    g = golden("transformer_bm")
    p = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in sd_from(g).items()}
    out = B.transformer_logits(p, _t(g["x"]), _t(g["mask"]), n_layers=2, n_heads=4)
    loss = F.cross_entropy(out, _t(g["y"]))
    loss.backward()
    np.testing.assert_allclose(out.detach().numpy(), g["out"], **TOL)
    assert abs(loss.item() - float(g["loss"])) < 1e-5
    assert _grad_check({k: v for k, v in p.items() if v.grad is not None}, g) >= 30


def _filled_state(module_ctor, seed):
    """the parameters the fixture used: the repo's drop-in module is only a container for names / shapes here"""
    torch.manual_seed(0)
    m = seeded_fill(module_ctor(), seed)
    return {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point and k in dict(m.named_parameters()))
            for k, v in m.state_dict().items()}


def test_transformer_oracle_at_benchmark_width():
    import speech_imagery_eeg_amd  # noqa
    from models.Transformer import Model
    g = golden("transformer_ch512")
    cfg = make_cfg(enc_in=122, seq_len=1000, num_class=3, c_out=3, d_model=512, n_heads=8, d_ff=2048)
    p = _filled_state(lambda: Model(cfg), int(g["seed"]))
    out = B.transformer_logits(p, _t(g["x"]), _t(g["mask"]), n_layers=2, n_heads=8)
    loss = F.cross_entropy(out, _t(g["y"]))
    loss.backward()
    np.testing.assert_allclose(out.detach().numpy(), g["out"], **TOL)
    assert abs(loss.item() - float(g["loss"])) < 2e-5
    assert _grad_check({k: v for k, v in p.items() if v.grad is not None}, g) >= 30


@pytest.mark.parametrize("name,d_model", [("eegcnn_ch", 128), ("eegcnn_ch512", 512)])
def test_eegcnn_oracle(name, d_model):
    import speech_imagery_eeg_amd  # noqa
    from models.eegcnn import EEGCNNTransformer
    g = golden(name)
    cfg = make_cfg(enc_in=122, seq_len=1000, num_class=3, c_out=3, d_model=d_model)
    if "seed" in g:
        p = _filled_state(lambda: EEGCNNTransformer(cfg), int(g["seed"]))
    else:
        sd = sd_from(g)
        p = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in sd.items()}
    x, y = _t(g["x"]), _t(g["y"])
    stats = {}
    out = B.eegcnn_logits(p, x, training=True, stats_out=stats)
    loss = F.cross_entropy(out, y)
    loss.backward()
    np.testing.assert_allclose(out.detach().numpy(), g["out"], **TOL)
    assert abs(loss.item() - float(g["loss"])) < 2e-5
    assert _grad_check({k: v for k, v in p.items() if v.grad is not None}, g) >= 20
    if name == "eegcnn_ch512":               # running statistics after ONE training forward, then eval mode
        for k, v in stats.items():
            np.testing.assert_allclose(v.numpy(), g["sd_after." + k], rtol=1e-4, atol=1e-6, err_msg=k)
        pe = {k: v.detach() for k, v in p.items()}
        pe.update(stats)
        with torch.no_grad():
            np.testing.assert_allclose(B.eegcnn_features(pe, x, training=False).numpy(), g["eval_feat"], **TOL)
            np.testing.assert_allclose(B.eegcnn_logits(pe, x, training=False).numpy(), g["eval_out"], **TOL)


@pytest.mark.parametrize("name", ["ch_lts", "ch_cos", "strided_l1", "strided_lts"])
def test_oracle_shapelet_chisco_row_and_strided(name):
    g = golden("shapelet_" + name)
    Bn, C, T, K, L, stride = (int(v) for v in g["dims"])
    gen = torch.Generator().manual_seed(int(g["seed"]))
    xn = torch.randn(Bn, C, T, generator=gen)
    w = torch.randn(K, C, L, generator=gen).requires_grad_(True)
    thr = torch.rand(1, K, C, generator=gen).requires_grad_(True)
    r = torch.randn(Bn, K * C, generator=gen)
    d = O.window_distance(xn, w, stride, O.MODE_COS if name == "ch_cos" else O.MODE_L1, chunk=16)
    p, dmin = O.lts_softmin_gate(d, thr) if name.endswith("lts") else O.rbf_straight_through_max(d, 1.0)
    (p * r).sum().backward()
    np.testing.assert_allclose(p.detach().numpy(), g["p"], **TOL)
    np.testing.assert_allclose(dmin.detach().numpy(), g["dmin"], **TOL)
    sc = float(np.abs(g["grad_w"]).max())
    assert float(np.abs(w.grad.numpy() - g["grad_w"]).max()) <= 1e-4 * sc
    if name.endswith("lts"):
        np.testing.assert_allclose(thr.grad.numpy(), g["grad_thr"], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("which", ["sbm"])
def test_oracle_three_adam_steps_sbm(which):
    g = golden("train_step_sbm")
    m = O.OracleSBM(make_cfg(), [10] * 6, [0.05, 0.1, 0.2, 0.3, 0.5, 0.8])
    m.load_state_dict(sd_from(g, "sd0."))
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=5e-3)
    xs, ys = _t(g["xs"]), _t(g["ys"])
    for i in range(3):
        out, info = m(xs[i])
        loss = F.cross_entropy(out, ys[i]) + info.loss.mean()
        loss.backward()
        opt.step()
        opt.zero_grad()
        assert abs(loss.item() - g["losses"][i]) < 1e-5 * max(1.0, abs(g["losses"][i]))


@pytest.mark.parametrize("name,lts", [("sbm_ch_6x10", False), ("lts_ch_6x10", True)])
def test_oracle_sbm_and_lts_at_the_chisco_shape_with_the_6x10_bank(name, lts):
    g = golden(name)
    cfg = make_cfg(enc_in=122, seq_len=1000, num_class=3, c_out=3)
    torch.manual_seed(0)
    m = seeded_fill(O.OracleSBM(cfg, [10] * 6, [0.05, 0.1, 0.2, 0.3, 0.5, 0.8], lts=lts), int(g["seed"]))
    with torch.no_grad():
        for sh in m.shapelets:
            sh.weights.mul_(sh.weights.shape[-1] ** 0.5)
            if getattr(sh, "threshold", None) is not None:
                sh.threshold.abs_().mul_(10.0)
    m.train()
    out, info = m(_t(g["x"]))
    loss = F.cross_entropy(out, _t(g["y"])) + info.loss.mean()
    loss.backward()
    np.testing.assert_allclose(out.detach().numpy(), g["out"], **TOL)
    np.testing.assert_allclose(info.p.detach().numpy(), g["p"], **TOL)
    np.testing.assert_allclose(info.d.detach().numpy(), g["d"], **TOL)
    assert abs(loss.item() - float(g["train_loss"])) < 2e-5
    # gradients: small tensors and the fixed samples of the large ones; rows whose arg-max ties to the last bit are allowed to
    # differ (tests/test_gpu_golden_r2.py explains), so the sample check tolerates a handful of outliers
    gmax = max(float(np.abs(g[k]).max()) for k in g if k.startswith(("grad.", "gradsample.")))
    for k, p_ in m.named_parameters():
        if "grad." + k in g:
            assert float(np.abs(p_.grad.numpy() - g["grad." + k]).max()) <= 1e-4 * max(float(np.abs(g["grad." + k]).max()), 1e-4 * gmax), k
        else:
            idx = torch.linspace(0, p_.numel() - 1, 2048).long()
            bad = np.abs(p_.grad.flatten()[idx].numpy() - g["gradsample." + k]) > 1e-4 * max(float(np.abs(g["gradsample." + k]).max()), 1e-4 * gmax)
            assert bad.mean() <= 0.01, (k, bad.mean())
