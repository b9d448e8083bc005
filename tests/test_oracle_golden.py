"""Pin the CPU oracle (oracle/ign_oracle.py) against outputs of the reference itself
(tests/golden/*.npz, produced by tests/golden/make_golden.py in the build container).
Tolerance: 1e-5 abs/rel fp32 (same torch, different summation order inside the chunks)."""
import numpy as np
import pytest
import torch

from conftest import golden, make_cfg, sd_from
from oracle import ign_oracle as O

TOL = dict(rtol=1e-5, atol=1e-5)


def _t(a):
    return torch.from_numpy(np.asarray(a))


@pytest.mark.parametrize("name,mode", [("l1", O.MODE_L1), ("mse", O.MODE_MSE), ("cos", O.MODE_COS),
                                       ("pearson", O.MODE_PEARSON)])
def test_shapelet_modes(name, mode):
    g = golden(f"shapelet_{name}")
    xn, w, r = _t(g["xn"]), _t(g["w"]).requires_grad_(True), _t(g["r"])
    d = O.window_distance(xn, w, 1, mode, chunk=7)
    p, dmin = O.rbf_straight_through_max(d, float(g["eps"]))
    (p * r).sum().backward()
    np.testing.assert_allclose(p.detach().numpy(), g["p"], **TOL)
    np.testing.assert_allclose(dmin.detach().numpy(), g["dmin"], **TOL)
    np.testing.assert_allclose(w.grad.numpy(), g["grad_w"], **TOL)


def test_shapelet_lts():
    g = golden("shapelet_lts")
    xn, w, r = _t(g["xn"]), _t(g["w"]).requires_grad_(True), _t(g["r"])
    thr = _t(g["thr"]).requires_grad_(True)
    d = O.window_distance(xn, w, 1, O.MODE_L1, chunk=5)
    p, dmin = O.lts_softmin_gate(d, thr)
    (p * r).sum().backward()
    np.testing.assert_allclose(p.detach().numpy(), g["p"], **TOL)
    np.testing.assert_allclose(dmin.detach().numpy(), g["dmin"], **TOL)
    np.testing.assert_allclose(w.grad.numpy(), g["grad_w"], **TOL)
    np.testing.assert_allclose(thr.grad.numpy(), g["grad_thr"], **TOL)


@pytest.mark.parametrize("name", ["l1", "lts"])
def test_shapelet_exact_ties_follow_sgn0(name):
    """x == w exactly (a shapelet planted from the input, single planted elements): the reference's autograd uses
    aten::sgn, sign(0) = 0 (IGN/model/Shapelet.py:74) -- so does the oracle."""
    g = golden(f"shapelet_tie_{name}")
    xn, w, r = _t(g["xn"]), _t(g["w"]).requires_grad_(True), _t(g["r"])
    assert torch.equal(w[0].detach(), xn[1, :, 20:29]) and float(w[1, 2, 4]) == float(xn[0, 2, 13])
    d = O.window_distance(xn, w, 1, O.MODE_L1, chunk=7)
    if name == "lts":
        thr = _t(g["thr"]).requires_grad_(True)
        p, dmin = O.lts_softmin_gate(d, thr)
    else:
        p, dmin = O.rbf_straight_through_max(d, float(g["eps"]))
    (p * r).sum().backward()
    np.testing.assert_allclose(p.detach().numpy(), g["p"], **TOL)
    np.testing.assert_allclose(dmin.detach().numpy(), g["dmin"], **TOL)
    np.testing.assert_allclose(w.grad.numpy(), g["grad_w"], **TOL)
    if name == "lts":
        np.testing.assert_allclose(thr.grad.numpy(), g["grad_thr"], **TOL)


def test_shapelet_bm_groups_and_instance_norm():
    g = golden("shapelet_bm")
    xn = O.instance_norm(_t(g["x"]))
    np.testing.assert_allclose(xn.numpy(), g["xn"], **TOL)
    for gi in range(4):
        w = _t(g[f"w{gi}"]).requires_grad_(True)
        d = O.window_distance(xn, w, 1, O.MODE_L1, chunk=16)
        p, dmin = O.rbf_straight_through_max(d, 1.0)
        (p * _t(g[f"r{gi}"])).sum().backward()
        np.testing.assert_allclose(p.detach().numpy(), g[f"p{gi}"], **TOL)
        np.testing.assert_allclose(dmin.detach().numpy(), g[f"dmin{gi}"], **TOL)
        np.testing.assert_allclose(w.grad.numpy(), g[f"grad_w{gi}"], **TOL)


def test_closed_form_backward_matches_autograd_fp64():
    """SURVEY App. A closed form (what the HIP backward implements) vs autograd, float64."""
    torch.manual_seed(3)
    B, C, T, K, L = 3, 2, 30, 4, 7
    xn = torch.randn(B, C, T, dtype=torch.float64)
    w = torch.randn(K, C, L, dtype=torch.float64, requires_grad=True)
    g = torch.randn(B, K * C, dtype=torch.float64)
    p, _ = O.rbf_straight_through_max(O.window_distance(xn, w, 1, O.MODE_L1, 4), 0.9)
    (p * g).sum().backward()
    gw = O.shapelet_l1_grad_w_closed_form(xn, w.detach(), 0.9, g)
    np.testing.assert_allclose(gw.numpy(), w.grad.numpy(), rtol=1e-12, atol=1e-14)


@pytest.mark.parametrize("name,sbm_cls,lts", [("sbm_linear", "linear", False), ("sbm_bilinear", "bilinear", False),
                                              ("sbm_attention", "attention", False), ("lts", "linear", True)])
def test_sbm_heads(name, sbm_cls, lts):
    g = golden(name)
    k = int(g["num_shapelet"])
    m = O.OracleSBM(make_cfg(sbm_cls=sbm_cls), [k] * 6, [0.05, 0.1, 0.2, 0.3, 0.5, 0.8], lts=lts)
    m.load_state_dict(sd_from(g))
    m.train()
    x, y = _t(g["x"]), _t(g["y"])
    out, info = m(x)
    loss = torch.nn.functional.cross_entropy(out, y) + info.loss.mean()
    loss.backward()
    np.testing.assert_allclose(out.detach().numpy(), g["out"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(info.p.detach().numpy(), g["p"], **TOL)
    np.testing.assert_allclose(info.d.detach().numpy(), g["d"], **TOL)
    np.testing.assert_allclose(info.loss.detach().numpy(), g["model_loss"], **TOL)
    np.testing.assert_allclose(loss.item(), g["train_loss"], **TOL)
    for n, p in m.named_parameters():
        np.testing.assert_allclose(p.grad.numpy(), g["grad." + n], rtol=1e-4, atol=2e-6, err_msg=n)


def _ign_from(g, cfg):
    m = O.OracleIGN(cfg)
    m.load_state_dict(sd_from(g))
    return m


def test_ign_fcn_bm():
    g = golden("ign_fcn_bm")
    m = _ign_from(g, make_cfg())
    x, y = _t(g["x"]), _t(g["y"])
    m.train()
    out, info = m(x, torch.ones(8, 100), None, None)
    loss = O.train_loss('InterpGN', out, info, y, beta=1.0)
    loss.backward()
    for k, v in (("out", out), ("eta", info.eta), ("shapelet_preds", info.shapelet_preds),
                 ("dnn_preds", info.dnn_preds), ("p", info.p), ("d", info.d), ("model_loss", info.loss)):
        np.testing.assert_allclose(v.detach().numpy(), g[k], rtol=1e-4, atol=2e-5, err_msg=k)
    assert info.loss.shape == (1,)
    np.testing.assert_allclose(loss.item(), g["train_loss"], rtol=1e-5)
    for n, p in m.named_parameters():
        if n.startswith("deep_model.block") and n.endswith(".0.bias"):
            continue
        np.testing.assert_allclose(p.grad.numpy(), g["grad." + n], rtol=2e-4, atol=2e-6, err_msg=n)
    for k, v in m.state_dict().items():      # BN running stats after the train-mode forward
        np.testing.assert_allclose(v.numpy(), g["sd_after." + k], rtol=1e-5, atol=1e-6, err_msg=k)
    m.eval()
    with torch.no_grad():
        oe, ie = m(x, None, None, None)
        og, ig = m(x, None, None, None, gating_value=0.05)
    np.testing.assert_allclose(oe.numpy(), g["eval_out"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(og.numpy(), g["gated_out"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(ig.eta.numpy(), g["gated_eta"], rtol=1e-4, atol=2e-5)


def test_ign_fcn_chisco_shape():
    g = golden("ign_fcn_ch")
    m = O.OracleIGN(make_cfg(enc_in=122, seq_len=1000, num_class=3, c_out=3), chunk=8)
    m.load_state_dict(sd_from(g))
    m.train()
    x, y = _t(g["x"]), _t(g["y"])
    out, info = m(x, None, None, None)
    loss = O.train_loss('InterpGN', out, info, y)
    loss.backward()
    np.testing.assert_allclose(out.detach().numpy(), g["out"], rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose(info.p.detach().numpy(), g["p"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(info.d.detach().numpy(), g["d"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(loss.item(), g["train_loss"], rtol=1e-5)
    for n, p in m.named_parameters():
        ref = g["grad." + n]
        if n.startswith("deep_model.block") and n.endswith(".0.bias"):
            continue     # conv bias in front of BatchNorm: the true gradient is 0, both sides hold rounding noise
        scale = max(1e-7, float(np.abs(ref).max()))
        assert np.abs(p.grad.numpy() - ref).max() <= 2e-4 * scale + 1e-9, n


@pytest.mark.parametrize("tag,posw", [("", False), ("_posw", True)])
def test_three_adam_steps(tag, posw):
    g = golden("train_step_ign" + tag)
    m = O.OracleIGN(make_cfg())
    m.load_state_dict(sd_from(g, "sd0."))
    opt = torch.optim.Adam(m.parameters(), lr=5e-3)
    xs, ys = _t(g["xs"]), _t(g["ys"])
    m.train()
    for i in range(3):
        out, info = m(xs[i], None, None, None)
        loss = O.train_loss('InterpGN', out, info, ys[i])
        loss.backward()
        opt.step()
        if posw:
            m.step()
        opt.zero_grad()
        np.testing.assert_allclose(loss.item(), g["losses"][i], rtol=2e-5)
    for k, v in m.state_dict().items():
        # Adam divides by sqrt(v)+eps: tiny grad differences are amplified on near-zero grads
        np.testing.assert_allclose(v.numpy(), g["sd3." + k], rtol=2e-3, atol=2e-4, err_msg=k)


# ----------------------------------------------------------------------------- remaining deep experts
def _leaf_params(g):
    p = {}
    for k, v in sd_from(g).items():
        p[k] = v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()
    return p


def _check_grads(p, g, tol=2e-4):
    gmax = max(float(np.abs(g[k]).max()) for k in g if k.startswith("grad."))
    for k in g:
        if k.startswith("grad."):
            ref = g[k]
            err = float(np.abs(p[k[5:]].grad.numpy() - ref).max())
            assert err <= tol * max(float(np.abs(ref).max()), 1e-3 * gmax) + 1e-8, f"{k}: {err:.3e}"


@pytest.mark.parametrize("tag", ["bm", "odd"])
def test_resnet_oracle(tag):
    from oracle import experts_oracle as X
    g = golden("resnet_" + tag)
    p = _leaf_params(g)
    x, y = torch.from_numpy(g["x"]), torch.from_numpy(g["y"])
    after = {}
    out = X.resnet_logits(p, x, training=True, stats_out=after)
    loss = torch.nn.functional.cross_entropy(out, y)
    loss.backward()
    np.testing.assert_allclose(out.detach().numpy(), g["out"], rtol=1e-4, atol=1e-5)
    assert abs(loss.item() - float(g["loss"])) < 1e-5
    _check_grads(p, g)
    for k in g:
        if k.startswith("sd_after.") and "num_batches" not in k:
            np.testing.assert_allclose(after[k[9:]].numpy(), g[k], rtol=1e-4, atol=1e-6, err_msg=k)
    with torch.no_grad():
        q = dict(p)
        q.update(after)
        np.testing.assert_allclose(X.resnet_logits(q, x, training=False).numpy(), g["eval_out"], rtol=1e-4, atol=1e-5)


def test_patchtst_oracle():
    from oracle import experts_oracle as X
    g = golden("patchtst_bm")
    p = _leaf_params(g)
    x, y = torch.from_numpy(g["x"]), torch.from_numpy(g["y"])
    out = X.patchtst_logits(p, x, n_layers=2, n_heads=4)
    loss = torch.nn.functional.cross_entropy(out, y)
    loss.backward()
    np.testing.assert_allclose(out.detach().numpy(), g["out"], rtol=1e-4, atol=1e-5)
    assert abs(loss.item() - float(g["loss"])) < 1e-5
    _check_grads(p, g)


def test_timesnet_oracle():
    from oracle import experts_oracle as X
    g = golden("timesnet_t96")
    p = _leaf_params(g)
    x, mask, y = torch.from_numpy(g["x"]), torch.from_numpy(g["mask"]), torch.from_numpy(g["y"])
    periods = []
    out = X.timesnet_logits(p, x, mask, n_layers=2, top_k=3, n_kernels=6, periods_out=periods)
    assert np.array_equal(np.array(periods), g["periods"])
    loss = torch.nn.functional.cross_entropy(out, y)
    loss.backward()
    np.testing.assert_allclose(out.detach().numpy(), g["out"], rtol=1e-4, atol=1e-5)
    assert abs(loss.item() - float(g["loss"])) < 1e-5
    _check_grads(p, g)
