"""GPU parity of the HIP shapelet path (through the C ABI) against the golden fixtures produced by the
reference and against the CPU oracle on seeded inputs.  Tolerance: 1e-4 (north_star: "within 1e-4 fp32")."""
import numpy as np
import pytest
import torch

from conftest import golden, make_cfg, parity, sd_from

pytestmark = pytest.mark.gpu

RTOL = ATOL = 1e-4


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _t(a, dev=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t.to(dev) if dev is not None else t


_N = [0]


def _label(msg, kind):
    _N[0] += 1
    return f"{kind}{_N[0]:03d}" + (f" {msg}" if msg else "")


def _close(a, b, rtol=RTOL, atol=ATOL, msg=""):
    """element-wise |a-b| <= tol (1 + |b|), recorded in the parity ledger"""
    assert rtol == atol
    parity(_label(msg, "val"), a, b, tol=rtol, kind="elem")


def _grad_close(a, b, msg="", tol=1e-4):
    """gradients: 1e-4 relative to the tensor's own scale (entries are sums of O(B*Tw) signed terms)."""
    parity(_label(msg, "grad"), a, b, tol=tol, kind="scale", floor=1e-12)


def test_library_loaded_is_in_tree():
    import speech_imagery_eeg_amd  # noqa
    from ign_hip import _lib
    assert _lib.lib_path().endswith("speech-imagery-eeg_amd/csrc/libign_hip.so")
    assert _lib.lib().ign_abi_version() == 1


def test_cpu_tensor_is_refused():
    import speech_imagery_eeg_amd  # noqa
    from ign_hip import ops, _lib
    with pytest.raises(_lib.IgnError):
        ops.instance_norm(torch.randn(2, 10, 3))


def test_instance_norm_matches_oracle():
    dev = _dev()
    import speech_imagery_eeg_amd  # noqa
    from ign_hip import ops
    from oracle import ign_oracle as O
    g = torch.Generator().manual_seed(1)
    for (B, T, C) in [(3, 100, 6), (2, 1000, 122), (5, 37, 17), (1, 64, 1)]:
        x = torch.randn(B, T, C, generator=g) * 3.0 + 50.0        # large offset: two-pass variance matters
        xn, xt = ops.instance_norm(x.to(dev), want_raw=True)
        _close(xn, O.instance_norm(x), msg=f"{B},{T},{C}")
        assert torch.equal(xt.cpu(), x.permute(0, 2, 1).contiguous())


@pytest.mark.parametrize("name,mode", [("l1", 0), ("mse", 1), ("cos", 2), ("pearson", 3)])
def test_shapelet_golden_small(name, mode):
    dev = _dev()
    import speech_imagery_eeg_amd  # noqa
    from ign_hip import ops
    g = golden(f"shapelet_{name}")
    xn, r = _t(g["xn"], dev), _t(g["r"], dev)
    w = _t(g["w"], dev).requires_grad_(True)
    p, dmin = ops.shapelet_bank(xn, [w], float(g["eps"]), mode)
    (p * r).sum().backward()
    _close(p, g["p"])
    _close(dmin, g["dmin"])
    _grad_close(w.grad, g["grad_w"], name)


def test_shapelet_golden_lts():
    dev = _dev()
    import speech_imagery_eeg_amd  # noqa
    from ign_hip import ops
    g = golden("shapelet_lts")
    xn, r = _t(g["xn"], dev), _t(g["r"], dev)
    w = _t(g["w"], dev).requires_grad_(True)
    thr = _t(g["thr"], dev).requires_grad_(True)
    p, dmin = ops.shapelet_bank(xn, [w], float(g["eps"]), ops.DIST_L1 | ops.GATE_LTS, thresholds=[thr])
    (p * r).sum().backward()
    _close(p, g["p"])
    _close(dmin, g["dmin"])
    _grad_close(w.grad, g["grad_w"], "lts w")
    _grad_close(thr.grad, g["grad_thr"], "lts thr")


@pytest.mark.parametrize("name", ["l1", "lts"])
def test_exact_ties_differ_from_sgn0_by_exactly_the_documented_term(name):
    """x == w exactly.  The reference's gradient uses sign(0) = 0 (aten::sgn behind abs, IGN/model/Shapelet.py:74); the L1
    backward kernel accumulates P = sum_{x > w} A and forms 2P - S, i.e. it counts a tie as sign = -1 (include/ign_abi.h,
    ign_shapelet_bwd).  So   grad_kernel[k,c,j] = grad_ref[k,c,j] - sum_{(b,t): x[b,c,t+j] == w[k,c,j]} A[b,k,c,t],
    A = -(dl/dd[b,t,k,c]) / L.  The fixture (reference run) plants one perfect-match shapelet and three single-element
    ties; the tie term is evaluated from the oracle's dl/dd (the oracle is pinned to the same fixture) and the kernel must
    equal reference-minus-tie-term at 1e-4 -- no other difference is allowed, forward values included."""
    dev = _dev()
    import speech_imagery_eeg_amd  # noqa
    from ign_hip import ops
    from oracle import ign_oracle as O
    g = golden(f"shapelet_tie_{name}")
    xn_c, w_c, r_c = _t(g["xn"]), _t(g["w"]), _t(g["r"])
    lts = name == "lts"
    # the tie term from the oracle
    wo = w_c.clone().requires_grad_(True)
    d = O.window_distance(xn_c, wo, 1, O.MODE_L1, chunk=7)
    d.retain_grad()
    if lts:
        po, _ = O.lts_softmin_gate(d, _t(g["thr"]))
    else:
        po, _ = O.rbf_straight_through_max(d, float(g["eps"]))
    (po * r_c).sum().backward()
    B, C, T = xn_c.shape
    K, _, L = w_c.shape
    Tw = T - L + 1
    A = -d.grad / L                                                   # (B, Tw, K, C)
    win = xn_c.unfold(2, L, 1)                                        # (B, C, Tw, L)
    tie = (win.permute(0, 2, 1, 3).unsqueeze(2) == w_c.view(1, 1, K, C, L))     # (B, Tw, K, C, L)
    delta = (A.unsqueeze(-1) * tie).sum(dim=(0, 1))                   # (K, C, L)
    assert int(tie.sum()) >= 3 + C * L
    if lts:
        assert float(delta[0].abs().max()) > 1e-3 * float(np.abs(g["grad_w"]).max())     # the perfect match carries gradient
    assert float(delta[1:].abs().max()) > 1e-4 * float(np.abs(g["grad_w"]).max())       # the test has teeth
    # the kernel
    xn, r = xn_c.to(dev), r_c.to(dev)
    w = w_c.to(dev).requires_grad_(True)
    kw = dict(thresholds=[_t(g["thr"], dev).requires_grad_(True)]) if lts else {}
    p, dmin = ops.shapelet_bank(xn, [w], float(g["eps"]), ops.DIST_L1 | (ops.GATE_LTS if lts else ops.GATE_RBF), **kw)
    (p * r).sum().backward()
    _close(p, g["p"], msg=f"tie {name} p")
    _close(dmin, g["dmin"], msg=f"tie {name} dmin")
    _grad_close(w.grad, g["grad_w"] - delta.numpy(), f"tie {name}: reference gradient minus the documented tie term")
    if lts:
        _grad_close(kw["thresholds"][0].grad, g["grad_thr"], f"tie {name} thr")


def test_shapelet_bm_groups_one_bank():
    """The four driver-default groups of the BasicMotions shape in ONE bank call (column offsets, K=5 tiles)."""
    dev = _dev()
    import speech_imagery_eeg_amd  # noqa
    from ign_hip import ops
    g = golden("shapelet_bm")
    xn, _ = ops.instance_norm(_t(g["x"], dev))
    _close(xn, g["xn"])
    ws = [_t(g[f"w{i}"], dev).requires_grad_(True) for i in range(4)]
    p, dmin = ops.shapelet_bank(xn, ws, 1.0)
    r = torch.cat([_t(g[f"r{i}"], dev) for i in range(4)], dim=1)
    (p * r).sum().backward()
    _close(p, np.concatenate([g[f"p{i}"] for i in range(4)], axis=1))
    _close(dmin, np.concatenate([g[f"dmin{i}"] for i in range(4)], axis=1))
    for i in range(4):
        _grad_close(ws[i].grad, g[f"grad_w{i}"], f"group {i}")


@pytest.mark.parametrize("B,C,T,K,L,mode", [
    (4, 122, 1000, 5, 100, 0),     # CHISCO shape, TT=15
    (3, 122, 1000, 5, 500, 0),     # TT=8, JJ=8 backward
    (3, 7, 1000, 10, 300, 0),      # two K-tiles of 5
    (5, 3, 200, 7, 33, 0),         # K = 5 + 2 tiles, odd L (tail loop)
    (2, 4, 1300, 3, 10, 0),        # Tw = 1291 > 1024: two passes, K = 2 + 1 tiles
    (3, 5, 300, 4, 50, 1),         # MSE
    (3, 5, 300, 5, 50, 2),         # cosine
    (2, 122, 1000, 5, 200, 2),     # cosine, CHISCO shape
    (3, 5, 300, 7, 33, 3),         # pearson, K = 5 + 2 tiles, odd L
    (2, 16, 1000, 5, 500, 3),      # pearson, long shapelets
    (3, 5, 120, 6, 3, 0),          # minimum shapelet length
    (2, 3, 50, 2, 50, 0),          # L == T: a single window
])
def test_shapelet_vs_oracle_random(B, C, T, K, L, mode):
    dev = _dev()
    import speech_imagery_eeg_amd  # noqa
    from ign_hip import ops
    from oracle import ign_oracle as O
    g = torch.Generator().manual_seed(B * 1000 + L)
    x = torch.randn(B, T, C, generator=g)
    w0 = torch.randn(K, C, L, generator=g)
    r = torch.randn(B, K * C, generator=g)
    xn_o = O.instance_norm(x)
    wo = w0.clone().requires_grad_(True)
    po, do = O.rbf_straight_through_max(O.window_distance(xn_o, wo, 1, mode, chunk=8), 0.8)
    (po * r).sum().backward()
    xn, _ = ops.instance_norm(x.to(dev))
    wg = w0.clone().to(dev).requires_grad_(True)
    p, dmin = ops.shapelet_bank(xn, [wg], 0.8, mode)
    (p * r.to(dev)).sum().backward()
    _close(p, po)
    _close(dmin, do)
    _grad_close(wg.grad, wo.grad.numpy(), f"B{B} C{C} T{T} K{K} L{L}")


@pytest.mark.parametrize("B,C,T,K,L,mode", [
    (4, 122, 1000, 5, 100, 0),     # CHISCO row, TT=15
    (3, 122, 1000, 10, 500, 0),    # two K-tiles, JJ=8 backward
    (5, 3, 200, 7, 33, 0),         # K = 5 + 2 tiles, odd L
    (2, 4, 1300, 3, 10, 0),        # two passes per row: the soft-min statistics are merged across passes
    (3, 5, 300, 4, 50, 1),         # LTS over the MSE distance
    (3, 5, 300, 5, 50, 2),         # ... cosine
    (3, 5, 120, 6, 3, 0),          # minimum shapelet length
    (2, 3, 50, 2, 50, 0),          # a single window: soft-min weight 1
])
def test_lts_gate_vs_oracle_random(B, C, T, K, L, mode):
    """The LTS soft-min / threshold gate (IGN/model/Shapelet.py:96-111) over the same seeded shape sweep the RBF gate gets:
    p, d_min, grad_w and grad_threshold against the oracle."""
    dev = _dev()
    import speech_imagery_eeg_amd  # noqa
    from ign_hip import ops
    from oracle import ign_oracle as O
    g = torch.Generator().manual_seed(B * 1000 + L + 7)
    x = torch.randn(B, T, C, generator=g)
    w0 = torch.randn(K, C, L, generator=g)
    t0 = torch.rand(1, K, C, generator=g) * 2.0
    r = torch.randn(B, K * C, generator=g)
    xn_o = O.instance_norm(x)
    wo, to = w0.clone().requires_grad_(True), t0.clone().requires_grad_(True)
    po, do = O.lts_softmin_gate(O.window_distance(xn_o, wo, 1, mode, chunk=8), to)
    (po * r).sum().backward()
    xn, _ = ops.instance_norm(x.to(dev))
    wg, tg = w0.clone().to(dev).requires_grad_(True), t0.clone().to(dev).requires_grad_(True)
    p, dmin = ops.shapelet_bank(xn, [wg], 1.0, mode | ops.GATE_LTS, thresholds=[tg])
    (p * r.to(dev)).sum().backward()
    _close(p, po, msg="p")
    _close(dmin, do, msg="dmin")
    _grad_close(wg.grad, wo.grad.numpy(), f"w B{B} C{C} T{T} K{K} L{L} mode{mode}")
    _grad_close(tg.grad, to.grad.numpy(), "threshold")


def test_strided_windows_forward():
    """seq_len >= 3000 switches the reference to stride int(log2(L)) (Shapelet.py:162): TT=1 kernel path."""
    dev = _dev()
    import speech_imagery_eeg_amd  # noqa
    from ign_hip import ops
    from oracle import ign_oracle as O
    g = torch.Generator().manual_seed(9)
    xn = torch.randn(2, 3, 3100, generator=g)
    w = torch.randn(4, 3, 310, generator=g)
    stride = 8
    po, do = O.rbf_straight_through_max(O.window_distance(xn, w, stride, 0, chunk=32), 1.0)
    with torch.no_grad():
        p, dmin = ops.shapelet_bank(xn.to(dev), [w.to(dev)], 1.0, 0, strides=[stride])
    _close(p, po)
    _close(dmin, do)


@pytest.mark.parametrize("B,C,T,K,L,stride,mode", [
    (2, 3, 3100, 4, 310, 8, 0),          # MotorImagery-like: T >= 3000 -> stride int(log2 L) (Shapelet.py:162)
    (2, 3, 3100, 7, 155, 7, 0),          # K = 7: shapelet tiles of the strided kernel, odd stride
    (2, 2, 3000, 3, 2400, 11, 0),        # 0.8 T: L > 2048 -> two j-tiles per shapelet
    (1, 2, 17984, 2, 1799, 10, 0),       # EigenWorms row length: the forward stages 72 KB of LDS per row
    (1, 2, 17984, 2, 8992, 13, 0),       # ... and its 0.5 T shapelets: five j-tiles
    (2, 3, 3100, 3, 310, 8, 1),          # MSE
    (2, 3, 3100, 3, 310, 8, 2),          # cosine
    (2, 3, 3100, 3, 310, 8, 3),          # pearson
])
def test_strided_and_long_rows_forward_backward_vs_oracle(B, C, T, K, L, stride, mode):
    """seq_len >= 3000: strided windows, forward AND backward (run_uea.sh's MotorImagery / EigenWorms cannot train
    without it), including rows longer than the 64 KB default LDS window and shapelets longer than one block of lanes."""
    dev = _dev()
    import speech_imagery_eeg_amd  # noqa
    from ign_hip import ops
    from oracle import ign_oracle as O
    g = torch.Generator().manual_seed(T + L + mode)
    xn = torch.randn(B, C, T, generator=g)
    w0 = torch.randn(K, C, L, generator=g)
    r = torch.randn(B, K * C, generator=g)
    wo = w0.clone().requires_grad_(True)
    po, do = O.rbf_straight_through_max(O.window_distance(xn, wo, stride, mode, chunk=32), 1.0)
    (po * r).sum().backward()
    wg = w0.clone().to(dev).requires_grad_(True)
    p, dmin = ops.shapelet_bank(xn.to(dev), [wg], 1.0, mode, strides=[stride])
    (p * r.to(dev)).sum().backward()
    _close(p, po, msg="p")
    _close(dmin, do, msg="dmin")
    _grad_close(wg.grad, wo.grad.numpy(), f"T{T} L{L} stride{stride} mode{mode}")


def test_sbm_trains_at_seq_len_3100():
    """The whole SBM at seq_len 3100 (stride rule applied by the model itself): forward, loss, backward against the oracle."""
    dev = _dev()
    import speech_imagery_eeg_amd  # noqa
    import torch.nn.functional as F
    from models.Shapelet import ShapeBottleneckModel
    from oracle import ign_oracle as O
    cfg = make_cfg(enc_in=3, seq_len=3100, num_class=3)
    torch.manual_seed(0)
    ref = O.OracleSBM(cfg, [3, 3], [0.05, 0.1])
    m = ShapeBottleneckModel(cfg, [3, 3], [0.05, 0.1])
    m.load_state_dict(ref.state_dict())
    m.to(dev)
    g = torch.Generator().manual_seed(5)
    x, y = torch.randn(3, 3100, 3, generator=g), torch.tensor([0, 1, 2])
    o_r, i_r = ref(x)
    (F.cross_entropy(o_r, y) + i_r.loss.mean()).backward()
    o, i = m(x.to(dev))
    (F.cross_entropy(o, y.to(dev)) + i.loss.mean()).backward()
    _close(o, o_r.detach(), msg="out")
    _close(i.p, i_r.p.detach(), msg="p")
    for (n, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
        _grad_close(p.grad, q.grad.numpy(), n)


@pytest.mark.parametrize("B,C,T,K,L,stride,lts", [(3, 6, 100, 5, 20, 1, False), (2, 4, 1300, 3, 10, 1, False),
                                                   (2, 3, 3100, 3, 310, 8, False), (3, 5, 120, 4, 30, 1, True)])
def test_match_location_is_the_argmax_window(B, C, T, K, L, stride, lts):
    """Tstar (ModelInfo.t): the window index the straight-through max selects -- arg-max_t p (first index on ties, like
    torch.argmax: IGN/model/Shapelet.py:79) for the RBF gate, arg-min_t d for LTS (:101) -- bit-exact against the oracle's
    distances; its sample range is what IGN/utils/shapelet_util.py:153 plots."""
    dev = _dev()
    import speech_imagery_eeg_amd  # noqa
    from ign_hip import ops
    from oracle import ign_oracle as O
    g = torch.Generator().manual_seed(B + T + L)
    xn = torch.randn(B, C, T, generator=g)
    w = torch.randn(K, C, L, generator=g)
    thr = torch.rand(1, K, C, generator=g)
    d = O.window_distance(xn, w, stride, 0, chunk=16)                       # (B, Tw, K, C)
    want = (d.argmin(dim=1) if lts else torch.exp(-d.pow(2)).argmax(dim=1)).reshape(B, K * C)
    mode = ops.DIST_L1 | (ops.GATE_LTS if lts else 0)
    with torch.no_grad():
        p, dmin, t = ops.shapelet_bank(xn.to(dev), [w.to(dev)], 1.0, mode, strides=[stride],
                                       thresholds=[thr.to(dev)] if lts else None, return_tstar=True)
    assert t.dtype == torch.int32 and t.shape == p.shape
    t = t.cpu().long()
    # a window whose distance ties the optimum to the last bit may be picked instead: compare through the distances
    picked = d.reshape(B, -1, K * C).gather(1, t.unsqueeze(1)).squeeze(1)
    best = d.reshape(B, -1, K * C).gather(1, want.unsqueeze(1)).squeeze(1)
    assert float((picked - best).abs().max()) <= 1e-6
    assert float((t != want).float().mean()) <= 0.01
    assert int(t.max()) * stride + L <= T


@pytest.mark.parametrize("B,C,T,Ks,Ls", [(5, 7, 1000, (5, 5, 5, 5), (100, 200, 300, 500)),      # the driver-default bank
                                         (3, 6, 100, (10, 5, 10), (5, 30, 80)),                    # K = 10: two tiles per row
                                         (2, 4, 200, (5, 3), (20, 50))])                            # K = 3: falls back per group
def test_bank_call_is_bitwise_the_per_group_result(B, C, T, Ks, Ls):
    """ign_shapelet_fwd_bank (all length groups in one ABI call, column offsets into shared outputs) must reproduce separate
    single-group banks bit for bit -- outputs, arg-max windows and, through the saved distances / statistics, the gradients."""
    dev = _dev()
    import speech_imagery_eeg_amd  # noqa
    from ign_hip import ops
    g = torch.Generator().manual_seed(sum(Ls))
    xn = torch.randn(B, C, T, generator=g).to(dev)
    ws = [torch.randn(K, C, L, generator=g).to(dev).requires_grad_(True) for K, L in zip(Ks, Ls)]
    r = torch.randn(B, sum(Ks) * C, generator=g).to(dev)
    p, d, t = ops.shapelet_bank(xn, ws, 0.9, return_tstar=True)
    gw = torch.autograd.grad((p * r).sum(), ws)
    col = 0
    for i, w in enumerate(ws):
        n = Ks[i] * C
        pi, di, ti = ops.shapelet_bank(xn, [w], 0.9, return_tstar=True)
        gi, = torch.autograd.grad((pi * r[:, col:col + n]).sum(), [w])
        assert torch.equal(pi, p[:, col:col + n]) and torch.equal(di, d[:, col:col + n]) and torch.equal(ti, t[:, col:col + n])
        assert torch.equal(gi, gw[i])
        col += n


def test_full_size_properties():
    """B=256, C=122, T=1000 (BASELINE config 1): size-independent checks the oracle cannot reach in seconds."""
    dev = _dev()
    import speech_imagery_eeg_amd  # noqa
    from ign_hip import ops
    torch.manual_seed(0)
    B, C, T, K = 256, 122, 1000, 5
    x = torch.randn(B, T, C, device=dev)
    xn, _ = ops.instance_norm(x)
    assert abs(float(xn.mean())) < 1e-5 and abs(float(xn.std(dim=-1, unbiased=True).mean()) - 1.0) < 1e-4
    ws = [torch.randn(K, C, L, device=dev, requires_grad=True) for L in (100, 200, 300, 500)]
    # plant shapelet 0 of group 1 at t=123 of row 7, channel 11: distance 0, p = 1, arg-max = 123
    with torch.no_grad():
        ws[1][0, 11] = xn[7, 11, 123:323]
    p, dmin = ops.shapelet_bank(xn, ws, 1.0)
    col = K * C + 0 * C + 11
    assert float(dmin[7, col]) == 0.0 and float(p[7, col]) == 1.0
    # RBF is monotone: max_t p == exp(-(eps * min_t d)^2)
    _close(p, torch.exp(-dmin.double() ** 2).float(), rtol=1e-5, atol=1e-5)
    assert bool(((p > 0) & (p <= 1)).all())
    g1 = torch.randn_like(p)
    g2 = torch.randn_like(p)
    grads1 = torch.autograd.grad(p, ws, g1, retain_graph=True)
    grads1b = torch.autograd.grad(p, ws, g1, retain_graph=True)
    grads2 = torch.autograd.grad(p, ws, g2, retain_graph=True)
    grads12 = torch.autograd.grad(p, ws, g1 + g2)
    for a, b in zip(grads1, grads1b):
        assert torch.equal(a, b), "backward is not bitwise reproducible"
    for a, b, c in zip(grads1, grads2, grads12):            # backward is linear in the upstream gradient
        scale = float(c.abs().max())
        assert float((a + b - c).abs().max()) <= 2e-4 * scale
    # batch-chunk additivity: grad over the batch == sum of grads over two halves
    p_a, _ = ops.shapelet_bank(xn[:128].contiguous(), ws, 1.0)
    p_b, _ = ops.shapelet_bank(xn[128:].contiguous(), ws, 1.0)
    assert torch.equal(torch.cat([p_a, p_b]), p)
    ga = torch.autograd.grad(p_a, ws, g1[:128].contiguous())
    gb = torch.autograd.grad(p_b, ws, g1[128:].contiguous())
    for a, b, c in zip(ga, gb, grads1):
        assert float((a + b - c).abs().max()) <= 2e-4 * float(c.abs().max())


def test_abi_argument_errors():
    dev = _dev()
    import ctypes
    import speech_imagery_eeg_amd  # noqa
    from ign_hip import _lib
    L = _lib.lib()
    x = torch.zeros(16, device=dev)
    pp = ctypes.c_void_p(x.data_ptr())
    assert L.ign_instnorm_fwd(None, pp, None, 1, 4, 4, 1e-8, None) == -1001
    assert L.ign_shapelet_fwd(pp, pp, None, pp, pp, 4, 0, pp, pp, None, None, 1, 2, 8, 2, 9, 1, 1.0, 0, None) == -1001  # L > T
    assert L.ign_shapelet_fwd(pp, pp, None, pp, pp, 4, 0, pp, pp, None, None, 1, 2, 8, 2, 3, 1, 1.0, 7, None) == -1001  # mode
    assert b"unknown mode" in L.ign_last_error()
    assert L.ign_shapelet_fwd(pp, pp, None, pp, pp, 4, 0, pp, pp, None, None, 1, 2, 8, 2, 3, 1, 1.0, 0x10, None) == -1001  # LTS, no thr
    assert L.ign_shapelet_bwd_workspace_bytes(1, 2, 8, 2, 3, 2, 0) == 1 * 2 * 2 * 3 * 4      # strided plans exist since round 2
    assert L.ign_shapelet_bwd_workspace_bytes(1, 2, 8, 2, 3, 0, 0) == 0


def test_abi_argument_errors_of_the_round3_entry_points():
    """Raw ctypes calls: every round-3 entry point refuses bad arguments with IGN_E_ARG / IGN_E_UNSUP before launching anything."""
    dev = _dev()
    import ctypes
    import speech_imagery_eeg_amd  # noqa
    from ign_hip import _lib
    L = _lib.lib()
    x = torch.zeros(4096, device=dev)
    pp = ctypes.c_void_p(x.data_ptr())
    v1, i1, l1 = ctypes.c_void_p * 1, ctypes.c_int * 1, ctypes.c_longlong * 1
    ARG, UNSUP = -1001, -1002
    # operand bounds
    assert L.ign_absmax(None, 16, pp, None) == ARG and L.ign_absmax(pp, 0, pp, None) == ARG
    assert L.ign_absmax(ctypes.c_void_p(x.data_ptr() + 4), 16, pp, None) == ARG                      # not 16-byte aligned
    assert L.ign_fcn_scan(0, v1(x.data_ptr()), l1(16), None, None, None, None, pp, None, 0, None) == ARG
    assert L.ign_fcn_scan(9, v1(x.data_ptr()), l1(16), None, None, None, None, pp, None, 0, None) == ARG
    assert L.ign_fcn_scan(1, v1(x.data_ptr()), l1(16), v1(x.data_ptr()), None, None, None, pp, None, 0, None) == ARG      # gamma without beta
    assert L.ign_fcn_scan(1, v1(x.data_ptr()), l1(16), v1(x.data_ptr()), v1(x.data_ptr()), i1(4), l1(1), pp, None, 0, None) == ARG   # R <= 1
    assert L.ign_fcn_scan(1, v1(x.data_ptr()), l1(16), None, None, None, None, pp, None, 5, None) == ARG         # nzero without a buffer
    assert L.ign_instnorm_fwd_amax(pp, pp, None, 1, 4, 4, ctypes.c_float(1e-8), None, None) == ARG             # null slot
    # fp16 GEMMs need both bounds
    assert L.ign_clconv_fwd_h3(pp, pp, None, None, None, pp, None, None, pp, 1, 16, 4, 4, 3, None) == ARG
    assert L.ign_clconv_dgrad_h3(pp, pp, pp, pp, pp, pp, pp, pp, pp, pp, None, 1, 16, 4, 4, 3, None) == ARG
    assert L.ign_clconv_wgrad_h3(pp, 2, pp, None, None, pp, pp, None, pp, 1, 16, 4, 4, 3, None) == ARG
    assert L.ign_linear_wgrad_h3(pp, pp, pp, None, pp, pp, pp, 0, 4, 4, None) == ARG
    assert L.ign_clconv_pack_weights_h2_multi(1, v1(x.data_ptr()), v1(x.data_ptr()), None, i1(4), i1(4), i1(3), None, None, None) == ARG
    assert L.ign_clconv_pack_weights_x3_multi(9, v1(x.data_ptr()), v1(x.data_ptr()), None, i1(4), i1(4), i1(3), None, None) == ARG
    assert L.ign_clconv_wgrad_reduce_multi(0, v1(x.data_ptr()), v1(x.data_ptr()), i1(1), i1(4), i1(4), i1(3), None) == ARG
    assert L.ign_clconv_wgrad_x6_nsplit(1, 16, 4, 4, 7) == 0                                       # k = 7 has no multi-tap kernel
    # attention on fp16 planes: E <= 64, all bounds present
    z = (pp,) * 5
    assert L.ign_attn_fwd_h3(*z, 1, 8, 8, 1, 128, 128, 128, 128, 128, 128, 128, 1.0, None, pp, pp, pp) == UNSUP
    assert L.ign_attn_fwd_h3(*z, 1, 8, 8, 1, 64, 64, 64, 64, 64, 64, 64, 1.0, None, pp, None, pp) == ARG
    assert L.ign_attn_bwd_h3(*((pp,) * 10), 1, 8, 8, 1, 64, 64, 64, 64, 64, 64, 64, 1.0, None, 0, 0, pp, pp, pp, None, None) == ARG
    # fused regulariser, bank backward, head backward with an addend
    assert L.ign_sbm_reg_fwd_bwd(pp, pp, 16, 0.1, 9, v1(x.data_ptr()), v1(x.data_ptr()), i1(2), i1(3), 2, 0.1, 1e-6, pp, pp, None) == ARG
    assert L.ign_sbm_reg_fwd_bwd(pp, pp, 16, 0.1, 1, v1(x.data_ptr()), v1(x.data_ptr()), i1(17), i1(3), 2, 0.1, 1e-6, pp, pp, None) == UNSUP
    assert L.ign_sbm_reg_fwd_bwd(pp, pp, 16, 0.1, 0, None, None, None, None, 2, 0.1, 1e-6, None, pp, None) == ARG          # no output
    assert L.ign_sbm_reg_workspace_bytes(9, 2, 16) == 0
    assert L.ign_shapelet_bwd_bank_workspace_bytes(0, 1, 2, 8, i1(2), i1(3), i1(1), 0) == 0
    assert L.ign_shapelet_bwd_bank(pp, 0, None, pp, pp, pp, 4, None, None, None, None, None, None, None, None, None, pp, 1, 2, 8, None,
                                   None, None, 1.0, 0, None) == ARG
    assert L.ign_head_bwd_acc(None, pp, pp, pp, pp, None, pp, pp, 1, 4, 2, 4, None) == ARG
    assert L.ign_loss_fwd_bwd_reg(pp, pp, pp, pp, pp, pp, pp, pp, pp, 4, 1, 1.0, None) == ARG                                # N < 2
    assert L.ign_edge_lagprod_fwd(pp, pp, 4, 16, 1, 0, None) == ARG and L.ign_edge_lagprod_fwd(pp, pp, 4, 16, 130, 0, None) == UNSUP


@pytest.mark.parametrize("bound", [0.0, float("inf"), float("nan"), 1e-30, 1e20])
def test_fp16_gemm_with_degenerate_magnitude_bounds(bound):
    """ign_pow2_scale: a bound of zero (an all-zero operand), a non-finite one, or one at the ends of the supported range
    (the scale exponent is clamped to +-60: magnitudes from ~1e-14, below which precision degrades gracefully towards "the operand
    is zero", up to 1.9e22 -- include/ign_abi.h) must not produce NaN / inf; the all-zero case returns the bias."""
    dev = _dev()
    import ctypes
    import speech_imagery_eeg_amd  # noqa
    from ign_hip import _lib
    L = _lib.lib()
    B, T, Ci, Co, k = 2, 40, 8, 128, 3
    g = torch.Generator().manual_seed(4)
    zero_in = bound == 0.0
    # (a bound must BE a bound: the finite non-zero cases use data a factor ~10 below it; the scale exponent is clamped to +-60,
    # so operands of 1e-31 underflow to zero -- the output is the bias -- instead of overflowing)
    import math
    fallback = bound == 0.0 or not math.isfinite(bound)          # scale 1
    mag = 1.0 if fallback else bound / 30.0
    x = torch.zeros(B, T, Ci) if zero_in else torch.randn(B, T, Ci, generator=g).clamp(-3, 3) * mag
    w = torch.randn(Co, Ci, k, generator=g) * 0.1
    bias = torch.randn(Co, generator=g)
    xd, wd, bd = x.to(dev), w.to(dev), bias.to(dev)
    wt = torch.zeros(int(L.ign_clconv_x3_elems(Co, Ci, k)), device=dev, dtype=torch.bfloat16)
    slots = torch.tensor([float(w.abs().max()), bound, 0.0, 0.0], device=dev)
    v1, i1 = ctypes.c_void_p * 1, ctypes.c_int * 1
    _lib.check(L.ign_clconv_pack_weights_h2_multi(1, v1(wd.data_ptr()), v1(wt.data_ptr()), None, i1(Co), i1(Ci), i1(k), None,
                                                  v1(slots.data_ptr()), None), "pack")
    y = torch.full((B, T - k + 1, Co), float("nan"), device=dev)
    _lib.check(L.ign_clconv_fwd_h3(ctypes.c_void_p(xd.data_ptr()), ctypes.c_void_p(wt.data_ptr()), ctypes.c_void_p(bd.data_ptr()), None,
                                   None, ctypes.c_void_p(y.data_ptr()), None, ctypes.c_void_p(slots.data_ptr() + 4),
                                   ctypes.c_void_p(slots.data_ptr()), B, T, Ci, Co, k, None), "fwd_h3")
    torch.cuda.synchronize()
    assert torch.isfinite(y).all()
    ref = torch.nn.functional.conv1d(x.double().permute(0, 2, 1), w.double(), bias.double()).permute(0, 2, 1)
    # unscaled (scale 1) operands of O(1) are inside fp16's range: full accuracy; the clamped / tiny cases keep the bias exact
    tol = 3e-6 if (fallback or bound == 1e20) else 1e-3
    assert float((y.double().cpu() - ref).abs().max() / ref.abs().max()) < tol


def test_instance_norm_takes_the_input_bound_in_the_same_pass():
    """ops.instance_norm(..., input_bound=True): outputs bitwise those of the plain pass, and the bound attached to the raw batch
    is exactly max |x| (what ign_absmax returns); a second call finds the bound attached and takes the plain route."""
    dev = _dev()
    from ign_hip import ops
    g = torch.Generator().manual_seed(3)
    for (B, T, C, sc) in [(5, 100, 6, 1.0), (3, 1000, 122, 37.5), (2, 64, 33, 1e-4)]:
        x = (torch.randn(B, T, C, generator=g) * sc).to(dev)
        xn0, _ = ops.instance_norm(x.clone())
        xn1, xt1 = ops.instance_norm(x, want_raw=True, input_bound=True)
        b = ops.cached_bound(x)
        assert b is not None and float(b) == float(x.abs().max())
        assert torch.equal(xn0, xn1) and torch.equal(xt1, x.permute(0, 2, 1))
        xn2, _ = ops.instance_norm(x, input_bound=True)
        assert ops.cached_bound(x) is b and torch.equal(xn2, xn0)
        x.mul_(2.0)                                         # modified in place: the attached bound no longer describes it
        assert ops.cached_bound(x) is None
