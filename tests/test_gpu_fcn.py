"""GPU parity of the FCN expert's hand-written convolution stack (csrc/ign_clconv_{f32,x6}.hip, ign_bn.hip through the C ABI):
implicit-GEMM Conv1d forward / data gradient / weight gradient with the BatchNorm + ReLU prologues and epilogues,
against a float64 torch reference of IGN/model/FullyConvNet.py:31-59 on the CPU.  Tolerance 1e-4 (north_star)."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from conftest import make_cfg

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _rel(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _s():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize("B,T,Ci,Co,k,pro", [
    (4, 100, 6, 128, 8, False),      # BasicMotions first block: float2 staging, K = 48
    (3, 40, 3, 128, 3, False),       # odd channel count: scalar staging, K = 9 (one ragged chunk)
    (2, 50, 7, 64, 2, False),        # N < tile, odd K
    (2, 1000, 122, 128, 8, False),   # CHISCO first block
    (2, 993, 128, 256, 5, True),     # second block with the BatchNorm+ReLU prologue, two n-tiles
    (3, 77, 256, 128, 3, True),      # third block, ragged m-tiles
    (5, 9, 128, 256, 3, True),       # seq_len <= 10 kernels
])
@pytest.mark.parametrize("math", ["f32", "bf16x6", "f16x3", "f16x3 tiny", "f16x3 huge"])
def test_clconv_fwd_matches_conv1d(B, T, Ci, Co, k, pro, math):
    """"f16x3": two fp16 planes of power-of-two-scaled operands, three products (ign_clconv_fwd_h3) -- same bound as the other two;
    "tiny" / "huge": inputs of magnitude 1e-6 / 3e6 with a loose magnitude bound (4x): fp16's exponent range must not matter."""
    dev = _dev()
    import speech_imagery_eeg_amd  # noqa
    from ign_hip import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(B * 1000 + T + Ci + k)
    xs = 1e-6 if math.endswith("tiny") else 3e6 if math.endswith("huge") else 1.0
    math = math.split()[0]
    x = torch.randn(B, T, Ci, generator=g) * xs
    w = torch.randn(Co, Ci, k, generator=g) / (Ci * k) ** 0.5 / xs ** 0.5
    bias = torch.randn(Co, generator=g) * xs ** 0.5
    pa = (torch.rand(Ci, generator=g) + 0.5) if pro else None
    pb = torch.randn(Ci, generator=g) * 0.3 * xs if pro else None
    xin = torch.relu(x.double() * pa.double() + pb.double()) if pro else x.double()
    ref = F.conv1d(xin.permute(0, 2, 1), w.double(), bias.double()).permute(0, 2, 1)      # (B,Tout,Co)
    Tout = T - k + 1
    xd, wd_, bd = x.to(dev), w.to(dev), bias.to(dev)
    pad, pbd = (pa.to(dev), pb.to(dev)) if pro else (None, None)
    y = torch.full((B, Tout, Co), float("nan"), device=dev)
    nparts = int(L.ign_clconv_x6_mtiles(B, Tout) if math != "f32" else L.ign_clconv_mtiles(B * Tout))
    part = torch.full((nparts, 2, Co), float("nan"), device=dev)
    if math == "f32":
        wt = torch.empty(Co, k * Ci, device=dev)
        _lib.check(L.ign_clconv_pack_weights(_p(wd_), _p(wt), None, Co, Ci, k, _s()), "pack")
        _lib.check(L.ign_clconv_fwd(_p(xd), _p(wt), _p(bd), _p(pad), _p(pbd), _p(y), _p(part), B, T, Ci, Co, k, _s()), "fwd")
    elif math == "f16x3":
        import ctypes
        wt = torch.zeros(int(L.ign_clconv_x3_elems(Co, Ci, k)), device=dev, dtype=torch.bfloat16)
        slots = torch.tensor([float(w.abs().max()), 4.0 * float(xin.abs().max()), 0.0, 0.0], device=dev)     # |W| exact, |input| loose
        v1, i1 = ctypes.c_void_p * 1, ctypes.c_int * 1
        _lib.check(L.ign_clconv_pack_weights_h2_multi(1, v1(wd_.data_ptr()), v1(wt.data_ptr()), None, i1(Co), i1(Ci), i1(k), None,
                                                      v1(slots.data_ptr()), _s()), "pack_h2")
        _lib.check(L.ign_clconv_fwd_h3(_p(xd), _p(wt), _p(bd), _p(pad), _p(pbd), _p(y), _p(part), ctypes.c_void_p(slots.data_ptr() + 4),
                                       _p(slots), B, T, Ci, Co, k, _s()), "fwd_h3")
    else:       # split-bf16 product on the bf16 matrix cores: must be as accurate as the fp32 kernel
        wt = torch.empty(int(L.ign_clconv_x3_elems(Co, Ci, k)), device=dev, dtype=torch.bfloat16)
        _lib.check(L.ign_clconv_pack_weights_x3(_p(wd_), _p(wt), None, Co, Ci, k, _s()), "pack_x3")
        _lib.check(L.ign_clconv_fwd_x6(_p(xd), _p(wt), _p(bd), _p(pad), _p(pbd), _p(y), _p(part), B, T, Ci, Co, k, _s()), "fwd_x6")
    assert _rel(y, ref) < 3e-6, "fp32 rounding level (K <= 1280 terms)"
    # the BatchNorm statistics partials of the epilogue
    s1 = part[:, 0].double().sum(0).cpu()
    s2 = part[:, 1].double().sum(0).cpu()
    assert float((s1 - ref.sum((0, 1))).abs().max() / ref.abs().sum((0, 1)).max()) < TOL
    assert float((s2 - ref.square().sum((0, 1))).abs().max() / ref.square().sum((0, 1)).max()) < TOL


def _ref_blocks(cfg_in, widths, ks, seed):
    torch.manual_seed(seed)
    blocks, ci = [], cfg_in
    for co, k in zip(widths, ks):
        blocks.append(nn.Sequential(nn.Conv1d(ci, co, k), nn.BatchNorm1d(co), nn.ReLU()))
        ci = co
    for b in blocks:                    # non-trivial affine / running state
        with torch.no_grad():
            b[1].weight.uniform_(0.5, 1.5)
            b[1].bias.normal_(0, 0.3)
            b[1].running_mean.normal_(0, 0.2)
            b[1].running_var.uniform_(0.5, 2.0)
    return nn.Sequential(*blocks)


@pytest.mark.parametrize("B,T,C,widths,ks,training", [
    (8, 100, 6, (128, 256, 128), (8, 5, 3), True),       # BasicMotions
    (8, 100, 6, (128, 256, 128), (8, 5, 3), False),      # eval: running statistics
    (4, 37, 3, (64, 128, 64), (3, 3, 2), True),          # odd channels, ragged tiles, short kernels
    (2, 1000, 122, (128, 256, 128), (8, 5, 3), True),    # CHISCO shape (small batch)
    (3, 61, 9, (128, 256, 128), (8, 5, 3), False),
    (1, 140, 1, (128, 256, 128), (8, 5, 3), True),       # univariate series (several UEA sets), one sample
    (5, 30, 2, (128, 256, 128), (8, 5, 3), True),        # T barely above the kernel sizes
])
@pytest.mark.parametrize("math", ["f32", "bf16x6", "f16x3"])
def test_fcn_body_forward_backward(B, T, C, widths, ks, training, math, monkeypatch):
    dev = _dev()
    import copy
    import speech_imagery_eeg_amd  # noqa
    from ign_hip import fcn
    monkeypatch.setattr(fcn, "CONV_MATH", math)
    ref = _ref_blocks(C, widths, ks, seed=B + T + C).double().train(training)
    mod = copy.deepcopy(ref).float().to(dev).train(training)
    g = torch.Generator().manual_seed(11 * T + C)
    x = torch.randn(B, T, C, generator=g)
    gp = torch.randn(B, widths[-1], generator=g)

    # float64 reference, keeping the BatchNorm outputs: an element whose pre-activation is ~0 sits on the ReLU kink, where
    # fp32 rounding may legitimately pick the other side than float64 (one such element moves a B=2 gradient by ~1e-2).
    h = x.double().permute(0, 2, 1)
    near = []
    for blk in ref:
        n = blk[1](blk[0](h))
        near.append(int((n.abs() < 1e-5).sum()))
        h = blk[2](n)
    pooled_ref = h.mean(-1)
    (pooled_ref * gp.double()).sum().backward()

    pooled = fcn.fcn_body(x.to(dev), [(b[0], b[1]) for b in mod])
    (pooled * gp.to(dev)).sum().backward()
    assert _rel(pooled, pooled_ref) < TOL
    top_kink = max([l for l, c in enumerate(near) if c] or [-1])      # blocks <= top_kink see the kink elements
    for (n, p), (_, q) in zip(mod.named_parameters(), ref.named_parameters()):
        if training and n.endswith("0.bias"):
            # true gradient is zero (batch statistics remove the bias); both sides hold at most rounding noise
            assert float(p.grad.abs().max()) <= 1e-4 * max(1.0, float(q.grad.abs().max())), n
            continue
        tol = 2 * TOL if int(n.split(".")[0]) > top_kink else 5e-2
        assert _rel(p.grad, q.grad) < tol, (n, near)
    for (n, p), (_, q) in zip(mod.named_buffers(), ref.named_buffers()):
        if p.dtype.is_floating_point:
            assert _rel(p, q) < TOL, n
        else:
            assert int(p) == int(q), n


def _away_from_kink(y, a, b, margin=1e-3):
    """Move pre-activations that are within `margin` of the ReLU kink away from it (keeps the test well-conditioned)."""
    pre = y * a + b
    bad = pre.abs() < margin
    return torch.where(bad, y + 4 * margin / a, y)


@pytest.mark.parametrize("B,Tin,Ci,Co,k", [(2, 989, 256, 128, 3), (2, 993, 128, 256, 5), (3, 50, 64, 128, 3), (1, 9, 128, 256, 2),
                                           (2, 100, 120, 128, 8), (3, 77, 36, 64, 8)])
@pytest.mark.parametrize("math", ["f32", "bf16x6", "f16x3", "f16x3 tiny"])
def test_clconv_dgrad_and_wgrad_kernels(B, Tin, Ci, Co, k, math):
    """ign_clconv_dgrad (ReLU mask + BatchNorm-backward sums in the epilogue) and ign_clconv_wgrad (prologue recomputes
    relu(bn(y))) against float64 autograd of the same maths, with pre-activations kept off the ReLU kink."""
    dev = _dev()
    import speech_imagery_eeg_amd  # noqa
    from ign_hip import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(Tin + Ci + k)
    Tout = Tin - k + 1
    a = torch.rand(Ci, generator=g) + 0.5
    b = torch.randn(Ci, generator=g) * 0.3
    mean = torch.randn(Ci, generator=g) * 0.2
    invstd = torch.rand(Ci, generator=g) + 0.5
    y_in = _away_from_kink(torch.randn(B, Tin, Ci, generator=g), a, b)
    w = torch.randn(Co, Ci, k, generator=g) / (Ci * k) ** 0.5
    dy = torch.randn(B, Tout, Co, generator=g) * (1e-7 if math.endswith("tiny") else 1.0)     # "tiny": gradients of a late epoch
    math = math.split()[0]

    yd = y_in.double().requires_grad_(True)
    pre = yd * a.double() + b.double()
    z = torch.relu(pre)
    pre.retain_grad()
    wdbl = w.double().requires_grad_(True)
    out = F.conv1d(z.permute(0, 2, 1), wdbl).permute(0, 2, 1)
    (out * dy.double()).sum().backward()
    g_ref = pre.grad                                          # dL/dz masked by the ReLU
    yhat = (y_in.double() - mean.double()) * invstd.double()
    s1_ref, s2_ref = g_ref.sum((0, 1)), (g_ref * yhat).sum((0, 1))

    pad = k - 1
    dyp = torch.zeros(B, Tout + 2 * pad, Co)
    dyp[:, pad:pad + Tout] = dy
    to = lambda t: t.contiguous().to(dev)
    dypd, yind, ad, bd, md, isd, wd_ = to(dyp), to(y_in), to(a), to(b), to(mean), to(invstd), to(w)
    wt = torch.empty(Co, k * Ci, device=dev)
    wdg = torch.empty(Ci, k * Co, device=dev)
    _lib.check(L.ign_clconv_pack_weights(_p(wd_), _p(wt), _p(wdg), Co, Ci, k, _s()), "pack")
    gin = torch.full((B, Tin, Ci), float("nan"), device=dev)
    nparts = int(L.ign_clconv_x6_mtiles(B, Tin) if math != "f32" else L.ign_clconv_mtiles(B * Tin))
    part = torch.full((nparts, 2, Ci), float("nan"), device=dev)
    if math == "f32":
        _lib.check(L.ign_clconv_dgrad(_p(dypd), _p(wdg), _p(yind), _p(ad), _p(bd), _p(md), _p(isd), _p(gin), _p(part),
                                      B, Tin, Ci, Co, k, _s()), "dgrad")
    elif math == "f16x3":
        import ctypes
        wt3 = torch.zeros(int(L.ign_clconv_x3_elems(Co, Ci, k)), device=dev, dtype=torch.bfloat16)
        wd3 = torch.zeros(int(L.ign_clconv_x3_elems(Ci, Co, k)), device=dev, dtype=torch.bfloat16)
        # bounds: |W|, |relu(a y + b)| (a loose one), |dy| (taken by the producer kernel in the product; here from the tensor)
        slots = torch.tensor([float(w.abs().max()), 3.0 * float(z.detach().abs().max()), float(dy.abs().max()), 0.0], device=dev)
        v1, i1 = ctypes.c_void_p * 1, ctypes.c_int * 1
        sp = lambda i: ctypes.c_void_p(slots.data_ptr() + 4 * i)
        _lib.check(L.ign_clconv_pack_weights_h2_multi(1, v1(wd_.data_ptr()), v1(wt3.data_ptr()), v1(wd3.data_ptr()), i1(Co), i1(Ci), i1(k),
                                                      None, v1(slots.data_ptr()), _s()), "pack_h2")
        _lib.check(L.ign_clconv_dgrad_h3(_p(dypd), _p(wd3), _p(yind), _p(ad), _p(bd), _p(md), _p(isd), _p(gin), _p(part), sp(2), sp(0),
                                         B, Tin, Ci, Co, k, _s()), "dgrad_h3")
    else:
        wt3 = torch.empty(int(L.ign_clconv_x3_elems(Co, Ci, k)), device=dev, dtype=torch.bfloat16)
        wd3 = torch.empty(int(L.ign_clconv_x3_elems(Ci, Co, k)), device=dev, dtype=torch.bfloat16)
        _lib.check(L.ign_clconv_pack_weights_x3(_p(wd_), _p(wt3), _p(wd3), Co, Ci, k, _s()), "pack_x3")
        _lib.check(L.ign_clconv_dgrad_x6(_p(dypd), _p(wd3), _p(yind), _p(ad), _p(bd), _p(md), _p(isd), _p(gin), _p(part),
                                         B, Tin, Ci, Co, k, _s()), "dgrad_x6")
    assert _rel(gin, g_ref) < 3e-6
    assert _rel(part[:, 0].double().sum(0), s1_ref) < TOL and _rel(part[:, 1].double().sum(0), s2_ref) < TOL

    x6w = math != "f32" and k in (2, 3, 5, 8)
    nbytes = (L.ign_clconv_wgrad_x6_workspace_bytes if x6w else L.ign_clconv_wgrad_workspace_bytes)(B, Tin, Ci, Co, k)
    ws = torch.empty(int(nbytes) // 4, device=dev)
    dw = torch.full((Co, Ci, k), float("nan"), device=dev)
    if math == "f16x3" and x6w:
        _lib.check(L.ign_clconv_wgrad_h3(_p(dypd), pad, _p(yind), _p(ad), _p(bd), _p(dw), _p(ws), sp(2), sp(1), B, Tin, Ci, Co, k, _s()),
                   "wgrad_h3")
    else:
        wgrad = L.ign_clconv_wgrad_x6 if x6w else L.ign_clconv_wgrad
        _lib.check(wgrad(_p(dypd), pad, _p(yind), _p(ad), _p(bd), _p(dw), _p(ws), B, Tin, Ci, Co, k, _s()), "wgrad")
    assert _rel(dw, wdbl.grad) < 3e-6


def test_fcn_body_is_deterministic_and_matches_full_batch_shape():
    """Bitwise reproducibility at the benchmark shape (fixed-order reductions everywhere: no float atomics), plus a
    size-independent property: the pooled feature of a sample does not depend on which other samples share the batch
    when the statistics are frozen (eval mode)."""
    dev = _dev()
    import speech_imagery_eeg_amd  # noqa
    from ign_hip import fcn
    mod = _ref_blocks(122, (128, 256, 128), (8, 5, 3), seed=5).to(dev).train()
    x = torch.randn(16, 1000, 122, generator=torch.Generator().manual_seed(3)).to(dev)
    blocks = [(b[0], b[1]) for b in mod]
    outs, grads = [], []
    for _ in range(2):
        mod.zero_grad(set_to_none=True)
        p = fcn.fcn_body(x, blocks)
        p.square().sum().backward()
        outs.append(p.detach().clone())
        grads.append([q.grad.clone() for q in mod.parameters()])
    assert torch.equal(outs[0], outs[1])
    for a, b in zip(*grads):
        assert torch.equal(a, b)
    mod.eval()
    with torch.no_grad():
        full = fcn.fcn_body(x, blocks)
        half = fcn.fcn_body(x[5:9].contiguous(), blocks)
    assert _rel(half, full[5:9]) < 1e-6


@pytest.mark.parametrize("num_class,bias", [(4, True), (3, False), (40, True)], ids=["4 classes", "no head bias", "40 classes"])
def test_fcn_model_matches_its_own_modules_on_the_library_kernels(num_class, bias):
    """The drop-in FullyConvNetwork (blocks + pooling + class head as ONE autograd node, the head inside the pooling launch for up
    to 16 classes) against the SAME nn modules called one by one on torch's GPU kernels (MIOpen convolution / batch norm, rocBLAS
    head): logits, every parameter gradient, running statistics.  40 classes: the head falls back to its own launch."""
    dev = _dev()
    import speech_imagery_eeg_amd  # noqa
    import models.FullyConvNet as M
    cfg = make_cfg(enc_in=6, seq_len=100, num_class=num_class)
    torch.manual_seed(0)
    net = M.FullyConvNetwork(cfg)
    if not bias:
        net.fc = torch.nn.Linear(128, num_class, bias=False)
    net = net.to(dev).train()
    x = torch.randn(8, 100, 6, generator=torch.Generator().manual_seed(1)).to(dev)
    y = torch.arange(8, device=dev) % num_class
    res = {}
    for ours in (True, False):
        net.zero_grad(set_to_none=True)
        sd = {k: v.clone() for k, v in net.state_dict().items()}
        if ours:
            out = net(x)
        else:
            h = net.block3(net.block2(net.block1(x.permute(0, 2, 1))))
            out = net.fc(net.pooling(h).squeeze(-1))
        F.cross_entropy(out, y).backward()
        res[ours] = (out.detach().clone(), [p.grad.clone() for p in net.parameters()],
                     {k: v.clone() for k, v in net.state_dict().items()})
        net.load_state_dict(sd)
    assert _rel(res[True][0], res[False][0]) < TOL
    for (n, _), a, b in zip(net.named_parameters(), res[True][1], res[False][1]):
        if n.endswith("0.bias") and "fc" not in n:
            continue                                  # conv bias in front of BatchNorm: exactly zero here, rounding noise there
        assert _rel(a, b) < 5 * TOL, n
    for k in res[True][2]:
        if "running" in k:
            assert _rel(res[True][2][k], res[False][2][k]) < TOL, k


# --------------------------------------------------------------------------------- bf16-autocast mode
def _bf16r(t):
    return t.float().bfloat16().double()


@pytest.mark.parametrize("B,Tin,Ci,Co,k", [(3, 50, 6, 128, 8), (2, 140, 128, 256, 5), (2, 77, 256, 128, 3), (4, 33, 10, 64, 2),
                                           (2, 300, 64, 64, 1)])
def test_bf16_single_product_kernels_equal_rounded_operand_products(B, Tin, Ci, Co, k):
    """ign_clconv_{fwd,wgrad}_bf16 and the input gradient through ops: one product per MFMA step on operands rounded to bf16,
    fp32 accumulation == float64 convolution of the ROUNDED operands (bf16 x bf16 products are exact in fp32)."""
    dev = _dev()
    import speech_imagery_eeg_amd  # noqa
    from ign_hip import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(B + Tin + Ci + Co + k)
    x = torch.randn(B, Tin, Ci, generator=g)
    w = torch.randn(Co, Ci, k, generator=g) / (Ci * k) ** 0.5
    b = torch.randn(Co, generator=g)
    Tout = Tin - k + 1
    gy = torch.randn(B, Tout, Co, generator=g)
    xg, wg, bg, gyg = (t.to(dev).contiguous() for t in (x, w, b, gy))
    wt3 = torch.empty(int(L.ign_clconv_x3_elems(Co, Ci, k)), device=dev, dtype=torch.bfloat16)
    wd3 = torch.empty(int(L.ign_clconv_x3_elems(Ci, Co, k)), device=dev, dtype=torch.bfloat16)
    _lib.check(L.ign_clconv_pack_weights_x3(_p(wg), _p(wt3), _p(wd3), Co, Ci, k, _s()), "pack")
    y = torch.empty(B, Tout, Co, device=dev)
    _lib.check(L.ign_clconv_fwd_bf16(_p(xg), _p(wt3), _p(bg), None, None, _p(y), None, B, Tin, Ci, Co, k, _s()), "fwd_bf16")
    ref = F.conv1d(_bf16r(x).permute(0, 2, 1), _bf16r(w), b.double()).permute(0, 2, 1)
    assert _rel(y, ref) < 3e-6
    # input gradient = the same forward kernel on the zero-padded dy with the tap-reversed transposed weights
    gyp = F.pad(gyg, (0, 0, k - 1, k - 1)).contiguous()
    dx = torch.empty(B, Tin, Ci, device=dev)
    _lib.check(L.ign_clconv_fwd_bf16(_p(gyp), _p(wd3), None, None, None, _p(dx), None, B, Tout + 2 * (k - 1), Co, Ci, k, _s()), "dx")
    dref = F.conv_transpose1d(_bf16r(gy).permute(0, 2, 1), _bf16r(w)).permute(0, 2, 1)
    assert _rel(dx, dref) < 3e-6
    if k in (1, 2, 3, 5, 8) and Ci % 4 == 0 or k in (2, 3, 5, 8):
        ws = torch.empty(int(L.ign_clconv_wgrad_x6_workspace_bytes(B, Tin, Ci, Co, k)) // 4, device=dev)
        dw = torch.empty(Co, Ci, k, device=dev)
        _lib.check(L.ign_clconv_wgrad_bf16(_p(gyp), k - 1, _p(xg), None, None, _p(dw), _p(ws), B, Tin, Ci, Co, k, _s()), "wgrad_bf16")
        xr, gr = _bf16r(x), _bf16r(gy)
        wref = torch.stack([torch.einsum("bto,bti->oi", gr, xr[:, j:j + Tout]) for j in range(k)], dim=2)
        assert _rel(dw, wref) < 1e-5


class _RoundedConv(torch.autograd.Function):
    """float64 model of the single-product kernels: every GEMM operand (input, weight, upstream gradient) is rounded to bf16
    where the kernel stages it; products and sums are exact / float64."""

    @staticmethod
    def forward(ctx, x, w, b):
        xr, wr = _bf16r(x), _bf16r(w)
        ctx.save_for_backward(xr, wr)
        return F.conv1d(xr, wr, b)

    @staticmethod
    def backward(ctx, gy):
        xr, wr = ctx.saved_tensors
        gr = _bf16r(gy)
        return (F.conv_transpose1d(gr, wr), torch.nn.grad.conv1d_weight(xr, wr.shape, gr), gy.sum(dim=(0, 2)))


def test_fcn_body_inside_autocast_runs_the_bf16_kernels(monkeypatch):
    """Inside torch.autocast(bfloat16) the FCN body runs the single-product kernels: it matches a float64 model that rounds
    every GEMM operand to bf16 where the kernels do (tight bound), is NOT the fp32-accurate result, and stays within bf16-level
    distance of it; the output stays float32."""
    dev = _dev()
    import copy
    import speech_imagery_eeg_amd  # noqa
    from ign_hip import fcn
    ref = _ref_blocks(6, (128, 256, 128), (8, 5, 3), seed=5).train()
    m32, m16 = copy.deepcopy(ref).to(dev), copy.deepcopy(ref).to(dev)
    g = torch.Generator().manual_seed(17)
    x = torch.randn(8, 100, 6, generator=g)
    gp = torch.randn(8, 128, generator=g)
    p32 = fcn.fcn_body(x.to(dev), [(b[0], b[1]) for b in m32])
    with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
        p16 = fcn.fcn_body(x.to(dev), [(b[0], b[1]) for b in m16])
    assert p16.dtype == torch.float32
    (p16 * gp.to(dev)).sum().backward()
    assert 1e-5 < _rel(p16, p32) < 2e-2
    # float64 model with the kernels' rounding points
    md = copy.deepcopy(ref).double()
    h = x.double().permute(0, 2, 1)
    near = 0
    for blk in md:
        y = _RoundedConv.apply(h, blk[0].weight, blk[0].bias)
        n = F.batch_norm(y, None, None, blk[1].weight, blk[1].bias, True, 0.1, blk[1].eps)
        near += int((n.abs() < 1e-5).sum())
        h = F.relu(n)
    pd = h.mean(-1)
    (pd * gp.double()).sum().backward()
    # an fp32 activation within rounding of a bf16 tie rounds the other way than its float64 twin (2^-9 on that element):
    # the bound is a few 1e-4, two orders below the distance to the unrounded result
    assert _rel(p16, pd) < 3e-4
    if near == 0:
        for (n_, a), (_, b) in zip(m16.named_parameters(), md.named_parameters()):
            if n_.endswith("0.bias"):
                continue
            assert _rel(a.grad, b.grad) < 1e-2, n_
