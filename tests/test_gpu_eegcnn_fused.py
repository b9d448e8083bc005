"""The fused EEG-CNN block ops (csrc/ign_eegcnn_fused.hip) against float64 torch of the reference's layer sequence
(IGN/model/eegcnn.py:71-74,79-82): channel contraction fwd / input gradient / weight gradient, and BatchNorm2d(batch statistics)
+ ELU + AvgPool as one op, including the affine map in front of it and the running-statistics update."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import parity

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


@pytest.mark.parametrize("B,Ci,Co,T", [(3, 122, 64, 1000), (2, 20, 12, 200), (4, 64, 64, 500), (1, 5, 3, 70), (2, 128, 64, 257)])
def test_chan_contract_vs_float64(B, Ci, Co, T):
    dev = _dev()
    import speech_imagery_eeg_amd  # noqa
    from ign_hip import ops
    g = torch.Generator().manual_seed(B + Ci + T)
    x = torch.randn(B, Ci, T, generator=g)
    w = torch.randn(Co, Ci, generator=g) / Ci ** 0.5
    gu = torch.randn(B, Co, T, generator=g)
    need_gx = Ci <= 64
    xd, wd = x.double().requires_grad_(need_gx), w.double().requires_grad_(True)
    ud = torch.einsum('oc,bct->bot', wd, xd)
    (ud * gu.double()).sum().backward()
    xg, wg = x.to(dev).requires_grad_(need_gx), w.to(dev).requires_grad_(True)
    u = ops.chan_contract(xg, wg)
    (u * gu.to(dev)).sum().backward()
    parity("u", u, ud, tol=1e-5, kind="scale", ref_is="float64 torch")
    parity("dW", wg.grad, wd.grad, tol=1e-5, kind="scale", ref_is="float64 torch")
    if need_gx:
        parity("dx", xg.grad, xd.grad, tol=1e-5, kind="scale", ref_is="float64 torch")
    u2 = ops.chan_contract(xg, wg)
    g1, = torch.autograd.grad((u2 * gu.to(dev)).sum(), wg)
    assert torch.equal(g1, wg.grad), "weight gradient not bitwise reproducible"


@pytest.mark.parametrize("B,C,T,P,affine", [(6, 64, 1000, 2, True), (5, 64, 500, 5, False), (3, 12, 201, 2, True), (4, 7, 33, 5, False),
                                            (2, 3, 10, 1, True)])
def test_bn_elu_pool_vs_float64(B, C, T, P, affine):
    dev = _dev()
    import speech_imagery_eeg_amd  # noqa
    from ign_hip import ops
    g = torch.Generator().manual_seed(B * 7 + C + T)
    v = torch.randn(B, C, T, generator=g) * 1.7 + 0.3
    alpha = (torch.rand(C, generator=g) + 0.5) * torch.where(torch.rand(C, generator=g) > 0.3, 1.0, -1.0) if affine else None
    cshift = torch.randn(C, generator=g) if affine else None
    gout = torch.randn(B, C, T // P, generator=g)
    bn = torch.nn.BatchNorm2d(C)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5, generator=g); bn.bias.uniform_(-0.5, 0.5, generator=g)
        bn.running_mean.normal_(generator=g); bn.running_var.uniform_(0.5, 2.0, generator=g)
    import copy
    bn_ref = copy.deepcopy(bn).double()
    bng = copy.deepcopy(bn).to(dev)
    for training in (True, False):
        bng.train(training); bn_ref.train(training)
        # float64 reference: the reference's own layer sequence
        vd = v.double().requires_grad_(True)
        ad = alpha.double().requires_grad_(True) if affine else None
        cd = cshift.double().requires_grad_(True) if affine else None
        y = vd * ad.view(1, -1, 1) + cd.view(1, -1, 1) if affine else vd
        for p_ in bn_ref.parameters():
            p_.grad = None
        outd = F.avg_pool2d(F.elu(bn_ref(y.unsqueeze(2))), (1, P)).squeeze(2)
        (outd * gout.double()).sum().backward()
        # HIP
        bng.weight.grad = bng.bias.grad = None
        vg = v.to(dev).requires_grad_(True)
        ag = alpha.to(dev).requires_grad_(True) if affine else None
        cg = cshift.to(dev).requires_grad_(True) if affine else None
        out = ops.bn_elu_pool(vg, bng, P, alpha=ag, cshift=cg)
        (out * gout.to(dev)).sum().backward()
        tag = "train" if training else "eval"
        parity(f"{tag} out", out, outd, tol=2e-5, kind="scale", ref_is="float64 torch")
        parity(f"{tag} dv", vg.grad, vd.grad, tol=5e-5, kind="scale", ref_is="float64 torch")
        parity(f"{tag} dgamma", bng.weight.grad, bn_ref.weight.grad, tol=5e-5, kind="scale", ref_is="float64 torch")
        parity(f"{tag} dbeta", bng.bias.grad, bn_ref.bias.grad, tol=5e-5, kind="scale", ref_is="float64 torch")
        if affine:
            gmax = float(bn_ref.weight.grad.abs().max())
            parity(f"{tag} dalpha", ag.grad, ad.grad, tol=5e-5, kind="scale", floor=1e-3 * gmax, ref_is="float64 torch")
            if training:
                assert cg.grad is None or float(cg.grad.abs().max()) == 0.0       # exactly zero: c cancels in the normalisation
                assert float(cd.grad.abs().max()) < 1e-9 * max(1.0, gmax)
            else:
                parity(f"{tag} dc", cg.grad, cd.grad, tol=5e-5, kind="scale", ref_is="float64 torch")
        if training:
            parity("running_mean", bng.running_mean, bn_ref.running_mean, tol=1e-5, kind="elem", ref_is="float64 torch")
            parity("running_var", bng.running_var, bn_ref.running_var, tol=1e-5, kind="elem", ref_is="float64 torch")
            assert int(bng.num_batches_tracked) == int(bn_ref.num_batches_tracked)
