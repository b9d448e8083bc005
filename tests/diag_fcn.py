"""Diagnostic (not a test): per-parameter and per-intermediate error of fcn_body vs float64 torch."""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import speech_imagery_eeg_amd  # noqa
from ign_hip import fcn
from test_gpu_fcn import _ref_blocks, _rel
dev = torch.device("cuda:0")
for (B, T, C) in [(2, 1000, 122), (2, 1000, 128)]:
    widths, ks = (128, 256, 128), (8, 5, 3)
    ref = _ref_blocks(C, widths, ks, seed=B + T + C).double().train()
    mod = copy.deepcopy(ref).float().to(dev).train()
    g = torch.Generator().manual_seed(11 * T + C)
    x = torch.randn(B, T, C, generator=g); gp = torch.randn(B, widths[-1], generator=g)
    cap = {}
    h = x.double().permute(0, 2, 1)
    for l, blk in enumerate(ref):
        y = blk[0](h); y.retain_grad(); cap[f"y{l}"] = y
        n = blk[1](y); n.retain_grad(); cap[f"n{l}"] = n
        h = blk[2](n)
    p = h.mean(-1); (p * gp.double()).sum().backward()
    fcn.DEBUG = {}
    q = fcn.fcn_body(x.to(dev), [(b[0], b[1]) for b in mod]); (q * gp.to(dev)).sum().backward()
    print((B, T, C), "pooled", f"{_rel(q, p):.2e}")
    for l in (2, 1, 0):
        gref = cap[f"n{l}"].grad.permute(0, 2, 1)       # dL/d(bn out) = masked g
        dyref = cap[f"y{l}"].grad.permute(0, 2, 1)
        gm = fcn.DEBUG[f"g{l}"].double().cpu()
        pad = (ks[l] - 1) if l > 0 else 0
        dy = fcn.DEBUG[f"dyp{l}"].double().cpu()
        dyi = dy[:, pad:dy.shape[1] - pad]
        err = (gm - gref).abs()
        bad = (err > 1e-4 * gref.abs().max()).nonzero()
        print(f"  block{l}: g rel {_rel(gm, gref):.1e} bad {len(bad)} of {gm.numel()}  dy rel {_rel(dyi, dyref):.1e}"
              f"  pads zero {float(dy[:, :pad].abs().max()) if pad else 0.0:.1e} {float(dy[:, dy.shape[1]-pad:].abs().max()) if pad else 0.0:.1e}"
              f"  dbeta {_rel(fcn.DEBUG[f'dbeta{l}'], ref[l][1].bias.grad):.1e} dgamma {_rel(fcn.DEBUG[f'dgamma{l}'], ref[l][1].weight.grad):.1e}")
        if len(bad):
            print("   first bad (b,t,c):", bad[:8].tolist(), " rows:", sorted(set((int(b_) * gm.shape[1] + int(t_)) for b_, t_, _ in bad.tolist()))[:20])
            b_, t_, c_ = bad[0].tolist()
            print("   values", float(gm[b_, t_, c_]), float(gref[b_, t_, c_]))
    fcn.DEBUG = None
