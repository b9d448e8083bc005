"""Diagnostic (not a test): per-group device time of the shapelet forward / backward kernels at the benchmark shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import speech_imagery_eeg_amd  # noqa
from ign_hip import ops, _lib

dev = torch.device("cuda:0")
B, C, T, K = 256, 122, 1000, 5
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
torch.manual_seed(0)
xn, _ = ops.instance_norm(torch.randn(B, T, C, device=dev))
tot_f = tot_b = 0.0
for L in (100, 200, 300, 500):
    w = torch.randn(K, C, L, device=dev, requires_grad=True)
    p, d = ops.shapelet_bank(xn, [w], 1.0)
    g = torch.randn_like(p)
    torch.autograd.grad(p, [w], g)          # warm-up
    _lib.timing_enable(True)
    for _ in range(iters):
        p, d = ops.shapelet_bank(xn, [w], 1.0)
        torch.autograd.grad(p, [w], g)
    torch.cuda.synchronize()
    f, nf = _lib.timing_read("shp_fwd")
    b, nb = _lib.timing_read("shp_bwd")
    _lib.timing_enable(False)
    _lib.timing_enable(True)
    with torch.no_grad():
        for _ in range(iters):
            ops.shapelet_bank(xn, [w], 1.0)
    torch.cuda.synchronize()
    f0, nf0 = _lib.timing_read("shp_fwd")
    _lib.timing_enable(False)
    print(f"        fwd without d_save: {f0/nf0:7.3f} ms")
    E = B * K * C * (T - L + 1) * L
    print(f"L={L:4d}  fwd {f/nf:7.3f} ms ({E/(f/nf*1e-3)/1e12:5.1f} T elem/s)   bwd {b/nb:7.3f} ms ({E/(b/nb*1e-3)/1e12:5.1f} T elem/s)", flush=True)
    tot_f += f / nf; tot_b += b / nb
print(f"total fwd {tot_f:.3f} ms  bwd {tot_b:.3f} ms")
