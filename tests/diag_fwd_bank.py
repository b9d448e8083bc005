"""Diagnostic (not a test): shapelet forward at the benchmark shape through ign_shapelet_fwd (per group) and ign_shapelet_fwd_bank,
same box, same buffers, plus each group against the measured 32.3 T elements/s issue ceiling.  Round 2 used it to A/B a single-grid
kernel over all groups against the per-group launches (4.07 vs 3.97 ms: rejected, see csrc/ign_abi.hip)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import speech_imagery_eeg_amd  # noqa
from ign_hip import _lib
L = _lib.lib()
dev = torch.device("cuda:0")
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
s = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
B, C, T = 256, 122, 1000
Ks, Ls = [5, 5, 5, 5], [100, 200, 300, 500]
if len(sys.argv) > 1 and sys.argv[1] == "6x10":
    Ks, Ls = [10] * 6, [50, 100, 200, 300, 500, 800]
G = len(Ks)
torch.manual_seed(0)
xn = torch.randn(B, C, T, device=dev)
ws = [torch.randn(K, C, Lg, device=dev) for K, Lg in zip(Ks, Ls)]
ld = sum(Ks) * C
P = torch.empty(B, ld, device=dev); D = torch.empty_like(P)
ts = [torch.empty(B, K, C, device=dev, dtype=torch.int32) for K in Ks]
zs = [torch.empty(B, K, C, 2, device=dev) for K in Ks]
ds = [torch.empty(B, C, K, T - Lg + 1, device=dev) for K, Lg in zip(Ks, Ls)]
col0 = [sum(Ks[:i]) * C for i in range(G)]


def per_group():
    for g in range(G):
        _lib.check(L.ign_shapelet_fwd(p(xn), p(ws[g]), None, p(P), p(D), ld, col0[g], p(ts[g]), p(zs[g]), p(ds[g]), None, B, C, T,
                                      Ks[g], Ls[g], 1, 1.0, 0, s()), "fwd")


vpa, ia = ctypes.c_void_p * G, ctypes.c_int * G
pv = lambda xs: vpa(*[(t.data_ptr() if t is not None else None) for t in xs])


def bank():
    _lib.check(L.ign_shapelet_fwd_bank(p(xn), G, pv(ws), pv([None] * G), p(P), p(D), ld, ia(*col0), pv(ts), pv(zs), pv(ds),
                                       pv([None] * G), B, C, T, ia(*Ks), ia(*Ls), ia(*[1] * G), 1.0, 0, s()), "bank")


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for rep in range(3):
    print(f"per-group launches: {timeit(per_group):.3f} ms   one launch: {timeit(bank):.3f} ms", flush=True)
for g in range(G):
    t = timeit(lambda: _lib.check(L.ign_shapelet_fwd(p(xn), p(ws[g]), None, p(P), p(D), ld, col0[g], p(ts[g]), p(zs[g]), p(ds[g]), None,
                                                     B, C, T, Ks[g], Ls[g], 1, 1.0, 0, s()), "fwd"))
    E = B * Ks[g] * C * (T - Ls[g] + 1) * Ls[g]
    print(f"  group L={Ls[g]}: {t:.3f} ms = {E / t / 1e9:.1f} T elements/s ({E / t / 1e9 / 32.3:.2f} of the 32.3 T/s issue ceiling)", flush=True)
