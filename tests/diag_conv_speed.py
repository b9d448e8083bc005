"""Diagnostic (not a test): device time of the FCN convolution kernels at the benchmark shape, f32 MFMA vs split-bf16."""
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
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for (Tin, Ci, Co, k, pro) in [(1000, 122, 128, 8, False), (993, 128, 256, 5, True), (989, 256, 128, 3, True)]:
    Tout = Tin - k + 1
    x = torch.randn(B, Tin, Ci, device=dev); w = torch.randn(Co, Ci, k, device=dev) / (Ci * k) ** 0.5; b = torch.randn(Co, device=dev)
    pa = (torch.rand(Ci, device=dev) + 0.5) if pro else None; pb = torch.randn(Ci, device=dev) if pro else None
    y = torch.empty(B, Tout, Co, device=dev); y6 = torch.empty_like(y)
    part = torch.empty(int(L.ign_clconv_x6_mtiles(B, Tout)), 2, Co, device=dev)
    wt = torch.empty(Co, k * Ci, device=dev); wd = torch.empty(Ci, k * Co, device=dev)
    wt3 = torch.empty(int(L.ign_clconv_x3_elems(Co, Ci, k)), device=dev, dtype=torch.bfloat16)
    wd3 = torch.empty(int(L.ign_clconv_x3_elems(Ci, Co, k)), device=dev, dtype=torch.bfloat16)
    _lib.check(L.ign_clconv_pack_weights(p(w), p(wt), p(wd), Co, Ci, k, s()), "pack")
    _lib.check(L.ign_clconv_pack_weights_x3(p(w), p(wt3), p(wd3), Co, Ci, k, s()), "pack3")
    flops = 2.0 * B * Tout * Co * Ci * k
    t32 = timeit(lambda: _lib.check(L.ign_clconv_fwd(p(x), p(wt), p(b), p(pa), p(pb), p(y), p(part), B, Tin, Ci, Co, k, s()), "f"))
    t6 = timeit(lambda: _lib.check(L.ign_clconv_fwd_x6(p(x), p(wt3), p(b), p(pa), p(pb), p(y6), p(part), B, Tin, Ci, Co, k, s()), "f6"))
    err = float((y - y6).abs().max() / y.abs().max())
    print(f"fwd  Ci={Ci} Co={Co} k={k}: f32 {t32:.3f} ms ({flops/t32/1e9:.0f} TFLOP/s)  bf16x6 {t6:.3f} ms ({flops/t6/1e9:.0f} TFLOP/s-equiv)  max rel diff {err:.1e}", flush=True)
    dy0 = torch.randn(B, Tout, Co, device=dev)
    dw = torch.empty(Co, Ci, k, device=dev); dw6 = torch.empty_like(dw)
    ws = torch.empty(int(L.ign_clconv_wgrad_workspace_bytes(B, Tin, Ci, Co, k)) // 4, device=dev)
    ws6 = torch.empty(int(L.ign_clconv_wgrad_x6_workspace_bytes(B, Tin, Ci, Co, k)) // 4, device=dev)
    t32 = timeit(lambda: _lib.check(L.ign_clconv_wgrad(p(dy0), 0, p(x), p(pa), p(pb), p(dw), p(ws), B, Tin, Ci, Co, k, s()), "w"))
    t6 = timeit(lambda: _lib.check(L.ign_clconv_wgrad_x6(p(dy0), 0, p(x), p(pa), p(pb), p(dw6), p(ws6), B, Tin, Ci, Co, k, s()), "w6"))
    err = float((dw - dw6).abs().max() / dw.abs().max())
    print(f"wgrad Ci={Ci} Co={Co} k={k}: f32 {t32:.3f} ms ({flops/t32/1e9:.0f} TFLOP/s)  bf16x6 {t6:.3f} ms ({flops/t6/1e9:.0f} TFLOP/s-equiv)  max rel diff {err:.1e}", flush=True)
    if pro:
        dyp = torch.randn(B, Tout + 2 * (k - 1), Co, device=dev)
        g = torch.empty(B, Tin, Ci, device=dev); g6 = torch.empty_like(g)
        partd = torch.empty(int(L.ign_clconv_x6_mtiles(B, Tin)), 2, Ci, device=dev)
        mean = torch.randn(Ci, device=dev); inv = torch.rand(Ci, device=dev) + 0.5
        flops = 2.0 * B * Tin * Co * Ci * k
        t32 = timeit(lambda: _lib.check(L.ign_clconv_dgrad(p(dyp), p(wd), p(x), p(pa), p(pb), p(mean), p(inv), p(g), p(partd), B, Tin, Ci, Co, k, s()), "d"))
        t6 = timeit(lambda: _lib.check(L.ign_clconv_dgrad_x6(p(dyp), p(wd3), p(x), p(pa), p(pb), p(mean), p(inv), p(g6), p(partd), B, Tin, Ci, Co, k, s()), "d6"))
        err = float((g - g6).abs().max() / g.abs().max())
        print(f"dgrad Ci={Ci} Co={Co} k={k}: f32 {t32:.3f} ms ({flops/t32/1e9:.0f} TFLOP/s)  bf16x6 {t6:.3f} ms ({flops/t6/1e9:.0f} TFLOP/s-equiv)  max rel diff {err:.1e}", flush=True)
