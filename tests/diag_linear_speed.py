"""Diagnostic (not a test): device time of ops.linear's three GEMMs at the Transformer baseline's shapes."""
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

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

M = 256 * 1000
for (Ci, Co) in [(512, 512), (512, 2048), (2048, 512), (64, 512)]:
    x = torch.randn(M, Ci, device=dev); gy = torch.randn(M, Co, device=dev)
    dw = torch.empty(Co, Ci, device=dev); dw6 = torch.empty_like(dw)
    ws = torch.empty(int(L.ign_clconv_wgrad_workspace_bytes(1, M, Ci, Co, 1)) // 4, device=dev)
    ws6 = torch.empty(int(L.ign_clconv_wgrad_x6_workspace_bytes(1, M, Ci, Co, 1)) // 4, device=dev)
    flops = 2.0 * M * Ci * Co
    t32 = timeit(lambda: _lib.check(L.ign_clconv_wgrad(p(gy), 0, p(x), None, None, p(dw), p(ws), 1, M, Ci, Co, 1, s()), "w"))
    t6 = timeit(lambda: _lib.check(L.ign_clconv_wgrad_x6(p(gy), 0, p(x), None, None, p(dw6), p(ws6), 1, M, Ci, Co, 1, s()), "w6"))
    ref = gy.double().t() @ x.double()
    e32 = float((dw - ref).abs().max() / ref.abs().max()); e6 = float((dw6 - ref).abs().max() / ref.abs().max())
    print(f"wgrad Ci={Ci} Co={Co}: f32 {t32:.3f} ms ({flops/t32/1e9:.0f} TFLOP/s, err {e32:.1e})  bf16x6 {t6:.3f} ms "
          f"({flops/t6/1e9:.0f} TFLOP/s-equiv, err {e6:.1e})", flush=True)

# forward GEMM (ign_clconv_fwd_x6, k = 1) on the 128x256 eight-wave kernel (the 128x128 A/B of round 1 is in DESIGN 4.6)
for (Ci, Co) in [(512, 512), (512, 2048), (2048, 512), (512, 1536)]:
    x = torch.randn(M, Ci, device=dev); w = torch.randn(Co, Ci, device=dev) / Ci ** 0.5; b = torch.randn(Co, device=dev)
    wt3 = torch.empty(int(L.ign_clconv_x3_elems(Co, Ci, 1)), device=dev, dtype=torch.bfloat16)
    _lib.check(L.ign_clconv_pack_weights_x3(p(w), p(wt3), None, Co, Ci, 1, s()), "pack")
    y = torch.empty(M, Co, device=dev)
    t = timeit(lambda: _lib.check(L.ign_clconv_fwd_x6(p(x), p(wt3), p(b), None, None, p(y), None, 1, M, Ci, Co, 1, s()), "f"))
    ref = x[:4096].double() @ w.double().t() + b.double()
    err = float((y[:4096] - ref).abs().max() / ref.abs().max())
    print(f"fwd Ci={Ci} Co={Co}: {t:.3f} ms ({2.0*M*Ci*Co/t/1e9:.0f} TFLOP/s-equiv, err {err:.1e})", flush=True)
