"""Diagnostic (not a test): attention forward, fp32 MFMA vs split-bf16, at the Transformer baseline's shape."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import speech_imagery_eeg_amd  # noqa
from ign_hip import ops
dev = torch.device("cuda:0")

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for (B, L, H, E) in [(256, 1000, 8, 64), (256, 100, 8, 64), (64, 1000, 4, 128), (256, 1000, 16, 32)]:
    q, k, v = (torch.randn(B, L, H, E, device=dev) for _ in range(3))
    flops = 4.0 * B * H * L * L * E
    res = {}
    for m in ("f32", "bf16x6"):
        ops.ATTN_MATH = m
        with torch.no_grad():
            t = timeit(lambda: ops.attention(q, k, v, 1 / math.sqrt(E)))
            res[m] = (t, ops.attention(q, k, v, 1 / math.sqrt(E)))
    for m in ("f32", "bf16x6"):
        ops.ATTN_MATH = m
        qg, kg, vg = (t.clone().requires_grad_(True) for t in (q, k, v))
        o = ops.attention(qg, kg, vg, 1 / math.sqrt(E))
        go = torch.randn_like(o)
        tb = timeit(lambda: torch.autograd.grad(o, (qg, kg, vg), go, retain_graph=True))
        res[m] = res[m] + (tb,)
    print(f"   backward: f32 {res['f32'][2]:.3f} ms  bf16x6 {res['bf16x6'][2]:.3f} ms", flush=True)
    d = float((res["f32"][1] - res["bf16x6"][1]).abs().max() / res["f32"][1].abs().max())
    print(f"B={B} L={L} H={H} E={E}: f32 {res['f32'][0]:.3f} ms ({flops/res['f32'][0]/1e9:.0f} TFLOP/s)  "
          f"bf16x6 {res['bf16x6'][0]:.3f} ms ({flops/res['bf16x6'][0]/1e9:.0f} TFLOP/s-equiv)  max rel diff {d:.1e}", flush=True)
