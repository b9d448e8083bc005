"""Diagnostic: do the four length groups of the shapelet bank finish sooner when their launches go to four streams (one merged
tail) than back to back on one stream (four tails)?  Forward and backward, benchmark shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import speech_imagery_eeg_amd  # noqa
from ign_hip import ops

dev = torch.device("cuda:0")
B, C, T, K = 256, 122, 1000, 5
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
torch.manual_seed(0)
xn, _ = ops.instance_norm(torch.randn(B, T, C, device=dev))
Ls = (100, 200, 300, 500)
ws = [torch.randn(K, C, L, device=dev, requires_grad=True) for L in Ls]
streams = [torch.cuda.Stream() for _ in Ls]
main = torch.cuda.current_stream()


def run(multi, order):
    outs = []
    if not multi:
        for i in order:
            outs.append(ops.shapelet_bank(xn, [ws[i]], 1.0))
    else:
        for i in order:
            streams[i].wait_stream(main)
            with torch.cuda.stream(streams[i]):
                outs.append(ops.shapelet_bank(xn, [ws[i]], 1.0))
        for i in order:
            main.wait_stream(streams[i])
    return outs


def bwd(outs, gs, multi, order):
    if not multi:
        for (p, d), g, i in zip(outs, gs, order):
            torch.autograd.grad(p, [ws[i]], g)
    else:
        for (p, d), g, i in zip(outs, gs, order):
            streams[i].wait_stream(main)
            with torch.cuda.stream(streams[i]):
                torch.autograd.grad(p, [ws[i]], g)
        for i in order:
            main.wait_stream(streams[i])


for order in ((0, 1, 2, 3), (3, 2, 1, 0)):
    for multi in (False, True):
        outs = run(multi, order)
        gs = [torch.randn_like(p) for p, _ in outs]
        bwd(outs, gs, multi, order)
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        tf = tb = 0.0
        for _ in range(iters):
            e[0].record()
            outs = run(multi, order)
            e[1].record()
            bwd(outs, gs, multi, order)
            e[2].record()
            torch.cuda.synchronize()
            tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
        print(f"order {order} {'4 streams' if multi else '1 stream '}: fwd {tf / iters:.3f} ms  bwd {tb / iters:.3f} ms", flush=True)
