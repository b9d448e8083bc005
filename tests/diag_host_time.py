"""Diagnostic: host issue time of one IGN step (time until step() returns with an empty GPU queue in front of it) and a cProfile
of the host side.   python tests/diag_host_time.py [bm|ch] [batch]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import speech_imagery_eeg_amd  # noqa
from conftest import make_cfg
from ign_hip import ops as ign_ops
from ign_hip.ddp import FlatAdam, FlatParamBucket
from models.InterpGN import InterpGN

shape = sys.argv[1] if len(sys.argv) > 1 else "bm"
C, T, NC = (6, 100, 4) if shape == "bm" else (122, 1000, 3)
B = int(sys.argv[2]) if len(sys.argv) > 2 else (32 if shape == "bm" else 256)
dev = torch.device("cuda:0")
cfg = make_cfg(enc_in=C, seq_len=T, num_class=NC)
torch.manual_seed(0)
m = InterpGN(cfg).to(dev).train()
bucket = FlatParamBucket(m, 1)
opt = FlatAdam(bucket, lr=5e-3)
x = torch.randn(B, T, C, device=dev)
y = torch.randint(0, NC, (B,), device=dev)
mask = torch.ones(B, T, device=dev)


def step():
    out, info = m(x, mask, None, None)
    loss = ign_ops.ign_loss(info.shapelet_preds, info.dnn_preds, y, 1.0, reg=info.loss)[0]
    ign_ops.backward(loss)
    opt.step()
    bucket.zero_grad()


for _ in range(20):
    step()
torch.cuda.synchronize()
hs, ts = [], []
for _ in range(50):
    torch.cuda.synchronize(); t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    hs.append((t1 - t0) * 1e3); ts.append((t2 - t0) * 1e3)
print("host issue time per step: %.3f ms (min %.3f); step wall with sync %.3f ms" % (sum(hs) / len(hs), min(hs), sum(ts) / len(ts)))
t0 = time.perf_counter()
for _ in range(200):
    step()
torch.cuda.synchronize()
print("200 back-to-back steps: %.3f ms/step" % ((time.perf_counter() - t0) * 1e3 / 200))
pr = cProfile.Profile(); torch.cuda.synchronize(); pr.enable()
for _ in range(50):
    step()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumtime").print_stats(40)
