"""Diagnostic: host issue time of one IGN step (time until step() returns with an empty GPU queue in front of it)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import speech_imagery_eeg_amd  # noqa
from conftest import make_cfg
from ign_hip import ops as ign_ops
from ign_hip.ddp import FlatAdam, FlatParamBucket
from models.InterpGN import InterpGN
dev = torch.device("cuda:0")
cfg = make_cfg(enc_in=122, seq_len=1000, num_class=3)
torch.manual_seed(0)
m = InterpGN(cfg).to(dev).train()
bucket = FlatParamBucket(m, 1); opt = FlatAdam(bucket, lr=5e-3)
x = torch.randn(256, 1000, 122, device=dev); y = torch.randint(0, 3, (256,), device=dev); mask = torch.ones(256, 1000, device=dev)
def step():
    out, info = m(x, mask, None, None)
    loss = ign_ops.ign_loss(info.shapelet_preds, info.dnn_preds, y, 1.0)[0] + info.loss.mean()
    loss.backward(); opt.step(); bucket.zero_grad()
for _ in range(5): step()
torch.cuda.synchronize()
hs, ts = [], []
for _ in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    hs.append((t1 - t0) * 1e3); ts.append((t2 - t0) * 1e3)
print("host issue time per step: %.2f ms (min %.2f); step wall with sync %.2f ms" % (sum(hs) / len(hs), min(hs), sum(ts) / len(ts)))
import cProfile, pstats
pr = cProfile.Profile(); torch.cuda.synchronize(); pr.enable()
for _ in range(5): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
