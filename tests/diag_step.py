"""Diagnostic (not a test): wall-clock breakdown of one IGN training step at the benchmark shape."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.nn.functional as F
import speech_imagery_eeg_amd  # noqa
from ign_hip import ops, _lib
from models.InterpGN import InterpGN
from bench import ch_config

def T(msg, t0):
    torch.cuda.synchronize(); print(f"{msg}: {time.perf_counter()-t0:.3f}s", flush=True); return time.perf_counter()

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = ch_config()
t0 = time.perf_counter()
torch.manual_seed(0)
m = InterpGN(cfg).to(dev).train()
t0 = T("model build", t0)
x = torch.randn(B, 1000, 122, device=dev); y = torch.randint(0, 3, (B,), device=dev)
t0 = T("data", t0)
_lib.timing_enable(True)
for it in range(3):
    xn, xt = ops.instance_norm(x, want_raw=True)
    t0 = T(f"[{it}] instnorm", t0)
    p, d = m.sbm.shapelet_features(x)
    t0 = T(f"[{it}] shapelet fwd", t0)
    out = m.sbm.head(p)
    loss = out.square().mean()
    loss.backward()
    t0 = T(f"[{it}] shapelet bwd", t0)
    deep = m.deep_model(x)
    t0 = T(f"[{it}] fcn fwd", t0)
    deep.square().mean().backward()
    t0 = T(f"[{it}] fcn bwd", t0)
    for lab in ("instnorm", "shp_fwd", "shp_bwd", "reduce_parts"):
        print("   ", lab, _lib.timing_read(lab), flush=True)
