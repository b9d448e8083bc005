"""Diagnostic: sizes ign_absmax still scans in one training step of the Transformer baseline (small shape)."""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import speech_imagery_eeg_amd  # noqa
from conftest import make_cfg
from ign_hip import ops, _lib
import importlib
name = sys.argv[1] if len(sys.argv) > 1 else "Transformer"
cfg = make_cfg(enc_in=6, seq_len=100, num_class=4)
torch.manual_seed(0)
m = importlib.import_module(f"models.{name}").Model(cfg).cuda().train()
x = torch.randn(8, 100, 6, device="cuda")
L = _lib.lib()
orig = L.ign_absmax
def spy(*a):
    st = traceback.extract_stack()[:-1]
    print("absmax n=%d  <- %s" % (int(a[1]), " <- ".join(f"{f.name}:{f.lineno}" for f in st[-5:])))
    return orig(*a)
L.ign_absmax = spy
out = m(x, torch.ones(8, 100, device="cuda"), None, None)
print("---- backward")
out.square().mean().backward()
