"""Diagnostic: which tensors does ign_absmax still scan in a forward + backward of one encoder layer?"""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import speech_imagery_eeg_amd  # noqa
from ign_hip import ops, _lib
from models.eegcnn import _encoder_layer_forward
dev = torch.device("cuda:0")
torch.manual_seed(0)
layer = torch.nn.TransformerEncoderLayer(d_model=64, nhead=4, dim_feedforward=128, dropout=0.0, batch_first=True).to(dev).train()
x = torch.randn(4, 50, 64, device=dev)
L = _lib.lib()
orig = L.ign_absmax
def spy(*a):
    st = traceback.extract_stack()[:-1]
    print("absmax n=%d  <- %s" % (int(a[1]), " <- ".join(f"{f.name}:{f.lineno}" for f in st[-4:])))
    return orig(*a)
L.ign_absmax = spy
xin = x.clone().requires_grad_(True)
out = _encoder_layer_forward(layer, xin, 4)
print("---- backward")
(out * torch.randn_like(out)).sum().backward()
