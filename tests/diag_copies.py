"""Diagnostic: the aten::copy_ / add / fill launches of one EEG-CNN (or Transformer) training step, with shapes and Python call
sites (torch.profiler with stacks).   python tests/diag_copies.py [eegcnn|Transformer]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import speech_imagery_eeg_amd  # noqa
from conftest import make_cfg
import importlib
from torch.profiler import ProfilerActivity, profile
name = sys.argv[1] if len(sys.argv) > 1 else "eegcnn"
B = 256
if name == "eegcnn":
    from models.eegcnn import EEGCNNTransformer
    cfg = make_cfg(enc_in=122, seq_len=1000, num_class=3)
    cfg.eegcnn_layers, cfg.d_model, cfg.eegcnn_pooling = 2, 512, "mean"
    m = EEGCNNTransformer(cfg).cuda().train()
    x = torch.randn(B, 122, 1000, device="cuda")
    fwd = lambda: m(x)
else:
    cfg = make_cfg(enc_in=122, seq_len=1000, num_class=3)
    m = importlib.import_module(f"models.{name}").Model(cfg).cuda().train()
    x = torch.randn(B, 1000, 122, device="cuda")
    fwd = lambda: m(x, torch.ones(B, 1000, device="cuda"), None, None)


def step():
    out = fwd()
    out = out[0] if isinstance(out, tuple) else out
    out.square().mean().backward()
    m.zero_grad(set_to_none=True)


for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
rows = []
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::add", "aten::add_", "aten::fill_", "aten::mul", "aten::sum", "aten::index_select", "aten::repeat_interleave"):
        dev_us = getattr(ev, "device_time_total", 0) or getattr(ev, "cuda_time_total", 0)
        if dev_us <= 0:
            continue
        stack = [s for s in (ev.stack or []) if "speech-imagery" in s or "models/" in s or "ign_hip" in s or "layers/" in s][:3]
        rows.append((dev_us, ev.name, str(ev.input_shapes)[:70], " <- ".join(s.split("/")[-1][:60] for s in stack)))
rows.sort(reverse=True)
for r in rows[:45]:
    print("%7.1f us  %-18s %-70s %s" % r)
print("total device us:", sum(r[0] for r in rows), "in", len(rows), "ops")
