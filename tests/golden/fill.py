"""Deterministic parameter fill shared by `make_golden.py` (on the reference's modules) and the tests (on this repo's
modules): both sides walk the FLOATING-POINT PARAMETERS in sorted state-dict-key order and draw from one seeded CPU
generator, so a benchmark-width model (8 M parameters) needs no weights in its fixture.  Buffers (BatchNorm running
statistics, sin/cos tables) keep their constructor values, which are deterministic."""
import torch


def seeded_fill(model, seed, skip=()):
    g = torch.Generator().manual_seed(seed)
    params = dict(model.named_parameters())
    with torch.no_grad():
        for name in sorted(params):
            p = params[name]
            if any(name.startswith(s) for s in skip):
                continue
            if p.dim() >= 2:
                fan_in = p.numel() // p.shape[0]
                v = torch.randn(p.shape, generator=g) / max(fan_in, 1) ** 0.5
            elif name.endswith("weight"):                      # LayerNorm / BatchNorm scale
                v = 1.0 + 0.1 * torch.randn(p.shape, generator=g)
            else:                                              # biases, thresholds
                v = 0.05 * torch.randn(p.shape, generator=g)
            p.copy_(v.to(p.dtype))
    return model
