"""Where does the HIP harness run leave the reference's trajectory?  (round-2 verdict, weak #1)

Lock-step diagnostic on the `driver_smoke` fixture (run a: InterpGN(FCN), 40 training samples, batch 8, Adam lr 5e-3): the CPU
oracle model (pinned to the reference harness at 1e-6 by tests/test_driver_smoke.py) and the HIP model start from the fixture's
initial weights and see the same batches.  After every optimizer step it prints
  * the per-step GRADIENT error of the HIP path evaluated AT THE ORACLE'S PARAMETERS (a third model that copies the oracle's
    weights before each step): is any single step outside 1e-4?
  * the accumulated parameter distance oracle <-> HIP per tensor, and how many elements have moved apart by more than lr / 2;
  * for those elements: the size of their gradient relative to the tensor's largest (Adam turns a sign flip of a noise-level
    gradient into a 2 lr separation in ONE step, whatever its magnitude).
Run on the GPU box:  python tests/diag_trajectory.py [steps] [freeze]   (freeze = 1: elements whose oracle gradient is below 1e-6 of
the tensor's largest are masked to exact zero on both sides before Adam sees them)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import golden, make_cfg, sd_from  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 15
    freeze = len(sys.argv) > 2 and sys.argv[2] == "1"
    import speech_imagery_eeg_amd  # noqa: F401
    from ign_hip import ops
    from models.InterpGN import InterpGN
    from oracle import ign_oracle as O
    g = golden("driver_smoke")
    dev = torch.device("cuda:0")
    cfg = make_cfg()
    sd0 = sd_from(g, "sd0.")
    orc = O.OracleIGN(cfg).train()
    orc.load_state_dict(sd0)
    hip = InterpGN(cfg)
    hip.load_state_dict(sd0)
    hip = hip.to(dev).train()
    probe = InterpGN(cfg)
    probe.load_state_dict(sd0)
    probe = probe.to(dev).train()
    oo = torch.optim.Adam(orc.parameters(), lr=5e-3)
    oh = torch.optim.Adam(hip.parameters(), lr=5e-3)
    X, Y = torch.from_numpy(g["train_x"]), torch.from_numpy(g["train_y"]).long()
    torch.manual_seed(777)
    lr = 5e-3
    step = 0
    zero_bias = lambda n: n.startswith("deep_model.block") and n.endswith(".0.bias")
    while step < steps:
        perm = torch.randperm(X.shape[0])
        for i in range(0, X.shape[0], 8):
            if step >= steps:
                break
            idx = perm[i:i + 8]
            x, y = X[idx], Y[idx]
            # single-step error at the oracle's parameters
            probe.load_state_dict(orc.state_dict())
            probe.zero_grad(set_to_none=True)
            out, info = orc(x)
            lo = O.train_loss("InterpGN", out, info, y)
            oo.zero_grad(set_to_none=True)
            lo.backward()
            _, ip = probe(x.to(dev), None, None, None)
            lp = ops.ign_loss(ip.shapelet_preds, ip.dnn_preds, y.to(dev), 1.0, reg=ip.loss)[0]
            lp.backward()
            worst = ("", 0.0)
            for (n, p), (_, q) in zip(orc.named_parameters(), probe.named_parameters()):
                if zero_bias(n):
                    continue
                sc = float(p.grad.abs().max())
                e = float((q.grad.cpu() - p.grad).abs().max()) / max(sc, 1e-30)
                if e > worst[1]:
                    worst = (n, e)
            # the HIP run's own step
            _, ih = hip(x.to(dev), None, None, None)
            lh = ops.ign_loss(ih.shapelet_preds, ih.dnn_preds, y.to(dev), 1.0, reg=ih.loss)[0]
            oh.zero_grad(set_to_none=True)
            lh.backward()
            for (n, p), (_, q) in zip(orc.named_parameters(), hip.named_parameters()):
                if zero_bias(n):
                    p.grad.zero_()
                    q.grad.zero_()
                elif freeze:
                    m = p.grad.abs() < 1e-6 * p.grad.abs().max()
                    p.grad[m] = 0.0
                    q.grad[m.to(dev)] = 0.0
            oo.step()
            oh.step()
            step += 1
            line = f"step {step:3d}  loss oracle {float(lo):.6f} hip {float(lh):.6f} (diff {abs(float(lo) - float(lh)):.2e})  " \
                   f"single-step grad err {worst[1]:.2e} ({worst[0]})"
            print(line)
            for (n, p), (_, q) in zip(orc.named_parameters(), hip.named_parameters()):
                d = (q.detach().cpu() - p.detach()).abs()
                far = d > lr / 2
                if int(far.sum()):
                    gr = p.grad.abs() / p.grad.abs().max().clamp_min(1e-30)
                    print(f"      {n}: {int(far.sum())} of {d.numel()} elements apart by > lr/2 (max {float(d.max()):.2e}); their |grad| / max|grad| "
                          f"at this step: median {float(gr[far].median()):.1e}, max {float(gr[far].max()):.1e}")
    # validation loss of both on the fixture's validation set (eval mode)
    vx, vy = torch.from_numpy(g["val_x"]), torch.from_numpy(g["val_y"]).long()
    orc.eval(); hip.eval()
    with torch.no_grad():
        o, io = orc(vx)
        h, ih = hip(vx.to(dev), None, None, None)
        vo = float(torch.nn.functional.cross_entropy(o, vy) + io.loss.mean())
        vh = float(torch.nn.functional.cross_entropy(h, vy.to(dev)) + ih.loss.mean())
    print(f"validation loss after {steps} steps: oracle {vo:.6f}  hip {vh:.6f}  diff {abs(vo - vh):.2e}   freeze={freeze}")


if __name__ == "__main__":
    main()
