"""GPU parity against the round-2 reference fixtures (tests/golden/make_golden.py): both baselines at the BENCHMARK
width (BASELINE.json configs 3 / 4: d_model 512, enc_in 122, seq_len 1000), the gated model with the Transformer expert,
three Adam steps of SBM / DNN-Transformer / EEG-CNN, LTS / cosine banks at the CHISCO row shape and the strided bank.
Tolerance 1e-4 (north_star); every comparison is recorded in the parity ledger with the reference's own fp32-vs-float64
distance beside it where the fixture holds a float64 run."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN, golden, make_cfg, parity, sd_from

sys.path.insert(0, GOLDEN)
from fill import seeded_fill  # noqa: E402

pytestmark = pytest.mark.gpu
SAMPLE = 2048


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _sample_idx(numel):
    return torch.linspace(0, numel - 1, min(SAMPLE, numel)).long()


def _check_compact_grads(m, g):
    """small gradients element-wise, large ones by norm and by the fixture's fixed strided sample; scale floor 1e-4 of
    the model's largest gradient (true-zero gradients -- key-projection bias, block1_bn1.* -- hold rounding noise)."""
    gmax = max(float(np.abs(g[k]).max()) for k in g if k.startswith(("grad.", "gradsample.")))
    n_checked = 0
    for n, p in m.named_parameters():
        ref = g.get("grad." + n, g.get("gradsample." + n))
        if ref is not None and float(np.abs(ref).max()) < 1e-5 * gmax:
            # TRUE gradient zero (block1_bn1.* under block1_bn2; a key-projection bias): rounding noise on both sides
            assert p.grad is None or float(p.grad.abs().max()) < 1e-5 * gmax, f"{n}: not noise-level"
            n_checked += 1
            continue
        if "grad." + n in g:
            parity("grad." + n, p.grad, g["grad." + n], kind="scale", floor=1e-4 * gmax, f64=g.get("grad64." + n))
        elif "gradsample." + n in g:
            idx = _sample_idx(p.numel()).to(p.device)
            parity("gradsample." + n, p.grad.flatten()[idx], g["gradsample." + n], kind="scale", floor=1e-4 * gmax,
                   f64=g.get("grad64sample." + n))
            parity("gradnorm." + n, p.grad.double().norm(), g["gradnorm." + n], kind="scale", f64=g.get("grad64norm." + n))
        else:
            assert p.grad is None, n            # constructed, never called on this path (temporal_embedding)
            continue
        n_checked += 1
    assert n_checked >= 20


def test_transformer_baseline_at_benchmark_width():
    """BASELINE config 4 at model level: Transformer.Model (IGN/model/Transformer.py:99-110) with enc_in 122, seq_len 1000,
    d_model 512, 8 heads, d_ff 2048, 2 layers -- the 122->512 circular token embedding, the mask multiply and the
    512 000->3 head included."""
    dev = _dev()
    import speech_imagery_eeg_amd  # noqa
    from models.Transformer import Model
    g = golden("transformer_ch512")
    torch.manual_seed(0)
    m = seeded_fill(Model(make_cfg(enc_in=122, seq_len=1000, num_class=3, c_out=3, d_model=512, n_heads=8, d_ff=2048)),
                    int(g["seed"])).to(dev).train()
    x, mask, y = _t(g["x"], dev), _t(g["mask"], dev), _t(g["y"], dev)
    out = m(x, mask, None, None)
    loss = F.cross_entropy(out, y)
    loss.backward()
    parity("out", out, g["out"], kind="elem", f64=g["out64"])
    parity("loss", loss, g["loss"], kind="elem", f64=g["loss64"])
    _check_compact_grads(m, g)


def test_eegcnn_baseline_at_benchmark_width():
    """BASELINE config 3 at model level: EEGCNNTransformer with d_model 512 (IGN/model/eegcnn.py:111-397), B=4."""
    dev = _dev()
    import speech_imagery_eeg_amd  # noqa
    from models.eegcnn import EEGCNNTransformer
    g = golden("eegcnn_ch512")
    torch.manual_seed(0)
    m = seeded_fill(EEGCNNTransformer(make_cfg(enc_in=122, seq_len=1000, num_class=3, c_out=3, d_model=512)),
                    int(g["seed"])).to(dev).train()
    x, y = _t(g["x"], dev), _t(g["y"], dev)
    out, info = m(x)
    loss = F.cross_entropy(out, y) + info.loss.mean()
    loss.backward()
    parity("out", out, g["out"], kind="elem", f64=g["out64"])
    parity("loss", loss, g["loss"], kind="elem", f64=g["loss64"])
    _check_compact_grads(m, g)
    for k in (k for k in g if k.startswith("sd_after.")):
        parity(k, m.state_dict()[k[9:]], g[k], kind="elem")
    m.eval()
    with torch.no_grad():
        parity("eval_feat", m.eegcnn(x), g["eval_feat"], kind="elem")
        parity("eval_out", m(x)[0], g["eval_out"], kind="elem")


def test_ign_with_transformer_expert_golden():
    """InterpGN(dnn_type='Transformer') -- the second entry of dnn_dict (IGN/model/InterpGN.py:13-19) inside the gate."""
    dev = _dev()
    import speech_imagery_eeg_amd  # noqa
    from models.InterpGN import InterpGN
    g = golden("ign_transformer")
    m = InterpGN(make_cfg(dnn_type="Transformer"))
    want = {k[3:] for k in g if k.startswith("sd.")}
    assert {k for k in m.state_dict() if not k.endswith("position_embedding.pe")} == want
    m.load_state_dict(sd_from(g), strict=False)
    m.to(dev).train()
    x, mask, y = _t(g["x"], dev), _t(g["mask"], dev), _t(g["y"], dev)
    out, info = m(x, mask, None, None)
    loss = F.cross_entropy(out, y) + info.loss.mean() + F.cross_entropy(info.shapelet_preds, y)
    loss.backward()
    for k, v in (("out", out), ("eta", info.eta), ("shapelet_preds", info.shapelet_preds), ("dnn_preds", info.dnn_preds),
                 ("p", info.p), ("d", info.d), ("model_loss", info.loss)):
        parity(k, v, g[k], kind="elem")
    parity("train_loss", loss, g["train_loss"], kind="elem")
    gmax = max(float(np.abs(g[k]).max()) for k in g if k.startswith("grad."))
    for n, p in m.named_parameters():
        if "grad." + n not in g:
            assert p.grad is None, n
            continue
        parity("grad." + n, p.grad, g["grad." + n], kind="scale", floor=1e-4 * gmax)


def _adam_trajectory_close(m, g, skip=()):
    """Adam divides by sqrt(v): an entry whose gradient is ~0 turns last-bit differences into +-lr steps.  So at most 5 %
    of a tensor's entries may leave the tight band (5e-4 + 5e-3 |ref|) and none may move further than 3 steps x 2 lr;
    the fraction outside the band is recorded."""
    for k, v in m.state_dict().items():
        if "sd3." + k not in g or any(s in k for s in skip) or not v.dtype.is_floating_point:
            continue
        a, b = v.detach().cpu().numpy().astype(np.float64), g["sd3." + k].astype(np.float64)
        if k.endswith("in_proj_bias"):           # packed q|k|v bias: the k third has a true-zero gradient (see below)
            third = a.shape[0] // 3
            a, b = np.delete(a, np.s_[third:2 * third]), np.delete(b, np.s_[third:2 * third])
        diff = np.abs(a - b)
        bad = diff > (5e-4 + 5e-3 * np.abs(b))
        assert bad.mean() <= 0.05, f"{k}: {bad.mean():.3%} of entries outside tolerance"
        assert diff.max() <= 3 * 2 * 5e-3 + 1e-6, f"{k}: max diff {diff.max():.3e}"


@pytest.mark.parametrize("which", ["sbm", "dnn_transformer", "eegcnn"])
def test_three_adam_steps_other_models(which):
    """train_step_{sbm,dnn_transformer,eegcnn}: the reference's training step (exp:313-343) three times with Adam lr 5e-3;
    the three losses at 1e-4, the parameters after step 3 inside the Adam band."""
    dev = _dev()
    import speech_imagery_eeg_amd  # noqa
    g = golden("train_step_" + which)
    if which == "sbm":
        from models.Shapelet import ShapeBottleneckModel
        m = ShapeBottleneckModel(make_cfg(), [10] * 6, [0.05, 0.1, 0.2, 0.3, 0.5, 0.8])
        step = lambda x, y: (lambda o, i: F.cross_entropy(o, y) + i.loss.mean())(*m(x))
    elif which == "dnn_transformer":
        from models.Transformer import Model
        m = Model(make_cfg())
        step = lambda x, y: F.cross_entropy(m(x, torch.ones(x.shape[0], x.shape[1], device=dev), None, None), y)
    else:
        from models.eegcnn import EEGCNNTransformer
        m = EEGCNNTransformer(make_cfg(enc_in=8, seq_len=200, num_class=3, c_out=3, d_model=128))
        step = lambda x, y: F.cross_entropy(m(x)[0], y)
    m.load_state_dict(sd_from(g, "sd0."), strict=False)
    m.to(dev).train()
    opt = torch.optim.Adam(m.parameters(), lr=5e-3)
    xs, ys = _t(g["xs"], dev), _t(g["ys"], dev)
    for i in range(3):
        loss = step(xs[i], ys[i])
        loss.backward()
        opt.step()
        opt.zero_grad()
        parity(f"loss_step{i}", loss, np.float64(g["losses"][i]), kind="elem")
    # True-zero gradients are pure Adam noise (+-lr per step) in every implementation and are left out: EEG-CNN's block1_bn1.*
    # (BN2 removes what BN1's affine adds) and every key-projection bias (softmax scores are shift-invariant).
    _adam_trajectory_close(m, g, skip=("block1_bn1.weight", "block1_bn1.bias", "block1_bn2.running_mean", "num_batches_tracked",
                                       "key_projection.bias"))        # bn2.running_mean absorbs the noise-driven bn1.bias one-for-one


@pytest.mark.parametrize("name", ["ch_lts", "ch_cos", "strided_l1", "strided_lts"])
def test_shapelet_bank_chisco_row_and_strided_golden(name):
    """LTS gate and cosine distance at the CHISCO row shape (C=122, T=1000, K=5, L=200), and the strided bank
    (T=3100 => stride int(log2 310) = 8, IGN/model/Shapelet.py:162) forward AND backward, against the reference module."""
    dev = _dev()
    import speech_imagery_eeg_amd  # noqa
    from ign_hip import ops
    g = golden("shapelet_" + name)
    B, C, T, K, L, stride = (int(v) for v in g["dims"])
    gen = torch.Generator().manual_seed(int(g["seed"]))
    xn = torch.randn(B, C, T, generator=gen).to(dev)
    w = torch.randn(K, C, L, generator=gen).to(dev).requires_grad_(True)
    thr = torch.rand(1, K, C, generator=gen).to(dev).requires_grad_(True)
    r = torch.randn(B, K * C, generator=gen).to(dev)
    lts = name.endswith("lts")
    mode = (ops.DIST_COS if name == "ch_cos" else ops.DIST_L1) | (ops.GATE_LTS if lts else 0)
    p, dmin = ops.shapelet_bank(xn, [w], 1.0, mode, thresholds=[thr] if lts else None, strides=[stride])
    (p * r).sum().backward()
    parity("p", p, g["p"], kind="elem")
    parity("dmin", dmin, g["dmin"], kind="elem")
    parity("grad_w", w.grad, g["grad_w"], kind="scale")
    if lts:
        parity("grad_thr", thr.grad, g["grad_thr"], kind="scale")


@pytest.mark.parametrize("name,lts", [("sbm_ch_6x10", False), ("lts_ch_6x10", True)])
def test_sbm_and_lts_at_the_chisco_shape_with_the_6x10_bank(name, lts):
    """`--model SBM` / `--model LTS` as the driver builds them (6 lengths x 10 shapelets, exp:264-270) at C 122, T 1000: outputs,
    every ModelInfo field, loss and all gradients against the reference's own run (B=2: ~7.4 GB per sample there).

    14 640 (sample, shapelet, channel) rows x ~900 windows: a few rows have their two best windows tie to the last bit, and the
    straight-through arg-max (Shapelet.py:79 / :101) is discontinuous there -- whichever window an implementation's summation
    order favours gets the hard part of the gradient.  The test finds those rows from the oracle's distances (gate values of
    the two best windows within 1e-6), requires the kernel's own choice to be one of the tied windows, bounds their number, and
    compares the gradient on all other rows at 1e-4."""
    dev = _dev()
    import speech_imagery_eeg_amd  # noqa
    from models.Shapelet import DistThresholdSBM, ShapeBottleneckModel
    from oracle import ign_oracle as O
    g = golden(name)
    cfg = make_cfg(enc_in=122, seq_len=1000, num_class=3, c_out=3)
    torch.manual_seed(0)
    m = seeded_fill((DistThresholdSBM if lts else ShapeBottleneckModel)(cfg, [10] * 6, [0.05, 0.1, 0.2, 0.3, 0.5, 0.8]), int(g["seed"]))
    with torch.no_grad():                      # the same rescaling make_golden.py applies after the fill
        for sh in m.shapelets:
            sh.weights.mul_(sh.weights.shape[-1] ** 0.5)
            if hasattr(sh, "threshold"):
                sh.threshold.abs_().mul_(10.0)
    w_cpu = [sh.weights.detach().clone() for sh in m.shapelets]
    m.to(dev).train()
    x, y = _t(g["x"], dev), _t(g["y"], dev)
    out, info = m(x)
    loss = F.cross_entropy(out, y) + info.loss.mean()
    loss.backward()
    for k, v in (("out", out), ("p", info.p), ("d", info.d), ("model_loss", info.loss), ("train_loss", loss)):
        parity(k, v, g[k], kind="elem")
    # rows whose two best windows tie (from the oracle's distances), and the kernel's choice among them
    xn_o = O.instance_norm(torch.from_numpy(g["x"]))
    m.eval()                                   # the match locations are an eval-mode output (the training step never reads them)
    with torch.no_grad():
        t_hip = m(x)[1].t.cpu().long()
    m.train()
    C, col, tie_rows, n_tie = 122, 0, [], 0
    for gi, w in enumerate(w_cpu):
        K = w.shape[0]
        d = O.window_distance(xn_o, w, 1, O.MODE_L1, chunk=16)                     # (B, Tw, K, C)
        score = -d if lts else torch.exp(-d.pow(2))                                 # arg-min d / arg-max p
        top2 = score.topk(2, dim=1).values
        tied = (top2[:, 0] - top2[:, 1]) <= 1e-6                                    # (B, K, C)
        chosen = score.gather(1, t_hip[:, col:col + K * C].view(-1, 1, K, C)).squeeze(1)
        assert float((top2[:, 0] - chosen).max()) <= 1e-6, f"group {gi}: the kernel's window is not a best window"
        tie_rows.append(tied.any(dim=0))                                             # (K, C)
        n_tie += int(tied.sum())
        col += K * C
    assert n_tie <= 0.005 * t_hip.numel(), n_tie
    gmax = max(float(np.abs(g[k]).max()) for k in g if k.startswith(("grad.", "gradsample.")))
    n_checked = 0
    for n, p in m.named_parameters():
        if "grad." + n in g:
            parity("grad." + n, p.grad, g["grad." + n], kind="scale", floor=1e-4 * gmax)
        elif not n.startswith("shapelets."):                     # the class weights: a function of the gate values only
            idx = _sample_idx(p.numel()).to(dev)
            parity("gradsample." + n, p.grad.flatten()[idx], g["gradsample." + n], kind="scale", floor=1e-4 * gmax)
            parity("gradnorm." + n, p.grad.double().norm(), g["gradnorm." + n], kind="scale")
        else:
            gi = int(n.split(".")[1])
            keep = (~tie_rows[gi]).to(dev)                                           # (K, C)
            idx = _sample_idx(p.numel())
            row_ok = keep.flatten()[(idx // p.shape[-1]).to(dev)]
            got, ref = p.grad.flatten()[idx.to(dev)][row_ok], torch.from_numpy(g["gradsample." + n]).to(dev)[row_ok]
            parity("gradsample." + n, got, ref, kind="scale", floor=1e-4 * gmax)
            rn = p.grad.double().norm(dim=-1)
            parity("gradrownorm." + n, rn[keep], torch.from_numpy(g["gradrownorm." + n]).to(dev)[keep], kind="scale")
        n_checked += 1
    assert n_checked == (13 if lts else 7)
    parity("rows left out as arg-max ties (count / 14640)", np.float64(n_tie / 14640.0), np.float64(0.0), tol=5e-3, kind="elem",
           ref_is="bound on the number of tie rows, not a parity claim")
