"""CPU oracle for the IGN hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  The product path (``speech-imagery-eeg_amd/``) never does and fails
loudly when its HIP extension is missing.

What this is: a memory-lean CPU restatement (pure torch, fp32) of the reference's
shapelet-bottleneck + gated two-expert forward/backward, written from the maths in
SURVEY.md App. A.  The reference materialises a ``(B, Tw, K, C, L)`` broadcast
(~3.7 GB / sample at the benchmark shape); this restatement walks the window axis in
chunks and carries a hand-derived backward for the distance so it can run B=256 rows on a
host.  Everything downstream of the distance tensor ``d`` is ordinary torch autograd of
the same expressions the reference evaluates.

Pinning: the reference has no tests / golden vectors.  The oracle is
pinned by ``tests/golden/*.npz`` -- outputs of the reference itself, imported in the build
container by ``tests/golden/make_golden.py`` (committed) -- and checked in
``tests/test_oracle_golden.py``.

Reference lines followed (``IGN/`` = InterpretGatedNetwork/):
  instance norm            IGN/model/Shapelet.py:186-187
  window distance (L1)     IGN/model/Shapelet.py:61-62,74
  MSE / cosine / pearson   IGN/model/Shapelet.py:24-40,64-69,11-19
  RBF + straight-thru max  IGN/model/Shapelet.py:77-84
  LTS soft-min gate        IGN/model/Shapelet.py:96-111
  SBM heads / regularisers IGN/model/Shapelet.py:170-177,199-205,217-230
  IGN gini gate            IGN/model/InterpGN.py:39-60
  FCN expert               IGN/model/FullyConvNet.py:31-59
  train-step loss          IGN/exp/experiment_classification.py:319-329
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

MODE_L1, MODE_MSE, MODE_COS, MODE_PEARSON = 0, 1, 2, 3


@dataclass
class OracleInfo:
    """Mirror of the reference's ModelInfo (IGN/utils/shapelet_util.py:17-25)."""
    d: torch.Tensor = None
    p: torch.Tensor = None
    eta: torch.Tensor = None
    shapelet_preds: torch.Tensor = None
    dnn_preds: torch.Tensor = None
    preds: torch.Tensor = None
    loss: torch.Tensor = None


def shapelet_lengths(seq_len: int, fracs: Sequence[float]) -> List[int]:
    """max(3, ceil(l * seq_len))  -- IGN/model/Shapelet.py:153."""
    return [max(3, int(np.ceil(l * seq_len).astype(int))) for l in fracs]


def shapelet_stride(seq_len: int, sl: int) -> int:
    """IGN/model/Shapelet.py:162."""
    return 1 if seq_len < 3000 else max(1, int(np.log2(sl)))


def instance_norm(x_btc: torch.Tensor) -> torch.Tensor:
    """(B,T,C) -> (B,C,T); (x-mean_T)/(std_T(unbiased)+1e-8).  IGN/model/Shapelet.py:186-187."""
    x = x_btc.permute(0, 2, 1)
    return (x - x.mean(dim=-1, keepdim=True)) / (x.std(dim=-1, keepdim=True) + 1e-8)


# --------------------------------------------------------------------------------------
# window distances, chunked over the window axis
# --------------------------------------------------------------------------------------
def _windows(x: torch.Tensor, t0: int, t1: int, L: int, stride: int) -> torch.Tensor:
    """x (B,C,T) -> windows t0..t1 as (B, t, 1, C, L)  (unfold + 'b m t l -> b t 1 m l')."""
    seg = x[:, :, t0 * stride: (t1 - 1) * stride + L]
    u = seg.unfold(2, L, stride)            # (B, C, t, L)
    return u.permute(0, 2, 1, 3).unsqueeze(2)


class _WindowDistance(torch.autograd.Function):
    """d[b,t,k,c] for modes L1-mean (Shapelet.py:74) and MSE (Shapelet.py:28).

    Forward and backward both walk the window axis in chunks so the 5-D broadcast is never
    resident.  Backward: L1: dd/dw_j = -sign(x-w)/L ; MSE: -2(x-w)/L ;
    dd/dx is the negation scattered back over the overlapping windows.
    """

    @staticmethod
    def forward(ctx, x, w, stride, mode, chunk):
        B, C, T = x.shape
        K, _, L = w.shape
        Tw = (T - L) // stride + 1
        d = x.new_empty(B, Tw, K, C)
        for t0 in range(0, Tw, chunk):
            t1 = min(Tw, t0 + chunk)
            diff = _windows(x, t0, t1, L, stride) - w       # (B,t,K,C,L)
            if mode == MODE_L1:
                d[:, t0:t1] = diff.abs().mean(dim=-1)
            else:
                d[:, t0:t1] = diff.pow(2).mean(dim=-1)
        ctx.save_for_backward(x, w)
        ctx.stride, ctx.mode, ctx.chunk = stride, mode, chunk
        return d

    @staticmethod
    def backward(ctx, gd):
        x, w = ctx.saved_tensors
        stride, mode, chunk = ctx.stride, ctx.mode, ctx.chunk
        B, C, T = x.shape
        K, _, L = w.shape
        Tw = gd.shape[1]
        gw = torch.zeros_like(w)
        gx = torch.zeros_like(x) if ctx.needs_input_grad[0] else None
        for t0 in range(0, Tw, chunk):
            t1 = min(Tw, t0 + chunk)
            diff = _windows(x, t0, t1, L, stride) - w       # (B,t,K,C,L)
            if mode == MODE_L1:
                dd = torch.sign(diff) / L
            else:
                dd = diff * (2.0 / L)
            g = gd[:, t0:t1].unsqueeze(-1) * dd              # d loss / d diff
            gw -= g.sum(dim=(0, 1))
            if gx is not None:
                gxc = g.sum(dim=2)                           # (B,t,C,L)
                for i, t in enumerate(range(t0, t1)):
                    gx[:, :, t * stride: t * stride + L] += gxc[:, i]
        return gx, gw, None, None, None


def window_distance(x, w, stride=1, mode=MODE_L1, chunk=16):
    """d (B,Tw,K,C) for any of the reference's distance_func branches (Shapelet.py:64-74)."""
    if mode in (MODE_L1, MODE_MSE):
        return _WindowDistance.apply(x, w, stride, mode, chunk)
    B, C, T = x.shape
    K, _, L = w.shape
    Tw = (T - L) // stride + 1
    outs = []
    for t0 in range(0, Tw, chunk):
        t1 = min(Tw, t0 + chunk)
        win = _windows(x, t0, t1, L, stride)
        if mode == MODE_COS:       # Shapelet.py:64-66
            sim = F.cosine_similarity(win, w, dim=-1)
        else:                      # Shapelet.py:67-69, 11-19
            xc = win - win.mean(dim=-1, keepdim=True)
            wc = w - w.mean(dim=-1, keepdim=True)
            num = torch.sum(xc * wc, dim=-1)
            den = torch.sqrt(torch.sum(xc ** 2, dim=-1) * torch.sum(wc ** 2, dim=-1)) + 1e-8
            sim = num / den
        outs.append(torch.ones_like(sim) - sim)
    return torch.cat(outs, dim=1)


def rbf_straight_through_max(d: torch.Tensor, eps: float):
    """Shapelet.py:77-84: p=exp(-(eps d)^2); max_p = sum_t (hard+soft-soft.detach()) p."""
    p = torch.exp(-torch.pow(eps * d, 2))
    hard = torch.zeros_like(p).scatter_(1, p.argmax(dim=1, keepdim=True), 1.)
    soft = torch.softmax(p, dim=1)
    onehot_max = hard + soft - soft.detach()
    max_p = torch.sum(onehot_max * p, dim=1)
    return max_p.flatten(start_dim=1), d.min(dim=1).values.flatten(start_dim=1)


def lts_softmin_gate(d: torch.Tensor, threshold: torch.Tensor):
    """Shapelet.py:105-111: straight-through soft-min then sigmoid(threshold - min_d)."""
    hard = torch.zeros_like(d).scatter_(1, d.argmin(dim=1, keepdim=True), 1.)
    soft = F.softmin(d, dim=1)
    onehot = hard + soft - soft.detach()
    min_d = torch.sum(onehot * d, dim=1)
    p = torch.sigmoid(threshold - min_d)
    return p.flatten(start_dim=1), d.min(dim=1).values.flatten(start_dim=1)


_MODES = {'euclidean': MODE_L1, 'cosine': MODE_COS, 'pearson': MODE_PEARSON}


class OracleShapelet(nn.Module):
    """IGN/model/Shapelet.py:46-87 (and :90-114 when lts=True)."""

    def __init__(self, dim_data, shapelet_len, num_shapelet=10, stride=1, eps=1.,
                 distance_func='euclidean', memory_efficient=False, lts=False, chunk=16):
        super().__init__()
        self.dim, self.length, self.n, self.stride = dim_data, int(shapelet_len), num_shapelet, stride
        self.distance_func, self.memory_efficient, self.lts = distance_func, memory_efficient, lts
        self.chunk = chunk
        self.weights = nn.Parameter(torch.normal(0, 1, (self.n, self.dim, self.length)))
        self.eps = eps
        if lts:
            self.threshold = nn.Parameter(torch.rand(1, self.n, self.dim).abs())

    def mode(self):
        if self.lts:      # LTS ignores cosine/pearson (Shapelet.py:100-103)
            return MODE_MSE if self.memory_efficient else MODE_L1
        m = _MODES.get(self.distance_func, MODE_L1)
        if m == MODE_L1 and self.memory_efficient:
            return MODE_MSE   # intended behaviour of the (broken, D7) memory_efficient branch
        return m

    def forward(self, x):
        d = window_distance(x, self.weights, self.stride, self.mode(), self.chunk)
        if self.lts:
            return lts_softmin_gate(d, self.threshold)
        return rbf_straight_through_max(d, self.eps)


class _OracleSelfAttention(nn.Module):
    """IGN/model/Shapelet.py:117-131."""

    def __init__(self, dim_feature, dim_attn):
        super().__init__()
        self.q_proj = nn.Linear(1, dim_attn)
        self.k_proj = nn.Linear(1, dim_attn)
        self.pos_embed = nn.Embedding(num_embeddings=dim_feature, embedding_dim=dim_attn)

    def forward(self, x):
        pos = self.pos_embed(torch.arange(x.shape[1], device=x.device))
        q = self.q_proj(x.unsqueeze(-1)) + pos
        k = self.k_proj(x.unsqueeze(-1)) + pos
        return F.scaled_dot_product_attention(q, k, x.unsqueeze(-1)).squeeze(-1)


class OracleSBM(nn.Module):
    """IGN/model/Shapelet.py:134-238 (lts=True: :241-262).  Same state_dict keys."""

    def __init__(self, configs, num_shapelet=(5, 5, 5, 5), shapelet_len=(0.1, 0.2, 0.3, 0.5),
                 lts=False, chunk=16):
        super().__init__()
        self.num_shapelet = list(num_shapelet)
        self.num_channel = configs.enc_in
        self.num_class = configs.num_class
        self.configs = configs
        self.shapelet_len = shapelet_lengths(configs.seq_len, shapelet_len)
        self.shapelets = nn.ModuleList([
            OracleShapelet(self.num_channel, sl, self.num_shapelet[i], eps=configs.epsilon,
                           distance_func=configs.distance_func,
                           memory_efficient=configs.memory_efficient,
                           stride=shapelet_stride(configs.seq_len, sl), lts=lts, chunk=chunk)
            for i, sl in enumerate(self.shapelet_len)])
        self.total_shapelets = sum(self.num_shapelet) * self.num_channel
        self.output_layer = nn.Linear(self.total_shapelets, self.num_class, bias=False)
        if configs.sbm_cls == 'bilinear':
            self.output_bilinear = nn.Bilinear(self.total_shapelets, self.total_shapelets,
                                               self.num_class, bias=False)
        elif configs.sbm_cls == 'attention':
            self.attention = _OracleSelfAttention(self.total_shapelets, 16)
        self.dropout = nn.Dropout(p=configs.dropout)
        self.lambda_reg, self.lambda_div = configs.lambda_reg, configs.lambda_div

    def forward(self, x, *args, **kwargs):
        xn = instance_norm(x)
        ps, ds = zip(*[s(xn) for s in self.shapelets])
        p, d = torch.cat(ps, dim=-1), torch.cat(ds, dim=-1)
        cls = self.configs.sbm_cls
        if cls == 'linear':
            out = self.output_layer(self.dropout(p))
        elif cls == 'bilinear':
            out = self.output_layer(self.dropout(p)) + self.output_bilinear(self.dropout(p), self.dropout(p))
        else:
            out = self.output_layer(self.dropout(self.attention(p)))
        return out, OracleInfo(d=d, p=p, shapelet_preds=out, preds=out, loss=self.loss().unsqueeze(0))

    def step(self):
        with torch.no_grad():
            self.output_layer.weight.clamp_(0.)

    def diversity(self):     # Shapelet.py:223-230, nn.PairwiseDistance(p=2) eps=1e-6
        loss = 0.
        for s in self.shapelets:
            sh = s.weights.permute(1, 0, 2)
            dist = F.pairwise_distance(sh.unsqueeze(1), sh.unsqueeze(2), p=2)
            mask = torch.ones_like(dist) - torch.eye(sh.shape[1]).unsqueeze(0)
            loss = loss + (torch.exp(-dist) * mask).mean()
        return loss

    def get_shapelets(self):   # Shapelet.py:232-238
        return [(s.weights.data[k, c, :].cpu().numpy(), c) for s in self.shapelets
                for k in range(s.weights.shape[0]) for c in range(s.weights.shape[1])]

    def loss(self):
        reg = self.output_layer.weight.abs().mean()
        div = self.diversity() if self.lambda_div > 0. else 0.
        return reg * self.lambda_reg + div * self.lambda_div


class OracleFCN(nn.Module):
    """IGN/model/FullyConvNet.py:7-59."""

    def __init__(self, configs):
        super().__init__()
        ks = (3, 3, 2) if configs.seq_len <= 10 else (8, 5, 3)
        self.block1 = nn.Sequential(nn.Conv1d(configs.enc_in, 128, ks[0]), nn.BatchNorm1d(128), nn.ReLU())
        self.block2 = nn.Sequential(nn.Conv1d(128, 256, ks[1]), nn.BatchNorm1d(256), nn.ReLU())
        self.block3 = nn.Sequential(nn.Conv1d(256, 128, ks[2]), nn.BatchNorm1d(128), nn.ReLU())
        self.fc = nn.Linear(128, configs.num_class)

    def forward(self, x, *args, **kwargs):
        x = x.permute(0, 2, 1)
        x = self.block3(self.block2(self.block1(x)))
        return self.fc(x.mean(dim=-1))


def gini_gate(sbm_out, deep_out, gating_value=None):
    """IGN/model/InterpGN.py:44-52."""
    p = F.softmax(sbm_out, dim=-1)
    c = sbm_out.shape[-1]
    gini = p.pow(2).sum(-1, keepdim=True)
    eta = (c * gini - 1) / (c - 1)
    if gating_value is not None:
        m = (eta > gating_value).float()
        eta = torch.ones_like(eta) * m + eta * (1 - m)
    return eta * sbm_out + (torch.ones_like(eta) - eta) * deep_out, eta


class OracleIGN(nn.Module):
    """IGN/model/InterpGN.py:22-66 with an injected deep expert (default FCN)."""

    def __init__(self, configs, num_shapelet=(5, 5, 5, 5), shapelet_len=(0.1, 0.2, 0.3, 0.5),
                 deep_model: Optional[nn.Module] = None, chunk=16):
        super().__init__()
        self.configs = configs
        self.sbm = OracleSBM(configs, num_shapelet, shapelet_len, chunk=chunk)
        self.deep_model = deep_model if deep_model is not None else OracleFCN(configs)

    def forward(self, x, x_mark_enc=None, x_dec=None, x_mark_dec=None, mask=None, gating_value=None):
        sbm_out, info = self.sbm(x)
        deep_out = self.deep_model(x, x_mark_enc, x_dec, x_mark_dec, mask)
        out, eta = gini_gate(sbm_out, deep_out, gating_value)
        return out, OracleInfo(d=info.d, p=info.p, eta=eta, shapelet_preds=sbm_out,
                               dnn_preds=deep_out, preds=out, loss=self.sbm.loss().unsqueeze(0))

    def loss(self):
        return self.sbm.loss()

    def step(self):
        self.sbm.step()


def compute_beta(epoch, max_epoch, schedule='cosine'):
    """IGN/exp/experiment_classification.py:19-26."""
    if schedule == 'cosine':
        return 1 / 2 * (1 + np.cos(np.pi * epoch / max_epoch))
    if schedule == 'linear':
        return 1 - epoch / max_epoch
    return 1


def train_loss(model_name, logits, info, label, beta=1.0):
    """IGN/exp/experiment_classification.py:320-329."""
    if model_name == 'DNN':
        return F.cross_entropy(logits, label)
    loss = F.cross_entropy(logits, label) + info.loss.mean()
    if model_name == 'InterpGN':
        loss = loss + beta * F.cross_entropy(info.shapelet_preds, label)
    return loss


# --------------------------------------------------------------------------------------
# closed-form backward of the shapelet block (what the HIP backward kernel implements);
# kept here so tests can check the hand-derived formulas against autograd on the CPU.
# --------------------------------------------------------------------------------------
def shapelet_l1_grad_w_closed_form(xn, w, eps, g):
    """dl/dw given upstream g = dl/dP[b,k,c], float64 friendly, small sizes."""
    B, C, T = xn.shape
    K, _, L = w.shape
    Tw = T - L + 1
    win = xn.unfold(2, L, 1).permute(0, 2, 1, 3).unsqueeze(2)     # (B,Tw,1,C,L)
    diff = win - w
    d = diff.abs().mean(-1)                                        # (B,Tw,K,C)
    p = torch.exp(-(eps * d) ** 2)
    s = torch.softmax(p, dim=1)
    mu = (s * p).sum(1, keepdim=True)
    hard = torch.zeros_like(p).scatter_(1, p.argmax(dim=1, keepdim=True), 1.)
    dldp = g.view(B, 1, K, C) * (hard + s * (p - mu))
    dldd = dldp * (-2 * eps * eps * d * p)
    return -(dldd.unsqueeze(-1) * torch.sign(diff)).sum(dim=(0, 1)) / L
