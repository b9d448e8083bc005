"""Shapelet bottleneck model on the MI355X kernels.

Drop-in for IGN/model/Shapelet.py: same classes, constructor arguments, ``forward`` return values and
state_dict keys (``shapelets.{i}.weights``, ``shapelets.{i}.threshold``, ``output_layer.weight`` ...), so
checkpoints interchange with the reference.  What differs is underneath: the reference unfolds the input
and materialises a ``(B, Tw, K, C, L)`` broadcast per length group (Shapelet.py:61-74); here the whole bank
runs as one autograd node over ``ign_shapelet_fwd/bwd`` (include/ign_abi.h), which keeps only the
``(B, C, K, Tw)`` distances for the backward.

There is no CPU path: tensors must live on the GPU and libign_hip.so must be built (ign_hip._lib).
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ign_hip import ops
from ign_hip._lib import IgnError
from utils.shapelet_util import ModelInfo

_DIST_CODES = {'euclidean': ops.DIST_L1, 'cosine': ops.DIST_COS, 'pearson': ops.DIST_PEARSON}


def _distance_code(distance_func, memory_efficient, lts=False):
    """Map the reference's flags to a kernel mode.

    'euclidean' is mean |x-w| (Shapelet.py:74); ``memory_efficient`` selects the mean squared
    difference of ShapeletDistanceFunc (Shapelet.py:24-40; its call site is broken upstream, D7 -- the
    intended result is reproduced).  The LTS variant ignores cosine/pearson (Shapelet.py:100-103).
    """
    if lts:
        return ops.DIST_MSE if memory_efficient else ops.DIST_L1
    code = _DIST_CODES.get(distance_func, ops.DIST_L1)
    if code == ops.DIST_L1 and memory_efficient:
        return ops.DIST_MSE
    return code


class Shapelet(nn.Module):
    """One length group: K shapelets of length L on C channels (IGN/model/Shapelet.py:46-87)."""
    gate = ops.GATE_RBF

    def __init__(self, dim_data, shapelet_len, num_shapelet=10, stride=1, eps=1., distance_func='euclidean',
                 memory_efficient=False):
        super().__init__()
        self.dim = dim_data
        self.length = int(shapelet_len)
        self.n = num_shapelet
        self.stride = stride
        self.distance_func = distance_func
        self.memory_efficient = memory_efficient
        self.weights = nn.Parameter(torch.normal(0, 1, (self.n, self.dim, self.length)), requires_grad=True)
        self.eps = eps

    def mode(self):
        return _distance_code(self.distance_func, self.memory_efficient, self.gate == ops.GATE_LTS) | self.gate

    def forward(self, x):
        """x: instance-normalised (B, C, T) -> (p, d_min), each (B, K*C) with feature index k*C + c."""
        thr = [self.threshold] if self.gate == ops.GATE_LTS else None
        return ops.shapelet_bank(x, [self.weights], self.eps, self.mode(), [self.stride], thr)

    def derivative(self):
        return torch.diff(self.weights, dim=-1)


class DistThresholdShapelet(Shapelet):
    """LTS gate: straight-through soft-min then sigmoid(threshold - min_d) (IGN/model/Shapelet.py:90-114)."""
    gate = ops.GATE_LTS

    def __init__(self, dim_data, shapelet_len, num_shapelet=10, stride=1, eps=1., distance_func='euclidean',
                 memory_efficient=False):
        super().__init__(dim_data, shapelet_len, num_shapelet, stride, eps, distance_func, memory_efficient)
        self.threshold = nn.Parameter(torch.rand(1, self.n, self.dim).abs(), requires_grad=True)


class SelfAttention(nn.Module):
    """16-d attention over the shapelet features, used by sbm_cls='attention' (IGN/model/Shapelet.py:117-131)."""

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


class ShapeBottleneckModel(nn.Module):
    """SBM expert: instance norm -> shapelet bank -> linear / bilinear / attention head
    (IGN/model/Shapelet.py:134-238)."""
    _shapelet_cls = Shapelet

    def __init__(self, configs, num_shapelet=[5, 5, 5, 5], shapelet_len=[0.1, 0.2, 0.3, 0.5]):
        super().__init__()
        self.num_shapelet = list(num_shapelet)
        self.num_channel = configs.enc_in
        self.num_class = configs.num_class
        self.normalize = True
        self.configs = configs
        self.shapelet_len = []
        self.shapelets = nn.ModuleList()
        for i, frac in enumerate(shapelet_len):
            sl = max(3, int(np.ceil(frac * configs.seq_len)))                 # Shapelet.py:153
            stride = 1 if configs.seq_len < 3000 else max(1, int(np.log2(sl)))  # Shapelet.py:162
            self.shapelets.append(self._shapelet_cls(
                dim_data=self.num_channel, shapelet_len=sl, num_shapelet=self.num_shapelet[i],
                eps=configs.epsilon, distance_func=configs.distance_func,
                memory_efficient=configs.memory_efficient, stride=stride))
            self.shapelet_len.append(sl)
        self.total_shapelets = sum(self.num_shapelet) * self.num_channel

        if configs.sbm_cls not in ('linear', 'bilinear', 'attention'):
            raise ValueError(f"sbm_cls must be linear|bilinear|attention, got {configs.sbm_cls!r}")
        self.output_layer = nn.Linear(self.total_shapelets, self.num_class, bias=False)
        if configs.sbm_cls == 'bilinear':
            self.output_bilinear = nn.Bilinear(self.total_shapelets, self.total_shapelets, self.num_class, bias=False)
        elif configs.sbm_cls == 'attention':
            self.attention = SelfAttention(self.total_shapelets, 16)
        self.dropout = nn.Dropout(p=configs.dropout)
        self.lambda_reg = configs.lambda_reg
        self.lambda_div = configs.lambda_div

    # set by InterpGN when a deep expert on the same stream consumes the raw batch through the fp16 GEMMs: the instance-norm
    # pass then also takes max |x| (ops.instance_norm), which spares that expert a pass of its own
    input_bound = False

    # -- hot path ---------------------------------------------------------------------------------
    def shapelet_features(self, x, xn=None):
        """x (B,T,C) on the GPU -> (p, d_min) of every group, concatenated in the reference's order.
        `xn`: the instance-normalised (B,C,T) tensor if the caller already ran ign_instnorm_fwd (InterpGN does, to
        share the transpose with the FCN expert)."""
        if xn is None:
            xn, _ = ops.instance_norm(x, input_bound=self.input_bound)
        first = self.shapelets[0]
        thr = [s.threshold for s in self.shapelets] if first.gate == ops.GATE_LTS else None
        return ops.shapelet_bank(xn, [s.weights for s in self.shapelets], first.eps, first.mode(),
                                 [s.stride for s in self.shapelets], thr, return_tstar=True)

    def head(self, p):
        cls = self.configs.sbm_cls
        if cls == 'linear':
            return ops.head_linear(self.dropout(p), self.output_layer.weight)
        if cls == 'bilinear':
            return self.output_layer(self.dropout(p)) + self.output_bilinear(self.dropout(p), self.dropout(p))
        return self.output_layer(self.dropout(self.attention(p)))

    def _fused_forward(self, x, xn):
        """Bank + linear head + both regularisers as ONE autograd node (ops.SbmFn): 1 instance-norm launch, G shapelet launches,
        1 regulariser launch, 1 head launch -- and a backward without a single torch kernel.  Returns None when the
        configuration is outside that node (pearson centring runs through autograd; > 8 groups; > 16 shapelets per group)."""
        first = self.shapelets[0]
        mode = first.mode()
        G = len(self.shapelets)
        if (mode & 0xf) == ops.DIST_PEARSON or G > 8 or max(s.n for s in self.shapelets) > 16 or not x.is_cuda \
                or x.dtype != torch.float32:
            return None
        if xn is None:
            xn, _ = ops.instance_norm(x, input_bound=self.input_bound)
        lts = first.gate == ops.GATE_LTS
        fuse_head = self.configs.sbm_cls == 'linear' and (self.dropout.p == 0.0 or not self.training) \
            and self.total_shapelets % 4 == 0          # the streaming head kernel reads float4 rows (else: ops.head_linear's route)
        W = self.output_layer.weight
        ws = getattr(self, '_reg_ws', None)
        need = ops._lib.lib().ign_sbm_reg_workspace_bytes(G, self.num_channel, W.numel()) // 4
        if ws is None or ws.device != x.device or ws.numel() < need:
            ws = self._reg_ws = torch.zeros(need, device=x.device, dtype=torch.float32)   # zero-filled once; the kernel re-arms it
        cfg = (first.eps, mode, tuple(s.stride for s in self.shapelets), G, float(self.lambda_reg), float(self.lambda_div),
               fuse_head, not self.training, ws)
        params = [s.weights for s in self.shapelets] + ([s.threshold for s in self.shapelets] if lts else [])
        p, d, t, reg, out = ops.SbmFn.apply(xn, cfg, W, *params)
        if out is None:
            out = self.head(p)
        return out, ModelInfo(d=d, p=p, shapelet_preds=out, preds=out, loss=reg, t=t)

    def forward(self, x, *args, xn=None, **kwargs):
        """-> (logits, ModelInfo).  ModelInfo.t (match locations) is filled in eval mode only: the training step never reads it."""
        fused = self._fused_forward(x, xn)
        if fused is not None:
            return fused
        p, d, t = self.shapelet_features(x, xn)
        out = self.head(p)
        return out, ModelInfo(d=d, p=p, shapelet_preds=out, preds=out, loss=self.loss().unsqueeze(0), t=t)

    def match_layout(self):
        """per feature column (g*K*C + k*C + c): (window stride, shapelet length) -- turns ModelInfo.t into sample ranges"""
        stride = torch.cat([torch.full((s.n * self.num_channel,), s.stride, dtype=torch.int32) for s in self.shapelets])
        length = torch.cat([torch.full((s.n * self.num_channel,), s.length, dtype=torch.int32) for s in self.shapelets])
        return stride, length

    # -- regularisers / utilities -----------------------------------------------------------------
    def step(self):
        """Clamp the class weights to be non-negative (--pos_weight; Shapelet.py:212-215)."""
        with torch.no_grad():
            self.output_layer.weight.clamp_(0.)

    def diversity(self):
        """sum_g mean_{c,i,j} exp(-||w_i - w_j + 1e-6||_2) (1 - delta_ij)   (Shapelet.py:223-230)."""
        total = 0.
        for s in self.shapelets:
            fused = ops.diversity(s.weights)            # one HIP launch per group (loss + gradient)
            if fused is not None:
                total = total + fused
                continue
            sh = s.weights.permute(1, 0, 2)                                   # (C,K,L)
            dist = F.pairwise_distance(sh.unsqueeze(1), sh.unsqueeze(2), p=2)  # eps=1e-6 like nn.PairwiseDistance
            off_diag = 1.0 - torch.eye(sh.shape[1], device=dist.device, dtype=dist.dtype).unsqueeze(0)
            total = total + (torch.exp(-dist) * off_diag).mean()
        return total

    def loss(self):
        reg = self.output_layer.weight.abs().mean()
        div = self.diversity() if self.lambda_div > 0. else 0.
        return reg * self.lambda_reg + div * self.lambda_div

    def get_shapelets(self):
        out = []
        for s in self.shapelets:
            w = s.weights.data.cpu().numpy()
            for k in range(w.shape[0]):
                for c in range(w.shape[1]):
                    out.append((w[k, c, :], c))
        return out


class DistThresholdSBM(ShapeBottleneckModel):
    """'LTS' registry entry: SBM with the soft-min / threshold gate (IGN/model/Shapelet.py:241-262)."""

    def __init__(self, configs, num_shapelet=[5, 5, 5, 5], shapelet_len=[0.1, 0.2, 0.3, 0.5]):
        # The reference first builds the RBF bank (consuming RNG draws), then replaces it (Shapelet.py:248-262).
        # Same order here, so a given torch seed yields the reference's initial weights and thresholds.
        super().__init__(configs, num_shapelet, shapelet_len)
        self.shapelets = nn.ModuleList([
            DistThresholdShapelet(dim_data=self.num_channel, shapelet_len=s.length, num_shapelet=s.n, eps=s.eps,
                                  distance_func=s.distance_func, memory_efficient=s.memory_efficient,
                                  stride=s.stride)
            for s in self.shapelets])
