"""EEG-CNN + Transformer-encoder baseline (IGN/model/eegcnn.py:11-397), working semantics only.

What "working" means for the reference: the input is (B, C, T); no padding mask (the mask
down-sampling of eegcnn.py:260-263 cannot match the CNN output length, so the only path that runs is mask=None ->
all-ones); the CNN hyper-parameters are the constructor defaults F1=8, D=8, k1=125, k2=25, P1=2, P2=5 because
``run.py`` defines ``--eegcnn_cnn_f1`` ... while the model reads ``cnn_filter1`` ... (D11); the "positional" code is
indexed by BATCH position and broadcast over time (D10: ``pe[:x.size(0)]`` on a batch-first tensor).  All of that is
reproduced, since parity is defined against what the reference computes.  The per-forward debug prints are not.

State-dict keys match the reference (``eegcnn.block1_conv1.weight`` ... ``transformer_encoder.layers.{i}.self_attn.
in_proj_weight`` ... ``classifier.weight``).  The encoder layers are evaluated explicitly (post-norm, relu) so the
attention core can run on the fused fp32-MFMA kernel instead of ``nn.MultiheadAttention``'s materialised scores.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from ign_hip import ops
from utils.shapelet_util import ModelInfo


class PositionalEncoding(nn.Module):
    def __init__(self, d_model, dropout=0.1, max_len=5000):
        super().__init__()
        self.dropout = nn.Dropout(p=dropout)
        pos = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
        div = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
        pe = torch.zeros(max_len, d_model)
        pe[:, 0::2] = torch.sin(pos * div)
        pe[:, 1::2] = torch.cos(pos * div)
        self.register_buffer('pe', pe.unsqueeze(0).transpose(0, 1))      # (max_len, 1, d)

    def forward(self, x):
        # x is batch-first (B, S, d): the slice is taken along the BATCH axis and broadcast over S (D10)
        return self.dropout(x + self.pe[:x.size(0), :])


class EEGcnn(nn.Module):
    """Temporal conv 1xk1 -> BN -> depthwise spatial Cx1 -> BN/ELU/AvgPool(P1) -> depthwise 1xk2 -> pointwise ->
    BN/ELU/AvgPool(P2)   (IGN/model/eegcnn.py:67-108).  (B, C, T) -> (B, F2, T/(P1*P2))."""

    def __init__(self, Chans=122, kernLength1=125, kernLength2=25, F1=8, D=8, F2=64, P1=2, P2=5, dropoutRate=0.1):
        super().__init__()
        Chans = int(Chans)
        self.block1_conv1 = nn.Conv2d(1, F1, (1, kernLength1), padding='same', bias=False)
        self.block1_bn1 = nn.BatchNorm2d(F1)
        self.block1_depthwise = nn.Conv2d(F1, D * F1, (Chans, 1), groups=F1, bias=False)
        self.block1_bn2 = nn.BatchNorm2d(D * F1)
        self.block1_elu = nn.ELU()
        self.block1_pool = nn.AvgPool2d((1, P1))
        self.block1_drop = nn.Dropout(p=dropoutRate)
        self.block2_conv1 = nn.Conv2d(D * F1, D * F1, (1, kernLength2), padding='same', groups=D * F1, bias=False)
        self.block2_conv2 = nn.Conv2d(D * F1, F2, 1, bias=False)
        self.block2_bn = nn.BatchNorm2d(F2)
        self.block2_elu = nn.ELU()
        self.block2_pool = nn.AvgPool2d((1, P2))
        self.block2_drop = nn.Dropout(p=dropoutRate)

    @staticmethod
    def _shifted_sums(x, k, pl):
        """S[j] = sum_{b,c,t} xpad[b,c,t+j]: total minus the samples a tap at offset j - pl never sees (zero padding)."""
        T = x.shape[-1]
        total = x.sum()
        pr = k - 1 - pl
        head = x[..., :pr].sum(dim=(0, 1)).cumsum(0) if pr > 0 else x.new_zeros(0)        # first o samples, o = 1..pr
        tail = x[..., T - pl:].sum(dim=(0, 1)).flip(0).cumsum(0) if pl > 0 else x.new_zeros(0)   # last o samples
        left = total - tail.flip(0)            # taps j = 0 .. pl-1  (offset j - pl < 0: the last pl - j samples fall outside)
        right = total - head                   # taps j = pl+1 .. k-1
        return torch.cat([left, total.reshape(1), right])

    _GRAM_INDEX = {}

    @classmethod
    def _window_gram(cls, x2d, k, pl):
        """G[j,j'] = sum_{rows} sum_{t<T} xp[t+j] xp[t+j'] (float64, (k,k)), xp = the row zero-padded by (pl, k-1-pl): the matrix
        behind sum_t (w (*) x)[t]^2 = w^T G w.  G[j,j+d] = C[d] - sum_{s<j} xp[s] xp[s+d] - sum_{s>=T+j} xp[s] xp[s+d]: the full lag
        sums C (ops.autocorr, one pass over x) minus edge terms that only involve the first / last k-1 samples of each row."""
        R, T = x2d.shape
        dev = x2d.device
        C = ops.autocorr(x2d, k)
        m = k - 1
        key = (k, str(dev))
        if key not in cls._GRAM_INDEX:
            j, jp = torch.arange(k, device=dev).view(k, 1), torch.arange(k, device=dev).view(1, k)
            col = torch.arange(m, device=dev).view(m, 1) + torch.arange(k, device=dev).view(1, k)       # s + d
            cls._GRAM_INDEX[key] = ((jp - j).abs(), torch.minimum(j.expand(k, k), jp.expand(k, k)), col.clamp(max=m - 1), col < m)
        d, lo, colc, valid = cls._GRAM_INDEX[key]
        if m <= 124:
            # D[s, d] = sum_rows xp[s] xp[s+d] at the two ends of the rows, straight from ign_edge_lagprod_fwd
            Dh, Dt = ops.edge_lagprod(x2d, k, pl)
        else:
            head = F.pad(x2d[:, :k - 1 - pl], (pl, 0))                 # xp[:, 0 : k-1]
            tail = F.pad(x2d[:, T - pl:], (0, k - 1 - pl))             # xp[:, T : T+k-1]

            def gram64(a):
                # (m, R) @ (R, m) over ~1000-row chunks in fp32 (batched GEMM), chunks added in float64
                nch = max(1, (R + 1023) // 1024)
                rows = (R + nch - 1) // nch
                a = F.pad(a, (0, 0, 0, nch * rows - R)).view(nch, rows, m)
                return torch.bmm(a.transpose(1, 2), a).sum(dim=0, dtype=torch.float64)

            Hh, Ht = gram64(head), gram64(tail)
            zero = Hh.new_zeros(())
            Dh = torch.where(valid, Hh.gather(1, colc), zero)          # D[s, d] = H[s, s+d]
            Dt = torch.where(valid, Ht.gather(1, colc), zero)
        z = Dh.new_zeros(1, k)
        cumh = torch.cat([z, Dh.cumsum(0)], 0)                     # [j, d] = sum_{s<j}
        cumt = torch.cat([Dt.flip(0).cumsum(0).flip(0), z], 0)     # [j, d] = sum_{s>=j}
        return C[d] - cumh[lo, d] - cumt[lo, d]

    def forward(self, x):
        """(B, C, T) -> (B, F2, T / (P1*P2)).  GPU path: block 1 is evaluated as
        y2 = a_f * (w1[f] (*) (W2 x)) + b_f * rowsum(W2)  (see csrc/ign_eegcnn.hip), block 2 as a depthwise HIP convolution +
        the 64x64 pointwise contraction, both BatchNorm+ELU+AvgPool stages as fused two-pass ops (csrc/ign_eegcnn_fused.hip);
        BatchNorm-1's batch variance is a quadratic form over the input's lag sums, so the (B,F1,C,T) tensor of the reference
        (eegcnn.py:90-91) is never formed."""
        if not x.is_cuda:
            raise ops._lib.IgnError(f"EEG-CNN: tensor on {x.device}; the product path runs on the MI355X only (no CPU fallback)")
        if torch.is_autocast_enabled():
            # the reference's default bf16 mode: this block's VALU kernels have no bf16 form and run in fp32 (a superset of the
            # reference's accuracy); the encoder behind it picks the bf16 single-product GEMM / attention kernels itself
            with torch.autocast(device_type="cuda", enabled=False):
                return self._forward_hip(x.float())
        return self._forward_hip(x)

    def _forward_hip(self, x):
        B, C, T = x.shape
        bn1 = self.block1_bn1
        w1 = self.block1_conv1.weight.reshape(self.block1_conv1.weight.shape[0], -1)          # (F1, k1)
        w2 = self.block1_depthwise.weight.reshape(self.block1_depthwise.weight.shape[0], C)    # (F1*D, C)
        F1, k1 = w1.shape
        D = w2.shape[0] // F1
        pl1 = (k1 - 1) // 2
        x = ops.contiguous_bct(x)                   # (a permuted view of the loader's time-first batch: HIP transpose)
        batch_stats = bn1.training or not bn1.track_running_stats
        if batch_stats and _BN1_VARIANCE == "gram" and 2 <= k1 <= 125 and k1 <= T <= 1024 and bn1.weight is not None \
                and (bn1.momentum is not None or not bn1.track_running_stats):
            # BatchNorm-1's batch statistics are a linear / quadratic form of the filter over statistics of the INPUT (per-tap sums
            # S, window Gram matrix G: two passes over x for all filters, no pass over the (B,F1,C,T) convolution): data statistics,
            # then mean / variance / affine fold and their gradients in closed form (ops.bn1_fold), running statistics included
            with torch.no_grad():
                G, S = ops.bn1_data_stats(x.view(B * C, T), k1, pl1)
                track = bn1.track_running_stats and bn1.training
                if track:
                    bn1.num_batches_tracked.add_(1)
            alpha, cshift = ops.bn1_fold(w1, bn1.weight, bn1.bias, w2.sum(dim=1), G, S, B * C * T, bn1.eps,
                                         bn1.momentum if track else 0.0, bn1.running_mean if track else None,
                                         bn1.running_var if track else None, D)
            u = ops.chan_contract(x, w2)
            v = ops.dwconv1d(u, w1.repeat_interleave(D, dim=0), pl1)
            h = ops.bn_elu_pool(v, self.block1_bn2, self.block1_pool.kernel_size[1], alpha=alpha, cshift=cshift)
            return self._block2(self.block1_drop(h))
        if batch_stats:
            n = B * C * T
            mu1 = (w1 @ self._shifted_sums(x, k1, pl1)) / n
            if _BN1_VARIANCE == "gram" and k1 <= 128 and T >= k1:
                # E[y^2] = w^T G w / n with the window Gram matrix of the INPUT (no pass over the (B,F1,C,T) convolution, and
                # autograd's gradient 2 G w needs no pass over the data at all); float64 for the difference of two moments
                with torch.no_grad():
                    G = self._window_gram(x.view(B * C, T), k1, pl1)
                wd = w1.double()
                var1 = (torch.einsum('fj,jk,fk->f', wd, G, wd) / n - mu1.double().square()).float()
            else:
                var1 = ops.conv1_sumsq(x.view(B * C, T), w1, mu1, pl1) / n
            if bn1.track_running_stats:
                with torch.no_grad():
                    bn1.num_batches_tracked.add_(1)
                    m = bn1.momentum
                    bn1.running_mean.mul_(1 - m).add_(mu1.detach(), alpha=m)
                    bn1.running_var.mul_(1 - m).add_(var1.detach() * (n / (n - 1)), alpha=m)
        else:
            mu1, var1 = bn1.running_mean, bn1.running_var
        a1 = bn1.weight * torch.rsqrt(var1 + bn1.eps)
        b1 = bn1.bias - a1 * mu1
        # Block 1 behind the electrode contraction (csrc/ign_eegcnn.hip): y2 = a_f (w1[f] (*) (W2 x)) + b_f rowsum(W2).  The
        # affine map (a_f, b_f rowsum) goes INTO the BatchNorm-2 op (ops.bn_elu_pool absorbs it analytically), so the chain is
        # contraction -> depthwise temporal conv -> [stats pass + apply pass: BN2, ELU, AvgPool] with no element-wise glue.
        u = ops.chan_contract(x, w2)                                                           # (B, F1*D, T)
        v = ops.dwconv1d(u, w1.repeat_interleave(D, dim=0), pl1)
        h = ops.bn_elu_pool(v, self.block1_bn2, self.block1_pool.kernel_size[1], alpha=a1.repeat_interleave(D),
                            cshift=b1.repeat_interleave(D) * w2.sum(dim=1))
        return self._block2(self.block1_drop(h))

    def _block2(self, h):
        # Block 2: depthwise temporal conv -> pointwise 64 -> 64 (the same contraction kernel) -> [BN3, ELU, AvgPool]
        w3 = self.block2_conv1.weight.reshape(self.block2_conv1.weight.shape[0], -1)           # (F1*D, k2)
        q = ops.dwconv1d(h, w3, (w3.shape[1] - 1) // 2)
        g = ops.chan_contract(q, self.block2_conv2.weight.reshape(self.block2_conv2.weight.shape[0], -1))
        g = ops.bn_elu_pool(g, self.block2_bn, self.block2_pool.kernel_size[1])
        return self.block2_drop(g)


# BatchNorm-1 batch variance: "gram" = quadratic form over the input's window Gram matrix (ops.autocorr); "conv" (kernel lengths
# beyond 128 or rows shorter than the kernel; tests) = brute-force pass over the un-stored convolution (ign_conv1_sumsq_*:
# 3.1e10 FMA forward, 6.2e10 backward at the benchmark shape)
_BN1_VARIANCE = "gram"


def _encoder_layer_forward(layer, x, n_heads):
    """One post-norm nn.TransformerEncoderLayer (relu) evaluated with the fused attention core."""
    B, S, d = x.shape
    sa = layer.self_attn
    qkv = ops.linear(x, sa.in_proj_weight, sa.in_proj_bias).view(B, S, 3, n_heads, d // n_heads)
    o = ops.attention_packed(qkv, 1.0 / math.sqrt(d // n_heads))      # gradients land in one packed buffer
    a = ops.linear(o.reshape(B, S, d), sa.out_proj.weight, sa.out_proj.bias)
    x = ops.layer_norm(x, layer.norm1, residual=layer.dropout1(a))
    u = ops.linear(x, layer.linear1.weight, layer.linear1.bias)
    hdn = ops.relu(u)                                               # (magnitude bound handed on in both directions)
    if layer.training and layer.dropout.p > 0:
        hdn = layer.dropout(hdn)
    ff = ops.linear(hdn, layer.linear2.weight, layer.linear2.bias)
    return ops.layer_norm(x, layer.norm2, residual=layer.dropout2(ff))


class EEGCNNTransformer(nn.Module):
    def __init__(self, configs=None, **kwargs):
        super().__init__()
        g = (lambda *names, default=None: next((getattr(configs, n) for n in names if hasattr(configs, n)), default)) \
            if configs is not None and hasattr(configs, '__dict__') else \
            (lambda *names, default=None: next((kwargs[n] for n in names if n in kwargs), default))
        input_channels = g('enc_in', 'input_channels', default=122)
        seq_len = g('seq_len', default=845)
        num_classes = g('c_out', 'num_class', 'num_classes', default=3)
        dropout1 = g('eegcnn_dropout1', 'dropout1', default=0.1)
        dropout2 = g('eegcnn_dropout2', 'dropout2', default=0.1)
        num_layers = g('eegcnn_layers', 'num_layers', default=0)
        pooling = g('eegcnn_pooling', 'pooling', default=None)
        f1 = g('cnn_filter1', 'F1', default=8)                       # D11: NOT --eegcnn_cnn_f1
        dmul = g('cnn_filter2', 'D', default=8)
        k1 = g('kernel_length1', 'kernLength1', default=125)
        k2 = g('kernel_length2', 'kernLength2', default=25)
        p1 = g('pooling1', 'P1', default=2)
        p2 = g('pooling2', 'P2', default=5)
        d_model = g('d_model', default=None)
        self.n_heads = g('eegcnn_n_heads', 'n_heads', default=8)
        d_ff = g('eegcnn_d_ff', 'dim_feedforward', default=256)
        self.output_attention = g('output_attention', default=False)

        self.input_channels, self.seq_len, self.num_classes = input_channels, seq_len, num_classes
        self.num_layers, self.pooling = num_layers, pooling
        self.eegcnn = EEGcnn(Chans=input_channels, kernLength1=k1, kernLength2=k2, F1=f1, D=dmul, F2=f1 * dmul,
                             P1=p1, P2=p2, dropoutRate=dropout1)
        self.cnn_out_channels = f1 * dmul
        if num_layers > 0:
            d_model = d_model or self.cnn_out_channels
            self.pos_encoder = PositionalEncoding(d_model, dropout=dropout2, max_len=5000)
            layer = nn.TransformerEncoderLayer(d_model=d_model, nhead=self.n_heads, dim_feedforward=d_ff,
                                               dropout=dropout2, batch_first=True)
            self.transformer_encoder = nn.TransformerEncoder(layer, num_layers=num_layers)
            self.cnn_projection = nn.Linear(self.cnn_out_channels, d_model) if self.cnn_out_channels != d_model \
                else nn.Identity()
            self.d_model = d_model
        else:
            self.d_model = self.cnn_out_channels
        self.classifier = nn.Linear((seq_len if pooling is None else 1) * self.d_model, num_classes)

    def features(self, x):
        """(B,C,T) -> encoder output (B,S,d)."""
        h = self.eegcnn(x).permute(0, 2, 1)
        if self.num_layers > 0:
            proj = self.cnn_projection
            # bounds and packed planes of all dense-layer weights of the encoder in one scan + two packing launches
            ops.prepare_linear_weights(([proj.weight] if isinstance(proj, nn.Linear) else []) +
                                       [w for l in self.transformer_encoder.layers
                                        for w in (l.self_attn.in_proj_weight, l.self_attn.out_proj.weight, l.linear1.weight,
                                                  l.linear2.weight)], need_dx=torch.is_grad_enabled())
            h = ops.linear(h, proj.weight, proj.bias) if isinstance(proj, nn.Linear) else proj(h)
            h = self.pos_encoder(h)
            for layer in self.transformer_encoder.layers:
                h = _encoder_layer_forward(layer, h, self.n_heads)
        return h

    def forward(self, x, padding_mask=None, *unused, **kw):
        if padding_mask is not None:
            raise ValueError("EEGCNNTransformer: the reference's mask path cannot run (mask length 498 vs 100 CNN steps, "
                             "SURVEY D9); call it with the (B,C,T) tensor only")
        h = self.features(x)
        if self.pooling is None:
            h = h.reshape(h.shape[0], -1)
        elif self.pooling == "mean":
            h = h.sum(dim=1) / h.shape[1]            # masked mean with an all-ones mask
        elif self.pooling == "sum":
            h = h.sum(dim=1)
        elif self.pooling == "top":
            h = h[:, 0, :]
        else:
            raise ValueError(f"Unsupported pooling method: {self.pooling}")
        logits = ops.head_linear(h, self.classifier.weight, self.classifier.bias)      # N = 3: the streaming head kernel
        return logits, ModelInfo(loss=torch.zeros((), device=logits.device), preds=logits)
