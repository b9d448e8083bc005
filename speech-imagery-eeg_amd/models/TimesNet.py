"""TimesNet deep expert, classification head (IGN/model/TimesNet.py:9-69,72-104,185-202; IGN/layers/Conv_Blocks.py:5-33):
embedding -> e_layers x LayerNorm(TimesBlock) -> gelu -> x mask -> flatten -> Linear(T*d -> N).

A TimesBlock finds the top-k periods of the batch (rFFT amplitudes), folds the series into a (T/period, period) image per
period, runs Inception -> GELU -> Inception over it and mixes the k results with softmax(amplitude) weights.

State-dict keys match the reference (``model.{i}.conv.{0,2}.kernels.{j}.*``, ``enc_embedding.*``, ``layer_norm``,
``projection``).  On the GPU:
  * an Inception block is the MEAN of `num_kernels` same-padded Conv2d's with kernels 1,3,..,2n-1.  Convolution is linear in
    its weights, so the block is ONE (2n-1)x(2n-1) convolution whose kernel is the mean of the zero-padded kernels (286 -> 121
    taps per output at n = 6); gradients reach the individual kernels through that mean;
  * the folded series (B, T/period, period, d) already IS a channels-last image.  The 2-D convolution runs on the 1-D
    implicit-GEMM kernels (``ops.conv1d_cl``): the K kernel rows are stacked into the channel axis (K*d channels, rows as
    independent sequences), leaving a k = K convolution along the period axis;
  * period detection uses torch.fft (rocFFT) -- a library transform, as in the reference -- and its one host sync per block.
The forecasting / imputation / anomaly heads belong to tasks outside the scope contract.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ign_hip import ops
from ign_hip._lib import IgnError

from layers.Embed import DataEmbedding


def FFT_for_Period(x, k=2):
    """(B, T, C) -> the k dominant periods (host ints) and their per-sample amplitudes (TimesNet.py:9-18)."""
    xf = torch.fft.rfft(x, dim=1)
    amp = xf.abs()
    frequency_list = amp.mean(0).mean(-1)
    frequency_list[0] = 0
    _, top_list = torch.topk(frequency_list, k)
    top_list = top_list.detach().cpu().numpy()
    period = x.shape[1] // top_list
    return period, amp.mean(-1)[:, top_list]


class _RowStack(torch.autograd.Function):
    """(B, H+K-1, W, C) -> (B, H, W, K, C): out[b, h, w, r, :] = hp[b, h+r, w, :] (the K kernel rows stacked into the channel
    axis).  The backward adds the K slices back in place -- autograd's own slice gradients allocate and add a full-size zero
    tensor per slice."""

    @staticmethod
    def forward(ctx, hp, K):
        H = hp.shape[1] - K + 1
        ctx.K = K
        return torch.stack([hp[:, r:r + H] for r in range(K)], dim=3)

    @staticmethod
    def backward(ctx, g):
        K = ctx.K
        B, H, W, _, C = g.shape
        ghp = g.new_zeros(B, H + K - 1, W, C)
        for r in range(K):
            ghp[:, r:r + H] += g[:, :, :, r]
        return ghp, None


class Inception_Block_V1(nn.Module):
    def __init__(self, in_channels, out_channels, num_kernels=6, init_weight=True):
        super().__init__()
        self.in_channels, self.out_channels, self.num_kernels = in_channels, out_channels, num_kernels
        self.kernels = nn.ModuleList([nn.Conv2d(in_channels, out_channels, kernel_size=2 * i + 1, padding=i)
                                      for i in range(num_kernels)])
        if init_weight:
            for m in self.kernels:
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
                nn.init.constant_(m.bias, 0)

    def forward(self, x):                        # (B, C, H, W): the reference formulation (channel counts not divisible by 4)
        return torch.stack([conv(x) for conv in self.kernels], dim=-1).mean(-1)

    def fused_kernel(self):
        """(Co, Ci, K, K) kernel and (Co,) bias of the single convolution the block equals."""
        P = self.num_kernels - 1
        w = sum(F.pad(conv.weight, (P - i,) * 4) for i, conv in enumerate(self.kernels)) / self.num_kernels
        b = sum(conv.bias for conv in self.kernels) / self.num_kernels
        return w, b

    def forward_cl(self, h):                     # (B, H, W, C) channels-last on the hand-written kernels
        B, H, W, C = h.shape
        P, K = self.num_kernels - 1, 2 * self.num_kernels - 1
        w, b = self.fused_kernel()
        hp = F.pad(h, (0, 0, P, P, P, P))
        rows = _RowStack.apply(hp, K)                                            # (B, H, W+2P, K, C)
        w2 = w.permute(0, 2, 1, 3).reshape(self.out_channels, K * C, K)           # [co][(row r, ci)][column s]
        y = ops.conv1d_cl(rows.view(B * H, W + 2 * P, K * C), w2, b)
        return y.view(B, H, W, self.out_channels)


class TimesBlock(nn.Module):
    def __init__(self, configs):
        super().__init__()
        self.seq_len, self.pred_len, self.k = configs.seq_len, configs.pred_len, configs.top_k
        self.conv = nn.Sequential(Inception_Block_V1(configs.d_model, configs.d_ff, num_kernels=configs.num_kernels),
                                  nn.GELU(),
                                  Inception_Block_V1(configs.d_ff, configs.d_model, num_kernels=configs.num_kernels))

    def forward(self, x):
        B, T, N = x.size()
        period_list, period_weight = FFT_for_Period(x, self.k)
        total = self.seq_len + self.pred_len
        cl = x.dtype == torch.float32 and N % 4 == 0 and self.conv[0].out_channels % 4 == 0
        res = []
        for i in range(self.k):
            period = int(period_list[i])
            length = total if total % period == 0 else (total // period + 1) * period
            out = x if length == total else torch.cat([x, x.new_zeros(B, length - total, N)], dim=1)
            out = out.reshape(B, length // period, period, N)                    # channels-last image (B, H, W, C)
            if cl:
                out = self.conv[2].forward_cl(F.gelu(self.conv[0].forward_cl(out)))
            else:
                out = self.conv(out.permute(0, 3, 1, 2).contiguous()).permute(0, 2, 3, 1)
            res.append(out.reshape(B, -1, N)[:, :total, :])
        res = torch.stack(res, dim=-1)
        period_weight = F.softmax(period_weight, dim=1)
        res = torch.sum(res * period_weight[:, None, None, :], -1)
        return res + x


class Model(nn.Module):
    def __init__(self, configs):
        super().__init__()
        self.configs = configs
        self.task_name = configs.task_name
        if self.task_name not in ('classification', 'regression'):
            raise NotImplementedError("only the classification head of TimesNet is rebuilt (SURVEY section 2)")
        self.seq_len, self.label_len, self.pred_len = configs.seq_len, configs.label_len, configs.pred_len
        self.model = nn.ModuleList([TimesBlock(configs) for _ in range(configs.e_layers)])
        self.enc_embedding = DataEmbedding(configs.enc_in, configs.d_model, configs.embed, configs.freq, configs.dropout)
        self.layer = configs.e_layers
        self.layer_norm = nn.LayerNorm(configs.d_model)
        self.act = F.gelu
        self.dropout = nn.Dropout(configs.dropout)
        self.projection = nn.Linear(configs.d_model * configs.seq_len, configs.num_class)

    def classification(self, x_enc, x_mark_enc):
        if not x_enc.is_cuda:
            raise IgnError(f"TimesNet expert: tensor on {x_enc.device}; the deep experts run on the MI355X only (no CPU fallback)")
        enc_out = self.enc_embedding(x_enc, None)
        for i in range(self.layer):
            enc_out = ops.layer_norm(self.model[i](enc_out), self.layer_norm)
        output = self.dropout(self.act(enc_out))
        output = output * x_mark_enc.unsqueeze(-1)                              # zero the padded steps
        return ops.head_linear(output.reshape(output.shape[0], -1), self.projection.weight, self.projection.bias)

    def forward(self, x_enc, x_mark_enc, x_dec=None, x_mark_dec=None, mask=None):
        return self.classification(x_enc, x_mark_enc)
