"""Input embedding of the Transformer-encoder baseline (IGN/layers/Embed.py:8-42,109-126).

Only what the classification path touches: ``DataEmbedding`` = ``TokenEmbedding`` (Conv1d C->d, k=3, circular,
no bias) + ``PositionalEmbedding`` (fixed sin/cos table).  The reference also constructs a temporal embedding it
never calls on this path (``x_mark`` is None); it is kept so state_dict keys match (``temporal_embedding.embed.weight``).
"""
import math

import torch
import torch.nn as nn


class PositionalEmbedding(nn.Module):
    def __init__(self, d_model, max_len=5000):
        super().__init__()
        pos = torch.arange(0, max_len).float().unsqueeze(1)
        div = (torch.arange(0, d_model, 2).float() * -(math.log(10000.0) / d_model)).exp()
        pe = torch.zeros(max_len, d_model)
        pe[:, 0::2] = torch.sin(pos * div)
        pe[:, 1::2] = torch.cos(pos * div)
        self.register_buffer('pe', pe.unsqueeze(0))

    def forward(self, x):
        return self.pe[:, :x.size(1)]


class TokenEmbedding(nn.Module):
    def __init__(self, c_in, d_model):
        super().__init__()
        self.tokenConv = nn.Conv1d(c_in, d_model, kernel_size=3, padding=1, padding_mode='circular', bias=False)
        nn.init.kaiming_normal_(self.tokenConv.weight, mode='fan_in', nonlinearity='leaky_relu')

    def forward(self, x):                       # (B,T,C) -> (B,T,d)
        if not x.is_cuda:
            from ign_hip._lib import IgnError
            raise IgnError(f"token embedding: tensor on {x.device}; the product path runs on the MI355X only (no CPU fallback)")
        if x.dtype == torch.float32 and self.tokenConv.out_channels % 4 == 0:
            # circular padding = one wrapped row per side; then a valid channels-last convolution on the implicit-GEMM kernels
            from ign_hip import ops
            return ops.conv1d_cl(torch.cat([x[:, -1:], x, x[:, :1]], dim=1), self.tokenConv.weight)
        return self.tokenConv(x.permute(0, 2, 1)).transpose(1, 2)


class TimeFeatureEmbedding(nn.Module):
    """Unused on the classification path; present for state_dict compatibility (Embed.py:95-106)."""
    _D_INP = {'h': 4, 't': 5, 's': 6, 'm': 1, 'a': 1, 'w': 2, 'd': 3, 'b': 3}

    def __init__(self, d_model, embed_type='timeF', freq='h'):
        super().__init__()
        self.embed = nn.Linear(self._D_INP[freq], d_model, bias=False)

    def forward(self, x):
        return self.embed(x)


class DataEmbedding(nn.Module):
    def __init__(self, c_in, d_model, embed_type='fixed', freq='h', dropout=0.1):
        super().__init__()
        self.value_embedding = TokenEmbedding(c_in=c_in, d_model=d_model)
        self.position_embedding = PositionalEmbedding(d_model=d_model)
        if embed_type != 'timeF':
            raise NotImplementedError("only embed='timeF' (the run.py default) is part of the classification path")
        self.temporal_embedding = TimeFeatureEmbedding(d_model=d_model, embed_type=embed_type, freq=freq)
        self.dropout = nn.Dropout(p=dropout)

    def forward(self, x, x_mark):
        out = self.value_embedding(x) + self.position_embedding(x)
        if x_mark is not None:
            out = out + self.temporal_embedding(x_mark)
        return self.dropout(out)


class PatchEmbedding(nn.Module):
    """Patching + value projection + positional table of PatchTST (IGN/layers/Embed.py:165-190): (B, C, T) ->
    (B*C, patch_num, d_model).  The right edge is replicated `padding` steps, patches of `patch_len` are taken every `stride`
    steps; the projection Linear(patch_len -> d_model, no bias) runs on ``ops.linear``."""

    def __init__(self, d_model, patch_len, stride, padding, dropout):
        super().__init__()
        self.patch_len = patch_len
        self.stride = stride
        self.padding_patch_layer = nn.ReplicationPad1d((0, padding))
        self.value_embedding = nn.Linear(patch_len, d_model, bias=False)
        self.position_embedding = PositionalEmbedding(d_model)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x):
        from ign_hip import ops
        n_vars = x.shape[1]
        x = self.padding_patch_layer(x)
        x = x.unfold(dimension=-1, size=self.patch_len, step=self.stride)
        x = torch.reshape(x, (x.shape[0] * x.shape[1], x.shape[2], x.shape[3]))
        x = ops.linear(x, self.value_embedding.weight) + self.position_embedding(x)
        return self.dropout(x), n_vars
