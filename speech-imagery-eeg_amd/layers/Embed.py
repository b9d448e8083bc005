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
