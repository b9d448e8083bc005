"""Attention of the Transformer-encoder baseline (IGN/layers/SelfAttention_Family.py:48-75,179-213).

``FullAttention`` is softmax(scale * Q K^T) V without mask and with dropout 0 on every live path
(``mask_flag=False``: IGN/model/Transformer.py:30).  The reference materialises the (B,H,L,S) score tensor
(8.2 GB per layer at the benchmark shape); here the core runs as a fused flash-style fp32-MFMA kernel
(``ign_attn_fwd/bwd``) that never stores the scores.  The Prob/DS/Reformer/TwoStage variants of the reference file are
not used by any live path and are not rebuilt.
"""
from math import sqrt

import torch
import torch.nn as nn

from ign_hip import ops

class FullAttention(nn.Module):
    def __init__(self, mask_flag=True, factor=5, scale=None, attention_dropout=0.1, output_attention=False):
        super().__init__()
        self.scale = scale
        self.mask_flag = mask_flag
        self.output_attention = output_attention
        self.dropout = nn.Dropout(attention_dropout)

    def forward(self, queries, keys, values, attn_mask, tau=None, delta=None):
        B, L, H, E = queries.shape
        scale = self.scale or 1. / sqrt(E)
        if self.mask_flag or self.output_attention or (self.training and self.dropout.p > 0):
            raise NotImplementedError("causal mask / attention-map output / attention dropout are not on the "
                                      "classification path (mask_flag=False, dropout 0)")
        return ops.attention(queries, keys, values, scale), None


class AttentionLayer(nn.Module):
    def __init__(self, attention, d_model, n_heads, d_keys=None, d_values=None):
        super().__init__()
        d_keys = d_keys or (d_model // n_heads)
        d_values = d_values or (d_model // n_heads)
        self.inner_attention = attention
        self.query_projection = nn.Linear(d_model, d_keys * n_heads)
        self.key_projection = nn.Linear(d_model, d_keys * n_heads)
        self.value_projection = nn.Linear(d_model, d_values * n_heads)
        self.out_projection = nn.Linear(d_values * n_heads, d_model)
        self.n_heads = n_heads

    def forward(self, queries, keys, values, attn_mask, tau=None, delta=None):
        B, L, _ = queries.shape
        S, H = keys.shape[1], self.n_heads
        lin = lambda m, t: ops.linear(t, m.weight, m.bias)
        # (a stacked q/k/v projection was built twice -- packed output in round 1, three contiguous outputs + one K = 3d
        # input-gradient GEMM in round 2 -- and lost both times at this shape: DESIGN.md 4.4 / 4.4b)
        q = lin(self.query_projection, queries).view(B, L, H, -1)
        k = lin(self.key_projection, keys).view(B, S, H, -1)
        v = lin(self.value_projection, values).view(B, S, H, -1)
        out, attn = self.inner_attention(q, k, v, attn_mask, tau=tau, delta=delta)
        return lin(self.out_projection, out.reshape(B, L, -1)), attn
