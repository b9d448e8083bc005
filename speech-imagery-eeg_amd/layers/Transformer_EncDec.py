"""Encoder stack of the Transformer baseline (IGN/layers/Transformer_EncDec.py:27-80): post-norm layers with a
1x1-Conv1d feed-forward, and a final LayerNorm.  Decoder / ConvLayer of the reference file are unused by the
classification path and not rebuilt."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ign_hip import ops


class EncoderLayer(nn.Module):
    def __init__(self, attention, d_model, d_ff=None, dropout=0.1, activation="relu"):
        super().__init__()
        d_ff = d_ff or 4 * d_model
        self.attention = attention
        self.conv1 = nn.Conv1d(in_channels=d_model, out_channels=d_ff, kernel_size=1)
        self.conv2 = nn.Conv1d(in_channels=d_ff, out_channels=d_model, kernel_size=1)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.dropout = nn.Dropout(dropout)
        self.activation = F.relu if activation == "relu" else F.gelu

    def _fused(self, x):
        """the whole layer as one autograd node (ops.EncoderLayerFn) when it is the plain self-attention layer of the
        classification path: FullAttention without mask / attention maps, no active dropout, shapes the kernels cover"""
        from layers.SelfAttention_Family import AttentionLayer, FullAttention
        al = self.attention
        if not isinstance(al, AttentionLayer) or not isinstance(al.inner_attention, FullAttention) or torch.is_autocast_enabled():
            return None
        fa = al.inner_attention
        if fa.mask_flag or fa.output_attention or fa.scale is not None:
            return None
        drop = self.training and (self.dropout.p > 0 or fa.dropout.p > 0)
        d = x.shape[-1]
        params = (al.query_projection.weight, al.query_projection.bias, al.key_projection.weight, al.key_projection.bias,
                  al.value_projection.weight, al.value_projection.bias, al.out_projection.weight, al.out_projection.bias,
                  self.norm1.weight, self.norm1.bias, self.conv1.weight.squeeze(-1), self.conv1.bias,
                  self.conv2.weight.squeeze(-1), self.conv2.bias, self.norm2.weight, self.norm2.bias)
        if al.query_projection.out_features != d or al.value_projection.out_features != d or self.conv1.kernel_size != (1,) \
                or not ops.encoder_layer_supported(x, self.conv1.out_channels, al.n_heads, params, (self.norm1, self.norm2), drop):
            return None
        return ops.encoder_layer(x, al.n_heads, 1.0 / (d // al.n_heads) ** 0.5, self.activation is F.gelu, False, self.norm1.eps,
                                 self.norm2.eps, *params)

    def forward(self, x, attn_mask=None, tau=None, delta=None):
        if attn_mask is None and tau is None and delta is None:
            y = self._fused(x)
            if y is not None:
                return y, None
        new_x, attn = self.attention(x, x, x, attn_mask=attn_mask, tau=tau, delta=delta)
        x = ops.layer_norm(x + self.dropout(new_x), self.norm1)
        # the k=1 convolutions are plain GEMMs over (B*T, d): apply them without the two transposes
        y = ops.linear(x, self.conv1.weight.squeeze(-1), self.conv1.bias)
        y = self.dropout(self.activation(y))
        y = self.dropout(ops.linear(y, self.conv2.weight.squeeze(-1), self.conv2.bias))
        return ops.layer_norm(x + y, self.norm2), attn


class Encoder(nn.Module):
    def __init__(self, attn_layers, conv_layers=None, norm_layer=None):
        super().__init__()
        if conv_layers is not None:
            raise NotImplementedError("distilling conv layers are not used by the classification path")
        self.attn_layers = nn.ModuleList(attn_layers)
        self.conv_layers = None
        self.norm = norm_layer

    def forward(self, x, attn_mask=None, tau=None, delta=None):
        attns = []
        for layer in self.attn_layers:
            x, attn = layer(x, attn_mask=attn_mask, tau=tau, delta=delta)
            attns.append(attn)
        if self.norm is not None:
            x = ops.layer_norm(x, self.norm) if isinstance(self.norm, nn.LayerNorm) else self.norm(x)
        return x, attns
