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
        # ops.relu / ops.gelu: torch's element-wise kernels plus the magnitude bound handed on in both directions (fp16 GEMM scaling)
        self.activation = ops.relu if activation == "relu" else ops.gelu

    def forward(self, x, attn_mask=None, tau=None, delta=None):
        new_x, attn = self.attention(x, x, x, attn_mask=attn_mask, tau=tau, delta=delta)
        x = ops.layer_norm(x, self.norm1, residual=self.dropout(new_x))       # LayerNorm(x + attention(x)): the add rides on the pass
        # the k=1 convolutions are plain GEMMs over (B*T, d): apply them without the two transposes
        u = ops.linear(x, self.conv1.weight.squeeze(-1), self.conv1.bias)
        drop = self.training and self.dropout.p > 0
        if self.activation is ops.gelu and not drop:
            # gelu + second dense layer as one node: the activation's backward runs in the epilogue of the input-gradient GEMM
            y = ops.gelu_linear(u, self.conv2.weight.squeeze(-1), self.conv2.bias)
        else:
            y = self.activation(u)
            if drop:
                y = self.dropout(y)
            y = ops.linear(y, self.conv2.weight.squeeze(-1), self.conv2.bias)
        y = self.dropout(y)
        return ops.layer_norm(x, self.norm2, residual=y), attn


class Encoder(nn.Module):
    def __init__(self, attn_layers, conv_layers=None, norm_layer=None):
        super().__init__()
        if conv_layers is not None:
            raise NotImplementedError("distilling conv layers are not used by the classification path")
        self.attn_layers = nn.ModuleList(attn_layers)
        self.conv_layers = None
        self.norm = norm_layer

    def dense_weights(self):
        """The weight matrices of every dense layer of the stack, as `ops.linear` will see them (k = 1 convolutions squeezed)."""
        ws = []
        for layer in self.attn_layers:
            a = layer.attention
            for name in ("query_projection", "key_projection", "value_projection", "out_projection"):
                m = getattr(a, name, None)
                if isinstance(m, nn.Linear):
                    ws.append(m.weight)
            ws += [layer.conv1.weight.squeeze(-1), layer.conv2.weight.squeeze(-1)]
        return ws

    def forward(self, x, attn_mask=None, tau=None, delta=None):
        # bounds and packed planes of all dense-layer weights in one scan + a few packing launches (ops.prepare_linear_weights)
        ops.prepare_linear_weights(self.dense_weights(), need_dx=torch.is_grad_enabled())
        attns = []
        for layer in self.attn_layers:
            x, attn = layer(x, attn_mask=attn_mask, tau=tau, delta=delta)
            attns.append(attn)
        if self.norm is not None:
            x = ops.layer_norm(x, self.norm) if isinstance(self.norm, nn.LayerNorm) else self.norm(x)
        return x, attns
