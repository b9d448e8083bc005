"""PatchTST deep expert, classification head (IGN/model/PatchTST.py:23-76,165-189,207-212): per-series standardisation ->
patches (len 16, stride 8) -> Linear + positional table -> e_layers x (MHA + FFN, post-norm) over the B*C patch sequences ->
LayerNorm -> flatten (C * d_model * patch_num) -> Linear.

State-dict keys match the reference (``patch_embedding.{value_embedding,position_embedding}``, ``encoder.*``, ``projection``).
The encoder is the one the Transformer baseline uses (layers/): attention on ign_attn_*, dense layers on ``ops.linear``
(split-bf16 GEMM kernels), the class projection on the streaming head kernel.  The forecasting / imputation / anomaly heads of
the reference file belong to tasks outside the scope contract and are not rebuilt.
"""
import torch
import torch.nn as nn

from ign_hip import ops
from ign_hip._lib import IgnError

from layers.Embed import PatchEmbedding
from layers.SelfAttention_Family import AttentionLayer, FullAttention
from layers.Transformer_EncDec import Encoder, EncoderLayer


class Model(nn.Module):
    def __init__(self, configs, patch_len=16, stride=8):
        super().__init__()
        self.task_name = configs.task_name
        if self.task_name not in ('classification', 'regression'):
            raise NotImplementedError("only the classification head of PatchTST is rebuilt (SURVEY section 2)")
        self.seq_len = configs.seq_len
        self.pred_len = configs.pred_len
        padding = stride
        self.patch_embedding = PatchEmbedding(configs.d_model, patch_len, stride, padding, configs.dropout)
        self.encoder = Encoder(
            [EncoderLayer(
                AttentionLayer(FullAttention(False, configs.factor, attention_dropout=configs.dropout,
                                             output_attention=configs.output_attention),
                               configs.d_model, configs.n_heads),
                configs.d_model, configs.d_ff, dropout=configs.dropout, activation=configs.activation)
             for _ in range(configs.e_layers)],
            norm_layer=nn.LayerNorm(configs.d_model))
        self.head_nf = configs.d_model * int((configs.seq_len - patch_len) / stride + 2)
        self.flatten = nn.Flatten(start_dim=-2)
        self.dropout = nn.Dropout(configs.dropout)
        self.projection = nn.Linear(self.head_nf * configs.enc_in, configs.num_class)

    def classification(self, x_enc, x_mark_enc):
        if not x_enc.is_cuda:
            raise IgnError(f"PatchTST expert: tensor on {x_enc.device}; the deep experts run on the MI355X only (no CPU fallback)")
        # per-series standardisation over time (biased variance + 1e-5), statistics detached (PatchTST.py:167-171)
        means = x_enc.mean(1, keepdim=True).detach()
        x_enc = x_enc - means
        stdev = torch.sqrt(torch.var(x_enc, dim=1, keepdim=True, unbiased=False) + 1e-5)
        x_enc = x_enc / stdev
        x_enc = x_enc.permute(0, 2, 1)                                   # (B, C, T)
        enc_out, n_vars = self.patch_embedding(x_enc)                    # (B*C, P, d)
        enc_out, _ = self.encoder(enc_out)
        enc_out = torch.reshape(enc_out, (-1, n_vars, enc_out.shape[-2], enc_out.shape[-1]))
        enc_out = enc_out.permute(0, 1, 3, 2)                            # (B, C, d, P)
        output = self.dropout(self.flatten(enc_out))
        output = output.reshape(output.shape[0], -1)
        return ops.head_linear(output, self.projection.weight, self.projection.bias)

    def forward(self, x_enc, x_mark_enc=None, x_dec=None, x_mark_dec=None, mask=None):
        return self.classification(x_enc, x_mark_enc)
