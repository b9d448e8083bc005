"""Transformer-encoder baseline, classification head only (IGN/model/Transformer.py:17-39,69-72,99-124):
embed -> e_layers x (MHA + FFN, post-norm) -> LayerNorm -> gelu -> x mask -> flatten -> Linear(T*d -> N)."""
import torch.nn as nn
import torch.nn.functional as F

from ign_hip import ops

from layers.Embed import DataEmbedding
from layers.SelfAttention_Family import AttentionLayer, FullAttention
from layers.Transformer_EncDec import Encoder, EncoderLayer


class Model(nn.Module):
    def __init__(self, configs):
        super().__init__()
        self.task_name = configs.task_name
        if self.task_name != 'classification':
            raise NotImplementedError("only task_name='classification' is rebuilt (SURVEY section 2)")
        self.pred_len = configs.pred_len
        self.output_attention = configs.output_attention
        self.enc_embedding = DataEmbedding(configs.enc_in, configs.d_model, configs.embed, configs.freq, configs.dropout)
        self.encoder = Encoder(
            [EncoderLayer(
                AttentionLayer(FullAttention(False, configs.factor, attention_dropout=configs.dropout,
                                             output_attention=configs.output_attention),
                               configs.d_model, configs.n_heads),
                configs.d_model, configs.d_ff, dropout=configs.dropout, activation=configs.activation)
             for _ in range(configs.e_layers)],
            norm_layer=nn.LayerNorm(configs.d_model))
        self.act = F.gelu
        self.dropout = nn.Dropout(configs.dropout)
        self.projection = nn.Linear(configs.d_model * configs.seq_len, configs.num_class)

    def classification(self, x_enc, x_mark_enc):
        enc_out = self.enc_embedding(x_enc, None)
        enc_out, _ = self.encoder(enc_out, attn_mask=None)
        out = self.dropout(self.act(enc_out))
        out = out * x_mark_enc.unsqueeze(-1)            # zero the padded steps
        return ops.head_linear(out.reshape(out.shape[0], -1), self.projection.weight, self.projection.bias)

    def forward(self, x_enc, x_mark_enc, x_dec=None, x_mark_dec=None, mask=None):
        return self.classification(x_enc, x_mark_enc)
