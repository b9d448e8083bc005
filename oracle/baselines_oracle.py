"""CPU oracle for the two baselines of BASELINE.json configs 3 / 4 -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module; nothing under
``speech-imagery-eeg_amd/`` does.  FUNCTIONAL restatements: a state dict with the reference's key names and an input go in,
logits come out (plain torch fp32 ops on the CPU, autograd-able so a training step can be timed).  Pinned by fixtures the
reference itself produced (``tests/golden/make_golden.py``): ``transformer_bm``, ``transformer_ch512``, ``eegcnn_ch``,
``eegcnn_ch512`` in ``tests/test_oracle_golden.py``.

Reference lines followed (``IGN/`` = InterpretGatedNetwork/):
  Transformer.Model.classification      IGN/model/Transformer.py:99-110 (embed -> encoder -> gelu -> mask -> flatten -> Linear)
  DataEmbedding / TokenEmbedding         IGN/layers/Embed.py:29-42,109-126 (k=3 circular Conv1d, no bias, + sin/cos table)
  Encoder / EncoderLayer                 IGN/layers/Transformer_EncDec.py:27-80 (post-norm, 1x1-conv feed-forward)
  AttentionLayer / FullAttention         IGN/layers/SelfAttention_Family.py:179-213, 56-75
  EEGcnn.forward                         IGN/model/eegcnn.py:85-108 (layers :67-83)
  EEGCNNTransformer._original_forward    IGN/model/eegcnn.py:248-309; PositionalEncoding :11-27 (batch-index code, D10)
  nn.TransformerEncoderLayer             torch (post-norm, relu, batch_first; packed in_proj) -- third-party, un-pinned
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

from oracle.experts_oracle import _encoder, _layer_norm, _sincos_table, _token_embedding

Tensors = Dict[str, torch.Tensor]


def transformer_logits(p: Tensors, x_btc: torch.Tensor, pad_mask: torch.Tensor, n_layers: int, n_heads: int) -> torch.Tensor:
    """(B,T,C), mask (B,T) -> (B,N).  Dropout 0 (run_uea.sh's setting); the unused temporal embedding is not evaluated."""
    h = _token_embedding(p, "enc_embedding", x_btc)
    h = _encoder(p, "encoder", h, n_layers, n_heads, gelu=True)
    out = F.gelu(h) * pad_mask[:, :, None].to(h.dtype)
    return out.reshape(out.shape[0], -1) @ p["projection.weight"].t() + p["projection.bias"]


def _bn2d(p: Tensors, pre: str, h: torch.Tensor, training: bool, stats_out: Tensors | None) -> torch.Tensor:
    """BatchNorm2d over (B, F, H, W): batch statistics (biased variance) in training, running statistics otherwise."""
    w, b = p[pre + ".weight"], p[pre + ".bias"]
    if training:
        mean = h.mean(dim=(0, 2, 3))
        var = h.var(dim=(0, 2, 3), unbiased=False)
        if stats_out is not None:
            n = h.numel() // h.shape[1]
            stats_out[pre + ".running_mean"] = 0.9 * p[pre + ".running_mean"] + 0.1 * mean.detach()
            stats_out[pre + ".running_var"] = 0.9 * p[pre + ".running_var"] + 0.1 * var.detach() * n / max(n - 1, 1)
    else:
        mean, var = p[pre + ".running_mean"], p[pre + ".running_var"]
    s = (1, -1, 1, 1)
    return (h - mean.view(s)) / torch.sqrt(var.view(s) + 1e-5) * w.view(s) + b.view(s)


def _same_pad(k: int):
    """PyTorch padding='same' for stride 1: total k-1, the extra element on the right."""
    return (k - 1) // 2, (k - 1) - (k - 1) // 2


def eegcnn_features(p: Tensors, x_bct: torch.Tensor, training: bool, stats_out: Tensors | None = None, pre: str = "eegcnn",
                    P1: int = 2, P2: int = 5) -> torch.Tensor:
    """EEGcnn.forward (eegcnn.py:85-108): (B,C,T) -> (B,F2,T/(P1*P2)).  Evaluated block by block on chunks of filters so the
    (B,F1,C,T) intermediate never exceeds one filter's worth per step would change nothing numerically -- it is simply
    materialised here, as the reference does."""
    h = x_bct.unsqueeze(1)                                                        # (B,1,C,T)
    w1 = p[pre + ".block1_conv1.weight"]                                          # (F1,1,1,k1)
    h = F.conv2d(F.pad(h, _same_pad(w1.shape[-1])), w1)
    h = _bn2d(p, pre + ".block1_bn1", h, training, stats_out)
    w2 = p[pre + ".block1_depthwise.weight"]                                      # (D*F1,1,C,1), groups F1
    h = F.conv2d(h, w2, groups=w1.shape[0])                                       # (B,D*F1,1,T)
    h = F.elu(_bn2d(p, pre + ".block1_bn2", h, training, stats_out))
    h = F.avg_pool2d(h, (1, P1))
    w3 = p[pre + ".block2_conv1.weight"]                                          # (DF1,1,1,k2) depthwise
    h = F.conv2d(F.pad(h, _same_pad(w3.shape[-1])), w3, groups=w3.shape[0])
    h = F.conv2d(h, p[pre + ".block2_conv2.weight"])                              # pointwise
    h = F.elu(_bn2d(p, pre + ".block2_bn", h, training, stats_out))
    h = F.avg_pool2d(h, (1, P2))
    return h.squeeze(2)


def _torch_encoder_layer(p: Tensors, pre: str, h: torch.Tensor, n_heads: int) -> torch.Tensor:
    """nn.TransformerEncoderLayer(batch_first=True, norm_first=False, activation=relu), dropout 0, no mask."""
    B, S, D = h.shape
    E = D // n_heads
    qkv = h @ p[pre + ".self_attn.in_proj_weight"].t() + p[pre + ".self_attn.in_proj_bias"]
    q, k, v = (t.view(B, S, n_heads, E) for t in qkv.split(D, dim=-1))
    a = torch.softmax(torch.einsum("blhe,bshe->bhls", q, k) / math.sqrt(E), dim=-1)
    o = torch.einsum("bhls,bshe->blhe", a, v).reshape(B, S, D)
    o = o @ p[pre + ".self_attn.out_proj.weight"].t() + p[pre + ".self_attn.out_proj.bias"]
    h = _layer_norm(p, pre + ".norm1", h + o)
    y = F.relu(h @ p[pre + ".linear1.weight"].t() + p[pre + ".linear1.bias"])
    y = y @ p[pre + ".linear2.weight"].t() + p[pre + ".linear2.bias"]
    return _layer_norm(p, pre + ".norm2", h + y)


def eegcnn_logits(p: Tensors, x_bct: torch.Tensor, training: bool, n_layers: int = 2, n_heads: int = 8,
                  stats_out: Tensors | None = None) -> torch.Tensor:
    """EEGCNNTransformer._original_forward with the working semantics of SURVEY D9 ((B,C,T) input, no mask => all-ones) and
    the batch-index positional code of D10: pe[:B] has shape (B,1,D) and is broadcast over TIME."""
    f = eegcnn_features(p, x_bct, training, stats_out)                            # (B,F2,S)
    h = f.permute(0, 2, 1)
    if "cnn_projection.weight" in p:
        h = h @ p["cnn_projection.weight"].t() + p["cnn_projection.bias"]
    B, S, D = h.shape
    h = h + _sincos_table(B, D, h).unsqueeze(1)                                   # row b of the table for sample b (D10)
    for i in range(n_layers):
        h = _torch_encoder_layer(p, f"transformer_encoder.layers.{i}", h, n_heads)
    pooled = h.sum(dim=1) / float(S)                                              # masked mean with an all-ones mask
    return pooled @ p["classifier.weight"].t() + p["classifier.bias"]
