"""FCN deep expert (IGN/model/FullyConvNet.py:7-59): 3 x (Conv1d + BatchNorm1d + ReLU) -> GAP -> Linear.

State-dict keys match the reference (``block{1,2,3}.{0,1}.*``, ``fc.*``).  On the GPU the three blocks and the pool run as
one autograd node over the hand-written channels-last implicit-GEMM kernels (ign_hip/fcn.py, csrc/ign_clconv_{f32,x6}.hip, ign_bn.hip): the
loader's (B, T, C) batch IS the GEMM operand (the im2col row of (b, t) is x[b, t:t+k, :], contiguous), BatchNorm statistics
come out of the GEMM epilogue and BatchNorm's affine + ReLU are applied while the next GEMM stages its operand.
Inside a ``torch.autocast(bfloat16)`` region (the reference's default mode: ``--amp`` switches it OFF) the same kernels run with
their operands rounded to bf16 and ONE product per MFMA step (``fcn.CONV_MATH`` "bf16"), activations and BatchNorm stay fp32.
``IGN_FCN_MIOPEN=1`` routes the convolutions through torch's MIOpen backend instead (A/B measurements only).
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ign_hip import fcn, ops

# IGN_FCN_MIOPEN=1 routes the convolutions through torch's MIOpen backend instead (A/B measurements only)
_USE_CLCONV = os.environ.get("IGN_FCN_MIOPEN", "0") != "1"


class FullyConvNetwork(nn.Module):
    def __init__(self, configs):
        super().__init__()
        k1, k2, k3 = (3, 3, 2) if configs.seq_len <= 10 else (8, 5, 3)     # FullyConvNet.py:11-50
        self.block1 = nn.Sequential(nn.Conv1d(configs.enc_in, 128, k1), nn.BatchNorm1d(128), nn.ReLU())
        self.block2 = nn.Sequential(nn.Conv1d(128, 256, k2), nn.BatchNorm1d(256), nn.ReLU())
        self.block3 = nn.Sequential(nn.Conv1d(256, 128, k3), nn.BatchNorm1d(128), nn.ReLU())
        self.pooling = nn.AdaptiveAvgPool1d(1)
        self.fc = nn.Linear(128, configs.num_class)

    @staticmethod
    def _block(block, h):
        conv, bn = block[0], block[1]
        h = F.conv2d(h, conv.weight.unsqueeze(2), conv.bias)
        if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
            bn.num_batches_tracked.add_(1)
        h = F.batch_norm(h, bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.training or not bn.track_running_stats,
                         bn.momentum, bn.eps)
        return F.relu(h)

    def forward(self, x, x_mark_enc=None, x_dec=None, x_mark_dec=None, mask=None, x_bct=None):
        if not x.is_cuda:
            raise ops._lib.IgnError(f"FCN expert: tensor on {x.device}; the deep experts run on the MI355X only (no CPU fallback)")
        if x.dtype == torch.float32:
            if _USE_CLCONV:
                # x (B,T,C) is already the channels-last operand of the implicit GEMM: no transpose, no im2col
                pooled = fcn.fcn_body(x, [(b[0], b[1]) for b in (self.block1, self.block2, self.block3)])
            else:
                h = x.permute(0, 2, 1).unsqueeze(2)              # (B,C,1,T) view, channels-last strides, no copy
                h = self._block(self.block3, self._block(self.block2, self._block(self.block1, h)))
                pooled = h.mean(dim=(2, 3))
        else:
            h = x_bct if x_bct is not None else x.permute(0, 2, 1)
            h = self.block3(self.block2(self.block1(h)))
            pooled = self.pooling(h).flatten(start_dim=1)
        return ops.head_linear(pooled, self.fc.weight, self.fc.bias)
