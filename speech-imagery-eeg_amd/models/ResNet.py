"""ResNet deep expert (IGN/model/ResNet.py:6-83): stem Conv1d(k=7, stride 2) + BN + ReLU + MaxPool(3, 2), three BasicBlocks
(64 -> 64 -> 128 -> 128, stride 1, 1x1-conv shortcut where the width changes), global average pool, Linear.

State-dict keys match the reference (``conv1, bn1, layer{1,2,3}.0.{conv1,bn1,conv2,bn2,shortcut.{0,1}}, fc``).  On the GPU
every convolution runs channels-last on the implicit-GEMM kernels of the FCN expert (``ops.conv1d_cl`` -> ign_clconv_*):
the loader's (B, T, C) batch is the operand as it is, the k=3 / padding=1 convolutions read a zero-row-padded copy, the
stride-2 stem is a stride-1 k=4 convolution over the (B, T/2, 2C) space-to-depth VIEW of the padded input (weights
interleaved to match; no gather), the 1x1 shortcut is ``ops.linear``.  BatchNorm runs on the (B*T, C) view (same statistics as
BatchNorm1d over (B, C, T)); BatchNorm / ReLU / max-pool / residual add are torch element-wise ops on tensors of <= 33 MB.
Inside a torch.autocast(bfloat16) region (the reference's default mode) the same kernels run in their single-product bf16 form
(``ops._gemm``); BatchNorm and the activations stay fp32.  CPU tensors are refused (no CPU fallback).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ign_hip import ops
from ign_hip._lib import IgnError


def _bn_cl(bn, h):
    """BatchNorm1d over a channels-last (B, T, C) tensor: the (B*T, C) view has the same per-channel statistics."""
    B, T, C = h.shape
    return bn(h.reshape(B * T, C)).view(B, T, C)


def _conv3_cl(conv, h):
    """Conv1d(k=3, padding=1, bias=False) channels-last."""
    return ops.conv1d_cl(F.pad(h, (0, 0, 1, 1)), conv.weight)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, in_channels, out_channels, stride=1):
        super().__init__()
        self.conv1 = nn.Conv1d(in_channels, out_channels, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm1d(out_channels)
        self.conv2 = nn.Conv1d(out_channels, out_channels, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn2 = nn.BatchNorm1d(out_channels)
        self.relu = nn.ReLU(inplace=True)
        self.stride = stride
        self.shortcut = nn.Identity()
        if stride != 1 or in_channels != out_channels:
            self.shortcut = nn.Sequential(nn.Conv1d(in_channels, out_channels, kernel_size=1, stride=stride, bias=False),
                                          nn.BatchNorm1d(out_channels))

    def forward(self, h):                        # nn.Sequential entry point
        return self.forward_cl(h)

    def forward_cl(self, h):                     # (B, T, C) channels-last on the hand-written kernels
        if self.stride != 1:
            raise NotImplementedError("the reference builds every BasicBlock with stride 1 (ResNet.py:52-54)")
        if isinstance(self.shortcut, nn.Identity):
            identity = h
        else:
            identity = _bn_cl(self.shortcut[1], ops.linear(h, self.shortcut[0].weight.squeeze(-1)))
        out = F.relu(_bn_cl(self.bn1, _conv3_cl(self.conv1, h)))
        out = _bn_cl(self.bn2, _conv3_cl(self.conv2, out))
        return F.relu(out + identity)


class Model(nn.Module):
    def __init__(self, configs):
        super().__init__()
        self.in_channels = 64
        self.conv1 = nn.Conv1d(configs.enc_in, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm1d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool1d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(BasicBlock, 64, 1)
        self.layer2 = self._make_layer(BasicBlock, 128, 1, stride=1)
        self.layer3 = self._make_layer(BasicBlock, 128, 1, stride=1)
        self.avgpool = nn.AdaptiveAvgPool1d(1)
        self.fc = nn.Linear(128 * BasicBlock.expansion, configs.num_class)

    def _make_layer(self, block, out_channels, blocks, stride=1):
        layers = [block(self.in_channels, out_channels, stride)]
        self.in_channels = out_channels * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.in_channels, out_channels))
        return nn.Sequential(*layers)

    def _stem_cl(self, x):
        """Conv1d(C -> 64, k=7, stride 2, padding 3) as a stride-1 k=4 convolution over the space-to-depth view:
        y[t] = sum_{j<8} w8[:, :, j] x_p[2t + j] = sum_{j'<4} W2[:, :, j'] X2[t + j'] with X2[u] = (x_p[2u], x_p[2u+1])."""
        B, T, C = x.shape
        T1 = (T + 6 - 7) // 2 + 1
        Tp = 2 * T1 + 6
        xp = F.pad(x, (0, 0, 3, Tp - T - 3))
        w8 = F.pad(self.conv1.weight, (0, 1))                                   # (64, C, 8), tap 7 = 0
        w2 = w8.view(64, C, 4, 2).permute(0, 3, 1, 2).reshape(64, 2 * C, 4)      # [co][(parity, ci)][j'] = w8[co][ci][2j'+parity]
        return ops.conv1d_cl(xp.view(B, Tp // 2, 2 * C), w2)                    # (B, T1, 64)

    @staticmethod
    def _maxpool_cl(h):
        """MaxPool1d(3, stride 2, padding 1) over time, channels-last.  h >= 0 (it follows a ReLU), so zero padding equals the
        reference's -inf padding; ties only occur at 0, where the ReLU below passes no gradient either way."""
        B, T1, C = h.shape
        T2 = (T1 - 1) // 2 + 1
        hp = F.pad(h, (0, 0, 1, 1))
        return torch.maximum(torch.maximum(hp[:, 0:2 * T2 - 1:2], hp[:, 1:2 * T2:2]), hp[:, 2:2 * T2 + 1:2])

    def forward(self, x, x_mark_enc=None, x_dec=None, x_mark_dec=None, mask=None):
        if not x.is_cuda:
            raise IgnError(f"ResNet expert: tensor on {x.device}; the deep experts run on the MI355X only (no CPU fallback)")
        h = F.relu(_bn_cl(self.bn1, self._stem_cl(x.float())))
        h = self._maxpool_cl(h)
        for layer in (self.layer1, self.layer2, self.layer3):
            for blk in layer:
                h = blk.forward_cl(h)
        pooled = h.mean(dim=1)
        return ops.head_linear(pooled, self.fc.weight, self.fc.bias)
