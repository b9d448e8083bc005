"""FCN deep expert (IGN/model/FullyConvNet.py:7-59): 3 x (Conv1d + BatchNorm1d + ReLU) -> GAP -> Linear.

State-dict keys match the reference (``block{1,2,3}.{0,1}.*``, ``fc.*``).  On the GPU the three blocks and the pool run as
one autograd node over the hand-written channels-last implicit-GEMM kernels (ign_hip/fcn.py, csrc/ign_clconv_{f32,x6}.hip, ign_bn.hip): the
loader's (B, T, C) batch IS the GEMM operand (the im2col row of (b, t) is x[b, t:t+k, :], contiguous), BatchNorm statistics
come out of the GEMM epilogue and BatchNorm's affine + ReLU are applied while the next GEMM stages its operand.
Inside a ``torch.autocast(bfloat16)`` region (the reference's default mode: ``--amp`` switches it OFF) the same kernels run with
their operands rounded to bf16 and ONE product per MFMA step (``fcn.CONV_MATH`` "bf16"), activations and BatchNorm stay fp32.
(The MIOpen A/B comparison of round 1 lives in tests/diag_conv_speed.py; the product module has no library route.)
"""
import torch
import torch.nn as nn

from ign_hip import fcn, ops


class FullyConvNetwork(nn.Module):
    def __init__(self, configs):
        super().__init__()
        k1, k2, k3 = (3, 3, 2) if configs.seq_len <= 10 else (8, 5, 3)     # FullyConvNet.py:11-50
        self.block1 = nn.Sequential(nn.Conv1d(configs.enc_in, 128, k1), nn.BatchNorm1d(128), nn.ReLU())
        self.block2 = nn.Sequential(nn.Conv1d(128, 256, k2), nn.BatchNorm1d(256), nn.ReLU())
        self.block3 = nn.Sequential(nn.Conv1d(256, 128, k3), nn.BatchNorm1d(128), nn.ReLU())
        self.pooling = nn.AdaptiveAvgPool1d(1)
        self.fc = nn.Linear(128, configs.num_class)

    def forward(self, x, x_mark_enc=None, x_dec=None, x_mark_dec=None, mask=None, x_bct=None):
        if not x.is_cuda:
            raise ops._lib.IgnError(f"FCN expert: tensor on {x.device}; the deep experts run on the MI355X only (no CPU fallback)")
        # x (B,T,C) is already the channels-last operand of the implicit GEMM: no transpose, no im2col.  The kernels take fp32
        # activations (inside an autocast region they round the GEMM operands to bf16 themselves): other dtypes are cast, never
        # handed to a library convolution
        x = x if x.dtype == torch.float32 else x.float()
        blocks = [(b[0], b[1]) for b in (self.block1, self.block2, self.block3)]
        if self.fc.weight.shape[0] <= 16 and not torch.is_autocast_enabled():
            return fcn.fcn_body(x, blocks, head=self.fc)            # class head inside the pooling launch
        return ops.head_linear(fcn.fcn_body(x, blocks), self.fc.weight, self.fc.bias)
