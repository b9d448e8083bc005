"""FCN deep expert (IGN/model/FullyConvNet.py:7-59): 3 x (Conv1d + BatchNorm1d + ReLU) -> GAP -> Linear.

State-dict keys match the reference (``block{1,2,3}.{0,1}.*``, ``fc.*``).  Interim status: the
convolutions run through torch's ROCm backend (MIOpen); the hand-written implicit-GEMM MFMA kernel of
SURVEY K7 is the next kernel on this path.  ``x_bct`` lets InterpGN hand over the (B,C,T) transpose that
``ign_instnorm_fwd`` already produced instead of permuting again.
"""
import torch.nn as nn

from ign_hip import ops


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
        h = x_bct if x_bct is not None else x.permute(0, 2, 1)
        h = self.block3(self.block2(self.block1(h)))
        return ops.head_linear(self.pooling(h).flatten(start_dim=1), self.fc.weight, self.fc.bias)
