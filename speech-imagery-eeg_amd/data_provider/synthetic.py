"""Synthetic CHISCO-shaped EEG (``--data SYNTH``): the workload BASELINE.json is quoted on.

X ~ N(0,1) i.i.d. float32 in loader layout (n, T, C), labels uniform; optional "planted" variant adds a
class-specific sine burst to a few channels so the loss visibly decreases.  Items follow the UEA item
contract (x[T,C], y[1]) so the same ``collate_fn`` applies.  Generated with a CPU ``torch.Generator`` so every
rank and the CPU oracle see bit-identical values.
"""
import math

import torch
from torch.utils.data import Dataset


class SyntheticEEG(Dataset):
    SPLIT_SEEDS = {'train': 1234, 'val': 2345, 'test': 3456}

    def __init__(self, root_path=None, flag='train', n=None, seq_len=1000, enc_in=122, num_classes=3,
                 planted=True, **_):
        flag = flag.lower()
        self.seq_len, self.enc_in, self.num_classes = seq_len, enc_in, num_classes
        n = n if n is not None else (8192 if flag == 'train' else 1024)
        g = torch.Generator().manual_seed(self.SPLIT_SEEDS.get(flag, 99))
        self.x = torch.randn(n, seq_len, enc_in, generator=g)
        self.y = torch.randint(0, num_classes, (n,), generator=g)
        if planted:
            gp = torch.Generator().manual_seed(4321)              # same motifs for every split
            L = max(8, seq_len // 5)
            t = torch.arange(L, dtype=torch.float32)
            chans = torch.randperm(enc_in, generator=gp)[:min(4, enc_in)]
            motifs = [2.0 * torch.sin(2 * math.pi * (k + 1) * 3 * t / L) * torch.hann_window(L) for k in range(num_classes)]
            offs = torch.randint(0, seq_len - L + 1, (n,), generator=g)
            for i in range(n):
                self.x[i, offs[i]:offs[i] + L, chans] += motifs[int(self.y[i])].unsqueeze(1)

    def __len__(self):
        return self.x.shape[0]

    def __getitem__(self, i):
        return self.x[i], self.y[i:i + 1].to(torch.int8)
