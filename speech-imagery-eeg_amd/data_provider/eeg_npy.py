"""CHISCO tensor contract over pre-exported ``.npy`` shards (``--data EEG`` / ``EEG3`` without ``mne``).

The reference builds ``(n, 122, T)`` float32 arrays from ``.fif`` recordings with ``mne`` (not installed; the
OpenNeuro dataset is hundreds of GB and cannot be fetched) -- IGN/data_factory/eeg.py, eeg_processor.py.  What
the training path consumes is only the RESULT: per-sample z-scored tensors, integer labels (39 classes, or 3
via the 39->3 map of eeg_processor.py:455-467) and a 70/10/20 split (eeg.py:412-471).  This loader reads that
contract from ``<root>/{X.npy, y.npy}`` (``X``: (n, C, T) float32 in microvolts, ``y``: (n,) int) exported once
by whoever owns the raw data, applies the same per-sample standardisation and split, and yields the UEA item
contract ``(x[T, C], y[1])``.
"""
import os

import numpy as np
import torch
from torch.utils.data import Dataset


def per_sample_standardise(x_nct, eps=1e-8):
    """Normalizer('per_sample_std') of IGN/data_factory/eeg.py:332-367: per sample and channel over time."""
    m = x_nct.mean(axis=-1, keepdims=True)
    s = x_nct.std(axis=-1, ddof=1, keepdims=True)
    return (x_nct - m) / (s + eps)


def split_indices(n, flag, test_size=0.2, val_size=0.1, seed=42):
    """70/10/20 shuffle split with a fixed seed (eeg.py:412-471)."""
    perm = np.random.RandomState(seed).permutation(n)
    n_test, n_val = int(round(n * test_size)), int(round(n * val_size))
    if flag == 'test':
        return perm[:n_test]
    if flag == 'val':
        return perm[n_test:n_test + n_val]
    return perm[n_test + n_val:]


class EEGNpyDataset(Dataset):
    num_label_classes = 39

    def __init__(self, root_path, flag='train', label_map=None, test_size=0.2, val_size=0.1, mmap=True, raw=False, **_):
        self.raw = bool(raw)      # True: items are RAW (C,T) recordings; the batch is standardised on the GPU instead
        xp, yp = os.path.join(root_path, "X.npy"), os.path.join(root_path, "y.npy")
        if not (os.path.exists(xp) and os.path.exists(yp)):
            raise FileNotFoundError(f"{root_path}: expected X.npy (n,C,T) and y.npy (n,) exported from the CHISCO "
                                    f"recordings (the .fif/mne pipeline of the reference is out of scope)")
        X = np.load(xp, mmap_mode='r' if mmap else None, allow_pickle=False)
        y = np.load(yp, allow_pickle=False).astype(np.int64)
        if label_map is not None:
            y = np.asarray([label_map[int(v)] for v in y], dtype=np.int64)
        idx = np.sort(split_indices(len(y), flag.lower(), test_size, val_size))
        self.X, self.y, self.idx = X, y, idx
        self.enc_in, self.seq_len = int(X.shape[1]), int(X.shape[2])
        self.num_classes = int(y.max()) + 1 if label_map is None else len(set(label_map.values()))

    def __len__(self):
        return len(self.idx)

    def __getitem__(self, i):
        j = self.idx[i]
        x = np.asarray(self.X[j], dtype=np.float32)
        y = torch.tensor([self.y[j]], dtype=torch.int64)
        if self.raw:              # device pipeline: data_provider.device_prefetch.standardise_raw_batch does the rest
            return torch.from_numpy(np.ascontiguousarray(x)), y
        return torch.from_numpy(per_sample_standardise(x).T.copy()), y


def collate_raw(data):
    """list of (X[C,T] raw, y[1]) -> (X[B,C,T], y[B,1], None): the batch contract before the device transform."""
    feats, labels = zip(*data)
    return torch.stack(feats, dim=0), torch.stack(labels, dim=0), None


class EEGNpyDataset3Class(EEGNpyDataset):
    """39 sentence classes folded into 3 groups of 13 consecutive ids (stand-in for eeg_processor.py:455-467's
    semantic map, which needs the textmaps.json of the raw dataset; pass ``label_map`` to override)."""

    def __init__(self, root_path, flag='train', label_map=None, **kw):
        super().__init__(root_path, flag, label_map or {i: i // 13 for i in range(39)}, **kw)
