"""CHISCO tensor contract over pre-exported ``.npy`` shards (``--data EEG`` / ``EEG3`` without ``mne``).

The reference builds ``(n, 122, T)`` float32 arrays from ``.fif`` recordings with ``mne`` (not installed; the
OpenNeuro dataset is hundreds of GB and cannot be fetched) -- IGN/data_factory/eeg.py, eeg_processor.py.  What
the training path consumes is only the RESULT: per-sample z-scored tensors, integer labels (39 classes, or 3
via the 39->3 map of eeg_processor.py:455-467) and a 70/10/20 split (eeg.py:412-471).  This loader reads that
contract from ``<root>/{X.npy, y.npy}`` (``X``: (n, C, T) float32 in microvolts, ``y``: (n,) int) exported once
by whoever owns the raw data, applies the per-sample standardisation and the reference's split, and yields the
UEA item contract ``(x[T, C], y[1])``.

Fork defect D14 (reproduced as INTENDED, not as written): ``EEGDataset._normalize_sample`` (eeg.py:351-367) wraps
the (C, T) frame in a DataFrame with its default RangeIndex and calls ``Normalizer('per_sample_std')``
(uea.py:97-99), which groups rows BY INDEX VALUE -- every channel row is a group of one, ``x - mean`` is 0 and the
sample standard deviation of one value is NaN, so the reference as checked in turns every EEG sample into NaN
(probe: ``Normalizer('per_sample_std').normalize(pd.DataFrame(randn(4, 50)))`` is all-NaN).  What the upstream
UEA loader does with the same class -- rows are time steps, the index is the sample id -- is a per-sample,
per-channel z-score over time with the unbiased standard deviation; that is what this module computes (and what
``ign_standardise_nct_to_btc`` computes on the device).  ``tests/golden/uea_contract.npz`` pins it against the
reference's own ``Normalizer`` called the upstream way.
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


def split_sizes(n, test_size=0.2, val_size=0.1):
    """(n_train, n_val, n_test) exactly as IGN/data_factory/eeg.py:426-438: int() truncation, the remainder goes to
    train, at least one training sample, and at least one validation sample when there is more than one sample."""
    n_val, n_test = int(n * val_size), int(n * test_size)
    n_train = n - n_val - n_test
    if n_train < 1:
        n_train = 1
        n_val = min(n - 1, n_val)
        n_test = n - n_train - n_val
    elif n_val < 1 and n > 1:
        n_val = 1
        n_test = min(n - n_train - 1, n_test)
        n_train = n - n_val - n_test
    return n_train, n_val, n_test


def split_indices(n, flag, test_size=0.2, val_size=0.1, seed=42):
    """The reference's split (eeg.py:412-471): every dataset object seeds numpy's global generator with
    ``random_seed=42`` (eeg.py:149) and draws ONE ``np.random.permutation(n)`` (eeg.py:448) -- the same stream as
    ``RandomState(42)``, so train / val / test objects see the same permutation -- then takes train FIRST, then
    val, then test (eeg.py:451-453)."""
    if n == 0:
        return np.zeros(0, dtype=np.int64)
    n_train, n_val, n_test = split_sizes(n, test_size, val_size)
    perm = np.random.RandomState(seed).permutation(n)
    if flag == 'train':
        return perm[:n_train]
    if flag == 'val':
        return perm[n_train:n_train + n_val] if n_val > 0 else perm[:0]
    if flag == 'test':
        return perm[n_train + n_val:] if n_test > 0 else perm[:0]
    raise ValueError(f"invalid flag: {flag}")


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
        flag = flag.lower()
        flag = 'val' if flag == 'validation' else flag if flag in ('train', 'val', 'test') else 'train'   # eeg.py:117-126
        idx = split_indices(len(y), flag, test_size, val_size)       # permutation order kept, as the reference does
        if label_map is not None:     # folded AFTER the split; samples whose id has no entry are dropped (eeg.py:648-678)
            y = np.asarray([label_map.get(int(v), -1) for v in y], dtype=np.int64)
            idx = idx[y[idx] >= 0]
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


# The 39 -> 3 fold of the sentence classes (daily life 0 / social-emotional 1 / professional services 2): the constant
# table of IGN/data_factory/eeg_processor.py:455-461 (``create_3category_mapping``; 9 / 13 / 17 ids per class).  Only the
# text -> 39-id step before it needs the dataset's textmaps.json; y.npy already holds the 39-class ids.
_DAILY = (0, 13, 14, 18, 22, 23, 26, 35, 37)
_SOCIAL = (1, 2, 6, 7, 9, 12, 15, 17, 24, 29, 34, 36, 38)
_SERVICE = (3, 4, 5, 8, 10, 11, 16, 19, 20, 21, 25, 27, 28, 30, 31, 32, 33)
CHISCO_3CLASS_MAP = {i: c for c, ids in enumerate((_DAILY, _SOCIAL, _SERVICE)) for i in ids}
assert sorted(CHISCO_3CLASS_MAP) == list(range(39))


class EEGNpyDataset3Class(EEGNpyDataset):
    """``--data EEG3``: the 39 sentence classes folded to 3 with the reference's table (``label_map`` overrides)."""

    def __init__(self, root_path, flag='train', label_map=None, **kw):
        super().__init__(root_path, flag, label_map or CHISCO_3CLASS_MAP, **kw)
