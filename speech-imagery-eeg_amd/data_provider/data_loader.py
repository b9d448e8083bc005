"""Datasets behind ``data_dict`` (contract of IGN/data_factory/data_loader.py:600-719 for UEA).

Only the classification path of the reference is in scope; forecasting / anomaly loaders
(ETT, M4, PSM ...) are not rebuilt.
"""
import glob
import os
import re

import numpy as np
import torch
from torch.utils.data import Dataset

from data_provider.ts_reader import load_ts
from data_provider.uea import Normalizer, interpolate_missing, subsample


class UEAloader(Dataset):
    """UEA multivariate archive ``<root>/<Dataset>_{TRAIN,TEST}.ts``.

    Items are ``(x[T_i, C] float64 tensor, label[1] int8 tensor)`` like the reference; ``collate_fn`` pads
    them to ``args.seq_len``.  Attributes the harness reads: ``max_seq_len``, ``feature_df`` (rows x C array),
    ``class_names`` -- plus ``seq_len / enc_in / num_classes`` so ``_get_params_from_data`` does not fall
    through to its EEG defaults.  ``flag`` is matched case-insensitively and 'val' maps to the
    TEST file, which is what upstream does.
    """

    def __init__(self, root_path, file_list=None, limit_size=None, flag=None):
        self.root_path = root_path
        series, codes = self.load_all(root_path, file_list=file_list, flag=flag)
        if limit_size is not None:
            n = int(limit_size) if limit_size > 1 else int(limit_size * len(series))
            series, codes = series[:n], codes[:n]
        self.all_IDs = np.arange(len(series))
        self.labels_df = np.asarray(codes, dtype=np.int8).reshape(-1, 1)
        lengths = np.array([s.shape[0] for s in series])
        ids = np.repeat(self.all_IDs, lengths)
        stacked = np.concatenate(series, axis=0)
        self.feature_df = Normalizer().normalize(stacked, ids)          # standardise over all rows (uea.py:85-89)
        self._offsets = np.concatenate([[0], np.cumsum(lengths)])
        self.feature_names = [f"dim_{i}" for i in range(stacked.shape[1])]
        self.enc_in = stacked.shape[1]
        self.seq_len = int(self.max_seq_len)
        self.num_classes = len(self.class_names)
        print(len(self.all_IDs))

    def load_all(self, root_path, file_list=None, flag=None):
        paths = glob.glob(os.path.join(root_path, '*')) if file_list is None else \
            [os.path.join(root_path, p) for p in file_list]
        if not paths:
            raise Exception('No files found using: {}'.format(os.path.join(root_path, '*')))
        if flag is not None:
            want = 'TRAIN' if flag.lower() == 'train' else 'TEST'
            paths = [p for p in paths if re.search(want, os.path.basename(p), re.IGNORECASE)]
        paths = sorted(p for p in paths if os.path.isfile(p) and p.endswith('.ts'))
        if not paths:
            raise Exception("No .ts files found using pattern: '*.ts'")
        return self.load_single(paths[0])

    def load_single(self, filepath):
        cases, labels, _ = load_ts(filepath)
        if labels is None:
            raise Exception(f"{filepath}: no class labels")
        self.class_names = sorted(set(labels))
        codes = [self.class_names.index(l) for l in labels]
        lens = np.array([[len(d) for d in c] for c in cases])
        if np.abs(lens - lens[:, :1]).sum() > 0:          # dimensions of one case differ in length
            cases = [[subsample(d) for d in c] for c in cases]
            lens = np.array([[len(d) for d in c] for c in cases])
        self.max_seq_len = int(lens[:, 0].max())
        series = []
        for c in cases:
            n = len(c[0])
            arr = np.stack([interpolate_missing(np.asarray(d[:n], dtype=np.float64)) if len(d) >= n
                            else interpolate_missing(np.pad(np.asarray(d, dtype=np.float64), (0, n - len(d)),
                                                            constant_values=np.nan)) for d in c], axis=1)
            series.append(arr)                             # (T_i, C)
        return series, codes

    def instance_norm(self, case):
        if self.root_path.count('EthanolConcentration') > 0:      # data_loader.py:705-713
            case = case - case.mean(0, keepdim=True)
            return case / torch.sqrt(torch.var(case, dim=1, keepdim=True, unbiased=False) + 1e-5)
        return case

    def __getitem__(self, ind):
        lo, hi = self._offsets[ind], self._offsets[ind + 1]
        return self.instance_norm(torch.from_numpy(self.feature_df[lo:hi])), torch.from_numpy(self.labels_df[ind])

    def __len__(self):
        return len(self.all_IDs)
