"""Batch assembly and normalisation helpers of the UEA path (contract of IGN/data_factory/uea.py:7-125)."""
import numpy as np
import torch


def padding_mask(lengths, max_len=None):
    """(B,) lengths -> (B, max_len) bool, True = keep (IGN/data_factory/uea.py:45-55)."""
    lengths = torch.as_tensor(lengths)
    max_len = int(max_len or lengths.max())
    return torch.arange(max_len).unsqueeze(0) < lengths.unsqueeze(1).to(torch.int64)


def collate_fn(data, max_len=None):
    """list of (X[T_i, C], y[num_labels]) -> (X[B, max_len, C] float32 zero-padded / clipped,
    targets[B, num_labels], padding_masks[B, max_len] bool)       (IGN/data_factory/uea.py:7-42)."""
    feats, labels = zip(*data)
    lengths = [x.shape[0] for x in feats]
    if max_len is None:
        max_len = max(lengths)
    X = torch.zeros(len(feats), max_len, feats[0].shape[-1])
    for i, x in enumerate(feats):
        n = min(lengths[i], max_len)
        X[i, :n] = torch.as_tensor(x[:n], dtype=torch.float32)
    targets = torch.stack([torch.as_tensor(l) for l in labels], dim=0)
    return X, targets, padding_mask(torch.tensor(lengths), max_len)


class Normalizer:
    """Feature normalisation over ALL time steps of ALL samples ('standardization', 'minmax') or per sample
    ('per_sample_std', 'per_sample_minmax') -- IGN/data_factory/uea.py:58-107, on numpy instead of pandas.

    ``normalize(values, sample_ids)``: values (rows, C); sample_ids (rows,) maps each row to its sample.
    """

    def __init__(self, norm_type='standardization', mean=None, std=None, min_val=None, max_val=None):
        self.norm_type, self.mean, self.std, self.min_val, self.max_val = norm_type, mean, std, min_val, max_val

    def normalize(self, values, sample_ids=None):
        v = np.asarray(values, dtype=np.float64)
        eps = np.finfo(float).eps
        if self.norm_type == "standardization":
            if self.mean is None:
                self.mean, self.std = np.nanmean(v, axis=0), np.nanstd(v, axis=0, ddof=1)
            return (v - self.mean) / (self.std + eps)
        if self.norm_type == "minmax":
            if self.max_val is None:
                self.max_val, self.min_val = np.nanmax(v, axis=0), np.nanmin(v, axis=0)
            return (v - self.min_val) / (self.max_val - self.min_val + eps)
        if self.norm_type in ("per_sample_std", "per_sample_minmax"):
            if sample_ids is None:
                raise ValueError("per-sample normalisation needs sample_ids")
            out = np.empty_like(v)
            ids = np.asarray(sample_ids)
            for s in np.unique(ids):
                m = ids == s
                if self.norm_type == "per_sample_std":
                    out[m] = (v[m] - v[m].mean(axis=0)) / v[m].std(axis=0, ddof=1)
                else:
                    lo = v[m].min(axis=0)
                    out[m] = (v[m] - lo) / (v[m].max(axis=0) - lo + eps)
            return out
        raise NameError(f'Normalize method "{self.norm_type}" not implemented')


def interpolate_missing(y):
    """Linear interpolation of NaNs, edges filled with the nearest value (IGN/data_factory/uea.py:110-116)."""
    y = np.asarray(y, dtype=np.float64).copy()
    bad = np.isnan(y)
    if bad.any() and (~bad).any():
        idx = np.arange(len(y))
        y[bad] = np.interp(idx[bad], idx[~bad], y[~bad])
    return y


def subsample(y, limit=256, factor=2):
    return y[::factor] if len(y) > limit else y
