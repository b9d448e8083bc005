"""The data contract either side of the hot path against `tests/golden/uea_contract.npz`, which
`tests/golden/make_golden.py uea_contract` produced by running the reference's OWN data_factory code (eeg.py,
eeg_processor.py, uea.py, data_loader.py::UEAloader on its vendored `.ts` parser monash.py) in the build container.
Integer work (label table, split indices, class codes, masks) is compared bit-exactly."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden

import speech_imagery_eeg_amd  # noqa: F401


def test_eeg3_label_table_equals_reference_table():
    """IGN/data_factory/eeg_processor.py:455-461 (and its duplicate eeg.py:63-69)."""
    from data_provider.eeg_npy import CHISCO_3CLASS_MAP
    g = golden("uea_contract")
    assert [CHISCO_3CLASS_MAP[i] for i in range(39)] == g["label3"].tolist()
    assert np.bincount(g["label3"]).tolist() == [9, 13, 17]


def test_split_rule_equals_reference_split():
    """IGN/data_factory/eeg.py:412-471 run unmodified after the ctor's np.random.seed(42): same indices, same order."""
    from data_provider.eeg_npy import split_indices
    g = golden("uea_contract")
    for n in g["split_ns"].tolist():
        got = {f: split_indices(n, f) for f in ("train", "val", "test")}
        for f in got:
            assert got[f].tolist() == g[f"split_{n}_{f}"].tolist(), (n, f)
        if n > 2:       # the reference's n=1 / n=2 edge rules hand the same sample to two splits; above that: a partition
            assert sorted(np.concatenate(list(got.values())).tolist()) == list(range(n))
    with pytest.raises(ValueError):
        split_indices(5, "bogus")


def test_eeg3_dataset_folds_after_the_split_and_drops_unmapped(tmp_path):
    from data_provider.eeg_npy import CHISCO_3CLASS_MAP, EEGNpyDataset, EEGNpyDataset3Class, split_indices
    rng = np.random.RandomState(0)
    X = (rng.randn(40, 4, 30) * 20 + 100).astype(np.float32)
    y = rng.randint(0, 39, 40)
    y[3] = 41                                             # an id the table does not know: dropped (eeg.py:674-678)
    np.save(tmp_path / "X.npy", X)
    np.save(tmp_path / "y.npy", y)
    parts = {f: EEGNpyDataset3Class(str(tmp_path), f) for f in ("train", "val", "test")}
    for f, ds in parts.items():
        want = [i for i in split_indices(40, f).tolist() if i != 3]
        assert ds.idx.tolist() == want
        assert [int(ds[j][1]) for j in range(len(ds))] == [CHISCO_3CLASS_MAP[int(y[i])] for i in want]
    assert sum(len(p) for p in parts.values()) == 39 and parts["train"].num_classes == 3
    assert len(EEGNpyDataset(str(tmp_path), "validation")) == 4       # 'validation' -> 'val' (eeg.py:119-121)


def test_normalizer_variants_equal_reference_normalizer():
    """IGN/data_factory/uea.py:58-107 on a frame indexed by sample id (the upstream call form)."""
    from data_provider.uea import Normalizer
    g = golden("uea_contract")
    for nt in ("standardization", "minmax", "per_sample_std", "per_sample_minmax"):
        got = Normalizer(nt).normalize(g["norm_in"], g["norm_ids"])
        np.testing.assert_allclose(got, g["norm_" + nt], rtol=1e-12, atol=1e-12, err_msg=nt)


def test_eeg_standardisation_is_the_intended_per_channel_zscore_and_the_defect_is_real():
    """Fork defect D14: the EEG call site as written (eeg.py:351-367) yields NaN for every element; the repo computes
    what the same Normalizer gives when called the upstream way (time steps as rows, one sample id)."""
    from data_provider.eeg_npy import per_sample_standardise
    g = golden("uea_contract")
    assert float(g["eeg_norm_as_written_nan_fraction"]) == 1.0
    got = per_sample_standardise(g["eeg_norm_in"][None].astype(np.float64))[0]
    np.testing.assert_allclose(got, g["eeg_norm_intended"], rtol=1e-6, atol=1e-6)     # repo adds eps=1e-8 to std


def test_padding_mask_and_collate_equal_reference():
    """IGN/data_factory/uea.py:7-55: zero-pad / clip to max_len, targets stacked, boolean keep-mask."""
    from data_provider.uea import collate_fn, padding_mask
    g = golden("uea_contract")
    assert torch.equal(padding_mask(torch.from_numpy(g["pm_lengths"]), 8), torch.from_numpy(g["pm_mask"]))
    items = [(torch.from_numpy(g[f"col_x{i}"]), torch.tensor([i % 3], dtype=torch.int8)) for i in range(4)]
    for max_len in (8, 13):
        X, t, pm = collate_fn(items, max_len=max_len)
        assert X.dtype == torch.float32 and pm.dtype == torch.bool
        assert np.array_equal(X.numpy(), g[f"col{max_len}_X"])
        assert np.array_equal(t.numpy(), g[f"col{max_len}_t"]) and t.dtype == torch.int8
        assert np.array_equal(pm.numpy(), g[f"col{max_len}_mask"])


@pytest.mark.parametrize("stem", ["EqLen_TRAIN", "EqLen_TEST", "Ragged_TRAIN", "Missing_TRAIN"])
def test_ts_files_through_uealoader_equal_reference(stem):
    """The committed generated `.ts` text through the repo's sktime-free reader + UEAloader against the reference's
    UEAloader (data_loader.py:600-719) on its vendored parser: standardised feature table, row -> sample index, class codes,
    class names, max_seq_len, and the collated batch at max_len 18 (pads the 16-step file, clips the 20-step ones)."""
    from data_provider.data_loader import UEAloader
    from data_provider.uea import collate_fn
    g = golden("uea_contract")
    ds = UEAloader(os.path.join(GOLDEN, "ts"), file_list=[stem + ".ts"])
    np.testing.assert_allclose(ds.feature_df, g[f"ts_{stem}_feature"], rtol=1e-12, atol=1e-12)
    lengths = np.diff(ds._offsets)
    assert np.repeat(ds.all_IDs, lengths).tolist() == g[f"ts_{stem}_index"].tolist()
    assert ds.labels_df.tolist() == g[f"ts_{stem}_labels"].tolist() and ds.labels_df.dtype == np.int8
    assert [float(c) for c in ds.class_names] == [float(c) for c in g[f"ts_{stem}_classes"]]
    assert int(ds.max_seq_len) == int(g[f"ts_{stem}_maxlen"])
    X, t, pm = collate_fn([ds[i] for i in range(len(ds))], max_len=18)
    np.testing.assert_allclose(X.numpy(), g[f"ts_{stem}_X18"], rtol=1e-6, atol=1e-7)
    assert np.array_equal(t.numpy(), g[f"ts_{stem}_t18"]) and np.array_equal(pm.numpy(), g[f"ts_{stem}_mask18"])
