"""GPU parity of the on-device input pipeline: ign_standardise_nct_to_btc against the CPU
Normalizer('per_sample_std') restatement, and the DevicePrefetcher against the CPU loader path, batch for batch."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


@pytest.mark.parametrize("B,C,T", [(7, 122, 1000), (3, 5, 50), (2, 33, 130), (1, 1, 2), (256, 122, 1000)])
def test_standardise_matches_cpu_normalizer(B, C, T):
    dev = _dev()
    import speech_imagery_eeg_amd  # noqa: F401
    from data_provider.eeg_npy import per_sample_standardise
    from ign_hip import ops
    rng = np.random.RandomState(B + C + T)
    x = (rng.randn(B, C, T) * rng.uniform(5, 80, size=(B, C, 1)) + rng.uniform(-2e4, 2e4, size=(B, C, 1))).astype(np.float32)
    ref = per_sample_standardise(x.astype(np.float64), eps=1e-8).transpose(0, 2, 1)
    out = ops.standardise_nct_to_btc(torch.from_numpy(x).to(dev))
    assert out.shape == (B, T, C)
    # offsets of 2e4 uV at fp32: the subtraction x - mean itself carries ~1e-3 absolute = ~1e-4 of a 10 uV std
    assert float((out.double().cpu() - torch.from_numpy(ref)).abs().max()) < 5e-4
    # size-independent properties: zero mean, unit (unbiased) std per (b, c)
    assert float(out.mean(dim=1).abs().max()) < 1e-3
    if T > 2:
        assert float((out.std(dim=1, unbiased=True) - 1).abs().max()) < 1e-3


def test_standardise_equals_reference_normalizer_called_the_upstream_way():
    """The device kernel against the reference's own Normalizer('per_sample_std') (IGN/data_factory/uea.py:97-99) applied
    as upstream intends -- time steps as rows, one sample id (tests/golden/uea_contract.npz; the EEG call site as written
    returns NaN: fork defect D14)."""
    dev = _dev()
    import speech_imagery_eeg_amd  # noqa: F401
    from conftest import golden, parity
    from ign_hip import ops
    g = golden("uea_contract")
    x = torch.from_numpy(g["eeg_norm_in"]).to(dev)[None]                       # (1, C, T)
    out = ops.standardise_nct_to_btc(x)[0].T                                  # -> (C, T)
    parity("standardised frame", out, g["eeg_norm_intended"], kind="elem", ref_is="reference Normalizer (pandas, float32 in)")


def test_prefetcher_equals_cpu_loader_path(tmp_path):
    dev = _dev()
    import speech_imagery_eeg_amd  # noqa: F401
    from data_provider.device_prefetch import DevicePrefetcher, standardise_raw_batch
    from data_provider.eeg_npy import EEGNpyDataset3Class, collate_raw
    from data_provider.uea import collate_fn
    rng = np.random.RandomState(1)
    X = (rng.randn(50, 12, 200) * 30 + 500).astype(np.float32)
    np.save(tmp_path / "X.npy", X)
    np.save(tmp_path / "y.npy", rng.randint(0, 39, size=50))
    cpu = EEGNpyDataset3Class(str(tmp_path), flag="train")
    raw = EEGNpyDataset3Class(str(tmp_path), flag="train", raw=True)
    want = list(torch.utils.data.DataLoader(cpu, batch_size=16, shuffle=False, collate_fn=lambda b: collate_fn(b, max_len=200)))
    loader = torch.utils.data.DataLoader(raw, batch_size=16, shuffle=False, collate_fn=collate_raw, pin_memory=True)
    got = list(DevicePrefetcher(loader, dev, transform=standardise_raw_batch))
    assert len(got) == len(want)
    for (a, b, m), (c, d, n) in zip(got, want):
        assert a.is_cuda and a.shape == c.shape
        assert float((a.cpu() - c).abs().max()) < 1e-4
        assert torch.equal(b.cpu(), d) and torch.equal(m.cpu(), n)


def test_harness_runs_on_the_device_pipeline(tmp_path, monkeypatch):
    """Experiment on CHISCO-contract shards: the device pipeline (raw items, GPU standardise) and the CPU item path train
    to the same losses."""
    dev = _dev()
    import speech_imagery_eeg_amd  # noqa: F401
    import run as driver
    from exp.experiment_classification import Experiment
    rng = np.random.RandomState(2)
    y = rng.randint(0, 39, size=96)
    X = (rng.randn(96, 6, 100) * 20 + 300).astype(np.float32)
    X[:, 0, 10:30] += (y // 13)[:, None] * 40.0                 # a class-dependent burst so that the loss moves
    np.save(tmp_path / "X.npy", X)
    np.save(tmp_path / "y.npy", y)
    monkeypatch.chdir(tmp_path)
    losses = {}
    for flag in (True, False):
        args = driver.get_args(["--model", "InterpGN", "--dnn_type", "FCN", "--data", "EEG3", "--data_root", str(tmp_path),
                                "--dataset", "chisco_npy", "--train_epochs", "2", "--batch_size", "32", "--seed", "0", "--amp",
                                "--min_epochs", "5", "--num_workers", "0"])
        args.device_standardise = flag
        driver.set_seed(0)
        exp = Experiment(args)
        exp.train()
        losses[flag] = exp.validation()
    # the two input paths agree to ~1e-5 per element; six Adam steps at lr 5e-3 amplify that to the 1e-3 level in the loss
    assert abs(losses[True][0] - losses[False][0]) < 2e-2 and abs(losses[True][1] - losses[False][1]) <= 0.15


@pytest.mark.parametrize("B,T,C", [(3, 100, 6), (2, 1000, 122), (5, 65, 33), (1, 1, 1)])
def test_contiguous_bct_transposes_the_time_first_batch(B, T, C):
    """ops.contiguous_bct (ign_transpose_btc_to_bct): the permuted view of a time-first batch becomes a contiguous (B, C, T)
    tensor, bit for bit; other layouts take torch's route."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import speech_imagery_eeg_amd  # noqa
    from ign_hip import ops
    x = torch.randn(B, T, C, device="cuda")
    out = ops.contiguous_bct(x.permute(0, 2, 1))
    assert out.is_contiguous() and torch.equal(out, x.permute(0, 2, 1).contiguous())
    y = torch.randn(B, C, T, device="cuda")
    assert ops.contiguous_bct(y) is y
    z = torch.randn(B, C, 2 * T, device="cuda")[:, :, ::2]
    assert torch.equal(ops.contiguous_bct(z), z.contiguous())
