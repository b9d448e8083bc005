"""Multi-process CPU tests (gloo, world_size 2) of the batch-sharded data-parallel path: the flat gradient
bucket, the rank-sharded sampler, and the Experiment harness under torch.distributed (oracle models on CPU)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    import speech_imagery_eeg_amd  # noqa: F401
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _bucket_worker(rank, world, port, out):
    _init(rank, world, port)
    from ign_hip.ddp import FlatParamBucket, shard_indices
    from oracle import ign_oracle as O
    from argparse import Namespace
    cfg = Namespace(enc_in=3, seq_len=40, num_class=3, epsilon=1.0, distance_func='euclidean', memory_efficient=False,
                    sbm_cls='linear', dropout=0.0, lambda_reg=0.1, lambda_div=0.1)
    torch.manual_seed(100 + rank)                    # replicas start DIFFERENT on purpose
    model = O.OracleSBM(cfg, [2, 2], [0.2, 0.5])
    bucket = FlatParamBucket(model, world)
    bucket.broadcast_state(0)                        # ... and are made identical to rank 0
    g = torch.Generator().manual_seed(7)
    X = torch.randn(8, 40, 3, generator=g)
    y = torch.randint(0, 3, (8,), generator=g)
    idx = shard_indices(8, rank, world, epoch=0, seed=0, shuffle=False)
    opt = torch.optim.Adam(model.parameters(), lr=5e-3)
    for _ in range(2):
        out_, info = model(X[idx])
        loss = torch.nn.functional.cross_entropy(out_, y[idx]) + info.loss.mean()
        loss.backward()
        assert all(p.grad.data_ptr() >= bucket.flat_grad.data_ptr() for p in model.parameters())   # still views
        bucket.allreduce()
        opt.step()
        bucket.zero_grad()
    if rank == 0:
        torch.save({k: v.clone() for k, v in model.state_dict().items()}, out)
    sd = [None] * world
    dist.all_gather_object(sd, {k: v.numpy() for k, v in model.state_dict().items()})
    for k in sd[0]:
        np.testing.assert_array_equal(sd[0][k], sd[1][k])        # replicas stay bit-identical
    dist.barrier()
    dist.destroy_process_group()


def test_flat_bucket_matches_single_process(tmp_path):
    """2 ranks x 4 samples, averaged grads == 1 process x 8 samples (CE is a mean over equal shards; the model
    regulariser is replica-identical) -- SURVEY 8(e)."""
    out = str(tmp_path / "sd.pt")
    mp.spawn(_bucket_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    sys.path.insert(0, ROOT)
    from oracle import ign_oracle as O
    from argparse import Namespace
    cfg = Namespace(enc_in=3, seq_len=40, num_class=3, epsilon=1.0, distance_func='euclidean', memory_efficient=False,
                    sbm_cls='linear', dropout=0.0, lambda_reg=0.1, lambda_div=0.1)
    torch.manual_seed(100)
    model = O.OracleSBM(cfg, [2, 2], [0.2, 0.5])
    g = torch.Generator().manual_seed(7)
    X = torch.randn(8, 40, 3, generator=g)
    y = torch.randint(0, 3, (8,), generator=g)
    opt = torch.optim.Adam(model.parameters(), lr=5e-3)
    for _ in range(2):
        o, info = model(X)
        (torch.nn.functional.cross_entropy(o, y) + info.loss.mean()).backward()
        opt.step()
        opt.zero_grad()
    ddp = torch.load(out, weights_only=True)
    for k, v in model.state_dict().items():
        np.testing.assert_allclose(ddp[k].numpy(), v.numpy(), rtol=2e-4, atol=2e-5, err_msg=k)


def test_shard_indices_partition():
    sys.path.insert(0, ROOT)
    import speech_imagery_eeg_amd  # noqa: F401
    from ign_hip.ddp import shard_indices
    for world in (1, 2, 4, 8):
        parts = [shard_indices(2048, r, world, epoch=3, seed=5) for r in range(world)]
        allidx = torch.cat(parts)
        assert len(allidx) == 2048 and len(set(allidx.tolist())) == 2048
        assert all(len(p) == 2048 // world for p in parts)
    a = shard_indices(100, 0, 3, epoch=1, seed=0)
    assert len(a) == 33                                      # 100 // 3: the remainder is dropped (equal shards)
    assert not torch.equal(shard_indices(64, 0, 2, epoch=1), shard_indices(64, 0, 2, epoch=2))


def _harness_worker(rank, world, port, tmp):
    _init(rank, world, port)
    os.chdir(tmp)
    import run
    from exp.experiment_classification import Experiment
    from oracle import ign_oracle as O
    Experiment.model_dict['SBM'] = lambda configs, num_shapelet, shapelet_len: O.OracleSBM(configs, num_shapelet, shapelet_len)
    a = run.get_args(["--model", "SBM", "--data", "SYNTH", "--synthetic", "32,3,40,3", "--train_epochs", "2",
                      "--batch_size", "8", "--seed", "0", "--amp", "--log_interval", "1", "--num_shapelet", "2"])
    run.set_seed(0)
    e = Experiment(a)
    assert e.distributed and e.world == 2 and len(e.train_loader) == 2       # 32 samples / 2 ranks / batch 8
    e.train()
    sd = [None] * world
    dist.all_gather_object(sd, {k: v.cpu().numpy() for k, v in e.model.state_dict().items()})
    for k in sd[0]:
        np.testing.assert_array_equal(sd[0][k], sd[1][k])
    loss, res, _ = e.test(result_dir=os.path.join(tmp, "result"))
    assert np.isfinite(loss) and res.preds.shape[0] == len(e.test_data)
    if rank == 0:
        assert os.path.exists(os.path.join(e.checkpoint_dir, "checkpoint.pth"))
    dist.barrier()
    dist.destroy_process_group()


def test_experiment_under_gloo(tmp_path):
    mp.spawn(_harness_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)


def _early_stop_worker(rank, world, port, tmp):
    """A BatchNorm model (IGN with the FCN expert) whose per-rank running statistics differ, trained until early stopping
    fires: every rank must take the same stopping decision in the same epoch and end with the same weights AND buffers."""
    _init(rank, world, port)
    os.chdir(tmp)
    import run
    from exp.experiment_classification import Experiment
    from oracle import ign_oracle as O
    Experiment.model_dict['InterpGN'] = lambda cfg: O.OracleIGN(cfg)
    a = run.get_args(["--model", "InterpGN", "--data", "SYNTH", "--synthetic", "64,3,40,3", "--train_epochs", "12",
                      "--batch_size", "8", "--seed", "0", "--amp", "--log_interval", "1", "--patience", "1", "--min_epochs", "0",
                      "--lr", "0.02"])
    run.set_seed(0)
    e = Experiment(a)
    seen = []
    orig = e.validation

    def rec():
        # per-rank running statistics really differ before the sync would be needed: record what THIS rank computes
        r = orig()
        seen.append(r)
        return r
    e.validation = rec
    e.train()
    info = [None] * world
    dist.all_gather_object(info, dict(stop=e.epoch_stop, n_val=len(seen), vals=seen,
                                      sd={k: v.cpu().numpy() for k, v in e.model.state_dict().items()}))
    assert info[0]["stop"] == info[1]["stop"] and info[0]["n_val"] == info[1]["n_val"]
    assert info[0]["stop"] < 11, "early stopping never fired: the test would not cover the divergence"
    assert info[0]["vals"] == info[1]["vals"]                   # same model evaluated on every rank (buffers synced)
    for k in info[0]["sd"]:
        np.testing.assert_array_equal(info[0]["sd"][k], info[1]["sd"][k], err_msg=k)
    dist.barrier()
    dist.destroy_process_group()


def test_early_stopping_is_one_decision_for_all_ranks(tmp_path):
    mp.spawn(_early_stop_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
