"""The batch-sharded data-parallel path on the HIP models: two ranks on ONE GPU (collective over gloo -- RCCL needs one GPU per
rank; the driver's 8-GPU node is the only place that runs it) train the SBM for three steps through the flat gradient bucket and
the one-launch Adam.  Replicas must stay bit-identical, and equal a single process stepping on the concatenated batch (the CE is
a mean over equal shards, the regulariser is replica-identical: SURVEY 8(e))."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, has_gpu

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _cfg():
    from argparse import Namespace
    return Namespace(enc_in=5, seq_len=120, num_class=3, epsilon=1.0, distance_func='euclidean', memory_efficient=False,
                     sbm_cls='linear', dropout=0.0, lambda_reg=0.1, lambda_div=0.1)


def _data():
    g = torch.Generator().manual_seed(7)
    return torch.randn(3, 16, 120, 5, generator=g), torch.randint(0, 3, (3, 16), generator=g)


def _train(model, bucket, opt, xs, ys, sl):
    import torch.nn.functional as F
    for i in range(xs.shape[0]):
        out, info = model(xs[i, sl].cuda())
        (F.cross_entropy(out, ys[i, sl].cuda()) + info.loss.mean()).backward()
        bucket.allreduce()
        opt.step()
        bucket.zero_grad()


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, ROOT)
    import speech_imagery_eeg_amd  # noqa: F401
    from ign_hip.ddp import FlatAdam, FlatParamBucket
    from models.Shapelet import ShapeBottleneckModel
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                      # replicas start different ...
    model = ShapeBottleneckModel(_cfg(), [5, 3], [0.2, 0.5]).cuda().train()
    bucket = FlatParamBucket(model, world)
    bucket.broadcast_state(0)                          # ... and are made rank 0's
    opt = FlatAdam(bucket, lr=5e-3)
    xs, ys = _data()
    _train(model, bucket, opt, xs, ys, slice(rank * 8, rank * 8 + 8))
    sd = [None] * world
    dist.all_gather_object(sd, {k: v.cpu().numpy() for k, v in model.state_dict().items()})
    for k in sd[0]:
        np.testing.assert_array_equal(sd[0][k], sd[1][k], err_msg=k)                # bit-identical replicas
    if rank == 0:
        torch.save({k: torch.from_numpy(v) for k, v in sd[0].items()}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_the_hip_models_equal_one_process_on_the_joint_batch(tmp_path):
    if not has_gpu():
        pytest.skip("needs a GPU")
    out = str(tmp_path / "sd.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    import speech_imagery_eeg_amd  # noqa: F401
    from ign_hip.ddp import FlatAdam, FlatParamBucket
    from models.Shapelet import ShapeBottleneckModel
    torch.manual_seed(100)
    model = ShapeBottleneckModel(_cfg(), [5, 3], [0.2, 0.5]).cuda().train()
    bucket = FlatParamBucket(model, 1)
    opt = FlatAdam(bucket, lr=5e-3)
    xs, ys = _data()
    _train(model, bucket, opt, xs, ys, slice(0, 16))
    ddp = torch.load(out, weights_only=True)
    for k, v in model.state_dict().items():
        # the gradient of the mean over 16 = the mean of the two 8-sample means, up to fp32 summation order (then Adam's
        # normalisation: the band of tests/test_gpu_models.py::test_three_adam_steps)
        a, b = ddp[k].double().numpy(), v.double().cpu().numpy()
        diff = np.abs(a - b)
        assert (diff > 5e-4 + 5e-3 * np.abs(b)).mean() <= 0.05 and diff.max() <= 3 * 2 * 5e-3 + 1e-6, k


# ---------------------------------------------------------------------------------------------------------------- RCCL, one rank
def _rccl_worker(rank, port, out):
    """A fresh process (nothing has touched the GPU before the process group exists): backend nccl (= RCCL), world size 1."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, ROOT)
    from argparse import Namespace
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    import speech_imagery_eeg_amd  # noqa: F401
    from ign_hip import ops
    from ign_hip.ddp import FlatAdam, FlatParamBucket
    from models.InterpGN import InterpGN
    cfg = Namespace(enc_in=6, seq_len=100, num_class=4, epsilon=1.0, distance_func='euclidean', memory_efficient=False,
                    sbm_cls='linear', dropout=0.0, lambda_reg=0.1, lambda_div=0.1, dnn_type='FCN')
    g = torch.Generator().manual_seed(3)
    xs = [torch.randn(8, 100, 6, generator=g).to(dev) for _ in range(2)]
    ys = [torch.randint(0, 4, (8,), generator=g).to(dev) for _ in range(2)]
    finals = []
    for collective in (True, False):
        torch.manual_seed(11)
        model = InterpGN(cfg).to(dev).train()
        bucket = FlatParamBucket(model, 1, force_collective=collective)
        opt = FlatAdam(bucket, lr=5e-3)
        for x, y in zip(xs, ys):
            _, info = model(x, None, None, None)
            ops.backward(ops.ign_loss(info.shapelet_preds, info.dnn_preds, y, 1.0, reg=info.loss)[0])
            bucket.allreduce()                        # on the compute stream, between the library's backward kernels and Adam
            opt.step()
            bucket.zero_grad()
        torch.cuda.synchronize()
        finals.append({k: v.detach().cpu().clone() for k, v in model.state_dict().items()})
    same = all(torch.equal(finals[0][k], finals[1][k]) for k in finals[0])
    torch.save({"same": same, "backend": dist.get_backend(), "n": len(finals[0])}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_one_rank_rccl_allreduce_between_backward_and_adam_is_bitwise_neutral(tmp_path):
    """RCCL smoke on one GPU (the only part of BASELINE.json config 5 a one-GPU box can exercise): torch.distributed backend
    "nccl" with world size 1, FlatParamBucket.allreduce() forced through the communicator between the HIP backward kernels and
    the flat Adam launch, two IGN(FCN) steps -- the parameters must equal the no-collective run bit for bit (sum over one rank)."""
    if not has_gpu():
        pytest.skip("needs a GPU")
    out = str(tmp_path / "rccl.pt")
    mp.spawn(_rccl_worker, args=(_free_port(), out), nprocs=1, join=True)
    res = torch.load(out, weights_only=True)
    assert res["backend"] == "nccl" and res["same"] and res["n"] > 10


def test_flat_bucket_keeps_slots_of_gradient_free_parameters_at_zero_without_a_fill_per_step():
    """`FlatParamBucket.gather` zero-fills the slot of a parameter without a gradient only when the slot was written since it was
    last zero: a parameter used in step 1 and unused in step 2 must read zeros in step 2, one that never gets a gradient costs
    no fill kernel at all."""
    if not has_gpu():
        pytest.skip("needs a GPU")
    sys.path.insert(0, os.path.join(ROOT, "speech-imagery-eeg_amd"))
    from ign_hip.ddp import FlatParamBucket

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Parameter(torch.ones(7))
            self.b = torch.nn.Parameter(torch.ones(5))
            self.never = torch.nn.Parameter(torch.ones(3))

        def forward(self, use_b):
            return (self.a * 2).sum() + ((self.b * 3).sum() if use_b else 0.0)

    m = M().cuda()
    bucket = FlatParamBucket(m, 1)
    for use_b, want_b in ((True, 3.0), (False, 0.0), (False, 0.0), (True, 3.0)):
        m(use_b).backward()
        bucket.gather()
        assert torch.equal(m.a.grad, torch.full((7,), 2.0, device="cuda"))
        assert torch.equal(m.b.grad, torch.full((5,), want_b, device="cuda")), (use_b, m.b.grad)
        assert torch.equal(m.never.grad, torch.zeros(3, device="cuda"))
        assert m.b.grad.data_ptr() == bucket.views[1].data_ptr()
        bucket.zero_grad()
    assert bucket._dirty == [True, True, False]
