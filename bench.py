#!/usr/bin/env python3
"""Headline benchmark: epochs/sec of IGN 3-class training on synthetic CHISCO-shaped EEG
(B=256 per GPU, C=122, T=1000; N_train=8192 samples per epoch) -- BASELINE.json's metric.

  python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over RCCL.  Either the caller starts the ranks (`python -m torch.distributed.run --nproc-per-node N
bench.py --gpus N ...`: WORLD_SIZE / RANK / LOCAL_RANK come from the environment) or, when WORLD_SIZE is not set, this
script starts them itself as a child `torch.distributed.run` BEFORE anything here touches the GPU, relays rank 0's JSON line
and exits with the child's code.

A step = one optimizer step of the training hot path (IGN/exp/experiment_classification.py:313-343) on one batch per
rank: instance norm -> shapelet bank (HIP) -> SBM head || FCN expert (implicit-GEMM conv kernels, second HIP stream) ->
gini gate -> 3-term loss -> backward (HIP shapelet / conv backward) -> gradient all-reduce over RCCL (N>1) -> Adam.
Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.  At N = 1 the default run also times
the two baselines of BASELINE.json configs 3 / 4 on the same tensors (`baselines` object) and the CPU oracle of each
workload on the host cores (`cpu_baseline`).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time
from argparse import Namespace

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.nn.functional as F

N_TRAIN = 8192          # samples per epoch
PEAK_FP32_VALU_TFLOPS = 157.3     # MI355X_MICROARCH.md, "Peak FP32 (vector)"
PEAK_FP32_MFMA_TFLOPS = 157.3     # fp32 matrix = the vector rate on gfx950
PEAK_BF16_MFMA_TFLOPS = 2500.0    # dense bf16 (never the 2:1-sparsity figure)
MFMA_32x32x16_FLOP = 2 * 32 * 32 * 16


def ch_config(model='InterpGN', dnn_type='FCN', shape='ch'):
    """shape 'ch': the CHISCO shape of BASELINE.json configs 2-5 (122 channels x 1000 samples, 3 classes);
    'bm': UEA BasicMotions, config 1 / run_uea.sh (6 channels x 100 samples, 4 classes)."""
    C, T, N = (122, 1000, 3) if shape == 'ch' else (6, 100, 4)
    return Namespace(enc_in=C, seq_len=T, num_class=N, c_out=N, epsilon=1.0, distance_func='euclidean',
                     memory_efficient=False, sbm_cls='linear', dropout=0.0, lambda_reg=0.1, lambda_div=0.1,
                     dnn_type=dnn_type, model=model, task_name='classification', pred_len=0, label_len=0,
                     output_attention=False, d_model=512, embed='timeF', freq='h', factor=1, n_heads=8, d_ff=2048,
                     activation='gelu', e_layers=2,
                     eegcnn_layers=2, eegcnn_pooling='mean', eegcnn_dropout1=0.0, eegcnn_dropout2=0.0, eegcnn_n_heads=8,
                     eegcnn_d_ff=256)


def synth_batches(n_batches, B, T, C, n_class, rank, world, device):
    """Deterministic synthetic EEG, X ~ N(0,1): batch `i` of rank `r` is seeded by its GLOBAL batch index, so the
    global data set is the same for every world size (CPU generator)."""
    xs, ys = [], []
    for i in range(n_batches):
        g = torch.Generator().manual_seed(1234 + (i * world + rank))
        xs.append(torch.randn(B, T, C, generator=g).to(device))
        ys.append(torch.randint(0, n_class, (B,), generator=g).to(device))
    return xs, ys


def planted_batches(n_batches, B, T, C, n_class, rank, world, device):
    """SURVEY 8(d)'s learnable variant: N(0,1) noise plus, per sample, a class-specific sine burst of length 200 (1, 2 or 3
    periods, amplitude 2 sigma) on 4 random channels at a random offset (seed 4321).  Used only to show the loss falls at the
    full benchmark size (`--planted`): never the headline."""
    xs, ys = [], []
    tt = torch.arange(200, dtype=torch.float32)
    for i in range(n_batches):
        g = torch.Generator().manual_seed(4321 + (i * world + rank))
        x = torch.randn(B, T, C, generator=g)
        y = torch.randint(0, n_class, (B,), generator=g)
        off = torch.randint(0, T - 200, (B,), generator=g)
        ch = torch.stack([torch.randperm(C, generator=g)[:4] for _ in range(B)])
        for b in range(B):
            burst = 2.0 * torch.sin(2 * torch.pi * (int(y[b]) + 1) * tt / 200)
            x[b, off[b]:off[b] + 200, ch[b]] += burst[:, None]
        xs.append(x.to(device))
        ys.append(y.to(device))
    return xs, ys


def shapelet_algorithmic_flops(B, C, T, groups):
    """E = B * sum_g K*C*Tw*L element-ops; forward 2E, backward 3E flops."""
    E = B * sum(K * C * (T - L + 1) * L for (K, L) in groups)
    return 2 * E, 3 * E


def host_cores():
    """CPU cores this process may actually use: min(affinity, cgroup quota) -- os.cpu_count() reports the whole
    machine inside a container and oversubscribing torch's thread pool makes the CPU baseline meaningless."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def log(msg):
    print(f"[bench +{time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


_T0 = time.perf_counter()


# --------------------------------------------------------------------------------------------------------- CPU baselines
def _time_cpu_steps(step, sample_b, what, threads, timed=3):
    """BASELINE.md section 3 / SURVEY 8(d): 1 warm-up step + `timed` timed steps (fwd + bwd + Adam), core count and CPU model
    stated.  kind = "port": the oracle is a restatement of the reference path (the reference's files never travel)."""
    step()
    t0 = time.perf_counter()
    for _ in range(timed):
        step()
    dt = (time.perf_counter() - t0) / timed
    return dict(value=(sample_b / dt) / N_TRAIN, unit="epochs/s", cores=threads, cpu=cpu_model(), kind="port",
                sample=f"{what}: 1 warm-up + {timed} timed train steps (fwd+bwd+Adam) on {sample_b} samples of shape (1000,122): "
                       f"{dt:.2f} s/step = {dt / sample_b:.3f} s/sample")


def cpu_baseline_ign(cfg, sample_b, threads):
    from oracle import ign_oracle as O
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    m = O.OracleIGN(cfg, chunk=8)
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=5e-3)
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(sample_b, cfg.seq_len, cfg.enc_in, generator=g)
    y = torch.randint(0, cfg.num_class, (sample_b,), generator=g)

    def step():
        out, info = m(x)
        loss = O.train_loss('InterpGN', out, info, y)
        loss.backward()
        opt.step()
        opt.zero_grad()
    return _time_cpu_steps(step, sample_b, "CPU oracle of IGN(FCN) (oracle/ign_oracle.py)", threads)


def cpu_baseline_baseline(config, cfg, state_dict, sample_b, threads):
    """oracle/baselines_oracle.py on the product model's own initial weights (copied to the host)."""
    from oracle import baselines_oracle as OB
    torch.set_num_threads(threads)
    p = {k: v.detach().float().cpu().clone() for k, v in state_dict.items()}
    leaves = [v.requires_grad_(True) for k, v in p.items()
              if v.dtype.is_floating_point and "running_" not in k and not k.endswith(".pe")]
    opt = torch.optim.Adam(leaves, lr=5e-3)
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(sample_b, cfg.seq_len, cfg.enc_in, generator=g)
    y = torch.randint(0, cfg.num_class, (sample_b,), generator=g)
    mask = torch.ones(sample_b, cfg.seq_len)

    def step():
        if config == "transformer":
            out = OB.transformer_logits(p, x, mask, cfg.e_layers, cfg.n_heads)
        else:
            out = OB.eegcnn_logits(p, x.permute(0, 2, 1), training=True, n_layers=2, n_heads=8)
        F.cross_entropy(out, y).backward()
        opt.step()
        opt.zero_grad()
    return _time_cpu_steps(step, sample_b, f"CPU oracle of the {config} baseline (oracle/baselines_oracle.py)", threads)


# ------------------------------------------------------------------------------------------------- harness leg (f1)
def harness_leg(B, epochs=2):
    """epochs/s of the drop-in harness itself -- Experiment.train_one_epoch of exp/experiment_classification.py
    (IGN/exp/experiment_classification.py:313-343) on `--data SYNTH` (8192 x (1000, 122) samples on the HOST), batches collated
    by the DataLoader, pinned and copied per step by the DevicePrefetcher (the H2D copy of :315-317 is inside the timed region),
    validation excluded.  One warm-up epoch, then `epochs` timed ones."""
    import speech_imagery_eeg_amd  # noqa: F401
    import run as ign_run
    from exp.experiment_classification import Experiment
    import tempfile
    argv = ["--model", "InterpGN", "--dnn_type", "FCN", "--data", "SYNTH", "--dataset", "SYNTH", "--batch_size", str(B), "--amp",
            "--train_epochs", str(epochs + 1), "--num_workers", "0", "--seed", "0"]
    a = ign_run.get_args(argv)
    a.synthetic = f"{N_TRAIN},122,1000,3"
    import contextlib
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp, contextlib.redirect_stdout(sys.stderr):      # stdout carries ONE JSON line
        os.chdir(tmp)                                        # Experiment creates ./checkpoints
        try:
            ign_run.set_seed(0)
            exp = Experiment(a)
            steps_per_epoch = len(exp.train_loader)
            _, ts = exp.train_one_epoch(0, 0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for e in range(epochs):
                _, ts = exp.train_one_epoch(e + 1, ts)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        finally:
            os.chdir(cwd)
    return {"value": epochs / dt, "unit": "epochs/s", "ms_per_step": 1e3 * dt / (epochs * steps_per_epoch),
            "steps_per_epoch": steps_per_epoch, "epochs_timed": epochs,
            "what": "Experiment.train_one_epoch (the reference's exp:313-343 loop) on --data SYNTH: host-resident samples, "
                    "DataLoader collate (num_workers 0), pinned H2D copy of every batch through DevicePrefetcher INSIDE the timed "
                    "region, forward / fused loss / backward / flat Adam; validation excluded"}


# ------------------------------------------------------------------------------------------------- self-launch for N > 1
def spawn_ranks(n, argv):
    """`python bench.py --gpus N` with no rendezvous environment: start N ranks with torch.distributed.run as a CHILD process
    (this process has not touched the GPU and never does), relay rank 0's JSON line, propagate failure."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // n)))
    log(f"starting {n} ranks: {' '.join(cmd)}")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for out in proc.stdout:
        out = out.rstrip("\n")
        if out.startswith("{") and '"metric"' in out:
            line = out
        elif out:
            print(out, file=sys.stderr, flush=True)
    rc = proc.wait()
    if rc != 0:
        raise SystemExit(f"bench.py: the {n}-rank child exited with code {rc}")
    if line is None:
        raise SystemExit("bench.py: the ranks finished without printing a result line")
    print(line, flush=True)


# ---------------------------------------------------------------------------------------------------------- measurement
class Bench:
    """One workload (ign | eegcnn | transformer) resident on this rank's GPU: model, flat bucket, Adam, data, step()."""

    def __init__(self, config, args, dev, rank, world, dist):
        import speech_imagery_eeg_amd  # noqa: F401
        from ign_hip import ops as ign_ops
        from ign_hip.ddp import FlatAdam, FlatParamBucket
        self.config, self.args, self.dev, self.rank, self.world, self.dist = config, args, dev, rank, world, dist
        self.ops = ign_ops
        cfg = ch_config(dnn_type=args.dnn, shape=args.shape)
        if config == "ign" and args.dnn == "PatchTST":
            cfg.d_model, cfg.d_ff, cfg.n_heads = 64, 128, 4
        elif config == "ign" and args.dnn == "TimesNet":
            cfg.d_model, cfg.d_ff, cfg.top_k, cfg.num_kernels = 32, 32, 3, 6
        self.cfg = cfg
        self.B, self.T, self.C = args.batch, cfg.seq_len, cfg.enc_in
        torch.manual_seed(0)                       # same initial replica on every rank
        if config == "ign":
            from models.InterpGN import InterpGN
            model = InterpGN(cfg) if args.groups == "4x5" else \
                InterpGN(cfg, num_shapelet=[10] * 6, shapelet_len=[0.05, 0.1, 0.2, 0.3, 0.5, 0.8])
        elif config == "eegcnn":
            from models.eegcnn import EEGCNNTransformer
            model = EEGCNNTransformer(cfg)
        else:
            from models.Transformer import Model as TransformerModel
            model = TransformerModel(cfg)
        self.state0 = {k: v.detach().clone() for k, v in model.state_dict().items()} if config != "ign" else None
        self.model = model.to(dev).train()
        self.bucket = FlatParamBucket(self.model, world)     # flat fp32 grad bucket: one RCCL all-reduce per step
        self.use_graph = bool(args.graph and world == 1 and config == "ign")
        self.opt = FlatAdam(self.bucket, lr=5e-3, capturable=self.use_graph)   # one ign_adam_step launch over the flat buffer
        self.amp = args.precision == "bf16"
        self.mask = torch.ones(self.B, self.T, device=dev)
        self.graphed = None
        self.ar_events = []

    def load_data(self, n_batches):
        self.n_batches = n_batches
        make = planted_batches if self.args.planted else synth_batches
        self.xs, self.ys = make(n_batches, self.B, self.T, self.C, self.cfg.num_class, self.rank, self.world, self.dev)

    def loss_of(self, x, y):
        """the loss composition of Experiment.train (exp:319-329)"""
        model, mask, cfgname = self.model, self.mask, self.config
        if self.amp:
            with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
                if cfgname == "ign":
                    out, info = model(x, mask, None, None)
                    return F.cross_entropy(out, y) + info.loss.mean() + 1.0 * F.cross_entropy(info.shapelet_preds, y)
                if cfgname == "eegcnn":
                    out, info = model(x.permute(0, 2, 1))
                    return F.cross_entropy(out, y) + info.loss.mean()
                return F.cross_entropy(model(x, mask, None, None), y)
        if cfgname == "ign":
            out, info = model(x, mask, None, None)
            # = F.cross_entropy(out, y) + info.loss.mean() + 1.0 * F.cross_entropy(info.shapelet_preds, y), one launch
            return self.ops.ign_loss(info.shapelet_preds, info.dnn_preds, y, 1.0, reg=info.loss)[0]
        if cfgname == "eegcnn":
            out, info = model(x.permute(0, 2, 1))                     # (B,C,T) view of the time-first batch, no mask
            return F.cross_entropy(out, y) + info.loss.mean()
        return F.cross_entropy(model(x, mask, None, None), y)

    def step_xy(self, x, y):
        loss = self.loss_of(x, y)
        self.ops.backward(loss)
        self.opt.step()
        self.bucket.zero_grad()
        return loss.detach()

    def step(self, i, time_allreduce=False):
        x, y = self.xs[i % self.n_batches], self.ys[i % self.n_batches]
        if self.graphed is not None:
            return self.graphed(x, y)
        loss = self.loss_of(x, y)
        self.ops.backward(loss)                  # = loss.backward(), as Experiment.train_one_epoch calls it
        if time_allreduce and (self.world > 1 or self.bucket.force_collective):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self.bucket.allreduce()
            e1.record()
            self.ar_events.append((e0, e1))
        else:
            self.bucket.allreduce()
        self.opt.step()
        self.bucket.zero_grad()
        return loss.detach()

    def fence(self):
        torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
        torch.cuda.synchronize()

    def timed(self, steps, offset, time_allreduce=False):
        """EXACTLY `steps` steps between barrier + synchronize fences; MAX over ranks."""
        self.fence()
        t0 = time.perf_counter()
        last = None
        for i in range(steps):
            last = self.step(offset + i, time_allreduce)
        self.fence()
        dt = time.perf_counter() - t0
        if self.world > 1:
            tt = torch.tensor([dt], device=self.dev, dtype=torch.float64)
            self.dist.all_reduce(tt, op=self.dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, last


def count_launches(step_fn, n=2):
    """GPU kernel launches of one step, counted by torch.profiler over `n` steps (None if the profiler is unavailable)."""
    try:
        from torch.profiler import ProfilerActivity, profile
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for i in range(n):
                step_fn(i)
            torch.cuda.synchronize()
        names = {}
        for ev in prof.events():
            if str(getattr(ev, "device_type", "")).endswith("CUDA") and "memcpy" not in ev.name.lower() and "memset" not in ev.name.lower():
                names[ev.name] = names.get(ev.name, 0) + 1
        total = sum(names.values())
        return None if total == 0 else {"per_step": total / n,
                                        "by_kernel": {k[:60]: v / n for k, v in sorted(names.items(), key=lambda kv: -kv[1])[:40]}}
    except Exception as e:                                   # profiler not usable here: report that, never fail the bench
        return {"per_step": None, "error": repr(e)[:200]}


def attention_roofline(lib_timing, steps, Bh, Lq, E, layers, attn_math):
    """Attention core: forward 4*Bh*L*S*E flop per layer (QK^T + PV), backward 2.5x (five products; S and dP are recomputed
    for dQ, which is executed work, not algorithmic).  The split-bf16 kernels execute 48 (forward) and 192
    (dQ 72 + dK 72 + dV 48) v_mfma_f32_32x32x16_bf16 per 32x32 tile: priced as EXECUTED bf16 flops against the dense bf16
    peak; the fp32-MFMA kernels are priced as algorithmic fp32 flops against the fp32 matrix peak."""
    f_fwd = 4.0 * Bh * Lq * Lq * E * layers
    ms_f = lib_timing["attn_fwd"][0] / steps
    ms_b = (lib_timing["attn_bwd_dkdv"][0] + lib_timing["attn_bwd_dq"][0]) / steps
    tf_f = f_fwd / (ms_f * 1e-3) / 1e12 if ms_f > 0 else 0.0
    tf_b = 2.5 * f_fwd / (ms_b * 1e-3) / 1e12 if ms_b > 0 else 0.0
    if attn_math in ("bf16x6", "f16x3") and E <= 64:
        tiles = Bh * ((Lq + 31) // 32) ** 2 * layers
        half = 0.5 if attn_math == "f16x3" else 1.0      # three products per element pair instead of six
        ex_f, ex_b = 48.0 * half * MFMA_32x32x16_FLOP * tiles * (E / 64.0), 192.0 * half * MFMA_32x32x16_FLOP * tiles * (E / 64.0)
        ex_tf_f = ex_f / (ms_f * 1e-3) / 1e12 if ms_f > 0 else 0.0
        ex_tf_b = ex_b / (ms_b * 1e-3) / 1e12 if ms_b > 0 else 0.0
        return {"bound": "mfma", "kernel": "attn_bwd_x6 (dq + dk + dv kernels)", "achieved": ex_tf_b, "peak": PEAK_BF16_MFMA_TFLOPS,
                "unit": "TFLOP/s", "frac": ex_tf_b / PEAK_BF16_MFMA_TFLOPS, "traffic": None, "ms_per_step": ms_b,
                "arithmetic": attn_math,
                "pricing": "executed 16-bit MFMA flops / dense bf16 (= fp16) peak; " +
                           ("f16x3: fp32 operands scaled by a power of two and split into 2 fp16 terms, 3 products"
                            if attn_math == "f16x3" else "bf16x6: fp32 operands as 3 bf16 terms, 6 products"),
                "algorithmic_fp32_tflops": tf_b,
                "fwd_kernel": {"kernel": "attn_fwd_x6", "achieved": ex_tf_f, "frac": ex_tf_f / PEAK_BF16_MFMA_TFLOPS,
                               "algorithmic_fp32_tflops": tf_f, "ms_per_step": ms_f}}
    return {"bound": "mfma", "kernel": "attn_bwd_dkdv_kernel+attn_bwd_dq_kernel", "achieved": tf_b, "peak": PEAK_FP32_MFMA_TFLOPS,
            "unit": "TFLOP/s", "frac": tf_b / PEAK_FP32_MFMA_TFLOPS, "traffic": None, "ms_per_step": ms_b,
            "pricing": "algorithmic fp32 flops / fp32 matrix peak (v_mfma_f32_32x32x2_f32 kernels)",
            "fwd_kernel": {"kernel": "attn_fwd_kernel", "achieved": tf_f, "frac": tf_f / PEAK_FP32_MFMA_TFLOPS, "ms_per_step": ms_f}}


def gemm_groups(lib, steps, flops_by_label, nprod):
    """ms/step and pricing of the GEMM-shaped kernel groups (timer label -> algorithmic flops per step).  nprod = 16-bit MFMA
    products executed per algorithmic fp32 product (6: three bf16 planes; 3: two fp16 planes; 1: bf16 autocast; None: fp32 MFMA)."""
    out = {}
    for lab, fl in flops_by_label.items():
        ms, n = lib.timing_read(lab)
        ms /= steps
        if ms <= 0:
            continue
        tf = fl / (ms * 1e-3) / 1e12
        if nprod:
            out[lab] = {"ms_per_step": ms, "launches_per_step": n // steps, "algorithmic_fp32_tflops": tf,
                        "products_per_fp32_product": nprod, "executed_16bit_tflops": nprod * tf, "peak": PEAK_BF16_MFMA_TFLOPS,
                        "frac": nprod * tf / PEAK_BF16_MFMA_TFLOPS}
        else:
            out[lab] = {"ms_per_step": ms, "launches_per_step": n // steps, "achieved": tf, "peak": PEAK_FP32_MFMA_TFLOPS,
                        "frac": tf / PEAK_FP32_MFMA_TFLOPS}
    return out


def other_kernels(lib, steps, labels):
    out = {}
    for lab in labels:
        ms, n = lib.timing_read(lab)
        if n:
            out[lab] = {"ms_per_step": ms / steps, "launches_per_step": n // steps}
    return out


def baseline_result(b, lib, steps, dt, last, cpu_sample):
    """The JSON fields of one baseline run (configs 3 / 4): throughput, roofline of its attention core (the MFMA path
    north_star names for config 4), the GEMM / convolution kernel groups, and the CPU oracle of the same model."""
    from ign_hip import ops as ign_ops
    cfg, B, T, config = b.cfg, b.B, b.T, b.config
    h3 = ign_ops.GEMM_MATH == "f16x3" and not b.amp
    x6 = (1.0 if b.amp else 3.0 if h3 else 6.0) if ign_ops.LINEAR_WGRAD == "bf16x6" else None
    amath = "f16x3" if (h3 and ign_ops.ATTN_MATH == "bf16x6") else ign_ops.ATTN_MATH
    timing = {k: lib.timing_read(k) for k in ("attn_fwd", "attn_bwd_dkdv", "attn_bwd_dq")}
    M = float(B * T)
    if config == "transformer":
        desc = "Transformer-encoder baseline (d_model 512, 8 heads, d_ff 2048, 2 layers)"
        roof = attention_roofline(timing, steps, B * cfg.n_heads, T, cfg.d_model // cfg.n_heads, cfg.e_layers, amath)
        lin = 2.0 * M * (4 * 512 * 512 + 2 * 512 * 2048) * cfg.e_layers        # Q,K,V,O + the two 1x1-conv FFN layers
        emb = 2.0 * M * 512 * 122 * 3                                            # k=3 circular token embedding
        # timer label "clconv_fwd" = every launch of ign_clconv_fwd_x6: the Linear / embedding FORWARD GEMMs and the Linear
        # INPUT-GRADIENT GEMMs (the same kernel with W in place of W^T; the embedding's input is data and takes none)
        groups = gemm_groups(lib, steps, {"clconv_fwd": lin + emb + lin, "clconv_wgrad": lin + emb}, x6)
        if "clconv_fwd" in groups:
            groups["clconv_fwd"]["what"] = "Linear + token-embedding forward GEMMs and the Linear input-gradient GEMMs (one kernel)"
        others = other_kernels(lib, steps, ("layernorm_fwd", "layernorm_bwd", "head_fwd", "head_bwd_xw", "head_bwd_x", "head_bwd_w", "absmax", "adam"))
    else:
        desc = "EEG-CNN baseline (CNN 8x8 filters k=125/25 + 2-layer encoder d_model 512, 8 heads, d_ff 256)"
        S = T // 10
        roof = attention_roofline(timing, steps, B * 8, S, cfg.d_model // 8, 2, amath)
        Me = float(B * S)
        lin = 2.0 * Me * (64 * 512 + 2 * (4 * 512 * 512 + 2 * 512 * 256))       # projection + 2 x (in_proj, out_proj, FFN)
        groups = gemm_groups(lib, steps, {"clconv_fwd": lin + lin, "clconv_wgrad": lin}, x6)      # forward + input gradient (see above)
        if "clconv_fwd" in groups:
            groups["clconv_fwd"]["what"] = "Linear forward GEMMs and the Linear input-gradient GEMMs (one kernel)"
        others = other_kernels(lib, steps, ("chan_contract", "chan_contract_bwd_w", "dwconv1d", "dwconv1d_bwd_w", "chan_stats",
                                            "affine_elu_pool", "bn_elu_pool_bwd_sums", "bn_elu_pool_bwd_apply", "autocorr",
                                            "conv1_sumsq_fwd", "conv1_sumsq_bwd", "layernorm_fwd", "layernorm_bwd", "absmax", "adam"))
        # config 3's evidence is HBM GB/s of the conv-kernel path: algorithmic bytes of the CNN block per step =
        # forward x 125 MB + u / v 4 x 65.5 MB + h 32.8 + q / g 3 x 32.8 + out 6.6; backward about twice that
        cnn_labels = ("chan_contract", "chan_contract_bwd_w", "dwconv1d", "dwconv1d_bwd_w", "chan_stats", "affine_elu_pool",
                      "bn_elu_pool_bwd_sums", "bn_elu_pool_bwd_apply", "autocorr")
        cnn_ms = sum(others[k]["ms_per_step"] for k in cnn_labels if k in others)
        Bf = float(B)
        fwd_bytes = 4.0 * Bf * (122 * T + 4 * 64 * T + 64 * (T // 2) * 4 + 64 * (T // 10))
        cnn_bytes = 3.0 * fwd_bytes
        others["cnn_block_total"] = {"ms_per_step": cnn_ms, "algorithmic_bytes_per_step": cnn_bytes,
                                     "achieved_GBps": cnn_bytes / (cnn_ms * 1e-3) / 1e9 if cnn_ms > 0 else 0.0, "peak_GBps": 8000.0,
                                     "frac_of_hbm_peak": cnn_bytes / (cnn_ms * 1e-3) / 8e12 if cnn_ms > 0 else 0.0}
    res = {"metric": f"epochs/sec (B=256, C=122, T=1000) {config} baseline 3-class",
           "value": (steps * B * b.world / N_TRAIN) / dt, "unit": "epochs/s", "ms_per_step": 1e3 * dt / steps, "steps": steps,
           "workload": f"Synthetic CHISCO-shape EEG (122ch x 1000, 3-class) {desc}, Adam lr 5e-3, fp32", "final_loss": float(last),
           "gemm_math": ("bf16 autocast (one product)" if b.amp else ign_ops.GEMM_MATH), "roofline": roof, "gemm_kernels": groups,
           "other_kernels": others}
    if b.world == 1 and cpu_sample > 0 and b.rank == 0:
        log(f"cpu baseline ({config}) ...")
        res["cpu_baseline"] = cpu_baseline_baseline(config, cfg, b.state0, 16 if config == "transformer" else 64, host_cores())
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (weak scaling)")
    ap.add_argument("--alt-steps", type=int, default=16, help="steps of the second timed region that runs the FCN convolution "
                    "GEMMs on the fp32-MFMA kernels instead of the split-bf16 ones (0 = skip)")
    ap.add_argument("--iso-steps", type=int, default=8, help="steps of the serial (one-stream) pass that measures isolated "
                    "kernel durations for the roofline object")
    ap.add_argument("--cpu-sample", type=int, default=8, help="samples per step of the IGN CPU baseline (1 warm-up + 3 timed "
                    "steps; 0 = skip every CPU baseline)")
    ap.add_argument("--baseline-steps", type=int, default=6, help="timed steps of each baseline (configs 3 / 4) appended to the "
                    "default single-GPU IGN run as the `baselines` object (0 = skip)")
    ap.add_argument("--harness-epochs", type=int, default=2,
                    help="timed epochs of the harness leg (Experiment.train_one_epoch on --data SYNTH with per-step H2D copies; "
                         "single GPU, default config only; 0 = skip)")
    ap.add_argument("--graph", action="store_true",
                    help="replay the whole step (fwd + loss + bwd + Adam) as one captured hipGraph (single GPU; pays off in "
                         "launch-bound regimes such as --batch 32, the UEA setting of run_uea.sh)")
    ap.add_argument("--groups", choices=["4x5", "6x10"], default="4x5",
                    help="shapelet bank of the IGN config: 4x5 = what the reference driver builds (4 length groups x 5 shapelets, "
                         "SURVEY D4; the headline); 6x10 = the SBM/LTS bank (6 groups x 10, lengths .05-.8 T) as a stress case")
    ap.add_argument("--precision", choices=["fp32", "bf16"], default="fp32",
                    help="fp32 = the parity path (the reference run with --amp, BASELINE.json's headline); bf16 = the "
                         "reference's DEFAULT mode, torch.autocast(bfloat16) around forward + loss: GEMM operands rounded to "
                         "bf16 on the same kernels (one product per MFMA step), shapelet path and softmax stay fp32")
    ap.add_argument("--dnn", choices=["FCN", "ResNet", "PatchTST", "TimesNet"], default="FCN",
                    help="deep expert of the gated mixture (config ign): FCN = the reference driver's default and the headline; "
                         "the others are the remaining `--dnn_type` experts (PatchTST: d_model 64, d_ff 128, 4 heads; TimesNet: "
                         "d_model 32, d_ff 32, top-3 periods, 6 inception kernels -- the reference's 512 / 2048 defaults put 3.9 M "
                         "patch tokens / 121-tap 2-D convolutions over 2048 channels on this input)")
    ap.add_argument("--config", choices=["ign", "eegcnn", "transformer"], default="ign",
                    help="ign = BASELINE.json's headline (config 1 / 5); eegcnn / transformer = the baselines of "
                         "configs 3 / 4 on the same synthetic tensors")
    ap.add_argument("--shape", choices=["ch", "bm"], default="ch",
                    help="ch = the CHISCO shape (122 ch x 1000, 3 classes: BASELINE.json's metric); bm = UEA BasicMotions (6 ch x 100, "
                         "4 classes: the shape run_uea.sh trains at --batch 32) -- a launch-bound regime, reported with the launch "
                         "count, never the headline")
    ap.add_argument("--count-launches", action="store_true",
                    help="count the GPU kernel launches of one step with torch.profiler (config.launches_per_step); off by default "
                         "so that a bench run under rocprofv3 does not start a second profiler")
    ap.add_argument("--planted", action="store_true",
                    help="SURVEY 8(d)'s learnable synthetic variant (class-specific sine bursts planted in the noise): shows the "
                         "training loss falling at the full benchmark size; reported in config.loss_curve, never the headline")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus, sys.argv[1:])           # nothing above has touched the GPU

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    # IGN_BENCH_REHEARSAL=1: every rank uses cuda:0 and the collective runs over gloo -- a functional rehearsal of the N > 1
    # path on a one-GPU box (never a measurement; the JSON line says so)
    rehearsal = os.environ.get("IGN_BENCH_REHEARSAL", "0") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # an explicit one-rank rendezvous (WORLD_SIZE=1 with RANK and MASTER_ADDR set, e.g. `torch.distributed.run --nproc-per-node 1`):
    # the collective path runs on a one-rank RCCL communicator -- the part of config 5 a one-GPU box can exercise
    solo_pg = world == 1 and all(k in os.environ for k in ("WORLD_SIZE", "RANK", "MASTER_ADDR", "MASTER_PORT"))
    if world > 1 or solo_pg:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    torch.set_num_threads(max(1, host_cores() // world))
    import speech_imagery_eeg_amd  # noqa: F401
    from ign_hip import _lib

    b = Bench(args.config, args, dev, rank, world, dist)
    b.bucket.force_collective = solo_pg
    B, T, C, cfg = b.B, b.T, b.C, b.cfg
    n_batches = min(args.steps + args.warmup, max(1, N_TRAIN // (B * world)))
    log(f"generating {n_batches} synthetic batches on the host ...")
    b.load_data(n_batches)
    log("warm-up ...")
    if b.use_graph:
        from ign_hip.graph import GraphedTrainStep
        b.graphed = GraphedTrainStep(b.step_xy, (b.xs[0], b.ys[0]))
    curve = []
    for i in range(args.warmup):
        l = b.step(i)
        if args.planted:
            curve.append(l)

    launches = count_launches(lambda i: b.step(i)) if args.count_launches else None
    log("timed region ...")
    _lib.timing_enable(True)
    if args.planted:             # the loss curve costs nothing inside the region: detached device scalars, read afterwards
        b.fence()
        t0 = time.perf_counter()
        last = None
        for i in range(args.steps):
            last = b.step(args.warmup + i, True)
            curve.append(last)
        b.fence()
        dt = time.perf_counter() - t0
    else:
        dt, last = b.timed(args.steps, args.warmup, time_allreduce=True)
    fwd_ms, fwd_n = _lib.timing_read("shp_fwd")
    bwd_ms, bwd_n = _lib.timing_read("shp_bwd")
    ar_ms = sum(e0.elapsed_time(e1) for e0, e1 in b.ar_events) / max(1, len(b.ar_events)) if b.ar_events else 0.0

    if args.config != "ign":
        res_b = baseline_result(b, _lib, args.steps, dt, last, args.cpu_sample)

    # The same K steps with the convolution GEMMs on the fp32-MFMA kernels (v_mfma_f32_32x32x2_f32), reported next to the
    # headline so that both arithmetic choices are measured in one run (results agree to ~1e-6).
    alt = None
    if args.config == "ign" and args.alt_steps > 0 and not args.planted:
        from ign_hip import fcn as _fcn_alt
        if _fcn_alt.CONV_MATH in ("bf16x6", "f16x3"):
            _was_math = _fcn_alt.CONV_MATH
            _fcn_alt.CONV_MATH = "f32"
            try:
                for i in range(2):
                    b.step(i)
                dt_alt, _ = b.timed(args.alt_steps, args.warmup)
                alt = {"conv_math": "f32 (v_mfma_f32_32x32x2_f32)", "steps": args.alt_steps,
                       "ms_per_step": 1e3 * dt_alt / args.alt_steps,
                       "value": (args.alt_steps * B * world / N_TRAIN) / dt_alt, "unit": "epochs/s"}
            finally:
                _fcn_alt.CONV_MATH = _was_math

    # Kernel durations free of co-running kernels: in the timed region the two experts of IGN run on two HIP streams, so
    # a kernel's event bracket also contains the time it shared the CUs with the other expert.  A short serial pass
    # (same step, one stream) right after the timed region gives the isolated durations the roofline fractions refer to.
    iso = {}
    if args.config == "ign":
        from models.InterpGN import InterpGN as _IGN
        was = _IGN.expert_streams
        _IGN.expert_streams = False
        b.graphed = None                                   # the isolated pass runs eagerly
        try:
            b.step(0)
            torch.cuda.synchronize()
            _lib.timing_enable(True)                 # resets the registry
            for i in range(args.iso_steps):
                b.step(args.warmup + i)
            torch.cuda.synchronize()
            for lab in ("shp_fwd", "shp_bwd", "clconv_fwd", "clconv_dgrad", "clconv_wgrad"):
                ms, n = _lib.timing_read(lab)
                iso[lab] = (ms / max(1, args.iso_steps), n // max(1, args.iso_steps))
        finally:
            _IGN.expert_streams = was

    res = None
    if rank == 0:
        amp = b.amp
        res = {
            "metric": f"epochs/sec (B={B}, C={C}, T={T}) IGN {cfg.num_class}-class" if args.config == "ign" else
                      f"epochs/sec (B={B}, C={C}, T={T}) {args.config} baseline {cfg.num_class}-class",
            "value": (args.steps * B * world / N_TRAIN) / dt,
            "unit": "epochs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if not amp else "bf16 (autocast: GEMM operands bf16, fp32 accumulate; shapelet path and softmax fp32)",
            "data": "synthetic",
        }
        if amp:
            res["metric"] += " [reference default precision: bf16 autocast]"
        if args.config == "ign" and args.dnn != "FCN":
            res["metric"] += f" [deep expert {args.dnn}]"
        common = {"per_gpu_batch": B, "global_batch": B * world, "samples_per_epoch": N_TRAIN,
                  "data": "resident in HBM (no H2D copy in the timed region; the harness path with the per-step copy is the "
                          "`harness` object)",
                  "parallelism": f"dp{world}", "final_loss": float(last), "hipgraph": bool(b.use_graph)}
        if world > 1 or solo_pg:
            common["collective"] = {"backend": "gloo (rehearsal)" if rehearsal else "nccl (RCCL over xGMI)", "ranks": world,
                                    "bucket_bytes": b.bucket.nbytes, "allreduce_ms_per_step": ar_ms,
                                    "what": "one all-reduce(sum) of the flat fp32 gradient bucket per optimizer step, then / ranks"}
        if launches is not None:
            common["launches_per_step"] = launches
        if rehearsal:
            common["rehearsal"] = "all ranks on cuda:0 over gloo: functional check of the N > 1 path, NOT a measurement"
        if args.planted:
            vals = [float(v) for v in curve]
            common["data_variant"] = "planted class-specific sine bursts (SURVEY 8(d)); NOT the headline data"
            common["loss_curve"] = {"first": vals[0], "last": vals[-1], "every_8th": vals[::8]}
        if args.config == "ign":
            groups = [(s.n, s.length) for s in b.model.sbm.shapelets]
            f_fwd, f_bwd = shapelet_algorithmic_flops(B, C, T, groups)
            # dominant kernel: the shapelet backward (3E of the 5E shapelet flops).  One step issues one launch per
            # length group; `achieved` aggregates the G launches of a step (algorithmic flops / measured device time).
            bwd_tflops = (f_bwd * args.steps) / (bwd_ms * 1e-3) / 1e12 if bwd_ms > 0 else 0.0
            fwd_tflops = (f_fwd * args.steps) / (fwd_ms * 1e-3) / 1e12 if fwd_ms > 0 else 0.0
            traffic, traffic_src = None, None
            try:    # HBM bytes per step of the dominant kernel, from the committed PMC passes (see profiles/traffic.json)
                if args.groups != "4x5" or args.shape != "ch" or B != 256:
                    raise KeyError("the PMC passes were collected for the headline workload only")
                tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
                traffic = tj["shp_bwd_kernel"]["bytes_per_step"]
                traffic_src = ("NOT measured in this run: read from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes "
                               "(profiles/traffic.json, " + str(tj.get("source", "see file")) + ")")
            except Exception:
                pass
            from ign_hip import fcn as _fcn
            res["config"] = dict(common, workload=("Synthetic CHISCO-shape EEG (122ch x 1000, 3-class)" if args.shape == "ch" else
                                                   "Synthetic BasicMotions-shape series (6ch x 100, 4-class; run_uea.sh's shape)")
                                 + f" IGN({args.dnn} expert), "
                                 + ("driver-default groups K=5 x L=" if args.groups == "4x5" else "stress bank K=10 x L=")
                                 + "{" + ",".join(str(L_) for _, L_ in groups) + "}" + ", Adam lr 5e-3, fp32",
                                 conv_math=_fcn.CONV_MATH + {
                                     "bf16x6": " (fp32 operands split exactly into 3 bf16 terms, 6 partial products accumulated in "
                                               "fp32; <= 3e-6 vs float64, same as the fp32-MFMA kernel; IGN_CONV_MATH=f32 selects that one)",
                                     "f16x3": " (fp32 operands scaled by a power of two from a device-side magnitude bound, split into 2 "
                                              "fp16 terms, 3 partial products accumulated in fp32; <= 3e-6 vs float64, same as the "
                                              "six-product bf16 and the fp32-MFMA kernels: IGN_CONV_MATH=bf16x6 / f32 select those)"
                                 }.get(_fcn.CONV_MATH, ""))
            res["roofline"] = {"bound": "valu", "kernel": "shp_bwd_kernel", "achieved": bwd_tflops,
                               "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s", "frac": bwd_tflops / PEAK_FP32_VALU_TFLOPS,
                               "traffic": traffic, "traffic_source": traffic_src,
                               "ms_per_step": bwd_ms / max(1, args.steps), "launches": bwd_n,
                               "note": "timed region = two HIP streams (SBM expert || FCN expert): these brackets include "
                                       "time shared with the other expert's kernels; 'isolated' = the same kernels in a "
                                       "serial pass of --iso-steps steps right after the timed region",
                               "fwd_kernel": {"kernel": "shp_fwd_kernel", "achieved": fwd_tflops,
                                              "frac": fwd_tflops / PEAK_FP32_VALU_TFLOPS,
                                              "ms_per_step": fwd_ms / max(1, args.steps), "launches": fwd_n}}
            if alt is not None:
                res["fp32_mfma_conv"] = alt
            if iso and args.shape == "ch":
                def _tf(flops, ms):
                    return flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
                nprod = {"bf16x6": 6.0, "f16x3": 3.0}.get(_fcn.CONV_MATH)

                def _conv_roof(tf_alg, ms):
                    # split operands: six bf16 (three fp16) MFMA products per algorithmic fp32 product, priced as EXECUTED 16-bit
                    # flops against the dense bf16 / fp16 peak (the two run at the same rate on gfx950)
                    if nprod:
                        return {"ms_per_step": ms, "achieved": tf_alg, "unit": "TFLOP/s fp32-equivalent (algorithmic)",
                                "products_per_fp32_product": nprod, "executed_16bit_tflops": nprod * tf_alg,
                                "peak": PEAK_BF16_MFMA_TFLOPS, "frac": nprod * tf_alg / PEAK_BF16_MFMA_TFLOPS}
                    if _fcn.CONV_MATH == "bf16":
                        # operands rounded to bf16, one product (the reference's autocast arithmetic): priced against the bf16 peak
                        return {"ms_per_step": ms, "achieved": tf_alg, "unit": "TFLOP/s bf16 (one product, fp32 accumulate)",
                                "peak": PEAK_BF16_MFMA_TFLOPS, "frac": tf_alg / PEAK_BF16_MFMA_TFLOPS}
                    return {"ms_per_step": ms, "achieved": tf_alg, "peak": PEAK_FP32_MFMA_TFLOPS, "frac": tf_alg / PEAK_FP32_MFMA_TFLOPS}
                fcn_f = 2.0 * B * (993 * 128 * 976 + 989 * 256 * 640 + 987 * 128 * 768)
                fcn_d = 2.0 * B * (993 * 128 * 1280 + 989 * 256 * 384)                          # data gradients of blocks 3, 2
                # measured instruction-issue ceilings of the two inner loops in isolation (profiles/r1_valu_microbench.txt):
                # forward v_sub + v_add|abs| 32.3 T elements/s, backward v_cmpx + masked v_add 24.4 T elements/s; the nominal
                # 157.3 TFLOP/s counts an FMA as 2 flop, which neither loop can issue
                E_ops = f_fwd / 2.0
                res["roofline"]["isolated"] = {
                    "shp_bwd_kernel": {"ms_per_step": iso["shp_bwd"][0], "achieved": _tf(f_bwd, iso["shp_bwd"][0]),
                                       "frac": _tf(f_bwd, iso["shp_bwd"][0]) / PEAK_FP32_VALU_TFLOPS,
                                       "frac_of_issue_ceiling": (E_ops / (iso["shp_bwd"][0] * 1e-3) / 24.4e12)
                                       if iso["shp_bwd"][0] > 0 else 0.0},
                    "shp_fwd_kernel": {"ms_per_step": iso["shp_fwd"][0], "achieved": _tf(f_fwd, iso["shp_fwd"][0]),
                                       "frac": _tf(f_fwd, iso["shp_fwd"][0]) / PEAK_FP32_VALU_TFLOPS,
                                       "frac_of_issue_ceiling": (E_ops / (iso["shp_fwd"][0] * 1e-3) / 32.3e12)
                                       if iso["shp_fwd"][0] > 0 else 0.0},
                    **{f"clconv_{nm} (mfma)": _conv_roof(_tf(fl, iso[f"clconv_{nm}"][0]), iso[f"clconv_{nm}"][0])
                       for nm, fl in (("fwd", fcn_f), ("dgrad", fcn_d), ("wgrad", fcn_f))},
                    "steps": args.iso_steps}
        else:
            res["config"] = dict(common, workload=res_b["workload"])
            res["roofline"] = res_b["roofline"]
            res["gemm_kernels"], res["other_kernels"] = res_b["gemm_kernels"], res_b["other_kernels"]
            if "cpu_baseline" in res_b:
                res["cpu_baseline"] = res_b["cpu_baseline"]

    # ---- configs 3 / 4 in the same run (single GPU, default invocation): a few timed steps each, own roofline + CPU oracle
    if world == 1 and args.config == "ign" and args.baseline_steps > 0 and not args.planted and args.dnn == "FCN" \
            and args.groups == "4x5" and not b.amp:
        del b
        torch.cuda.empty_cache()
        baselines = {}
        for name in ("eegcnn", "transformer"):
            log(f"baseline {name} ...")
            bb = Bench(name, args, dev, rank, world, dist)
            bb.load_data(min(4, N_TRAIN // B))
            for i in range(3):
                bb.step(i)
            _lib.timing_enable(True)
            dtb, lastb = bb.timed(args.baseline_steps, 3)
            baselines[name] = baseline_result(bb, _lib, args.baseline_steps, dtb, lastb, args.cpu_sample)
            del bb
            torch.cuda.empty_cache()
        res["baselines"] = baselines

    if rank == 0 and world == 1 and args.config == "ign" and args.harness_epochs > 0 and args.shape == "ch" and B == 256 \
            and args.groups == "4x5" and args.dnn == "FCN" and not args.planted and args.precision == "fp32":
        log("harness leg ...")
        torch.cuda.empty_cache()
        try:
            res["harness"] = harness_leg(B, args.harness_epochs)
        except Exception as e:                               # the headline must not die with the side leg
            res["harness"] = {"error": repr(e)[:300]}
    if rank == 0:
        if world == 1 and args.cpu_sample > 0 and args.config == "ign":
            log("cpu baseline (ign) ...")
            res["cpu_baseline"] = cpu_baseline_ign(cfg, args.cpu_sample, host_cores())
        print(json.dumps(res), flush=True)
    _lib.timing_enable(False)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
