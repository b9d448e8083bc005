#!/usr/bin/env python3
"""Headline benchmark: epochs/sec of IGN 3-class training on synthetic CHISCO-shaped EEG
(B=256 per GPU, C=122, T=1000; N_train=8192 samples per epoch) -- BASELINE.json's metric.

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

A step = one optimizer step of the training hot path (IGN/exp/experiment_classification.py:313-343) on one
batch per rank: instance norm -> shapelet bank (HIP) -> SBM head || FCN expert (implicit-GEMM conv kernels, second HIP
stream) -> gini gate -> 3-term loss -> backward (HIP shapelet / conv backward) -> gradient all-reduce over RCCL (N>1) -> Adam.  Inputs are resident in HBM
before the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
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


def ch_config(model='InterpGN', dnn_type='FCN'):
    return Namespace(enc_in=122, seq_len=1000, num_class=3, c_out=3, epsilon=1.0, distance_func='euclidean',
                     memory_efficient=False, sbm_cls='linear', dropout=0.0, lambda_reg=0.1, lambda_div=0.1,
                     dnn_type=dnn_type, model=model, task_name='classification', pred_len=0, label_len=0,
                     output_attention=False, d_model=512, embed='timeF', freq='h', factor=1, n_heads=8, d_ff=2048,
                     activation='gelu', e_layers=2)


def synth_batches(n_batches, B, T, C, n_class, rank, world, device):
    """Deterministic synthetic EEG, X ~ N(0,1): batch `i` of rank `r` is seeded by its GLOBAL batch index, so the
    global data set is the same for every world size (CPU generator)."""
    xs, ys = [], []
    for i in range(n_batches):
        g = torch.Generator().manual_seed(1234 + (i * world + rank))
        xs.append(torch.randn(B, T, C, generator=g).to(device))
        ys.append(torch.randint(0, n_class, (B,), generator=g).to(device))
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


def log(msg):
    print(f"[bench +{time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def cpu_baseline(cfg, sample_b, threads):
    """The CPU oracle (a restatement of the reference path, kind="port") timed on the host: one full training step
    (fwd + bwd + Adam) on `sample_b` samples of the same synthetic distribution."""
    from oracle import ign_oracle as O
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    m = O.OracleIGN(cfg, chunk=8)
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=5e-3)
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(sample_b, cfg.seq_len, cfg.enc_in, generator=g)
    y = torch.randint(0, cfg.num_class, (sample_b,), generator=g)
    t0 = time.perf_counter()
    out, info = m(x)
    loss = O.train_loss('InterpGN', out, info, y)
    loss.backward()
    opt.step()
    dt = time.perf_counter() - t0
    return dict(value=(sample_b / dt) / N_TRAIN, unit="epochs/s", cores=threads, kind="port",
                sample=f"1 train step (fwd+bwd+Adam) of the CPU oracle on {sample_b} samples of shape (1000,122): "
                       f"{dt:.1f} s = {dt / sample_b:.2f} s/sample")


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
    ap.add_argument("--cpu-sample", type=int, default=12, help="samples for the CPU baseline step (0 = skip)")
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
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run "
                         f"--nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    # IGN_BENCH_REHEARSAL=1: every rank uses cuda:0 and the collective runs over gloo -- a functional rehearsal of the N > 1
    # path on a one-GPU box (never a measurement; the JSON line says so)
    rehearsal = os.environ.get("IGN_BENCH_REHEARSAL", "0") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    torch.set_num_threads(host_cores())
    import speech_imagery_eeg_amd  # noqa: F401
    from ign_hip import _lib
    from ign_hip import ops as ign_ops
    from ign_hip.ddp import FlatAdam, FlatParamBucket
    from models.InterpGN import InterpGN

    cfg = ch_config(dnn_type=args.dnn)
    if args.dnn == "PatchTST":
        cfg.d_model, cfg.d_ff, cfg.n_heads = 64, 128, 4
    elif args.dnn == "TimesNet":
        cfg.d_model, cfg.d_ff, cfg.top_k, cfg.num_kernels = 32, 32, 3, 6
    B, T, C = args.batch, cfg.seq_len, cfg.enc_in
    torch.manual_seed(0)                       # same initial replica on every rank
    if args.config == "ign":
        model = InterpGN(cfg) if args.groups == "4x5" else \
            InterpGN(cfg, num_shapelet=[10] * 6, shapelet_len=[0.05, 0.1, 0.2, 0.3, 0.5, 0.8])
    elif args.config == "eegcnn":
        from models.eegcnn import EEGCNNTransformer
        cfg.eegcnn_layers, cfg.eegcnn_pooling, cfg.eegcnn_dropout1, cfg.eegcnn_dropout2 = 2, 'mean', 0.0, 0.0
        cfg.eegcnn_n_heads, cfg.eegcnn_d_ff = 8, 256
        model = EEGCNNTransformer(cfg)
    else:
        from models.Transformer import Model as TransformerModel
        model = TransformerModel(cfg)
    model = model.to(dev).train()
    bucket = FlatParamBucket(model, world)     # flat fp32 grad bucket: one RCCL all-reduce per step
    use_graph = args.graph and world == 1 and args.config == "ign"
    opt = FlatAdam(bucket, lr=5e-3, capturable=use_graph)   # one ign_adam_step launch over the flat parameter buffer

    n_batches = min(args.steps + args.warmup, max(1, N_TRAIN // (B * world)))
    log(f"generating {n_batches} synthetic batches on the host ...")
    xs, ys = synth_batches(n_batches, B, T, C, cfg.num_class, rank, world, dev)
    log("warm-up ...")
    mask = torch.ones(B, T, device=dev)

    def step_xy(x, y):
        out, info = model(x, mask, None, None)
        loss = ign_ops.ign_loss(info.shapelet_preds, info.dnn_preds, y, 1.0)[0] + info.loss.mean()
        loss.backward()
        opt.step()
        bucket.zero_grad()
        return loss.detach()

    graphed = None

    amp = args.precision == "bf16"

    def step(i):
        x, y = xs[i % n_batches], ys[i % n_batches]
        if graphed is not None:
            return graphed(x, y)
        if amp:
            # the composition of Experiment.train inside the autocast region (exp:319-329)
            with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
                if args.config == "ign":
                    out, info = model(x, mask, None, None)
                    loss = F.cross_entropy(out, y) + info.loss.mean() + 1.0 * F.cross_entropy(info.shapelet_preds, y)
                elif args.config == "eegcnn":
                    out, info = model(x.permute(0, 2, 1).contiguous())
                    loss = F.cross_entropy(out, y) + info.loss.mean()
                else:
                    loss = F.cross_entropy(model(x, mask, None, None), y)
        elif args.config == "ign":
            out, info = model(x, mask, None, None)
            # = F.cross_entropy(out, y) + info.loss.mean() + 1.0 * F.cross_entropy(info.shapelet_preds, y), as Experiment.train
            loss = ign_ops.ign_loss(info.shapelet_preds, info.dnn_preds, y, 1.0)[0] + info.loss.mean()
        elif args.config == "eegcnn":
            out, info = model(x.permute(0, 2, 1).contiguous())        # (B,C,T), no mask
            loss = F.cross_entropy(out, y) + info.loss.mean()
        else:
            loss = F.cross_entropy(model(x, mask, None, None), y)
        loss.backward()
        bucket.allreduce()
        opt.step()
        bucket.zero_grad()
        return loss.detach()

    if use_graph:
        from ign_hip.graph import GraphedTrainStep
        graphed = GraphedTrainStep(step_xy, (xs[0], ys[0]))
    for i in range(args.warmup):
        step(i)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    log("timed region ...")
    _lib.timing_enable(True)
    fence()
    t0 = time.perf_counter()
    last = None
    for i in range(args.steps):
        last = step(args.warmup + i)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    fwd_ms, fwd_n = _lib.timing_read("shp_fwd")
    bwd_ms, bwd_n = _lib.timing_read("shp_bwd")

    # The same K steps with the convolution GEMMs on the fp32-MFMA kernels (v_mfma_f32_32x32x2_f32), reported next to the
    # headline so that both arithmetic choices are measured in one run (results agree to ~1e-6).
    alt = None
    if args.config == "ign" and args.alt_steps > 0:
        from ign_hip import fcn as _fcn_alt
        if _fcn_alt.CONV_MATH == "bf16x6":
            _fcn_alt.CONV_MATH = "f32"
            try:
                for i in range(2):
                    step(i)
                fence()
                t1 = time.perf_counter()
                for i in range(args.alt_steps):
                    step(args.warmup + i)
                fence()
                dt_alt = time.perf_counter() - t1
                if world > 1:
                    tt = torch.tensor([dt_alt], device=dev, dtype=torch.float64)
                    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                    dt_alt = float(tt.item())
                alt = {"conv_math": "f32 (v_mfma_f32_32x32x2_f32)", "steps": args.alt_steps,
                       "ms_per_step": 1e3 * dt_alt / args.alt_steps,
                       "value": (args.alt_steps * B * world / N_TRAIN) / dt_alt, "unit": "epochs/s"}
            finally:
                _fcn_alt.CONV_MATH = "bf16x6"

    # Kernel durations free of co-running kernels: in the timed region the two experts of IGN run on two HIP streams, so
    # a kernel's event bracket also contains the time it shared the CUs with the other expert.  A short serial pass
    # (same step, one stream) right after the timed region gives the isolated durations the roofline fractions refer to.
    iso = {}
    if args.config == "ign":
        from models.InterpGN import InterpGN as _IGN
        was = _IGN.expert_streams
        _IGN.expert_streams = False
        graphed = None                                   # the isolated pass runs eagerly
        try:
            step(0)
            torch.cuda.synchronize()
            _lib.timing_enable(True)                 # resets the registry
            for i in range(args.iso_steps):
                step(args.warmup + i)
            torch.cuda.synchronize()
            for lab in ("shp_fwd", "shp_bwd", "clconv_fwd", "clconv_dgrad", "clconv_wgrad"):
                ms, n = _lib.timing_read(lab)
                iso[lab] = (ms / max(1, args.iso_steps), n // max(1, args.iso_steps))
        finally:
            _IGN.expert_streams = was

    attn = {k: _lib.timing_read(k) for k in ("attn_fwd", "attn_bwd_dkdv", "attn_bwd_dq")} if args.config != "ign" else {}
    if rank == 0:
        res = {
            "metric": "epochs/sec (B=256, C=122, T=1000) IGN 3-class" if args.config == "ign" else
                      f"epochs/sec (B=256, C=122, T=1000) {args.config} baseline 3-class",
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
                  "parallelism": f"dp{world}", "final_loss": float(last), "hipgraph": bool(use_graph)}
        if rehearsal:
            common["rehearsal"] = "all ranks on cuda:0 over gloo: functional check of the N > 1 path, NOT a measurement"
        if args.config == "ign":
            groups = [(s.n, s.length) for s in model.sbm.shapelets]
            f_fwd, f_bwd = shapelet_algorithmic_flops(B, C, T, groups)
            # dominant kernel: the shapelet backward (3E of the 5E shapelet flops).  One step issues one launch per
            # length group; `achieved` aggregates the G launches of a step (algorithmic flops / measured device time).
            bwd_tflops = (f_bwd * args.steps) / (bwd_ms * 1e-3) / 1e12 if bwd_ms > 0 else 0.0
            fwd_tflops = (f_fwd * args.steps) / (fwd_ms * 1e-3) / 1e12 if fwd_ms > 0 else 0.0
            traffic = None
            try:    # HBM bytes per step of the dominant kernel, from the committed PMC passes (see profiles/traffic.json)
                if args.groups != "4x5":
                    raise KeyError("the PMC passes were collected for the headline bank only")
                traffic = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["shp_bwd_kernel"]["bytes_per_step"]
            except Exception:
                pass
            from ign_hip import fcn as _fcn
            res["config"] = dict(common, workload="Synthetic CHISCO-shape EEG (122ch x 1000, 3-class) IGN(FCN expert), "
                                 + ("driver-default groups K=5 x L{100,200,300,500}" if args.groups == "4x5" else
                                    "stress bank K=10 x L{50,100,200,300,500,800}") + ", Adam lr 5e-3, fp32",
                                 conv_math=_fcn.CONV_MATH + (" (fp32 operands split exactly into 3 bf16 terms, 6 partial products "
                                                            "accumulated in fp32; <= 3e-6 vs float64, same as the fp32-MFMA "
                                                            "kernel; IGN_CONV_MATH=f32 selects that one)"
                                                            if _fcn.CONV_MATH == "bf16x6" else ""))
            res["roofline"] = {"bound": "valu", "kernel": "shp_bwd_kernel", "achieved": bwd_tflops,
                               "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s", "frac": bwd_tflops / PEAK_FP32_VALU_TFLOPS,
                               "traffic": traffic, "ms_per_step": bwd_ms / max(1, args.steps), "launches": bwd_n,
                               "note": "timed region = two HIP streams (SBM expert || FCN expert): these brackets include "
                                       "time shared with the other expert's kernels; 'isolated' = the same kernels in a "
                                       "serial pass of --iso-steps steps right after the timed region",
                               "fwd_kernel": {"kernel": "shp_fwd_kernel", "achieved": fwd_tflops,
                                              "frac": fwd_tflops / PEAK_FP32_VALU_TFLOPS,
                                              "ms_per_step": fwd_ms / max(1, args.steps), "launches": fwd_n}}
            if alt is not None:
                res["fp32_mfma_conv"] = alt
            if iso:
                def _tf(flops, ms):
                    return flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
                x6 = _fcn.CONV_MATH == "bf16x6"

                def _conv_roof(tf_alg, ms):
                    # split-bf16: six bf16 MFMA products per algorithmic fp32 product, priced against the dense bf16 peak
                    if x6:
                        return {"ms_per_step": ms, "achieved": tf_alg, "unit": "TFLOP/s fp32-equivalent (algorithmic)",
                                "executed_bf16_tflops": 6.0 * tf_alg, "peak": 2500.0, "frac": 6.0 * tf_alg / 2500.0}
                    if _fcn.CONV_MATH == "bf16":
                        # operands rounded to bf16, one product (the reference's autocast arithmetic): priced against the bf16 peak
                        return {"ms_per_step": ms, "achieved": tf_alg, "unit": "TFLOP/s bf16 (one product, fp32 accumulate)",
                                "peak": 2500.0, "frac": tf_alg / 2500.0}
                    return {"ms_per_step": ms, "achieved": tf_alg, "peak": 157.3, "frac": tf_alg / 157.3}
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
            # attention core: fwd 4*B*H*L*S*E flop per layer (QK^T + PV); the two backward kernels execute 7 products
            # (S and dP are recomputed for dQ) but the ALGORITHMIC count is 5 products = 2.5x forward.
            if args.config == "transformer":
                Bh, Lq, E_, layers, desc = B * cfg.n_heads, T, cfg.d_model // cfg.n_heads, cfg.e_layers, \
                    "Transformer-encoder baseline (d_model 512, 8 heads, d_ff 2048, 2 layers)"
            else:
                Bh, Lq, E_, layers, desc = B * 8, T // 10, cfg.d_model // 8, 2, \
                    "EEG-CNN baseline (CNN 8x8 filters k=125/25 + 2-layer encoder d_model 512, 8 heads, d_ff 256)"
            f_attn_fwd = 4.0 * Bh * Lq * Lq * E_ * layers
            ms_f, ms_b = attn["attn_fwd"][0], attn["attn_bwd_dkdv"][0] + attn["attn_bwd_dq"][0]
            tf_f = f_attn_fwd * args.steps / (ms_f * 1e-3) / 1e12 if ms_f > 0 else 0.0
            tf_b = 2.5 * f_attn_fwd * args.steps / (ms_b * 1e-3) / 1e12 if ms_b > 0 else 0.0
            res["config"] = dict(common, workload=f"Synthetic CHISCO-shape EEG (122ch x 1000, 3-class) {desc}, Adam lr 5e-3, fp32")
            res["roofline"] = {"bound": "mfma", "kernel": "attn_bwd_dkdv_kernel+attn_bwd_dq_kernel", "achieved": tf_b,
                               "peak": 157.3, "unit": "TFLOP/s", "frac": tf_b / 157.3, "traffic": None,
                               "ms_per_step": ms_b / max(1, args.steps),
                               "fwd_kernel": {"kernel": "attn_fwd_kernel", "achieved": tf_f, "frac": tf_f / 157.3,
                                              "ms_per_step": ms_f / max(1, args.steps)}}
        if world == 1 and args.cpu_sample > 0 and args.config == "ign":
            log("cpu baseline ...")
            res["cpu_baseline"] = cpu_baseline(cfg, args.cpu_sample, host_cores())
        print(json.dumps(res), flush=True)
    _lib.timing_enable(False)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
