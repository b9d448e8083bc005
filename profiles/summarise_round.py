#!/usr/bin/env python3
"""Turn the scratch output of profiles/collect_round.sh (gpurun_out/r3_*) into the committed evidence under profiles/:
bench lines, rocprofv3 kernel statistics, the HBM traffic table (profiles/traffic.json) and the matrix-pipe counters of the
GEMM / attention kernels for both split arithmetics.   python profiles/summarise_round.py [tag]"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
T = sys.argv[1] if len(sys.argv) > 1 else "r3"


def one_json_line(path):
    for line in open(path):
        if line.startswith("{"):
            return json.loads(line)
    raise SystemExit(f"{path}: no JSON line")


def family(name, keep_np=False):
    m = re.match(r"(?:void )?([A-Za-z0-9_:]+)(<[^>]*>)?", name)
    if not m:
        return name
    fam = m.group(1)
    if keep_np and m.group(2):                       # the last template argument of the split kernels is the plane count
        fam += "<planes=" + m.group(2)[1:-1].split(",")[-1].strip() + ">"
    return fam


def pmc_table(pattern, steps, keep_np=False):
    out = collections.defaultdict(lambda: collections.defaultdict(float))
    for path in glob.glob(os.path.join(O, pattern, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            out[family(r["Kernel_Name"], keep_np)][r["Counter_Name"]] += float(r["Counter_Value"]) / steps
    return {k: dict(v) for k, v in out.items()}


def copy(src, dst):
    if os.path.exists(src):
        shutil.copyfile(src, os.path.join(P, dst))
        print("wrote", dst)
    else:
        print("MISSING", src)


# ---- bench lines
for src, dst in ((f"{T}_bench.json", f"{T}_bench_line.json"), (f"{T}_bm32_eager.json", f"{T}_bm32_eager_bench_line.json"),
                 (f"{T}_bm32_graph.json", f"{T}_bm32_graph_bench_line.json"), (f"{T}_bench_bf16.json", f"{T}_bench_line_bf16_autocast.json"),
                 (f"{T}_bench_6x10.json", f"{T}_bench_line_6x10.json"), (f"{T}_bench_convx6.json", f"{T}_bench_line_conv_bf16x6.json"),
                 (f"{T}_tr_x6.json", f"{T}_transformer_bench_line_bf16x6.json"),
                 (f"{T}_tr_bf16.json", f"{T}_transformer_bench_line_bf16_autocast.json"),
                 (f"{T}_ign_ResNet.json", f"{T}_bench_line_ign_ResNet.json"), (f"{T}_ign_PatchTST.json", f"{T}_bench_line_ign_PatchTST.json"),
                 (f"{T}_ign_TimesNet.json", f"{T}_bench_line_ign_TimesNet.json")):
    p = os.path.join(O, src)
    if os.path.exists(p):
        json.dump(one_json_line(p), open(os.path.join(P, dst), "w"), indent=1)
        print("wrote", dst)
    else:
        print("MISSING", src)
for src, dst in ((f"{T}_prof_eegcnn.json", f"{T}_eegcnn_bench_line.json"), (f"{T}_prof_tr.json", f"{T}_transformer_bench_line.json")):
    p = os.path.join(O, src)
    if os.path.exists(p):
        json.dump(one_json_line(p), open(os.path.join(P, dst), "w"), indent=1)
copy(os.path.join(O, f"{T}_parity.json"), f"parity_{T}.json")
for d, dst in ((f"{T}_prof_ign/ign_kernel_stats.csv", f"{T}_bench_kernel_stats.csv"),
               (f"{T}_prof_ign_serial/ign_kernel_stats.csv", f"{T}_bench_kernel_stats_serial.csv"),
               (f"{T}_prof_eegcnn/eegcnn_kernel_stats.csv", f"{T}_eegcnn_kernel_stats.csv"),
               (f"{T}_prof_tr/tr_kernel_stats.csv", f"{T}_transformer_kernel_stats.csv")):
    hits = glob.glob(os.path.join(O, os.path.dirname(d), "**", os.path.basename(d)), recursive=True)
    copy(hits[0] if hits else os.path.join(O, d), dst)

# ---- HBM traffic (two PMC passes over the serial IGN step: 4 + 2 + 1 iso + 1 = 8 steps in the process)
STEPS_IGN = 8
fetch = pmc_table(f"{T}_pmc_ign_FETCH_SIZE", STEPS_IGN)
write = pmc_table(f"{T}_pmc_ign_WRITE_SIZE", STEPS_IGN)
WIDE = ("clconv_", "bn_", "reduce_", "wgrad_reduce", "absmax", "xcorr", "layernorm", "head_", "adam", "gather_flat")   # 16-byte loads
traffic = {"_how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes, MI355X_MICROARCH.md 'HBM'), round 3 "
                   "(profiles/collect_round.sh): IGN_EXPERT_STREAMS=0 bench.py --steps 4 --warmup 2 --iso-steps 1 = 8 steps in the process, "
                   "summed over the launches and divided by 8; bytes = (FETCH_SIZE*f + WRITE_SIZE) * 1024 (KB units); f = 1 for kernels with "
                   "4-byte-per-lane loads (shapelet, instance norm), f = 2 for kernels whose loads are 16 B per lane (convolution GEMMs, "
                   "BatchNorm passes, reductions), as the guide prescribes for gfx950.",
           "source": f"profiles/collect_round.sh, round 3 ({T})"}
for k in sorted(set(fetch) | set(write)):
    if not k.startswith(("shp_", "instnorm", "clconv_", "bn_", "reduce_", "wgrad_reduce", "head_", "adam", "gather_flat", "absmax", "fcn_scan",
                         "sbm_reg", "pack_weights", "ign_loss", "gate_")):
        continue
    f = 2 if k.startswith(WIDE) else 1
    fk, wk = fetch.get(k, {}).get("FETCH_SIZE", 0.0), write.get(k, {}).get("WRITE_SIZE", 0.0)
    traffic[k] = {"fetch_kb_per_step": fk, "write_kb_per_step": wk, "fetch_factor": f, "bytes_per_step": (fk * f + wk) * 1024.0}
if len(traffic) > 2:
    json.dump(traffic, open(os.path.join(P, "traffic.json"), "w"), indent=1)
    json.dump(traffic, open(os.path.join(P, f"{T}_pmc_traffic_summary.json"), "w"), indent=1)
    print("wrote traffic.json:", {k: round(v["bytes_per_step"] / 1e9, 3) for k, v in traffic.items() if isinstance(v, dict) and v["bytes_per_step"] > 5e7})

# ---- matrix-pipe counters of the GEMM / attention kernels, both arithmetics
pipe = {"_how": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU "
                "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY (one pass per arithmetic: IGN_CONV_MATH / IGN_GEMM_MATH = f16x3 | "
                "bf16x6) over the serial IGN step and the Transformer step; per kernel family, sums over the launches.  "
                "mfma_busy_fraction_of_4_simd = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES); the *_over_wave_cycles ratios are "
                "fractions of SQ_WAVE_CYCLES.  The f16x3 kernels execute HALF the matrix instructions of the bf16x6 kernels for the same "
                "result, so a lower pipe-busy fraction at a shorter duration is the intended outcome, not a regression."}
for wl, steps, pat in (("ign_serial", STEPS_IGN, f"{T}_pmc_sq_%s"), ("transformer", 3, f"{T}_pmc_sq_tr_%s")):
    for m in ("f16x3", "bf16x6"):
        tab = pmc_table(pat % m, steps, keep_np=True)
        sel = {}
        for k, v in tab.items():
            if not k.startswith(("clconv_", "attn_")) or "SQ_BUSY_CU_CYCLES" not in v or v["SQ_BUSY_CU_CYCLES"] <= 0:
                continue
            w = max(v.get("SQ_WAVE_CYCLES", 0.0), 1.0)
            sel[k] = dict(v, mfma_busy_fraction_of_4_simd=v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * v["SQ_BUSY_CU_CYCLES"]),
                          coexec_over_mfma_busy=v.get("SQ_VALU_MFMA_COEXEC_CYCLES", 0.0) / max(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), 1.0),
                          valu_active_over_wave_cycles=v.get("SQ_ACTIVE_INST_VALU", 0.0) / w,
                          wait_any_over_wave_cycles=v.get("SQ_WAIT_ANY", 0.0) / w, wait_inst_over_wave_cycles=v.get("SQ_WAIT_INST_ANY", 0.0) / w)
        if sel:
            pipe[f"{wl}:{m}"] = sel
if len(pipe) > 1:
    json.dump(pipe, open(os.path.join(P, f"{T}_pmc_gemm_pipe.json"), "w"), indent=1, sort_keys=True)
    for k, v in pipe.items():
        if isinstance(v, dict):
            print(k, {kk: round(vv["mfma_busy_fraction_of_4_simd"], 3) for kk, vv in v.items()})
