#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files per kernel family (sum over launches / number of steps).
Usage: summarize_pmc.py <steps> <counter_collection.csv> [...]   -> prints JSON {kernel: {counter: value_per_step}}"""
import collections, csv, json, re, sys

def family(name):
    m = re.match(r"(?:void )?([A-Za-z0-9_:]+)", name)
    return m.group(1) if m else name

steps = int(sys.argv[1])
out = collections.defaultdict(lambda: collections.defaultdict(float))
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        out[family(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"]) / steps
print(json.dumps({k: dict(v) for k, v in out.items() if k.startswith(("shp_", "instnorm", "attn_", "head_", "adam", "reduce_parts", "clconv_", "bn_", "wgrad_reduce", "gather_flat", "chan_", "affine_", "dwconv", "autocorr", "layernorm"))}, indent=1, sort_keys=True))
