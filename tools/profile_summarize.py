#!/usr/bin/env python
"""Turns gpurun_out/prof_<TAG> (written by tools/profile.sh on the GPU box) into the committed evidence under profiles/:
   <TAG>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary of `python bench.py`
   <TAG>_counters.json      per-launch HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and SQ counters of the fused kernel
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)


def find(sub, pat):
    g = glob.glob(os.path.join(src, sub, "**", pat), recursive=True)
    return g[0] if g else None


stats = find("stats", "*kernel_stats.csv")
if stats:
    shutil.copy(stats, os.path.join(dst, tag + "_kernel_stats.csv"))
out = {"tag": tag, "command": "python bench.py  (100 MP f32 RGGB frame, fused raw->sRGB, 20 steps)"}


def counters(sub):
    f = find(sub, "*counter_collection.csv")
    agg = defaultdict(list)
    if f:
        for r in csv.DictReader(open(f)):
            if "fused" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}


fetch, nf = counters("fetch")
write, nw = counters("write")
sq, _ = counters("sq")
if stats:
    for r in csv.DictReader(open(stats)):
        if "fused" in r.get("Name", ""):
            out["kernel_stats"] = {k: r[k] for k in r}
# MI355X_MICROARCH.md "HBM": FETCH_SIZE/WRITE_SIZE are in KiB-class units of 1024 B; on gfx950 FETCH_SIZE reports exactly
# half the bytes of a wide (16 B/lane) coalesced streaming read -> doubled; WRITE_SIZE is uncalibrated -> reported as is.
if "FETCH_SIZE" in fetch:
    out["FETCH_SIZE_raw_per_launch"] = fetch["FETCH_SIZE"]
    out["fetch_bytes_per_launch_corrected"] = fetch["FETCH_SIZE"] * 1024 * 2
if "WRITE_SIZE" in write:
    out["WRITE_SIZE_raw_per_launch"] = write["WRITE_SIZE"]
    out["write_bytes_per_launch"] = write["WRITE_SIZE"] * 1024
if "fetch_bytes_per_launch_corrected" in out and "write_bytes_per_launch" in out:
    out["hbm_traffic_bytes_per_launch"] = out["fetch_bytes_per_launch_corrected"] + out["write_bytes_per_launch"]
    out["algorithmic_bytes_per_launch"] = 16 * 10000 * 10000
out["sq_counters_per_launch"] = sq
bj = os.path.join(src, "bench_plain.json")
if os.path.exists(bj):
    lines = [l for l in open(bj).read().splitlines() if l.startswith("{")]
    if lines:
        out["bench_line"] = json.loads(lines[-1])
json.dump(out, open(os.path.join(dst, tag + "_counters.json"), "w"), indent=1, sort_keys=True)
print(json.dumps({k: v for k, v in out.items() if k not in ("bench_line", "kernel_stats")}, indent=1))
