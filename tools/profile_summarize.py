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
# ---- the VALU-issue model: dynamic instruction counts by class x measured issue cost per class ------------------------------------
ub = os.path.join(src, "ubench2.txt")
if os.path.exists(ub):
    shutil.copy(ub, os.path.join(dst, tag + "_ubench2.txt"))
    ns = {}
    for line in open(ub):
        f = line.split()
        if len(f) >= 7 and f[2] == "ms" and f[4] == "ns":
            ns[f[0]] = float(f[3])
    # issue cost per wave64 instruction per SIMD with 4 waves resident (tools/ubench2.hip): full-rate f32 (mul / add / fma), the half-rate
    # class (min / max / cmp / cndmask / cvt / fract / lshl ...), f64 arithmetic, the transcendental units
    cost = {"f32_add_mul_fma": round((ns["mul_vv"] + ns["add_vv"] + ns["fma_vvv"]) / 3, 3), "half_rate": round((ns["min"] + ns["cvt_u32"] + ns["fract"] + ns["cnd_e64"] + ns["cmp_e64"] + ns["lshl_add"]) / 6, 3),
            "cvt": ns["cvt_u32"], "int32": round((ns["sub_u32"] + ns["lshl_add"]) / 2, 3), "trans_f32": ns["rcp_f32"], "f64_arith": 1.95, "trans_f64": 6.8, "salu": 0.0}
    models = {}
    for d in ("noise", "photo"):
        c1, _ = counters("cls1_" + d)
        c2, _ = counters("cls2_" + d)
        if not c1 or not c2:
            continue
        c = dict(c1); c.update(c2)
        f32 = c["SQ_INSTS_VALU_ADD_F32"] + c["SQ_INSTS_VALU_MUL_F32"] + c["SQ_INSTS_VALU_FMA_F32"]
        f64 = c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + c["SQ_INSTS_VALU_FMA_F64"]
        listed = f32 + f64 + c["SQ_INSTS_VALU_TRANS_F32"] + c["SQ_INSTS_VALU_TRANS_F64"] + c["SQ_INSTS_VALU_INT32"] + c["SQ_INSTS_VALU_INT64"] + c["SQ_INSTS_VALU_CVT"]
        other = c["SQ_INSTS_VALU"] - listed          # v_min / max / med3 / cmp / cndmask / fract / mov / dpp ...: the counters have no class for them
        insts = {"f32_add_mul_fma": f32, "f64_arith": f64, "trans_f32": c["SQ_INSTS_VALU_TRANS_F32"], "trans_f64": c["SQ_INSTS_VALU_TRANS_F64"],
                 "int32": c["SQ_INSTS_VALU_INT32"] + c["SQ_INSTS_VALU_INT64"], "cvt": c["SQ_INSTS_VALU_CVT"], "half_rate": other, "salu": c.get("SQ_INSTS_SALU", 0)}
        simds = 1024
        pred = sum(insts[k] * cost[k] for k in insts) / simds * 1e-6
        m = {"insts_per_launch": {k: round(v) for k, v in insts.items()}, "valu_total_per_launch": round(c["SQ_INSTS_VALU"]), "ns_per_wave_inst": cost, "simds": simds,
             "predicted_ms": round(pred, 4), "raw_counters_per_launch": {k: round(v) for k, v in c.items()}}
        bd = os.path.join(src, "bench_%s.json" % d)
        if os.path.exists(bd):
            try:
                bl = json.loads(open(bd).read().strip().splitlines()[-1])
                m["measured_kernel_ms_same_session"] = bl["roofline"]["kernel_ms"]
                m["frac"] = round(pred / bl["roofline"]["kernel_ms"], 4)
            except Exception:
                pass
        models[d] = m
    if "noise" in models:
        vm = dict(models["noise"])
        vm["other_data"] = {k: v for k, v in models.items() if k != "noise"}
        vm["method"] = ("SQ_INSTS_VALU_* per-class counters of k_fused_bayer (rocprofv3 --pmc, two passes per data kind, tools/profile.sh) x the issue cost of each class measured "
                        "by tools/ubench2.hip on the same box (4 waves per SIMD, " + tag + "_ubench2.txt), summed and divided by the 1024 SIMDs of the chip. 'half_rate' = SQ_INSTS_VALU minus "
                        "every class the hardware counts separately (min / max / med3 / cmp / cndmask / fract / mov / dpp; mov and dpp are full-rate, so this slightly overestimates). "
                        "Scalar instructions are listed but priced at 0.")
        json.dump(vm, open(os.path.join(dst, tag + "_valu_model.json"), "w"), indent=1, sort_keys=True)
        print("valu model:", {k: (v["predicted_ms"], v.get("measured_kernel_ms_same_session")) for k, v in models.items()})

bj = os.path.join(src, "bench_plain.json")
if os.path.exists(bj):
    lines = [l for l in open(bj).read().splitlines() if l.startswith("{")]
    if lines:
        out["bench_line"] = json.loads(lines[-1])
json.dump(out, open(os.path.join(dst, tag + "_counters.json"), "w"), indent=1, sort_keys=True)
print(json.dumps({k: v for k, v in out.items() if k not in ("bench_line", "kernel_stats")}, indent=1))
