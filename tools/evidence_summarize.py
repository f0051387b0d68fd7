#!/usr/bin/env python
"""Turns gpurun_out/evid_<TAG> (tools/evidence.sh, run on the GPU box) into the tracked evidence under profiles/:
   <TAG>_kernel_stats.csv, <TAG>_c4/_c2/_c5_kernel_stats.csv   rocprofv3 --kernel-trace --stats summaries
   <TAG>_counters.json      HBM traffic per launch (FETCH_SIZE x 2 per MI355X_MICROARCH.md's gfx950 note, WRITE_SIZE; separate passes), headline and configs
   <TAG>_valu_model.json    the VALU-issue account of k_fused_bayer: instruction counts by class, priced two ways, the wave-cycle ratio and the stall split
   <TAG>_ubench2.txt        the micro-benchmark figures the prices come from
   <TAG>_variants.csv       one rocprofv3 line per fused variant (curve x linear x source x output) with its HBM and per-pixel figures
   <TAG>_staged_kernels.csv the staged 100 MP pipeline's kernels (the cache path), each with its algorithmic bytes and fraction of the HBM peak
"""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "evid_" + tag)
dst = os.path.join(root, "profiles")


def find(sub, pat):
    g = glob.glob(os.path.join(src, sub, "**", pat), recursive=True)
    return g[0] if g else None


def counters(sub, needle):
    f = find(sub, "*counter_collection.csv")
    agg = defaultdict(lambda: defaultdict(list))
    if f:
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, c in agg.items():
        if needle in k:
            for n, v in c.items():
                out.setdefault(n, []).extend(v)
    return {n: sum(v) / len(v) for n, v in out.items()}


def kernel_stats(sub, name, needle):
    f = find(sub, "*kernel_stats.csv")
    if not f:
        return None
    shutil.copy(f, os.path.join(dst, name))
    for r in csv.DictReader(open(f)):
        if needle in r.get("Name", ""):
            return {"name": r["Name"][:120], "calls": int(r["Calls"]), "average_us": float(r["AverageNs"]) / 1e3, "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3}


def bench_line(path):
    try:
        lines = [l for l in open(path).read().splitlines() if l.startswith("{")]
        return json.loads(lines[-1]) if lines else None
    except Exception:
        return None


out = {"tag": tag, "note": "FETCH_SIZE is doubled (gfx950 reports half the bytes of 16-B-per-lane streaming reads, MI355X_MICROARCH.md); WRITE_SIZE as reported; units of 1024 B; "
                           "one --pmc pass per counter group, --kernel-trace only"}
hs = kernel_stats("stats", tag + "_kernel_stats.csv", "k_fused_bayer<")
out["headline"] = {"command": "python bench.py --no-cpu-baseline --no-check --no-extras (10000x10000 RGGB f32 -> f32 RGB, noise)", "kernel_stats": hs,
                   "algorithmic_bytes_per_launch": 16 * 10000 * 10000}
f, w = counters("fetch", "k_fused_bayer"), counters("write", "k_fused_bayer")
if "FETCH_SIZE" in f and "WRITE_SIZE" in w:
    out["headline"].update(fetch_bytes_per_launch=f["FETCH_SIZE"] * 2048, write_bytes_per_launch=w["WRITE_SIZE"] * 1024)
    out["hbm_traffic_bytes_per_launch"] = f["FETCH_SIZE"] * 2048 + w["WRITE_SIZE"] * 1024
    out["headline"]["traffic_over_algorithmic"] = round(out["hbm_traffic_bytes_per_launch"] / (16e8), 4)
for c, needle, alg, what in (("c4", "k_fused_bayer_batch", 64 * 16 * 6000 * 4000, "bench.py --config c4: 64 x 6000x4000 f32 frames, one ipk_raw_to_srgb_batch launch"),
                             ("c2", "k_fused_bayer<", 16 * 6000 * 4000, "bench.py --config c2: one 6000x4000 f32 frame"),
                             ("c5", "k_raw_scaled_demosaic", 4 * 8640 * 5760 + 16 * 2160 * 1440, "bench.py --config c5: 8640x5760 X-Trans f32 -> 2160x1440, the scaled-demosaic kernel")):
    ks = kernel_stats("stats_" + c, "%s_%s_kernel_stats.csv" % (tag, c), needle)
    e = {"command": what, "kernel_stats": ks, "algorithmic_bytes_per_launch": alg}
    f, w = counters("fetch_" + c, needle), counters("write_" + c, needle)
    if "FETCH_SIZE" in f and "WRITE_SIZE" in w:
        e.update(fetch_bytes_per_launch=f["FETCH_SIZE"] * 2048, write_bytes_per_launch=w["WRITE_SIZE"] * 1024,
                 traffic_over_algorithmic=round((f["FETCH_SIZE"] * 2048 + w["WRITE_SIZE"] * 1024) / alg, 4))
    if ks:
        e["achieved_GBps"] = round(alg / (ks["average_us"] * 1e-6) / 1e9, 1)
        e["frac_of_8TBps"] = round(alg / (ks["average_us"] * 1e-6) / 8e12, 4)
    bl = bench_line(os.path.join(src, "bench_%s.log" % c))
    if bl:
        e["bench_line_roofline"] = bl.get("roofline")
        e["bench_ms_per_step"] = bl.get("ms_per_step")
    out[c] = e
# ---- the stream probe (the fused kernel's memory skeleton) under the kernel trace ------------------------------------------------------------
f = find("stats_probe", "*kernel_stats.csv")
if f:
    for r in csv.DictReader(open(f)):
        if "k_fused_bayer<float, true, 4," in r.get("Name", ""):
            out["stream_probe"] = {"kernel": r["Name"][:100], "calls": int(r["Calls"]), "average_us": float(r["AverageNs"]) / 1e3, "min_us": float(r["MinNs"]) / 1e3,
                                   "algorithmic_bytes_per_launch": 16e8, "frac_of_8TBps": round(16e8 / (float(r["AverageNs"]) * 1e-9) / 8e12, 4),
                                   "command": "python bench.py --no-cpu-baseline --no-check --steps 20 (its ceiling leg: ipk_stream_probe on the headline frame)"}
    bl = bench_line(os.path.join(src, "bench_probe.log"))
    if bl:
        out["stream_probe_bench_roofline"] = bl.get("roofline")

# ---- every fused variant -----------------------------------------------------------------------------------------------------------------------
rows = []
for aj in sorted(glob.glob(os.path.join(src, "variants", "v*.args")), key=lambda p: int(os.path.basename(p)[1:-5])):
    vdir = aj[:-5]
    bl = bench_line(vdir + ".json")
    ks = None
    g = glob.glob(os.path.join(vdir, "**", "*kernel_stats.csv"), recursive=True)
    if g:
        best = None
        for r in csv.DictReader(open(g[0])):
            if "k_fused_bayer<" in r.get("Name", "") and (best is None or float(r["TotalDurationNs"]) > float(best["TotalDurationNs"])):
                best = r
        ks = best
    if not (bl and ks):
        continue
    c = bl["config"]
    bpp = {"f32": 4, "u16": 2}[c["src"]] + {"f32": 12, "u8": 3, "u16": 6}[c["out"]]
    avg = float(ks["AverageNs"]) / 1e3
    rows.append({"src": c["src"], "out": c["out"], "curve": c["curve"], "linear": int(bool(c["linear"])), "exposure": c["exposure"],
                 "kernel": ks["Name"].replace("void ipk::", "").split("(")[0], "calls": int(ks["Calls"]), "avg_us": round(avg, 1), "min_us": round(float(ks["MinNs"]) / 1e3, 1),
                 "stddev_us": round(float(ks.get("StdDev", 0) or 0) / 1e3, 1), "alg_bytes_per_px": bpp, "achieved_GBps": round(bpp * 1e8 / (avg * 1e-6) / 1e9, 1),
                 "frac_of_8TBps": round(bpp * 1e8 / (avg * 1e-6) / 8e12, 4), "px_per_ns": round(1e8 / (avg * 1e3), 3), "bench_kernel_ms": bl["roofline"]["kernel_ms"]})
if rows:
    base = {(r["src"], r["out"]): r["avg_us"] for r in rows if r["curve"] == "default" and r["linear"] == (1 if r["out"] == "u16" else 0)}
    for r in rows:
        b = base.get((r["src"], r["out"]))
        r["time_vs_common_variant"] = round(r["avg_us"] / b, 3) if b else ""
    with open(os.path.join(dst, tag + "_variants.csv"), "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
    out["variants"] = {"file": tag + "_variants.csv", "n": len(rows), "worst_time_vs_common": max((r["time_vs_common_variant"] or 0) for r in rows),
                       "note": "100 MP frame, noise; one rocprofv3 --kernel-trace --stats run per variant; the u8 / u16 outputs are the output_8bit / output_16bit "
                               "boundary (5-7 / 8-10 B per pixel): bound by VALU issue, not HBM -- px_per_ns is the comparable figure across outputs"}

# ---- the staged kernels of the 100 MP pipeline (the cache path) -----------------------------------------------------------------------------
f = find("stats_staged", "*kernel_stats.csv")
if f:
    alg = [("k_gofloat_cfa", 8e8, "OpGoFloat 4+4 B/px"), ("k_fused_bayer<float, true, 3,", 20e8, "OpDemosaic (row walker, demosaic only) 4+16"), ("k_pointwise_chain<true>", 28e8, "OpToLab 16+12"),
           ("k_basecurve", 24e8, "OpBaseCurve 12+12"), ("k_fromlab", 24e8, "OpFromLab 12+12"), ("k_gamma", 24e8, "OpGamma 12+12"), ("k_output8", 15e8, "output8bit 12+3"),
           ("k_output16", 18e8, "output16bit 12+6")]
    srows = []
    for r in csv.DictReader(open(f)):
        for needle, b, what in alg:
            if needle in r.get("Name", ""):
                avg = float(r["AverageNs"]) / 1e3
                srows.append({"stage": what, "kernel": r["Name"].replace("void ipk::", "").replace("ipk::", "").split("(")[0], "calls": int(r["Calls"]), "avg_us": round(avg, 1),
                              "min_us": round(float(r["MinNs"]) / 1e3, 1), "alg_bytes": int(b), "achieved_GBps": round(b / (avg * 1e-6) / 1e9, 1), "frac_of_8TBps": round(b / (avg * 1e-6) / 8e12, 4)})
    if srows:
        with open(os.path.join(dst, tag + "_staged_kernels.csv"), "w", newline="") as fh:
            w = csv.DictWriter(fh, fieldnames=list(srows[0].keys())); w.writeheader(); w.writerows(srows)
        out["staged"] = {"file": tag + "_staged_kernels.csv", "command": "ONLY=C3 python tools/bench_configs.py (100 MP RGGB f32, staged pipeline) under rocprofv3 --kernel-trace --stats",
                         "frac_of_8TBps": {r["kernel"]: r["frac_of_8TBps"] for r in srows}}

bl = bench_line(os.path.join(src, "bench_plain.json"))
if bl:
    out["bench_line"] = bl
bl = bench_line(os.path.join(src, "bench_sp.json"))
if bl:
    out["single_process_two_contexts_one_gpu"] = {"command": "python bench.py --gpus 2 --single-process --devices 0,0 --no-cpu-baseline", "value": bl.get("value"),
                                                  "ms_per_step": bl.get("ms_per_step"), "scale": bl.get("scale")}
json.dump(out, open(os.path.join(dst, tag + "_counters.json"), "w"), indent=1, sort_keys=True)

# ---- the VALU-issue account ---------------------------------------------------------------------------------------------------------------
ub = os.path.join(src, "ubench2.txt")
ns = {}
if os.path.exists(ub):
    shutil.copy(ub, os.path.join(dst, tag + "_ubench2.txt"))
    for line in open(ub):
        p = line.split()
        if len(p) >= 7 and p[2] == "ms" and p[4] == "ns":
            ns[p[0]] = float(p[3])
model = {"method": (
    "Per launch of k_fused_bayer: SQ_INSTS_VALU_* class counters (two --pmc passes per data kind).  Two prices per wave64 instruction per SIMD, both from "
    "tools/ubench2.hip at 4 waves per SIMD on the same box: 'additive' -- every class at the cost of a loop of nothing but that class (f32 add/mul/fma, the half-rate "
    "class min/max/cmp/cndmask/cvt/fract/lshl, integer, f64, transcendental), summed: an UPPER bound on the issue time, because half-rate instructions hide behind "
    "full-rate ones when they alternate; 'interleaved' -- every f32 / half-rate / conversion / integer instruction at the cost of one instruction of seq_real, a loop "
    "with the kernel's kind of mix (mul, add, fma interleaved with cvt, fract, shifts, min, cmp, cndmask, med3), f64 and transcendental at their own cost: the "
    "issue time if the kernel interleaved as well as that loop does.  frac = issue time / measured time; the bench line carries the interleaved (lower) one as frac.  "
    "wave_cycles_per_valu: SQ_WAVE_CYCLES / SQ_INSTS_VALU, the resident-wave time per VALU instruction, for the kernel and for the seq_real loop -- model-free.  "
    "stall_split: SQ_WAIT_ANY (wave parked at s_waitcnt / barrier), SQ_WAIT_INST_ANY (ready, waiting for an issue slot), SQ_ACTIVE_INST_* as shares of "
    "SQ_WAVE_CYCLES; the three first add up to ~1."), "simds": 1024}
if ns:
    add = {"f32_add_mul_fma": round((ns["mul_vv"] + ns["add_vv"] + ns["fma_vvv"]) / 3, 3) if "mul_vv" in ns else round((ns["mul"] + ns["add"] + ns["fmaak"]) / 3, 3),
           "half_rate": round(sum(ns[k] for k in ("min", "cvt_u32", "fract", "cnd_e64", "cmp_e64", "lshl_add")) / 6, 3), "cvt": ns["cvt_u32"],
           "int32": round((ns["sub_u32"] + ns["lshl_add"]) / 2, 3), "trans_f32": ns.get("rcp_f32", 3.42), "f64_arith": ns.get("fma_f64", 1.95), "trans_f64": ns.get("rcp_f64", 6.8)}
    mix = dict(add)
    for k in ("f32_add_mul_fma", "half_rate", "cvt", "int32"):
        mix[k] = ns["seq_real"]
    model["ns_per_wave_inst"] = {"additive": add, "interleaved": mix}
uw = counters("ubench", "k_seq_real")
for d in ("noise", "photo"):
    c = dict(counters("cls1_" + d, "k_fused_bayer")); c.update(counters("cls2_" + d, "k_fused_bayer"))
    if "SQ_INSTS_VALU" not in c:
        continue
    f32 = c["SQ_INSTS_VALU_ADD_F32"] + c["SQ_INSTS_VALU_MUL_F32"] + c["SQ_INSTS_VALU_FMA_F32"]
    f64 = c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + c["SQ_INSTS_VALU_FMA_F64"]
    listed = f32 + f64 + c["SQ_INSTS_VALU_TRANS_F32"] + c["SQ_INSTS_VALU_TRANS_F64"] + c["SQ_INSTS_VALU_INT32"] + c["SQ_INSTS_VALU_INT64"] + c["SQ_INSTS_VALU_CVT"]
    insts = {"f32_add_mul_fma": f32, "f64_arith": f64, "trans_f32": c["SQ_INSTS_VALU_TRANS_F32"], "trans_f64": c["SQ_INSTS_VALU_TRANS_F64"],
             "int32": c["SQ_INSTS_VALU_INT32"] + c["SQ_INSTS_VALU_INT64"], "cvt": c["SQ_INSTS_VALU_CVT"], "half_rate": c["SQ_INSTS_VALU"] - listed}
    m = {"insts_per_launch": {k: round(v) for k, v in insts.items()}, "valu_total_per_launch": round(c["SQ_INSTS_VALU"]), "salu_per_launch": round(c.get("SQ_INSTS_SALU", 0)),
         "valu_per_pixel": round(c["SQ_INSTS_VALU"] * 64 / 1e8, 1)}
    bl = bench_line(os.path.join(src, "bench_%s.json" % d))
    meas = bl["roofline"]["kernel_ms"] if bl else None
    m["measured_kernel_ms_same_session"] = meas
    if ns:
        for name in ("additive", "interleaved"):
            pred = sum(insts[k] * model["ns_per_wave_inst"][name][k] for k in insts) / 1024 * 1e-6
            m["issue_ms_" + name] = round(pred, 4)
            if meas:
                m["frac_" + name] = round(pred / meas, 4)
    s = dict(counters("stall1_" + d, "k_fused_bayer")); s.update(counters("stall2_" + d, "k_fused_bayer")); s.update(counters("stall3_" + d, "k_fused_bayer"))
    if "SQ_WAVE_CYCLES" in s:
        wc = s["SQ_WAVE_CYCLES"]
        m["stall_split"] = {k: round(s[k] / wc, 4) for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
                                                            "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC", "SQ_ACTIVE_INST_VMEM") if k in s}
        m["per_launch_millions"] = {k: round(v / 1e6, 2) for k, v in s.items()}
        m["wave_cycles_per_valu"] = {"kernel": round(wc / s["SQ_INSTS_VALU"], 3)}
        if "SQ_WAVE_CYCLES" in uw and uw.get("SQ_INSTS_VALU"):
            m["wave_cycles_per_valu"]["seq_real_loop"] = round(uw["SQ_WAVE_CYCLES"] / uw["SQ_INSTS_VALU"], 3)
            m["wave_cycles_per_valu"]["ratio"] = round(m["wave_cycles_per_valu"]["seq_real_loop"] / m["wave_cycles_per_valu"]["kernel"], 4)
        if "GRBM_GUI_ACTIVE" in s and meas:
            m["shader_clock_GHz_during_kernel"] = round(s["GRBM_GUI_ACTIVE"] / 8 / (meas * 1e-3) / 1e9, 3)
    model[d] = m
json.dump(model, open(os.path.join(dst, tag + "_valu_model.json"), "w"), indent=1, sort_keys=True)
print(json.dumps({k: v for k, v in out.items() if k != "bench_line"}, indent=1)[:3000])
print(json.dumps({d: {k: v for k, v in model.get(d, {}).items() if k not in ("per_launch_millions",)} for d in ("noise", "photo")}, indent=1))
