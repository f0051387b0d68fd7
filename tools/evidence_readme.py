#!/usr/bin/env python
"""Writes the number-carrying parts of the documentation FROM the tracked evidence files, so that text cannot drift from them:
   DESIGN.md            the block between <!-- BEGIN GENERATED: at-a-glance --> and <!-- END GENERATED: at-a-glance -->
   profiles/README.md   the block between <!-- BEGIN GENERATED: round --> and <!-- END GENERATED: round -->
Sources: profiles/<TAG>_counters.json, _variants.csv, _staged_kernels.csv, _valu_model.json (tools/evidence_summarize.py).
usage: tools/evidence_readme.py [TAG]          rewrite both blocks
       tools/evidence_readme.py [TAG] --check  exit 1 when a block is stale (tests/test_docs_generated.py)"""
import csv, glob, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:] if not a.startswith("--")]
CHECK = "--check" in sys.argv
P = os.path.join(ROOT, "profiles")
tags = sorted({os.path.basename(f)[:3] for f in glob.glob(os.path.join(P, "r[0-9][0-9]_counters.json"))})
TAG = args[0] if args else tags[-1]


def load_csv(name):
    f = os.path.join(P, name)
    return list(csv.DictReader(open(f))) if os.path.exists(f) else []


C = json.load(open(os.path.join(P, TAG + "_counters.json")))
V = load_csv(TAG + "_variants.csv")
S = load_csv(TAG + "_staged_kernels.csv")
M = json.load(open(os.path.join(P, TAG + "_valu_model.json"))) if os.path.exists(os.path.join(P, TAG + "_valu_model.json")) else {}


def f3(x):
    return "%.3f" % x


def ks_line(e, alg=None):
    k = e["kernel_stats"]
    alg = alg or e.get("algorithmic_bytes_per_launch")
    return "%d calls, average %.1f µs (min %.1f, max %.1f) = %.2f TB/s = **%.3f of 8 TB/s**" % (
        k["calls"], k["average_us"], k["min_us"], k["max_us"], alg / (k["average_us"] * 1e-6) / 1e12, alg / (k["average_us"] * 1e-6) / 8e12)


def glance():
    rows = []
    h = C["headline"]
    rows.append(("headline kernel, rocprofv3 (`%s_kernel_stats.csv`)" % TAG, "`%s`: %s" % (h["kernel_stats"]["name"].replace("void ipk::", "").split("(")[0], ks_line(h))))
    if "traffic_over_algorithmic" in h:
        rows.append(("HBM traffic, PMC (`%s_counters.json`)" % TAG, "FETCH×2 + WRITE = %.3f GB per launch = **%.3f× algorithmic** (1.6 GB)" % (C["hbm_traffic_bytes_per_launch"] / 1e9, h["traffic_over_algorithmic"])))
    b = C.get("bench_line")
    if b:
        r = b["roofline"]
        rows.append(("bench line of that session (`python bench.py`)", "value %.0f MP/s, ms_per_step %.4f, kernel_ms %.4f (median %.4f), frac %.4f, cold %.4f ms" % (
            b["value"], b["ms_per_step"], r["kernel_ms"], r["kernel_ms_median"], r["frac"], b["config"].get("cold_ms", float("nan")))))
        if "ceiling_ms" in r:
            rows.append(("measured ceiling: the kernel's memory skeleton (`ipk_stream_probe`)", "ceiling_ms %.4f = %.3f of peak; **frac_of_ceiling %.3f**; 1:1 copy ceiling %.0f GB/s (frac_of_copy_ceiling %.3f)" % (
                r["ceiling_ms"], r["ceiling_frac_of_peak"], r["frac_of_ceiling"], r.get("copy_ceiling_GBps", float("nan")), r.get("frac_of_copy_ceiling", float("nan")))))
        if "mix_ceiling_GBps" in r:
            rows.append(("the traffic mix's own ceiling (`ipk_mix_probe`: 4 B read : 12 B written per pixel, flat, no arithmetic)", "%.0f GB/s = **%.3f of peak**; the fused kernel is at %.3f of it" % (
                r["mix_ceiling_GBps"], r["mix_ceiling_frac_of_peak"], r["frac_of_mix_ceiling"])))
        if "launch_stats" in r:
            ls = r["launch_stats"]
            t = "min %.4f / median %.4f / p95 %.4f / max %.4f ms, stddev %.1f %%, %d above 1.2× median" % (
                ls["min_ms"], ls["median_ms"], ls["p95_ms"], ls["max_ms"], 100 * ls["stddev_frac"], ls["over_1p2x_median"])
            if "after_50ms_idle" in ls:
                t += "; the 20 launches behind a 50 ms idle gap: first %.4f, max %.4f, mean %.4f ms (clock ramp)" % (ls["after_50ms_idle"]["first_ms"], ls["after_50ms_idle"]["max_ms"], ls["after_50ms_idle"]["mean_of_20_ms"])
            rows.append(("launch-to-launch spread, steady state (%d single launches)" % ls["launches"], t))
        if "other_data" in b:
            o = b["other_data"]
            t = ", ".join("%s %.4f ms (%.3f)" % (k, o[k]["kernel_ms"], o[k]["frac"]) for k in sorted(o))
            if "schedule_split" in o.get("photo", {}):
                t += "; photo under `IPK_SCHED_SPLIT` %.4f ms (%.3f)" % (o["photo"]["schedule_split"]["kernel_ms"], o["photo"]["schedule_split"]["frac"])
            rows.append(("other data kinds", t))
        cfg = b.get("config", {})
        if "shader_clock_GHz" in cfg:
            rows.append(("the box's state under the timed workload (`config.*`, `ipk_clock_probe` + hwmon)", "shader clock %.3f GHz (spans %.3f–%.3f)%s; **%.4f M shader cycles per launch** (`roofline.kernel_Mcycles` = kernel_ms × clock: the box-independent figure)" % (
                cfg["shader_clock_GHz"], cfg["shader_clock_GHz_min_max"][0], cfg["shader_clock_GHz_min_max"][1],
                ", socket power %.0f W (max %.0f)" % (cfg["socket_power_W"], cfg["socket_power_W_max"]) if "socket_power_W" in cfg else "", r.get("kernel_Mcycles", float("nan")))))
        hb = b.get("host_boundary")
        if hb and "to_u8" in hb:
            rows.append(("host boundary: 24 MP u16 in HOST memory → sRGB in HOST memory (`host_boundary`, `ipk_host_pipeline_run(_batch)`, page-locked)", "; ".join(
                "%s %.2f ms alone / **%.2f ms per frame batched** (full-duplex PCIe floor %.2f: %.2f%s)" % (
                    k.replace("to_", "→ "), hb[k]["ms_per_frame_single"], hb[k]["ms_per_frame_batched"], hb[k]["pcie_floor_ms"]["batched (directions overlap)"], hb[k]["frac_of_pcie_floor"]["batched"],
                    "; the link measured in the same run: %.0f / %.0f GB/s up / down alone, %.0f combined when both run -- one upload + one download at once take %.2f ms" % (
                        hb[k]["link_measured"]["up_GBps"], hb[k]["link_measured"]["down_GBps"], hb[k]["link_measured"]["combined_GBps_both_at_once"], hb[k]["link_measured"]["both_at_once_ms"])
                    if "link_measured" in hb[k] else "") for k in ("to_u8", "to_f32") if k in hb)))
        if "cpu_baseline" in b:
            cb = b["cpu_baseline"]
            rows.append(("CPU baseline (`kind: %s`)" % cb["kind"], "%.1f %s on %d cores" % (cb["value"], cb["unit"], cb["cores"])))
    sp1 = C.get("single_process_two_contexts_one_gpu", {}).get("scale")
    if sp1 and "compute_only" in sp1:
        t = "64 × 24 MP over two contexts on ONE GPU %.2f ms against %.2f ms on one context (%.2f×; expected %.0f× for %d physical device)" % (
            sp1["compute_only"]["ms"], sp1["n1_batch_ms"], sp1["compute_only"]["speedup"], sp1["compute_only"]["expected_speedup"], sp1["distinct_devices"])
        h2 = sp1.get("host_in_host_out_u8", {})
        if "ms_per_frame" in h2:
            t += "; host-to-host 8-bit %.2f ms per frame; same bytes as one context: %s" % (h2["ms_per_frame"], h2.get("same_bytes_as_one_context"))
        rows.append(("single-process multi-device mode (`bench.py --gpus 2 --single-process --devices 0,0`; no second GPU exists here)", t))
    if "stream_probe" in C:
        sp = C["stream_probe"]
        rows.append(("stream probe under rocprofv3", "%d calls, average %.1f µs = %.3f of peak" % (sp["calls"], sp["average_us"], sp["frac_of_8TBps"])))
    for key, label in (("c2", "configs[1] 24 MP, one frame"), ("c4", "configs[3] 64 × 24 MP, one launch"), ("c5", "configs[4] X-Trans 50 MP → 2160×1440, scaled-demosaic kernel")):
        e = C.get(key)
        if e and e.get("kernel_stats"):
            t = ks_line(e)
            if "traffic_over_algorithmic" in e:
                t += "; traffic %.2f×" % e["traffic_over_algorithmic"]
            rows.append((label + " (`%s_%s_kernel_stats.csv`)" % (TAG, key), t))
    if V:
        worst = max(V, key=lambda r: float(r["time_vs_common_variant"] or 0))
        rows.append(("all %d fused variants (`%s_variants.csv`)" % (len(V), TAG), "worst %s× the common variant (%s→%s, curve %s, linear %s); u8 output %s–%s µs, u16 %s–%s µs" % (
            worst["time_vs_common_variant"], worst["src"], worst["out"], worst["curve"], worst["linear"],
            min(float(r["avg_us"]) for r in V if r["out"] == "u8"), max(float(r["avg_us"]) for r in V if r["out"] == "u8"),
            min(float(r["avg_us"]) for r in V if r["out"] == "u16"), max(float(r["avg_us"]) for r in V if r["out"] == "u16"))))
    if S:
        rows.append(("staged kernels at 100 MP (`%s_staged_kernels.csv`)" % TAG, ", ".join("%s %.0f µs (%.2f)" % (r["kernel"].split("<")[0] if "fused" not in r["kernel"] else "demosaic-only", float(r["avg_us"]), float(r["frac_of_8TBps"])) for r in S)))
    for d in ("noise", "photo"):
        m = M.get(d)
        if m:
            rows.append(("VALU account, %s (`%s_valu_model.json`)" % (d, TAG), "%.1f VALU/px, issue floor %.3f ms (interleaved pricing) vs %.4f measured = %.3f; wave-cycle ratio %.3f; shader clock %.2f GHz" % (
                m["valu_per_pixel"], m["issue_ms_interleaved"], m["measured_kernel_ms_same_session"], m["frac_interleaved"], m["wave_cycles_per_valu"].get("ratio", float("nan")), m.get("shader_clock_GHz_during_kernel", float("nan")))))
    out = ["**At a glance** *(MI355X, round tag `%s`; generated by `tools/evidence_readme.py` from `profiles/%s_*` — do not edit by hand)*" % (TAG, TAG), "", "| | |", "|---|---|"]
    out += ["| %s | %s |" % r for r in rows]
    return "\n".join(out)


def round_block():
    out = ["## Round %d (tag `%s`)" % (int(TAG[1:]), TAG), "",
           "Collected on the MI355X box with `tools/evidence.sh %s` (one gpurun call; every `--pmc` pass is its own run with `--kernel-trace` only), condensed by" % TAG,
           "`tools/evidence_summarize.py %s`; this block is generated from those files by `tools/evidence_readme.py`." % TAG, "", "| file | content |", "|---|---|"]
    h = C["headline"]
    out.append("| `%s_kernel_stats.csv` | `rocprofv3 --kernel-trace --stats` of `python bench.py --no-cpu-baseline --no-check --no-extras` (100 MP f32 RGGB frame, noise, clock pre-warmed): %s |" % (TAG, ks_line(h)))
    for key, what in (("c4", "`bench.py --config c4` (64 × 24 MP, one `ipk_raw_to_srgb_batch` launch)"), ("c2", "`bench.py --config c2` (one 24 MP frame)"), ("c5", "`bench.py --config c5` (50 MP X-Trans → 2160×1440)")):
        e = C.get(key)
        if e and e.get("kernel_stats"):
            out.append("| `%s_%s_kernel_stats.csv` | %s: `%s` %s |" % (TAG, key, what, e["kernel_stats"]["name"].replace("void ipk::", "").split("(")[0], ks_line(e)))
    if "hbm_traffic_bytes_per_launch" in C:
        tr = ["headline %.3f GB = %.3f×" % (C["hbm_traffic_bytes_per_launch"] / 1e9, h["traffic_over_algorithmic"])]
        for key in ("c4", "c2", "c5"):
            if C.get(key, {}).get("traffic_over_algorithmic"):
                tr.append("%s %.2f×" % (key, C[key]["traffic_over_algorithmic"]))
        out.append("| `%s_counters.json` | HBM traffic per launch from separate `FETCH_SIZE` / `WRITE_SIZE` passes (FETCH × 2 per the gfx950 note): %s of the algorithmic bytes; the kernel-trace lines above; `stream_probe` (the fused kernel's memory skeleton under the trace%s); `bench_line` = the full default `python bench.py` line of that session |" % (
            TAG, ", ".join(tr), ": %.1f µs = %.3f of peak" % (C["stream_probe"]["average_us"], C["stream_probe"]["frac_of_8TBps"]) if "stream_probe" in C else ""))
    if V:
        out.append("| `%s_variants.csv` | one rocprofv3 line per fused variant a caller can reach (curve default / none / 5 user points × linear × f32 / u16 source × f32 / u8 / u16 output, 100 MP noise): %d variants, %s–%s µs, worst %s× its common variant |" % (
            TAG, len(V), min(float(r["avg_us"]) for r in V), max(float(r["avg_us"]) for r in V), max(float(r["time_vs_common_variant"] or 0) for r in V)))
    if S:
        out.append("| `%s_staged_kernels.csv` | the staged 100 MP pipeline's kernels (cache path): %s |" % (TAG, "; ".join("%s %.1f µs = %.3f" % (r["kernel"][:40], float(r["avg_us"]), float(r["frac_of_8TBps"])) for r in S)))
    if M.get("noise"):
        m = M["noise"]
        out.append("| `%s_valu_model.json`, `%s_ubench2.txt` | the VALU-issue account (instruction classes × `tools/ubench2.hip` prices): noise %.1f VALU/px, issue floor %.3f ms interleaved / %.3f additive, measured %.4f; photo-like %.1f VALU/px, %.3f ms; `bench.py` reads it for `roofline_valu` |" % (
            TAG, TAG, m["valu_per_pixel"], m["issue_ms_interleaved"], m["issue_ms_additive"], m["measured_kernel_ms_same_session"], M["photo"]["valu_per_pixel"], M["photo"]["issue_ms_interleaved"]))
    return "\n".join(out)


def splice(path, marker, text):
    s = open(path).read()
    a, b = "<!-- BEGIN GENERATED: %s -->" % marker, "<!-- END GENERATED: %s -->" % marker
    m = re.search(re.escape(a) + r".*?" + re.escape(b), s, flags=re.S)
    if not m:
        raise SystemExit("%s: markers for %r not found" % (path, marker))
    new = s[:m.start()] + a + "\n" + text + "\n" + b + s[m.end():]
    if CHECK:
        if new != s:
            sys.stderr.write("%s: generated block %r is stale -- run tools/evidence_readme.py\n" % (path, marker))
            return False
        return True
    open(path, "w").write(new)
    return True


ok = splice(os.path.join(ROOT, "DESIGN.md"), "at-a-glance", glance())
ok = splice(os.path.join(P, "README.md"), "round", round_block()) and ok
sys.exit(0 if ok else 1)
