#!/bin/bash
# Development tool: the shader clock DURING a kernel = GRBM_GUI_ACTIVE cycles of the dispatch / its duration from the kernel trace, for the headline
# kernel and for the instruction-class micro-benchmarks (run on the GPU box through gpurun).  Output: gpurun_out/clock/{fused,ubench}.txt
OUT=gpurun_out/clock
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/f -o f -- python bench.py --no-cpu-baseline --no-check --no-extras --steps 20 --warmup 3 > $OUT/bench.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/u -o u -- tools/build/ubench2 > $OUT/ubench2.log 2>&1
python - <<'P'
import csv, glob, collections
for tag in ("f", "u"):
    trace = glob.glob("gpurun_out/clock/%s/**/*kernel_trace.csv" % tag, recursive=True)
    ctr = glob.glob("gpurun_out/clock/%s/**/*counter_collection.csv" % tag, recursive=True)
    if not trace or not ctr: print(tag, "missing", trace, ctr); continue
    dur = {}
    for r in csv.DictReader(open(trace[0])):
        dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    agg = collections.defaultdict(lambda: [0, 0.0, 0])
    for r in csv.DictReader(open(ctr[0])):
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE": continue
        name, ns = dur.get(r["Dispatch_Id"], (r.get("Kernel_Name", "?"), 0))
        a = agg[name[:70]]; a[0] += 1; a[1] += float(r["Counter_Value"]); a[2] += ns
    with open("gpurun_out/clock/%s.txt" % ("fused" if tag == "f" else "ubench"), "w") as o:
        for name, (n, cyc, ns) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
            if ns: o.write("%-72s n=%4d  %.3f ms  %.0f cycles  %.3f GHz\n" % (name, n, ns / n / 1e6, cyc / n, cyc / ns))
P
head -5 $OUT/fused.txt; head -80 $OUT/ubench.txt
