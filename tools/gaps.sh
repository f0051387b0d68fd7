#!/bin/bash
# Development tool (GPU box): idle gaps between consecutive kernels of a bench configuration, from the rocprofv3 kernel trace.  usage: tools/gaps.sh c5
CFG=${1:-c5}
OUT=gpurun_out/gaps_$CFG
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python bench.py --config $CFG --no-cpu-baseline --no-check --no-extras --steps 40 --warmup 5 > $OUT/bench.log 2>&1
python - "$OUT" <<'P'
import csv, glob, sys, statistics
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]) for r in csv.DictReader(open(f))), key=lambda r: r[0])
rows = rows[-160:]
gaps = {}
for a, b in zip(rows, rows[1:]):
    gaps.setdefault((a[2], b[2]), []).append(b[0] - a[1])
for k, v in gaps.items():
    print("%-42s -> %-42s n=%3d  median gap %.2f us  (min %.2f, max %.2f)" % (k[0], k[1], len(v), statistics.median(v) / 1e3, min(v) / 1e3, max(v) / 1e3))
dur = {}
for r in rows: dur.setdefault(r[2], []).append(r[1] - r[0])
for k, v in dur.items(): print("%-42s n=%3d  median duration %.2f us" % (k, len(v), statistics.median(v) / 1e3))
P
