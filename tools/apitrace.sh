#!/bin/bash
# Development tool (GPU box): the HIP runtime calls one bench step makes (rocprofv3 --hip-trace, no counters).  usage: tools/apitrace.sh c5
CFG=${1:-c5}
OUT=gpurun_out/api_$CFG
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --output-format csv -d $OUT -o t -- python bench.py --config $CFG --no-cpu-baseline --no-check --no-extras --steps 6 --warmup 2 --prewarm-ms 0 > $OUT/bench.log 2>&1
python - "$OUT" <<'P'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*hip_api_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Function"] for r in rows]
# the last 60 calls before the final synchronize
print(" ".join(names[-90:]))
m = glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True)
if m: print("memory copies:", sum(1 for _ in open(m[0])) - 1)
P
