#!/bin/bash
# Development tool: per-kernel averages (rocprofv3 --kernel-trace --stats) of the staged 100 MP pipeline (ONLY=C3 tools/bench_configs.py) and of the
# X-Trans full-resolution frame (C5b); library kernels only.  usage (GPU box): tools/staged_stats.sh [TAG]
TAG=${1:-r03}
OUT=gpurun_out/staged_$TAG
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf $OUT; mkdir -p $OUT
for cfg in ${CFGS:-C3 C5b}; do
  ONLY=$cfg rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$cfg -o s -- python tools/bench_configs.py > $OUT/bench_$cfg.log 2>&1
  f=$(find $OUT/stats_$cfg -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && grep -E '^"Name"|ipk::' $f | cut -c1-400 > $OUT/${cfg}_kernel_stats.csv
  echo "== $cfg"; python3 - <<PY
import csv
for r in csv.DictReader(open('$OUT/${cfg}_kernel_stats.csv')):
    print('%-110s calls %5s avg %9.1f us  min %9.1f us' % (r['Name'][:110], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3))
PY
  tail -n 3 $OUT/bench_$cfg.log | cut -c1-300
done
