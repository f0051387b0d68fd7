#!/bin/bash
# Development tool: same-box A/B of variant libraries on the STAGED 100 MP pipeline: per-kernel averages from rocprofv3 (library kernels only).
# usage (GPU box): VARIANTS="old new" tools/staged_ab.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in ${VARIANTS}; do
  OUT=gpurun_out/staged_ab_$v; rm -rf $OUT; mkdir -p $OUT
  IPK_SO_OVERRIDE=$PWD/imagepipe_amd/csrc/build/ablate/lib$v.so ONLY=${CFG:-C3} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o s -- python tools/bench_configs.py > $OUT/bench.log 2>&1
  f=$(find $OUT -name "*kernel_stats.csv" | head -1)
  python3 - <<PY
import csv
for r in csv.DictReader(open('$f')):
    if 'ipk::' in r['Name'] and int(r['Calls']) >= 4 and 'selftest' not in r['Name']:
        print('$v %-90s calls %4s avg %8.1f us min %8.1f' % (r['Name'][:90], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3))
PY
done
