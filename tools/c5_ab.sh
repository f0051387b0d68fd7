#!/bin/bash
# Development tool: same-box A/B of variant libraries on config 5 (X-Trans 50 MP -> 2160x1440): per-kernel averages from rocprofv3's kernel trace
# usage (GPU box): VARIANTS="a b" tools/c5_ab.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in main ${VARIANTS}; do
  if [ $v = main ]; then so=""; else so="$PWD/imagepipe_amd/csrc/build/ablate/lib$v.so"; fi
  OUT=/tmp/c5ab_$v; rm -rf $OUT
  IPK_SO_OVERRIDE=$so rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o s -- python bench.py --config c5 --no-cpu-baseline --no-check --steps 400 --warmup 20 > $OUT.log 2>&1
  f=$(find $OUT -name "*kernel_stats.csv" | head -1)
  python3 - <<PY
import csv
for r in csv.DictReader(open('$f')):
    if 'ipk::' in r['Name'] and int(r['Calls']) >= 100:
        print('$v %-60s calls %5s avg %7.2f us min %7.2f' % (r['Name'][10:70], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3))
PY
done
done
