#!/bin/bash
# Development tool: instruction-cache and wait counters of the fused kernel for variant builds (tools/build_variant.sh)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for n in ${ABL:-0}; do
  if [ $n = 0 ]; then so=""; else so="$PWD/imagepipe_amd/csrc/build/ablate/lib$n.so"; fi
  for pass in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY SQ_INSTS_VALU"; do
  rm -rf gpurun_out/pmci$n
  IPK_SO_OVERRIDE=$so rocprofv3 --kernel-trace --pmc $pass --output-format csv -d gpurun_out/pmci$n -o p -- python bench.py --no-cpu-baseline --no-check --steps 2 --warmup 1 --prewarm-ms 0 --data ${DATA:-noise} > /dev/null 2>&1
  python3 - <<PY
import csv, collections
rows=list(csv.DictReader(open('gpurun_out/pmci$n/p_counter_collection.csv')))
agg=collections.defaultdict(list)
for r in rows:
    if 'fused' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
it = 100e6/256
print('variant $n ${DATA:-noise}', {k: round(sum(v)/len(v)/it,1) for k,v in agg.items()}, '(per wave-iteration of 256 px)')
PY
  done
done
