#!/bin/bash
# Development tool: dynamic instruction counts of the fused kernel for the ablation builds
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for n in ${ABL:-0 1 4}; do
  if [ $n = 0 ]; then so=""; else so="$PWD/imagepipe_amd/csrc/build/ablate/lib$n.so"; fi
  rm -rf gpurun_out/pmcv$n
  IPK_SO_OVERRIDE=$so rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d gpurun_out/pmcv$n -o p -- python bench.py --no-cpu-baseline --no-check --steps 2 --warmup 1 --prewarm-ms 0 --data ${DATA:-noise} > /dev/null 2>&1
  python3 - <<PY
import csv, collections
rows=list(csv.DictReader(open('gpurun_out/pmcv$n/p_counter_collection.csv')))
agg=collections.defaultdict(list)
for r in rows:
    if 'fused' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
it = 100e6/256
print('ablate $n', {k: round(sum(v)/len(v)/it,1) for k,v in agg.items()}, '(per wave-iteration of 256 px)')
PY
done
