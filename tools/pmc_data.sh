#!/bin/bash
# Development tool: SQ counters of the fused kernel for the two synthetic data kinds (noise / smooth)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for kind in ${KINDS:-noise smooth}; do
  for pass in 1 2; do
    if [ $pass -eq 1 ]; then C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR";
    else C="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; fi
    rm -rf gpurun_out/pmcd_$kind$pass
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/pmcd_$kind$pass -o p -- python bench.py --no-cpu-baseline --no-check --steps 2 --warmup 1 --data $kind ${BENCH_ARGS} > gpurun_out/pmcd_$kind$pass.log 2>&1
    python3 - <<PY
import csv, collections
rows=list(csv.DictReader(open('gpurun_out/pmcd_$kind$pass/p_counter_collection.csv')))
agg=collections.defaultdict(list)
for r in rows:
    if 'fused' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
it = 100e6/256
print('$kind', {k: round(sum(v)/len(v)/it,1) for k,v in agg.items()}, '(per wave-iteration of 256 px)')
PY
  done
done
