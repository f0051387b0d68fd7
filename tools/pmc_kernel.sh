#!/bin/bash
# Development tool: SQ / cache counters of one kernel (name substring $KNAME) in `$CMD`
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
KNAME=${KNAME:-raw_scaled}
CMD=${CMD:-"python tools/bench_configs.py"}
i=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "FETCH_SIZE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1)); rm -rf gpurun_out/pmck$i
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/pmck$i -o p -- $CMD > gpurun_out/pmck$i.log 2>&1
  python3 - <<PY
import csv, collections
try:
    rows=list(csv.DictReader(open('gpurun_out/pmck$i/p_counter_collection.csv')))
except Exception as e:
    print('pass $i failed', e); rows=[]
agg=collections.defaultdict(list)
for r in rows:
    if '$KNAME' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
print({k: round(sum(v)/len(v),1) for k,v in agg.items()})
PY
done
