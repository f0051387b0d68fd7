#!/bin/bash
# Development tool: where a wave of the fused kernel spends its cycles (issue vs waits), two PMC passes.  usage: DATA=noise tools/pmc_stalls.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
D=${DATA:-noise}; OUT=gpurun_out/pmc_stalls_$D; rm -rf $OUT; mkdir -p $OUT
run() { rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $OUT/$1 -o p -- python bench.py --no-cpu-baseline --no-check --no-extras --steps 3 --warmup 1 --prewarm-ms 0 --data $D > /dev/null 2>&1; }
run a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"
run b "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"
python3 - <<PY
import csv, collections, glob
for sub in "ab":
    fs = glob.glob('$OUT/%s/**/p_counter_collection.csv' % sub, recursive=True)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if 'fused' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print('$D', {k: round(sum(v)/len(v)/1e6, 2) for k, v in agg.items()}, '(millions per launch)')
PY
