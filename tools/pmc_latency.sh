#!/bin/bash
# Development tool: how long the fused kernel's LDS and memory instructions are in flight (SQ_INST_LEVEL_* / SQ_INSTS_*), instruction fetch, and the
# FIFO-full counters of the LDS and texture-address paths.  usage: DATA=noise tools/pmc_latency.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
D=${DATA:-noise}; OUT=gpurun_out/pmc_lat_$D; rm -rf $OUT; mkdir -p $OUT
run() { rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $OUT/$1 -o p -- python bench.py --no-cpu-baseline --no-check --no-extras --steps 3 --warmup 1 --prewarm-ms 0 --data $D > /dev/null 2>&1; }
run a "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
run b "SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT"
run c "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY"
python3 - <<PY
import csv, collections, glob
for sub in "abc":
    fs = glob.glob('$OUT/%s/**/p_counter_collection.csv' % sub, recursive=True)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if 'fused' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print('$D', {k: round(sum(v)/len(v)/1e6, 2) for k, v in agg.items()}, '(millions per launch)')
PY
