#!/bin/bash
# Development tool: PMC counters of the config-5 kernel (k_raw_scaled_demosaic_w8m), separate passes.  usage: tools/pmc_c5.sh [tag]
TAG=${1:-c5}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_$TAG; rm -rf $OUT; mkdir -p $OUT
run() { rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $OUT/$1 -o p -- python bench.py --config c5 --no-cpu-baseline --no-check --steps 3 --warmup 1 --prewarm-ms 0 > /dev/null 2>&1; }
run a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT"
run b "SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES"
run c "FETCH_SIZE"
run d "WRITE_SIZE"
python3 - <<PY
import csv, collections, glob
for sub in "abcd":
    fs = glob.glob('$OUT/%s/**/p_counter_collection.csv' % sub, recursive=True)
    if not fs: continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if 'scaled_demosaic' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print({k: round(sum(v)/len(v)) for k, v in agg.items()})
PY
