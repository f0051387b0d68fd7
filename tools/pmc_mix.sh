#!/bin/bash
# Development tool (GPU box): the instruction mix of the fused kernel by issue port (VALU / SALU / branch / SMEM / LDS / VMEM), per wave-row.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_mix; rm -rf $OUT; mkdir -p $OUT
run() { rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $OUT/$1 -o p -- python bench.py --no-cpu-baseline --no-check --no-extras --steps 3 --warmup 1 --prewarm-ms 0 --data ${DATA:-noise} > $OUT/$1.log 2>&1; }
run a "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SENDMSG"
run b "SQ_INSTS_VSKIPPED SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY"
python3 - <<PY
import csv, collections, glob
for sub in "ab":
    fs = glob.glob('$OUT/%s/**/p_counter_collection.csv' % sub, recursive=True)
    if not fs: print(sub, 'no output'); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if 'fused' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print({k: round(sum(v)/len(v)/390625, 1) for k, v in agg.items()}, '(per 256-pixel wave-row)')
PY
tail -3 $OUT/a.log | cut -c1-200
