#!/bin/bash
# Development tool: shader clock during the fused kernel = GRBM_GUI_ACTIVE / kernel duration
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_clk
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d gpurun_out/pmc_clk -o p -- python bench.py --no-cpu-baseline --no-check --steps 10 --warmup 2 ${BENCH_ARGS} > gpurun_out/pmc_clk.log 2>&1
python3 - <<PY
import csv, collections
rows=list(csv.DictReader(open('gpurun_out/pmc_clk/p_counter_collection.csv')))
tr={r['Dispatch_Id']:(int(r['End_Timestamp'])-int(r['Start_Timestamp'])) for r in csv.DictReader(open('gpurun_out/pmc_clk/p_kernel_trace.csv'))}
for r in rows:
    if 'fused' in r['Kernel_Name']:
        d=tr[r['Dispatch_Id']]
        print(r['Counter_Name'], float(r['Counter_Value']), 'dur_ns', d, 'GHz(if single counter)', float(r['Counter_Value'])/d)
PY
rocm-smi --showclocks 2>/dev/null | head -20
