#!/bin/bash
# Development tool: per-launch SQ counters of the fused kernel for the shipped library and variant libraries (tools/build_variant.sh), one --pmc pass each.
# usage (through gpurun): VARIANTS="a b" DATA="noise" tools/pmc_ab.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CTRS=${CTRS:-"SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64"}
for d in ${DATA:-noise}; do
for v in main ${VARIANTS}; do
  if [ $v = main ]; then so=""; else so="$PWD/imagepipe_amd/csrc/build/ablate/lib$v.so"; fi
  rm -rf gpurun_out/pmcab_$v
  IPK_SO_OVERRIDE=$so rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d gpurun_out/pmcab_$v -o p -- python bench.py --no-cpu-baseline --no-check --no-extras --steps 3 --warmup 1 --prewarm-ms 0 --data $d > gpurun_out/pmcab_$v.log 2>&1
  python3 - <<PY
import csv, collections, glob
agg=collections.defaultdict(list)
for f in glob.glob('gpurun_out/pmcab_$v/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_fused_bayer' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
print('$d $v', {k: round(sum(x)/len(x)/1e6,2) for k,x in sorted(agg.items())})
PY
done
done
