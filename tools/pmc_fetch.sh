#!/bin/bash
# Development tool: HBM fetch / write bytes per launch of the fused kernel for variant builds (FETCH_SIZE doubled per the gfx950 note)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for n in ${ABL:-0}; do
  if [ $n = 0 ]; then so=""; else so="$PWD/imagepipe_amd/csrc/build/ablate/lib$n.so"; fi
  for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmcf$n
  IPK_SO_OVERRIDE=$so rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmcf$n -o p -- python bench.py --no-cpu-baseline --no-check --steps 3 --warmup 1 --prewarm-ms 0 --data ${DATA:-noise} > /dev/null 2>&1
  python3 - <<PY
import csv
rows=[float(r['Counter_Value']) for r in csv.DictReader(open('gpurun_out/pmcf$n/p_counter_collection.csv')) if 'fused' in r['Kernel_Name'] and r['Counter_Name']=='$c']
v=sum(rows)/len(rows)
print('variant $n $c per launch: raw', round(v,1), '-> MB', round(v*(2048 if '$c'=='FETCH_SIZE' else 1024)/1e6,1))
PY
  done
done
