#!/bin/bash
# Development tool (round 3): which part of the fused kernel its waves are PARKED in.  For every variant library (tools/build_variant.sh: the complete
# kernel, and builds with one part switched off -- IPK_ABLATE=1..4, IPK_ABL_STORE, IPK_ABL_LOAD) the kernel time and one PMC pass with the stall split
# (shares of SQ_WAVE_CYCLES).  The difference between two variants is what the part that was switched off costs in issue, in waiting and in time.
# usage: VARIANTS="base abl1 ..." DATA="noise photo" tools/stall_sections.sh      (on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/stall_sections; rm -rf $OUT; mkdir -p $OUT
B="python bench.py --no-cpu-baseline --no-check --no-extras"
for d in ${DATA:-noise photo}; do
for v in ${VARIANTS}; do
  so="$PWD/imagepipe_amd/csrc/build/ablate/lib$v.so"
  ms=$(IPK_SO_OVERRIDE=$so $B --steps 20 --data $d 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'])")
  IPK_SO_OVERRIDE=$so rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_BUSY_CYCLES \
    --output-format csv -d $OUT/$v-$d -o p -- $B --steps 3 --warmup 1 --prewarm-ms 0 --data $d > /dev/null 2>&1
  python3 - <<PY
import csv, collections, glob
fs = glob.glob('$OUT/$v-$d/**/p_counter_collection.csv', recursive=True)
agg = collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    if 'fused' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
m = {k: sum(x) / len(x) / 1e6 for k, x in agg.items()}
wc = m['SQ_WAVE_CYCLES']
print('%-6s %-8s kernel %.4f ms | VALU %.1f M  LDS %.1f M | wave-cycles %.0f M: issuing %.3f  ready-waiting %.3f  parked %.3f (%.0f M) | wait-LDS-issue %.3f' % (
    '$d', '$v', $ms, m['SQ_INSTS_VALU'], m['SQ_INSTS_LDS'], wc, m['SQ_ACTIVE_INST_ANY'] / wc, m['SQ_WAIT_INST_ANY'] / wc, m['SQ_WAIT_ANY'] / wc, m['SQ_WAIT_ANY'], m['SQ_WAIT_INST_LDS'] / wc))
PY
done
done | tee $OUT/summary.txt
