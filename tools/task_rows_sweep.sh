#!/bin/bash
# Development tool (GPU box): task height / small-frame threshold of the persistent row-walking kernel
# (IPK_DEV_TASK_ROWS, IPK_DEV_SHARE_MIN on a build from tools/build_variant.sh knobs -DIPK_DEV_KNOBS=1)
so=$PWD/imagepipe_amd/csrc/build/ablate/libknobs.so
echo "knobs | 100MP noise | 100MP photo | 24MP | c5b | 48MP"
for knobs in ${KNOBS:-"16 40" "24 40" "32 40" "48 40" "16 10" "32 100000"}; do
  set -- $knobs
  line="rows=$1 share_min=$2"
  for cfg in "--width 10000 --height 10000" "--width 10000 --height 10000 --data photo" "--width 6000 --height 4000" "--config c5b" "--width 8000 --height 6000"; do
    v=$(IPK_SO_OVERRIDE=$so IPK_DEV_TASK_ROWS=$1 IPK_DEV_SHARE_MIN=$2 python bench.py $cfg --no-cpu-baseline --no-check --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
    line="$line | $v"
  done
  echo "$line"
done
