#!/bin/bash
# Development tool: fused-kernel time over frame sizes for variant libraries.  usage (GPU box): VARIANTS="old new" SIZES="1440x2160 3000x4000" tools/size_sweep.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for sz in ${SIZES:-1440x2160 2000x3000 3000x4000 4000x6000}; do
  h=${sz%x*}; w=${sz#*x}
  for d in ${DATA:-noise photo}; do
    for v in ${VARIANTS}; do
      so="$PWD/imagepipe_amd/csrc/build/ablate/lib$v.so"; [ "$v" = main ] && so=""
      for rep in 1 2; do
        IPK_SO_OVERRIDE=$so python bench.py --no-cpu-baseline --no-check --no-extras --steps ${STEPS:-50} --data $d --height $h --width $w 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$sz $d $v', d['roofline']['kernel_ms'], d['roofline']['kernel_ms_median'], 'ms')"
      done
    done
  done
done
