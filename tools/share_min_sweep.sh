#!/bin/bash
# Development tool: static schedule + takeovers against the drawn schedule (IPK_DEV_SHARE_MIN, -DIPK_DEV_KNOBS build): frames of several sizes.
# usage: SIZES="4000x6000 6000x8000 10000x10000" MINS="40 1000000" tools/share_min_sweep.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for sz in ${SIZES:-4000x6000 6000x8000 10000x10000}; do
  h=${sz%x*}; w=${sz#*x}
  for d in ${DATA:-noise photo}; do
    for m in ${MINS:-40 1000000}; do
      for rep in 1 2; do
      IPK_DEV_SHARE_MIN=$m IPK_SO_OVERRIDE=$PWD/imagepipe_amd/csrc/build/ablate/lib${LIB:-knobs}.so python bench.py --no-cpu-baseline --no-check --no-extras --steps 20 --data $d --height $h --width $w 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$sz $d share_min=$m', d['roofline']['kernel_ms'], 'ms')"
      done
    done
  done
done
if [ -n "$BATCH_TOO" ]; then
  for m in ${MINS:-64 1000000}; do
    for rep in 1 2; do
    IPK_DEV_SHARE_MIN=$m IPK_SO_OVERRIDE=$PWD/imagepipe_amd/csrc/build/ablate/lib${LIB:-knobs}.so python bench.py --config c4 --no-cpu-baseline --no-check --steps 5 --warmup 1 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c4 64x24MP share_min=$m', d['ms_per_step'], 'ms per batch')"
    done
  done
fi
