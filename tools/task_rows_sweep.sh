#!/bin/bash
# Development tool: rows per task (IPK_DEV_TASK_ROWS, -DIPK_DEV_KNOBS build) on one box.  usage: LIBS="knobs fair" ROWS="32 33" DATA="noise photo" [ARGS=..] tools/task_rows_sweep.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for d in ${DATA:-noise photo}; do
for lib in ${LIBS:-knobs}; do
for r in ${ROWS:-32}; do
  IPK_DEV_TASK_ROWS=$r IPK_SO_OVERRIDE=$PWD/imagepipe_amd/csrc/build/ablate/lib$lib.so python bench.py --no-cpu-baseline --no-check --no-extras --steps 20 --data $d $ARGS 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$d $lib rows=$r', d['roofline']['kernel_ms'], d['roofline']['kernel_ms_median'], 'ms')"
done; done; done
