#!/bin/bash
# Development tool: sweep of the schedule's knobs (IPK_DEV_HMAX / HMIN / PCT, -DIPK_DEV_KNOBS build) on one box.
# usage: SETS="32,4,100 32,8,150 ..." DATA="noise photo" [LIB=knobs] [ARGS="--config c2"] tools/gss_sweep.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
so="$PWD/imagepipe_amd/csrc/build/ablate/lib${LIB:-knobs}.so"
for d in ${DATA:-noise photo}; do
for s in ${SETS:-32,4,100}; do
  IFS=, read hmax hmin pct <<< "$s"
  IPK_DEV_HMAX=$hmax IPK_DEV_HMIN=$hmin IPK_DEV_PCT=$pct IPK_SO_OVERRIDE=$so python bench.py --no-cpu-baseline --no-check --no-extras --steps 20 --data $d $ARGS 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$d ${LIB:-knobs} hmax=$hmax hmin=$hmin pct=$pct', d['roofline']['kernel_ms'], d['roofline']['kernel_ms_median'], 'ms')"
done
done
