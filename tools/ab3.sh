#!/bin/bash
# Development tool: same-box A/B of variant libraries, timing only (no parity leg).  usage (through gpurun): VARIANTS="a b" DATA="noise" REPS=3 tools/ab3.sh [bench args]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in $(seq 1 ${REPS:-2}); do
for d in ${DATA:-noise}; do
  for v in main ${VARIANTS}; do
    if [ $v = main ]; then so=""; else so="$PWD/imagepipe_amd/csrc/build/ablate/lib$v.so"; fi
    IPK_SO_OVERRIDE=$so python bench.py --no-cpu-baseline --no-check --no-extras --steps 20 --data $d "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$d $v', d['roofline']['kernel_ms'], 'ms')"
  done
done
done
