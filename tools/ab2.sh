#!/bin/bash
# Development tool: same-box A/B of variant libraries with a parity check first.  usage (through gpurun): VARIANTS="a b" DATA="noise photo" REPS=2 tools/ab2.sh [bench args]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in main ${VARIANTS}; do
  if [ $v = main ]; then so=""; else so="$PWD/imagepipe_amd/csrc/build/ablate/lib$v.so"; fi
  for d in noise photo; do
    IPK_SO_OVERRIDE=$so python bench.py --no-cpu-baseline --no-extras --steps 3 --prewarm-ms 0 --data $d "$@" > /tmp/chk.out 2> /tmp/chk.err && echo "check $v $d ok" || { echo "check $v $d FAILED"; tail -3 /tmp/chk.err; }
  done
done
for rep in $(seq 1 ${REPS:-2}); do
for d in ${DATA:-noise photo smooth}; do
  for v in main ${VARIANTS}; do
    if [ $v = main ]; then so=""; else so="$PWD/imagepipe_amd/csrc/build/ablate/lib$v.so"; fi
    IPK_SO_OVERRIDE=$so python bench.py --no-cpu-baseline --no-check --no-extras --steps 20 --data $d "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$d $v', d['roofline']['kernel_ms'], 'ms')"
  done
done
done
