#!/bin/bash
# Development tool: same-box A/B of variant libraries on the point-wise chain alone (config 5's second kernel at its own size, and at 100 MP)
# usage (GPU box): VARIANTS="a b" tools/chain_ab.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for sz in "2160 1440" "6000 4000" "10000 10000"; do
  for v in main ${VARIANTS}; do
    if [ $v = main ]; then so=""; else so="$PWD/imagepipe_amd/csrc/build/ablate/lib$v.so"; fi
    echo -n "$v $sz: "; IPK_SO_OVERRIDE=$so python tools/stage_probe.py chain $sz 2>/dev/null | tail -1
  done
done
done
