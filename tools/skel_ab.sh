#!/bin/bash
# Development tool: same-box A/B of variant libraries on the fused kernel AND its memory skeleton (ipk_stream_probe = roofline.ceiling_ms), 100 MP and 24 MP
# usage (GPU box): VARIANTS="a b" [SCHEDS="auto lockstep"] tools/skel_ab.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in main ${VARIANTS}; do
  if [ $v = main ]; then so=""; else so="$PWD/imagepipe_amd/csrc/build/ablate/lib$v.so"; fi
  IPK_SO_OVERRIDE=$so python bench.py --no-cpu-baseline --no-extras --steps 3 --prewarm-ms 0 > /tmp/chk.out 2> /tmp/chk.err && echo "check $v ok" || { echo "check $v FAILED"; tail -3 /tmp/chk.err; }
done
for rep in 1 2; do
for cfg in "10000 10000" "6000 4000"; do set -- $cfg
for d in ${DATA:-noise photo}; do
  for v in main ${VARIANTS}; do for sc in ${SCHEDS:-auto}; do
    if [ $v = main ]; then so=""; else so="$PWD/imagepipe_amd/csrc/build/ablate/lib$v.so"; fi
    IPK_SO_OVERRIDE=$so python bench.py --width $1 --height $2 --no-cpu-baseline --no-check --steps 20 --data $d --no-live-traffic --schedule $sc 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1x$2 $d $v $sc kernel %.4f ms  skeleton %.4f ms (%.3f of peak)' % (r['kernel_ms'], r.get('ceiling_ms', 0), r.get('ceiling_frac_of_peak', 0)))"
  done; done
done
done
done
