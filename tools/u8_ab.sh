#!/bin/bash
# Development tool: same-box A/B of variant libraries on the 8-bit output variants (u16 / f32 source -> u8), 100 MP, noise and photo-like, parity-checked first
# usage (GPU box): VARIANTS="old" [OUTK=u16] tools/u8_ab.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in main ${VARIANTS}; do
  if [ $v = main ]; then so=""; else so="$PWD/imagepipe_amd/csrc/build/ablate/lib$v.so"; fi
  for src in u16 f32; do
    IPK_SO_OVERRIDE=$so python bench.py --no-cpu-baseline --no-extras --steps 3 --prewarm-ms 0 --src $src --out ${OUTK:-u8} > /tmp/chk.out 2> /tmp/chk.err && echo "check $v $src->${OUTK:-u8} ok" || { echo "check $v $src FAILED"; tail -3 /tmp/chk.err; }
  done
done
for rep in 1 2; do
for src in u16 f32; do for d in noise photo; do for curve in default user5; do
  for v in main ${VARIANTS}; do
    if [ $v = main ]; then so=""; else so="$PWD/imagepipe_amd/csrc/build/ablate/lib$v.so"; fi
    IPK_SO_OVERRIDE=$so python bench.py --no-cpu-baseline --no-check --no-extras --steps 20 --data $d --src $src --out ${OUTK:-u8} --curve $curve 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$src->${OUTK:-u8} $d curve=$curve $v', d['roofline']['kernel_ms'], 'ms')"
  done
done; done; done
done
