#!/bin/bash
# Development tool: quick same-box A/B of variant libraries with a parity gate.  usage: VARIANTS="v1 v2" DATA="noise photo" [CHECK=1] tools/abq.sh [bench args]
# For every variant: (CHECK=1) the cube-root proof and a few fused parity tests through IPK_SO_OVERRIDE, then bench.py kernel times, two repetitions.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in ${VARIANTS}; do
  so="$PWD/imagepipe_amd/csrc/build/ablate/lib$v.so"
  if [ "${CHECK:-1}" = 1 ]; then
    IPK_SO_OVERRIDE=$so python -m pytest -x -q tests/test_gpu_selftest.py -k "fast_form or libm_on_every" tests/test_gpu_fused.py -k "fused_f32_vs_oracle_with_specials or fused_u16_vs_oracle or config0 or full_size_strip or strip_geometry" 2>&1 | tail -n 2 | sed "s/^/$v parity: /"
  fi
done
for rep in 1 2; do
for d in ${DATA:-noise photo}; do
  for v in ${VARIANTS}; do
    so="$PWD/imagepipe_amd/csrc/build/ablate/lib$v.so"
    IPK_SO_OVERRIDE=$so python bench.py --no-cpu-baseline --no-check --no-extras --steps 20 --data $d "$@" 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$d $v', d['roofline']['kernel_ms'], d['roofline']['kernel_ms_median'], 'ms')"
  done
done
done
