#!/bin/bash
# Development tool: times the fused kernel with parts switched off (see IPK_ABLATE in ipk_kernels.hip).
for n in ${ABL:-0 1 4}; do
  if [ $n -eq 0 ]; then so=""; else so="$PWD/imagepipe_amd/csrc/build/ablate/lib$n.so"; fi
  IPK_SO_OVERRIDE=$so python bench.py --no-cpu-baseline --no-check --steps 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ablate $n', d['roofline']['kernel_ms'], 'ms', d['roofline']['achieved'], 'GB/s')"
done
