#!/bin/bash
# Development tool: like abq.sh without the parity gate, N repetitions, mean of kernel_ms per variant and data kind.  usage: VARIANTS="a b" DATA="noise photo" REPS=4 tools/abq3.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in $(seq 1 ${REPS:-4}); do
for d in ${DATA:-noise photo}; do
  for v in ${VARIANTS}; do
    so="$PWD/imagepipe_amd/csrc/build/ablate/lib$v.so"
    IPK_SO_OVERRIDE=$so python bench.py --no-cpu-baseline --no-check --no-extras --steps 20 --data $d $ARGS 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$d $v', d['roofline']['kernel_ms'])"
  done
done
done | python -c "
import sys, collections
a = collections.defaultdict(list)
for l in sys.stdin:
    d, v, ms = l.split(); a[(d, v)].append(float(ms))
for k in sorted(a): print(k[0], k[1], 'mean %.4f  min %.4f  n=%d' % (sum(a[k]) / len(a[k]), min(a[k]), len(a[k])))
"
