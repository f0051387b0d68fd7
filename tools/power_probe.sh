#!/bin/bash
# Development tool (round 3): socket power and shader clock (rocm-smi, polled twice a second) while bench.py launches one kernel back to back for ~10 s.
# usage (GPU box): VARIANTS="main probe4" DATA="noise photo" tools/power_probe.sh   (main = the built library; others = build/ablate/lib<name>.so)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocm-smi --showmaxpower 2>/dev/null | grep -i "max graphics" | sed 's/^.*: //' | sed 's/^/power cap (W): /'
for d in ${DATA:-noise photo}; do
for v in ${VARIANTS:-main}; do
  so="$PWD/imagepipe_amd/csrc/build/ablate/lib$v.so"; [ "$v" = main ] && so=""
  IPK_SO_OVERRIDE=$so python bench.py --no-cpu-baseline --no-check --no-extras --steps ${STEPS:-20000} --data $d > /tmp/pp_$v.json 2>/dev/null &
  pid=$!
  : > /tmp/pp_$v.log
  while kill -0 $pid 2>/dev/null; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -i "Socket Graphics Package Power\|sclk clock level" | sed 's/^.*: //' | tr '\n' ' ' >> /tmp/pp_$v.log; echo >> /tmp/pp_$v.log
    sleep 0.5
  done
  python3 - <<PY
import re, json
rows = []
for l in open('/tmp/pp_$v.log'):
    m = re.search(r'\((\d+)Mhz\).*?([\d.]+)\s*$', l.strip())
    if m: rows.append((float(m.group(2)), int(m.group(1))))
busy = sorted(rows, reverse=True)[:max(1, len(rows) // 5)]           # the fifth of the samples with the highest power: the timed region
d = json.loads(open('/tmp/pp_$v.json').read().strip().splitlines()[-1])
print('$d $v: kernel %.4f ms | %d samples; under load: power %.0f W (max %.0f), sclk %d MHz (median of those samples)' % (
    d['roofline']['kernel_ms'], len(rows), sum(p for p, _ in busy) / len(busy), busy[0][0], sorted(c for _, c in busy)[len(busy) // 2]))
PY
done; done
