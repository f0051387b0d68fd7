#!/bin/bash
# Every fused variant a caller can reach through ipk_raw_to_srgb, timed at 100 MP (and checked against the oracle on the way): curve x linear x source x output.
# usage (GPU box): tools/variants.sh [outfile] [extra bench args]   -- one JSON line per variant
OUT=${1:-gpurun_out/variants.jsonl}; shift
: > $OUT
run() { python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 "$@" 2>>${OUT%.jsonl}.err | tail -n 1 >> $OUT; }
for src in f32 u16; do
  for out in f32 u8 u16; do
    for curve in default none user5; do
      run --src $src --out $out --curve $curve "$@"
    done
  done
  run --src $src --out f32 --curve default --linear "$@"
  run --src $src --out f32 --curve none --exposure 0.5 --linear "$@"
done
python - "$OUT" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    l = l.strip()
    if not l: continue
    d = json.loads(l); c = d["config"]; r = d["roofline"]
    print("%-4s %-4s %-8s lin=%d exp=%.1f  %.4f ms  frac %.3f  %s" % (c["src"], c["out"], c["curve"], c["linear"], c["exposure"], r["kernel_ms"], r["frac"], d.get("parity_check", "")[:40]))
PY
