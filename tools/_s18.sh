cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in main spread8 spread16; do so=""; [ $v != main ] && so=$PWD/imagepipe_amd/csrc/build/ablate/lib$v.so
  echo -n "$v demosaic: "; IPK_SO_OVERRIDE=$so python tools/stage_probe.py demosaic 2>/dev/null | tail -1
  echo -n "$v fused+skeleton: "; IPK_SO_OVERRIDE=$so IPK_BENCH_NO_LIVE_PMC=1 python bench.py --no-cpu-baseline --no-check --width 10000 --height 10000 --steps 30 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read())['roofline']; print(r['kernel_ms'], r['ceiling_ms'], r['ceiling_frac_of_peak'])"
done; done
