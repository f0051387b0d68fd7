cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_stages.py tests/test_gpu_fused.py -m gpu -q -x -k "scaled or config5 or xtrans or maxwidth or randomized" 2>&1 | tail -2
for rep in 1 2 3; do for v in base main; do
  so=""; [ $v != main ] && so=$PWD/imagepipe_amd/csrc/build/ablate/lib$v.so
  echo -n "c5 $v: "; IPK_SO_OVERRIDE=$so python bench.py --config c5 --no-cpu-baseline --no-check --steps 60 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'], d['ms_per_step'])"
done; done
