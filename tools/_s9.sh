cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_fused.py -m gpu -q -x -k "not full_size and not config4 and not 100MP and not 24MP and not quantised" 2>&1 | tail -3
B="python bench.py --no-cpu-baseline --no-check --no-extras --steps 30"
for rep in 1 2 3; do
for d in noise photo; do
for v in base main; do
  so=""; [ $v != main ] && so=$PWD/imagepipe_amd/csrc/build/ablate/lib$v.so
  echo -n "100MP $d $v: "; IPK_SO_OVERRIDE=$so $B --data $d 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'])"
done; done; done
for d in noise photo; do for v in base main; do
  so=""; [ $v != main ] && so=$PWD/imagepipe_amd/csrc/build/ablate/lib$v.so
  echo -n "24MP $d $v: "; IPK_SO_OVERRIDE=$so $B --config c2 --steps 100 --data $d 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'])"
  echo -n "c4 $d $v: "; IPK_SO_OVERRIDE=$so $B --config c4 --steps 5 --data $d 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'])"
done; done
