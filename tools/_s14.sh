cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-check --no-extras --steps 30"
for rep in 1 2; do
for d in noise photo smooth; do
for v in main drawn32 drawn48 drawn64; do
  so=""; [ $v != main ] && so=$PWD/imagepipe_amd/csrc/build/ablate/lib$v.so
  echo -n "100MP $d $v: "; IPK_SO_OVERRIDE=$so $B --data $d 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'])"
done; done; done
