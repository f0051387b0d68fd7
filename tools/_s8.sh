cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-check --no-extras --steps 30"
for rep in 1 2 3; do
for d in noise photo; do
for v in main dupmask; do
  so=""; [ $v != main ] && so=$PWD/imagepipe_amd/csrc/build/ablate/lib$v.so
  echo -n "$d $v: "; IPK_SO_OVERRIDE=$so $B --data $d 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'])"
done; done; done
for w in 9984 10000 10240; do echo -n "width $w x 10000 noise: "; $B --width $w --height 10000 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'])"; done
for c in c2; do python bench.py --config c2 --no-cpu-baseline --no-check --steps 100 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('c2', r['kernel_ms'], r.get('ceiling_ms'), r.get('frac_of_ceiling'), r.get('launch_stats'))"; done
