cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest -x -q tests/test_gpu_fused.py tests/test_gpu_stages.py tests/test_gpu_selftest.py 2>&1 | tail -n 2
for i in 1 2; do
LIB=v3 SETS=0 tools/gss_sweep.sh
for d in noise photo; do python bench.py --no-cpu-baseline --no-check --no-extras --steps 20 --data $d 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$d main', d['roofline']['kernel_ms'], d['roofline']['kernel_ms_median'], 'ms')"; done
done
python bench.py --config c2 --no-cpu-baseline --no-check --steps 20 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c2', d['roofline']['kernel_ms'], d['roofline']['kernel_ms_median'], 'ms')"
python bench.py --config c4 --no-cpu-baseline --no-check --steps 5 --warmup 1 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c4', d['ms_per_step'], 'ms per 64 frames')"
