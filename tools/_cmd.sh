b() { python bench.py --config c5 --no-cpu-baseline --no-check 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"; }
python -m pytest tests/test_gpu_stages.py tests/test_gpu_fused.py -m gpu -x -q -k "scaled or config5" 2>&1 | tail -2
for rep in 1 2; do b main; IPK_SO_OVERRIDE=$PWD/imagepipe_amd/csrc/build/ablate/libw8.so b w8; done
