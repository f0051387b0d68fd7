mkdir -p gpurun_out/r02f
python -m pytest tests/test_gpu_stages.py tests/test_gpu_fused.py -m gpu -x -q -k "scaled or config5 or randomized or transform" > gpurun_out/r02f/tests.log 2>&1; tail -2 gpurun_out/r02f/tests.log
b() { python bench.py --config c5 --no-cpu-baseline --no-check 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"; }
b main; b main
for n in 1024 2048 3072 4096 6144 8192; do IPK_DEV_W8_BLOCKS=$n IPK_SO_OVERRIDE=$PWD/imagepipe_amd/csrc/build/ablate/libknobs.so b blocks$n; done
