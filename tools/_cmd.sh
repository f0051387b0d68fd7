python -m pytest tests/test_gpu_fused.py tests/test_gpu_selftest.py tests/test_golden.py -m gpu -x -q -k "not config4" 2>&1 | tail -2
VARIANTS="nolin" DATA="noise photo smooth flat" tools/ab.sh
