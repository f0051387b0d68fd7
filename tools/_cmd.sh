python -m pytest tests/test_gpu_fused.py tests/test_golden.py -m gpu -x -q -k "not config4" 2>&1 | tail -2
VARIANTS="notree nolo neither" DATA="noise photo smooth" tools/ab.sh
