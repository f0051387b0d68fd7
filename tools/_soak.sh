cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
IPK_RANDOM_SEEDS=8000 IPK_RANDOM_SEEDS_DRIVER=12000 python -m pytest tests/test_gpu_fused.py -x -q -k randomized 2>&1 | tail -2 | tee gpurun_out/soak_r04.txt
for i in 1 2 3; do python -m pytest tests/test_gpu_fused.py -x -q -k "queue or takeover or drawn or graph or concurrent or stream_probe or variants_full" 2>&1 | tail -1 | tee -a gpurun_out/soak_r04.txt; done
