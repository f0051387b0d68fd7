cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s4
( time python -m pytest tests/test_gpu_stages.py tests/test_gpu_fused.py tests/test_cache_contract.py -m gpu -q -x -k "not full_size and not config4 and not 100MP and not 24MP" ) > gpurun_out/s4/pytest.log 2>&1
tail -3 gpurun_out/s4/pytest.log
CFGS=C3 tools/staged_stats.sh r04b > gpurun_out/s4/staged.txt 2>&1; grep "avg" gpurun_out/s4/staged.txt | cut -c1-200
