cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s6
( time python -m pytest tests -m gpu -q ) > gpurun_out/s6/pytest.log 2>&1
tail -4 gpurun_out/s6/pytest.log
CFGS=C3 tools/staged_stats.sh r04c > gpurun_out/s6/staged.txt 2>&1; grep "avg" gpurun_out/s6/staged.txt | cut -c1-180
python bench.py > gpurun_out/s6/bench_default.json 2> gpurun_out/s6/bench_default.err; tail -c 600 gpurun_out/s6/bench_default.json
