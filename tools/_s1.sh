cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s1
( time python -m pytest tests -m gpu -x -q ) > gpurun_out/s1/pytest.log 2>&1
tail -5 gpurun_out/s1/pytest.log
python bench.py > gpurun_out/s1/bench_default.json 2> gpurun_out/s1/bench_default.err
tail -c 1500 gpurun_out/s1/bench_default.json
tools/variants.sh gpurun_out/s1/variants.jsonl --no-check
