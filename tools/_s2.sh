cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s2
( time python -m pytest tests -m gpu -x -q ) > gpurun_out/s2/pytest.log 2>&1
tail -5 gpurun_out/s2/pytest.log
tools/variants.sh gpurun_out/s2/variants.jsonl
