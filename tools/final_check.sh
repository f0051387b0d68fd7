#!/bin/bash
# The round-end check on the GPU box (through gpurun): the whole -m gpu suite, smoke(), then the evidence collection for tools/evidence_summarize.py.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
( time python -m pytest tests -m gpu -q ) > gpurun_out/final/pytest.log 2>&1
tail -4 gpurun_out/final/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/evidence.sh r05 > gpurun_out/final/evidence.log 2>&1
tail -3 gpurun_out/final/evidence.log
