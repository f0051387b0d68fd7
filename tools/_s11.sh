cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/outlier_probe.py 3000 2>/dev/null
