cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02g
(rocprofv3 --list-avail 2>&1 || rocprofv3 -L 2>&1) > gpurun_out/r02g/avail.txt
grep -o "SQ_INSTS_VALU[A-Z0-9_]*\|SQ_INST_CYCLES[A-Z0-9_]*\|SQ_ACTIVE_INST[A-Z0-9_]*\|SQ_INST_LEVEL[A-Z0-9_]*\|SQ_VALU[A-Z0-9_]*\|SQ_WAIT[A-Z0-9_]*\|SQ_BUSY[A-Z0-9_]*" gpurun_out/r02g/avail.txt | sort -u | tr '\n' ' '
