#!/bin/bash
# Collects the rocprofv3 evidence bench.py's roofline block refers to (run on the GPU box through gpurun):
#   1. --kernel-trace --stats of the default bench command  -> kernel duration summary
#   2. --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes (TCC slots; never combined with tracing domains other
#      than --kernel-trace) -> HBM traffic per launch
# Everything lands under gpurun_out/$TAG; tools/profile_summarize.py turns it into profiles/$TAG_*.{csv,json}.
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python bench.py --no-cpu-baseline --no-check > $OUT/bench_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- python bench.py --no-cpu-baseline --no-check --steps 5 --warmup 1 > $OUT/bench_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- python bench.py --no-cpu-baseline --no-check --steps 5 --warmup 1 > $OUT/bench_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/sq -o q -- python bench.py --no-cpu-baseline --no-check --steps 5 --warmup 1 > $OUT/bench_sq.log 2>&1
python bench.py > $OUT/bench_plain.json 2> $OUT/bench_plain.err
find $OUT -name "*.csv" | head -20
tail -1 $OUT/bench_plain.json | cut -c1-300
