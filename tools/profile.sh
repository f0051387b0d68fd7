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
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python bench.py --no-cpu-baseline --no-check --no-extras > $OUT/bench_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- python bench.py --no-cpu-baseline --no-check --no-extras --steps 5 --warmup 1 > $OUT/bench_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- python bench.py --no-cpu-baseline --no-check --no-extras --steps 5 --warmup 1 > $OUT/bench_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/sq -o q -- python bench.py --no-cpu-baseline --no-check --no-extras --steps 5 --warmup 1 > $OUT/bench_sq.log 2>&1
# dynamic VALU instruction counts by class (the VALU-issue model of DESIGN.md section 4), noise and photo-like data, two passes each
for d in noise photo; do
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_SALU --output-format csv -d $OUT/cls1_$d -o c -- python bench.py --no-cpu-baseline --no-check --no-extras --steps 3 --warmup 1 --prewarm-ms 0 --data $d > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT64 SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAVES --output-format csv -d $OUT/cls2_$d -o c -- python bench.py --no-cpu-baseline --no-check --no-extras --steps 3 --warmup 1 --prewarm-ms 0 --data $d > /dev/null 2>&1
  python bench.py --no-cpu-baseline --no-check --no-extras --data $d 2>/dev/null | tail -1 > $OUT/bench_$d.json
done
tools/build/ubench2 > $OUT/ubench2.txt 2>&1
python bench.py > $OUT/bench_plain.json 2> $OUT/bench_plain.err
find $OUT -name "*.csv" | head -20
tail -1 $OUT/bench_plain.json | cut -c1-300
