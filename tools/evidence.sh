#!/bin/bash
# Collects the secondary rocprofv3 evidence DESIGN.md section 5 quotes (run on the GPU box through gpurun):
#   micro-benchmarks of the VALU instruction classes (tools/ubench*.hip), kernel-trace stats of the staged 100 MP
#   pipeline (C3), of config 5 (X-Trans -> 2160x1440) and of the 24 MP configuration.
# Output: gpurun_out/evid_$TAG/ ; copy the *_kernel_stats.csv / *.txt you want judged into profiles/.
TAG=${1:-r02}
OUT=gpurun_out/evid_$TAG
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf $OUT; mkdir -p $OUT
tools/build/ubench2 > $OUT/ubench2.txt 2>&1
tools/build/ubench > $OUT/ubench.txt 2>&1
for cfg in C3 C5_ C5b C2_24MP_rggb_f32; do
  ONLY=$cfg rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$cfg -o s -- python tools/bench_configs.py > $OUT/bench_$cfg.log 2>&1
  f=$(find $OUT/stats_$cfg -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/${cfg}_kernel_stats.csv
done
rocm-smi --showclocks > $OUT/clocks_idle.txt 2>&1
ls $OUT
