#!/bin/bash
# Evidence collection (run on the GPU box through gpurun: `tools/evidence.sh r04`): everything DESIGN.md section 5 and the bench line quote, for
# tools/evidence_summarize.py to turn into tracked files under profiles/.  PMC passes are separate runs with --kernel-trace only.
TAG=${1:-r04}
OUT=gpurun_out/evid_$TAG
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf $OUT; mkdir -p $OUT
B="python bench.py --no-cpu-baseline --no-check --no-extras"
pmc() { local sub=$1; shift; local ctrs=$1; shift; rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OUT/$sub -o p -- "$@" > /dev/null 2>&1; }
# 1. headline: kernel-trace stats of the default command (pre-warmed clock), then HBM traffic in separate passes
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- $B > $OUT/bench_stats.log 2>&1
pmc fetch FETCH_SIZE $B --steps 5 --warmup 1
pmc write WRITE_SIZE $B --steps 5 --warmup 1
# 2. instruction classes and the stall split, noise and photo-like data
for d in noise photo; do
  pmc cls1_$d "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_SALU" $B --steps 3 --warmup 1 --prewarm-ms 0 --data $d
  pmc cls2_$d "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT64 SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAVES" $B --steps 3 --warmup 1 --prewarm-ms 0 --data $d
  pmc stall1_$d "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" $B --steps 3 --warmup 1 --prewarm-ms 0 --data $d
  pmc stall2_$d "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" $B --steps 3 --warmup 1 --prewarm-ms 0 --data $d
  pmc stall3_$d "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INSTS_SMEM" $B --steps 3 --warmup 1 --prewarm-ms 0 --data $d
  $B --data $d 2>/dev/null | tail -n 1 > $OUT/bench_$d.json
done
# 3. the micro-benchmarks the issue model prices with, and their wave-cycles per instruction
mkdir -p tools/build; [ -x tools/build/ubench2 ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench2.hip -o tools/build/ubench2
tools/build/ubench2 > $OUT/ubench2.txt 2>&1
pmc ubench "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU" tools/build/ubench2
# 4. configs[3] (the batch kernel), configs[1] (24 MP), configs[4] (scaled X-Trans): kernel-trace stats at the loaded clock + traffic
for c in c4 c2 c5; do
  extra=""; [ $c = c4 ] && extra="--steps 5 --warmup 1"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$c -o s -- python bench.py --config $c --no-cpu-baseline --no-check --no-box-state $extra > $OUT/bench_$c.log 2>&1
  pmc fetch_$c FETCH_SIZE python bench.py --config $c --no-cpu-baseline --no-check --steps 3 --warmup 1 --prewarm-ms 0
  pmc write_$c WRITE_SIZE python bench.py --config $c --no-cpu-baseline --no-check --steps 3 --warmup 1 --prewarm-ms 0
done
# 4b. every fused variant a caller can reach (curve x linear x source x output), one rocprofv3 kernel-trace run each (tools/evidence_summarize.py -> <TAG>_variants.csv)
mkdir -p $OUT/variants
vi=0
vrun() { vi=$((vi+1)); rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/variants/v$vi -o s -- $B --steps 20 --warmup 3 "$@" > $OUT/variants/v$vi.json 2>/dev/null; echo "$@" > $OUT/variants/v$vi.args; }
for src in f32 u16; do
  for out in f32 u8 u16; do for curve in default none user5; do vrun --src $src --out $out --curve $curve; done; done
  vrun --src $src --out f32 --curve default --linear
  vrun --src $src --out f32 --curve none --exposure 0.5 --linear
done
# 4c. the stream probe (the fused kernel's memory skeleton) under the kernel trace, and the staged kernels of the 100 MP pipeline
IPK_BENCH_NO_LIVE_PMC=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_probe -o s -- python bench.py --no-cpu-baseline --no-check --steps 20 > $OUT/bench_probe.log 2>&1
ONLY=C3 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_staged -o s -- python tools/bench_configs.py > $OUT/bench_staged.log 2>&1
# 4d. the single-process multi-device mode on the hardware there is: two contexts on the one GPU
python bench.py --gpus 2 --single-process --devices 0,0 --no-cpu-baseline > $OUT/bench_sp.json 2> $OUT/bench_sp.err
# 5. the plain default run, exactly as the driver issues it
python bench.py > $OUT/bench_plain.json 2> $OUT/bench_plain.err
tail -n 1 $OUT/bench_plain.json | cut -c1-400
ls $OUT | head -60
