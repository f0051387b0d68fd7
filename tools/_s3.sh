cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s3
( time python -m pytest tests -m gpu -q ) > gpurun_out/s3/pytest.log 2>&1
tail -5 gpurun_out/s3/pytest.log
CFGS=C3 tools/staged_stats.sh r04a > gpurun_out/s3/staged.txt 2>&1; cat gpurun_out/s3/staged.txt | cut -c1-200
SO=$PWD/imagepipe_amd/csrc/build/ablate/libsweep.so
for even in 0 1; do for g in 0 2; do for b in 1620 1792 2688 3584 5376 7168; do
  r=$(IPK_SO_OVERRIDE=$SO IPK_W8_BLOCKS=$b IPK_W8_GROUP=$g IPK_W8_EVEN=$even python bench.py --config c5 --no-cpu-baseline --no-check --steps 40 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'], d['ms_per_step'])")
  echo "c5 even=$even group=$g blocks=$b : $r" | tee -a gpurun_out/s3/c5_sweep.txt
done; done; done
