#!/bin/bash
# Development tool (round 3): per-wave timeline and section timers (tools/wave_timeline.py) of probe builds -- the complete kernel, the kernel without its
# stores, the load / demosaic / store skeleton.  Build first: tools/build_variant.sh probe -DIPK_DEV_PROBE=1; ... probest1 -DIPK_DEV_PROBE=1 -DIPK_ABL_STORE=1;
# ... probe4 -DIPK_DEV_PROBE=1 -DIPK_ABLATE=4.   usage (GPU box): tools/section_probe.sh > gpurun_out/wave_probe.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in ${VARIANTS:-probe probest1 probe4}; do
  for d in ${DATA:-noise photo}; do
    echo "== $v $d"
    IPK_SO_OVERRIDE=$PWD/imagepipe_amd/csrc/build/ablate/lib$v.so timeout 200 python tools/wave_timeline.py $d 2>&1 | grep -v amdgpu.ids
  done
done
