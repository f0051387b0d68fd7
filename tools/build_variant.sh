#!/bin/bash
# Development tool: builds imagepipe_amd/csrc/build/ablate/lib<name>.so with extra -D flags, for same-box A/B runs
# (IPK_SO_OVERRIDE=<that .so> python bench.py ...).   usage: tools/build_variant.sh name -DIPK_NOQUEUE=1 ...
set -e
name=$1; shift
cd "$(dirname "$0")/../imagepipe_amd/csrc"
mkdir -p build/ablate
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -I../../include"
/opt/rocm/bin/hipcc $F -fno-slp-vectorize "$@" -c ipk_kernels.hip -o build/ablate/k_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/ablate/lib$name.so build/ablate/k_$name.o build/ipk_api.o build/ipk_comm.o -ldl
echo build/ablate/lib$name.so
