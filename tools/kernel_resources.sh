#!/bin/bash
# Development tool: registers / LDS / scratch of every kernel in a built object (default: the product's ipk_kernels.o), from the code object's metadata.
# usage: tools/kernel_resources.sh [object] [grep pattern]
O=${1:-$(dirname "$0")/../imagepipe_amd/csrc/build/ipk_kernels.o}
PAT=${2:-.}
T=$(mktemp -d); L=/opt/rocm/lib/llvm/bin
cp "$O" $T/in.o   # (objcopy without an output file rewrites its input and bumps the mtime make looks at)
$L/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin $T/in.o $T/out.o 2>/dev/null
$L/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/k.co --unbundle
$L/llvm-readelf --notes $T/k.co | python3 -c '
import sys, re, subprocess
txt = sys.stdin.read()
rows = []
for blk in txt.split("- .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "?"])[1]
    rows.append((g("name"), g("vgpr_count"), g("sgpr_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size"), g("vgpr_spill_count")))
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
print("%5s %5s %7s %7s %5s  kernel" % ("vgpr", "sgpr", "lds", "scratch", "spill"))
for r, n in zip(rows, names):
    n = re.sub(r"^void ipk::", "", n); n = re.sub(r"\(.*$", "", n)
    print("%5s %5s %7s %7s %5s  %s" % (r[1], r[2], r[3], r[4], r[5], n))
' | grep -E "vgpr|$PAT"
rm -rf $T
