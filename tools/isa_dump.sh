#!/bin/bash
# Development tool: disassembles one kernel of a built object.  usage: tools/isa_dump.sh <object> '<demangled kernel name substring>' > kernel.s
O=$1; PAT=$2; L=/opt/rocm/lib/llvm/bin; T=$(mktemp -d)
cp "$O" $T/in.o   # (objcopy without an output file rewrites its input and bumps the mtime make looks at)
$L/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin $T/in.o $T/out.o 2>/dev/null
$L/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/k.co --unbundle
$L/llvm-objdump -d --no-show-raw-insn $T/k.co | c++filt > $T/all.s
awk -v pat="$PAT" 'BEGIN{on=0} /^[0-9a-f]+ </{ on = index($0, pat) > 0 } on{print}' $T/all.s
rm -rf $T
