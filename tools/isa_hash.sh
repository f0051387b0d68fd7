#!/bin/bash
# Development tool: one line per kernel of a built object -- "<sha1 of its disassembly> <instructions> <demangled name>" -- to show that a source
# clean-up left the generated code untouched (diff two outputs).  usage: tools/isa_hash.sh [object]
O=${1:-$(dirname "$0")/../imagepipe_amd/csrc/build/ipk_kernels.o}
T=$(mktemp -d); L=/opt/rocm/lib/llvm/bin
cp "$O" $T/in.o; $L/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin $T/in.o $T/out.o 2>/dev/null
$L/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/k.co --unbundle
$L/llvm-objdump -d --no-show-raw-insn --no-leading-addr $T/k.co | python3 -c '
import sys, re, hashlib, subprocess
cur, body, out = None, [], []
def flush():
    if cur is not None:
        txt = "\n".join(re.sub(r"//.*$", "", l).strip() for l in body)
        out.append((cur, hashlib.sha1(txt.encode()).hexdigest()[:12], len(body)))
for line in sys.stdin:
    m = re.match(r"^<(.+)>:$", line.strip())
    if m:
        flush(); cur, body = m.group(1), []
    elif cur is not None and line.strip():
        body.append(line)
flush()
names = subprocess.run(["c++filt"], input="\n".join(o[0] for o in out), capture_output=True, text=True).stdout.split("\n")
for (n, h, c), d in sorted(zip(out, names), key=lambda t: t[1]):
    print(h, c, re.sub(r"\(.*$", "", d.replace("void ipk::", "")))
'
rm -rf $T
