"""The hot kernels' register budgets, read from the code object inside the built library (no GPU): a change that tips hipcc's allocation into scratch --
it happened twice in round 5, once costing the config-5 kernel 60 % -- fails here instead of in a profile three steps later."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
SO = os.path.join(ROOT, "imagepipe_amd", "libimagepipe_amd.so")

# (demangled-name regex, max VGPRs or None): every match must have no scratch and no spills
NO_SCRATCH = [
    (r"k_fused_bayer<float, true, [012], true, false, false, [12], false>", 128),          # the common-parameter Bayer variants, f32 source (CM = 1, 2)
    (r"k_fused_bayer<unsigned short, false, [012], true, false, false, [012], false>", 128),
    (r"k_fused_bayer_batch<", 128),
    (r"k_fused_bayer<(float, true|unsigned short, false), 4, ", 128),                       # the stream probe
    (r"k_raw_scaled_demosaic_w8m<(float|unsigned short), \du, false>", 72),                 # seven waves per SIMD
    (r"k_raw_scaled_demosaic_w8m<(float|unsigned short), \du, true>", 80),                  # four-colour filters: six
    (r"k_pointwise_chain<true>", 64),                                                       # two resident 1024-thread blocks
    (r"k_pointwise_chain<false>", 128),
    (r"k_pointwise_chain_small", 128),                                                      # the two-pixel form of the full chain (small frames)
    (r"k_gamma|k_fromlab|k_basecurve|k_output8|k_output16", 128),
]


def _kernels():
    if not (os.path.exists(SO) and all(os.path.exists(os.path.join(LLVM, t)) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf"))):
        pytest.skip("library or LLVM tools not present")
    d = tempfile.mkdtemp(prefix="ipk_res_")
    try:
        shutil.copy(SO, os.path.join(d, "in.so"))
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + os.path.join(d, "fat.bin"), os.path.join(d, "in.so"), os.path.join(d, "out.so")],
                       check=True, capture_output=True)
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + os.path.join(d, "fat.bin"),
                        "--output=" + os.path.join(d, "k.co"), "--unbundle"], check=True, capture_output=True)
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(d, "k.co")], check=True, capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(d, ignore_errors=True)
    rows = []
    for blk in notes.split("- .agpr_count:")[1:]:
        g = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "0"])[1]
        rows.append((g("name"), int(g("vgpr_count")), int(g("private_segment_fixed_size")), int(g("vgpr_spill_count"))))
    names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
    return [(n, v, s, sp) for (_, v, s, sp), n in zip(rows, names)]


def test_hot_kernels_have_no_scratch_and_fit_their_occupancy():
    ks = _kernels()
    assert len(ks) > 100
    problems = []
    for pat, max_vgpr in NO_SCRATCH:
        hits = [k for k in ks if re.search(pat, k[0])]
        assert hits, "no kernel matches %r" % pat
        for name, vgpr, scratch, spills in hits:
            short = re.sub(r"\(.*$", "", name.replace("void ipk::", ""))
            if scratch or spills:
                problems.append("%s: %d bytes of scratch, %d spills" % (short, scratch, spills))
            if max_vgpr is not None and vgpr > max_vgpr:
                problems.append("%s: %d VGPRs > %d" % (short, vgpr, max_vgpr))
    assert not problems, "\n".join(problems)
