#!/usr/bin/env python
"""Writes tests/golden/pin/: the INPUTS of the rows of SURVEY.md 8 that no reference test pins (OpGoFloat::run_raw, demosaic::full, scaled_demosaic at a
non-identity scale, rawloader's CFA::new / color_at) as raw little-endian files a Rust owner can feed through the reference itself
(bindings/rust/dump_goldens.rs writes the reference's outputs into tests/golden/ref/; tests/test_golden.py::test_reference_dump_* then compares the oracle
and the HIP path with them).  Inputs only -- nothing here comes from the oracle.

  python tests/golden/make_pin_inputs.py     # rewrites tests/golden/pin/*

cases.txt, one case per line:  name cfa width height black white demosaic_width demosaic_height
  <name>.raw.u16      height x width sensor values, row-major
  <name>.mosaic.f32   the same frame as OpGoFloat's CFA branch must produce it -- stored as the INPUT of the demosaic step, computed here in numpy
                      with the reference's own expression ((v - black) / (white - black)).min(1.0) in f32, so that the demosaic pin does not hang on the
                      gofloat pin
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import util  # noqa: E402

XTRANS = "GGRGGBGGBGGRBRGRBGGGBGGRGGRGGBRBGBRG"
# name -> (cfa, (h, w), (demosaic_width, demosaic_height)); demosaic size == frame size: demosaic::full; smaller: OpDemosaic::run picks
# scaled_demosaic (scale >= minscale) or full + scale_down_opbuf (1 < scale < minscale)
CASES = {
    "p_rggb_full": ("RGGB", (38, 46), (46, 38)),
    "p_bggr_full_odd": ("BGGR", (37, 45), (45, 37)),
    "p_grbg_full": ("GRBG", (36, 44), (44, 36)),
    "p_gbrg_full_odd": ("GBRG", (35, 43), (43, 35)),
    "p_xtrans_full": (XTRANS, (42, 54), (54, 42)),
    "p_rggb_scaled4": ("RGGB", (40, 48), (12, 10)),
    "p_rggb_scaled2p7": ("RGGB", (41, 53), (19, 15)),         # non-integer scale above the Bayer minscale of 2
    "p_rggb_scaled1p5": ("RGGB", (39, 51), (34, 26)),         # 1 < scale < minscale: full, then scale_down_opbuf
    "p_xtrans_scaled4": (XTRANS, (48, 72), (18, 12)),
    "p_xtrans_scaled3p3": (XTRANS, (47, 71), (21, 14)),       # non-integer scale above the X-Trans minscale of 3
    "p_xtrans_scaled2": (XTRANS, (48, 60), (30, 24)),         # below minscale 3: full, then scale_down_opbuf
}


def main():
    out = os.path.join(HERE, "pin")
    os.makedirs(out, exist_ok=True)
    lines = ["# name cfa width height black white demosaic_width demosaic_height   (tests/golden/make_pin_inputs.py)"]
    for i, (name, (cfa, (h, w), (dw, dh))) in enumerate(sorted(CASES.items())):
        raw = util.noise_u16(util.SEED + 300 + i, h, w)
        raw[:2, :6] = [[0, 512, 511, 513, 16383, 16382], [600, 16383, 16383, 16383, 0, 0]]   # below black, at black, saturated runs
        raw.astype("<u2").tofile(os.path.join(out, name + ".raw.u16"))
        mosaic = np.minimum((raw.astype(np.float32) - np.float32(util.BLACK)) / (np.float32(util.WHITE) - np.float32(util.BLACK)), np.float32(1.0)).astype(np.float32)
        mosaic.astype("<f4").tofile(os.path.join(out, name + ".mosaic.f32"))
        lines.append("%s %s %d %d %d %d %d %d" % (name, cfa, w, h, int(util.BLACK), int(util.WHITE), dw, dh))
    open(os.path.join(out, "cases.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
