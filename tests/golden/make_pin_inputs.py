#!/usr/bin/env python
"""Writes tests/golden/pin/: the INPUTS of the rows of SURVEY.md 8 that no reference test pins (OpGoFloat::run_raw, demosaic::full, scaled_demosaic at a
non-identity scale, rawloader's CFA::new / color_at) as raw little-endian files a Rust owner can feed through the reference itself
(bindings/rust/dump_goldens.rs writes the reference's outputs into tests/golden/ref/; tests/test_golden.py::test_reference_dump_* then compares the oracle
and the HIP path with them).  Inputs only -- nothing here comes from the oracle.

  python tests/golden/make_pin_inputs.py     # rewrites tests/golden/pin/*

cases.txt, one case per line:  name cfa width height black white demosaic_width demosaic_height
  <name>.raw.u16      height x width sensor values, row-major
  pointwise.rgbe.f32 / .params.f32 / .curves.f32   inputs of the point-wise functions (see main)
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


POINTWISE_N = 4096
POINTWISE_MUL = (2.0, 1.0, 1.5, 1.0)                      # normalize_wbs of the synthetic camera's coefficients
POINTWISE_CURVES = ([(0.5, 0.6)], [(0.1, 0.05), (0.3, 0.35), (0.6, 0.7), (0.9, 0.85)])      # SplineFunc::new adds the (0, 0) / (1, 1) ends


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
    # the point-wise functions (color_conversions.rs, curves.rs): the reference's own tests pin them through round trips only, which a table of another
    # length or another interpolation would pass too.  4096 RGBE pixels (E = 0) over [-0.05, 1.2] with the special values in front; the multipliers and
    # the camera matrix the Rust side passes to camera_to_lab; a three-knot and a six-knot curve.
    px = util.uniform_f32(util.SEED + 400, POINTWISE_N * 4, -0.05, 1.2).reshape(POINTWISE_N, 4)
    sp = util.SPECIALS[np.isfinite(util.SPECIALS) & (np.abs(util.SPECIALS) < 1e20)]
    px[: sp.size, 0] = sp; px[: sp.size, 1] = sp[::-1]; px[: sp.size, 2] = np.roll(sp, 7)
    px[:, 3] = 0.0
    px.astype("<f4").tofile(os.path.join(out, "pointwise.rgbe.f32"))
    params = np.concatenate([np.array(POINTWISE_MUL, np.float32), util.cam_matrix().ravel()]).astype("<f4")       # mul[4], cmatrix[3][4]
    params.tofile(os.path.join(out, "pointwise.params.f32"))
    np.array(POINTWISE_CURVES[0] + POINTWISE_CURVES[1], np.float32).astype("<f4").tofile(os.path.join(out, "pointwise.curves.f32"))   # 1 + 4 (x, y) pairs
    print("\n".join(lines))


if __name__ == "__main__":
    main()
