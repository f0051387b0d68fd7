#!/usr/bin/env python
"""Generates the committed golden vectors under tests/golden/.

The Rust reference cannot run here (no rustc/cargo, absent path dependencies), so the vectors are produced by the CPU
oracle (oracle/imagepipe_oracle.c), which is itself pinned by the reference's known-answer tests
(tests/test_oracle_reference_kats.py).  What the fixtures buy: drift detection -- a different libm (the three 13-bit
tables are built with the host's cbrtf/powf), compiler or flag change shows up as a fixture mismatch on either side
(tests/test_golden.py checks the oracle on CPU and the HIP path on the GPU against the same files).

  python tests/golden/make_golden.py        # rewrites tests/golden/*.npz and lut_fixture.json
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as orc  # noqa: E402
import util  # noqa: E402

XTRANS = "GGRGGBGGBGGRBRGRBGGGBGGRGGRGGBRBGBRG"
CASES = {  # name -> (cfa, (h, w), maxwidth)
    "rggb_full": ("RGGB", (40, 48), 0),
    "rggb_scaled4": ("RGGB", (40, 48), 12),
    "xtrans_full": (XTRANS, (36, 48), 0),
    "xtrans_scaled4": (XTRANS, (48, 72), 18),
}


def stages(cfa, raw, maxwidth):
    """per-stage outputs of the reference's op order on a u16 raw frame"""
    h, w = raw.shape
    desc = orc.make_pipeline(raw, cfa=cfa, blacklevels=[util.BLACK] * 4, whitelevels=[util.WHITE] * 4, wb_coeffs=util.WB,
                             cam_to_xyz_normalized=util.cam_matrix(), maxwidth=maxwidth)
    (dw, dh), (fw, fh) = orc.pipeline_sizes(desc)
    out = {"raw": raw, "sizes": np.array([dw, dh, fw, fh], np.int64)}
    out["gofloat"] = orc.gofloat_cfa(raw, 0, 0, w, h, util.BLACK, util.WHITE)
    branch, out["demosaic"] = orc.demosaic_run(cfa, out["gofloat"], dw, dh)
    out["demosaic_branch"] = np.array([branch], np.int64)
    out["tolab"] = orc.tolab(out["demosaic"], util.WB, util.cam_matrix())
    out["basecurve"] = orc.basecurve(out["tolab"], 0.0, [(0.5, 0.6)])
    out["fromlab"] = orc.fromlab(out["basecurve"])
    out["gamma"] = orc.gamma(out["fromlab"])
    out["out8"] = orc.output8bit(out["gamma"])
    whole = orc.pipeline_run(desc)
    assert np.array_equal(whole.view(np.uint32), out["gamma"].view(np.uint32)), "stage chain != Pipeline::run restatement"
    return out


RASTER_CASES = {  # name -> (bits, (h, w), maxwidth, exposure): raster sources with an edited op (default ops take the integer fast path)
    "rgb8_edited": (8, (24, 40), 0, 0.3),
    "rgb16_scaled3": (16, (30, 48), 16, 0.0),
}


def raster_stages(img, maxwidth, exposure):
    desc = lambda: orc.make_pipeline(img, maxwidth=maxwidth, exposure=exposure, points=[(0.4, 0.5)], use_fastpath=True)
    out = {"img": img, "sizes": np.array(sum(orc.pipeline_sizes(desc()), ()), np.int64)}
    out["gofloat"] = orc.gofloat_other(img, 0, 0, img.shape[1], img.shape[0])
    out["run"] = orc.pipeline_run(desc())
    out["out8"] = orc.pipeline_output_8bit(desc())
    out["out16"] = orc.pipeline_output_16bit(desc())
    return out


def lut_fixture():
    fx = {"note": "TransformLookup tables built with this image's libm (glibc 2.35): sha256 of the little-endian f32 bytes + sampled entries as u32 bits"}
    for which, name in ((orc.LUT_XYZ_LAB, "xyz_lab"), (orc.LUT_SRGB_GAMMA_REVERSE, "srgb_gamma_reverse"), (orc.LUT_SRGB_GAMMA, "srgb_gamma")):
        t = orc.lut_table(which)
        idx = list(range(0, 8193, 128)) + [1, 2, 3, 71, 72, 73, 8190, 8191, 8192]
        fx[name] = {"sha256": hashlib.sha256(t.astype("<f4").tobytes()).hexdigest(), "samples": {str(i): int(t[i:i + 1].view(np.uint32)[0]) for i in idx}}
    return fx


def main():
    for i, (name, (cfa, (h, w), mw)) in enumerate(sorted(CASES.items())):
        raw = util.noise_u16(util.SEED + 100 + i, h, w)
        raw[:2, :6] = [[0, 512, 511, 513, 16383, 16382], [600, 16383, 16383, 16383, 0, 0]]   # below black, at black, saturated runs
        st = stages(cfa, raw, mw)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), cfa=np.array(cfa), maxwidth=np.array([mw], np.int64), **st)
        print(name, {k: v.shape for k, v in st.items()})
    for i, (name, (bits, (h, w), mw, ex)) in enumerate(sorted(RASTER_CASES.items())):
        rng = np.random.default_rng(util.SEED + 200 + i)
        img = rng.integers(0, 256 if bits == 8 else 65536, (h, w, 3)).astype(np.uint8 if bits == 8 else np.uint16)
        img[0, :3] = [[0, 0, 0], [255 if bits == 8 else 65535] * 3, [1, 128, 254]]
        st = raster_stages(img, mw, ex)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), maxwidth=np.array([mw], np.int64), exposure=np.array([ex], np.float32), **st)
        print(name, {k: v.shape for k, v in st.items()})
    json.dump(lut_fixture(), open(os.path.join(HERE, "lut_fixture.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
