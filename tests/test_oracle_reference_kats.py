"""Pins the CPU oracle against every known-answer test the reference holds for the hot path
(SURVEY.md section 8c).  Each test restates one reference `#[test]`, cited by file:line; the data are
the reference tests' own inputs and expected outputs, the code is ours.  CPU only.
"""
import numpy as np
import pytest

U8 = np.arange(0, 255, dtype=np.uint8)        # Rust `0..u8::MAX`
U16 = np.arange(0, 65535, dtype=np.uint32)    # Rust `0..u16::MAX`


# ---------------------------------------------------------------------------------------------
# src/color_conversions.rs:337-349
# ---------------------------------------------------------------------------------------------
def test_roundtrip_8bit(orc):
    assert np.array_equal(orc.output8bit(orc.input8bit(U8)), U8)


def test_roundtrip_16bit(orc):
    v = U16.astype(np.uint16)
    assert np.array_equal(orc.output16bit(orc.input16bit(v)), v)


# src/color_conversions.rs:352-383: the image crate widens 8->16 bit as v*257 and keeps 16 bit
def test_roundtrip_8bit_values_image_crate(orc):
    vin = np.concatenate([np.zeros(300, np.uint8), np.arange(256, dtype=np.uint8)])
    widened = vin.astype(np.uint16) * 257
    assert np.array_equal(orc.output8bit(orc.input16bit(widened)), vin)


def test_roundtrip_16bit_values_image_crate(orc):
    vin = np.arange(65536, dtype=np.uint32).astype(np.uint16)
    assert np.array_equal(orc.output16bit(orc.input16bit(vin)), vin)


# src/color_conversions.rs:385-402
def test_roundtrip_8bit_gamma(orc):
    rt = orc.apply_srgb_gamma(orc.expand_srgb_gamma(orc.input8bit(U8)))
    assert np.array_equal(orc.output8bit(rt), U8)


def test_roundtrip_16bit_gamma(orc):
    v = U16.astype(np.uint16)
    rt = orc.apply_srgb_gamma(orc.expand_srgb_gamma(orc.input16bit(v)))
    assert np.array_equal(orc.output16bit(rt), v)


# ---------------------------------------------------------------------------------------------
# src/color_conversions.rs:420-495 -- exhaustive 255^3 8-bit Lab round trips
# ---------------------------------------------------------------------------------------------
def _all_rgb8():
    a = np.arange(0, 255, dtype=np.uint8)
    r, g, b = np.meshgrid(a, a, a, indexing="ij")
    return np.stack([r.ravel(), g.ravel(), b.ravel()], axis=1)


def test_roundtrip_8bit_lab_xyz(orc):
    xyz8 = _all_rgb8()
    out = orc.lab_to_xyz(orc.xyz_to_lab(orc.input8bit(xyz8)))
    assert np.array_equal(orc.output8bit(out), xyz8)


def test_roundtrip_8bit_lab_rgb(orc):
    rgb8 = _all_rgb8()
    pix = np.concatenate([orc.input8bit(rgb8), np.zeros((rgb8.shape[0], 1), np.float32)], axis=1)
    lab = orc.camera_to_lab([1, 1, 1, 1], orc.const_srgb_d65_43(), pix)
    out = orc.lab_to_rgb(orc.const_xyz_d65_33(), lab)
    assert np.array_equal(orc.output8bit(out), rgb8)


def test_roundtrip_8bit_lab_rgb_gamma(orc):
    rgb8 = _all_rgb8()
    pix = np.concatenate([orc.expand_srgb_gamma(orc.input8bit(rgb8)), np.zeros((rgb8.shape[0], 1), np.float32)], axis=1)
    lab = orc.camera_to_lab([1, 1, 1, 1], orc.const_srgb_d65_43(), pix)
    out = orc.apply_srgb_gamma(orc.lab_to_rgb(orc.const_xyz_d65_33(), lab))
    assert np.array_equal(orc.output8bit(out), rgb8)


# ---------------------------------------------------------------------------------------------
# src/color_conversions.rs:497-611 -- 16-bit strided (89/97/101) Lab round trips, 323M triples
# each; chunked over the outer loop.
# ---------------------------------------------------------------------------------------------
R16 = np.arange(0, 65535, 89, dtype=np.uint32).astype(np.uint16)
G16 = np.arange(0, 65535, 97, dtype=np.uint32).astype(np.uint16)
B16 = np.arange(0, 65535, 101, dtype=np.uint32).astype(np.uint16)


def _chunks16(outer_step=1):
    g, b = np.meshgrid(G16, B16, indexing="ij")
    g = g.ravel(); b = b.ravel()
    for r in R16[::outer_step]:
        yield np.stack([np.full_like(g, r), g, b], axis=1)


@pytest.mark.slow
def test_roundtrip_16bit_lab_xyz(orc):
    for xyz16 in _chunks16():
        out = orc.lab_to_xyz(orc.xyz_to_lab(orc.input16bit(xyz16)))
        assert np.array_equal(orc.output16bit(out), xyz16)
        assert np.array_equal(orc.output8bit(out), (xyz16 >> 8).astype(np.uint8))


@pytest.mark.slow
def test_roundtrip_16bit_lab_rgb(orc):
    cm, rm = orc.const_srgb_d65_43(), orc.const_xyz_d65_33()
    for rgb16 in _chunks16():
        pix = np.concatenate([orc.input16bit(rgb16), np.zeros((rgb16.shape[0], 1), np.float32)], axis=1)
        out = orc.lab_to_rgb(rm, orc.camera_to_lab([1, 1, 1, 1], cm, pix))
        assert np.array_equal(orc.output16bit(out), rgb16)
        assert np.array_equal(orc.output8bit(out), (rgb16 >> 8).astype(np.uint8))


@pytest.mark.slow
def test_roundtrip_16bit_lab_rgb_gamma(orc):
    cm, rm = orc.const_srgb_d65_43(), orc.const_xyz_d65_33()
    for rgb16 in _chunks16():
        pix = np.concatenate([orc.expand_srgb_gamma(orc.input16bit(rgb16)), np.zeros((rgb16.shape[0], 1), np.float32)], axis=1)
        lab = orc.camera_to_lab([1, 1, 1, 1], cm, pix)
        lab[:, 0] = orc.apply_srgb_gamma(orc.expand_srgb_gamma(lab[:, 0]))      # `roundtrip_gamma(ll)` :583
        out = orc.apply_srgb_gamma(orc.lab_to_rgb(rm, lab))
        o16 = orc.output16bit(out).astype(np.int32)
        assert np.all(np.abs(o16 - rgb16.astype(np.int32)) <= 1)                # assert_offby(.., 1, 1) :596
        assert np.array_equal(orc.output8bit(out), (rgb16 >> 8).astype(np.uint8))


# ---------------------------------------------------------------------------------------------
# src/ops/curves.rs:165-188
# ---------------------------------------------------------------------------------------------
def test_spline_extremes(orc):
    assert orc.spline_interpolate([], [0.0, 1.0]).tolist() == [0.0, 1.0]


def test_spline_saturates(orc):
    assert orc.spline_interpolate([], [1.5, -0.2]).tolist() == [1.0, 0.0]


def test_spline_high_blackpoint(orc):
    assert orc.spline_interpolate([(0.0, 0.2)], [0.0])[0] == np.float32(0.2)


def test_spline_low_whitepoint(orc):
    assert orc.spline_interpolate([(1.0, 0.8)], [1.0])[0] == np.float32(0.8)


# ---------------------------------------------------------------------------------------------
# src/ops/transform.rs:152-278 -- the nine orientation goldens on the "F" glyph
# ---------------------------------------------------------------------------------------------
def from_rgb_str_vec(rows):
    """src/buffer.rs:82-113"""
    m = {"R": (1, 0, 0), "G": (0, 1, 0), "B": (0, 0, 1), "O": (1, 1, 1), " ": (0, 0, 0)}
    return np.array([[m[c] for c in row] for row in rows], dtype=np.float32)


F = ["        ", " RRRRRR ", " GG     ", " BBBB   ", " GG     ", " GG     ", "        "]
ROTATE_GOLDENS = {
    "Unknown": F,
    "Normal": F,
    "HorizontalFlip": ["        ", " RRRRRR ", "     GG ", "   BBBB ", "     GG ", "     GG ", "        "],
    "VerticalFlip": ["        ", " GG     ", " GG     ", " BBBB   ", " GG     ", " RRRRRR ", "        "],
    "Rotate90": ["       ", " GGBGR ", " GGBGR ", "   B R ", "   B R ", "     R ", "     R ", "       "],
    "Rotate270": ["       ", " R     ", " R     ", " R B   ", " R B   ", " RGBGG ", " RGBGG ", "       "],
    "Rotate180": ["        ", "     GG ", "     GG ", "   BBBB ", "     GG ", " RRRRRR ", "        "],
    "Transpose": ["       ", " RGBGG ", " RGBGG ", " R B   ", " R B   ", " R     ", " R     ", "       "],
    "Transverse": ["       ", "     R ", "     R ", "   B R ", "   B R ", " GGBGR ", " GGBGR ", "       "],
}


def orientation_id(orc, name):
    return {"Normal": orc.OR_NORMAL, "HorizontalFlip": orc.OR_HFLIP, "Rotate180": orc.OR_ROT180,
            "VerticalFlip": orc.OR_VFLIP, "Transpose": orc.OR_TRANSPOSE, "Rotate90": orc.OR_ROT90,
            "Transverse": orc.OR_TRANSVERSE, "Rotate270": orc.OR_ROT270, "Unknown": orc.OR_UNKNOWN}[name]


@pytest.mark.parametrize("name", sorted(ROTATE_GOLDENS))
def test_rotate_goldens(orc, name):
    out = orc.rotate_buffer(from_rgb_str_vec(F), orientation_id(orc, name))
    want = from_rgb_str_vec(ROTATE_GOLDENS[name])
    assert out.shape == want.shape
    assert np.array_equal(out, want)


# ---------------------------------------------------------------------------------------------
# src/scaling.rs:189-203
# ---------------------------------------------------------------------------------------------
def test_scaling_noop(orc):
    data = (np.arange(150 * 150 * 3, dtype=np.uint32) & 0xFFFF).astype(np.uint16).reshape(150, 150, 3)  # `i as u16` wraps
    out = orc.transform_buffer(data, 150, 150, (0, 0), (149, 0), (0, 149), 150, 150, 3)
    assert np.array_equal(out, data)


# ---------------------------------------------------------------------------------------------
# src/ops/rotatecrop.rs:170-270
# ---------------------------------------------------------------------------------------------
def _rc_setup():
    return np.arange(100 * 100 * 3, dtype=np.float32).reshape(100, 100, 3)


RC_CASES = [  # (top, right, bottom, left), (w, h), index of first element in the source
    ((0.1, 0, 0, 0), (100, 90), 100 * 10 * 3),
    ((0, 0, 0.1, 0), (100, 90), 0),
    ((0.1, 0, 0.1, 0), (100, 80), 100 * 10 * 3),
    ((0, 0, 0, 0.1), (90, 100), 10 * 3),
    ((0, 0.1, 0, 0), (90, 100), 0),
    ((0, 0.1, 0, 0.1), (80, 100), 10 * 3),
    ((0.1, 0.1, 0.1, 0.1), (80, 80), 100 * 10 * 3 + 10 * 3),
]


@pytest.mark.parametrize("crops,size,first", RC_CASES)
def test_rotatecrop_crops(orc, crops, size, first):
    buf = _rc_setup()
    out = orc.rotatecrop_run(list(crops) + [0.0], buf)
    assert (out.shape[1], out.shape[0]) == size
    assert out.ravel()[0] == buf.ravel()[first]


def test_rotatecrop_rotate_45(orc):
    out = orc.rotatecrop_run([0, 0, 0, 0, 0.5], _rc_setup())
    assert out.shape[:2] == (141, 141)


def test_rotatecrop_rotate_90(orc):
    out = orc.rotatecrop_run([0, 0, 0, 0, 1.0], _rc_setup())
    assert out.shape[:2] == (100, 100)


# src/ops/rotatecrop.rs:273-312 (loop nests run inside the oracle; 49.6M and 2.7M cases)
@pytest.mark.slow
def test_rotatecrop_roundtrip_transform(orc):
    assert orc.lib().orc_selftest_rotatecrop_roundtrip_transform() == 0


def test_rotatecrop_roundtrip_transform_rotation(orc):
    assert orc.lib().orc_selftest_rotatecrop_roundtrip_rotation() == 0


# ---------------------------------------------------------------------------------------------
# tests/maxsize_test.rs:31-90 -- size negotiation on a blank 128x64 RGB8 source
# (checked on both the slow-path buffer dims and the negotiated sizes)
# ---------------------------------------------------------------------------------------------
def _blank(orc, **kw):
    return orc.make_pipeline(np.zeros((64, 128, 3), np.uint8), **kw)


def _assert_width(orc, desc, w, h):
    assert orc.pipeline_sizes(desc)[1] == (w, h)
    o8 = orc.pipeline_output_8bit(desc)
    assert (o8.shape[1], o8.shape[0]) == (w, h)
    o16 = orc.pipeline_output_16bit(desc)
    assert (o16.shape[1], o16.shape[0]) == (w, h)


def test_maxsize_default_same_size(orc):
    _assert_width(orc, _blank(orc), 128, 64)


def test_maxsize_no_upscaling(orc):
    _assert_width(orc, _blank(orc, maxwidth=128), 128, 64)
    _assert_width(orc, _blank(orc, maxwidth=256), 128, 64)


def test_maxsize_downscale_keeps_ratio(orc):
    _assert_width(orc, _blank(orc, maxwidth=64), 64, 32)


def test_maxsize_rotation(orc):
    _assert_width(orc, _blank(orc, maxwidth=64, rotation=orc.ROT_90), 64, 128)
    _assert_width(orc, _blank(orc, maxwidth=32, rotation=orc.ROT_90), 32, 64)
    _assert_width(orc, _blank(orc, maxwidth=256, rotation=orc.ROT_90), 64, 128)


def test_maxsize_crops(orc):
    _assert_width(orc, _blank(orc, maxwidth=64, crops=(1, 1, 1, 1)), 64, 31)


def test_maxsize_rotatecrop(orc):
    _assert_width(orc, _blank(orc, maxwidth=64, rotatecrop=(0.1, 0.1, 0.1, 0.1, 0.0)), 64, 32)


# ---------------------------------------------------------------------------------------------
# tests/roundtrip_test.rs:4-35 -- all 2^24 RGB8 colours as a 4096x4096 image through the slow path
# ---------------------------------------------------------------------------------------------
def test_roundtrip_8bit_slowpath(orc):
    a = np.arange(256, dtype=np.uint8)
    r, g, b = np.meshgrid(a, a, a, indexing="ij")
    img = np.stack([r.ravel(), g.ravel(), b.ravel()], axis=1).reshape(4096, 4096, 3)
    out = orc.pipeline_output_8bit(orc.make_pipeline(img))
    assert np.array_equal(out, img)


# tests/roundtrip_test.rs:37-84 -- strided 16-bit triples, 4096x4096 blocks, through output_16bit
@pytest.mark.slow
def test_roundtrip_16bit_slowpath(orc):
    r16 = np.arange(0, 65536, 89, dtype=np.uint32).astype(np.uint16)     # `0..=u16::MAX`
    g16 = np.arange(0, 65536, 97, dtype=np.uint32).astype(np.uint16)
    b16 = np.arange(0, 65536, 101, dtype=np.uint32).astype(np.uint16)
    g, b = np.meshgrid(g16, b16, indexing="ij")
    plane = np.stack([g.ravel(), b.ravel()], axis=1)                     # (g,b) pairs for one r
    per_block = 4096 * 4096
    # The reference restarts the inner ranges from `start.1/start.2` after each block (a quirk of its
    # generator); what it asserts is per-pixel identity, which is order independent, so the blocks
    # here simply walk the full strided cube.
    triples = []
    count = 0
    for r in r16:
        t = np.concatenate([np.full((plane.shape[0], 1), r, np.uint16), plane], axis=1)
        triples.append(t); count += t.shape[0]
        if count >= per_block or r == r16[-1]:
            allt = np.concatenate(triples)
            while allt.shape[0] >= per_block or (r == r16[-1] and allt.shape[0] > 0):
                blk = allt[:per_block]; allt = allt[per_block:]
                img = np.zeros((per_block, 3), np.uint16); img[: blk.shape[0]] = blk
                img = img.reshape(4096, 4096, 3)
                out = orc.pipeline_output_16bit(orc.make_pipeline(img))
                assert np.array_equal(out, img)
            triples = [allt] if allt.shape[0] else []
            count = allt.shape[0]
