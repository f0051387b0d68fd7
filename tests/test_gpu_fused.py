"""GPU parity of the fused raw->sRGB kernel and of the Pipeline driver (fused and staged) against the CPU oracle's
Pipeline::run restatement, at sizes the oracle finishes in seconds; plus size-independent properties at the
BASELINE.json frame sizes.  Bar: bit-exact (0 ULP; BASELINE.json allows 1 ULP f32)."""
import os

import numpy as np
import pytest

import util
from util import assert_bits_equal

pytestmark = pytest.mark.gpu

CFAS = ["RGGB", "BGGR", "GRBG", "GBRG"]


@pytest.fixture(scope="module")
def ipa():
    import imagepipe_amd
    imagepipe_amd.init(0)
    return imagepipe_amd


def _raw(ipa, raw, cfa="RGGB", is_float=False, **kw):
    import torch
    h, w = raw.shape[:2]
    data = torch.from_numpy(np.ascontiguousarray(raw, np.float32).ravel()).cuda() if is_float else ipa.upload_u16(raw)
    kw.setdefault("blacklevels", [util.BLACK] * 4); kw.setdefault("whitelevels", [util.WHITE] * 4)
    kw.setdefault("wb_coeffs", util.WB); kw.setdefault("cam_to_xyz_normalized", util.cam_matrix())
    return ipa.RawImage(width=w, height=h, data=data, cfa=cfa, is_float=is_float, **kw)


def _oracle_desc(orc, raw, cfa="RGGB", crops=(0, 0, 0, 0), **kw):
    kw.setdefault("blacklevels", [util.BLACK] * 4); kw.setdefault("whitelevels", [util.WHITE] * 4)
    kw.setdefault("wb_coeffs", util.WB); kw.setdefault("cam_to_xyz_normalized", util.cam_matrix())
    return orc.make_pipeline(raw, cfa=orc.cfa_shift(cfa, crops[3], crops[0]) if cfa else "", crops=crops, **kw)


@pytest.mark.parametrize("cfa", CFAS)
@pytest.mark.parametrize("shape", [(10, 10), (11, 13), (64, 256), (97, 257), (130, 1030), (33, 2500)])
def test_fused_u16_vs_oracle(ipa, orc, cfa, shape):
    h, w = shape
    raw = util.noise_u16(util.SEED + h * w, h, w)
    pipe = ipa.Pipeline.new_from_source(_raw(ipa, raw, cfa))
    got = pipe.run()
    assert pipe.last_used_fused and (got.width, got.height, got.colors) == (w, h, 3)
    assert_bits_equal(got.numpy(), orc.pipeline_run(_oracle_desc(orc, raw, cfa)), "fused u16 %s %dx%d" % (cfa, w, h))


@pytest.mark.parametrize("shape", [(12, 16), (65, 259), (100, 1000)])
def test_fused_f32_vs_oracle_with_specials(ipa, orc, shape):
    h, w = shape
    raw = util.noise_u16(util.SEED + 31, h, w).astype(np.float32) + util.uniform_f32(util.SEED + 32, h * w).reshape(h, w)
    raw.ravel()[7: 7 + util.SPECIALS.size] = util.SPECIALS * np.float32(16383.0)       # NaN, inf, negatives, denormals in the mosaic
    pipe = ipa.Pipeline.new_from_source(_raw(ipa, raw, "GRBG", is_float=True))
    got = pipe.run()
    assert pipe.last_used_fused
    assert_bits_equal(got.numpy(), orc.pipeline_run(_oracle_desc(orc, raw, "GRBG")), "fused f32 specials")


@pytest.mark.parametrize("shape", [(65, 259), (100, 1000), (40, 300)])
@pytest.mark.parametrize("points", [[(0.5, 0.6)], [(0.1, 0.07), (0.3, 0.27), (0.5, 0.6), (0.7, 0.82), (0.9, 0.95)], []])
def test_fused_f32_with_specials_quantised_outputs(ipa, orc, shape, points):
    """the 8-bit variants run OpGamma + output8bit as one step lookup, and the lanes the fast form flags (NaN, inf, denormal, huge samples in the mosaic)
    go through the literal form, which reads the gamma table where it lives and quantises itself: both against output_8bit / output_16bit of the oracle,
    full strips (width >= 256: the step-table variants) and a narrow frame (the plain ones), 3-knot / grid / no curve"""
    h, w = shape
    raw = util.noise_u16(util.SEED + 35, h, w).astype(np.float32) + util.uniform_f32(util.SEED + 36, h * w).reshape(h, w)
    raw.ravel()[11: 11 + util.SPECIALS.size] = util.SPECIALS * np.float32(16383.0)
    raw[h // 2, :] = 16383.0 * 3.0                                                     # a blown row: every Lab ratio out of the table
    pipe = ipa.Pipeline.new_from_source(_raw(ipa, raw, "GRBG", is_float=True))
    pipe.ops.basecurve.points = points
    desc = lambda: _oracle_desc(orc, raw, "GRBG", points=points)
    ww, hh, o8 = pipe.output_8bit()
    assert pipe.last_used_fused and (ww, hh) == (w, h)
    got, want = o8.cpu().numpy().reshape(h, w, 3), orc.pipeline_output_8bit(desc())
    assert np.array_equal(got, want), (int((got != want).sum()), np.argwhere(got != want)[:4])
    ww, hh, o16 = pipe.output_16bit()
    assert np.array_equal(o16.cpu().numpy().view(np.uint16).reshape(h, w, 3), orc.pipeline_output_16bit(desc()))


@pytest.mark.parametrize("kind", ["noise", "smooth"])
def test_fused_256x256_config0(ipa, orc, kind):
    """BASELINE.json configs[0]: 256x256 synthetic RGGB -> sRGB; the reference-shaped CPU path is the oracle."""
    raw = (util.noise_u16 if kind == "noise" else util.smooth_u16)(util.SEED, 256, 256)
    pipe = ipa.Pipeline.new_from_source(_raw(ipa, raw))
    assert_bits_equal(pipe.run().numpy(), orc.pipeline_run(_oracle_desc(orc, raw)), "config0 " + kind)
    w, h, o8 = pipe.output_8bit()
    assert np.array_equal(o8.cpu().numpy().reshape(h, w, 3), orc.pipeline_output_8bit(_oracle_desc(orc, raw)))
    w, h, o16 = pipe.output_16bit()
    assert np.array_equal(o16.cpu().numpy().view(np.uint16).reshape(h, w, 3), orc.pipeline_output_16bit(_oracle_desc(orc, raw)))


@pytest.mark.parametrize("crops", [(0, 0, 0, 0), (1, 0, 0, 0), (0, 0, 0, 1), (3, 2, 5, 7), (2, 3, 4, 6)])
@pytest.mark.parametrize("is_float", [False, True])
def test_fused_crops_shift_the_cfa(ipa, orc, crops, is_float):
    """sensor crops: gofloat's offsets + cropped_cfa() phase; odd crops also exercise unaligned u16 rows"""
    h, w = 70, 135
    raw = util.noise_u16(util.SEED + 33, h, w)
    src = raw.astype(np.float32) if is_float else raw
    pipe = ipa.Pipeline.new_from_source(_raw(ipa, src, "RGGB", is_float=is_float, crops=crops))
    got = pipe.run()
    assert pipe.last_used_fused
    assert_bits_equal(got.numpy(), orc.pipeline_run(_oracle_desc(orc, src, "RGGB", crops=crops)), "fused crops %r" % (crops,))


@pytest.mark.parametrize("points,exposure,linear", [([(0.5, 0.6)], 0.0, False), ([], 0.0, False), ([(0.5, 0.6)], 0.7, True),
                                                    ([(0.2, 0.1), (0.4, 0.5), (0.6, 0.55), (0.8, 0.9)], -0.3, False), ([], 1.0, False)])
def test_fused_curve_and_linear_variants(ipa, orc, points, exposure, linear):
    h, w = 48, 300
    raw = util.smooth_u16(util.SEED + 34, h, w)
    pipe = ipa.Pipeline.new_from_source(_raw(ipa, raw))
    pipe.ops.basecurve.points = points; pipe.ops.basecurve.exposure = exposure
    pipe.globals.settings.linear = linear
    got = pipe.run()
    assert pipe.last_used_fused
    want = orc.pipeline_run(_oracle_desc(orc, raw, points=points, exposure=exposure, linear=linear))
    assert_bits_equal(got.numpy(), want, "fused curve variant")


def test_fused_equals_staged_equals_ops(ipa, orc):
    """three routes to the same frame: fused kernel, the C driver's staged kernels, the Python op-by-op loop"""
    h, w = 120, 516
    raw = util.noise_u16(util.SEED + 35, h, w)
    pipe = ipa.Pipeline.new_from_source(_raw(ipa, raw, "BGGR"))
    fused = pipe.run().numpy(); assert pipe.last_used_fused
    pipe.allow_fused = False
    staged = pipe.run().numpy(); assert not pipe.last_used_fused
    ops = pipe.run_ops().numpy()
    want = orc.pipeline_run(_oracle_desc(orc, raw, "BGGR"))
    assert_bits_equal(fused, want, "fused"); assert_bits_equal(staged, want, "staged"); assert_bits_equal(ops, want, "op loop")


def test_fused_row_bands_equal_whole_frame(ipa, orc):
    """multi-GPU band form: each band (with its 1-row halos) reproduces its rows of the whole frame"""
    import torch
    h, w = 90, 260
    raw = util.noise_u16(util.SEED + 36, h, w)
    want = orc.pipeline_run(_oracle_desc(orc, raw, "GBRG"))
    dev = ipa.upload_u16(raw)
    for r0, r1 in [(0, 30), (30, 31), (31, 77), (77, 90)]:
        s0, s1 = max(0, r0 - 1), min(h, r1 + 1)
        band = dev[s0 * w: s1 * w].clone()
        out = ipa.raw_to_srgb(band, width=w, height=h, is_float=False, black0=util.BLACK, white0=util.WHITE, cfa="GBRG",
                              wb_coeffs=util.WB, cam_to_xyz_normalized=util.cam_matrix(), band=(s0, s1 - s0, r0, r1 - r0))
        torch.cuda.synchronize()
        assert_bits_equal(out.cpu().numpy().reshape(r1 - r0, w, 3), want[r0:r1], "band %d..%d" % (r0, r1))


@pytest.mark.parametrize("points,exposure", [([], 0.5), ([], -1.0), ([(0.5, 0.6)], 0.0), ([(0.0, 0.1)], 0.0), ([(1.0, 0.9)], 0.2), ([(0.3, 0.2), (0.6, 0.8)], 0.0),
                                             ([(0.2, 0.9), (0.4, 0.1), (0.6, 0.95), (0.8, 0.05)], 0.0), ([], 0.0)])
@pytest.mark.parametrize("is_float", [False, True])
def test_fused_curve_shapes(ipa, orc, points, exposure, is_float):
    """2-knot (padded to 3 for the compiled-in 3-knot form), 3-knot, many-knot and no-op base curves; samples land exactly on knots"""
    h, w = 24, 512
    raw = util.noise_u16(util.SEED + 66, h, w)
    src = raw.astype(np.float32) if is_float else raw
    pipe = ipa.Pipeline.new_from_source(_raw(ipa, src, "RGGB", is_float=is_float))
    pipe.ops.basecurve.points = list(points); pipe.ops.basecurve.exposure = exposure
    for linear in (False, True):
        pipe.globals.settings.linear = linear
        got = pipe.run(); assert pipe.last_used_fused
        assert_bits_equal(got.numpy(), orc.pipeline_run(_oracle_desc(orc, src, "RGGB", points=points, exposure=exposure, linear=linear)), "curve %r" % (points,))
    ww, hh, o16 = pipe.output_16bit()
    assert np.array_equal(o16.cpu().numpy().view(np.uint16).reshape(hh, ww, 3), orc.pipeline_output_16bit(_oracle_desc(orc, src, "RGGB", points=points, exposure=exposure)))


def test_fused_rejects_four_colour_filters(ipa):
    """the fused kernel covers every three-colour filter; RGBE-style mosaics (the fast point-wise form drops the E term) run staged"""
    import torch
    src = torch.zeros(36 * 36, dtype=torch.float32, device="cuda")
    with pytest.raises(ipa.IpkError):
        ipa.raw_to_srgb(src, width=36, height=36, cfa="RGBE")
    with pytest.raises(ipa.IpkError):
        ipa.raw_to_srgb(src, width=36, height=36, cfa="RGXB")


@pytest.mark.parametrize("cfa", ["RGGB", "GBRG"])
def test_fused_negative_zero_samples_with_zero_black_level(ipa, orc, cfa):
    """f32 mosaics may hold -0.0; with a zero black level the normalised sample is -0.0 too.  The reference's demosaic sums start at
    +0.0 (`0.0 + -0.0 = +0.0`); the fused kernel starts its sums from the first tap (ipk_kernels.hip demosaic_inner_px<.., Z = false>),
    which can only change the SIGN of an exactly-zero channel -- and that sign must not reach any output bit.  Frames made of 0.0 /
    -0.0 / tiny / ordinary samples in every arrangement the 3x3 window can see, at both row parities; the staged demosaic (whose output IS
    the RGBE buffer) keeps the literal sums -- tests/test_gpu_stages.py::test_demosaic_full_vs_oracle holds -0.0 samples at widths >= 256."""
    rng = np.random.default_rng(77)
    h, w = 64, 520
    vals = np.array([0.0, -0.0, -0.0, 0.0, 1e-30, -1e-30, 0.25, 1.0, 0.5, -0.0], np.float32)
    raw = vals[rng.integers(0, vals.size, size=(h, w))]
    raw[10:20, 100:300] = -0.0                                   # whole neighbourhoods of negative zeros
    raw[30:40, 100:300] = 0.0
    raw[44:50, :] = np.where(rng.integers(0, 2, size=(6, w)) == 0, np.float32(-0.0), np.float32(0.0))
    kw = dict(blacklevels=[0.0] * 4, whitelevels=[1.0] * 4)
    pipe = ipa.Pipeline.new_from_source(_raw(ipa, raw, cfa, is_float=True, **kw))
    got = pipe.run()
    assert pipe.last_used_fused
    assert_bits_equal(got.numpy(), orc.pipeline_run(_oracle_desc(orc, raw, cfa, **kw)), "fused, -0.0 samples")
    pipe.allow_fused = False
    assert_bits_equal(pipe.run().numpy(), got.numpy(), "staged == fused with -0.0 samples")
    ww, hh, o8 = pipe.output_8bit()
    assert np.array_equal(o8.cpu().numpy().reshape(hh, ww, 3), orc.pipeline_output_8bit(_oracle_desc(orc, raw, cfa, **kw)))


# ---------------------------------------------------------------------------------------------
# generic-CFA mode of the fused kernel: X-Trans and other three-colour filters
# ---------------------------------------------------------------------------------------------
XT = "GGRGGBGGBGGRBRGRBGGGBGGRGGRGGBRBGBRG"
W16 = "RGGBGRBGGBRGBGGR"                                    # 16 letters: refused (tile shape unverified against rawloader)
W12 = (XT[0:6] + XT[18:24] + XT[6:12] + XT[24:30] + XT[12:18] + XT[30:36]) * 2 + (XT[18:24] + XT[0:6] + XT[24:30] + XT[6:12] + XT[30:36] + XT[12:18]) * 2
W12 = (W12 * 2)[:144]


@pytest.mark.parametrize("cfa", [XT, W12])
@pytest.mark.parametrize("shape", [(10, 10), (13, 37), (48, 257), (61, 530), (30, 1100)])
@pytest.mark.parametrize("is_float", [False, True])
def test_fused_generic_cfa_vs_oracle(ipa, orc, cfa, shape, is_float):
    h, w = shape
    raw = util.noise_u16(util.SEED + 80 + h * w, h, w)
    src = raw.astype(np.float32) if is_float else raw
    crops = (1, 0, 2, 5) if w > 40 else (0, 0, 0, 0)
    pipe = ipa.Pipeline.new_from_source(_raw(ipa, src, cfa, is_float=is_float, crops=crops))
    got = pipe.run()
    assert pipe.last_used_fused
    assert_bits_equal(got.numpy(), orc.pipeline_run(_oracle_desc(orc, src, cfa, crops=crops)), "generic cfa fused")
    pipe.allow_fused = False
    assert_bits_equal(pipe.run().numpy(), got.numpy(), "generic cfa staged == fused")
    pipe.allow_fused = True
    ww, hh, o8 = pipe.output_8bit()
    assert np.array_equal(o8.cpu().numpy().reshape(hh, ww, 3), orc.pipeline_output_8bit(_oracle_desc(orc, src, cfa, crops=crops)))
    ww, hh, o16 = pipe.output_16bit()
    assert np.array_equal(o16.cpu().numpy().view(np.uint16).reshape(hh, ww, 3), orc.pipeline_output_16bit(_oracle_desc(orc, src, cfa, crops=crops)))


L16 = "RGBGRBGGGBGRGRBG"                                    # sixteen letters without symmetry: 2 wide x 8 high and 8 wide x 2 high are different filters


def test_sixteen_letter_cfa_without_a_stated_shape_is_refused(ipa, orc):
    """rawloader's tile shape for a 16-letter pattern (8x2 or 2x8) cannot be verified here: without the caller's statement of it, product and
    oracle refuse the string instead of guessing"""
    raw = util.noise_u16(util.SEED + 81, 32, 300)
    with pytest.raises(ipa.IpkError) as e:                              # OpDemosaic::new's cropped_cfa() is the first to parse it
        ipa.Pipeline.new_from_source(_raw(ipa, raw, W16)).run()
    assert e.value.code == -5 and "16-letter" in str(e.value)           # IPK_ERR_UNSUPPORTED
    plan = ipa.FusedPlan(width=300, height=32, is_float=False, black0=util.BLACK, white0=util.WHITE, cfa=W16)
    with pytest.raises(ipa.IpkError) as e:
        plan.run(ipa.upload_u16(raw), plan.new_output())
    assert e.value.code == -5
    with pytest.raises(Exception):
        orc.pipeline_run(_oracle_desc(orc, raw, W16))


@pytest.mark.parametrize("shape_prefix", ["2x8:", "8x2:", "4x4:"])
@pytest.mark.parametrize("is_float", [False, True])
def test_sixteen_letter_cfa_with_the_callers_shape(ipa, orc, shape_prefix, is_float):
    """The same sixteen letters read 2 wide x 8 high (dcraw's layout), 8 wide x 2 high (what imagepipe's `8 => 2.0` minscale arm describes,
    demosaic.rs:36-37) and 4 x 4, the shape stated by the caller: the fused kernel (generic-CFA mode, pattern 2x8 / 8x2 / 4x4), the staged ops, a
    cropped frame (cropped_cfa() keeps the shape) and a downscaled one (scaled_demosaic; minscale from cfa.width) against the oracle -- whose own
    known answers for the two readings are hand-derived in tests/test_oracle_second_restatement.py -- and the three readings differ from each other."""
    import torch
    cfa = shape_prefix + L16
    H, W = 70, 530
    src = util.noise_u16(util.SEED + 83, H, W)
    raw = src.astype(np.float32) if is_float else src
    pipe = ipa.Pipeline.new_from_source(_raw(ipa, raw, cfa, is_float=is_float))
    got = pipe.run()
    assert pipe.last_used_fused
    want = orc.pipeline_run(_oracle_desc(orc, raw, cfa))
    assert_bits_equal(got.numpy(), want, "fused, " + cfa)
    pipe.allow_fused = False
    assert_bits_equal(pipe.run().numpy(), want, "staged, " + cfa)
    others = [orc.pipeline_run(_oracle_desc(orc, raw, p + L16)) for p in ("2x8:", "8x2:", "4x4:") if p != shape_prefix]
    assert all(not np.array_equal(want, o) for o in others)
    # the shape through the descriptor's fields instead of the string (what a Rust caller copies from its CFA object)
    w_, h_ = (int(v) for v in shape_prefix[:-1].split("x"))
    plan = ipa.FusedPlan(width=W, height=H, is_float=is_float, black0=util.BLACK, white0=util.WHITE, cfa=L16, wb_coeffs=util.WB, cam_to_xyz_normalized=util.cam_matrix())
    plan.params.cfa_width, plan.params.cfa_height = w_, h_
    dev = torch.from_numpy(np.ascontiguousarray(raw).ravel()).cuda() if is_float else ipa.upload_u16(raw)
    out = plan.run(dev, plan.new_output()); torch.cuda.synchronize()
    assert_bits_equal(out.cpu().numpy().reshape(H, W, 3), want, "cfa_width / cfa_height fields, " + cfa)
    plan.params.cfa_width, plan.params.cfa_height = h_ + 1, w_
    with pytest.raises(ipa.IpkError):
        plan.run(dev, plan.new_output())
    # crops shift the pattern (cropped_cfa) and keep its stated shape
    crops = (3, 2, 1, 5)
    pc = ipa.Pipeline.new_from_source(_raw(ipa, raw, cfa, is_float=is_float, crops=crops))
    assert_bits_equal(pc.run().numpy(), orc.pipeline_run(_oracle_desc(orc, raw, cfa, crops=crops)), "cropped, " + cfa)
    # a size limit: OpDemosaic's scaled branch, entered at scale >= minscale(cfa.width) -- 2.0 for widths 2 and 8 and, by the default arm, 4
    ps = ipa.Pipeline.new_from_source(_raw(ipa, raw, cfa, is_float=is_float))
    ps.globals.settings.maxwidth = 130
    assert_bits_equal(ps.run().numpy(), orc.pipeline_run(_oracle_desc(orc, raw, cfa, maxwidth=130)), "scaled, " + cfa)


def test_fused_generic_cfa_specials_take_the_literal_form(ipa, orc):
    """NaN / inf / denormal / huge samples in an X-Trans mosaic: the arithmetic demosaic form is not proven there, the row
    windows that hold them use the literal form; results stay bit-identical"""
    h, w = 40, 300
    raw = util.noise_u16(util.SEED + 83, h, w).astype(np.float32) + util.uniform_f32(util.SEED + 84, h * w).reshape(h, w)
    sp = util.SPECIALS * np.float32(16383.0)
    raw[7, 5: 5 + sp.size] = sp; raw[21, 290: 300] = sp[:10]; raw[0, :3] = [np.nan, np.inf, -np.inf]; raw[39, 297:] = [1e-38, -1e-42, 3e38]
    for black, white in [(util.BLACK, util.WHITE), (0.0, 1.0), (1e-30, 16383.0)]:
        kw = dict(blacklevels=[black] * 4, whitelevels=[white] * 4)
        pipe = ipa.Pipeline.new_from_source(_raw(ipa, raw, XT, is_float=True, **kw))
        got = pipe.run(); assert pipe.last_used_fused
        assert_bits_equal(got.numpy(), orc.pipeline_run(_oracle_desc(orc, raw, XT, **kw)), "xtrans specials %r" % ((black, white),))
    # u16 levels that let a normalised sample leave [2^-60, 2^60]: the u16 kernel switches its sample check on
    raw16 = util.noise_u16(util.SEED + 85, h, w)
    for black, white in [(1e-30, 16383.0), (0.0, 1e-30), (100.0, 100.0 + 2.0 ** -40)]:
        kw = dict(blacklevels=[black] * 4, whitelevels=[white] * 4)
        pipe = ipa.Pipeline.new_from_source(_raw(ipa, raw16, XT, **kw))
        got = pipe.run(); assert pipe.last_used_fused
        assert_bits_equal(got.numpy(), orc.pipeline_run(_oracle_desc(orc, raw16, XT, **kw)), "xtrans u16 levels %r" % ((black, white),))


def test_fused_generic_cfa_row_bands_equal_whole_frame(ipa, orc):
    import torch
    h, w = 90, 300
    raw = util.noise_u16(util.SEED + 86, h, w)
    want = orc.pipeline_run(_oracle_desc(orc, raw, XT))
    dev = ipa.upload_u16(raw)
    for r0, r1 in [(0, 30), (30, 66), (66, 90), (7, 8)]:
        s0, s1 = max(0, r0 - 1), min(h, r1 + 1)
        band = dev[s0 * w: s1 * w].clone()
        out = ipa.raw_to_srgb(band, width=w, height=h, is_float=False, black0=util.BLACK, white0=util.WHITE, cfa=XT,
                              wb_coeffs=util.WB, cam_to_xyz_normalized=util.cam_matrix(), band=(s0, s1 - s0, r0, r1 - r0))
        torch.cuda.synchronize()
        assert_bits_equal(out.cpu().numpy().reshape(r1 - r0, w, 3), want[r0:r1], "xtrans band %d..%d" % (r0, r1))


# ---------------------------------------------------------------------------------------------
# staged driver on the paths the fused kernel does not take
# ---------------------------------------------------------------------------------------------
XTRANS = "GGRGGBGGBGGRBRGRBGGGBGGRGGRGGBRBGBRG"


@pytest.mark.parametrize("case", [dict(cfa=XTRANS, shape=(72, 108)), dict(cfa=XTRANS, shape=(96, 144), maxwidth=36),
                                  dict(cfa="RGGB", shape=(80, 120), maxwidth=40), dict(cfa="RGGB", shape=(80, 120), maxwidth=90),
                                  dict(cfa="RGGB", shape=(80, 120), maxheight=33), dict(cfa="RGBE", shape=(40, 60)),
                                  dict(cfa="RGGB", shape=(60, 90), rotation=1), dict(cfa="RGGB", shape=(60, 90), rotation=3, fliph=True, maxwidth=30),
                                  dict(cfa="RGGB", shape=(60, 90), rotation=2, flipv=True),
                                  dict(cfa="RGGB", shape=(100, 100), rotatecrop=(0.1, 0.05, 0.2, 0.0, 0.0)),
                                  dict(cfa="RGGB", shape=(100, 120), rotatecrop=(0.0, 0.0, 0.0, 0.0, 0.3), maxwidth=64),
                                  dict(cfa="GRBG", shape=(64, 64), crops=(1, 1, 1, 1), maxwidth=20)])
def test_staged_pipeline_vs_oracle(ipa, orc, case):
    case = dict(case)
    h, w = case.pop("shape"); cfa = case.pop("cfa")
    crops = case.pop("crops", (0, 0, 0, 0))
    raw = util.noise_u16(util.SEED + 37, h, w)
    pipe = ipa.Pipeline.new_from_source(_raw(ipa, raw, cfa, crops=crops))
    okw = {}
    if "rotatecrop" in case:
        rc = case.pop("rotatecrop")
        pipe.ops.rotatecrop.crop_top, pipe.ops.rotatecrop.crop_right, pipe.ops.rotatecrop.crop_bottom, pipe.ops.rotatecrop.crop_left, pipe.ops.rotatecrop.rotation = rc
        okw["rotatecrop"] = rc
    for k in ("rotation", "fliph", "flipv"):
        if k in case:
            v = case.pop(k); setattr(pipe.ops.transform, k, v); okw[k] = v
    for k in ("maxwidth", "maxheight"):
        if k in case:
            v = case.pop(k); setattr(pipe.globals.settings, k, v); okw[k] = v
    desc = _oracle_desc(orc, raw, cfa, crops=crops, **okw)
    assert pipe.sizes() == orc.pipeline_sizes(desc)
    assert pipe.negotiate()[0] == orc.pipeline_sizes(desc)[0]           # the negotiation proper (demosaic size); run() decides the final size
    want = orc.pipeline_run(desc)
    got = pipe.run()
    # full-scale three-colour mosaics are fused (an orientation change only adds rotate_buffer behind the fused launch); scaling,
    # rotatecrop and four-colour filters run staged
    fusable = cfa != "RGBE" and not any(k in okw for k in ("maxwidth", "maxheight", "rotatecrop"))
    assert pipe.last_used_fused == fusable
    assert (got.height, got.width) == want.shape[:2]
    assert_bits_equal(got.numpy(), want, "driver")
    pipe.allow_fused = False
    assert_bits_equal(pipe.run().numpy(), want, "staged driver")
    assert_bits_equal(pipe.run_ops().numpy(), want, "op loop")


# ---------------------------------------------------------------------------------------------
# tests/roundtrip_test.rs on the GPU: all 2^24 RGB8 colours through output_8bit, fast path and slow path
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("fast", [True, False])
def test_roundtrip_8bit_gpu(ipa, fast):
    import torch
    a = np.arange(256, dtype=np.uint8)
    r, g, b = np.meshgrid(a, a, a, indexing="ij")
    img = np.stack([r.ravel(), g.ravel(), b.ravel()], axis=1).reshape(4096, 4096, 3)
    pipe = ipa.Pipeline.new_from_source(ipa.OtherImage(4096, 4096, torch.from_numpy(img.ravel()).cuda(), bits=8))
    pipe.globals.settings.use_fastpath = fast
    assert bool(ipa.lib().ipk_pipeline_takes_fastpath(pipe.desc(), ipa.OUT_U8)) == fast
    w, h, o8 = pipe.output_8bit()
    assert (w, h) == (4096, 4096)
    assert np.array_equal(o8.cpu().numpy().reshape(4096, 4096, 3), img)


@pytest.mark.parametrize("fast", [True, False])
def test_roundtrip_16bit_gpu(ipa, fast):
    """strided 16-bit colours (89/97/101) through output_16bit: identity (tests/roundtrip_test.rs:37-84)"""
    r16 = np.arange(0, 65536, 89, dtype=np.uint32).astype(np.uint16)
    g16 = np.arange(0, 65536, 97, dtype=np.uint32).astype(np.uint16)
    b16 = np.arange(0, 65536, 101, dtype=np.uint32).astype(np.uint16)
    g, b = np.meshgrid(g16, b16, indexing="ij")
    plane = np.stack([g.ravel(), b.ravel()], axis=1)
    per = 38                                                       # 38 r-planes of 676*649 triples ~ one 4096x4096 block
    for i in range(0, r16.size, per):
        rs = r16[i: i + per]
        t = np.concatenate([np.concatenate([np.full((plane.shape[0], 1), r, np.uint16), plane], axis=1) for r in rs])
        n = t.shape[0]
        wdt = 4096; hgt = (n + wdt - 1) // wdt
        img = np.zeros((hgt * wdt, 3), np.uint16); img[:n] = t
        img = img.reshape(hgt, wdt, 3)
        pipe = ipa.Pipeline.new_from_source(ipa.OtherImage(wdt, hgt, ipa.upload_u16(img), bits=16))
        pipe.globals.settings.use_fastpath = fast
        w, h, o16 = pipe.output_16bit()
        assert np.array_equal(o16.cpu().numpy().view(np.uint16).reshape(hgt, wdt, 3), img)


@pytest.mark.parametrize("bits", [8, 16])
@pytest.mark.parametrize("maxwidth", [0, 100, 37, 29, 20, 6])      # scales 2.1, 5.7, 7.3 (windows of eight and nine columns side by side), 10.6, 35
def test_raster_fastpath_vs_oracle(ipa, orc, bits, maxwidth):
    """output_8bit / output_16bit of a raster source with default ops (pipeline.rs:381-402, :428-449): no float pipeline, integer
    resampling (scale_down_srgb/16), channel-depth conversion when the depths differ; and the slow path beside it"""
    import torch
    h, w = 150, 211
    rng = np.random.default_rng(8)
    img = rng.integers(0, 256 if bits == 8 else 65536, (h, w, 3)).astype(np.uint8 if bits == 8 else np.uint16)
    data = torch.from_numpy(img.ravel()).cuda() if bits == 8 else ipa.upload_u16(img)
    for fast in (True, False):
        pipe = ipa.Pipeline.new_from_source(ipa.OtherImage(w, h, data, bits=bits))
        pipe.globals.settings.maxwidth = maxwidth; pipe.globals.settings.use_fastpath = fast
        ww, hh, o8 = pipe.output_8bit()
        want = orc.pipeline_output_8bit(orc.make_pipeline(img, maxwidth=maxwidth, use_fastpath=fast))
        assert (hh, ww) == want.shape[:2] and np.array_equal(o8.cpu().numpy().reshape(hh, ww, 3), want), ("8", fast)
        ww, hh, o16 = pipe.output_16bit()
        want = orc.pipeline_output_16bit(orc.make_pipeline(img, maxwidth=maxwidth, use_fastpath=fast))
        assert (hh, ww) == want.shape[:2] and np.array_equal(o16.cpu().numpy().view(np.uint16).reshape(hh, ww, 3), want), ("16", fast)
    # an edited op leaves the fast path (default_ops() false)
    pipe = ipa.Pipeline.new_from_source(ipa.OtherImage(w, h, data, bits=bits))
    pipe.ops.basecurve.exposure = 0.5
    assert ipa.lib().ipk_pipeline_takes_fastpath(pipe.desc(), ipa.OUT_U8) == 0
    ww, hh, o8 = pipe.output_8bit()
    assert np.array_equal(o8.cpu().numpy().reshape(hh, ww, 3), orc.pipeline_output_8bit(orc.make_pipeline(img, exposure=0.5, use_fastpath=True)))


# ---------------------------------------------------------------------------------------------
# full-size frames: size-independent properties (the oracle is too slow to run per test at 100 MP)
# ---------------------------------------------------------------------------------------------
def _tile_frame(tile, reps_y, reps_x):
    return np.tile(tile, (reps_y, reps_x))


@pytest.mark.parametrize("H,W,is_float", [(4000, 6000, False), (4000, 6000, True), (10000, 10000, True)])
def test_config2_config3_full_size_vs_oracle(ipa, orc, H, W, is_float):
    """BASELINE.json configs[1] (24 MP, u16 and f32 mosaic) and configs[2] (100 MP f32, the bench workload) at their real sizes:
    every output sample of the fused kernel against the oracle's unfused pipeline (the oracle takes a few seconds per frame)"""
    import torch
    raw = util.noise_u16(util.SEED + 40 + H, H, W)
    src = raw.astype(np.float32) if is_float else raw
    pipe = ipa.Pipeline.new_from_source(_raw(ipa, src, "RGGB", is_float=is_float))
    got = pipe.run()
    assert pipe.last_used_fused
    want = torch.from_numpy(orc.pipeline_run(_oracle_desc(orc, src, "RGGB")).reshape(-1))
    same = torch.equal(got.data.cpu().view(torch.int32), want.view(torch.int32))
    if not same:                                             # NaN payloads aside, report where
        assert_bits_equal(got.numpy(), want.numpy().reshape(H, W, 3), "full-size frame")


def test_config4_batch_of_64_frames_full_size(ipa, orc):
    """BASELINE.json configs[3] at its real size on one GPU: a batch of 64 independent 24 MP (6000x4000) RGGB f32 frames through ONE
    prepared fused launch descriptor, every frame with its own resident input and output (6 GB in, 18 GB out).  Frames are
    independent pipelines (src/pipeline.rs:246-249).  Five frames spread over the batch are compared sample by sample with the
    oracle; all the others through size-independent properties: a frame computed inside the batch equals the same frame computed
    alone afterwards (no cross-frame state, no buffer aliasing), two slots that hold the SAME mosaic produce the same bits, and
    every frame equals the staged kernels' result (an independent set of kernels, themselves oracle-checked stage by stage)."""
    import torch
    H, W, B = 4000, 6000, 64
    plan = ipa.FusedPlan(width=W, height=H, is_float=True, black0=util.BLACK, white0=util.WHITE, cfa="RGGB", wb_coeffs=util.WB,
                         cam_to_xyz_normalized=util.cam_matrix())
    g = torch.Generator(device="cuda")
    srcs, outs = [], []
    for i in range(B):
        g.manual_seed(util.SEED + 4000 + (i if i != 37 else 5))          # slot 37 repeats frame 5's mosaic
        srcs.append(torch.randint(0, 16384, (H * W,), generator=g, device="cuda", dtype=torch.int32).to(torch.float32))
        outs.append(plan.new_output())
    for s, o in zip(srcs, outs):                                          # the batch: 64 launches back to back on one stream
        plan.run(s, o)
    torch.cuda.synchronize()
    # (1) a handful of frames against the oracle, every sample
    for i in (0, 5, 21, 42, 63):
        raw = srcs[i].cpu().numpy().reshape(H, W)
        want = torch.from_numpy(orc.pipeline_run(_oracle_desc(orc, raw, "RGGB")).reshape(-1))
        if not torch.equal(outs[i].cpu().view(torch.int32), want.view(torch.int32)):
            assert_bits_equal(outs[i].cpu().numpy().reshape(H, W, 3), want.numpy().reshape(H, W, 3), "batch frame %d" % i)
    # (2) independence: recompute alone into a fresh buffer; identical inputs -> identical outputs
    assert torch.equal(outs[37].view(torch.int32), outs[5].view(torch.int32))
    alone = plan.new_output()
    sums = []
    for i in range(B):
        plan.run(srcs[i], alone); torch.cuda.synchronize()
        assert torch.equal(alone.view(torch.int32), outs[i].view(torch.int32)), i
        sums.append(int(outs[i].view(torch.int32).to(torch.int64).sum().item()))
    assert len(set(sums)) == B - 1                                        # 63 distinct frames + the deliberate repeat
    # (3) every frame against the staged path (gofloat, demosaic, tolab, basecurve, fromlab, gamma as separate kernels)
    for i in range(0, B, 7):
        pipe = ipa.Pipeline.new_from_source(ipa.RawImage(width=W, height=H, data=srcs[i], cfa="RGGB", is_float=True, blacklevels=[util.BLACK] * 4,
                                                         whitelevels=[util.WHITE] * 4, wb_coeffs=util.WB, cam_to_xyz_normalized=util.cam_matrix()))
        pipe.allow_fused = False
        st = pipe.run()
        assert not pipe.last_used_fused
        assert torch.equal(st.data.view(torch.int32), outs[i].view(torch.int32)), i
        del st, pipe


def test_config4_batch_launch_full_size_vs_oracle(ipa, orc):
    """BASELINE.json configs[3] through the path bench.py TIMES: ipk_raw_to_srgb_batch, 64 x 6000x4000 RGGB f32 frames as ONE persistent launch
    (k_fused_bayer_batch: the queue holds the tasks of all 64 frames).  Six frames spread over the batch -- the first, the last, one next to a
    frame boundary of the task list -- against the oracle, every sample; every frame against the same frame through a launch of its own; and
    the same once more with the 8-bit output (the first and the last frame against the oracle's output_8bit)."""
    import torch
    H, W, B = 4000, 6000, 64
    kw = dict(width=W, height=H, is_float=True, black0=util.BLACK, white0=util.WHITE, cfa="RGGB", wb_coeffs=util.WB, cam_to_xyz_normalized=util.cam_matrix())
    plan = ipa.FusedPlan(**kw)
    g = torch.Generator(device="cuda")
    srcs = []
    for i in range(B):
        g.manual_seed(util.SEED + 7000 + i)
        srcs.append(torch.randint(0, 16384, (H * W,), generator=g, device="cuda", dtype=torch.int32).to(torch.float32))
    outs = [plan.new_output() for _ in range(B)]
    batch = ipa.FusedBatchPlan(plan, srcs, outs)
    batch.run(); torch.cuda.synchronize()
    for i in (0, 1, 17, 32, 62, 63):
        raw = srcs[i].cpu().numpy().reshape(H, W)
        want = torch.from_numpy(orc.pipeline_run(_oracle_desc(orc, raw, "RGGB")).reshape(-1))
        if not torch.equal(outs[i].cpu().view(torch.int32), want.view(torch.int32)):
            assert_bits_equal(outs[i].cpu().numpy().reshape(H, W, 3), want.numpy().reshape(H, W, 3), "batch-launch frame %d" % i)
    alone = plan.new_output()
    for i in range(B):
        plan.run(srcs[i], alone); torch.cuda.synchronize()
        assert torch.equal(alone.view(torch.int32), outs[i].view(torch.int32)), i
    del outs, alone, batch
    torch.cuda.empty_cache()
    plan8 = ipa.FusedPlan(out_type=ipa.OUT_U8, **kw)
    outs8 = [plan8.new_output() for _ in range(B)]
    ipa.FusedBatchPlan(plan8, srcs, outs8).run(); torch.cuda.synchronize()
    one8 = plan8.new_output()
    for i in range(B):
        plan8.run(srcs[i], one8); torch.cuda.synchronize()
        assert torch.equal(one8, outs8[i]), i
    for i in (0, 63):
        raw = srcs[i].cpu().numpy().reshape(H, W)
        assert np.array_equal(outs8[i].cpu().numpy().reshape(H, W, 3), orc.pipeline_output_8bit(_oracle_desc(orc, raw, "RGGB"))), i


def test_config5_full_size_f32_mosaic_vs_oracle(ipa, orc):
    """BASELINE.json configs[4] as BASELINE.md words it -- an f32 mosaic (RawImageData::Float) -- at its real size: 8640x5760 X-Trans, maxwidth 2160
    -> 2160x1440 through k_raw_scaled_demosaic_w8m<float> and the point-wise chain, the whole output against the oracle.  The samples are 14-bit
    values plus a fraction, so that the float path is not fed integers only."""
    H, W = 5760, 8640
    xt = "GGRGGBGGBGGRBRGRBGGGBGGRGGRGGBRBGBRG"
    raw = (util.noise_u16(util.SEED + 41, H, W).astype(np.float32) + util.uniform_f32(util.SEED + 42, H * W).reshape(H, W)).astype(np.float32)
    pipe = ipa.Pipeline.new_from_source(_raw(ipa, raw, xt, is_float=True))
    pipe.globals.settings.maxwidth = 2160
    got = pipe.run()
    assert (got.width, got.height) == (2160, 1440) and not pipe.last_used_fused
    want = orc.pipeline_run(_oracle_desc(orc, raw, xt, maxwidth=2160))
    assert_bits_equal(got.numpy(), want, "config 5 full size, f32 mosaic")


def test_config5_full_size_vs_oracle(ipa, orc):
    """BASELINE.json configs[4] at its real size: 8640x5760 X-Trans mosaic, maxwidth 2160 -> 2160x1440 through the one-pass
    gofloat+scaled-demosaic kernel and the point-wise chain; the whole output against the oracle (the oracle needs a few seconds)"""
    H, W = 5760, 8640
    xt = "GGRGGBGGBGGRBRGRBGGGBGGRGGRGGBRBGBRG"
    raw = util.noise_u16(util.SEED + 39, H, W)
    pipe = ipa.Pipeline.new_from_source(_raw(ipa, raw, xt))
    pipe.globals.settings.maxwidth = 2160
    got = pipe.run()
    assert (got.width, got.height) == (2160, 1440) and not pipe.last_used_fused
    want = orc.pipeline_run(_oracle_desc(orc, raw, xt, maxwidth=2160))
    assert_bits_equal(got.numpy(), want, "config 5 full size")
    w, h, o8 = pipe.output_8bit()
    assert np.array_equal(o8.cpu().numpy().reshape(h, w, 3), orc.pipeline_output_8bit(_oracle_desc(orc, raw, xt, maxwidth=2160)))
    # (both quantised previews leave through ipk_pointwise_chain_out: tolab..gamma + the quantise loop in one pass, the 8-bit one on the step table)
    w, h, o16 = pipe.output_16bit()
    assert np.array_equal(o16.cpu().numpy().view(np.uint16).reshape(h, w, 3), orc.pipeline_output_16bit(_oracle_desc(orc, raw, xt, maxwidth=2160)))


@pytest.mark.parametrize("H,W,cfa,th,tw", [(4000, 6000, "RGGB", 50, 40), (10000, 10000, "RGGB", 50, 40),
                                          (5760, 8640, "GGRGGBGGBGGRBRGRBGGGBGGRGGRGGBRBGBRG", 60, 48),
                                          (20000, 20000, "RGGB", 50, 40)])       # 400 MP: 4.8 GB of output, byte offsets past 2^32
def test_full_size_frame_is_periodic_like_its_input(ipa, orc, H, W, cfa, th, tw):
    """A frame built by tiling a tile whose sides are multiples of the CFA period is periodic, so the output interior must
    repeat the oracle's output of a 3x3 tiling of that tile (every interior pixel sees the same 3x3 neighbourhood): the
    BASELINE.json frame sizes, 24 MP and 100 MP Bayer and the 50 MP X-Trans frame (generic-CFA mode)."""
    import torch
    tile = util.noise_u16(util.SEED + 38, th, tw)
    frame = _tile_frame(tile, H // th, W // tw)
    assert frame.shape == (H, W)
    pipe = ipa.Pipeline.new_from_source(_raw(ipa, frame, cfa))
    out = pipe.run()
    assert pipe.last_used_fused
    got = out.data.view(H, W, 3)
    small = orc.pipeline_run(_oracle_desc(orc, _tile_frame(tile, 3, 3), cfa))
    centre = torch.from_numpy(small[th:2 * th, tw:2 * tw].copy()).cuda()
    # interior tiles: compare a spread of them bit-for-bit on the device
    ys = sorted(set([1, 2, H // th // 2, H // th - 2]))
    xs = sorted(set([1, 2, W // tw // 2, W // tw - 2]))
    for ty in ys:
        for tx in xs:
            blk = got[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw]
            assert torch.equal(blk.view(torch.int32), centre.view(torch.int32)), (ty, tx)
    # every interior tile equals the first interior tile (whole-frame periodicity), checked in one shot
    inner = got[th:H - th, tw:W - tw].reshape(H // th - 2, th, W // tw - 2, tw, 3)
    ref = inner[0:1, :, 0:1]
    assert bool((inner.view(torch.int32) == ref.view(torch.int32)).all())
    # frame edges: the first/last tile rows and columns against the oracle's 3x3 result edges
    edge = small
    assert torch.equal(got[:th, :tw].view(torch.int32), torch.from_numpy(edge[:th, :tw].copy()).cuda().view(torch.int32))
    assert torch.equal(got[H - th:, W - tw:].view(torch.int32), torch.from_numpy(edge[2 * th:, 2 * tw:].copy()).cuda().view(torch.int32))
    assert torch.equal(got[:th, W - tw:].view(torch.int32), torch.from_numpy(edge[:th, 2 * tw:].copy()).cuda().view(torch.int32))
    assert torch.equal(got[H - th:, :tw].view(torch.int32), torch.from_numpy(edge[2 * th:, :tw].copy()).cuda().view(torch.int32))


def test_full_size_strip_matches_oracle(ipa, orc):
    """24 MP noise frame: three horizontal strips (top edge, middle, bottom edge) bit-identical to the oracle run on
    just those rows plus their halo rows"""
    import torch
    H, W = 4000, 6000
    frame = util.noise_u16(util.SEED + 39, H, W)
    pipe = ipa.Pipeline.new_from_source(_raw(ipa, frame))
    got = pipe.run().data.view(H, W, 3)
    for r0, r1 in [(0, 24), (1988, 2012), (3976, 4000)]:
        s0, s1 = max(0, r0 - 2), min(H, r1 + 2)              # even offsets keep the CFA phase
        sub = orc.pipeline_run(_oracle_desc(orc, frame[s0:s1]))
        assert_bits_equal(got[r0:r1].cpu().numpy(), sub[r0 - s0: r1 - s0], "strip %d..%d" % (r0, r1))


# ---------------------------------------------------------------------------------------------
# the C++ mirror of Pipeline / ImageOp / OpBuffer (include/imagepipe_amd.hpp), built by __graft_entry__.build()
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cfa,shape,maxwidth,rotation", [("RGGB", (64, 256), 0, 0), ("GRBG", (50, 70), 0, 0), ("RGGB", (64, 96), 24, 0),
                                                       ("GGRGGBGGBGGRBRGRBGGGBGGRGGRGGBRBGBRG", (48, 72), 0, 1)])
def test_cpp_mirror_pipeline(orc, tmp_path, cfa, shape, maxwidth, rotation):
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp", "build", "mirror_test")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.dirname(os.path.dirname(exe))])
    h, w = shape
    raw = util.noise_u16(util.SEED + 60, h, w)
    raw.tofile(tmp_path / "in.u16")
    out = subprocess.run([exe, str(tmp_path / "in.u16"), str(w), str(h), cfa, str(maxwidth), str(rotation), str(tmp_path / "out.f32")],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    ow, oh, fused, n8 = [int(v) for v in out.stdout.split()]
    want = orc.pipeline_run(_oracle_desc(orc, raw, cfa, maxwidth=maxwidth, rotation=rotation))
    assert (oh, ow) == want.shape[:2] and n8 == ow * oh * 3
    assert bool(fused) == (maxwidth == 0)                   # every three-colour filter at full scale, any orientation
    assert_bits_equal(np.fromfile(tmp_path / "out.f32", np.float32).reshape(oh, ow, 3), want, "C++ mirror")


# ---------------------------------------------------------------------------------------------
# strip geometry of the FULL kernel (256-pixel strips, the last one shifted left to end at the last column: any width,
# odd strip starts, element-aligned vector loads/stores) x output types x source types
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(10, 256), (12, 257), (12, 259), (12, 260), (23, 511), (23, 512), (13, 769), (17, 1023), (17, 1028), (40, 2052),
                                   (40, 2053), (11, 4096 + 8), (11, 4096 + 7)])
@pytest.mark.parametrize("is_float", [False, True])
def test_fused_full_kernel_strip_geometry_all_outputs(ipa, orc, shape, is_float):
    h, w = shape
    raw = util.noise_u16(util.SEED + 61 + w, h, w)
    src = raw.astype(np.float32) if is_float else raw
    pipe = ipa.Pipeline.new_from_source(_raw(ipa, src, "GBRG", is_float=is_float))
    d = _oracle_desc(orc, src, "GBRG")
    got = pipe.run(); assert pipe.last_used_fused
    assert_bits_equal(got.numpy(), orc.pipeline_run(d), "f32 out %dx%d" % (w, h))
    ww, hh, o8 = pipe.output_8bit()
    assert np.array_equal(o8.cpu().numpy().reshape(hh, ww, 3), orc.pipeline_output_8bit(_oracle_desc(orc, src, "GBRG")))
    ww, hh, o16 = pipe.output_16bit()
    assert np.array_equal(o16.cpu().numpy().view(np.uint16).reshape(hh, ww, 3), orc.pipeline_output_16bit(_oracle_desc(orc, src, "GBRG")))


@pytest.mark.parametrize("crops", [(1, 2, 3, 5), (0, 1, 0, 1), (3, 0, 2, 7)])
@pytest.mark.parametrize("is_float", [False, True])
def test_fused_full_kernel_odd_crops_and_pitch(ipa, orc, crops, is_float):
    """sensor pitch, crop offset and cropped width all odd: the strip loads start at odd element offsets of the source"""
    h, w = 31, 777
    raw = util.noise_u16(util.SEED + 65, h, w)
    src = raw.astype(np.float32) if is_float else raw
    pipe = ipa.Pipeline.new_from_source(_raw(ipa, src, "GRBG", is_float=is_float, crops=crops))
    got = pipe.run(); assert pipe.last_used_fused
    assert_bits_equal(got.numpy(), orc.pipeline_run(_oracle_desc(orc, src, "GRBG", crops=crops)), "odd crops %r" % (crops,))
    ww, hh, o8 = pipe.output_8bit()
    assert np.array_equal(o8.cpu().numpy().reshape(hh, ww, 3), orc.pipeline_output_8bit(_oracle_desc(orc, src, "GRBG", crops=crops)))
    ww, hh, o16 = pipe.output_16bit()
    assert np.array_equal(o16.cpu().numpy().view(np.uint16).reshape(hh, ww, 3), orc.pipeline_output_16bit(_oracle_desc(orc, src, "GRBG", crops=crops)))


def test_fused_extreme_levels_fall_back_to_true_division(ipa, orc):
    """white-black ranges the host cannot validate for the multiply/fma division (tiny, huge, negative) and absurd
    parameters must still be bit-exact (exact_norm / literal point-wise path)"""
    h, w = 24, 300
    raw = util.noise_u16(util.SEED + 62, h, w).astype(np.float32)
    for black, white in [(0.0, 1e-30), (0.0, 3e38), (16383.0, 512.0), (512.0, 512.0), (1e-30, 16383.0)]:
        kw = dict(blacklevels=[black] * 4, whitelevels=[white] * 4)
        pipe = ipa.Pipeline.new_from_source(_raw(ipa, raw, "RGGB", is_float=True, **kw))
        assert_bits_equal(pipe.run().numpy(), orc.pipeline_run(_oracle_desc(orc, raw, "RGGB", **kw)), "levels %r" % ((black, white),))
    for wb, scale in [((2.0, 0.0, 1.5, 1.0), 1.0), ((1e30, 1.0, 1e-30, 1.0), 1.0), ((2.0, 1.0, 1.5, np.nan), 1e25), ((-2.0, 1.0, -1.5, 1.0), -3.0)]:
        cm = (util.cam_matrix() * np.float32(scale)).astype(np.float32)
        kw = dict(wb_coeffs=wb, cam_to_xyz_normalized=cm)
        # the op's public field, set after OpToLab::new (which would replace abnormal as-shot coefficients by neutralwb(),
        # colorspaces.rs:33-39); run() normalises whatever it finds there (:100)
        pipe = ipa.Pipeline.new_from_source(_raw(ipa, raw, "RGGB", is_float=True, cam_to_xyz_normalized=cm))
        pipe.ops.tolab.wb_coeffs = list(wb)
        assert_bits_equal(pipe.run().numpy(), orc.pipeline_run(_oracle_desc(orc, raw, "RGGB", **kw)), "params %r" % ((wb, scale),))


def test_fused_denormal_and_huge_samples_take_the_guarded_path(ipa, orc):
    """f32 mosaics full of denormals / 1e-35-class values / huge values: every dividend guard fires somewhere"""
    h, w = 16, 512
    rng = np.random.default_rng(5)
    expo = rng.integers(-149, 127, size=(h, w)).astype(np.float64)
    raw = (np.ldexp(rng.uniform(1, 2, size=(h, w)), expo.astype(np.int64)) * rng.choice([-1.0, 1.0], size=(h, w))).astype(np.float32)
    for black, white in [(0.0, 1.0), (0.0, 16383.0)]:
        kw = dict(blacklevels=[black] * 4, whitelevels=[white] * 4)
        pipe = ipa.Pipeline.new_from_source(_raw(ipa, raw, "BGGR", is_float=True, **kw))
        assert_bits_equal(pipe.run().numpy(), orc.pipeline_run(_oracle_desc(orc, raw, "BGGR", **kw)), "wild samples")


def test_per_stage_timing_uses_the_reference_op_names(ipa):
    """do_timing! (pipeline.rs:68-80): ipk_timing_begin / ipk_timing_end bracket every stage of the driver with hipEvents; stage
    names are the reference's op names, in its op order"""
    raw = util.noise_u16(util.SEED + 700, 600, 800)
    pipe = ipa.Pipeline.new_from_source(_raw(ipa, raw, "RGGB"))
    out, st = pipe.run_timed()
    assert pipe.last_used_fused and len(st) == 1 and st[0][0].startswith("fused gofloat+demosaic+to_lab") and st[0][1] > 0
    pipe.allow_fused = False
    out2, st = pipe.run_timed()
    assert [n for n, _ in st] == ["gofloat", "demosaic", "rotatecrop", "to_lab", "basecurve", "from_lab", "gamma"]
    assert all(ms >= 0 for _, ms in st) and sum(ms for _, ms in st) > 0
    import torch
    assert torch.equal(out.view(torch.int32), out2.view(torch.int32))
    pipe.ops.transform.rotation = 1
    _, st = pipe.run_timed(ipa.OUT_U8)
    assert [n for n, _ in st][-1] == "quantise+transform" and "gamma" in [n for n, _ in st]
    pipe.allow_fused = True; pipe.ops.transform.rotation = 0
    pipe.globals.settings.maxwidth = 200
    _, st = pipe.run_timed()
    assert [n for n, _ in st] == ["gofloat+demosaic", "rotatecrop", "to_lab+basecurve+from_lab+gamma"]
    # an unarmed run records nothing, and a session ends cleanly with no run at all
    import ctypes as C
    pipe.run()
    n = C.c_int(-1)
    assert ipa.lib().ipk_timing_begin() == 0 and ipa.lib().ipk_timing_end(None, 0, C.byref(n)) == 0 and n.value == 0


def test_errors_are_reported_not_computed(ipa):
    import ctypes as C
    import torch
    L = ipa.lib()
    z = torch.zeros(64, device="cuda")
    assert L.ipk_tolab(None, 4, 4, 0, None, None, None, None) == -2                         # null buffers
    assert L.ipk_demosaic_full(C.c_void_p(z.data_ptr()), 4, 4, b"RGXB", C.c_void_p(z.data_ptr()), None) == -2   # bad CFA
    assert b"CFA" in L.ipk_last_error()
    out = (C.c_size_t * 4)()
    assert L.ipk_size_image(0, 0, 0, 0, 9, 9, out) == -2                                    # the reference underflows below 10x10
    with pytest.raises(ipa.IpkError):
        ipa.Pipeline.new_from_source(ipa.RawImage(width=8, height=8, data=ipa.upload_u16(np.zeros((8, 8), np.uint16)), cfa="RGGB")).run()


# ---------------------------------------------------------------------------------------------
# randomized sweep: sizes, CFA phase, crops, levels, white balance, camera matrix, curve, output depth
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed", range(int(os.environ.get("IPK_RANDOM_SEEDS", "120"))))   # IPK_RANDOM_SEEDS=1000 for a soak run
def test_fused_randomized_configurations(ipa, orc, seed):
    rng = np.random.default_rng(1000 + seed)
    h = int(rng.integers(10, 80)); w = int(rng.choice([rng.integers(10, 300), 256, 260, 512, 4 * int(rng.integers(64, 200))]))
    cfa = (CFAS + [XT, W12])[int(rng.integers(0, 6))]
    crops = tuple(int(v) for v in rng.integers(0, 4, 4)) if rng.integers(0, 2) else (0, 0, 0, 0)
    orient = dict(rotation=int(rng.integers(0, 4)), fliph=bool(rng.integers(0, 2)), flipv=bool(rng.integers(0, 2))) if rng.integers(0, 3) == 0 else {}
    if h - crops[0] - crops[2] < 10 or w - crops[1] - crops[3] < 10:
        crops = (0, 0, 0, 0)
    is_float = bool(rng.integers(0, 2))
    black = float(rng.choice([0.0, 64.0, 256.5, 512.0, 1024.0])); white = float(rng.choice([1023.0, 4095.0, 16383.0, 65535.0]))
    raw = (rng.integers(0, int(white) + 200, size=(h, w))).astype(np.uint16)
    src = (raw.astype(np.float32) + rng.uniform(-0.5, 0.5, size=(h, w)).astype(np.float32)) if is_float else raw
    wb = (float(rng.uniform(0.5, 3.0)), float(rng.uniform(0.8, 1.2)), float(rng.uniform(0.5, 3.0)), float(rng.choice([np.nan, 1.0, 0.0])))
    cm = (util.cam_matrix() * rng.uniform(0.7, 1.3, size=(3, 1)).astype(np.float32) + rng.normal(0, 0.05, size=(3, 4)).astype(np.float32)).astype(np.float32)
    cm[:, 3] = 0.0 if rng.integers(0, 2) else rng.normal(0, 0.1, 3).astype(np.float32)
    npts = int(rng.integers(0, 5))
    xs = np.sort(rng.uniform(0.05, 0.95, npts)); ys = np.sort(rng.uniform(0.05, 0.95, npts))
    points = [(float(np.float32(a)), float(np.float32(b))) for a, b in zip(xs, ys)]
    exposure = float(rng.choice([0.0, 0.0, 0.3, -0.7]))
    linear = bool(rng.integers(0, 2))
    kw = dict(blacklevels=[black] * 4, whitelevels=[white] * 4, wb_coeffs=wb, cam_to_xyz_normalized=cm)
    pipe = ipa.Pipeline.new_from_source(_raw(ipa, src, cfa, is_float=is_float, crops=crops, **kw))
    pipe.ops.basecurve.points = points; pipe.ops.basecurve.exposure = exposure
    pipe.globals.settings.linear = linear
    for k, v in orient.items():
        setattr(pipe.ops.transform, k, v)
    d = lambda: _oracle_desc(orc, src, cfa, crops=crops, points=points, exposure=exposure, linear=linear, **orient, **kw)
    got = pipe.run(); assert pipe.last_used_fused
    assert_bits_equal(got.numpy(), orc.pipeline_run(d()), "randomized seed %d (%dx%d %s crops %r float %s)" % (seed, w, h, cfa, crops, is_float))
    ww, hh, o8 = pipe.output_8bit()
    assert np.array_equal(o8.cpu().numpy().reshape(hh, ww, 3), orc.pipeline_output_8bit(d()))
    ww, hh, o16 = pipe.output_16bit()
    assert np.array_equal(o16.cpu().numpy().view(np.uint16).reshape(hh, ww, 3), orc.pipeline_output_16bit(d()))


# ---------------------------------------------------------------------------------------------
# randomized sweep of the whole driver: every source kind, scaling, rotatecrop, orientation, cache on/off
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed", range(int(os.environ.get("IPK_RANDOM_SEEDS_DRIVER", "80"))))
def test_driver_randomized_configurations(ipa, orc, seed):
    import torch
    rng = np.random.default_rng(5000 + seed)
    h, w = int(rng.integers(24, 140)), int(rng.integers(24, 400))
    kind = int(rng.integers(0, 4))                              # 0 raw u16, 1 raw f32, 2 rgb8, 3 rgb16
    okw = {}
    if kind < 2:
        cfa = (CFAS + [XT, W12, "RGBE"])[int(rng.integers(0, 7))]
        crops = tuple(int(v) for v in rng.integers(0, 5, 4)) if rng.integers(0, 2) else (0, 0, 0, 0)
        raw = rng.integers(0, 16384, size=(h, w)).astype(np.uint16)
        src = raw.astype(np.float32) if kind == 1 else raw
        img = _raw(ipa, src, cfa, is_float=(kind == 1), crops=crops)
        od = lambda **kw: _oracle_desc(orc, src, cfa, crops=crops, **kw)
    else:
        src = rng.integers(0, 256 if kind == 2 else 65536, size=(h, w, 3)).astype(np.uint8 if kind == 2 else np.uint16)
        data = torch.from_numpy(src.ravel()).cuda() if kind == 2 else ipa.upload_u16(src)
        img = ipa.OtherImage(w, h, data, bits=8 if kind == 2 else 16)
        od = lambda **kw: orc.make_pipeline(src, **kw)
    pipe = ipa.Pipeline.new_from_source(img)
    if rng.integers(0, 2):
        okw["maxwidth"] = int(rng.integers(8, w + 20)); pipe.globals.settings.maxwidth = okw["maxwidth"]
    if rng.integers(0, 4) == 0:
        okw["maxheight"] = int(rng.integers(8, h + 20)); pipe.globals.settings.maxheight = okw["maxheight"]
    if rng.integers(0, 3) == 0:
        rc = [float(np.float32(v)) for v in rng.uniform(0, 0.2, 4)] + [float(np.float32(rng.choice([0.0, 0.0, rng.uniform(0.0, 0.6)])))]
        okw["rotatecrop"] = tuple(rc)
        pipe.ops.rotatecrop.crop_top, pipe.ops.rotatecrop.crop_right, pipe.ops.rotatecrop.crop_bottom, pipe.ops.rotatecrop.crop_left, pipe.ops.rotatecrop.rotation = rc
    if rng.integers(0, 2):
        okw.update(rotation=int(rng.integers(0, 4)), fliph=bool(rng.integers(0, 2)), flipv=bool(rng.integers(0, 2)))
        pipe.ops.transform.rotation, pipe.ops.transform.fliph, pipe.ops.transform.flipv = okw["rotation"], okw["fliph"], okw["flipv"]
    if rng.integers(0, 2):
        okw["exposure"] = float(rng.choice([0.4, -0.5])); pipe.ops.basecurve.exposure = okw["exposure"]
        if kind >= 2:
            okw["points"] = []
    okw["linear"] = bool(rng.integers(0, 2)); pipe.globals.settings.linear = okw["linear"]
    pipe.globals.settings.use_fastpath = bool(rng.integers(0, 2)); okw["use_fastpath"] = pipe.globals.settings.use_fastpath
    tag = "driver seed %d kind %d %dx%d %r" % (seed, kind, w, h, okw)
    assert pipe.sizes() == orc.pipeline_sizes(od(**okw)), tag
    if 0 in pipe.sizes()[0] + pipe.sizes()[1]:                  # an extreme aspect ratio scaled to nothing: an error, not a crash
        with pytest.raises(ipa.IpkError):
            pipe.run()
        return
    want = orc.pipeline_run(od(**okw))
    got = pipe.run()
    assert (got.height, got.width) == want.shape[:2], tag
    assert_bits_equal(got.numpy(), want, tag)
    cache = ipa.PipelineCache(1 << 28)
    assert_bits_equal(pipe.run(cache).numpy(), want, tag + " (cached, cold)")
    assert_bits_equal(pipe.run(cache).numpy(), want, tag + " (cached, hit)"); assert pipe.last_ops_run == 0
    cache.close()
    ww, hh, o8 = pipe.output_8bit()
    assert np.array_equal(o8.cpu().numpy().reshape(hh, ww, 3), orc.pipeline_output_8bit(od(**okw))), tag + " 8 bit"
    ww, hh, o16 = pipe.output_16bit()
    assert np.array_equal(o16.cpu().numpy().view(np.uint16).reshape(hh, ww, 3), orc.pipeline_output_16bit(od(**okw))), tag + " 16 bit"


def test_staged_pipeline_on_alternating_streams(ipa, orc):
    """The staged driver's scratch pool is stream-ordered: runs that alternate between two HIP streams without synchronising in
    between (blocks of the pool change streams) still equal the oracle"""
    import torch
    frames = [util.noise_u16(util.SEED + 500 + i, 120, 384) for i in range(2)]
    pipes = []
    for f in frames:
        p = ipa.Pipeline.new_from_source(_raw(ipa, f, "GBRG"))
        p.allow_fused = False
        pipes.append(p)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = []
    for rep in range(6):
        i = rep % 2
        with torch.cuda.stream(streams[(rep // 2 + i) % 2]):          # each pipeline visits both streams
            outs.append((i, pipes[i].run()))
    torch.cuda.synchronize()
    wants = [orc.pipeline_run(_oracle_desc(orc, f, "GBRG")) for f in frames]
    for i, o in outs:
        assert_bits_equal(o.numpy(), wants[i], "alternating streams, frame %d" % i)


@pytest.mark.parametrize("is_float", [False, True])
@pytest.mark.parametrize("cfa", ["RGGB", "GRBG", "GBRG", "BGGR"])
@pytest.mark.parametrize("shape,crops", [((256, 300), (0, 0, 0, 0)), ((257, 301), (0, 0, 0, 0)), ((300, 513), (1, 2, 0, 3)), ((321, 256), (3, 0, 1, 1)), ((512, 258), (0, 1, 1, 0))])
def test_orientations_run_in_rotated_space(ipa, orc, cfa, shape, crops, is_float):
    """Rotate90 / Rotate270 of a Bayer frame: the mosaic is permuted and the fused kernel works in rotated space (taps renamed into
    the original orientation's order, role table from the sensor parities, edge masks mapped) -- every Bayer phase, odd and even
    sizes, odd crops, all three outputs, bit-identical to the oracle's pipeline with OpTransform"""
    h, w = shape
    raw = util.noise_u16(util.SEED + 600 + h + w, h, w)
    src = raw.astype(np.float32) if is_float else raw
    if is_float:                                               # guarded samples (denormal, huge, non-finite): the literal redo in rotated space
        with np.errstate(over="ignore"):
            sp = util.SPECIALS * np.float32(16383.0)
        src[33, 20: 20 + sp.size] = sp
        src[100, 150] = -np.inf; src[h - 1, 0] = np.nan; src[0, w - 1] = np.float32(3e38)
    for rotation, fliph in [(1, False), (3, False), (0, True), (2, False), (2, True), (1, True), (3, True)]:     # the seven non-Normal orientations
        pipe = ipa.Pipeline.new_from_source(_raw(ipa, src, cfa, is_float=is_float, crops=crops))
        pipe.ops.transform.rotation = rotation; pipe.ops.transform.fliph = fliph
        okw = dict(crops=crops, rotation=rotation, fliph=fliph)
        want = orc.pipeline_run(_oracle_desc(orc, src, cfa, **okw))
        got = pipe.run()
        assert pipe.last_used_fused
        assert_bits_equal(got.numpy(), want, "rotated space rotation=%d fliph=%s %s %r %r" % (rotation, fliph, cfa, shape, crops))
        ww, hh, o8 = pipe.output_8bit()
        assert np.array_equal(o8.cpu().numpy().reshape(hh, ww, 3), orc.pipeline_output_8bit(_oracle_desc(orc, src, cfa, **okw)))
        ww, hh, o16 = pipe.output_16bit()
        assert np.array_equal(o16.cpu().numpy().view(np.uint16).reshape(hh, ww, 3), orc.pipeline_output_16bit(_oracle_desc(orc, src, cfa, **okw)))



@pytest.mark.parametrize("is_float", [False, True])
@pytest.mark.parametrize("cfa", [XT, W12])
@pytest.mark.parametrize("shape,crops", [((258, 300), (0, 0, 0, 0)), ((263, 301), (0, 0, 0, 0)), ((300, 517), (1, 2, 0, 3)), ((331, 262), (5, 0, 1, 4))])
def test_orientations_in_rotated_space_generic_cfa(ipa, orc, cfa, shape, crops, is_float):
    """The same for filters in generic-CFA mode (X-Trans 6 x 6 and a 12 x 12 pattern): the cell records are laid out for the rotated
    pattern (dimensions swapped, phase from the frame size), the taps keep the sensor's order; f32 frames carry NaN / inf /
    denormal samples, whose row windows take the literal bins in rotated space as well"""
    h, w = shape
    raw = util.noise_u16(util.SEED + 620 + h + w, h, w)
    src = raw.astype(np.float32) if is_float else raw
    if is_float:
        with np.errstate(over="ignore"):
            sp = util.SPECIALS * np.float32(16383.0)
        src[40, 30: 30 + sp.size] = sp
        src[90, 200] = -np.inf; src[150, 7] = np.nan; src[h - 1, w - 1] = np.inf; src[0, 0] = np.float32(1e-40)
    for rotation, fliph in [(1, False), (3, False), (0, True), (2, False), (2, True), (1, True), (3, True)]:
        pipe = ipa.Pipeline.new_from_source(_raw(ipa, src, cfa, is_float=is_float, crops=crops))
        pipe.ops.transform.rotation = rotation; pipe.ops.transform.fliph = fliph
        okw = dict(crops=crops, rotation=rotation, fliph=fliph)
        want = orc.pipeline_run(_oracle_desc(orc, src, cfa, **okw))
        got = pipe.run()
        assert pipe.last_used_fused
        assert_bits_equal(got.numpy(), want, "rotated space generic rotation=%d fliph=%s %r %r" % (rotation, fliph, shape, crops))
        ww, hh, o8 = pipe.output_8bit()
        assert np.array_equal(o8.cpu().numpy().reshape(hh, ww, 3), orc.pipeline_output_8bit(_oracle_desc(orc, src, cfa, **okw)))


@pytest.mark.parametrize("H,W,cfa", [(4000, 6000, "RGGB"), (10000, 10000, "GBRG"), (5760, 8640, XT)])
def test_full_size_rotated_space_equals_rotating_the_normal_result(ipa, H, W, cfa):
    """BASELINE.json frame sizes: the rotated-space launch (mosaic permuted, kernel renames its taps) against OpTransform applied to
    the Normal-orientation result of the same frame, which the tests above pin to the oracle -- every sample, all seven orientations"""
    import torch
    g = torch.Generator(device="cuda"); g.manual_seed(77)
    data = torch.randint(0, 16384, (H * W,), generator=g, device="cuda", dtype=torch.int32).to(torch.int16)
    def pipe_for(rotation, fliph):
        img = ipa.RawImage(width=W, height=H, data=data, cfa=cfa, blacklevels=[util.BLACK] * 4, whitelevels=[util.WHITE] * 4,
                           wb_coeffs=util.WB, cam_to_xyz_normalized=util.cam_matrix())
        p = ipa.Pipeline.new_from_source(img)
        p.ops.transform.rotation = rotation; p.ops.transform.fliph = fliph
        return p
    normal = pipe_for(0, False).run()
    for rotation, fliph in [(1, False), (3, False), (2, False), (0, True), (1, True), (2, True), (3, True)]:
        p = pipe_for(rotation, fliph)
        got = p.run()
        assert p.last_used_fused
        want = p.ops.transform.run(p.globals, normal)                    # rotate_buffer on the verified result
        assert (got.width, got.height) == (want.width, want.height)
        assert torch.equal(got.data.view(torch.int32), want.data.view(torch.int32)), (rotation, fliph)
        del got, want


@pytest.mark.parametrize("cfa", ["RGGB", XT])
@pytest.mark.parametrize("shape,is_float", [((10, 150001), False), ((12, 131072), True), ((120011, 10), False), ((65537, 12), True), ((10, 65535), False)])
def test_fused_extreme_aspect_ratios(ipa, orc, cfa, shape, is_float):
    """frames that are one strip wide and very tall, or ten rows high and wider than any strip / segment plan has seen: the wave-strip
    walker's task grid, its 32-bit indices and the row-band bookkeeping at their ends of the range (reference: any usize dimensions >= 10)"""
    h, w = shape
    raw = util.noise_u16(util.SEED + h + w, h, w)
    if is_float:
        raw = raw.astype(np.float32)
    pipe = ipa.Pipeline.new_from_source(_raw(ipa, raw, cfa, is_float=is_float))
    got = pipe.run()
    assert pipe.last_used_fused and (got.width, got.height) == (w, h)
    assert_bits_equal(got.numpy(), orc.pipeline_run(_oracle_desc(orc, raw, cfa)), "extreme shape %s %dx%d" % (cfa[:4], w, h))


@pytest.mark.parametrize("is_float,out_type,n,shape", [(True, "f32", 5, (96, 512)), (False, "u8", 3, (70, 300)), (False, "u16", 2, (64, 256)), (True, "f32", 1, (40, 260)),
                                                       (False, "f32", 70, (24, 256)), (True, "u8", 67, (30, 258))])
def test_batch_launch_equals_single_launches(ipa, is_float, out_type, n, shape):
    """ipk_raw_to_srgb_batch (one persistent launch per 64 frames) against ipk_raw_to_srgb frame by frame, bit for bit; n = 70 spans two launches"""
    import torch
    h, w = shape
    ot = {"f32": ipa.OUT_F32, "u8": ipa.OUT_U8, "u16": ipa.OUT_U16}[out_type]
    plan = ipa.FusedPlan(width=w, height=h, is_float=is_float, black0=util.BLACK, white0=util.WHITE, cfa="GRBG", wb_coeffs=util.WB,
                         cam_to_xyz_normalized=util.cam_matrix(), out_type=ot, linear=(out_type == "u16"))
    srcs = []
    for i in range(n):
        raw = util.noise_u16(util.SEED + 4000 + i, h, w)
        srcs.append(torch.from_numpy(raw.astype(np.float32).ravel()).cuda() if is_float else ipa.upload_u16(raw).reshape(-1))
    want = [plan.run(s, plan.new_output()).clone() for s in srcs]
    outs = [torch.zeros_like(x) for x in want]
    ipa.FusedBatchPlan(plan, srcs, outs).run()
    torch.cuda.synchronize()
    for i in range(n):
        assert torch.equal(outs[i].view(torch.uint8), want[i].view(torch.uint8)), "frame %d of %d" % (i, n)


def test_batch_launch_falls_back_frame_by_frame(ipa, orc):
    """a filter / parameter set without a batch variant of the kernel (X-Trans; no base curve): the same entry point, one launch per frame"""
    import torch
    h, w = 60, 270
    for kw in (dict(cfa=XT), dict(cfa="RGGB", points=())):
        plan = ipa.FusedPlan(width=w, height=h, is_float=False, black0=util.BLACK, white0=util.WHITE, wb_coeffs=util.WB, cam_to_xyz_normalized=util.cam_matrix(), **kw)
        srcs = [ipa.upload_u16(util.noise_u16(util.SEED + 4100 + i, h, w)).reshape(-1) for i in range(3)]
        want = [plan.run(s, plan.new_output()).clone() for s in srcs]
        outs = [torch.zeros_like(x) for x in want]
        ipa.FusedBatchPlan(plan, srcs, outs).run()
        torch.cuda.synchronize()
        for a, b in zip(outs, want):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32))


def test_task_queues_under_concurrent_launches(ipa):
    """the persistent kernel draws its tasks from a per-stream queue pair that consecutive launches use alternately: four host threads launch
    144 MP frames (large enough for drawn tasks) on two shared streams and on the null stream at once; every output must equal the frame's
    single-threaded result"""
    import threading
    import torch
    h, w = 12000, 12000
    plan = ipa.FusedPlan(width=w, height=h, is_float=False, black0=util.BLACK, white0=util.WHITE, cfa="RGGB", wb_coeffs=util.WB,
                         cam_to_xyz_normalized=util.cam_matrix(), out_type=ipa.OUT_U8)
    g = torch.Generator(device="cuda"); g.manual_seed(77)
    srcs = [torch.randint(0, 16384, (h * w,), device="cuda", generator=g, dtype=torch.int32).to(torch.int16) for _ in range(4)]
    want = [plan.run(s, plan.new_output()).clone() for s in srcs]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [[plan.new_output() for _ in range(6)] for _ in range(4)]
    errs = []

    def worker(t):
        try:
            for i in range(6):
                st = 0 if i % 3 == 2 else streams[(t + i) % 2].cuda_stream
                plan.run(srcs[t], outs[t][i], st)
        except Exception as e:      # noqa
            errs.append(repr(e))
    th = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    [x.start() for x in th]; [x.join() for x in th]
    torch.cuda.synchronize()
    assert not errs, errs
    for t in range(4):
        for i in range(6):
            assert torch.equal(outs[t][i], want[t]), (t, i)


@pytest.mark.parametrize("cfa,shape", [("GBRG", (270001, 300)), (XT, (180001, 517)), ("RGGB", (70000, 200))])
def test_drawn_tasks_on_tall_narrow_frames(ipa, orc, cfa, shape):
    """frames whose waves get 128+ rows each run with tasks DRAWN from the stream's queue (two or three strips, thousands of 32-row segments); the
    200-pixel-wide one takes the predicated-tail variant on the static schedule, one task per wave with takeovers inside the blocks: every sample against
    the oracle"""
    h, w = shape
    raw = util.noise_u16(util.SEED + h, h, w)
    pipe = ipa.Pipeline.new_from_source(_raw(ipa, raw, cfa))
    got = pipe.run()
    assert pipe.last_used_fused
    assert_bits_equal(got.numpy(), orc.pipeline_run(_oracle_desc(orc, raw, cfa)), "drawn tasks %s %dx%d" % (cfa[:4], w, h))


@pytest.mark.parametrize("h,w", [(6000, 8000), (12000, 12000)])
def test_takeovers_do_not_change_the_result(ipa, orc, h, w):
    """A wave that has run out of work (one task per wave at 48 MP; the stream's queue dry at 144 MP) takes over the lower half of the rows a slower wave
    of its block has not begun (one compare-and-swap on that wave's descriptor in LDS).  Which rows change hands depends on how the waves happened to
    run, so: a frame whose left half saturates (those strips run the cube-root branch in every slot and take about twice as long per row) against the
    oracle, and twenty further launches of it, alternating with launches of another frame on a second stream, bit-identical to the first."""
    import torch
    raw = util.noise_u16(util.SEED + 91, h, w)
    raw[:, : w // 2] = np.maximum(raw[:, : w // 2], np.uint16(16000))       # saturated half: every slot of these strips leaves the table
    raw[h // 3: h // 2, :] = np.uint16(700)                                  # and a dark band that takes lab_to_xyz's linear branches
    plan = ipa.FusedPlan(width=w, height=h, is_float=False, black0=util.BLACK, white0=util.WHITE, cfa="RGGB", wb_coeffs=util.WB,
                         cam_to_xyz_normalized=util.cam_matrix())
    src = ipa.upload_u16(raw)
    first = plan.run(src, plan.new_output()).clone()
    torch.cuda.synchronize()
    want = torch.from_numpy(orc.pipeline_run(_oracle_desc(orc, raw, "RGGB")).reshape(-1))
    assert torch.equal(first.cpu().view(torch.int32), want.view(torch.int32))
    other = ipa.upload_u16(util.noise_u16(util.SEED + 92, h, w))
    side = torch.cuda.Stream()
    outs = [plan.new_output() for _ in range(4)]
    scratch = plan.new_output()
    for i in range(20):
        plan.run(other, scratch, side.cuda_stream)                           # a second launch competing for the CUs changes every wave's pace
        plan.run(src, outs[i % 4])
        if i % 4 == 3:
            torch.cuda.synchronize()
            for o in outs:
                assert torch.equal(o.view(torch.int32), first.view(torch.int32)), i
                o.zero_()
    torch.cuda.synchronize()


def test_takeovers_generic_cfa_8bit_output(ipa, orc):
    """the same for the generic-CFA variant with the 8-bit epilogue: a 50 MP X-Trans frame (one static task per wave, takeovers inside the blocks) whose left
    half saturates, `output_8bit` six times, every one equal to the oracle's"""
    h, w = 5760, 8640
    raw = util.noise_u16(util.SEED + 93, h, w)
    raw[:, : w // 2] = np.maximum(raw[:, : w // 2], np.uint16(16000))
    pipe = ipa.Pipeline.new_from_source(_raw(ipa, raw, XTRANS))
    want = orc.pipeline_output_8bit(_oracle_desc(orc, raw, XTRANS))
    for i in range(6):
        ww, hh, o8 = pipe.output_8bit()
        assert pipe.last_used_fused and (ww, hh) == (w, h)
        assert np.array_equal(o8.cpu().numpy().reshape(hh, ww, 3), want), i


def test_drawn_tasks_in_a_batch_launch(ipa):
    """64 frames of 2304 x 480 in one persistent launch: 64 x 9 strips x 15 segments queue up behind the first round; each frame against its own launch"""
    import torch
    h, w, n = 480, 2304, 64
    plan = ipa.FusedPlan(width=w, height=h, is_float=True, black0=util.BLACK, white0=util.WHITE, cfa="BGGR", wb_coeffs=util.WB, cam_to_xyz_normalized=util.cam_matrix())
    g = torch.Generator(device="cuda"); g.manual_seed(11)
    srcs = [torch.randint(0, 16384, (h * w,), device="cuda", generator=g, dtype=torch.int32).to(torch.float32) for _ in range(n)]
    want = [plan.run(s, plan.new_output()).clone() for s in srcs]
    outs = [torch.zeros_like(x) for x in want]
    ipa.FusedBatchPlan(plan, srcs, outs).run()
    torch.cuda.synchronize()
    for i in range(n):
        assert torch.equal(outs[i].view(torch.int32), want[i].view(torch.int32)), i


def test_task_queue_survives_a_launch_that_failed_to_enqueue(ipa, orc):
    """A launch on a stream handle HIP rejects (the stream was destroyed) must come back as IPK_ERR_HIP -- nothing was enqueued -- and leave the
    library's task queues as they were: the next launches, on a live stream and large enough to draw tasks, equal the oracle.  (Round 2's queues
    alternated two counters per stream and a launch that never ran left the next-but-one frame incomplete.)  Runs in a child process: a HIP build
    that faults on a destroyed handle instead of rejecting it must not take the test session with it."""
    code = r'''
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, "tests")
import imagepipe_amd as ipa, oracle, util
ipa.init(0)
hip = C.CDLL("libamdhip64.so")
st = C.c_void_p()
assert hip.hipStreamCreate(C.byref(st)) == 0
dead = st.value
assert hip.hipStreamSynchronize(st) == 0 and hip.hipStreamDestroy(st) == 0
H, W = 12000, 12000                                 # 144 MP: above the queue's threshold, tasks are drawn
raw = util.noise_u16(util.SEED + 5, H, W)
cm = util.cam_matrix()
plan = ipa.FusedPlan(width=W, height=H, is_float=False, black0=util.BLACK, white0=util.WHITE, cfa="RGGB", wb_coeffs=util.WB, cam_to_xyz_normalized=cm)
src = ipa.upload_u16(raw)
out = plan.new_output()
try:
    plan.run(src, out, dead)
    print("NOTE: the destroyed stream was accepted")
except ipa._lib.IpkError as e:
    assert e.code == -3, e.code                     # IPK_ERR_HIP
    print("rejected:", e)
want = oracle.pipeline_run(oracle.make_pipeline(raw, cfa="RGGB", blacklevels=[util.BLACK] * 4, whitelevels=[util.WHITE] * 4, wb_coeffs=util.WB,
                                                cam_to_xyz_normalized=cm))
live = torch.cuda.current_stream().cuda_stream
for i in range(3):
    out.zero_()
    plan.run(src, out, live); torch.cuda.synchronize()
    util.assert_bits_equal(out.cpu().numpy().reshape(H, W, 3), want, "launch %d after the failed one" % i)
print("QUEUE_OK")
'''
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "QUEUE_OK" in r.stdout, (r.stdout + r.stderr)[-3000:]


def test_fused_launch_captured_in_a_graph_replays_correctly(ipa, orc):
    """The fused launch (tasks drawn from the stream's queue) captured into a HIP graph and replayed: every replay finds the queue as the last
    one left it -- zeroed by its last wave -- so each replay produces the whole frame (round 2's queues broke on the second replay), and a direct
    launch afterwards is still right.  Nothing is allocated at launch time (capture forbids it)."""
    import torch
    H, W = 12000, 12000                                    # 144 MP: tasks are drawn
    raw = util.noise_u16(util.SEED + 6, H, W)
    cm = util.cam_matrix()
    plan = ipa.FusedPlan(width=W, height=H, is_float=False, black0=util.BLACK, white0=util.WHITE, cfa="RGGB", wb_coeffs=util.WB, cam_to_xyz_normalized=cm)
    src = ipa.upload_u16(raw)
    out = plan.new_output()
    plan.run(src, out); torch.cuda.synchronize()          # warm: the stream's queue slot exists
    want = torch.from_numpy(orc.pipeline_run(orc.make_pipeline(raw, cfa="RGGB", blacklevels=[util.BLACK] * 4, whitelevels=[util.WHITE] * 4,
                                                                wb_coeffs=util.WB, cam_to_xyz_normalized=cm)).reshape(-1))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        plan.run(src, out, torch.cuda.current_stream().cuda_stream)
    for i in range(3):
        out.zero_()
        g.replay(); torch.cuda.synchronize()
        assert torch.equal(out.cpu().view(torch.int32), want.view(torch.int32)), "replay %d" % i
    out.zero_()
    plan.run(src, out); torch.cuda.synchronize()
    assert torch.equal(out.cpu().view(torch.int32), want.view(torch.int32)), "direct launch after the replays"


# ---------------------------------------------------------------------------------------------
# round 4: launches without a task queue that hold more tasks than the chip has waves; the stream probe
# ---------------------------------------------------------------------------------------------
def test_queue_less_launch_runs_every_task(ipa):
    """No queue slot for the stream (ipk_selftest_task_queue(0) forces what a full slot table does) and more tasks than waves: a 64-frame batch
    whose strips x frames exceed the chip's 4096 waves, and a single frame wider than 4096 strips would be -- here 70 x 72 strips.  Every wave
    then walks the tasks of its index a whole round of waves apart; each frame must equal the same frame through a queued launch of its own."""
    import torch
    L = ipa.lib()
    h, w, n = 40, 72 * 256, 64                              # 72 strips x 64 frames = 4608 one-segment tasks > 4096 waves
    plan = ipa.FusedPlan(width=w, height=h, is_float=True, black0=util.BLACK, white0=util.WHITE, cfa="GRBG", wb_coeffs=util.WB, cam_to_xyz_normalized=util.cam_matrix())
    g = torch.Generator(device="cuda"); g.manual_seed(12)
    srcs = [torch.randint(0, 16384, (h * w,), device="cuda", generator=g, dtype=torch.int32).to(torch.float32) for _ in range(n)]
    want = [plan.run(s, plan.new_output()).clone() for s in srcs]
    torch.cuda.synchronize()
    assert L.ipk_selftest_task_queue(0) == 0
    try:
        outs = [torch.zeros_like(x) for x in want]
        ipa.FusedBatchPlan(plan, srcs, outs).run()
        torch.cuda.synchronize()
        for i in range(n):
            assert torch.equal(outs[i].view(torch.int32), want[i].view(torch.int32)), i
        # single frames, queue-less, at sizes that would otherwise draw (144 MP) and that would not (24 MP)
        for hh, ww in ((12000, 12000), (4000, 6000)):
            p1 = ipa.FusedPlan(width=ww, height=hh, is_float=False, black0=util.BLACK, white0=util.WHITE, cfa="RGGB", wb_coeffs=util.WB, cam_to_xyz_normalized=util.cam_matrix())
            src = torch.randint(0, 16384, (hh * ww,), device="cuda", generator=g, dtype=torch.int32).to(torch.int16)
            a = p1.run(src, p1.new_output()).clone(); torch.cuda.synchronize()
            assert L.ipk_selftest_task_queue(1) == 0
            b = p1.run(src, p1.new_output()); torch.cuda.synchronize()
            assert L.ipk_selftest_task_queue(0) == 0
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (hh, ww)
            del a, b, src
    finally:
        assert L.ipk_selftest_task_queue(1) == 0


@pytest.mark.parametrize("cfa", CFAS)
@pytest.mark.parametrize("is_float,shape", [(True, (48, 256)), (False, (37, 515)), (True, (4000, 6000))])
def test_stream_probe_is_gofloat_plus_demosaic(ipa, orc, cfa, is_float, shape):
    """ipk_stream_probe -- the fused kernel's memory skeleton, the launch bench.py times as the kernel's ceiling -- has a defined result: the first
    three channels of demosaic::full(OpGoFloat(src)) (src/ops/gofloat.rs:126, src/ops/demosaic.rs:67-119).  Compared with the oracle's two stages
    sample by sample; the fused kernels start each demosaic sum from its first tap, so an exactly-zero channel may differ in its sign."""
    import torch
    h, w = shape
    if h > 1000 and cfa != "RGGB":
        pytest.skip("full size once")
    raw = util.noise_u16(util.SEED + 77 + h, h, w)
    src = raw.astype(np.float32) if is_float else raw
    plan = ipa.FusedPlan(width=w, height=h, is_float=is_float, black0=util.BLACK, white0=util.WHITE, cfa=cfa, wb_coeffs=util.WB, cam_to_xyz_normalized=util.cam_matrix())
    dev = torch.from_numpy(src.ravel()).cuda() if is_float else ipa.upload_u16(raw)
    out = torch.zeros(h * w * 3, dtype=torch.float32, device="cuda")
    plan.probe(dev, out); torch.cuda.synchronize()
    want = orc.demosaic_full(cfa, orc.gofloat_cfa(src, 0, 0, w, h, util.BLACK, util.WHITE))[:, :, :3]
    got = out.cpu().numpy().reshape(h, w, 3)
    zero = (got == 0.0) & (want == 0.0)
    assert_bits_equal(np.where(zero, np.float32(0.0), got), np.where(zero, np.float32(0.0), want), "stream probe %s %dx%d" % (cfa, w, h))


def test_stream_probe_refuses_what_it_has_no_variant_for(ipa):
    import torch
    xt = "GGRGGBGGBGGRBRGRBGGGBGGRGGRGGBRBGBRG"
    for kw in (dict(width=600, height=48, cfa=xt), dict(width=200, height=48, cfa="RGGB")):
        plan = ipa.FusedPlan(is_float=True, black0=util.BLACK, white0=util.WHITE, wb_coeffs=util.WB, cam_to_xyz_normalized=util.cam_matrix(), **kw)
        src = torch.zeros(kw["width"] * kw["height"], dtype=torch.float32, device="cuda")
        with pytest.raises(ipa._lib.IpkError) as e:
            plan.probe(src, torch.zeros(kw["width"] * kw["height"] * 3, dtype=torch.float32, device="cuda"))
        assert e.value.code == -5, e.value.code           # IPK_ERR_UNSUPPORTED


# ---------------------------------------------------------------------------------------------
# round 4: every fused variant a caller can reach, at the BASELINE frame sizes, against the oracle (drawn tasks and takeovers in play)
# ---------------------------------------------------------------------------------------------
USER5 = [(0.1, 0.07), (0.3, 0.27), (0.5, 0.6), (0.7, 0.82), (0.9, 0.95)]
VARIANTS = [                                   # (id, points, exposure, linear)
    ("user5", USER5, 0.0, False),             # 7 knots: the grid form of the curve in the common-parameter variant (CM = 2)
    ("nocurve", [], 0.0, False),              # OpBaseCurve returns its input (curves.rs:34-36): the generic variant, no curve
    ("exposure_linear", [], 0.5, True),       # 2 knots scaled by exp2(exposure), no OpGamma (gamma.rs:17): generic variant, runtime flags
    ("user5_linear", USER5, -0.4, True),      # grid curve in the generic variant
    ("unsorted", [(0.6, 0.5), (0.4, 0.3), (0.8, 0.9)], 0.0, False),   # knots out of order: the reference's search as it is (literal form)
]


@pytest.mark.parametrize("vid,points,exposure,linear", VARIANTS, ids=[v[0] for v in VARIANTS])
@pytest.mark.parametrize("is_float", [True, False], ids=["f32", "u16"])
@pytest.mark.parametrize("H,W", [(4000, 6000), (10000, 10000)], ids=["24MP", "100MP"])
def test_fused_variants_full_size_vs_oracle(ipa, orc, H, W, is_float, vid, points, exposure, linear):
    """Pipeline::run at BASELINE.json's frame sizes for the parameter sets that do NOT take the headline instantiation: user curves of more knots
    (src/ops/curves.rs:12-49), no curve, exposure, linear (src/ops/gamma.rs:16-26), f32 and u16 sources; every output sample against the oracle.
    The 100 MP u16 cases skip the costliest combinations to keep the suite's run time bounded (the oracle needs ~10 s per 100 MP frame)."""
    import torch
    if H == 10000 and not is_float and vid in ("exposure_linear", "unsorted"):
        pytest.skip("covered at 24 MP and, at 100 MP, from the f32 source")
    raw = util.noise_u16(util.SEED + 900 + H + len(points), H, W)
    src = raw.astype(np.float32) if is_float else raw
    pipe = ipa.Pipeline.new_from_source(_raw(ipa, src, "RGGB", is_float=is_float))
    pipe.ops.basecurve.points = points; pipe.ops.basecurve.exposure = exposure
    pipe.globals.settings.linear = linear
    got = pipe.run()
    assert pipe.last_used_fused
    want = torch.from_numpy(orc.pipeline_run(_oracle_desc(orc, src, "RGGB", points=points, exposure=exposure, linear=linear)).reshape(-1))
    if not torch.equal(got.data.cpu().view(torch.int32), want.view(torch.int32)):
        assert_bits_equal(got.numpy(), want.numpy().reshape(H, W, 3), "variant %s %s %dx%d" % (vid, "f32" if is_float else "u16", W, H))


@pytest.mark.parametrize("vid,points,exposure", [("user5", USER5, 0.0), ("nocurve", [], 0.0), ("exposure", [(0.5, 0.6)], 0.8)], ids=["user5", "nocurve", "exposure"])
@pytest.mark.parametrize("is_float", [True, False], ids=["f32", "u16"])
def test_fused_variants_quantised_outputs_24mp_vs_oracle(ipa, orc, is_float, vid, points, exposure):
    """output_8bit / output_16bit (src/pipeline.rs:404-421, :451-468: linear forced off / on) for the same parameter sets at 24 MP, every sample"""
    H, W = 4000, 6000
    raw = util.noise_u16(util.SEED + 950 + len(points), H, W)
    src = raw.astype(np.float32) if is_float else raw
    pipe = ipa.Pipeline.new_from_source(_raw(ipa, src, "RGGB", is_float=is_float))
    pipe.ops.basecurve.points = points; pipe.ops.basecurve.exposure = exposure
    desc = _oracle_desc(orc, src, "RGGB", points=points, exposure=exposure)
    w, h, o8 = pipe.output_8bit()
    assert pipe.last_used_fused
    assert np.array_equal(o8.cpu().numpy().reshape(h, w, 3), orc.pipeline_output_8bit(desc))
    w, h, o16 = pipe.output_16bit()
    assert np.array_equal(o16.cpu().numpy().view(np.uint16).reshape(h, w, 3), orc.pipeline_output_16bit(desc))


@pytest.mark.parametrize("h,w,src", [(6000, 8000, "u16"), (4000, 6000, "f32"), (1200, 3000, "f32"), (97, 300, "u16")])
def test_schedule_split_is_bit_identical(h, w, src):
    """ipk_fused_params.schedule = IPK_SCHED_SPLIT deals every wave two pieces half a frame apart (48 MP: it applies; the small frames: it falls back to
    the contiguous schedule); whatever the launch does with the field, every output sample equals the AUTO launch's, and a bad value is refused"""
    import torch
    import imagepipe_amd as ipa
    ipa.init(0)
    raw = util.noise_u16(util.SEED + 1234, h, w)
    raw[h // 3: h // 3 + h // 8, w // 4: w // 2] = 16383                       # a blown patch: the region the split schedule exists for
    src_t = torch.from_numpy(raw.astype(np.float32)).cuda().reshape(-1) if src == "f32" else torch.from_numpy(raw.view(np.int16)).cuda().reshape(-1)
    kw = dict(width=w, height=h, is_float=src == "f32", black0=util.BLACK, white0=util.WHITE, cfa="RGGB", wb_coeffs=util.WB, cam_to_xyz_normalized=util.cam_matrix())
    a = ipa.FusedPlan(**kw)
    b = ipa.FusedPlan(schedule=1, **kw)
    oa, ob = a.new_output(), b.new_output()
    a.run(src_t, oa); b.run(src_t, ob)
    torch.cuda.synchronize()
    assert torch.equal(oa.view(torch.int32), ob.view(torch.int32))
    if h * w < 1000000:
        import oracle
        want = oracle.pipeline_run(oracle.make_pipeline(raw.astype(np.float32) if src == "f32" else raw, cfa="RGGB", source_kind=1 if src == "f32" else 0,
                                                        blacklevels=[util.BLACK] * 4, whitelevels=[util.WHITE] * 4, wb_coeffs=util.WB, cam_to_xyz_normalized=util.cam_matrix()))
        util.assert_bits_equal(ob.cpu().numpy().reshape(h, w, 3), want, "split schedule vs oracle")
    with pytest.raises(ipa.IpkError, match="schedule"):
        ipa.FusedPlan(schedule=7, **kw).run(src_t, ob)
    # ... and through the pipeline driver: ipk_pipeline_desc.schedule reaches the fused launch, the hash chain does not see it (same results, same keys)
    img = ipa.RawImage(width=w, height=h, data=src_t, cfa="RGGB", is_float=src == "f32", blacklevels=[util.BLACK] * 4, whitelevels=[util.WHITE] * 4,
                       wb_coeffs=util.WB, cam_to_xyz_normalized=util.cam_matrix())
    pipe = ipa.Pipeline.new_from_source(img)
    keys = pipe.hashes()
    pipe.schedule = 1
    got = pipe.run().data
    assert pipe.last_used_fused and pipe.hashes() == keys
    assert torch.equal(got.view(torch.int32), oa.view(torch.int32))
