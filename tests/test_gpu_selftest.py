"""Exhaustive on-device proofs of the arithmetic shortcuts used by the fused kernel (bit-exactness is the bar):
constant division as multiply/fma, v_fract for the lookup weight, v_med3 for the gamma clamp, and the device
cbrtf against the HOST libm (what Rust's f32::cbrt calls) over enumerable domains."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# every divisor constant of the hot path (color_conversions.rs:7,158,168,177-179,181-187) + typical level ranges
CONSTS = [0.95047, 1.08883, 100.0, 255.0, 116.0, 500.0, 200.0, float(np.float32(24389.0) / np.float32(27.0)), 15871.0, 4031.0, 959.0, 65535.0, 16319.0]


@pytest.fixture(scope="module")
def L():
    import imagepipe_amd
    imagepipe_amd.init(0)
    return imagepipe_amd.lib()


def _cdiv(L, c, variant, lo, hi, special):
    n = C.c_uint64(); first = C.c_uint32()
    assert L.ipk_selftest_cdiv(c, variant, lo, hi, special, C.byref(n), C.byref(first)) == 0
    return n.value, first.value


@pytest.mark.parametrize("c", CONSTS)
def test_cdiv_fast_exact_on_every_f32_in_the_proven_zone(L, c):
    """variant 0 (with v_div_fixup): every f32 with 2^-100 <= |x| <= 2^100, plus +-0, +-inf and every NaN"""
    bad, first = _cdiv(L, c, 0, 2.0 ** -100, 2.0 ** 100, 1)
    assert bad == 0, "c=%r: %d mismatches, first x bits 0x%08x" % (c, bad, first)


@pytest.mark.parametrize("c", CONSTS)
def test_cdiv_three_steps_exact_on_finite_nonzero(L, c):
    """variant 1 (no fix-up): exact for finite nonzero x in the zone (what the guarded call sites rely on)"""
    bad, first = _cdiv(L, c, 1, 2.0 ** -100, 2.0 ** 100, 0)
    assert bad == 0, "c=%r: %d mismatches, first x bits 0x%08x" % (c, bad, first)


@pytest.mark.parametrize("c", [c for c in CONSTS if c > 1.0])
def test_cdiv_divisors_above_one_need_no_upper_guard(L, c):
    bad, first = _cdiv(L, c, 1, 2.0 ** -100, float(np.finfo(np.float32).max), 0)
    assert bad == 0, "c=%r: %d mismatches, first x bits 0x%08x" % (c, bad, first)


def test_cdiv_two_step_variant(L):
    """variant 2, q = fma(x, rc_hi, x*rc_lo) with 1/c = rc_hi + rc_lo: exact on every f32 in the zone for SOME constants only
    (the quotient of two 24-bit numbers can sit within 2^-49 of a rounding boundary, the size of this form's error), so the
    kernel uses it exactly for the constants proven here and the three-step form elsewhere"""
    res = {c: _cdiv(L, c, 2, 2.0 ** -100, 2.0 ** 100, 0)[0] for c in CONSTS}
    print("two-step constant division mismatches:", res)
    # the kernel's cdiv2() call sites (ipk_kernels.hip pointwise4_fast): the white point and the Lab scale constants
    for c in (0.95047, 1.08883, 100.0, 255.0, 116.0, 500.0, 200.0):
        assert res[c] == 0, "two-step division is not exact for c=%r" % c


def test_lut_weight_fract(L):
    n = C.c_uint64(); first = C.c_uint32()
    assert L.ipk_selftest_lut_weight(C.byref(n), C.byref(first)) == 0
    assert n.value == 0, "v_fract differs from pos - trunc(pos) at bits 0x%08x" % first.value


def test_clamp_med3(L):
    n = C.c_uint64(); first = C.c_uint32()
    assert L.ipk_selftest_clamp01(C.byref(n), C.byref(first)) == 0
    assert n.value == 0, "v_med3(v,0,1) differs from v.max(0).min(1) at bits 0x%08x" % first.value


@pytest.mark.parametrize("count", [1, 2, 3, 4, 5, 6, 7, 8, 9])
def test_cdiv_two_step_by_tap_counts(L, count):
    """generic-CFA demosaic (ipk_kernels.hip demosaic_gen_px): sum / count as fma(sum, rc_hi, sum*rc_lo) for every tap count a 3x3
    window can produce, every finite f32 sum in [2^-100, 2^100] (a zero sum gives zero in both forms)"""
    bad, first = _cdiv(L, float(count), 2, 2.0 ** -100, 2.0 ** 100, 0)
    assert bad == 0, "count=%d: %d mismatches, first x bits 0x%08x" % (count, bad, first)


def _device_cbrt(L, x, variant):
    import torch
    d = torch.from_numpy(x).cuda(); o = torch.empty_like(d)
    assert L.ipk_selftest_cbrtf(C.c_void_p(d.data_ptr()), C.c_void_p(o.data_ptr()), x.size, variant, None) == 0
    torch.cuda.synchronize()
    return o.cpu().numpy()


@pytest.mark.parametrize("variant", [0, 1])
def test_device_cbrtf_equals_host_libm_on_every_f32_in_1_to_8_and_beyond(L, orc, variant):
    """cbrtf depends on the mantissa and on xe mod 3 only (glibc 2.35 s_cbrtf.c), so [1,8) covers every case; a
    sweep of larger exponents and +inf covers the ldexp step"""
    lo, hi = np.float32(1.0).view(np.uint32), np.float32(8.0).view(np.uint32)
    x = np.arange(int(lo) + 1, int(hi), dtype=np.uint32).view(np.float32)
    got = _device_cbrt(L, x, variant)
    want = orc.lookup(orc.LUT_XYZ_LAB, x)             # x > 1: the oracle's lookup() calls the host libm's cbrtf directly
    bad = np.flatnonzero(got.view(np.uint32) != want.view(np.uint32))
    assert bad.size == 0, "%d mismatches, first x=%r" % (bad.size, x[bad[0]])
    big = np.concatenate([np.float32(2.0) ** np.arange(3, 128, dtype=np.float32) * np.float32(1.2345678), np.array([np.inf, 3.4e38], np.float32)]).astype(np.float32)
    assert np.array_equal(_device_cbrt(L, big, variant).view(np.uint32), orc.lookup(orc.LUT_XYZ_LAB, big).view(np.uint32))


def test_device_cbrtf_fast_form_every_f32_in_1_to_2(L, orc):
    lo, hi = np.float32(1.0).view(np.uint32), np.float32(2.0).view(np.uint32)
    x = np.arange(int(lo) + 1, int(hi), dtype=np.uint32).view(np.float32)
    got = _device_cbrt(L, x, 2)
    want = orc.lookup(orc.LUT_XYZ_LAB, x)
    bad = np.flatnonzero(got.view(np.uint32) != want.view(np.uint32))
    assert bad.size == 0, "%d mismatches, first x=%r" % (bad.size, x[bad[0]])




@pytest.mark.parametrize("exposure,points", [(0.0, [(0.5, 0.6)]), (0.0, []), (0.7, []), (-1.3, [(0.5, 0.6)]), (0.0, [(0.25, 0.1)]), (0.0, [(0.7, 0.95)]),
                                             (0.4, [(0.1, 0.3)]), (0.0, [(0.5, 0.5)]), (0.0, [(0.999, 0.001)])])
def test_base_curve_arithmetic_form_equals_the_literal_search_on_every_f32(L, exposure, points):
    """SplineFunc::interpolate (curves.rs:126-157: end clamps, binary search, exact knot hit) against the fused kernels' form for curves of 2 or 3
    knots, which folds the lower clamp and the knot hit into the polynomial (ipk_device.hpp spline_interpolate_3a): all 2^32 arguments"""
    if not points and abs(exposure) < 0.001:
        pytest.skip("no curve at all: OpBaseCurve returns its input (curves.rs:34-36)")
    flat = (C.c_float * max(2, 2 * len(points)))(*[c for p in points for c in p])
    n = C.c_uint64(); first = C.c_uint32()
    assert L.ipk_selftest_spline3(C.c_float(exposure), flat, len(points), C.byref(n), C.byref(first)) == 0
    assert n.value == 0, (n.value, hex(first.value))


USER5 = [(0.1, 0.07), (0.3, 0.27), (0.5, 0.6), (0.7, 0.82), (0.9, 0.95)]


@pytest.mark.parametrize("exposure,points", [(0.0, USER5), (0.6, USER5), (0.0, [(0.2, 0.1), (0.4, 0.5), (0.6, 0.55), (0.8, 0.9)]),
                                             (0.0, [(0.2, 0.9), (0.4, 0.1), (0.6, 0.95), (0.8, 0.05)]), (0.0, [(0.3, 0.2), (0.6, 0.8)]),
                                             (0.0, [((i + 1) / 65.0, ((i + 1) / 65.0) ** 0.8) for i in range(64)]),           # 66 knots, the most the library takes
                                             (0.0, [(0.25, 0.25), (0.2541, 0.26), (0.75, 0.8)]),                            # two knots in neighbouring grid cells (64 and 65)
                                             (-0.5, [(0.0, 0.1), (0.5, 0.4), (1.0, 0.9)])])                                   # user points on the ends: no auto-added knots
def test_base_curve_grid_form_equals_the_literal_search_on_every_f32(L, exposure, points):
    """SplineFunc::interpolate (curves.rs:126-157) against the grid form the kernels use for curves of four or more knots (ipk_device.hpp
    spline_interpolate_grid: segment from a 256-cell grid plus one comparison, lower clamp and knot hits as arithmetic): all 2^32 arguments
    (NaN arguments excepted, which never reach the curve of a lane whose result is used)"""
    flat = (C.c_float * (2 * len(points)))(*[c for p in points for c in p])
    n = C.c_uint64(); first = C.c_uint32()
    assert L.ipk_selftest_spline3(C.c_float(exposure), flat, len(points), C.byref(n), C.byref(first)) == 0
    assert n.value == 0, (n.value, hex(first.value))


@pytest.mark.parametrize("points", [[(0.3, 0.2), (0.3004, 0.21), (0.3008, 0.22), (0.7, 0.8)],       # three knots inside one grid cell (1/256 wide)
                                    [(0.6, 0.5), (0.4, 0.3), (0.8, 0.9)],                            # not sorted: the reference searches it as it is
                                    [(0.2, 0.1), (0.4, -0.0), (0.6, 0.55), (0.8, 0.9)]])            # a knot ordinate of -0.0
def test_base_curve_grid_form_is_refused_where_it_would_not_hold(L, points):
    flat = (C.c_float * (2 * len(points)))(*[c for p in points for c in p])
    n = C.c_uint64(); first = C.c_uint32()
    assert L.ipk_selftest_spline3(C.c_float(0.0), flat, len(points), C.byref(n), C.byref(first)) == -5    # IPK_ERR_UNSUPPORTED: the literal search stays


def test_base_curve_arithmetic_form_is_refused_for_a_negative_zero_ordinate(L):
    flat = (C.c_float * 2)(0.5, -0.0)
    n = C.c_uint64(); first = C.c_uint32()
    assert L.ipk_selftest_spline3(C.c_float(0.0), flat, 1, C.byref(n), C.byref(first)) == -5       # IPK_ERR_UNSUPPORTED: the kernels keep the select form


def test_output8bit_packed_form_is_exact_on_every_f32(L):
    """output8bit = (v*256).max(0).min(255) as u8 (color_conversions.rs:323-326) against v_cvt_pk_u8_f32 of floor(v*256), the form the
    kernels pack their 8-bit output with, and against the saturating v_cvt_u32_f32 + unsigned min: all 2^32 inputs.  (Without the floor the
    instruction rounds to nearest: 41 910 144 mismatches.)"""
    for variant, expect_zero in ((1, True), (2, True), (0, False)):
        n = C.c_uint64(); first = C.c_uint32()
        assert L.ipk_selftest_quant8(variant, C.byref(n), C.byref(first)) == 0
        assert (n.value == 0) == expect_zero, (variant, n.value, hex(first.value))


def test_output16bit_packed_form_is_exact_on_every_f32(L):
    """output16bit = (v*65535).round().max(0).min(65535) as u16 (color_conversions.rs:327-330; round = halves away from zero) against
    floor(v*65535 + 0.5) through the saturating v_cvt_u32_f32 / v_cvt_pk_u16_u32, both halves of the packed dword: all 2^32 inputs"""
    n = C.c_uint64(123); first = C.c_uint32()
    assert L.ipk_selftest_quant16(C.byref(n), C.byref(first)) == 0, L.ipk_last_error()
    assert n.value == 0, (n.value, hex(first.value))


def test_gamma_plus_output8bit_as_one_step_lookup_is_exact_on_every_f32(L):
    """the 8-bit-output kernels replace OpGamma's table step + output8bit (src/ops/gamma.rs:22, src/color_conversions.rs:323-326) by ONE lookup in a table of
    8192 {k, threshold} steps built on the device from the gamma table (inside one table segment the 8-bit value changes at most once): the literal
    composition against clamp + step lookup on all 2^32 inputs -- negative, above 1, NaN, inf, denormal included"""
    n = C.c_uint64(123); first = C.c_uint32()
    assert L.ipk_selftest_q8(C.byref(n), C.byref(first)) == 0, L.ipk_last_error()
    assert n.value == 0, (n.value, hex(first.value))


def test_init_time_libm_check_agrees_with_the_exhaustive_one(L):
    """ipk_init compares the host's cbrtf with the device routine on 65 536 arguments and reports it (ipk_host_libm_matches); on this
    host (glibc 2.35, the routine the device ports) they agree -- the exhaustive tests above say the same for every argument"""
    n = C.c_size_t(123)
    assert L.ipk_host_libm_matches(C.byref(n)) == 1 and n.value == 0


def test_shutdown_and_init_again():
    """ipk_shutdown releases everything the context holds (tables, CFA records, scratch pool, host lanes, the streams' task queues); a second
    ipk_init in the same process starts clean and computes the same frame.  Runs in its own process: the session fixture keeps its context."""
    import subprocess, sys, os
    code = ("import numpy as np, torch, imagepipe_amd as ipa, util; from imagepipe_amd import _lib\n"
            "def frame():\n"
            "    raw = util.noise_u16(util.SEED + 9, 6000, 8000)\n"
            "    plan = ipa.FusedPlan(width=8000, height=6000, is_float=False, black0=util.BLACK, white0=util.WHITE, cfa='RGGB', wb_coeffs=util.WB, cam_to_xyz_normalized=util.cam_matrix(), out_type=ipa.OUT_U8)\n"
            "    o = plan.run(ipa.upload_u16(raw), plan.new_output()); torch.cuda.synchronize(); return o.cpu()\n"
            "ipa.init(0); a = frame(); a2 = frame()\n"
            "_lib.load().ipk_shutdown(); _lib.check(_lib.load().ipk_init(0), 'ipk_init')\n"
            "b = frame(); assert torch.equal(a, b) and torch.equal(a, a2); print('REINIT_OK')\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, PYTHONPATH=os.path.join(root, "tests") + os.pathsep + root))
    assert r.returncode == 0 and "REINIT_OK" in r.stdout, (r.stdout + r.stderr)[-3000:]


def test_mix_probe_moves_the_bytes_it_claims(L):
    """ipk_mix_probe (bench.py's mix_ceiling: 16 bytes read, 48 written per lane) really reads every input group and writes three output groups for it:
    block b's 256 input groups land three times in its 768 output groups, as lane-contiguous sweeps"""
    import torch
    n16 = 256 * 37                                         # whole blocks only: the probe refuses anything else
    src = torch.arange(n16 * 4, dtype=torch.float32, device="cuda")
    dst = torch.full((n16 * 12,), -1.0, dtype=torch.float32, device="cuda")
    assert L.ipk_mix_probe(src.data_ptr(), dst.data_ptr(), n16 * 16, None) == 0
    torch.cuda.synchronize()
    s = src.cpu().numpy().reshape(n16, 4); d = dst.cpu().numpy().reshape(n16 * 3, 4)
    for g in (0, 1, 255, 256, 257, 256 * 36 + 255, n16 - 1):
        b, t = divmod(g, 256)
        for k in range(3):
            assert (d[b * 768 + k * 256 + t] == s[g]).all(), (g, k)
    assert (dst.cpu().numpy() >= 0).all()                   # every output group written
    assert L.ipk_mix_probe(src.data_ptr() + 4, dst.data_ptr(), n16 * 16, None) == -2       # misaligned source
    assert L.ipk_mix_probe(src.data_ptr(), dst.data_ptr(), n16 * 16 - 1600, None) == -2     # not whole blocks
