"""Why the kernels carry glibc's cbrtf bit for bit (ipk_device.hpp cbrtf_glibc*) instead of a cheaper cube root, although
BASELINE.json's north_star allows 1 ULP on the FINAL output: the experiment VERDICT r01 asked for, run on the CPU with the oracle as
the pipeline.

XYZ_LAB_TRANSFORM.lookup calls the host libm's cbrtf for every Lab ratio above 1 (src/color_conversions.rs:103-104,123).  glibc's
routine is not correctly rounded, so ANY other cube root -- here the best possible one, the correctly rounded result -- differs from it
by 1 ULP on a few per cent of the inputs.  This test replaces only that one function value, pushes both Lab buffers through the
reference's remaining stages (base curve, lab_to_rgb with its cubes and its matrix with negative entries, gamma table) and measures the
distance of the final sRGB samples: a 1-ULP change of one cube root reaches the output amplified (x3 by the cube, then by the
cancellation in XYZ->RGB), far beyond the 1 ULP the tolerance grants.  A faster cbrtf would therefore not be a legal opt-in
(it costs 75 of the fused kernel's 259 instructions per pixel on uniform noise); the exact routine stays."""
import numpy as np

import util

F = np.float32


def _restated_tolab(orc, px4, wb, cm, f_of_ratio):
    """OpToLab::run / camera_to_lab / xyz_to_lab (src/color_conversions.rs:42-55,156-169) in numpy float32 -- every operation
    individually rounded, left to right as the reference writes it -- with the table lookup injected"""
    mul = orc.normalize_wbs(wb)
    ch = [np.minimum(px4[..., i] * F(mul[i]), F(1.0)) for i in range(4)]
    m = np.asarray(cm, np.float32).reshape(3, 4)
    xyz = [((ch[0] * m[i, 0] + ch[1] * m[i, 1]) + ch[2] * m[i, 2]) + ch[3] * m[i, 3] for i in range(3)]
    xr, yr, zr = xyz[0] / F(0.95047), xyz[1] / F(1.0), xyz[2] / F(1.08883)
    fx, fy, fz = f_of_ratio(xr), f_of_ratio(yr), f_of_ratio(zr)
    l = F(116.0) * fy - F(16.0)
    a = F(500.0) * (fx - fy)
    b = F(200.0) * (fy - fz)
    return np.stack([l / F(100.0), (a + F(127.0)) / F(255.0), (b + F(127.0)) / F(255.0)], axis=-1), (xr, yr, zr)


def test_one_ulp_in_cbrtf_exceeds_the_one_ulp_output_tolerance(orc):
    h, w = 256, 1024
    rng = np.random.default_rng(11)
    # demosaiced-looking RGBE pixels of the bench's synthetic camera: uniform channels, E = 0; white balance (2, 1, 1.5) clips R and B
    px4 = np.zeros((h, w, 4), np.float32)
    px4[..., :3] = rng.uniform(-0.03, 1.0, size=(h, w, 3)).astype(np.float32)
    cm = util.cam_matrix()

    def exact(r):                      # the reference's lookup: table + linear interpolation inside [0,1], libm cbrtf above 1
        return orc.lookup(0, r)

    def alt(r):                        # the same, with the CORRECTLY ROUNDED cube root above 1
        f = orc.lookup(0, r)
        hi = r > F(1.0)
        f[hi] = np.cbrt(r[hi].astype(np.float64)).astype(np.float32)
        return f

    lab_exact, ratios = _restated_tolab(orc, px4, util.WB, cm, exact)
    # the numpy restatement with the exact lookup IS the oracle's OpToLab, bit for bit -- so the only difference below is the cube root
    util.assert_bits_equal(lab_exact, orc.tolab(px4, util.WB, cm), "numpy restatement of OpToLab")
    lab_alt, _ = _restated_tolab(orc, px4, util.WB, cm, alt)

    touched = np.zeros((h, w), bool)
    n_hi = n_diff = 0
    for r in ratios:
        hi = r > F(1.0)
        n_hi += int(hi.sum())
        d = exact(r)[hi].view(np.int32).astype(np.int64) - alt(r)[hi].view(np.int32)
        assert np.abs(d).max() <= 1                           # the two cube roots never differ by more than one ULP ...
        n_diff += int((d != 0).sum())
    frac = n_diff / n_hi
    assert 0.02 < frac < 0.2, frac                            # ... and do differ on a few per cent of the arguments (8 % on (1,2))
    touched = np.any(lab_exact.view(np.uint32) != lab_alt.view(np.uint32), axis=-1)

    def rest(lab):                     # OpBaseCurve (default raw curve) -> OpFromLab -> OpGamma, the oracle's own stages
        return orc.gamma(orc.fromlab(orc.basecurve(lab, 0.0, [(0.5, 0.6)])))

    out_exact, out_alt = rest(lab_exact), rest(lab_alt)
    a = out_exact[touched].view(np.int32).astype(np.int64); b = out_alt[touched].view(np.int32).astype(np.int64)
    ulps = np.abs(a - b)
    worst = int(ulps.max())
    over = float((ulps.max(axis=-1) > 1).mean())
    print("pixels with a changed cube root: %d; worst final-output distance %d ULP; %.1f %% of them beyond 1 ULP" % (int(touched.sum()), worst, 100 * over))
    assert touched.sum() > 1000
    assert worst > 1, "a 1-ULP cube root stayed within the 1-ULP output tolerance: the cheaper routine would be legal after all"
    assert over > 0.05
