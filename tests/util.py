"""Shared helpers for the parity tests: deterministic synthetic frames and bit-level comparison."""
import numpy as np

SEED = 0x1A6E9195
MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(seed, n):
    """n outputs of SplitMix64 started at `seed` (vectorised: output i uses state seed + (i+1)*golden)."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        z = (np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)) & MASK
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & MASK
        return z ^ (z >> np.uint64(31))


def noise_u16(seed, h, w, maxval=16383):
    """uniform integers in [0, maxval] -- worst case for LUT locality"""
    r = splitmix64(seed, h * w)
    return (r % np.uint64(maxval + 1)).astype(np.uint16).reshape(h, w)


def smooth_u16(seed, h, w):
    """diagonal gradient ((row+col) mod 4096)*4 + 6-bit noise -- realistic LUT locality (max 16383+63 clipped)"""
    rr, cc = np.meshgrid(np.arange(h, dtype=np.uint32), np.arange(w, dtype=np.uint32), indexing="ij")
    base = ((rr + cc) % 4096) * 4
    n = (splitmix64(seed, h * w) & np.uint64(63)).astype(np.uint32).reshape(h, w)
    return np.minimum(base + n, 16383).astype(np.uint16)


def uniform_f32(seed, n, lo=0.0, hi=1.0):
    r = splitmix64(seed, n)
    u = (r >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    return (lo + (hi - lo) * u).astype(np.float32)


# the synthetic camera of SURVEY.md 8(d)
BLACK, WHITE = 512.0, 16383.0
WB = (2.0, 1.0, 1.5, float("nan"))


def cam_matrix():
    """SRGB_D65_43 with rows scaled so that saturated pixels exceed the white point (exercises the cbrtf path)"""
    m = np.array([[0.4124564, 0.3575761, 0.1804375, 0.0],
                  [0.2126729, 0.7151522, 0.0721750, 0.0],
                  [0.0193339, 0.1191920, 0.9503041, 0.0]], dtype=np.float32)
    return (m * np.array([[1.10], [1.05], [1.20]], dtype=np.float32)).astype(np.float32)


SPECIALS = np.array([0.0, -0.0, 1.0, -1.0, 0.5, 1.5, 2.0, 8.0, 1e-3, -1e-3, 0.008856452, 0.0088564521, 0.04045, 0.0031308,
                     1e-30, -1e-30, 1e-40, -1e-40, 1e30, -1e30, np.inf, -np.inf, np.nan, 0.99999994, 1.0000001, 3.4e38,
                     1.17549435e-38, 0.9504700, 1.08883, 0.95047, 255.0, 65535.0, 0.33333334, 0.6, 0.5, 0.49999997, 0.50000006],
                    dtype=np.float32)


def ulp_diff(a, b):
    """max |ulp distance| between two f32 arrays (NaN vs NaN counts as 0, NaN vs number as inf)"""
    a = np.ascontiguousarray(a, np.float32).ravel(); b = np.ascontiguousarray(b, np.float32).ravel()
    na, nb = np.isnan(a), np.isnan(b)
    if np.any(na != nb):
        return np.inf
    ai = a.view(np.int32).astype(np.int64); bi = b.view(np.int32).astype(np.int64)
    ai = np.where(ai < 0, np.int64(-2147483648) - ai, ai); bi = np.where(bi < 0, np.int64(-2147483648) - bi, bi)
    d = np.abs(ai - bi); d[na] = 0
    return int(d.max()) if d.size else 0


def assert_bits_equal(got, want, what=""):
    """Bit-exact f32 equality (any NaN == any NaN).  The product's bar is 0 ULP; BASELINE.json allows 1 ULP."""
    got = np.ascontiguousarray(got, np.float32); want = np.ascontiguousarray(want, np.float32)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    g = got.ravel().view(np.uint32); w = want.ravel().view(np.uint32)
    both_nan = np.isnan(got.ravel()) & np.isnan(want.ravel())
    bad = (g != w) & ~both_nan
    if bad.any():
        i = int(np.flatnonzero(bad)[0])
        raise AssertionError("%s: %d of %d samples differ (max %s ULP); first at flat index %d: got %r (0x%08x) want %r (0x%08x)" % (
            what, int(bad.sum()), bad.size, ulp_diff(got, want), i, got.ravel()[i], g[i], want.ravel()[i], w[i]))
