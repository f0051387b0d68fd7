"""A SECOND, independent restatement of the reference functions that no reference test pins (OpGoFloat::run_raw,
demosaic::full, transform_buffer / scaled_demosaic at a non-identity scale), written in plain Python loops straight from the Rust text --
sums/counts arrays, `lookups[48][48][9]`, the window loops -- with every f32 operation a numpy float32 scalar operation.  It shares
no code with oracle/imagepipe_oracle.c (different language, different loop shapes: the C oracle indexes precomputed tables), so a
transcription slip in either one shows up as a bit mismatch on these small frames.  What it CANNOT catch is a shared misreading of
rawloader's CFA::color_at (row-major letters over the tile, tiled periodically) -- rawloader's source is not on disk."""
import numpy as np
import pytest

import util

F = np.float32
XTRANS = "GGRGGBGGBGGRBRGRBGGGBGGRGGRGGBRBGBRG"


def _color_at(pat):
    """rawloader's CFA as the reference uses it (demosaic.rs:80,86; scaling.rs:110): letters row-major over a width x height tile; the shape is
    implied by 4 / 36 / 144 letters or stated as "WxH:" in front of them (the only way to pass 16 letters)"""
    if ":" in pat:
        dims, pat = pat.split(":")
        wide, high = (int(v) for v in dims.split("x"))
        assert wide * high == len(pat)
    else:
        wide = high = {4: 2, 36: 6, 144: 12}[len(pat)]
    tile = [["RGBE".index(c) for c in pat[r * wide:(r + 1) * wide]] for r in range(high)]
    return lambda row, col: tile[row % high][col % wide]


def _full(pat, buf):
    """src/ops/demosaic.rs:67-119, literally"""
    color_at = _color_at(pat)
    h, w = buf.shape
    offsets = [(-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 0), (0, 1), (1, -1), (1, 0), (1, 1)]
    lookups = [[[0] * 9 for _ in range(48)] for _ in range(48)]
    for row in range(48):
        for col in range(48):
            pix = color_at(row, col)
            for i, (dy, dx) in enumerate(offsets):
                o = color_at(48 + dy + row, 48 + dx + col)
                lookups[row][col][i] = o if (o != pix or (dx == 0 and dy == 0)) else 4
    out = np.zeros((h, w, 4), np.float32)
    for row in range(h):
        for col in range(w):
            colors = lookups[row % 48][col % 48]
            sums = [F(0.0)] * 5
            counts = [F(0.0)] * 5
            for i, (dy, dx) in enumerate(offsets):
                r, c = row + dy, col + dx
                if 0 <= r < h and 0 <= c < w:
                    sums[colors[i]] = F(sums[colors[i]] + buf[r, c])
                    counts[colors[i]] = F(counts[colors[i]] + F(1.0))
            for c in range(4):
                if counts[c] > 0.0:
                    out[row, col, c] = F(sums[c] / counts[c])
    return out


def _as_usize(f):
    return 0 if not (f > 0) else int(f)           # `as usize`: saturating, NaN -> 0 (values here are far below 2^63)


def _transform_buffer(src, width, height, tl, tr, bl, nwidth, nheight, components, pat=None):
    """src/scaling.rs:51-130, literally (f32 source)"""
    color_at = _color_at(pat) if pat else None
    out = np.zeros(nwidth * nheight * components, np.float32)
    sxx = F(F(F(tr[0]) - F(tl[0])) / F(nwidth - 1)); sxy = F(F(F(tr[1]) - F(tl[1])) / F(nwidth - 1))
    syx = F(F(F(bl[0]) - F(tl[0])) / F(nheight - 1)); syy = F(F(F(bl[1]) - F(tl[1])) / F(nheight - 1))
    for row in range(nheight):
        from_x_r = F(F(tl[0]) + F(syx * F(row))); to_x_r = F(F(tl[0]) + F(syx * F(row + 1)))
        from_y_r = F(F(tl[1]) + F(syy * F(row))); to_y_r = F(F(tl[1]) + F(syy * F(row + 1)))
        cx_r = F(F(F(F(tl[0]) + F(syx * F(row))) + F(syx / F(2.0))) - F(0.5))
        cy_r = F(F(F(F(tl[1]) + F(syy * F(row))) + F(syy / F(2.0))) - F(0.5))
        for col in range(nwidth):
            from_x = min(width - 1, _as_usize(np.floor(F(from_x_r + F(sxx * F(col))))))
            to_x = min(width - 1, _as_usize(np.floor(F(to_x_r + F(sxx * F(col + 1))))))
            from_y = min(height - 1, _as_usize(np.floor(F(from_y_r + F(sxy * F(col))))))
            to_y = min(height - 1, _as_usize(np.floor(F(to_y_r + F(sxy * F(col + 1))))))
            cx = F(F(cx_r + F(sxx * F(col))) + F(sxx / F(2.0)))
            cy = F(F(cy_r + F(sxy * F(col))) + F(sxy / F(2.0)))
            sums = [F(0.0)] * 4
            counts = [F(0.0)] * 4
            for y in range(from_y, to_y + 1):
                for x in range(from_x, to_x + 1):
                    dx = F(F(F(x) - cx) / sxx)
                    dy = F(F(F(y) - cy) / syy)
                    factor = F(F(F(1.0) - F(dx * dx)) - F(dy * dy))
                    if factor < 0.0:
                        factor = F(0.0)
                    if color_at:
                        c = color_at(y, x)
                        sums[c] = F(sums[c] + F(src[y * width + x] * factor))
                        counts[c] = F(counts[c] + factor)
                    else:
                        for c in range(components):
                            sums[c] = F(sums[c] + F(src[(y * width + x) * components + c] * factor))
                            counts[c] = F(counts[c] + factor)
            for c in range(components):
                if counts[c] > 0.0:
                    out[(row * nwidth + col) * components + c] = F(sums[c] / counts[c])
    return out


def test_gofloat_run_raw_hand_values(orc):
    """src/ops/gofloat.rs:122-130: ((v - black) / (white - black)).min(1.0), no lower clamp; only levels[0] count for a CFA image"""
    raw = np.zeros((10, 10), np.uint16)
    vals = [0, 32, 64, 65, 543, 1023, 1024, 4095, 65535]
    raw[0, :len(vals)] = vals
    got = orc.gofloat_cfa(raw, 0, 0, 10, 10, 64.0, 1023.0)[0, :len(vals)]
    want = [min(np.float32(np.float32(np.float32(v) - np.float32(64.0)) / np.float32(959.0)), np.float32(1.0)) for v in vals]
    assert got.tolist() == [float(x) for x in want]
    assert got[0] == np.float32(-64.0) / np.float32(959.0) and got[2] == 0.0 and got[5] == 1.0 and got[6] == 1.0 and got[1] < 0
    # crop: out[row][col] = raw[row + y][col + x]
    raw = (np.arange(12 * 14, dtype=np.uint16) * 37 % 4096).reshape(12, 14)
    g2 = orc.gofloat_cfa(raw, 3, 1, 10, 10, 0.0, 4095.0)
    assert g2.shape == (10, 10) and g2[0, 0] == np.float32(raw[1, 3]) / np.float32(4095.0) and g2[9, 9] == np.float32(raw[10, 12]) / np.float32(4095.0)


L16 = "RGBGRBGGGBGRGRBG"                        # sixteen letters, no symmetry: read 2 wide x 8 high and 8 wide x 2 high they are different filters


@pytest.mark.parametrize("pat,h,w", [("RGGB", 11, 13), ("GBRG", 10, 12), ("RGBE", 10, 11), (XTRANS, 13, 14), ("2x8:" + L16, 19, 13), ("8x2:" + L16, 11, 21),
                                     ("4x4:" + L16, 10, 10)])
def test_demosaic_full_against_a_second_restatement(orc, pat, h, w):
    buf = util.uniform_f32(util.SEED + 900 + h, h * w, -0.05, 1.0).reshape(h, w)
    buf[2, 3] = np.float32(-0.0); buf[5, 5] = 0.0
    util.assert_bits_equal(orc.demosaic_full(pat, buf), _full(pat, buf), "demosaic::full %s" % pat[:4])


@pytest.mark.parametrize("pat,h,w,nh,nw", [("RGGB", 24, 30, 6, 7), (XTRANS, 36, 30, 9, 7), ("GRBG", 20, 22, 10, 11), ("2x8:" + L16, 40, 30, 10, 7),
                                           ("8x2:" + L16, 24, 40, 8, 13)])
def test_scaled_demosaic_against_a_second_restatement(orc, pat, h, w, nh, nw):
    buf = util.uniform_f32(util.SEED + 950 + h, h * w, -0.05, 1.0).reshape(h, w)
    want = _transform_buffer(buf.ravel(), w, h, (0, 0), (w - 1, 0), (0, h - 1), nw, nh, 4, pat).reshape(nh, nw, 4)
    util.assert_bits_equal(orc.scaled_demosaic(pat, buf, nw, nh), want, "scaled_demosaic %s" % pat[:4])


def _random_cfa(rng):
    """a random colour filter: 2x2, 6x6, 12x12 by length, or a stated WxH tile with W and H dividing 48 (16 letters and more); three
    colours mostly, a fourth sometimes; every colour the draw uses appears at least once"""
    shapes = [(2, 2, ""), (6, 6, ""), (12, 12, ""), (2, 8, "2x8:"), (8, 2, "8x2:"), (4, 4, "4x4:"),
              (24, 2, "24x2:"), (16, 3, "16x3:"), (48, 1, "48x1:"), (3, 16, "3x16:"), (1, 48, "1x48:"), (12, 4, "12x4:")]    # wide and tall tiles too
    wide, high, prefix = shapes[int(rng.integers(0, len(shapes)))]
    ncol = 4 if rng.random() < 0.25 else 3
    letters = [int(v) for v in rng.integers(0, ncol, wide * high)]
    for c in range(ncol):
        letters[int(rng.integers(0, wide * high)) if c not in letters else letters.index(c)] = c
    for c in range(ncol):                                  # the repair above can overwrite a colour's only cell: place the missing ones on distinct cells
        if c not in letters:
            cells = [i for i in range(wide * high) if letters.count(letters[i]) > 1]
            letters[cells[int(rng.integers(0, len(cells)))]] = c
    return prefix + "".join("RGBE"[c] for c in letters)


@pytest.mark.parametrize("seed", range(40))
def test_random_filters_odd_sizes_and_scales_against_a_second_restatement(orc, seed):
    """fuzz over what the fixed cases leave out: RANDOM pattern strings (not only the shipped filters), odd frame sizes, non-integer scales"""
    rng = np.random.default_rng(0x5EC0ED + seed)
    pat = _random_cfa(rng)
    h, w = int(rng.integers(10, 30)), int(rng.integers(10, 30))
    buf = util.uniform_f32(util.SEED + 2000 + seed, h * w, -0.05, 1.05).reshape(h, w)
    buf[int(rng.integers(0, h)), int(rng.integers(0, w))] = np.float32(-0.0)
    util.assert_bits_equal(orc.demosaic_full(pat, buf), _full(pat, buf), "demosaic::full %s %dx%d" % (pat, w, h))
    nh, nw = int(rng.integers(2, max(3, h // 2))), int(rng.integers(2, max(3, w // 2)))     # scales between 2 and ~15, rarely whole numbers
    want = _transform_buffer(buf.ravel(), w, h, (0, 0), (w - 1, 0), (0, h - 1), nw, nh, 4, pat).reshape(nh, nw, 4)
    util.assert_bits_equal(orc.scaled_demosaic(pat, buf, nw, nh), want, "scaled_demosaic %s %dx%d -> %dx%d" % (pat, w, h, nw, nh))


def test_transform_buffer_rotated_against_a_second_restatement(orc):
    """three components, corners of a rotated crop (negative skips, windows clamped at the frame edge)"""
    h, w = 20, 26
    src = util.uniform_f32(util.SEED + 990, h * w * 3, 0.0, 1.0)
    for tl, tr, bl, nw, nh in [((2, 1), (22, 4), (0, 17), 9, 7), ((24, 2), (3, 1), (23, 18), 8, 6)]:
        want = _transform_buffer(src, w, h, tl, tr, bl, nw, nh, 3)
        got = orc.transform_buffer(src.reshape(h, w, 3), w, h, tl, tr, bl, nw, nh, 3)
        util.assert_bits_equal(np.asarray(got).ravel(), want, "transform_buffer %r" % (tl,))


# ---- the colour path: TransformLookup tables, to_lab, basecurve, from_lab, gamma, quantisation ----------------------------------
# Plain Python, one numpy float32 scalar operation per Rust operation, Rust's left-to-right association; cbrtf / powf / exp2f are the
# platform libm's through ctypes -- what Rust's f32::cbrt / powf / exp2 call -- NOT numpy's vectorised versions.
import ctypes
import ctypes.util
import math

_libm = ctypes.CDLL(ctypes.util.find_library("m"))
for _n in ("cbrtf", "exp2f"):
    getattr(_libm, _n).restype = ctypes.c_float; getattr(_libm, _n).argtypes = [ctypes.c_float]
_libm.powf.restype = ctypes.c_float; _libm.powf.argtypes = [ctypes.c_float, ctypes.c_float]
cbrtf = lambda v: F(_libm.cbrtf(ctypes.c_float(float(v))))
powf = lambda v, e: F(_libm.powf(ctypes.c_float(float(v)), ctypes.c_float(float(e))))
exp2f = lambda v: F(_libm.exp2f(ctypes.c_float(float(v))))


def _rmin(a, b):    # f32::min: a NaN operand is ignored
    return b if np.isnan(a) else (a if np.isnan(b) else (a if a < b else b))


def _rmax(a, b):
    return b if np.isnan(a) else (a if np.isnan(b) else (a if a > b else b))


def _f_lab(v):      # color_conversions.rs:120-124
    e = F(F(216.0) / F(24389.0)); k = F(F(24389.0) / F(27.0))
    return cbrtf(v) if v > e else F(F(F(k * v) + F(16.0)) / F(116.0))


def _f_gamma(v):    # :134-140
    if v < F(0.0031308):
        return F(v * F(12.92))
    return F(F(F(1.055) * powf(v, F(F(1.0) / F(2.4)))) - F(0.055))


def _f_gamma_rev(v):  # :126-132
    if v < F(0.04045):
        return F(v / F(12.92))
    return powf(F(F(v + F(0.055)) / F(1.055)), F(2.4))


class _Lookup:      # color_conversions.rs:79-115
    def __init__(self, fn):
        self.max = 8191; self.fn = fn
        self.table = [fn(F(F(i) / F(self.max))) for i in range(self.max + 2)]

    def lookup(self, val):
        if val < 0.0 or val > 1.0:
            return self.fn(val)
        pos = F(val * F(self.max))
        key = 0 if np.isnan(pos) else int(pos)
        a = F(pos - np.trunc(pos))
        v1, v2 = self.table[key], self.table[key + 1]
        return F(v1 + F(a * F(v2 - v1)))


_TABLES = {}


def _tab(name):
    if name not in _TABLES:
        _TABLES[name] = _Lookup({"lab": _f_lab, "gamma": _f_gamma, "rev": _f_gamma_rev}[name])
    return _TABLES[name]


_SRGB = [[F(0.4124564), F(0.3575761), F(0.1804375)], [F(0.2126729), F(0.7151522), F(0.0721750)], [F(0.0193339), F(0.1191920), F(0.9503041)]]
_WHITE = (F(0.95047), F(1.000), F(1.08883))


def _inverse(m):    # color_conversions.rs:20-39
    det = F(F(F(m[0][0] * F(F(m[1][1] * m[2][2]) - F(m[2][1] * m[1][2]))) - F(m[0][1] * F(F(m[1][0] * m[2][2]) - F(m[1][2] * m[2][0])))) +
            F(m[0][2] * F(F(m[1][0] * m[2][1]) - F(m[1][1] * m[2][0]))))
    inv = F(F(1.0) / det)
    d = lambda a, b, c, e: F(F(a * b) - F(c * e))
    return [[F(d(m[1][1], m[2][2], m[2][1], m[1][2]) * inv), F(-d(m[0][1], m[2][2], m[0][2], m[2][1]) * inv), F(d(m[0][1], m[1][2], m[0][2], m[1][1]) * inv)],
            [F(-d(m[1][0], m[2][2], m[1][2], m[2][0]) * inv), F(d(m[0][0], m[2][2], m[0][2], m[2][0]) * inv), F(-d(m[0][0], m[1][2], m[1][0], m[0][2]) * inv)],
            [F(d(m[1][0], m[2][1], m[2][0], m[1][1]) * inv), F(-d(m[0][0], m[2][1], m[2][0], m[0][1]) * inv), F(d(m[0][0], m[1][1], m[1][0], m[0][1]) * inv)]]


def _normalize_wbs(v):   # ops/colorspaces.rs:12-27
    def normal(x):
        return np.isfinite(x) and x != 0 and abs(x) >= np.finfo(np.float32).tiny
    return [F(x / v[1]) if normal(x) else F(1.0) for x in v]


def _camera_to_lab(mul, cm, p):    # color_conversions.rs:42-55, 156-169
    r, g, b, e = [_rmin(F(p[i] * mul[i]), F(1.0)) for i in range(4)]
    x, y, z = [F(F(F(F(r * cm[i][0]) + F(g * cm[i][1])) + F(b * cm[i][2])) + F(e * cm[i][3])) for i in range(3)]
    xr, yr, zr = F(x / _WHITE[0]), F(y / _WHITE[1]), F(z / _WHITE[2])
    t = _tab("lab")
    fx, fy, fz = t.lookup(xr), t.lookup(yr), t.lookup(zr)
    l = F(F(F(116.0) * fy) - F(16.0)); a = F(F(500.0) * F(fx - fy)); bb = F(F(200.0) * F(fy - fz))
    return F(l / F(100.0)), F(F(a + F(127.0)) / F(255.0)), F(F(bb + F(127.0)) / F(255.0))


def _lab_to_rgb(m, l, a, b):       # :58-65, 172-191
    cl = F(l * F(100.0)); ca = F(F(a * F(255.0)) - F(127.0)); cb = F(F(b * F(255.0)) - F(127.0))
    fy = F(F(cl + F(16.0)) / F(116.0)); fx = F(F(ca / F(500.0)) + fy); fz = F(fy - F(cb / F(200.0)))
    e = F(F(216.0) / F(24389.0)); k = F(F(24389.0) / F(27.0))
    fx3 = F(F(fx * fx) * fx)
    xr = fx3 if fx3 > e else F(F(F(F(116.0) * fx) - F(16.0)) / k)
    yr = F(F(fy * fy) * fy) if cl > F(k * e) else F(cl / k)
    fz3 = F(F(fz * fz) * fz)
    zr = fz3 if fz3 > e else F(F(F(F(116.0) * fz) - F(16.0)) / k)
    x, y, z = F(xr * _WHITE[0]), F(yr * _WHITE[1]), F(zr * _WHITE[2])
    return [F(F(F(x * m[i][0]) + F(y * m[i][1])) + F(z * m[i][2])) for i in range(3)]


class _Spline:      # ops/curves.rs:68-157
    def __init__(self, p):
        pts = []
        if len(p) == 0 or (p[0][0] > 0.0 and p[0][1] > 0.0):
            pts.append((F(0.0), F(0.0)))
        pts += [(F(x), F(y)) for x, y in p]
        if len(p) == 0 or (p[-1][0] < 1.0 and p[-1][1] < 1.0):
            pts.append((F(1.0), F(1.0)))
        dxs = [F(pts[i + 1][0] - pts[i][0]) for i in range(len(pts) - 1)]
        dys = [F(pts[i + 1][1] - pts[i][1]) for i in range(len(pts) - 1)]
        sl = [F(dy / dx) for dx, dy in zip(dxs, dys)]
        c1 = [sl[0]]
        for i in range(len(dxs) - 1):
            m, nx = sl[i], sl[i + 1]
            if F(m * nx) <= 0.0:
                c1.append(F(0.0))
            else:
                common = F(dxs[i] + dxs[i + 1])
                c1.append(F(F(F(3.0) * common) / F(F(F(common + dxs[i + 1]) / m) + F(F(common + dxs[i]) / nx))))
        c1.append(sl[-1])
        c2, c3 = [], []
        for i in range(len(c1) - 1):
            inv = F(F(1.0) / dxs[i])
            common = F(F(F(c1[i] + c1[i + 1]) - sl[i]) - sl[i])
            c2.append(F(F(F(sl[i] - c1[i]) - common) * inv))
            c3.append(F(F(common * inv) * inv))
        self.pts, self.c1, self.c2, self.c3 = pts, c1, c2, c3

    def interpolate(self, val):
        if val >= self.pts[-1][0]:
            return self.pts[-1][1]
        if val <= self.pts[0][0]:
            return self.pts[0][1]
        low, high = 0, len(self.c3) - 1
        while low <= high:
            mid = (low + high) // 2
            xh = self.pts[mid][0]
            if xh < val:
                low = mid + 1
            elif xh > val:
                high = mid - 1
            else:
                return self.pts[mid][1]
        i = max(0, high)
        d = F(val - self.pts[i][0])
        return F(F(F(self.pts[i][1] + F(self.c1[i] * d)) + F(F(self.c2[i] * d) * d)) + F(F(F(self.c3[i] * d) * d) * d))


def test_lookup_tables_against_a_second_restatement(orc):
    """every entry of the three TransformLookup tables, built with the platform libm through ctypes"""
    import oracle
    for which, name in ((oracle.LUT_XYZ_LAB, "lab"), (oracle.LUT_SRGB_GAMMA, "gamma"), (oracle.LUT_SRGB_GAMMA_REVERSE, "rev")):
        util.assert_bits_equal(orc.lut_table(which), np.array(_tab(name).table, np.float32), "table " + name)
    vals = np.concatenate([util.uniform_f32(util.SEED + 70, 600, -0.2, 1.6), np.array([0.0, -0.0, 1.0, 0.5, np.nan, 2.0, 8.0, 1e-30, 1 - 2 ** -24], np.float32)])
    for which, name in ((oracle.LUT_XYZ_LAB, "lab"), (oracle.LUT_SRGB_GAMMA, "gamma")):
        util.assert_bits_equal(orc.lookup(which, vals), np.array([_tab(name).lookup(v) for v in vals], np.float32), "lookup " + name)


def test_xyz_d65_33_against_a_second_restatement(orc):
    util.assert_bits_equal(orc.const_xyz_d65_33(), np.array(_inverse(_SRGB), np.float32), "inverse(SRGB_D65_33)")


def test_tolab_fromlab_gamma_against_a_second_restatement(orc):
    n = 700
    px = util.uniform_f32(util.SEED + 71, n * 4, -0.06, 1.0).reshape(1, n, 4)
    px[0, :, 3] = 0.0
    px[0, :40, :3] *= np.float32(0.02)                       # dark tones: the linear branches of both Lab directions
    px[0, 40:60, 2] = 1.0                                      # clipped blue: ratios above 1, cbrtf
    wb = np.array([2.0, 1.0, 1.5, np.nan], np.float32)
    cam = util.cam_matrix()
    mul = _normalize_wbs([F(v) for v in wb])
    util.assert_bits_equal(orc.normalize_wbs(wb), np.array(mul, np.float32), "normalize_wbs")
    cm = [[F(v) for v in row] for row in cam]
    lab = np.array([_camera_to_lab(mul, cm, px[0, i]) for i in range(n)], np.float32).reshape(1, n, 3)
    util.assert_bits_equal(orc.tolab(px, wb, cam), lab, "to_lab")
    sp = _Spline([(0.5, 0.6)])
    cur = lab.copy()
    cur[0, :, 0] = [sp.interpolate(v) for v in lab[0, :, 0]]
    util.assert_bits_equal(orc.basecurve(lab, 0.0, [(0.5, 0.6)]), cur, "basecurve")
    m = _inverse(_SRGB)
    rgb = np.array([_lab_to_rgb(m, *cur[0, i]) for i in range(n)], np.float32).reshape(1, n, 3)
    util.assert_bits_equal(orc.fromlab(cur), rgb, "from_lab")
    g = _tab("gamma")
    out = np.array([g.lookup(_rmin(_rmax(v, F(0.0)), F(1.0))) for v in rgb.ravel()], np.float32).reshape(1, n, 3)
    util.assert_bits_equal(orc.gamma(rgb), out, "gamma")
    # quantisation, color_conversions.rs:323-330
    q8 = [int(_rmin(_rmax(F(v * F(256.0)), F(0.0)), F(255.0))) for v in out.ravel()]
    q16 = [int(_rmin(_rmax(F(math.copysign(math.floor(abs(float(F(v * F(65535.0)))) + 0.5), float(v))), F(0.0)), F(65535.0))) for v in rgb.ravel() if np.isfinite(v)]
    assert orc.output8bit(out.ravel()).tolist() == q8
    assert orc.output16bit(rgb.ravel()[np.isfinite(rgb.ravel())]).tolist() == q16


@pytest.mark.parametrize("points,exposure", [([(0.5, 0.6)], 0.0), ([], 0.5), ([(0.3, 0.2), (0.6, 0.8)], 0.0), ([(0.0, 0.1)], 0.0), ([(1.0, 0.9)], 0.2),
                                             ([(0.2, 0.3), (0.4, 0.35), (0.7, 0.9)], -0.3)])
def test_spline_against_a_second_restatement(orc, points, exposure):
    pts = [(F(x), F(F(y) * exp2f(F(exposure)))) for x, y in points]
    sp = _Spline(pts)
    vals = np.concatenate([util.uniform_f32(util.SEED + 72, 400, -0.1, 1.1), np.array([0.0, 1.0, 0.5, 0.3, 0.6, 0.2, 0.4, 0.7, -0.0], np.float32)])
    buf = np.zeros((1, vals.size, 3), np.float32); buf[0, :, 0] = vals; buf[0, :, 1] = 0.25; buf[0, :, 2] = 0.75
    want = buf.copy(); want[0, :, 0] = [sp.interpolate(v) for v in vals]
    util.assert_bits_equal(orc.basecurve(buf, exposure, points), want, "basecurve %r" % (points,))


# ---- white-balance helpers: temp_to_xyz / xyz_to_temp / OpToLab::set_temp / get_temp, host-only f64 + f32 maths -----------
import json
import os

_CIE = [(w, float(x), float(y), float(z)) for w, x, y, z in json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cie1931_2deg_5nm.json")))]


def _temp_to_xyz(temp):          # color_conversions.rs:276-292 (f64 throughout, f32 at the end)
    c1, c2 = 3.7417717905326694e-16, 0.014387773457709927
    xyz = [0.0, 0.0, 0.0]
    t = float(F(temp))
    for w, vx, vy, vz in _CIE:
        wl = float(w) / 1.0e9
        p5 = wl * wl; p5 = p5 * p5; p5 = p5 * wl                        # powi(5): square, square, times x
        power = c1 / (p5 * (math.exp(c2 / (t * wl)) - 1.0))
        xyz[0] += power * vx; xyz[1] += power * vy; xyz[2] += power * vz
    m = max(max(xyz[0], xyz[1]), xyz[2])
    return [F(v / m) for v in xyz]


def _xyz_to_temp(xyz):           # :294-310
    lo, hi = F(1000.0), F(40000.0)
    temp = F(0.0); new = [F(0.0)] * 3
    while F(hi - lo) > 1.0:
        temp = F(F(hi + lo) / F(2.0))
        new = _temp_to_xyz(temp)
        if F(new[2] / new[0]) > F(xyz[2] / xyz[0]):
            hi = temp
        else:
            lo = temp
    return temp, F(F(new[1] / new[0]) / F(xyz[1] / xyz[0]))


def test_white_balance_helpers_against_a_second_restatement(orc):
    for temp in (2000.0, 3200.0, 5003.5, 6504.0, 10000.0, 25000.0):
        util.assert_bits_equal(orc.temp_to_xyz(temp), np.array(_temp_to_xyz(temp), np.float32), "temp_to_xyz %g" % temp)
    for xyz in ([0.95047, 1.0, 1.08883], [1.0, 0.9, 0.4], [0.8, 1.0, 1.3], _temp_to_xyz(4321.0)):
        xyz = [F(v) for v in xyz]
        t, ti = _xyz_to_temp(xyz)
        got = orc.xyz_to_temp(np.array(xyz, np.float32))
        assert (np.float32(got[0]), np.float32(got[1])) == (t, ti), (xyz, got, t, ti)
    # OpToLab::set_temp / get_temp (ops/colorspaces.rs:59-85)
    xyz_to_cam = np.array([[0.9, -0.3, -0.1], [-0.4, 1.2, 0.2], [-0.1, 0.2, 0.7], [0.0, 0.0, 0.0]], np.float32)
    for temp, tint in ((5000.0, 1.0), (3000.0, 1.1), (7500.0, 0.93)):
        xyz = _temp_to_xyz(temp)
        xyz = [xyz[0], F(xyz[1] / F(tint)), xyz[2]]
        wb = []
        for i in range(4):
            acc = F(0.0)
            for j in range(3):
                acc = F(acc + F(xyz_to_cam[i][j] * xyz[j]))
            with np.errstate(all="ignore"):
                wb.append(F(F(1.0) / acc))                            # f32::recip; the E row is all zeros: 1/0 = inf -> not normal -> 1.0
        with np.errstate(all="ignore"):
            want = _normalize_wbs(wb)
        util.assert_bits_equal(orc.tolab_set_temp(xyz_to_cam, temp, tint), np.array(want, np.float32), "set_temp %g" % temp)
    cam_to_xyz = util.cam_matrix()
    for wbv in ([2.0, 1.0, 1.5, 0.0], [1.7, 1.0, 2.2, float("nan")], [2.4, 1.0, 1.2, 1.0]):
        acc = [F(0.0)] * 3
        for i in range(3):
            for j in range(4):
                if F(wbv[j]) > 0.0:
                    acc[i] = F(acc[i] + F(F(cam_to_xyz[i][j]) / F(wbv[j])))
        t, ti = _xyz_to_temp(acc)
        got = orc.tolab_get_temp(cam_to_xyz, np.array(wbv, np.float32))
        assert (np.float32(got[0]), np.float32(got[1])) == (t, ti), (wbv, got, t, ti)


def test_sixteen_letter_patterns_known_answers(orc):
    """Hand-derived known answers for a 16-letter filter under both shapes a caller may state (rawloader's own shape for 16 letters cannot be
    checked here, so the caller states it: "2x8:" / "8x2:").  Letters L16 = R G B G R B G G G B G R G R B G.
      2 wide x 8 high: tile rows  RG / BG / RB / GG / GB / GR / GR / BG        8 wide x 2 high: tile rows  RGBGRBGG / GBGRGRBG
    (1) colour of a pixel: (row 2, col 1) is B in the first reading (third tile row "RB"), G in the second (row 2 = tile row 0, col 1);
        (row 1, col 5) is G ("BG", col 5 % 2 = 1) against R ("GBGRGRBG"[5]).
    (2) demosaic::full of a mosaic whose every sample is 10 x (colour index + 1) -- 10 on R, 20 on G, 30 on B sites: a colour that has a tap in
        a pixel's 3x3 neighbourhood comes out as exactly its constant (mean of equal values), a colour without one stays 0.0 (demosaic.rs:110-114).
        At (row 3, col 0) -- a G site in both readings -- the 2x8 neighbourhood is  R B / G G / G B  (cols 0..1 and the wrapped col -1 = col 1 of
        the tile): R, G, B all present -> (10, 20, 30, 0).  In the 8x2 reading row 3 is tile row 1 and the neighbourhood of col 0 -- cols 7, 0, 1 of
        the tile rows 0, 1, 0 -- is  G R G / G G B / G R G: also all three.  At (row 4, col 3): 2x8 reads rows 3..5 = GG / GB / GR, cols 2..4 ->
        G G G / G B G / G R G -> own colour B = 30, G = 20, R = 10.  8x2 reads tile rows 1, 0, 1 at cols 2, 3, 4 -> G R G / B G R / G R G: the
        centre is G (own value 20), R present (10), B present (30).
    (3) a single bright sample: the filters disagree about which channel it lands in -- a 1000 at (row 2, col 1) is B (2x8) or G (8x2)."""
    two, eight = "2x8:" + L16, "8x2:" + L16
    (w2, pat2), (w8, pat8) = orc.cfa_pattern(two), orc.cfa_pattern(eight)
    assert (w2, w8) == (2, 8)                                         # cfa.width, which picks OpDemosaic's minscale arm (demosaic.rs:33-39)
    assert pat2[2, 1] == 2 and pat8[2, 1] == 1 and pat2[1, 5] == 1 and pat8[1, 5] == 0
    assert [int(v) for v in pat2[:8, :2].ravel()] == ["RGBE".index(c) for c in L16] and [int(v) for v in pat8[:2, :8].ravel()] == ["RGBE".index(c) for c in L16]
    h, w = 16, 16
    for name, pat in ((two, pat2), (eight, pat8)):
        mosaic = (10.0 * (pat[:h, :w] + 1)).astype(np.float32)
        out = orc.demosaic_full(name, mosaic)
        for (r, c) in ((3, 0), (4, 3), (8, 8), (5, 9)):
            window = pat[r - 1:r + 2, max(c - 1, 0):c + 2]
            want = [10.0 * (k + 1) if (window == k).any() else 0.0 for k in range(3)] + [0.0]
            assert out[r, c].tolist() == want, (name, r, c, out[r, c], want)
    assert orc.demosaic_full(two, (10.0 * (pat2[:h, :w] + 1)).astype(np.float32))[3, 0].tolist() == [10.0, 20.0, 30.0, 0.0]
    assert orc.demosaic_full(eight, (10.0 * (pat8[:h, :w] + 1)).astype(np.float32))[4, 3].tolist() == [10.0, 20.0, 30.0, 0.0]
    spike = np.zeros((h, w), np.float32); spike[2, 1] = 1000.0
    o2, o8 = orc.demosaic_full(two, spike), orc.demosaic_full(eight, spike)
    assert o2[2, 1].tolist() == [0.0, 0.0, 1000.0, 0.0] and o8[2, 1].tolist() == [0.0, 1000.0, 0.0, 0.0]
    # cropped_cfa(): shifting keeps the stated shape in the string; a square 4x4 reading needs the prefix too (16 letters are never guessed)
    assert orc.cfa_shift(two, 0, 0) == two and orc.cfa_shift(eight, 0, 0) == eight and orc.cfa_shift("2x2:RGGB", 1, 0) == "GRBG"
    assert orc.cfa_shift(two, 1, 0) == "2x8:" + "".join(L16[2 * r + (c + 1) % 2] for r in range(8) for c in range(2))
    assert orc.cfa_shift(eight, 0, 1) == "8x2:" + L16[8:] + L16[:8]
    with pytest.raises(Exception):
        orc.cfa_shift(L16, 0, 0)
    with pytest.raises(Exception):
        orc.cfa_shift("5x2:RGBGRGBGRG", 0, 0)                          # 5 does not divide 48
