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
    side = {4: 2, 36: 6, 144: 12}[len(pat)]
    tile = [["RGBE".index(c) for c in pat[r * side:(r + 1) * side]] for r in range(side)]
    return lambda row, col: tile[row % side][col % side]


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


@pytest.mark.parametrize("pat,h,w", [("RGGB", 11, 13), ("GBRG", 10, 12), ("RGBE", 10, 11), (XTRANS, 13, 14)])
def test_demosaic_full_against_a_second_restatement(orc, pat, h, w):
    buf = util.uniform_f32(util.SEED + 900 + h, h * w, -0.05, 1.0).reshape(h, w)
    buf[2, 3] = np.float32(-0.0); buf[5, 5] = 0.0
    util.assert_bits_equal(orc.demosaic_full(pat, buf), _full(pat, buf), "demosaic::full %s" % pat[:4])


@pytest.mark.parametrize("pat,h,w,nh,nw", [("RGGB", 24, 30, 6, 7), (XTRANS, 36, 30, 9, 7), ("GRBG", 20, 22, 10, 11)])
def test_scaled_demosaic_against_a_second_restatement(orc, pat, h, w, nh, nw):
    buf = util.uniform_f32(util.SEED + 950 + h, h * w, -0.05, 1.0).reshape(h, w)
    want = _transform_buffer(buf.ravel(), w, h, (0, 0), (w - 1, 0), (0, h - 1), nw, nh, 4, pat).reshape(nh, nw, 4)
    util.assert_bits_equal(orc.scaled_demosaic(pat, buf, nw, nh), want, "scaled_demosaic %s" % pat[:4])


def test_transform_buffer_rotated_against_a_second_restatement(orc):
    """three components, corners of a rotated crop (negative skips, windows clamped at the frame edge)"""
    h, w = 20, 26
    src = util.uniform_f32(util.SEED + 990, h * w * 3, 0.0, 1.0)
    for tl, tr, bl, nw, nh in [((2, 1), (22, 4), (0, 17), 9, 7), ((24, 2), (3, 1), (23, 18), 8, 6)]:
        want = _transform_buffer(src, w, h, tl, tr, bl, nw, nh, 3)
        got = orc.transform_buffer(src.reshape(h, w, 3), w, h, tl, tr, bl, nw, nh, 3)
        util.assert_bits_equal(np.asarray(got).ravel(), want, "transform_buffer %r" % (tl,))
