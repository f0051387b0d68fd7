"""The host-pointer half of the C ABI (ipk_host_*): what a Rust `impl ImageOp::run` that keeps its OpBuffers in host Vec<f32>s
binds.  Every entry point against the oracle, plain numpy buffers in and out, no torch on the data path."""
import ctypes as C

import numpy as np
import pytest

import util
from util import assert_bits_equal

pytestmark = pytest.mark.gpu
XT = "GGRGGBGGBGGRBRGRBGGGBGGRGGRGGBRBGBRG"


@pytest.fixture(scope="module")
def L():
    import imagepipe_amd
    imagepipe_amd.init(0)
    return imagepipe_amd.lib()


def P(a):
    return a.ctypes.data_as(C.c_void_p)


def fa(v):
    return (C.c_float * len(v))(*[float(x) for x in v])


def _rgbe(h, w, seed):
    v = util.uniform_f32(seed, h * w * 4, -0.05, 1.2).reshape(h, w, 4)
    v[..., 3] = 0.0
    return v


@pytest.mark.parametrize("is_float", [False, True])
def test_host_gofloat_and_demosaic(L, orc, is_float):
    oh, ow, x, y, w, h = 70, 300, 3, 2, 290, 60
    raw = util.noise_u16(util.SEED + 100, oh, ow)
    src = raw.astype(np.float32) if is_float else raw
    out = np.empty((h, w), np.float32)
    fn = L.ipk_host_gofloat_cfa_f32 if is_float else L.ipk_host_gofloat_cfa_u16
    assert fn(P(src), ow, oh, x, y, w, h, util.BLACK, util.WHITE, P(out)) == 0, L.ipk_last_error()
    want = orc.gofloat_cfa(src, x, y, w, h, util.BLACK, util.WHITE)
    assert_bits_equal(out, want, "host gofloat")
    for cfa in ("GRBG", XT, "RGBE"):
        d4 = np.empty((h, w, 4), np.float32)
        assert L.ipk_host_demosaic_full(P(out), w, h, cfa.encode(), P(d4)) == 0, L.ipk_last_error()
        assert_bits_equal(d4, orc.demosaic_full(cfa, want), "host demosaic " + cfa)


def test_host_transform_buffer(L, orc):
    h, w, nh, nw = 64, 96, 16, 24
    buf = util.uniform_f32(util.SEED + 101, h * w).reshape(h, w)
    out = np.empty((nh, nw, 4), np.float32)
    assert L.ipk_host_transform_buffer_f32(P(buf), w, h, 0, 0, w - 1, 0, 0, h - 1, nw, nh, 4, XT.encode(), P(out)) == 0, L.ipk_last_error()
    assert_bits_equal(out, orc.scaled_demosaic(XT, buf, nw, nh), "host scaled demosaic")
    b4 = _rgbe(h, w, util.SEED + 102)
    out = np.empty((nh, nw, 4), np.float32)
    assert L.ipk_host_transform_buffer_f32(P(b4), w, h, 0, 0, w - 1, 0, 0, h - 1, nw, nh, 4, None, P(out)) == 0
    assert_bits_equal(out, orc.scale_down_opbuf(b4, nw, nh), "host scale_down_opbuf")


def test_host_pointwise_stages(L, orc):
    h, w = 40, 101
    b4 = _rgbe(h, w, util.SEED + 103)
    lab = np.empty((h, w, 3), np.float32)
    assert L.ipk_host_tolab(P(b4), w, h, 0, fa(util.WB), fa(util.cam_matrix().ravel()), P(lab)) == 0, L.ipk_last_error()
    want = orc.tolab(b4, util.WB, util.cam_matrix())
    assert_bits_equal(lab, want, "host tolab")
    cur = np.empty_like(lab)
    rc = L.ipk_host_basecurve(P(lab), w, h, 0.2, fa([0.5, 0.6]), 1, P(cur))
    assert rc == 0
    want = orc.basecurve(want, 0.2, [(0.5, 0.6)])
    assert_bits_equal(cur, want, "host basecurve")
    assert L.ipk_host_basecurve(P(lab), w, h, 0.0, fa([0.0, 0.0]), 0, P(cur)) == 1          # IPK_NOOP: the caller keeps its input Arc
    rgb = np.empty_like(lab)
    assert L.ipk_host_fromlab(P(want), w, h, P(rgb)) == 0
    want = orc.fromlab(want)
    assert_bits_equal(rgb, want, "host fromlab")
    g = np.empty_like(rgb)
    assert L.ipk_host_gamma(P(want), w, h, 3, 0, P(g)) == 0
    assert_bits_equal(g, orc.gamma(want), "host gamma")
    assert L.ipk_host_gamma(P(want), w, h, 3, 1, P(g)) == 1                                   # linear: no-op
    o8 = np.empty(g.size, np.uint8); o16 = np.empty(g.size, np.uint16)
    gw = orc.gamma(want)
    assert L.ipk_host_output8bit(P(gw), gw.size, P(o8)) == 0 and L.ipk_host_output16bit(P(gw), gw.size, P(o16)) == 0
    assert np.array_equal(o8, orc.output8bit(gw).ravel()) and np.array_equal(o16, orc.output16bit(gw).ravel())
    for orientation in range(8):
        out = np.empty(gw.size, np.float32); ow_, oh_ = C.c_size_t(), C.c_size_t()
        rc = L.ipk_host_rotate_buffer(P(gw), w, h, orientation, P(out), C.byref(ow_), C.byref(oh_))
        assert rc in (0, 1)
        wantr = orc.rotate_buffer(gw, orientation)
        if rc == 0:
            assert (oh_.value, ow_.value) == wantr.shape[:2]
            assert_bits_equal(out.reshape(wantr.shape), wantr, "host rotate %d" % orientation)


@pytest.mark.parametrize("cfa,maxwidth,out_type", [("RGGB", 0, 0), ("BGGR", 0, 1), ("RGGB", 0, 2), (XT, 0, 0), ("RGGB", 100, 0), (XT, 64, 1), ("RGBE", 0, 0)])
def test_host_pipeline_run_raw(L, orc, cfa, maxwidth, out_type):
    """Pipeline::run / output_8bit / output_16bit on host buffers: upload, kernels, download, synchronous"""
    import imagepipe_amd as ipa
    h, w = 120, 384
    raw = util.noise_u16(util.SEED + 104, h, w)
    img = ipa.RawImage(width=w, height=h, data=None, cfa=cfa, blacklevels=[util.BLACK] * 4, whitelevels=[util.WHITE] * 4,
                       wb_coeffs=util.WB, cam_to_xyz_normalized=util.cam_matrix())
    pipe = ipa.Pipeline(img)
    pipe.globals.settings.maxwidth = maxwidth
    d = pipe.desc()
    _, (fw, fh) = pipe.sizes()
    out = np.empty(fw * fh * 3, {0: np.float32, 1: np.uint8, 2: np.uint16}[out_type])
    used = C.c_int(-1)
    assert L.ipk_host_pipeline_run(C.byref(d), P(raw), P(out), out_type, C.byref(used)) == 0, L.ipk_last_error()
    od = orc.make_pipeline(raw, cfa=cfa, blacklevels=[util.BLACK] * 4, whitelevels=[util.WHITE] * 4, wb_coeffs=util.WB,
                           cam_to_xyz_normalized=util.cam_matrix(), maxwidth=maxwidth)
    want = [orc.pipeline_run, orc.pipeline_output_8bit, orc.pipeline_output_16bit][out_type](od)
    assert want.shape[:2] == (fh, fw)
    if out_type == 0:
        assert_bits_equal(out.reshape(fh, fw, 3), want, "host pipeline")
    else:
        assert np.array_equal(out.reshape(fh, fw, 3), want)
    assert used.value == (1 if (maxwidth == 0 and cfa != "RGBE") else 0)


@pytest.mark.parametrize("bits,fast,maxwidth", [(8, True, 0), (8, False, 0), (16, True, 50), (16, False, 50), (8, True, 33)])
def test_host_pipeline_run_raster(L, orc, bits, fast, maxwidth):
    import imagepipe_amd as ipa
    h, w = 90, 130
    rng = np.random.default_rng(9)
    img = rng.integers(0, 256 if bits == 8 else 65536, (h, w, 3)).astype(np.uint8 if bits == 8 else np.uint16)
    pipe = ipa.Pipeline(ipa.OtherImage(w, h, None, bits=bits))
    pipe.globals.settings.maxwidth = maxwidth; pipe.globals.settings.use_fastpath = fast
    d = pipe.desc()
    _, (fw, fh) = pipe.sizes()
    for out_type, dt, fn in ((1, np.uint8, orc.pipeline_output_8bit), (2, np.uint16, orc.pipeline_output_16bit)):
        out = np.empty(fw * fh * 3, dt)
        assert L.ipk_host_pipeline_run(C.byref(d), P(img), P(out), out_type, None) == 0, L.ipk_last_error()
        want = fn(orc.make_pipeline(img, maxwidth=maxwidth, use_fastpath=fast))
        assert np.array_equal(out.reshape(fh, fw, 3), want), (bits, fast, maxwidth, out_type)


def test_host_raw_to_srgb(L, orc):
    from imagepipe_amd._lib import FusedParams
    h, w, ow, x, y = 50, 300, 310, 4, 3
    raw = util.noise_u16(util.SEED + 105, h + 5, ow)
    p = FusedParams()
    p.src_type = 0; p.owidth = ow; p.x = x; p.y = y; p.width = w; p.height = h
    p.black0 = util.BLACK; p.white0 = util.WHITE; p.cfa = b"GBRG"
    p.wb_coeffs[:] = list(util.WB); p.cam_to_xyz_normalized[:] = [float(v) for v in util.cam_matrix().ravel()]
    p.exposure = 0.0; p.npoints = 1; p.points[0] = 0.5; p.points[1] = 0.6; p.linear = 0; p.out_type = 0
    out = np.empty((h, w, 3), np.float32)
    assert L.ipk_host_raw_to_srgb(C.byref(p), P(raw), P(out)) == 0, L.ipk_last_error()
    want = orc.pipeline_run(orc.make_pipeline(raw, cfa="GBRG", crops=(y, ow - x - w, raw.shape[0] - y - h, x), blacklevels=[util.BLACK] * 4,
                                              whitelevels=[util.WHITE] * 4, wb_coeffs=util.WB, cam_to_xyz_normalized=util.cam_matrix()))
    assert_bits_equal(out, want, "host raw_to_srgb")


@pytest.mark.parametrize("cfa,maxwidth,out_type,pinned", [("RGGB", 0, 0, True), ("RGGB", 0, 1, False), (XT, 0, 2, True), ("GBRG", 90, 0, True), ("RGBE", 0, 1, False)])
def test_host_pipeline_run_batch(L, orc, cfa, maxwidth, out_type, pinned):
    """A batch of host frames through the three-stream driver (upload / compute / download over two device slots): every frame
    equals the oracle, with page-locked (ipk_host_alloc) and with pageable buffers, for more frames than slots"""
    import imagepipe_amd as ipa
    h, w, n = 100, 320, 5
    frames = [util.noise_u16(util.SEED + 300 + i, h, w) for i in range(n)]
    img = ipa.RawImage(width=w, height=h, data=None, cfa=cfa, blacklevels=[util.BLACK] * 4, whitelevels=[util.WHITE] * 4,
                       wb_coeffs=util.WB, cam_to_xyz_normalized=util.cam_matrix())
    pipe = ipa.Pipeline(img)
    pipe.globals.settings.maxwidth = maxwidth
    d = pipe.desc()
    _, (fw, fh) = pipe.sizes()
    dt = {0: np.float32, 1: np.uint8, 2: np.uint16}[out_type]
    out_bytes = fw * fh * 3 * np.dtype(dt).itemsize
    if pinned:
        L.ipk_host_alloc.restype = C.c_void_p
        sp = [L.ipk_host_alloc(h * w * 2) for _ in range(n)]
        dp = [L.ipk_host_alloc(out_bytes) for _ in range(n)]
        assert all(sp) and all(dp)
        for p, f in zip(sp, frames):
            C.memmove(p, f.ctypes.data, f.nbytes)
        outs = [np.frombuffer((C.c_char * out_bytes).from_address(p), dtype=dt) for p in dp]
    else:
        outs = [np.empty(fw * fh * 3, dt) for _ in range(n)]
        sp = [f.ctypes.data for f in frames]
        dp = [o.ctypes.data for o in outs]
    used = C.c_int(-1)
    srcs = (C.c_void_p * n)(*sp); dsts = (C.c_void_p * n)(*dp)
    assert L.ipk_host_pipeline_run_batch(C.byref(d), srcs, dsts, n, out_type, C.byref(used)) == 0, L.ipk_last_error()
    for i in range(n):
        od = orc.make_pipeline(frames[i], cfa=cfa, blacklevels=[util.BLACK] * 4, whitelevels=[util.WHITE] * 4, wb_coeffs=util.WB,
                               cam_to_xyz_normalized=util.cam_matrix(), maxwidth=maxwidth)
        want = [orc.pipeline_run, orc.pipeline_output_8bit, orc.pipeline_output_16bit][out_type](od)
        if out_type == 0:
            assert_bits_equal(outs[i].reshape(fh, fw, 3), want, "batch frame %d" % i)
        else:
            assert np.array_equal(outs[i].reshape(fh, fw, 3), want), i
    assert L.ipk_host_pipeline_run_batch(C.byref(d), srcs, dsts, 0, out_type, None) == 0          # empty batch
    bad = (C.c_void_p * n)(*([sp[0], None] + sp[2:]))
    assert L.ipk_host_pipeline_run_batch(C.byref(d), bad, dsts, n, out_type, None) == -2         # IPK_ERR_INVALID, nothing computed
    if pinned:
        del outs
        for p in sp + dp:
            L.ipk_host_free(p)


def test_host_pipeline_from_another_thread(L, orc):
    """HIP's current device is per host thread: an entry point called from a thread other than ipk_init's (a Rayon worker in the
    reference's host code) binds the context's device itself"""
    import threading
    import imagepipe_amd as ipa
    h, w = 64, 256
    raw = util.noise_u16(util.SEED + 310, h, w)
    img = ipa.RawImage(width=w, height=h, data=None, cfa="RGGB", blacklevels=[util.BLACK] * 4, whitelevels=[util.WHITE] * 4,
                       wb_coeffs=util.WB, cam_to_xyz_normalized=util.cam_matrix())
    d = ipa.Pipeline(img).desc()
    out = np.empty(h * w * 3, np.float32)
    res = {}

    def work():
        res["rc"] = L.ipk_host_pipeline_run(C.byref(d), P(raw), P(out), 0, None)
    t = threading.Thread(target=work); t.start(); t.join()
    assert res["rc"] == 0, L.ipk_last_error()
    want = orc.pipeline_run(orc.make_pipeline(raw, cfa="RGGB", blacklevels=[util.BLACK] * 4, whitelevels=[util.WHITE] * 4, wb_coeffs=util.WB,
                                              cam_to_xyz_normalized=util.cam_matrix()))
    assert_bits_equal(out.reshape(h, w, 3), want, "host pipeline from a worker thread")
