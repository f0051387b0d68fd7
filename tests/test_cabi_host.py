"""CPU-only checks of the product library: it loads, exports every symbol include/imagepipe_amd.h declares, its
host-side maths (sizes, curves, tables, CFA, orientation) agrees with the oracle, and every compute entry point
refuses to run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    from imagepipe_amd import _lib
    return _lib.load()


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "imagepipe_amd.h")).read()
    return sorted(set(re.findall(r"IPK_API\s+[\w\s\*]+?\b(ipk_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol(L):
    from imagepipe_amd import _lib
    syms = _header_symbols()
    assert len(syms) >= 60
    for s in syms:
        assert hasattr(L, s), "libimagepipe_amd.so does not export " + s
    assert sorted(_lib.SIGNATURES) == syms, "ctypes table and header disagree"


def test_compute_entry_points_fail_loudly_without_gpu(L):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert L.ipk_init(0) == -4                                  # IPK_ERR_NO_DEVICE
    assert b"no HIP device" in L.ipk_last_error()
    assert L.ipk_is_initialized() == 0
    buf = (C.c_float * 64)()
    assert L.ipk_fromlab(buf, 4, 4, buf, None) == -1            # IPK_ERR_NOT_INIT
    assert L.ipk_host_fromlab(buf, 4, 4, buf) == -1
    assert L.ipk_host_gamma(buf, 4, 4, 3, 0, buf) == -1
    assert b"no CPU fallback" in L.ipk_last_error()
    import imagepipe_amd
    with pytest.raises(imagepipe_amd.IpkError):
        imagepipe_amd.init()


def test_lut_tables_match_oracle_and_fixture(L, orc):
    for which in range(3):
        t = np.empty(8193, np.float32)
        assert L.ipk_lut_table(which, t.ctypes.data_as(C.POINTER(C.c_float))) == 0
        assert np.array_equal(t.view(np.uint32), orc.lut_table(which).view(np.uint32))


def test_size_image_and_scaling(L, orc):
    rng = np.random.default_rng(1)
    out = (C.c_size_t * 4)()
    for _ in range(300):
        ow, oh = int(rng.integers(10, 300)), int(rng.integers(10, 300))
        cr = [int(v) for v in rng.integers(0, 200, 4)]
        assert L.ipk_size_image(*cr, ow, oh, out) == 0
        assert tuple(out) == orc.size_image(*cr, ow, oh)
    assert L.ipk_size_image(0, 0, 0, 0, 9, 100, out) == -2
    s, nw, nh = C.c_float(), C.c_size_t(), C.c_size_t()
    for _ in range(2000):
        w, h = int(rng.integers(1, 12000)), int(rng.integers(1, 12000))
        mw, mh = int(rng.integers(0, 3) and rng.integers(1, 9000)), int(rng.integers(0, 3) and rng.integers(1, 9000))
        L.ipk_calculate_scaling_total(w, h, mw, mh, C.byref(s), C.byref(nw), C.byref(nh))
        es, ew, eh = orc.calculate_scaling_total(w, h, mw, mh)
        assert (np.float32(s.value), nw.value, nh.value) == (np.float32(es), ew, eh)


def test_normalize_wbs(L, orc):
    for wb in [(2.0, 1.0, 1.5, np.nan), (0.0, 2.0, np.inf, 1e-40), (1.9, 0.93, 1.4, 0.0), (-1, -2, 3, 4), (1, 0, 1, 1)]:
        a = (C.c_float * 4)(*wb); o = (C.c_float * 4)()
        L.ipk_normalize_wbs(a, o)
        assert np.array_equal(np.array(o[:], np.float32).view(np.uint32), orc.normalize_wbs(wb).view(np.uint32))


def test_spline_new(L, orc):
    for pts in [[], [(0.5, 0.6)], [(0.0, 0.2)], [(1.0, 0.8)], [(0.25, 0.2), (0.5, 0.6), (0.75, 0.8)], [(0.7, 0.2), (0.3, 0.9)],
                [(0.1, 0.05), (0.2, 0.3), (0.3, 0.25), (0.6, 0.7), (0.9, 0.95)]]:
        p = np.asarray(pts, np.float32).ravel()
        n = p.size // 2
        arr = (C.c_float * max(2, p.size))(*p)
        outs = [(C.c_float * (n + 2))() for _ in range(5)]
        k = L.ipk_spline_new(arr, n, *outs)
        want = orc.spline_new(pts)
        assert k == want[0].size
        for got, w, m in zip(outs, want, [k, k, k, k - 1, k - 1]):
            assert np.array_equal(np.array(got[:m], np.float32).view(np.uint32), w.view(np.uint32))


def test_rotatecrop_calc_size(L, orc):
    rng = np.random.default_rng(2)
    ow, oh = C.c_size_t(), C.c_size_t()
    for _ in range(3000):
        p = [float(np.float32(v)) for v in (rng.uniform(0, 0.45, 4).tolist() + [rng.uniform(0, 1.1)])]
        if rng.integers(0, 4) == 0:
            p[4] = 0.0
        w, h = int(rng.integers(1, 9000)), int(rng.integers(1, 9000))
        ratio = float(np.float32(w) / np.float32(h))
        for rev in (0, 1):
            L.ipk_rotatecrop_calc_size((C.c_float * 5)(*p), ratio, w, h, rev, C.byref(ow), C.byref(oh))
            assert (ow.value, oh.value) == orc.rotatecrop_calc_size(p, w, h, bool(rev), ratio)


def test_cfa_shift_and_orientation(L, orc):
    out = C.create_string_buffer(200)
    xt = "GGRGGBGGBGGRBRGRBGGGBGGRGGRGGBRBGBRG"
    for pat in ["RGGB", "GBRG", xt, (xt[:6] * 2 + xt[6:12] * 2) * 6]:
        for x in range(7):
            for y in range(7):
                assert L.ipk_cfa_shift(pat.encode(), x, y, out) == 0
                assert out.value.decode() == orc.cfa_shift(pat, x, y)
    assert orc.cfa_shift("RGGB", 1, 0) == "GRBG" and orc.cfa_shift("RGGB", 0, 1) == "GBRG" and orc.cfa_shift("RGGB", 1, 1) == "BGGR"
    assert L.ipk_cfa_shift(b"RGXB", 0, 0, out) == -2
    # 16 letters: the tile shape (8x2 or 2x8) is unverified against rawloader -- refused by product and oracle alike unless the caller states it
    assert L.ipk_cfa_shift(b"RGGBGRBGGBRGBGGR", 0, 0, out) == -5            # IPK_ERR_UNSUPPORTED
    assert b"16-letter" in L.ipk_last_error()
    with pytest.raises(Exception):
        orc.cfa_shift("RGGBGRBGGBRGBGGR", 0, 0)
    for pat in ["2x8:RGBGRBGGGBGRGRBG", "8x2:RGBGRBGGGBGRGRBG", "4x4:RGBGRBGGGBGRGRBG", "2x2:RGGB", "6x6:" + xt, "2x4:RGGBRGBG", "16x1:RGBGRBGGGBGRGRBG"]:
        for x in range(9):
            for y in range(9):
                assert L.ipk_cfa_shift(pat.encode(), x, y, out) == 0, pat
                assert out.value.decode() == orc.cfa_shift(pat, x, y)
    for bad in [b"5x2:RGBGRGBGRG", b"2x8:RGGB", b"x8:RGGBGRBGGBRGBGGR", b"2x:RGGB", b"2x8RGGB", b"0x4:", b"2x8:RGBGRBGGGBGRGRBX"]:
        assert L.ipk_cfa_shift(bad, 0, 0, out) == -2, bad
    f = (C.c_int * 3)()
    for o in range(9):
        L.ipk_orientation_to_flips(o, f)
        assert tuple(bool(v) for v in f) == orc.orientation_to_flips(o)
    for rot in range(4):
        for fh in (0, 1):
            for fv in (0, 1):
                assert L.ipk_transform_orientation(rot, fh, fv) == orc.transform_orientation(rot, fh, fv)


def _desc(**kw):
    from imagepipe_amd._lib import PipelineDesc
    d = PipelineDesc()
    d.src_type = 2; d.width = 128; d.height = 64; d.cpp = 3
    for k, v in kw.items():
        if k == "rotatecrop":
            d.rotatecrop[:] = v
        else:
            setattr(d, k, v)
    return d


@pytest.mark.parametrize("kw,size", [({}, (128, 64)), (dict(maxwidth=128), (128, 64)), (dict(maxwidth=256), (128, 64)),
                                     (dict(maxwidth=64), (64, 32)), (dict(maxwidth=64, rotation=1), (64, 128)),
                                     (dict(maxwidth=32, rotation=1), (32, 64)), (dict(maxwidth=256, rotation=1), (64, 128)),
                                     (dict(maxwidth=64, crop_top=1, crop_bottom=1, crop_left=1, crop_right=1), (64, 31)),
                                     (dict(maxwidth=64, rotatecrop=[0.1, 0.1, 0.1, 0.1, 0.0]), (64, 32))])
def test_pipeline_sizes_maxsize_cases(L, kw, size):
    """tests/maxsize_test.rs:31-90 through the product's own size negotiation"""
    d = _desc(**kw)
    a, b, c, e = C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_size_t()
    assert L.ipk_pipeline_sizes(C.byref(d), C.byref(a), C.byref(b), C.byref(c), C.byref(e)) == 0
    assert (c.value, e.value) == size


def test_pipeline_sizes_random_vs_oracle(L, orc):
    rng = np.random.default_rng(3)
    a, b, c, e = C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_size_t()
    for _ in range(1500):
        w, h = int(rng.integers(10, 7000)), int(rng.integers(10, 5000))
        kw = dict(width=w, height=h, maxwidth=int(rng.integers(0, 2) * rng.integers(1, 4000)), maxheight=int(rng.integers(0, 2) * rng.integers(1, 4000)),
                  rotation=int(rng.integers(0, 4)), crop_top=int(rng.integers(0, 50)), crop_left=int(rng.integers(0, 50)),
                  crop_right=int(rng.integers(0, 50)), crop_bottom=int(rng.integers(0, 50)))
        rc = [float(np.float32(v)) for v in rng.uniform(0, 0.3, 4)] + [float(np.float32(rng.uniform(0, 1.0)))]
        if rng.integers(0, 2):
            rc = [0.0] * 5
        d = _desc(rotatecrop=rc, **kw)
        assert L.ipk_pipeline_sizes(C.byref(d), C.byref(a), C.byref(b), C.byref(c), C.byref(e)) == 0
        od = orc.make_pipeline(np.zeros((h, w, 3), np.uint8), maxwidth=kw["maxwidth"], maxheight=kw["maxheight"], rotation=kw["rotation"],
                               crops=(kw["crop_top"], kw["crop_right"], kw["crop_bottom"], kw["crop_left"]), rotatecrop=rc)
        assert ((a.value, b.value), (c.value, e.value)) == orc.pipeline_sizes(od)


def test_wb_temperature_helpers_vs_oracle(L, orc):
    """temp_to_xyz / xyz_to_temp / OpToLab::set_temp / get_temp (host-only maths; no reference test pins them)"""
    fa = lambda v: (C.c_float * len(v))(*[float(x) for x in v])
    for temp in [1000.0, 2500.0, 3200.5, 5003.0, 6504.0, 9000.0, 25000.0, 40000.0]:
        o = (C.c_float * 3)(); L.ipk_temp_to_xyz(temp, o)
        want = orc.temp_to_xyz(temp)
        assert np.array_equal(np.array(o[:], np.float32).view(np.uint32), want.view(np.uint32))
        assert max(o[:]) == 1.0
        t, ti = C.c_float(), C.c_float()
        L.ipk_xyz_to_temp(fa(want), C.byref(t), C.byref(ti))
        assert (np.float32(t.value), np.float32(ti.value)) == tuple(np.float32(v) for v in orc.xyz_to_temp(want))
        assert abs(t.value - temp) <= 2.0 and abs(ti.value - 1.0) < 1e-3        # round trip of the bisection
    # independent anchor: the Planckian locus at 6500 K has CIE 1931 chromaticity (0.3135, 0.3237) (published value)
    p = orc.temp_to_xyz(6500.0).astype(np.float64)
    assert abs(p[0] / p.sum() - 0.3135) < 2e-4 and abs(p[1] / p.sum() - 0.3237) < 2e-4
    xyz_to_cam = np.array([[0.9, 0.1, -0.05], [-0.3, 1.2, 0.1], [0.05, -0.2, 1.1], [0.0, 0.0, 0.0]], np.float32)
    cam_to_xyz = np.array([[0.6, 0.3, 0.1, 0.0], [0.25, 0.7, 0.05, 0.0], [0.02, 0.1, 0.9, 0.0]], np.float32)
    for temp, tint in [(5000.0, 1.0), (3000.0, 1.1), (7500.0, 0.9)]:
        wb = (C.c_float * 4)(); L.ipk_tolab_set_temp(fa(xyz_to_cam.ravel()), temp, tint, wb)
        assert np.array_equal(np.array(wb[:], np.float32).view(np.uint32), orc.tolab_set_temp(xyz_to_cam, temp, tint).view(np.uint32))
        t, ti = C.c_float(), C.c_float()
        L.ipk_tolab_get_temp(fa(cam_to_xyz.ravel()), wb, C.byref(t), C.byref(ti))
        assert (np.float32(t.value), np.float32(ti.value)) == tuple(np.float32(v) for v in orc.tolab_get_temp(cam_to_xyz, np.array(wb[:], np.float32)))


def test_const_matrices_and_mirror_temp(L, orc):
    m = (C.c_float * 12)()
    L.ipk_const_matrix(0, m); assert np.array_equal(np.array(m[:9], np.float32), orc.const_srgb_d65_33().ravel())
    L.ipk_const_matrix(1, m); assert np.array_equal(np.array(m[:9], np.float32).view(np.uint32), orc.const_xyz_d65_33().ravel().view(np.uint32))
    L.ipk_const_matrix(2, m); assert np.array_equal(np.array(m[:], np.float32), orc.const_srgb_d65_43().ravel())
    L.ipk_const_matrix(3, m); assert np.array_equal(np.array(m[:9], np.float32), orc.const_xyz_d65_33().ravel()) and m[9:] == [0.0] * 3
    assert L.ipk_const_matrix(4, m) == -2
    import imagepipe_amd as ip
    op = ip.OpToLab(ip.OtherImage(16, 16, None))
    op.set_temp(5000.0, 1.0)
    want = orc.tolab_set_temp(np.vstack([orc.const_xyz_d65_33(), np.zeros((1, 3), np.float32)]), 5000.0, 1.0)
    assert np.array_equal(np.array(op.wb_coeffs, np.float32).view(np.uint32), want.view(np.uint32))
    t, ti = op.get_temp()
    assert abs(t - 5000.0) < 3.0 and abs(ti - 1.0) < 2e-3


def test_fastpath_decision(L):
    """Pipeline::default_ops for a raster source (pipeline.rs:286-288, :381-383): bitwise equality with PipelineOps::new(Other)"""
    def d(**kw):
        x = _desc(**{k: v for k, v in kw.items() if k not in ("wb_coeffs", "cam")}); x.use_fastpath = kw.get("use_fastpath", 1)
        m = (C.c_float * 12)(); L.ipk_const_matrix(2, m); x.cam_to_xyz_normalized[:] = m[:]
        x.wb_coeffs[:] = [1.0, 1.0, 1.0, 0.0]
        for k, v in kw.items():
            if k in ("wb_coeffs", "cam"):
                (x.wb_coeffs if k == "wb_coeffs" else x.cam_to_xyz_normalized)[:] = v
        return x
    for out_type, want in [(0, 0), (1, 1), (2, 1)]:
        assert L.ipk_pipeline_takes_fastpath(C.byref(d()), out_type) == want
    assert L.ipk_pipeline_takes_fastpath(C.byref(d(src_type=3)), 2) == 1          # RGB16
    for kw in [dict(use_fastpath=0), dict(src_type=0), dict(crop_left=1), dict(rotation=1), dict(fliph=1), dict(exposure=0.1), dict(exposure=-0.0),
               dict(rotatecrop=[0.0, 0.0, 0.0, 0.0, -0.0]), dict(rotatecrop=[0.1, 0.0, 0.0, 0.0, 0.0]), dict(wb_coeffs=[1.0, 1.0, 1.0, 1.0]), dict(is_cfa=1),
               dict(npoints=1)]:
        assert L.ipk_pipeline_takes_fastpath(C.byref(d(**kw)), 1) == 0, kw
    assert L.ipk_pipeline_takes_fastpath(C.byref(d(maxwidth=64, maxheight=10)), 1) == 1      # settings are not ops


def test_multi_gpu_and_timing_entry_points_without_a_gpu(L):
    """host-only parts of the round-2 entry points: band plans, argument checks, an empty timing session, the RCCL loader's failure
    mode and the libm report before ipk_init"""
    from imagepipe_amd import _lib
    bands = (_lib.Band * 4)()
    assert L.ipk_band_plan(100, 4, 2, bands) == 0
    assert [(b.out_row0, b.out_rows, b.src_row0, b.src_rows) for b in bands] == [(0, 26, 0, 27), (26, 26, 25, 28), (52, 24, 51, 26), (76, 24, 75, 25)]
    assert L.ipk_band_plan(0, 4, 2, bands) == -2 and L.ipk_band_plan(100, 0, 2, bands) == -2 and L.ipk_band_plan(100, 4, 2, None) == -2
    assert L.ipk_band_plan_scaled(5760, 1440, 4, bands) == 0
    assert sum(b.out_rows for b in bands) == 1440 and bands[0].src_row0 == 0 and bands[3].src_row0 + bands[3].src_rows == 5760
    assert all(bands[k].src_row0 + bands[k].src_rows >= bands[k + 1].src_row0 for k in range(3))      # neighbouring source ranges touch or overlap
    assert L.ipk_band_plan_scaled(5760, 1, 4, bands) == -2
    n = C.c_int(-1)
    assert L.ipk_timing_begin() == 0 and L.ipk_timing_end(None, 0, C.byref(n)) == 0 and n.value == 0
    assert L.ipk_timing_end(None, 0, None) == -2
    h = C.c_void_p()
    assert L.ipk_comm_init_host(0, 0, _lib.EXCHANGE_FN(lambda *a: 0), None, C.byref(h)) == -2
    assert L.ipk_comm_info(None, None, None, None) == -2 and L.ipk_comm_free(None) == 0
    import torch
    if not torch.cuda.is_available():
        idb = C.create_string_buffer(128)
        assert L.ipk_comm_init_rccl(idb.raw, 0, 1, C.byref(h)) == -1                                 # IPK_ERR_NOT_INIT: no device bound
        assert L.ipk_host_libm_matches(None) == -1


@pytest.mark.parametrize("nranks,w,h", [(1, 64, 37), (2, 64, 37), (3, 300, 50), (8, 300, 12)])
def test_multi_gpu_entry_points_from_a_plain_cpp_host(nranks, w, h):
    """tests/cpp/comm_test.cpp: ranks as threads, the host transport's callback a shared-memory mailbox, host slabs and frames -- the
    band plan, halo exchange and in-place gather through the C ABI with no Python in the data path (8 ranks on 12 rows: empty bands)"""
    import subprocess
    exe = os.path.join(ROOT, "tests", "cpp", "build", "comm_test")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp")])
    r = subprocess.run([exe, str(nranks), str(w), str(h)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "COMM_OK" in r.stdout, r.stdout + r.stderr
