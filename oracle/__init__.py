"""ctypes front-end of the CPU oracle (oracle/imagepipe_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never from the imagepipe_amd package (the product path has no CPU fallback).

Every wrapper mirrors one reference function; see the C file for the reference file:line of each.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "build", "liboracle.so")
_SRC = os.path.join(_HERE, "imagepipe_oracle.c")


def build(force=False):
    """Compile the oracle with gcc (recipe: oracle/Makefile)."""
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


_lib = None


def usable_cpus():
    """CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota.  OpenMP sizes its pool by
    the visible CPU count; under a quota (e.g. 16 CPUs' worth on a 256-thread host) that oversubscribes the row loops
    several hundred-fold in wall time, so the oracle pins its pool to this number."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]              # cgroup v2
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())            # cgroup v1
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, -(-quota // period)))
        except (OSError, ValueError):
            pass
    return n


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _declare(_lib)
        _lib.orc_set_num_threads(usable_cpus())
        _lib.orc_luts_init()
    return _lib


_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_u16p = np.ctypeslib.ndpointer(np.uint16, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_szp = C.POINTER(C.c_size_t)
_sz = C.c_size_t


class PipelineDesc(C.Structure):
    """Mirror of `orc_pipeline` (the fields of PipelineOps + PipelineSettings + the source)."""
    _fields_ = [
        ("source_kind", C.c_int), ("data", C.c_void_p),
        ("width", _sz), ("height", _sz),
        ("cpp", C.c_int), ("is_cfa", C.c_int), ("cfa", C.c_char * 160),
        ("crop_top", _sz), ("crop_right", _sz), ("crop_bottom", _sz), ("crop_left", _sz),
        ("blacklevels", C.c_float * 4), ("whitelevels", C.c_float * 4),
        ("rc", C.c_float * 5),
        ("cam_to_xyz_normalized", C.c_float * 12), ("wb_coeffs", C.c_float * 4),
        ("exposure", C.c_float), ("npoints", C.c_int), ("points", C.c_float * 128),
        ("rotation", C.c_int), ("fliph", C.c_int), ("flipv", C.c_int),
        ("maxwidth", _sz), ("maxheight", _sz), ("linear", C.c_int), ("use_fastpath", C.c_int),
    ]


def _declare(L):
    def sig(name, res, *args):
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = list(args)
    sig("orc_set_num_threads", None, C.c_int)
    sig("orc_get_max_threads", C.c_int)
    sig("orc_luts_init", None)
    sig("orc_lut_table", C.POINTER(C.c_float), C.c_int)
    sig("orc_lut_len", C.c_int)
    sig("orc_const_srgb_d65_33", None, _f32p)
    sig("orc_const_xyz_d65_33", None, _f32p)
    sig("orc_const_srgb_d65_43", None, _f32p)
    sig("orc_inverse33", None, _f32p, _f32p)
    sig("orc_lookup", None, C.c_int, _f32p, _f32p, _sz)
    sig("orc_input8bit", None, _u8p, _f32p, _sz)
    sig("orc_input16bit", None, _u16p, _f32p, _sz)
    sig("orc_output8bit", None, _f32p, _u8p, _sz)
    sig("orc_output16bit", None, _f32p, _u16p, _sz)
    sig("orc_xyz_to_lab", None, _f32p, _f32p, _sz)
    sig("orc_lab_to_xyz", None, _f32p, _f32p, _sz)
    sig("orc_camera_to_lab", None, _f32p, _f32p, _f32p, _f32p, _sz)
    sig("orc_lab_to_rgb", None, _f32p, _f32p, _f32p, _sz)
    sig("orc_cfa_shift", C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_char_p)
    sig("orc_cfa_pattern", C.c_int, C.c_char_p, _i32p)
    sig("orc_size_image", C.c_int, _sz, _sz, _sz, _sz, _sz, _sz, _szp)
    for k, p in (("u16", _u16p), ("f32", _f32p)):
        sig("orc_gofloat_cfa_" + k, None, p, _sz, _sz, _sz, _sz, _sz, C.c_float, C.c_float, _f32p)
        sig("orc_gofloat_mono_" + k, None, p, _sz, _sz, _sz, _sz, _sz, C.c_float, C.c_float, _f32p)
        sig("orc_gofloat_rgb_" + k, None, p, _sz, _sz, _sz, _sz, _sz, _f32p, _f32p, _f32p)
    sig("orc_gofloat_other_u8", None, _u8p, _sz, _sz, _sz, _sz, _sz, _f32p)
    sig("orc_gofloat_other_u16", None, _u16p, _sz, _sz, _sz, _sz, _sz, _f32p)
    sig("orc_demosaic_full", C.c_int, C.c_char_p, _f32p, _sz, _sz, _f32p)
    sig("orc_calculate_scaling_total", None, _sz, _sz, _sz, _sz, C.POINTER(C.c_float), _szp, _szp)
    for k, p in (("f32", _f32p), ("u8", _u8p), ("u16", _u16p)):
        sig("orc_transform_buffer_" + k, C.c_int, p, _sz, _sz,
            C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
            _sz, _sz, _sz, C.c_char_p, p)
    sig("orc_scaled_demosaic", C.c_int, C.c_char_p, _f32p, _sz, _sz, _sz, _sz, _f32p)
    sig("orc_scale_down_opbuf", C.c_int, _f32p, _sz, _sz, _sz, _sz, _f32p)
    sig("orc_scale_down_srgb", C.c_int, _u8p, _sz, _sz, _sz, _sz, _u8p)
    sig("orc_scale_down_srgb16", C.c_int, _u16p, _sz, _sz, _sz, _sz, _u16p)
    sig("orc_demosaic_run", C.c_int, C.c_char_p, _f32p, _sz, _sz, _sz, _sz, _sz, _f32p, _szp, _szp)
    sig("orc_normalize_wbs", None, _f32p, _f32p)
    sig("orc_tolab", None, _f32p, _sz, _sz, C.c_int, _f32p, _f32p, _f32p)
    sig("orc_fromlab", None, _f32p, _sz, _sz, _f32p)
    sig("orc_temp_to_xyz", None, C.c_float, _f32p)
    sig("orc_xyz_to_temp", None, _f32p, _f32p)
    sig("orc_tolab_set_temp", None, _f32p, C.c_float, C.c_float, _f32p)
    sig("orc_tolab_get_temp", None, _f32p, _f32p, _f32p)
    sig("orc_spline_new", C.c_int, _f32p, C.c_int, _f32p, _f32p, _f32p, _f32p, _f32p)
    sig("orc_spline_interpolate", C.c_int, _f32p, C.c_int, _f32p, _f32p, _sz)
    sig("orc_basecurve", C.c_int, _f32p, _sz, _sz, C.c_float, _f32p, C.c_int, _f32p)
    sig("orc_gamma", C.c_int, _f32p, _sz, _sz, _sz, C.c_int, _f32p)
    sig("orc_orientation_to_flips", None, C.c_int, C.POINTER(C.c_int))
    sig("orc_orientation_from_flips", C.c_int, C.c_int, C.c_int, C.c_int)
    sig("orc_transform_new", None, C.c_int, C.POINTER(C.c_int))
    sig("orc_transform_orientation", C.c_int, C.c_int, C.c_int, C.c_int)
    sig("orc_rotate_buffer", C.c_int, _f32p, _sz, _sz, C.c_int, _f32p, _szp, _szp)
    sig("orc_transform_forward", None, C.c_int, _sz, _sz, _szp, _szp)
    sig("orc_rotatecrop_calc_size", None, _f32p, C.c_float, _sz, _sz, C.c_int, _szp, _szp)
    sig("orc_rotatecrop_run", C.c_int, _f32p, _f32p, _sz, _sz, _sz, C.c_void_p, _szp, _szp)
    sig("orc_rotatecrop_corners", C.c_int, _f32p, _sz, _sz, _i64p, _szp, _szp)
    sig("orc_selftest_rotatecrop_roundtrip_transform", C.c_uint64)
    sig("orc_selftest_rotatecrop_roundtrip_rotation", C.c_uint64)
    sig("orc_pipeline_sizeof", _sz)
    sig("orc_pipeline_sizes", C.c_int, C.POINTER(PipelineDesc), _szp, _szp, _szp, _szp)
    sig("orc_pipeline_run", C.c_void_p, C.POINTER(PipelineDesc), _szp, _szp)
    sig("orc_pipeline_output_8bit", C.c_void_p, C.POINTER(PipelineDesc), _szp, _szp)
    sig("orc_pipeline_output_16bit", C.c_void_p, C.POINTER(PipelineDesc), _szp, _szp)
    sig("orc_free", None, C.c_void_p)
    assert L.orc_pipeline_sizeof() == C.sizeof(PipelineDesc), "orc_pipeline layout mismatch"


# Orientation / Rotation enums (same numbering as the C file and include/imagepipe_amd.h)
OR_NORMAL, OR_HFLIP, OR_ROT180, OR_VFLIP, OR_TRANSPOSE, OR_ROT90, OR_TRANSVERSE, OR_ROT270, OR_UNKNOWN = range(9)
ROT_NORMAL, ROT_90, ROT_180, ROT_270 = range(4)
LUT_XYZ_LAB, LUT_SRGB_GAMMA_REVERSE, LUT_SRGB_GAMMA = range(3)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def set_num_threads(n):
    lib().orc_set_num_threads(int(n))


def max_threads():
    return lib().orc_get_max_threads()


def lut_table(which):
    n = lib().orc_lut_len()
    p = lib().orc_lut_table(which)
    return np.ctypeslib.as_array(p, shape=(n,)).copy()


def const_srgb_d65_33():
    o = np.empty(9, np.float32); lib().orc_const_srgb_d65_33(o); return o.reshape(3, 3)


def const_xyz_d65_33():
    o = np.empty(9, np.float32); lib().orc_const_xyz_d65_33(o); return o.reshape(3, 3)


def const_srgb_d65_43():
    o = np.empty(12, np.float32); lib().orc_const_srgb_d65_43(o); return o.reshape(3, 4)


def lookup(which, vals):
    v = _f32(vals).ravel(); o = np.empty_like(v); lib().orc_lookup(which, v, o, v.size); return o.reshape(np.shape(vals))


def expand_srgb_gamma(v):
    return lookup(LUT_SRGB_GAMMA_REVERSE, v)


def apply_srgb_gamma(v):
    return lookup(LUT_SRGB_GAMMA, v)


def input8bit(v):
    v = np.ascontiguousarray(v, np.uint8); o = np.empty(v.shape, np.float32); lib().orc_input8bit(v.ravel(), o.ravel(), v.size); return o


def input16bit(v):
    v = np.ascontiguousarray(v, np.uint16); o = np.empty(v.shape, np.float32); lib().orc_input16bit(v.ravel(), o.ravel(), v.size); return o


def output8bit(v):
    v = _f32(v); o = np.empty(v.shape, np.uint8); lib().orc_output8bit(v.ravel(), o.ravel(), v.size); return o


def output16bit(v):
    v = _f32(v); o = np.empty(v.shape, np.uint16); lib().orc_output16bit(v.ravel(), o.ravel(), v.size); return o


def xyz_to_lab(xyz):
    v = _f32(xyz); o = np.empty_like(v); lib().orc_xyz_to_lab(v.ravel(), o.ravel(), v.size // 3); return o


def lab_to_xyz(lab):
    v = _f32(lab); o = np.empty_like(v); lib().orc_lab_to_xyz(v.ravel(), o.ravel(), v.size // 3); return o


def camera_to_lab(mul, cmatrix, pix4):
    v = _f32(pix4); n = v.size // 4
    o = np.empty(v.shape[:-1] + (3,), np.float32)
    lib().orc_camera_to_lab(_f32(mul).ravel(), _f32(cmatrix).ravel(), v.ravel(), o.ravel(), n); return o


def lab_to_rgb(rgbmatrix, pix3):
    v = _f32(pix3); o = np.empty_like(v); lib().orc_lab_to_rgb(_f32(rgbmatrix).ravel(), v.ravel(), o.ravel(), v.size // 3); return o


def cfa_shift(pat, x, y):
    out = C.create_string_buffer(160)
    if lib().orc_cfa_shift(pat.encode(), x, y, out):
        raise ValueError("invalid CFA pattern %r" % pat)
    return out.value.decode()


def cfa_pattern(pat):
    o = np.empty(48 * 48, np.int32)
    w = lib().orc_cfa_pattern(pat.encode(), o)
    if w < 0:
        raise ValueError("invalid CFA pattern %r" % pat)
    return w, o.reshape(48, 48)


def size_image(crop_top, crop_right, crop_bottom, crop_left, owidth, oheight):
    out = (_sz * 4)()
    if lib().orc_size_image(crop_top, crop_right, crop_bottom, crop_left, owidth, oheight, out):
        raise ValueError("image smaller than 10x10 (reference underflows)")
    return tuple(out)  # x, y, width, height


def gofloat_cfa(data, x, y, width, height, black0, white0):
    data = np.ascontiguousarray(data); oh, ow = data.shape
    out = np.empty((height, width), np.float32)
    fn = {np.dtype(np.uint16): lib().orc_gofloat_cfa_u16, np.dtype(np.float32): lib().orc_gofloat_cfa_f32}[data.dtype]
    fn(data.ravel(), ow, x, y, width, height, black0, white0, out.ravel()); return out


def gofloat_mono(data, x, y, width, height, black0, white0):
    data = np.ascontiguousarray(data); oh, ow = data.shape
    out = np.empty((height, width, 4), np.float32)
    fn = {np.dtype(np.uint16): lib().orc_gofloat_mono_u16, np.dtype(np.float32): lib().orc_gofloat_mono_f32}[data.dtype]
    fn(data.ravel(), ow, x, y, width, height, black0, white0, out.ravel()); return out


def gofloat_rgb(data, x, y, width, height, black4, white4):
    data = np.ascontiguousarray(data); oh, ow, _ = data.shape
    out = np.empty((height, width, 4), np.float32)
    fn = {np.dtype(np.uint16): lib().orc_gofloat_rgb_u16, np.dtype(np.float32): lib().orc_gofloat_rgb_f32}[data.dtype]
    fn(data.ravel(), ow, x, y, width, height, _f32(black4), _f32(white4), out.ravel()); return out


def gofloat_other(data, x, y, width, height):
    data = np.ascontiguousarray(data); oh, ow, _ = data.shape
    out = np.empty((height, width, 4), np.float32)
    fn = {np.dtype(np.uint8): lib().orc_gofloat_other_u8, np.dtype(np.uint16): lib().orc_gofloat_other_u16}[data.dtype]
    fn(data.ravel(), ow, x, y, width, height, out.ravel()); return out


def demosaic_full(cfa, buf):
    buf = _f32(buf); h, w = buf.shape
    out = np.empty((h, w, 4), np.float32)
    if lib().orc_demosaic_full(cfa.encode(), buf.ravel(), w, h, out.ravel()):
        raise ValueError("bad CFA")
    return out


def calculate_scaling_total(width, height, maxwidth, maxheight):
    s = C.c_float(); nw = _sz(); nh = _sz()
    lib().orc_calculate_scaling_total(width, height, maxwidth, maxheight, C.byref(s), C.byref(nw), C.byref(nh))
    return s.value, nw.value, nh.value


def transform_buffer(src, width, height, topleft, topright, bottomleft, nwidth, nheight, components, cfa=None):
    src = np.ascontiguousarray(src)
    fn = {np.dtype(np.float32): lib().orc_transform_buffer_f32, np.dtype(np.uint8): lib().orc_transform_buffer_u8,
          np.dtype(np.uint16): lib().orc_transform_buffer_u16}[src.dtype]
    out = np.empty(nwidth * nheight * components, src.dtype)
    rc = fn(src.ravel(), width, height, topleft[0], topleft[1], topright[0], topright[1], bottomleft[0], bottomleft[1],
            nwidth, nheight, components, cfa.encode() if cfa is not None else None, out)
    if rc:
        raise ValueError("transform_buffer failed")
    return out.reshape(nheight, nwidth, components)


def scaled_demosaic(cfa, buf, nwidth, nheight):
    buf = _f32(buf); h, w = buf.shape
    return transform_buffer(buf, w, h, (0, 0), (w - 1, 0), (0, h - 1), nwidth, nheight, 4, cfa)


def scale_down_opbuf(buf4, nwidth, nheight):
    buf4 = _f32(buf4); h, w, _ = buf4.shape
    return transform_buffer(buf4, w, h, (0, 0), (w - 1, 0), (0, h - 1), nwidth, nheight, 4, None)


def scale_down_srgb(img, nwidth, nheight):
    h, w, _ = img.shape
    return transform_buffer(img, w, h, (0, 0), (w - 1, 0), (0, h - 1), nwidth, nheight, 3, None)


def demosaic_run(cfa, buf, demosaic_width, demosaic_height):
    """OpDemosaic::run; returns (branch, out) with branch 0=pass-through 1=scale 2=scaled_demosaic 3=full 4=full+scale."""
    buf = _f32(buf)
    if buf.ndim == 2:
        h, w = buf.shape; colors = 1
    else:
        h, w, colors = buf.shape
    n = max(w * h, demosaic_width * demosaic_height)
    out = np.empty(n * 4, np.float32); ow = _sz(); oh = _sz()
    rc = lib().orc_demosaic_run(cfa.encode(), buf.ravel(), w, h, colors, demosaic_width, demosaic_height, out, C.byref(ow), C.byref(oh))
    if rc < 0:
        raise ValueError("demosaic failed")
    if rc == 0:
        return 0, buf
    return rc, out[: ow.value * oh.value * 4].reshape(oh.value, ow.value, 4).copy()


def normalize_wbs(vals):
    o = np.empty(4, np.float32); lib().orc_normalize_wbs(_f32(vals), o); return o


def tolab(buf4, wb_coeffs, cam_to_xyz_normalized, monochrome=False):
    buf4 = _f32(buf4); h, w, _ = buf4.shape
    out = np.empty((h, w, 3), np.float32)
    lib().orc_tolab(buf4.ravel(), w, h, int(monochrome), _f32(wb_coeffs), _f32(cam_to_xyz_normalized).ravel(), out.ravel()); return out


def fromlab(buf3):
    buf3 = _f32(buf3); h, w, _ = buf3.shape
    out = np.empty_like(buf3); lib().orc_fromlab(buf3.ravel(), w, h, out.ravel()); return out


def temp_to_xyz(temp):
    o = np.empty(3, np.float32); lib().orc_temp_to_xyz(temp, o); return o


def xyz_to_temp(xyz):
    o = np.empty(2, np.float32); lib().orc_xyz_to_temp(_f32(xyz), o); return float(o[0]), float(o[1])


def tolab_set_temp(xyz_to_cam, temp, tint):
    o = np.empty(4, np.float32); lib().orc_tolab_set_temp(_f32(xyz_to_cam).ravel(), temp, tint, o); return o


def tolab_get_temp(cam_to_xyz, wb_coeffs):
    o = np.empty(2, np.float32); lib().orc_tolab_get_temp(_f32(cam_to_xyz).ravel(), _f32(wb_coeffs), o); return float(o[0]), float(o[1])


def _pts(points):
    p = _f32(points).reshape(-1)
    return (p if p.size else np.zeros(2, np.float32)), p.size // 2


def spline_new(points):
    p, n = _pts(points)
    px = np.zeros(n + 2, np.float32); py = np.zeros(n + 2, np.float32)
    c1 = np.zeros(n + 2, np.float32); c2 = np.zeros(n + 2, np.float32); c3 = np.zeros(n + 2, np.float32)
    k = lib().orc_spline_new(p, n, px, py, c1, c2, c3)
    if k < 0:
        raise ValueError("bad spline")
    return px[:k], py[:k], c1[:k], c2[:k - 1], c3[:k - 1]


def spline_interpolate(points, vals):
    p, n = _pts(points); v = _f32(vals).ravel(); o = np.empty_like(v)
    if lib().orc_spline_interpolate(p, n, v, o, v.size):
        raise ValueError("bad spline")
    return o.reshape(np.shape(vals))


def basecurve(buf3, exposure, points):
    buf3 = _f32(buf3); h, w, _ = buf3.shape; p, n = _pts(points)
    out = np.empty_like(buf3)
    rc = lib().orc_basecurve(buf3.ravel(), w, h, exposure, p, n, out.ravel())
    if rc < 0:
        raise ValueError("bad curve")
    return buf3 if rc == 0 else out


def gamma(buf, linear=False):
    buf = _f32(buf); h, w, c = buf.shape
    out = np.empty_like(buf)
    return out if lib().orc_gamma(buf.ravel(), w, h, c, int(linear), out.ravel()) else buf


def orientation_to_flips(o):
    f = (C.c_int * 3)(); lib().orc_orientation_to_flips(o, f); return tuple(bool(x) for x in f)


def orientation_from_flips(t, fx, fy):
    return lib().orc_orientation_from_flips(int(t), int(fx), int(fy))


def transform_new(orientation):
    f = (C.c_int * 3)(); lib().orc_transform_new(orientation, f); return f[0], bool(f[1]), bool(f[2])


def transform_orientation(rotation, fliph, flipv):
    return lib().orc_transform_orientation(rotation, int(fliph), int(flipv))


def rotate_buffer(buf3, orientation):
    buf3 = _f32(buf3); h, w, _ = buf3.shape
    out = np.empty(buf3.size, np.float32); ow = _sz(); oh = _sz()
    lib().orc_rotate_buffer(buf3.ravel(), w, h, orientation, out, C.byref(ow), C.byref(oh))
    return out.reshape(oh.value, ow.value, 3)


def rotatecrop_calc_size(params5, width, height, reverse=False, input_ratio=1.0):
    ow = _sz(); oh = _sz()
    lib().orc_rotatecrop_calc_size(_f32(params5), input_ratio, width, height, int(reverse), C.byref(ow), C.byref(oh))
    return ow.value, oh.value


def rotatecrop_corners(params5, width, height):
    pts = np.zeros(6, np.int64); ow = _sz(); oh = _sz()
    rc = lib().orc_rotatecrop_corners(_f32(params5), width, height, pts, C.byref(ow), C.byref(oh))
    return None if rc == 0 else (tuple(pts[0:2]), tuple(pts[2:4]), tuple(pts[4:6]), ow.value, oh.value)


def rotatecrop_run(params5, buf):
    buf = _f32(buf); h, w, c = buf.shape; p = _f32(params5)
    ow = _sz(); oh = _sz()
    if lib().orc_rotatecrop_run(p, buf.ravel(), w, h, c, None, C.byref(ow), C.byref(oh)) == 0:
        return buf
    out = np.empty((oh.value, ow.value, c), np.float32)
    if lib().orc_rotatecrop_run(p, buf.ravel(), w, h, c, out.ctypes.data_as(C.c_void_p), C.byref(ow), C.byref(oh)) < 0:
        raise ValueError("rotatecrop failed")
    return out


def make_pipeline(data, *, source_kind=None, cfa="", is_cfa=None, cpp=1, crops=(0, 0, 0, 0),
                  blacklevels=(0, 0, 0, 0), whitelevels=(0, 0, 0, 0), rotatecrop=(0, 0, 0, 0, 0),
                  cam_to_xyz_normalized=None, wb_coeffs=(1.0, 1.0, 1.0, 0.0), exposure=0.0, points=None,
                  rotation=ROT_NORMAL, fliph=False, flipv=False, maxwidth=0, maxheight=0, linear=False, use_fastpath=False):
    """Builds the descriptor a `Pipeline::new_from_source` would hold.  For raw sources pass the
    already-cropped CFA string (`cropped_cfa()`), crops = (top, right, bottom, left)."""
    data = np.ascontiguousarray(data)
    d = PipelineDesc()
    if source_kind is None:
        if data.ndim == 3 and data.dtype == np.uint8: source_kind = 2
        elif data.ndim == 3 and data.dtype == np.uint16 and not cfa and cpp != 3: source_kind = 3
        elif data.dtype == np.uint16: source_kind = 0
        else: source_kind = 1
    d.source_kind = source_kind
    d._keep = data
    d.data = data.ctypes.data
    d.height, d.width = data.shape[0], data.shape[1]
    d.cpp = cpp
    d.cfa = cfa.encode()
    d.is_cfa = int(bool(cfa)) if is_cfa is None else int(is_cfa)
    d.crop_top, d.crop_right, d.crop_bottom, d.crop_left = crops
    d.blacklevels[:] = list(blacklevels); d.whitelevels[:] = list(whitelevels)
    d.rc[:] = list(rotatecrop)
    cm = const_srgb_d65_43() if cam_to_xyz_normalized is None else _f32(cam_to_xyz_normalized)
    d.cam_to_xyz_normalized[:] = list(cm.ravel())
    d.wb_coeffs[:] = list(wb_coeffs)
    d.exposure = exposure
    if points is None:
        points = [(0.5, 0.6)] if source_kind <= 1 else []
    p = _f32(points).ravel()
    d.npoints = p.size // 2
    for i, v in enumerate(p):
        d.points[i] = v
    d.rotation = rotation; d.fliph = int(fliph); d.flipv = int(flipv)
    d.maxwidth = maxwidth; d.maxheight = maxheight; d.linear = int(linear)
    d.use_fastpath = int(use_fastpath)          # the reference's default is true; tests of the slow path say so explicitly
    return d


def pipeline_sizes(desc):
    a, b, c, d = _sz(), _sz(), _sz(), _sz()
    if lib().orc_pipeline_sizes(C.byref(desc), C.byref(a), C.byref(b), C.byref(c), C.byref(d)):
        raise ValueError("size negotiation failed")
    return (a.value, b.value), (c.value, d.value)


def _take(ptr, w, h, dtype):
    if not ptr:
        raise RuntimeError("oracle pipeline failed")
    n = w * h * 3
    ct = {np.float32: C.c_float, np.uint8: C.c_uint8, np.uint16: C.c_uint16}[dtype]
    arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(n,)).copy().reshape(h, w, 3)
    lib().orc_free(ptr)
    return arr


def pipeline_run(desc):
    w = _sz(); h = _sz()
    p = lib().orc_pipeline_run(C.byref(desc), C.byref(w), C.byref(h))
    return _take(p, w.value, h.value, np.float32)


def pipeline_output_8bit(desc):
    w = _sz(); h = _sz()
    p = lib().orc_pipeline_output_8bit(C.byref(desc), C.byref(w), C.byref(h))
    return _take(p, w.value, h.value, np.uint8)


def pipeline_output_16bit(desc):
    w = _sz(); h = _sz()
    p = lib().orc_pipeline_output_16bit(C.byref(desc), C.byref(w), C.byref(h))
    return _take(p, w.value, h.value, np.uint16)
