"""ctypes binding of libimagepipe_amd.so (the C ABI in include/imagepipe_amd.h).

The library is the product; this module only loads it and declares the signatures.  If the shared
object is missing the import fails loudly -- there is no Python or CPU implementation to fall back to.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("IPK_SO_OVERRIDE") or os.path.join(_HERE, "libimagepipe_amd.so")   # override: dev ablation builds only

IPK_OK, IPK_NOOP = 0, 1
SRC_U16, SRC_F32, SRC_RGB8, SRC_RGB16 = 0, 1, 2, 3
OUT_F32, OUT_U8, OUT_U16 = 0, 1, 2
SCHED_AUTO, SCHED_SPLIT = 0, 1
OR_NORMAL, OR_HFLIP, OR_ROT180, OR_VFLIP, OR_TRANSPOSE, OR_ROT90, OR_TRANSVERSE, OR_ROT270, OR_UNKNOWN = range(9)
ROT_NORMAL, ROT_90, ROT_180, ROT_270 = range(4)

_sz = C.c_size_t
_vp = C.c_void_p
_fp = C.POINTER(C.c_float)
_szp = C.POINTER(C.c_size_t)
_i64 = C.c_int64


class _SizedDesc(C.Structure):
    """a descriptor with a leading struct_size ("Descriptor versioning", include/imagepipe_amd.h): a new object states its own size"""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.struct_size = C.sizeof(self)


class FusedParams(_SizedDesc):
    """ipk_fused_params"""
    _fields_ = [
        ("struct_size", C.c_uint32), ("src_type", C.c_int), ("owidth", _sz),
        ("x", _sz), ("y", _sz), ("width", _sz), ("height", _sz),
        ("black0", C.c_float), ("white0", C.c_float),
        ("cfa", C.c_char * 160),
        ("wb_coeffs", C.c_float * 4), ("cam_to_xyz_normalized", C.c_float * 12),
        ("exposure", C.c_float), ("npoints", C.c_int), ("points", C.c_float * 128),
        ("linear", C.c_int), ("out_type", C.c_int),
        ("band_src_row0", _sz), ("band_src_rows", _sz), ("band_out_row0", _sz), ("band_out_rows", _sz),
        ("cfa_width", C.c_int), ("cfa_height", C.c_int),           # later additions are appended, never inserted
        ("schedule", C.c_int), ("reserved0", C.c_int),
    ]


class PipelineDesc(_SizedDesc):
    """ipk_pipeline_desc"""
    _fields_ = [
        ("struct_size", C.c_uint32), ("src_type", C.c_int), ("width", _sz), ("height", _sz),
        ("cpp", C.c_int), ("is_cfa", C.c_int), ("cfa", C.c_char * 160),
        ("crop_top", _sz), ("crop_right", _sz), ("crop_bottom", _sz), ("crop_left", _sz),
        ("blacklevels", C.c_float * 4), ("whitelevels", C.c_float * 4),
        ("rotatecrop", C.c_float * 5),
        ("cam_to_xyz_normalized", C.c_float * 12), ("wb_coeffs", C.c_float * 4),
        ("exposure", C.c_float), ("npoints", C.c_int), ("points", C.c_float * 128),
        ("rotation", C.c_int), ("fliph", C.c_int), ("flipv", C.c_int),
        ("maxwidth", _sz), ("maxheight", _sz),
        ("linear", C.c_int), ("allow_fused", C.c_int), ("use_fastpath", C.c_int),
        ("cfa_width", C.c_int), ("cfa_height", C.c_int),
        ("schedule", C.c_int), ("reserved0", C.c_int), ("reserved1", C.c_int), ("reserved2", C.c_int),
    ]


class StageTime(C.Structure):
    """ipk_stage_time"""
    _fields_ = [("name", C.c_char * 88), ("ms", C.c_float)]


class Band(C.Structure):
    """ipk_band"""
    _fields_ = [("out_row0", _sz), ("out_rows", _sz), ("src_row0", _sz), ("src_rows", _sz)]


# ipk_exchange_fn: (ctx, send_peer, send, send_bytes, recv_peer, recv, recv_bytes) -> int
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, _vp, C.c_int, _vp, _sz, C.c_int, _vp, _sz)
_bp = C.POINTER(Band)
COMM_ID_BYTES = 128

# name -> (restype, argtypes).  Every symbol include/imagepipe_amd.h declares is listed here;
# tests/test_cabi_symbols.py cross-checks this table against the header text.
_GO = [_sz, _sz, _sz, _sz, _sz]                       # owidth, x, y, width, height
_CORNERS = [_i64] * 6
SIGNATURES = {
    "ipk_init": (C.c_int, [C.c_int]),
    "ipk_shutdown": (None, []),
    "ipk_is_initialized": (C.c_int, []),
    "ipk_last_error": (C.c_char_p, []),
    "ipk_device_cus": (C.c_int, []),
    "ipk_host_libm_matches": (C.c_int, [_szp]),
    "ipk_malloc": (C.c_int, [C.POINTER(_vp), _sz]),
    "ipk_free": (C.c_int, [_vp]),
    "ipk_memcpy_h2d": (C.c_int, [_vp, _vp, _sz, _vp]),
    "ipk_memcpy_d2h": (C.c_int, [_vp, _vp, _sz, _vp]),
    "ipk_stream_sync": (C.c_int, [_vp]),
    "ipk_lut_table": (C.c_int, [C.c_int, _fp]),
    "ipk_size_image": (C.c_int, [_sz] * 6 + [_szp]),
    "ipk_calculate_scaling_total": (C.c_int, [_sz] * 4 + [_fp, _szp, _szp]),
    "ipk_normalize_wbs": (C.c_int, [_fp, _fp]),
    "ipk_const_matrix": (C.c_int, [C.c_int, _fp]),
    "ipk_temp_to_xyz": (C.c_int, [C.c_float, _fp]),
    "ipk_xyz_to_temp": (C.c_int, [_fp, _fp, _fp]),
    "ipk_tolab_set_temp": (C.c_int, [_fp, C.c_float, C.c_float, _fp]),
    "ipk_tolab_get_temp": (C.c_int, [_fp, _fp, _fp, _fp]),
    "ipk_spline_new": (C.c_int, [_fp, C.c_int, _fp, _fp, _fp, _fp, _fp]),
    "ipk_rotatecrop_calc_size": (C.c_int, [_fp, C.c_float, _sz, _sz, C.c_int, _szp, _szp]),
    "ipk_cfa_shift": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_char_p]),
    "ipk_orientation_to_flips": (C.c_int, [C.c_int, C.POINTER(C.c_int)]),
    "ipk_orientation_from_flips": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "ipk_transform_orientation": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "ipk_gofloat_cfa_u16": (C.c_int, [_vp] + _GO + [C.c_float, C.c_float, _vp, _vp]),
    "ipk_gofloat_cfa_f32": (C.c_int, [_vp] + _GO + [C.c_float, C.c_float, _vp, _vp]),
    "ipk_gofloat_mono_u16": (C.c_int, [_vp] + _GO + [C.c_float, C.c_float, _vp, _vp]),
    "ipk_gofloat_mono_f32": (C.c_int, [_vp] + _GO + [C.c_float, C.c_float, _vp, _vp]),
    "ipk_gofloat_rgb_u16": (C.c_int, [_vp] + _GO + [_fp, _fp, _vp, _vp]),
    "ipk_gofloat_rgb_f32": (C.c_int, [_vp] + _GO + [_fp, _fp, _vp, _vp]),
    "ipk_gofloat_other_u8": (C.c_int, [_vp] + _GO + [_vp, _vp]),
    "ipk_gofloat_other_u16": (C.c_int, [_vp] + _GO + [_vp, _vp]),
    "ipk_demosaic_full": (C.c_int, [_vp, _sz, _sz, C.c_char_p, _vp, _vp]),
    "ipk_demosaic_full_band": (C.c_int, [_vp, _sz, _sz, _sz, _sz, _sz, _sz, C.c_char_p, _vp, _vp]),
    "ipk_transform_buffer_f32": (C.c_int, [_vp, _sz, _sz] + _CORNERS + [_sz, _sz, _sz, C.c_char_p, _vp, _vp]),
    "ipk_transform_buffer_u8": (C.c_int, [_vp, _sz, _sz] + _CORNERS + [_sz, _sz, _sz, C.c_char_p, _vp, _vp]),
    "ipk_transform_buffer_u16": (C.c_int, [_vp, _sz, _sz] + _CORNERS + [_sz, _sz, _sz, C.c_char_p, _vp, _vp]),
    "ipk_scaled_demosaic": (C.c_int, [_vp, _sz, _sz, C.c_char_p, _sz, _sz, _vp, _vp]),
    "ipk_scale_down_opbuf": (C.c_int, [_vp, _sz, _sz, _sz, _sz, _vp, _vp]),
    "ipk_raw_scaled_demosaic": (C.c_int, [_vp, C.c_int, _sz, _sz, _sz, _sz, _sz, C.c_float, C.c_float, C.c_char_p, _sz, _sz, _vp, _vp]),
    "ipk_raster_scale_down": (C.c_int, [_vp, C.c_int, _sz, _sz, _sz, _sz, _sz, _sz, _sz, _vp, _vp]),
    "ipk_demosaic_run": (C.c_int, [_vp, _sz, _sz, _sz, C.c_char_p, _sz, _sz, _vp, _szp, _szp, _vp]),
    "ipk_rotatecrop": (C.c_int, [_vp, _sz, _sz, _sz, _fp, _vp, _szp, _szp, _vp]),
    "ipk_tolab": (C.c_int, [_vp, _sz, _sz, C.c_int, _fp, _fp, _vp, _vp]),
    "ipk_basecurve": (C.c_int, [_vp, _sz, _sz, C.c_float, _fp, C.c_int, _vp, _vp]),
    "ipk_fromlab": (C.c_int, [_vp, _sz, _sz, _vp, _vp]),
    "ipk_gamma": (C.c_int, [_vp, _sz, _sz, _sz, C.c_int, _vp, _vp]),
    "ipk_rotate_buffer": (C.c_int, [_vp, _sz, _sz, C.c_int, _vp, _szp, _szp, _vp]),
    "ipk_rotate_image_u8": (C.c_int, [_vp, _sz, _sz, C.c_int, _vp, _szp, _szp, _vp]),
    "ipk_rotate_image_u16": (C.c_int, [_vp, _sz, _sz, C.c_int, _vp, _szp, _szp, _vp]),
    "ipk_transform": (C.c_int, [_vp, _sz, _sz, C.c_int, C.c_int, C.c_int, _vp, _szp, _szp, _vp]),
    "ipk_output8bit": (C.c_int, [_vp, _sz, _vp, _vp]),
    "ipk_output16bit": (C.c_int, [_vp, _sz, _vp, _vp]),
    "ipk_raw_to_srgb": (C.c_int, [C.POINTER(FusedParams), _vp, _vp, _vp]),
    "ipk_raw_to_srgb_batch": (C.c_int, [C.POINTER(FusedParams), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _sz, _vp]),
    "ipk_raw_to_srgb_oriented": (C.c_int, [C.POINTER(FusedParams), _vp, C.c_int, _vp, _szp, _szp, _vp]),
    "ipk_pipeline_sizes": (C.c_int, [C.POINTER(PipelineDesc), _szp, _szp, _szp, _szp]),
    "ipk_pipeline_run": (C.c_int, [C.POINTER(PipelineDesc), _vp, _vp, C.c_int, C.POINTER(C.c_int), _vp]),
    "ipk_host_pipeline_run": (C.c_int, [C.POINTER(PipelineDesc), _vp, _vp, C.c_int, C.POINTER(C.c_int)]),
    "ipk_host_pipeline_run_batch": (C.c_int, [C.POINTER(PipelineDesc), C.POINTER(_vp), C.POINTER(_vp), _sz, C.c_int, C.POINTER(C.c_int)]),
    "ipk_pipeline_run_batch": (C.c_int, [C.POINTER(PipelineDesc), C.POINTER(_vp), C.POINTER(_vp), _sz, C.c_int, C.POINTER(C.c_int), _vp]),
    "ipk_pipeline_run_batch_multi": (C.c_int, [C.POINTER(PipelineDesc), C.POINTER(_vp), C.POINTER(_vp), _sz, C.c_int, C.POINTER(C.c_int)]),
    "ipk_host_pipeline_run_batch_multi": (C.c_int, [C.POINTER(PipelineDesc), C.POINTER(_vp), C.POINTER(_vp), _sz, C.c_int, C.POINTER(C.c_int)]),
    "ipk_devices_sync": (C.c_int, []),
    "ipk_ctx_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "ipk_ctx_destroy": (C.c_int, [_vp]),
    "ipk_ctx_make_current": (C.c_int, [_vp]),
    "ipk_ctx_current": (_vp, []),
    "ipk_ctx_device": (C.c_int, [_vp]),
    "ipk_init_devices": (C.c_int, [C.POINTER(C.c_int), C.c_int]),
    "ipk_device_set_size": (C.c_int, []),
    "ipk_device_ctx": (_vp, [C.c_int]),
    "ipk_deal_frames": (C.c_int, [_sz, C.c_int, C.c_int, _szp, _szp, _szp]),
    "ipk_host_alloc": (_vp, [_sz]),
    "ipk_host_free": (None, [_vp]),
    "ipk_host_gofloat_cfa_u16": (C.c_int, [_vp, _sz, _sz, _sz, _sz, _sz, _sz, C.c_float, C.c_float, _vp]),
    "ipk_host_gofloat_cfa_f32": (C.c_int, [_vp, _sz, _sz, _sz, _sz, _sz, _sz, C.c_float, C.c_float, _vp]),
    "ipk_host_demosaic_full": (C.c_int, [_vp, _sz, _sz, C.c_char_p, _vp]),
    "ipk_host_transform_buffer_f32": (C.c_int, [_vp, _sz, _sz] + _CORNERS + [_sz, _sz, _sz, C.c_char_p, _vp]),
    "ipk_host_tolab": (C.c_int, [_vp, _sz, _sz, C.c_int, _fp, _fp, _vp]),
    "ipk_host_basecurve": (C.c_int, [_vp, _sz, _sz, C.c_float, _fp, C.c_int, _vp]),
    "ipk_host_fromlab": (C.c_int, [_vp, _sz, _sz, _vp]),
    "ipk_host_gamma": (C.c_int, [_vp, _sz, _sz, _sz, C.c_int, _vp]),
    "ipk_host_rotate_buffer": (C.c_int, [_vp, _sz, _sz, C.c_int, _vp, _szp, _szp]),
    "ipk_host_output8bit": (C.c_int, [_vp, _sz, _vp]),
    "ipk_host_output16bit": (C.c_int, [_vp, _sz, _vp]),
    "ipk_host_raw_to_srgb": (C.c_int, [C.POINTER(FusedParams), _vp, _vp]),
    "ipk_selftest_cdiv": (C.c_int, [C.c_float, C.c_int, C.c_float, C.c_float, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
    "ipk_selftest_lut_weight": (C.c_int, [C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
    "ipk_selftest_clamp01": (C.c_int, [C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
    "ipk_abi_sizeof": (_sz, [C.c_int]),
    "ipk_copy_probe": (C.c_int, [_vp, _vp, _sz, _vp]),
    "ipk_mix_probe": (C.c_int, [_vp, _vp, _sz, _vp]),
    "ipk_clock_probe": (C.c_int, [_vp, C.c_uint32, _vp]),
    "ipk_stream_probe": (C.c_int, [C.POINTER(FusedParams), _vp, _vp, _vp]),
    "ipk_selftest_spline3": (C.c_int, [C.c_float, _fp, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
    "ipk_selftest_quant8": (C.c_int, [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
    "ipk_selftest_q8": (C.c_int, [C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
    "ipk_selftest_quant16": (C.c_int, [C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
    "ipk_selftest_cbrtf": (C.c_int, [_vp, _vp, _sz, C.c_int, _vp]),
    "ipk_selftest_task_queue": (C.c_int, [C.c_int]),
    "ipk_selftest_cache_put": (C.c_int, [_vp, C.c_char_p, _sz]),
    "ipk_selftest_sha256": (C.c_int, [C.c_char_p, _sz, C.c_char_p]),
    "ipk_pointwise_chain": (C.c_int, [_vp, _sz, _sz, C.c_int, _fp, _fp, C.c_float, _fp, C.c_int, C.c_int, _vp, _vp]),
    "ipk_pointwise_chain_out": (C.c_int, [_vp, _sz, _sz, C.c_int, _fp, _fp, C.c_float, _fp, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "ipk_raster_to_srgb": (C.c_int, [_vp, C.c_int, _sz, _sz, _fp, _fp, C.c_float, _fp, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "ipk_pipeline_takes_fastpath": (C.c_int, [C.POINTER(PipelineDesc), C.c_int]),
    "ipk_pipeline_hashes": (C.c_int, [C.POINTER(PipelineDesc), C.c_int, C.c_uint64, C.c_char_p]),
    "ipk_cache_new": (C.c_int, [_sz, C.POINTER(C.c_void_p)]),
    "ipk_cache_free": (C.c_int, [_vp]),
    "ipk_cache_clear": (C.c_int, [_vp]),
    "ipk_cache_contains": (C.c_int, [_vp, C.c_char_p]),
    "ipk_cache_stats": (C.c_int, [_vp, C.POINTER(_sz), C.POINTER(_sz), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "ipk_cache_get": (C.c_int, [_vp, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(_sz), C.POINTER(_sz), C.POINTER(_sz), C.POINTER(C.c_int)]),
    "ipk_timing_begin": (C.c_int, []),
    "ipk_timing_end": (C.c_int, [C.POINTER(StageTime), C.c_int, C.POINTER(C.c_int)]),
    "ipk_band_plan": (C.c_int, [_sz, C.c_int, C.c_int, _bp]),
    "ipk_band_plan_scaled": (C.c_int, [_sz, _sz, C.c_int, _bp]),
    "ipk_raw_scaled_demosaic_band": (C.c_int, [_vp, C.c_int, _sz, _sz, _sz, _sz, C.c_float, C.c_float, C.c_char_p, _sz, _sz, _bp, _vp, _vp]),
    "ipk_comm_unique_id": (C.c_int, [C.c_char_p]),
    "ipk_comm_init_rccl": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.POINTER(_vp)]),
    "ipk_comm_init_host": (C.c_int, [C.c_int, C.c_int, EXCHANGE_FN, _vp, C.POINTER(_vp)]),
    "ipk_comm_free": (C.c_int, [_vp]),
    "ipk_comm_info": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ipk_band_exchange_halo": (C.c_int, [_vp, _vp, _sz, _bp, _vp]),
    "ipk_host_band_exchange_halo": (C.c_int, [_vp, _vp, _sz, _bp]),
    "ipk_band_gather": (C.c_int, [_vp, _vp, _sz, _bp, C.c_int, _vp]),
    "ipk_host_band_gather": (C.c_int, [_vp, _vp, _sz, _bp, C.c_int]),
    "ipk_band_gather_begin": (C.c_int, [_vp, _vp, _sz, _bp, C.c_int, _vp]),
    "ipk_comm_wait": (C.c_int, [_vp, _vp]),
    "ipk_comm_selftest": (C.c_int, [_vp]),
    "ipk_pipeline_run_cached": (C.c_int, [C.POINTER(PipelineDesc), _vp, C.c_uint64, _vp, C.c_int, _vp, C.POINTER(C.c_int), C.POINTER(C.c_int), _vp]),
}

_lib = None


class IpkError(RuntimeError):
    """a negative ipk_status; `code` holds it (None for failures raised by the Python plumbing itself)"""
    code = None


def load():
    """Loads the shared library (never builds it, never substitutes anything for it)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise ImportError(
            "imagepipe_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C imagepipe_amd/csrc`. There is no CPU fallback." % SO_PATH)
    lib = C.CDLL(SO_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        if os.environ.get("IPK_SO_OVERRIDE") and not hasattr(lib, name):
            continue                     # a development A/B against an OLDER build of the library (tools/*_ab.sh): what it lacks cannot be called
        fn = getattr(lib, name)          # AttributeError here = the .so does not export what the header declares
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc < 0:
        err = IpkError("%s failed (%d): %s" % (what or "ipk call", rc, load().ipk_last_error().decode()))
        err.code = rc
        raise err
    return rc
