"""imagepipe_amd -- MI355X-native raw->sRGB hot path behind imagepipe's Pipeline / ImageOp surface.

This package is plumbing: it loads libimagepipe_amd.so (HIP kernels + C ABI, include/imagepipe_amd.h),
uses torch only to own device memory / streams / process groups, and mirrors the reference's
interface names (OpBuffer, OpGoFloat ... OpTransform, Pipeline, PipelineSettings) so tests read like
the reference's.  All pixel work happens in the shared library on the GPU; nothing here computes
pixels and nothing falls back to the CPU.
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import (IpkError, FusedParams, PipelineDesc, OUT_F32, OUT_U8, OUT_U16, SRC_U16, SRC_F32, SRC_RGB8, SRC_RGB16,
                   OR_NORMAL, OR_HFLIP, OR_ROT180, OR_VFLIP, OR_TRANSPOSE, OR_ROT90, OR_TRANSVERSE, OR_ROT270, OR_UNKNOWN,
                   ROT_NORMAL, ROT_90, ROT_180, ROT_270, IPK_NOOP)

__all__ = ["init", "init_devices", "deal_frames", "Context", "lib", "OpBuffer", "RawImage", "OtherImage", "PipelineSettings", "PipelineGlobals", "PipelineOps",
           "Pipeline", "OpGoFloat", "OpDemosaic", "OpRotateCrop", "OpToLab", "OpBaseCurve", "OpFromLab", "OpGamma",
           "OpTransform", "raw_to_srgb", "FusedPlan", "IpkError"]

_initialized_device = None


def lib():
    return _lib.load()


def init(device: Optional[int] = None):
    """Binds the library to a GPU (torch's current device by default).  Raises without a GPU."""
    global _initialized_device
    L = lib()
    if not torch.cuda.is_available():
        raise IpkError("imagepipe_amd needs a HIP device (torch.cuda.is_available() is False); there is no CPU fallback")
    if device is None:
        device = torch.cuda.current_device()
    if _initialized_device != device:
        torch.cuda.set_device(device)
        torch.zeros(1, device="cuda")               # make sure torch's HIP context exists first
        _lib.check(L.ipk_init(device), "ipk_init")
        _initialized_device = device
    return device


class Context:
    """ipk_ctx: one more device binding in this process (a second GPU, or a second independent pipeline on the same GPU).  `with ctx:` makes it
    the calling thread's current context -- every ipk_* call inside runs on it -- and puts the previous one back on exit."""

    def __init__(self, device: int = 0, handle=None):
        if handle is None:
            h = C.c_void_p()
            _lib.check(lib().ipk_ctx_create(int(device), C.byref(h)), "ipk_ctx_create")
            handle, self._owned = h.value, True
        else:
            self._owned = False
        self.handle = handle
        self.device = lib().ipk_ctx_device(handle)
        self._prev = []

    def make_current(self):
        _lib.check(lib().ipk_ctx_make_current(self.handle), "ipk_ctx_make_current")

    def __enter__(self):
        self._prev.append(lib().ipk_ctx_current())
        self.make_current()
        return self

    def __exit__(self, *exc):
        prev = self._prev.pop()
        # the process default is restored as "no explicit choice" (NULL), anything else as itself
        lib().ipk_ctx_make_current(prev if prev and prev != self.handle else None)
        return False

    def destroy(self):
        if self.handle and self._owned:
            _lib.check(lib().ipk_ctx_destroy(self.handle), "ipk_ctx_destroy")
        self._owned = False


def init_devices(devices=None):
    """ipk_init_devices: the process's device set, one context per listed device ([] / None = every visible device; an ordinal may repeat).
    Returns the members as Context objects (not owned: ipk_init_devices / ipk_shutdown manage them)."""
    global _initialized_device
    L = lib()
    if not torch.cuda.is_available():
        raise IpkError("imagepipe_amd needs a HIP device (torch.cuda.is_available() is False); there is no CPU fallback")
    devs = list(devices or [])
    for d in sorted(set(devs or range(torch.cuda.device_count()))):
        torch.zeros(1, device="cuda:%d" % d)        # torch's HIP context on every device first
    arr = (C.c_int * max(1, len(devs)))(*devs)
    _lib.check(L.ipk_init_devices(arr if devs else None, len(devs)), "ipk_init_devices")
    members = [Context(handle=L.ipk_device_ctx(i)) for i in range(L.ipk_device_set_size())]
    if _initialized_device is None and members:
        _initialized_device = members[0].device
    return members


def deal_frames(n_frames, n_devices, index):
    """ipk_deal_frames: the frame indices set member `index` of `n_devices` gets of an n_frames batch (host arithmetic)"""
    f, s, c = C.c_size_t(), C.c_size_t(), C.c_size_t()
    _lib.check(lib().ipk_deal_frames(n_frames, n_devices, index, C.byref(f), C.byref(s), C.byref(c)), "ipk_deal_frames")
    return [f.value + k * s.value for k in range(c.value)]


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t: torch.Tensor):
    assert t.is_cuda and t.is_contiguous()
    return C.c_void_p(t.data_ptr())


def _farr(vals, n):
    a = (C.c_float * n)()
    v = np.asarray(vals, dtype=np.float32).ravel()
    assert v.size == n, (v.size, n)
    for i in range(n):
        a[i] = v[i]
    return a


def _points(points):
    p = np.asarray(points, dtype=np.float32).reshape(-1)
    arr = (C.c_float * max(2, p.size))()
    for i, v in enumerate(p):
        arr[i] = v
    return arr, p.size // 2


# ---------------------------------------------------------------------------------------------
# OpBuffer (src/buffer.rs:5-32): width, height, colors, monochrome, data (f32, row-major interleaved)
# ---------------------------------------------------------------------------------------------
@dataclass
class OpBuffer:
    width: int
    height: int
    colors: int
    monochrome: bool
    data: torch.Tensor                      # device, float32, width*height*colors elements

    @staticmethod
    def new(width, height, colors, monochrome=False):
        return OpBuffer(width, height, colors, monochrome,
                        torch.zeros(width * height * colors, dtype=torch.float32, device="cuda"))

    @staticmethod
    def from_numpy(a: np.ndarray, monochrome=False):
        a = np.ascontiguousarray(a, dtype=np.float32)
        if a.ndim == 2:
            h, w = a.shape; c = 1
        else:
            h, w, c = a.shape
        return OpBuffer(w, h, c, monochrome, torch.from_numpy(a.reshape(-1)).cuda())

    def numpy(self):
        a = self.data.cpu().numpy()
        return a.reshape(self.height, self.width) if self.colors == 1 else a.reshape(self.height, self.width, self.colors)


# ---------------------------------------------------------------------------------------------
# Sources: the fields of rawloader::RawImage / image::DynamicImage the hot path reads
# ---------------------------------------------------------------------------------------------
@dataclass
class RawImage:
    width: int
    height: int
    data: torch.Tensor                      # device uint16 (stored as int16 bits) or float32, width*height*cpp
    cpp: int = 1
    cfa: str = ""                           # uncropped CFA pattern; "" = not a CFA image
    crops: Sequence[int] = (0, 0, 0, 0)     # top, right, bottom, left
    blacklevels: Sequence[float] = (0, 0, 0, 0)
    whitelevels: Sequence[float] = (65535, 65535, 65535, 65535)
    wb_coeffs: Sequence[float] = (1.0, 1.0, 1.0, float("nan"))
    neutralwb: Optional[Sequence[float]] = None             # RawImage::neutralwb() (rawloader; caller-computed): OpToLab's fallback
    cam_to_xyz_normalized: Optional[np.ndarray] = None      # [[f32;4];3]; default SRGB_D65_43
    cam_to_xyz: Optional[np.ndarray] = None                 # [[f32;4];3]; only OpToLab.get_temp reads it
    xyz_to_cam: Optional[np.ndarray] = None                 # [[f32;3];4]; only OpToLab.set_temp reads it
    orientation: int = OR_NORMAL
    is_float: bool = False

    def cropped_cfa(self):
        if not self.cfa:
            return ""
        out = C.create_string_buffer(200)
        _lib.check(lib().ipk_cfa_shift(self.cfa.encode(), int(self.crops[3]), int(self.crops[0]), out), "ipk_cfa_shift")
        return out.value.decode()


@dataclass
class OtherImage:
    width: int
    height: int
    data: torch.Tensor                      # device uint8 or uint16(int16 bits), width*height*3
    bits: int = 8


def upload_u16(a: np.ndarray) -> torch.Tensor:
    """uint16 numpy -> device tensor (torch has no uint16 arithmetic; the bits are kept in int16)."""
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint16).view(np.int16).reshape(-1)).cuda()


def xyz_d65_34():
    """XYZ_D65_34 (color_conversions.rs:9-11) from the library's own f32 inverse"""
    o = (C.c_float * 12)()
    _lib.check(lib().ipk_const_matrix(3, o), "ipk_const_matrix")
    return np.array(o[:], np.float32).reshape(4, 3)


SRGB_D65_43 = np.array([[0.4124564, 0.3575761, 0.1804375, 0.0],
                        [0.2126729, 0.7151522, 0.0721750, 0.0],
                        [0.0193339, 0.1191920, 0.9503041, 0.0]], dtype=np.float32)


# ---------------------------------------------------------------------------------------------
# PipelineSettings / PipelineGlobals (src/pipeline.rs:110-151)
# ---------------------------------------------------------------------------------------------
@dataclass
class PipelineSettings:
    maxwidth: int = 0
    maxheight: int = 0
    demosaic_width: int = 0
    demosaic_height: int = 0
    linear: bool = False
    use_fastpath: bool = True


@dataclass
class PipelineGlobals:
    image: object
    settings: PipelineSettings = field(default_factory=PipelineSettings)


# ---------------------------------------------------------------------------------------------
# The eight ops (src/ops/*.rs).  Each `run` is one C-ABI call on device buffers and follows the
# reference's contract: returns either the *same* OpBuffer (no-op cases) or a fresh one.
# ---------------------------------------------------------------------------------------------
class ImageOp:
    name = "?"

    def run(self, pipeline: PipelineGlobals, buf: OpBuffer) -> OpBuffer:
        raise NotImplementedError

    def transform_forward(self, width, height):
        return width, height

    def transform_reverse(self, width, height):
        return width, height

    def reset(self):
        pass


class OpGoFloat(ImageOp):
    """src/ops/gofloat.rs"""
    name = "gofloat"

    def __init__(self, img):
        if isinstance(img, RawImage):
            self.crop_top, self.crop_right, self.crop_bottom, self.crop_left = [int(c) for c in img.crops]
            self.is_cfa = bool(img.cfa)
            self.blacklevels = [float(v) for v in img.blacklevels]
            self.whitelevels = [float(v) for v in img.whitelevels]
        else:
            self.crop_top = self.crop_right = self.crop_bottom = self.crop_left = 0
            self.is_cfa = False
            self.blacklevels = [0.0] * 4
            self.whitelevels = [0.0] * 4

    def size_image(self, owidth, oheight):
        out = (C.c_size_t * 4)()
        _lib.check(lib().ipk_size_image(self.crop_top, self.crop_right, self.crop_bottom, self.crop_left, owidth, oheight, out),
                   "ipk_size_image")
        return tuple(out)

    def transform_forward(self, width, height):
        _, _, w, h = self.size_image(width, height)
        return w, h

    def run(self, pipeline, _buf=None):
        img = pipeline.image
        L = lib()
        x, y, w, h = self.size_image(img.width, img.height)
        if isinstance(img, RawImage):
            sfx = "f32" if img.is_float else "u16"
            if img.cpp == 1 and not self.is_cfa:
                out = OpBuffer.new(w, h, 4, True)
                fn = getattr(L, "ipk_gofloat_mono_" + sfx)
                _lib.check(fn(_ptr(img.data), img.width, x, y, w, h, self.blacklevels[0], self.whitelevels[0], _ptr(out.data), _stream()), fn.__name__)
            elif img.cpp == 3:
                out = OpBuffer.new(w, h, 4, False)
                fn = getattr(L, "ipk_gofloat_rgb_" + sfx)
                _lib.check(fn(_ptr(img.data), img.width, x, y, w, h, _farr(self.blacklevels, 4), _farr(self.whitelevels, 4), _ptr(out.data), _stream()), fn.__name__)
            else:
                out = OpBuffer.new(w, h, img.cpp, False)
                fn = getattr(L, "ipk_gofloat_cfa_" + sfx)
                _lib.check(fn(_ptr(img.data), img.width, x, y, w, h, self.blacklevels[0], self.whitelevels[0], _ptr(out.data), _stream()), fn.__name__)
        else:
            out = OpBuffer.new(w, h, 4, False)
            fn = L.ipk_gofloat_other_u8 if img.bits == 8 else L.ipk_gofloat_other_u16
            _lib.check(fn(_ptr(img.data), img.width, x, y, w, h, _ptr(out.data), _stream()), "ipk_gofloat_other")
        return out


class OpDemosaic(ImageOp):
    """src/ops/demosaic.rs"""
    name = "demosaic"

    def __init__(self, img):
        self.cfa = img.cropped_cfa() if isinstance(img, RawImage) else ""

    def run(self, pipeline, buf):
        nw, nh = pipeline.settings.demosaic_width, pipeline.settings.demosaic_height
        out = torch.empty(max(buf.width * buf.height, nw * nh) * 4, dtype=torch.float32, device="cuda")
        ow, oh = C.c_size_t(), C.c_size_t()
        rc = _lib.check(lib().ipk_demosaic_run(_ptr(buf.data), buf.width, buf.height, buf.colors, self.cfa.encode(), nw, nh,
                                               _ptr(out), C.byref(ow), C.byref(oh), _stream()), "ipk_demosaic_run")
        if rc == IPK_NOOP:
            return buf
        return OpBuffer(ow.value, oh.value, 4, buf.monochrome, out[: ow.value * oh.value * 4])


class OpRotateCrop(ImageOp):
    """src/ops/rotatecrop.rs"""
    name = "rotatecrop"

    def __init__(self, _img=None):
        self.crop_top = self.crop_right = self.crop_bottom = self.crop_left = self.rotation = 0.0
        self.reset()

    def reset(self):
        self.input_ratio = 1.0
        self.output_size = None

    def _params(self):
        return _farr([self.crop_top, self.crop_right, self.crop_bottom, self.crop_left, self.rotation], 5)

    def calc_size(self, width, height, reverse):
        ow, oh = C.c_size_t(), C.c_size_t()
        _lib.check(lib().ipk_rotatecrop_calc_size(self._params(), self.input_ratio, width, height, int(reverse), C.byref(ow), C.byref(oh)),
                   "ipk_rotatecrop_calc_size")
        return ow.value, oh.value

    def transform_forward(self, width, height):
        if self.output_size is not None:
            return self.output_size
        # `width as f32 / height as f32` (rotatecrop.rs:71)
        self.input_ratio = float(np.float32(width) / np.float32(height)) if height else float("inf")
        return self.calc_size(width, height, False)

    def transform_reverse(self, width, height):
        self.output_size = (width, height)
        return self.calc_size(width, height, True)

    def run(self, pipeline, buf):
        ow, oh = C.c_size_t(), C.c_size_t()
        rc = _lib.check(lib().ipk_rotatecrop(_ptr(buf.data), buf.width, buf.height, buf.colors, self._params(), None,
                                             C.byref(ow), C.byref(oh), _stream()), "ipk_rotatecrop")
        if rc == IPK_NOOP:
            return buf
        out = OpBuffer.new(ow.value, oh.value, buf.colors, buf.monochrome)
        _lib.check(lib().ipk_rotatecrop(_ptr(buf.data), buf.width, buf.height, buf.colors, self._params(), _ptr(out.data),
                                        C.byref(ow), C.byref(oh), _stream()), "ipk_rotatecrop")
        return out


class OpToLab(ImageOp):
    """src/ops/colorspaces.rs:5-113"""
    name = "to_lab"

    def __init__(self, img):
        if isinstance(img, RawImage):
            cm = SRGB_D65_43 if img.cam_to_xyz_normalized is None else np.asarray(img.cam_to_xyz_normalized, np.float32)
            self.cam_to_xyz_normalized = cm.reshape(3, 4)
            self.cam_to_xyz = self.cam_to_xyz_normalized if img.cam_to_xyz is None else np.asarray(img.cam_to_xyz, np.float32).reshape(3, 4)
            self.xyz_to_cam = xyz_d65_34() if img.xyz_to_cam is None else np.asarray(img.xyz_to_cam, np.float32).reshape(4, 3)
            # colorspaces.rs:33-41: normalize_wbs(wb_coeffs), or normalize_wbs(neutralwb()) when any of wb[0..2] is not normal
            wb = np.asarray(img.wb_coeffs, np.float32)
            tiny = np.finfo(np.float32).tiny
            normal = bool(np.all(np.isfinite(wb[:3]) & (np.abs(wb[:3]) >= tiny)))
            if not normal:
                if img.neutralwb is None:
                    raise IpkError("OpToLab: as-shot wb_coeffs are not normal and the RawImage carries no neutralwb")
                wb = np.asarray(img.neutralwb, np.float32)
            out = (C.c_float * 4)()
            _lib.check(lib().ipk_normalize_wbs(_farr(wb, 4), out), "ipk_normalize_wbs")
            self.wb_coeffs = [float(v) for v in out]
        else:
            self.cam_to_xyz_normalized = self.cam_to_xyz = SRGB_D65_43
            self.xyz_to_cam = xyz_d65_34()
            self.wb_coeffs = [1.0, 1.0, 1.0, 0.0]

    def set_temp(self, temp, tint):
        """colorspaces.rs:59-70 (host-side)"""
        wb = (C.c_float * 4)()
        _lib.check(lib().ipk_tolab_set_temp(_farr(self.xyz_to_cam, 12), float(temp), float(tint), wb), "ipk_tolab_set_temp")
        self.wb_coeffs = list(wb)

    def get_temp(self):
        """colorspaces.rs:72-84 (host-side)"""
        t, ti = C.c_float(), C.c_float()
        _lib.check(lib().ipk_tolab_get_temp(_farr(self.cam_to_xyz, 12), _farr(self.wb_coeffs, 4), C.byref(t), C.byref(ti)), "ipk_tolab_get_temp")
        return t.value, ti.value

    def run(self, pipeline, buf):
        out = OpBuffer.new(buf.width, buf.height, 3, buf.monochrome)
        _lib.check(lib().ipk_tolab(_ptr(buf.data), buf.width, buf.height, int(buf.monochrome), _farr(self.wb_coeffs, 4),
                                   _farr(self.cam_to_xyz_normalized, 12), _ptr(out.data), _stream()), "ipk_tolab")
        return out


class OpBaseCurve(ImageOp):
    """src/ops/curves.rs:6-50"""
    name = "basecurve"

    def __init__(self, img):
        self.exposure = 0.0
        self.points = [(0.50, 0.60)] if isinstance(img, RawImage) else []

    def run(self, pipeline, buf):
        pts, n = _points(self.points)
        out = OpBuffer.new(buf.width, buf.height, 3, buf.monochrome)
        rc = _lib.check(lib().ipk_basecurve(_ptr(buf.data), buf.width, buf.height, self.exposure, pts, n, _ptr(out.data), _stream()),
                        "ipk_basecurve")
        return buf if rc == IPK_NOOP else out


class OpFromLab(ImageOp):
    """src/ops/colorspaces.rs:115-138"""
    name = "from_lab"

    def __init__(self, _img=None):
        pass

    def run(self, pipeline, buf):
        out = OpBuffer.new(buf.width, buf.height, 3, buf.monochrome)
        _lib.check(lib().ipk_fromlab(_ptr(buf.data), buf.width, buf.height, _ptr(out.data), _stream()), "ipk_fromlab")
        return out


class OpGamma(ImageOp):
    """src/ops/gamma.rs"""
    name = "gamma"

    def __init__(self, _img=None):
        pass

    def run(self, pipeline, buf):
        out = OpBuffer.new(buf.width, buf.height, buf.colors, buf.monochrome)
        rc = _lib.check(lib().ipk_gamma(_ptr(buf.data), buf.width, buf.height, buf.colors, int(pipeline.settings.linear),
                                        _ptr(out.data), _stream()), "ipk_gamma")
        return buf if rc == IPK_NOOP else out


class OpTransform(ImageOp):
    """src/ops/transform.rs:14-85"""
    name = "transform"
    _FROM_ORIENTATION = {
        OR_NORMAL: (ROT_NORMAL, False, False), OR_UNKNOWN: (ROT_NORMAL, False, False),
        OR_VFLIP: (ROT_NORMAL, False, True), OR_HFLIP: (ROT_NORMAL, True, False),
        OR_ROT180: (ROT_180, False, False), OR_TRANSPOSE: (ROT_90, False, True),
        OR_ROT90: (ROT_90, False, False), OR_ROT270: (ROT_270, False, False),
        OR_TRANSVERSE: (ROT_270, True, False),
    }

    def __init__(self, img):
        if isinstance(img, RawImage):
            self.rotation, self.fliph, self.flipv = self._FROM_ORIENTATION[img.orientation]
        else:
            self.rotation, self.fliph, self.flipv = ROT_NORMAL, False, False

    def transform_forward(self, width, height):
        return (height, width) if self.rotation in (ROT_90, ROT_270) else (width, height)

    transform_reverse = transform_forward

    def run(self, pipeline, buf):
        out = OpBuffer.new(buf.width, buf.height, 3, buf.monochrome)
        ow, oh = C.c_size_t(), C.c_size_t()
        rc = _lib.check(lib().ipk_transform(_ptr(buf.data), buf.width, buf.height, self.rotation, int(self.fliph), int(self.flipv),
                                            _ptr(out.data), C.byref(ow), C.byref(oh), _stream()), "ipk_transform")
        if rc == IPK_NOOP:
            return buf
        out.width, out.height = ow.value, oh.value
        return out


class PipelineOps:
    """src/pipeline.rs:153-179"""
    ORDER = ["gofloat", "demosaic", "rotatecrop", "tolab", "basecurve", "fromlab", "gamma", "transform"]

    def __init__(self, img):
        self.gofloat = OpGoFloat(img)
        self.demosaic = OpDemosaic(img)
        self.rotatecrop = OpRotateCrop(img)
        self.tolab = OpToLab(img)
        self.basecurve = OpBaseCurve(img)
        self.fromlab = OpFromLab(img)
        self.gamma = OpGamma(img)
        self.transform = OpTransform(img)

    def all_ops(self):
        return [getattr(self, n) for n in self.ORDER]


class PipelineCache:
    """Pipeline::new_cache(size) (src/pipeline.rs:43,258-260): byte-budgeted LRU of device OpBuffers keyed by op hash."""

    def __init__(self, max_bytes):
        h = C.c_void_p()
        _lib.check(lib().ipk_cache_new(int(max_bytes), C.byref(h)), "ipk_cache_new")
        self.handle = h

    def close(self):
        if self.handle:
            lib().ipk_cache_free(self.handle)
            self.handle = None

    __del__ = close

    def clear(self):
        _lib.check(lib().ipk_cache_clear(self.handle), "ipk_cache_clear")

    def contains(self, key: bytes) -> bool:
        return bool(_lib.check(lib().ipk_cache_contains(self.handle, key), "ipk_cache_contains"))

    def stats(self):
        b, e = C.c_size_t(), C.c_size_t()
        hi, mi, ev = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _lib.check(lib().ipk_cache_stats(self.handle, C.byref(b), C.byref(e), C.byref(hi), C.byref(mi), C.byref(ev)), "ipk_cache_stats")
        return dict(bytes=b.value, entries=e.value, hits=hi.value, misses=mi.value, evictions=ev.value)

    def get(self, key: bytes):
        """Copy of the memoised buffer as a numpy array (tests), or None."""
        p, w, h, c, m = C.c_void_p(), C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_int()
        rc = _lib.check(lib().ipk_cache_get(self.handle, key, C.byref(p), C.byref(w), C.byref(h), C.byref(c), C.byref(m)), "ipk_cache_get")
        if rc == IPK_NOOP:
            return None
        out = np.empty(w.value * h.value * c.value, np.float32)
        _lib.check(lib().ipk_stream_sync(_stream()), "ipk_stream_sync")
        _lib.check(lib().ipk_memcpy_d2h(out.ctypes.data, p, out.nbytes, None), "ipk_memcpy_d2h")
        return out.reshape(h.value, w.value, c.value)


class Pipeline:
    """src/pipeline.rs:246-470.  Settings (de)serialisation is out of scope.

    `run()` hands the whole op list to the C driver (ipk_pipeline_run), which uses the fused
    raw->sRGB kernel when every op between gofloat and gamma allows it and the staged kernels
    otherwise.  `run_ops()` walks the Python op objects one by one exactly like the reference's
    loop (pipeline.rs:364-372) and exists so tests can compare the two."""

    def __init__(self, img):
        self.globals = PipelineGlobals(img)
        self.ops = PipelineOps(img)
        self.allow_fused = True
        self.last_used_fused = None
        self.last_ops_run = None              # bit i set when op i executed in the last run (0 = served from the cache)
        self.source_id = 0                    # extension of the hash chain: identifies the frame inside a shared PipelineCache
        self.schedule = 0                     # ipk_pipeline_desc.schedule (IPK_SCHED_AUTO / IPK_SCHED_SPLIT): how a fused launch shares the rows out

    @staticmethod
    def new_from_source(img):
        init()
        return Pipeline(img)

    # -- size negotiation (pipeline.rs:314-338) in Python over the op objects
    def negotiate(self):
        ops = self.ops.all_ops()
        for op in ops:
            op.reset()
        w, h = self.globals.image.width, self.globals.image.height
        for op in ops:
            w, h = op.transform_forward(w, h)
        s, nw, nh = C.c_float(), C.c_size_t(), C.c_size_t()
        lib().ipk_calculate_scaling_total(w, h, self.globals.settings.maxwidth, self.globals.settings.maxheight,
                                          C.byref(s), C.byref(nw), C.byref(nh))
        w, h = nw.value, nh.value
        final = (w, h)
        for op in reversed(ops):
            w, h = op.transform_reverse(w, h)
        self.globals.settings.demosaic_width, self.globals.settings.demosaic_height = w, h
        return (w, h), final

    def run_ops(self) -> OpBuffer:
        self.negotiate()
        buf = None
        for op in self.ops.all_ops():
            buf = op.run(self.globals, buf)
        return buf

    # -- descriptor for the C driver
    def desc(self) -> PipelineDesc:
        img, ops, st = self.globals.image, self.ops, self.globals.settings
        d = PipelineDesc()
        if isinstance(img, RawImage):
            d.src_type = SRC_F32 if img.is_float else SRC_U16
            d.cpp = img.cpp
        else:
            d.src_type = SRC_RGB8 if img.bits == 8 else SRC_RGB16
            d.cpp = 3
        d.width, d.height = img.width, img.height
        d.is_cfa = int(ops.gofloat.is_cfa)
        d.cfa = ops.demosaic.cfa.encode()
        d.crop_top, d.crop_right, d.crop_bottom, d.crop_left = ops.gofloat.crop_top, ops.gofloat.crop_right, ops.gofloat.crop_bottom, ops.gofloat.crop_left
        d.blacklevels[:] = ops.gofloat.blacklevels
        d.whitelevels[:] = ops.gofloat.whitelevels
        rc = ops.rotatecrop
        d.rotatecrop[:] = [rc.crop_top, rc.crop_right, rc.crop_bottom, rc.crop_left, rc.rotation]
        d.cam_to_xyz_normalized[:] = [float(v) for v in np.asarray(ops.tolab.cam_to_xyz_normalized, np.float32).ravel()]
        d.wb_coeffs[:] = ops.tolab.wb_coeffs
        d.exposure = ops.basecurve.exposure
        p = np.asarray(ops.basecurve.points, np.float32).ravel()
        d.npoints = p.size // 2
        for i, v in enumerate(p):
            d.points[i] = v
        d.rotation, d.fliph, d.flipv = ops.transform.rotation, int(ops.transform.fliph), int(ops.transform.flipv)
        d.maxwidth, d.maxheight = st.maxwidth, st.maxheight
        d.linear = int(st.linear)
        d.allow_fused = int(self.allow_fused)
        d.use_fastpath = int(st.use_fastpath)
        d.schedule = int(self.schedule)
        return d

    def sizes(self):
        d = self.desc()
        a, b, c, e = C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_size_t()
        _lib.check(lib().ipk_pipeline_sizes(C.byref(d), C.byref(a), C.byref(b), C.byref(c), C.byref(e)), "ipk_pipeline_sizes")
        return (a.value, b.value), (c.value, e.value)

    def hashes(self, out_type=OUT_F32):
        """The eight chained op hashes of pipeline.rs:342-361 (host-only)."""
        d = self.desc()
        out = C.create_string_buffer(256)
        _lib.check(lib().ipk_pipeline_hashes(C.byref(d), out_type, int(self.source_id), out), "ipk_pipeline_hashes")
        return [out.raw[32 * i: 32 * i + 32] for i in range(8)]

    def _run(self, out_type, out: Optional[torch.Tensor] = None, cache: Optional[PipelineCache] = None):
        d = self.desc()
        _, (fw, fh) = self.sizes()
        dt = {OUT_F32: torch.float32, OUT_U8: torch.uint8, OUT_U16: torch.int16}[out_type]
        if out is None:
            out = torch.empty(fw * fh * 3, dtype=dt, device="cuda")
        used = C.c_int(0)
        if cache is None:
            _lib.check(lib().ipk_pipeline_run(C.byref(d), _ptr(self.globals.image.data), _ptr(out), out_type, C.byref(used), _stream()),
                       "ipk_pipeline_run")
            self.last_ops_run = 0xFF
        else:
            mask = C.c_int(0)
            _lib.check(lib().ipk_pipeline_run_cached(C.byref(d), _ptr(self.globals.image.data), int(self.source_id), cache.handle, out_type,
                                                     _ptr(out), C.byref(mask), C.byref(used), _stream()), "ipk_pipeline_run_cached")
            self.last_ops_run = mask.value
        self.last_used_fused = bool(used.value)
        return out, fw, fh

    def run_timed(self, out_type=OUT_F32):
        """do_timing! (pipeline.rs:68-80): one run with per-stage hipEvent times; returns (output tensor, [(stage name, ms), ...])"""
        _lib.check(lib().ipk_timing_begin(), "ipk_timing_begin")
        data, _w, _h = self._run(out_type)
        arr = (_lib.StageTime * 16)()
        n = C.c_int(0)
        _lib.check(lib().ipk_timing_end(arr, 16, C.byref(n)), "ipk_timing_end")
        return data, [(arr[i].name.decode(), float(arr[i].ms)) for i in range(min(n.value, 16))]

    def run(self, cache: Optional[PipelineCache] = None, out: Optional[torch.Tensor] = None) -> OpBuffer:
        """Pipeline::run(cache) (pipeline.rs:311-375)"""
        data, w, h = self._run(OUT_F32, out, cache)
        return OpBuffer(w, h, 3, False, data)

    def output_8bit(self, cache: Optional[PipelineCache] = None):
        """Pipeline::output_8bit slow path (pipeline.rs:404-421): returns (width, height, uint8 device tensor)."""
        data, w, h = self._run(OUT_U8, None, cache)
        return w, h, data

    def output_16bit(self, cache: Optional[PipelineCache] = None):
        """Pipeline::output_16bit slow path (pipeline.rs:451-468): uint16 bits in an int16 device tensor."""
        data, w, h = self._run(OUT_U16, None, cache)
        return w, h, data


class FusedPlan:
    """A prepared ipk_fused_params: build once, launch many times (keeps the per-launch host cost to one C call)."""

    def __init__(self, *, width, height, owidth=None, x=0, y=0, is_float=True, black0=0.0, white0=1.0,
                 cfa="RGGB", wb_coeffs=(1.0, 1.0, 1.0, float("nan")), cam_to_xyz_normalized=None, exposure=0.0,
                 points=((0.5, 0.6),), linear=False, out_type=OUT_F32, band=None, schedule=0):
        init()
        p = FusedParams()
        p.schedule = int(schedule)                  # ipk_schedule: IPK_SCHED_AUTO / IPK_SCHED_SPLIT (results do not depend on it)
        p.src_type = SRC_F32 if is_float else SRC_U16
        p.owidth = owidth if owidth is not None else width
        p.x, p.y, p.width, p.height = x, y, width, height
        p.black0, p.white0 = black0, white0
        p.cfa = cfa.encode()
        p.wb_coeffs[:] = list(wb_coeffs)
        cm = SRGB_D65_43 if cam_to_xyz_normalized is None else np.asarray(cam_to_xyz_normalized, np.float32)
        p.cam_to_xyz_normalized[:] = [float(v) for v in cm.ravel()]
        p.exposure = exposure
        pts = np.asarray(points, np.float32).ravel()
        p.npoints = pts.size // 2
        for i, v in enumerate(pts):
            p.points[i] = v
        p.linear = int(linear)
        p.out_type = out_type
        self.rows = height
        if band is not None:
            p.band_src_row0, p.band_src_rows, p.band_out_row0, p.band_out_rows = band
            self.rows = band[3]
        self.params = p
        self.width = width
        self.out_dtype = {OUT_F32: torch.float32, OUT_U8: torch.uint8, OUT_U16: torch.int16}[out_type]
        self._fn = lib().ipk_raw_to_srgb
        self._ref = C.byref(p)

    def new_output(self):
        return torch.empty(self.rows * self.width * 3, dtype=self.out_dtype, device="cuda")

    def run(self, src: torch.Tensor, out: torch.Tensor, stream=None):
        rc = self._fn(self._ref, src.data_ptr(), out.data_ptr(), stream if stream is not None else torch.cuda.current_stream().cuda_stream)
        if rc < 0:
            _lib.check(rc, "ipk_raw_to_srgb")
        return out


    def probe(self, src: torch.Tensor, out_f32: torch.Tensor, stream=None):
        """ipk_stream_probe: the fused kernel's memory skeleton (loads, OpGoFloat, demosaic, staging, stores) without the point-wise stages;
        out_f32 receives the demosaiced R, G, B as rows*width*3 f32"""
        # the probe always writes rows*width*3 f32, whatever the plan's out_type: a u8 / u16 buffer from new_output() would be overrun
        assert out_f32.is_cuda and out_f32.is_contiguous() and out_f32.dtype == torch.float32 and out_f32.numel() >= self.rows * self.width * 3, \
            "probe() needs a contiguous cuda float32 buffer of at least rows*width*3 elements"
        rc = lib().ipk_stream_probe(self._ref, src.data_ptr(), out_f32.data_ptr(), stream if stream is not None else torch.cuda.current_stream().cuda_stream)
        if rc < 0:
            _lib.check(rc, "ipk_stream_probe")
        return out_f32


class FusedBatchPlan:
    """ipk_raw_to_srgb_batch over fixed lists of same-shaped frames: the pointer arrays are built once, run() is one C call (one persistent
    launch per 64 frames where the kernel has a batch variant)."""

    def __init__(self, plan: "FusedPlan", srcs, outs):
        assert len(srcs) == len(outs)
        self.plan, self.srcs, self.outs = plan, list(srcs), list(outs)          # keeps the tensors alive
        n = len(self.srcs)
        self._n = n
        self._s = (C.c_void_p * max(n, 1))(*[t.data_ptr() for t in self.srcs])
        self._d = (C.c_void_p * max(n, 1))(*[t.data_ptr() for t in self.outs])

    def run(self, stream=None):
        rc = lib().ipk_raw_to_srgb_batch(self.plan._ref, self._s, self._d, self._n, stream if stream is not None else torch.cuda.current_stream().cuda_stream)
        if rc < 0:
            _lib.check(rc, "ipk_raw_to_srgb_batch")
        return self.outs


def raw_to_srgb(src: torch.Tensor, *, out: Optional[torch.Tensor] = None, **kw):
    """The fused kernel through the C ABI (ipk_raw_to_srgb); `band` = (src_row0, src_rows, out_row0, out_rows)."""
    plan = FusedPlan(**kw)
    if out is None:
        out = plan.new_output()
    return plan.run(src, out)
