#!/usr/bin/env python
"""Times the BASELINE.json configurations that are not the headline bench line (device-resident, HIP events):
   C2  24 MP RGGB, fused and staged;  C3 100 MP fused (same as bench.py);  C5 8640x5760 X-Trans -> 2160x1440 (staged)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import imagepipe_amd as ipa
import util

XTRANS = "GGRGGBGGBGGRBRGRBGGGBGGRGGRGGBRBGBRG"


def timeit(fn, n=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def frame(h, w, seed, is_float):
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    v = torch.randint(0, 16384, (h * w,), generator=g, device="cuda", dtype=torch.int32)
    return v.to(torch.float32) if is_float else v.to(torch.int16)


def main():
    ipa.init(0)
    res = {}
    for name, (h, w, cfa, maxw, is_float) in {"C2_24MP_rggb_f32": (4000, 6000, "RGGB", 0, True), "C2_24MP_rggb_u16": (4000, 6000, "RGGB", 0, False),
                                                 "C3_100MP_rggb_f32": (10000, 10000, "RGGB", 0, True),
                                                 "C5_50MP_xtrans_to_2160": (5760, 8640, XTRANS, 2160, True),
                                                 "C5b_50MP_xtrans_fullres": (5760, 8640, XTRANS, 0, True)}.items():
        if os.environ.get("ONLY") and os.environ["ONLY"] not in name:
            continue
        img = ipa.RawImage(width=w, height=h, data=frame(h, w, 7, is_float), cfa=cfa, is_float=is_float, blacklevels=[util.BLACK] * 4,
                           whitelevels=[util.WHITE] * 4, wb_coeffs=util.WB, cam_to_xyz_normalized=util.cam_matrix())
        pipe = ipa.Pipeline.new_from_source(img)
        pipe.globals.settings.maxwidth = maxw
        out = pipe.run()
        r = {"out": [out.width, out.height], "fused": pipe.last_used_fused, "ms": round(timeit(lambda: pipe.run(out=out.data)), 4)}
        r["MP_per_s_in"] = round(h * w / 1e6 / (r["ms"] * 1e-3), 1)
        if pipe.last_used_fused:
            pipe.allow_fused = False
            o2 = pipe.run()
            r["staged_ms"] = round(timeit(lambda: pipe.run(out=o2.data), n=5), 4)
            assert torch.equal(o2.data.view(torch.int32), out.data.view(torch.int32)), "fused != staged"
            r["staged_u8_ms"] = round(timeit(lambda: pipe.output_8bit(), n=3, warm=1), 4)
            r["staged_u16_ms"] = round(timeit(lambda: pipe.output_16bit(), n=3, warm=1), 4)
        res[name] = r
        del pipe, img, out
        torch.cuda.empty_cache()
    # raster sources: 24 MP RGB8 / RGB16 -> output_8bit / output_16bit, full float pipe vs the integer fast path
    if not os.environ.get("ONLY") or "raster" in os.environ["ONLY"]:
        h, w = 4000, 6000
        rng = np.random.default_rng(11)
        for bits in (8, 16):
            img = rng.integers(0, 256 if bits == 8 else 65536, (h, w, 3)).astype(np.uint8 if bits == 8 else np.uint16)
            data = torch.from_numpy(img.ravel()).cuda() if bits == 8 else ipa.upload_u16(img)
            for maxw in (0, 1500):
                for fast in (False, True):
                    pipe = ipa.Pipeline.new_from_source(ipa.OtherImage(w, h, data, bits=bits))
                    pipe.globals.settings.maxwidth = maxw; pipe.globals.settings.use_fastpath = fast
                    key = "raster_24MP_rgb%d_maxw%d_%s" % (bits, maxw, "fastpath" if fast else "floatpipe")
                    res[key] = {"u8_ms": round(timeit(lambda: pipe.output_8bit(), n=5, warm=1), 4), "u16_ms": round(timeit(lambda: pipe.output_16bit(), n=5, warm=1), 4)}
                    del pipe
            del data
            torch.cuda.empty_cache()
    # PCIe-inclusive: the host-pointer form (upload 24 MP u16, fused kernel, download f32 RGB), pageable host memory
    import ctypes as C
    h, w = 4000, 6000
    raw = util.noise_u16(util.SEED + 2, h, w)
    img = ipa.RawImage(width=w, height=h, data=ipa.upload_u16(raw), cfa="RGGB", blacklevels=[util.BLACK] * 4, whitelevels=[util.WHITE] * 4,
                       wb_coeffs=util.WB, cam_to_xyz_normalized=util.cam_matrix())
    d = ipa.Pipeline.new_from_source(img).desc()
    out = np.empty(h * w * 3, np.float32)
    used = C.c_int(0)
    L = ipa.lib()
    L.ipk_host_pipeline_run(C.byref(d), raw.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), 0, C.byref(used))
    t0 = time.perf_counter()
    for _ in range(3):
        assert L.ipk_host_pipeline_run(C.byref(d), raw.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), 0, C.byref(used)) == 0
    dt = (time.perf_counter() - t0) / 3
    res["host_pcie_inclusive_24MP_u16_to_f32"] = {"ms": round(dt * 1e3, 2), "MP_per_s": round(h * w / 1e6 / dt, 1), "bytes_moved": h * w * 2 + h * w * 12}
    # the same through the three-stream batch driver, page-locked buffers: 8 frames, u16 -> f32 and u16 -> u8
    L.ipk_host_alloc.restype = C.c_void_p
    nb = 8
    for out_type, ob in ((0, h * w * 12), (1, h * w * 3)):
        sp = [L.ipk_host_alloc(h * w * 2) for _ in range(nb)]
        dp = [L.ipk_host_alloc(ob) for _ in range(nb)]
        for p_ in sp:
            C.memmove(p_, raw.ctypes.data, raw.nbytes)
        srcs = (C.c_void_p * nb)(*sp); dsts = (C.c_void_p * nb)(*dp)
        assert L.ipk_host_pipeline_run_batch(C.byref(d), srcs, dsts, nb, out_type, None) == 0
        t0 = time.perf_counter()
        assert L.ipk_host_pipeline_run_batch(C.byref(d), srcs, dsts, nb, out_type, None) == 0
        dt = (time.perf_counter() - t0) / nb
        # one frame at a time with the same page-locked buffers, for the overlap's share
        t0 = time.perf_counter()
        for i in range(nb):
            assert L.ipk_host_pipeline_run(C.byref(d), sp[i], dp[i], out_type, None) == 0
        dt1 = (time.perf_counter() - t0) / nb
        res["host_batch_pinned_24MP_u16_to_%s" % ("f32" if out_type == 0 else "u8")] = {
            "ms_per_frame": round(dt * 1e3, 3), "MP_per_s": round(h * w / 1e6 / dt, 1), "one_at_a_time_ms": round(dt1 * 1e3, 3),
            "GB_per_s_up_plus_down": round((h * w * 2 + ob) / dt / 1e9, 1)}
        for p_ in sp + dp:
            L.ipk_host_free(p_)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
