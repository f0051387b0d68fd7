import sys, time, ctypes as C, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import imagepipe_amd as ipa, util
ipa.init(0); L = ipa.lib(); L.ipk_host_alloc.restype = C.c_void_p
h, w = 4000, 6000
raw = util.noise_u16(util.SEED + 2, h, w)
img = ipa.RawImage(width=w, height=h, data=ipa.upload_u16(raw), cfa="RGGB", blacklevels=[util.BLACK] * 4, whitelevels=[util.WHITE] * 4, wb_coeffs=util.WB, cam_to_xyz_normalized=util.cam_matrix())
d = ipa.Pipeline.new_from_source(img).desc()
for out_type, ob in ((0, h * w * 12), (1, h * w * 3)):
    sp = [L.ipk_host_alloc(h * w * 2) for _ in range(4)]; dp = [L.ipk_host_alloc(ob) for _ in range(4)]
    for p in sp: C.memmove(p, raw.ctypes.data, raw.nbytes)
    for p in dp: C.memset(p, 0, ob)
    for n in (1, 4, 16, 64):
        srcs = (C.c_void_p * n)(*[sp[i % 4] for i in range(n)]); dsts = (C.c_void_p * n)(*[dp[i % 4] for i in range(n)])
        L.ipk_host_pipeline_run_batch(C.byref(d), srcs, dsts, n, out_type, None)
        t0 = time.perf_counter(); assert L.ipk_host_pipeline_run_batch(C.byref(d), srcs, dsts, n, out_type, None) == 0; dt = time.perf_counter() - t0
        print("out", out_type, "n", n, "total ms", round(dt * 1e3, 2), "per frame", round(dt * 1e3 / n, 3))
