"""Development tool (GPU box): host-side cost of one Pipeline.run / ipk_pipeline_run call -- a tiny frame, so the GPU work is negligible --
next to config 5's per-step time.   python tools/host_overhead.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import imagepipe_amd as ipa
import util

XT = "GGRGGBGGBGGRBRGRBGGGBGGRGGRGGBRBGBRG"


def per_call(fn, n):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6


def make(W, H, cfa, maxw):
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    src = torch.randint(0, 16384, (H * W,), device="cuda", generator=g, dtype=torch.int32).to(torch.float32)
    img = ipa.RawImage(width=W, height=H, data=src, cfa=cfa, is_float=True, blacklevels=[util.BLACK] * 4, whitelevels=[util.WHITE] * 4,
                       wb_coeffs=util.WB, cam_to_xyz_normalized=util.cam_matrix())
    pipe = ipa.Pipeline.new_from_source(img)
    pipe.globals.settings.maxwidth = maxw
    out = pipe.run()
    return pipe, out


ipa.init(0)
for name, (W, H, cfa, maxw) in {"tiny X-Trans 144x96 -> 36 wide": (144, 96, XT, 36), "tiny RGGB 256x64 fused": (256, 64, "RGGB", 0),
                                "config 5 (8640x5760 X-Trans -> 2160x1440)": (8640, 5760, XT, 2160), "24 MP RGGB fused": (6000, 4000, "RGGB", 0)}.items():
    pipe, out = make(W, H, cfa, maxw)
    issue, total = per_call(lambda: pipe.run(out=out.data), 2000 if W < 1000 else 300)
    print("%-45s host issue %.1f us per call, with the GPU drained %.1f us per call" % (name, issue, total))
