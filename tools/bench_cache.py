#!/usr/bin/env python
"""Times Pipeline::run(Some(cache)) on the device: cold run, cache hit, and resumption after edits (24 MP full-size and a
1500-px preview of the same frame)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import imagepipe_amd as ipa
import util


def wall(fn, n=5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


def main():
    ipa.init(0)
    h, w = 4000, 6000
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    data = torch.randint(0, 16384, (h * w,), generator=g, device="cuda", dtype=torch.int32).to(torch.int16)
    res = {}
    for name, maxw in (("full_24MP", 0), ("preview_1500", 1500)):
        img = ipa.RawImage(width=w, height=h, data=data, cfa="RGGB", blacklevels=[util.BLACK] * 4, whitelevels=[util.WHITE] * 4,
                           wb_coeffs=util.WB, cam_to_xyz_normalized=util.cam_matrix())
        pipe = ipa.Pipeline.new_from_source(img)
        pipe.globals.settings.maxwidth = maxw
        cache = ipa.PipelineCache(64 << 30)
        r = {}
        t0 = time.perf_counter(); pipe.run(cache); torch.cuda.synchronize(); r["cold_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
        r["cold_ops_run"] = hex(pipe.last_ops_run); r["cold_fused"] = pipe.last_used_fused
        r["hit_ms"] = round(wall(lambda: pipe.run(cache)), 3)
        def edit_curve():
            pipe.ops.basecurve.exposure += 0.01
            pipe.run(cache)
        r["edit_curve_ms"] = round(wall(edit_curve), 3); r["edit_curve_ops"] = hex(pipe.last_ops_run)
        def edit_wb():
            pipe.ops.tolab.wb_coeffs[0] += 0.01
            pipe.run(cache)
        r["edit_wb_ms"] = round(wall(edit_wb), 3); r["edit_wb_ops"] = hex(pipe.last_ops_run)
        r["nocache_ms"] = round(wall(lambda: pipe.run()), 3)
        r["cache"] = cache.stats()
        res[name] = r
        cache.close()
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
