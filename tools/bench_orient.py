#!/usr/bin/env python
"""Times Pipeline::run / output_8bit / output_16bit of a 100 MP (or H W from argv) Bayer frame for the Normal orientation, a flip and
a 90-degree rotation (the portrait shot): the fused launch + the permutation, which for the 8/16-bit outputs runs on the quantised image."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import imagepipe_amd as ipa
import util


def timeit(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize(); return round((time.perf_counter() - t0) / n * 1e3, 4)


def main():
    ipa.init(0)
    h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (10000, 10000)
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    data = torch.randint(0, 16384, (h * w,), generator=g, device="cuda", dtype=torch.int32).to(torch.int16)
    res = {}
    for name, (rot, fh) in {"normal": (0, False), "hflip": (0, True), "rot90": (1, False), "rot270": (3, False)}.items():
        img = ipa.RawImage(width=w, height=h, data=data, cfa="RGGB", blacklevels=[util.BLACK] * 4, whitelevels=[util.WHITE] * 4,
                           wb_coeffs=util.WB, cam_to_xyz_normalized=util.cam_matrix())
        pipe = ipa.Pipeline.new_from_source(img)
        pipe.ops.transform.rotation = rot; pipe.ops.transform.fliph = fh
        out = pipe.run()
        res[name] = {"f32_ms": timeit(lambda: pipe.run(out=out.data)), "u8_ms": timeit(lambda: pipe.output_8bit()), "u16_ms": timeit(lambda: pipe.output_16bit()),
                     "fused": pipe.last_used_fused}
        del pipe, out
        torch.cuda.empty_cache()
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
