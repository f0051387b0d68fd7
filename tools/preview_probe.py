#!/usr/bin/env python
"""Development tool: the PREVIEW call -- output_8bit / output_16bit / run under a size limit (OpDemosaic's scaled branch, then tolab..gamma, then the
quantise loop) -- on BASELINE.json configs[4]'s frame (8640x5760 X-Trans -> 2160x1440) and on a 24 MP Bayer frame -> 1500x1000; HIP-event mean per call.
usage (GPU box): [IPK_SO_OVERRIDE=old.so] tools/preview_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import imagepipe_amd as ipa, util, bench
ipa.init(0)
for name, W, H, cfa, mw in (("X-Trans 50 MP -> 2160x1440", 8640, 5760, bench.XTRANS, 2160), ("RGGB 24 MP -> 1500x1000", 6000, 4000, "RGGB", 1500)):
    src = bench.synth_frame(torch, H, W, "noise", 5).to(torch.int16).reshape(-1).contiguous()
    img = ipa.RawImage(width=W, height=H, data=src, cfa=cfa, blacklevels=[util.BLACK] * 4, whitelevels=[util.WHITE] * 4, wb_coeffs=util.WB,
                       cam_to_xyz_normalized=util.cam_matrix())
    pipe = ipa.Pipeline.new_from_source(img)
    pipe.globals.settings.maxwidth = mw
    for what, fn in (("run (f32)", pipe.run), ("output_8bit", pipe.output_8bit), ("output_16bit", pipe.output_16bit)):
        for _ in range(50): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(300): fn()
        e1.record(); torch.cuda.synchronize()
        print("%-28s %-13s %.1f us per call" % (name, what, e0.elapsed_time(e1) / 300 * 1e3))
