#!/usr/bin/env python
"""Development tool: one staged entry point alone on a 100 MP buffer (HIP-event mean over 20 launches after 30 warm-ups), for grid-shape sweeps.
usage: tools/stage_probe.py gamma|tolab|chain|demosaic [W [H]]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import imagepipe_amd as ipa, util
ipa.init(0)
L = ipa.lib()
which = sys.argv[1]
W = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
H = int(sys.argv[3]) if len(sys.argv) > 3 else W
g = torch.Generator(device="cuda"); g.manual_seed(3)
st = torch.cuda.current_stream().cuda_stream
cm = (C.c_float * 12)(*[float(v) for v in util.cam_matrix().ravel()]); wb = (C.c_float * 4)(*util.WB)
if which == "gamma":
    src = torch.rand(W * H * 3, generator=g, device="cuda") * 1.2 - 0.1; dst = torch.empty_like(src); nbytes = 24.0 * W * H
    run = lambda: L.ipk_gamma(src.data_ptr(), W, H, 3, 0, dst.data_ptr(), st)
elif which == "tolab":
    src = torch.rand(W * H * 4, generator=g, device="cuda"); src.view(-1, 4)[:, 3] = 0; dst = torch.empty(W * H * 3, device="cuda"); nbytes = 28.0 * W * H
    run = lambda: L.ipk_tolab(src.data_ptr(), W, H, 0, wb, cm, dst.data_ptr(), st)
elif which == "chain":
    src = torch.rand(W * H * 4, generator=g, device="cuda"); src.view(-1, 4)[:, 3] = 0; dst = torch.empty(W * H * 3, device="cuda"); nbytes = 28.0 * W * H
    pts = (C.c_float * 2)(0.5, 0.6)
    run = lambda: L.ipk_pointwise_chain(src.data_ptr(), W, H, 0, wb, cm, C.c_float(0.0), pts, 1, 0, dst.data_ptr(), st)
else:
    src = torch.rand(W * H, generator=g, device="cuda"); dst = torch.empty(W * H * 4, device="cuda"); nbytes = 20.0 * W * H
    run = lambda: L.ipk_demosaic_full(src.data_ptr(), W, H, b"RGGB", dst.data_ptr(), st)
for _ in range(30): assert run() >= 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
N = 20 if W * H > 2e7 else 400
for _ in range(N): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / N
print("%s %.4f ms  %.0f GB/s  frac %.3f" % (which, ms, nbytes / ms / 1e6, nbytes / ms / 1e6 / 8000))
