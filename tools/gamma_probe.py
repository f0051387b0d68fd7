#!/usr/bin/env python
"""Development tool: ipk_gamma alone on a 100 MP 3-channel buffer (HIP-event mean over 20 launches) -- for grid-shape sweeps of k_gamma."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import imagepipe_amd as ipa
ipa.init(0)
L = ipa.lib()
W = H = 10000
g = torch.Generator(device="cuda"); g.manual_seed(3)
src = torch.rand(W * H * 3, generator=g, device="cuda", dtype=torch.float32) * 1.2 - 0.1
dst = torch.empty_like(src)
st = torch.cuda.current_stream().cuda_stream
def run():
    rc = L.ipk_gamma(src.data_ptr(), W, H, 3, 0, dst.data_ptr(), st); assert rc == 0, rc
for _ in range(30): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print("k_gamma %.4f ms  %.0f GB/s  frac %.3f" % (ms, 2.4e9 / ms / 1e6, 2.4e9 / ms / 1e6 / 8000))
