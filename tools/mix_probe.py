#!/usr/bin/env python
"""Development tool: ipk_copy_probe (1:1) and ipk_mix_probe (1:3 read:write) on 1.2 GB / 0.4 + 1.2 GB, HIP-event mean over 20 launches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import imagepipe_amd as ipa
ipa.init(0); L = ipa.lib()
st = torch.cuda.current_stream().cuda_stream
def t(fn):
    for _ in range(10): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / 20
for mp in (24, 100, 400):
    n = mp * 1000 * 1000 * 4 // 4096 * 4096
    a = torch.ones(n // 4, device="cuda"); c = torch.empty(3 * n // 4, device="cuda")
    ms = t(lambda: L.ipk_mix_probe(a.data_ptr(), c.data_ptr(), n, st))
    b = torch.empty(4 * n // 4 // 2 * 2, device="cuda")[: n // 4 * 2]
    a2 = torch.ones(n // 2, device="cuda"); b2 = torch.empty_like(a2)
    msc = t(lambda: L.ipk_copy_probe(a2.data_ptr(), b2.data_ptr(), 2 * n, st))
    print("%d MP: mix 1:3 %.4f ms = %.0f GB/s (%.3f of peak)   copy 1:1 of the same bytes %.4f ms = %.0f GB/s (%.3f)" % (mp, ms, 4 * n / ms / 1e6, 4 * n / ms / 1e6 / 8000, msc, 4 * n / msc / 1e6, 4 * n / msc / 1e6 / 8000))
    del a, c, a2, b2, b
