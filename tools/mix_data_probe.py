#!/usr/bin/env python
"""Development tool: the flat 1:3 mix probe (ipk_mix_probe) and the 1:1 copy probe on CONSTANT data and on the bench's noise / photo frames: the
memory system's rate depends on what the bytes are (the socket is power-capped), so a ceiling measured on constant data is not the ceiling of a noise frame.
usage: tools/mix_data_probe.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import imagepipe_amd as ipa, util
import bench
ipa.init(0)
L = ipa.lib()
st = torch.cuda.current_stream().cuda_stream
W = H = 10000
n_in = W * H * 4 // 4096 * 4096
dst = torch.empty(n_in * 3 // 4, dtype=torch.float32, device="cuda")
cp = torch.empty(n_in // 4, dtype=torch.float32, device="cuda")


def t(fn, n=20):
    for _ in range(30): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for kind in ("constant", "noise", "photo", "smooth"):
    if kind == "constant":
        src = torch.empty(n_in // 4, dtype=torch.float32, device="cuda").fill_(1.0)
    else:
        src = bench.synth_frame(torch, H, W, kind, 7).to(torch.float32).reshape(-1)[: n_in // 4].contiguous()
    ms = t(lambda: L.ipk_mix_probe(src.data_ptr(), dst.data_ptr(), n_in, st))
    mc = t(lambda: L.ipk_copy_probe(src.data_ptr(), cp.data_ptr(), n_in, st))
    print("%-9s mix 1:3  %.4f ms %6.0f GB/s (%.3f of peak) | copy 1:1 %.4f ms %6.0f GB/s (%.3f)" % (
        kind, ms, 4.0 * n_in / ms / 1e6, 4.0 * n_in / ms / 1e6 / 8000, mc, 2.0 * n_in / mc / 1e6, 2.0 * n_in / mc / 1e6 / 8000))
