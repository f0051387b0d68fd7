#!/usr/bin/env python
"""Development tool: what the PCIe link gives the host-pointer pipeline driver -- 48 MB up (24 MP u16) and 72 / 288 MB down (8-bit / f32 results), alone and
concurrently on two streams, page-locked buffers from ipk_host_alloc: the practical floor of host_boundary's per-frame times.
usage: tools/pcie_probe.py"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import imagepipe_amd as ipa
ipa.init(0)
L = ipa.lib()
up_b, dn8, dn32 = 48_000_000, 72_000_000, 288_000_000
hu, hd = L.ipk_host_alloc(up_b), L.ipk_host_alloc(dn32)
du = torch.empty(up_b, dtype=torch.uint8, device="cuda"); dd = torch.empty(dn32, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


up = lambda: L.ipk_memcpy_h2d(du.data_ptr(), hu, up_b, s1.cuda_stream)
for name, nb in (("8-bit", dn8), ("f32", dn32)):
    dn = lambda: L.ipk_memcpy_d2h(hd, dd.data_ptr(), nb, s2.cuda_stream)
    a, b = t(up), t(dn)
    c = t(lambda: (up(), dn()))
    print("%-5s up 48 MB %.3f ms (%.1f GB/s) | down %d MB %.3f ms (%.1f GB/s) | both at once %.3f ms (up %.1f + down %.1f GB/s)" % (
        name, a, up_b / a / 1e6, nb // 1000000, b, nb / b / 1e6, c, up_b / c / 1e6, nb / c / 1e6))

# the same transfers driven by a SHADER instead of the copy engines: ipk_copy_probe reading / writing the page-locked host buffer directly
pull = lambda: L.ipk_copy_probe(hu, du.data_ptr(), up_b, s1.cuda_stream)
for name, nb in (("8-bit", dn8), ("f32", dn32)):
    push = lambda: L.ipk_copy_probe(dd.data_ptr(), hd, nb, s2.cuda_stream)
    a, b = t(pull), t(push)
    c = t(lambda: (pull(), push()))
    print("shader %-5s pull 48 MB %.3f ms (%.1f GB/s) | push %d MB %.3f ms (%.1f GB/s) | both at once %.3f ms (%.1f GB/s combined)" % (
        name, a, up_b / a / 1e6, nb // 1000000, b, nb / b / 1e6, c, (up_b + nb) / c / 1e6))
