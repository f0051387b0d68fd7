"""Development tool (GPU box): the staged demosaic::full (row-walking kernel, demosaic-only variant) on a 100 MP and a 24 MP f32 buffer.  python tools/demosaic_time.py"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import imagepipe_amd as ipa
from imagepipe_amd import _lib
ipa.init(0)
L = _lib.load()
for h, w, cfa in ((10000, 10000, "RGGB"), (4000, 6000, "RGGB"), (5760, 8640, "GGRGGBGGBGGRBRGRBGGGBGGRGGRGGBRBGBRG")):
    src = torch.rand(h * w, device="cuda"); dst = torch.empty(h * w * 4, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    def run():
        _lib.check(L.ipk_demosaic_full(src.data_ptr(), w, h, cfa.encode(), dst.data_ptr(), st), "demosaic_full")
    t_end = time.perf_counter() + 0.2
    while time.perf_counter() < t_end: run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): run()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
    print("%dx%d %s: %.4f ms  %.2f TB/s" % (w, h, cfa[:4], dt * 1e3, 20.0 * h * w / dt / 1e12))
