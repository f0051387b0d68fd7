"""Development tool: what ipk_init / ipk_ctx_create cost (tables, libm check, the 8-bit step table and its exhaustive verification).  usage (GPU box): tools/init_time.py"""
import time, ctypes as C, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.zeros(1, device="cuda")
import imagepipe_amd as ipa
t0 = time.perf_counter(); ipa.init(0); t1 = time.perf_counter()
L = ipa.lib()
h = C.c_void_p()
t2 = time.perf_counter(); rc = L.ipk_ctx_create(0, C.byref(h)); t3 = time.perf_counter()
print("ipk_init %.1f ms, second ipk_ctx_create %.1f ms rc %d" % ((t1 - t0) * 1e3, (t3 - t2) * 1e3, rc))
n = C.c_uint64(); f = C.c_uint32()
t4 = time.perf_counter(); L.ipk_selftest_q8(C.byref(n), C.byref(f)); t5 = time.perf_counter()
print("ipk_selftest_q8 %.1f ms, bad %d" % ((t5 - t4) * 1e3, n.value))
