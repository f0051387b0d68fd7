"""CPU-only checks of round 6's boundary work: the multi-device dealing rule (frame i -> device-set member i mod N, src/pipeline.rs:246-249:
frames are independent pipelines), the context entry points' behaviour without a GPU (loud, never a CPU fallback), and descriptor versioning
(the leading struct_size of ipk_fused_params / ipk_pipeline_desc: include/imagepipe_amd.h, "Descriptor versioning")."""
import ctypes as C

import numpy as np
import pytest


@pytest.fixture(scope="module")
def L():
    from imagepipe_amd import _lib
    return _lib.load()


def _deal(L, n, nd, k):
    f, s, c = C.c_size_t(99), C.c_size_t(99), C.c_size_t(99)
    rc = L.ipk_deal_frames(n, nd, k, C.byref(f), C.byref(s), C.byref(c))
    return rc, f.value, s.value, c.value


@pytest.mark.parametrize("n", [0, 1, 5, 8, 63, 64, 65, 1000])
@pytest.mark.parametrize("nd", [1, 2, 3, 8])
def test_deal_frames_is_a_partition_round_robin(L, n, nd):
    seen = []
    for k in range(nd):
        rc, first, stride, count = _deal(L, n, nd, k)
        assert rc == 0 and first == k and stride == nd
        mine = [first + j * stride for j in range(count)]
        assert all(i < n and i % nd == k for i in mine)
        seen += mine
        # balanced: no member holds more than one frame above another
        assert count in (n // nd, n // nd + 1)
    assert sorted(seen) == list(range(n))                     # every frame exactly once


def test_deal_frames_rejects_bad_indices(L):
    assert _deal(L, 10, 0, 0)[0] == -2 and _deal(L, 10, 4, 4)[0] == -2 and _deal(L, 10, 4, -1)[0] == -2
    assert L.ipk_deal_frames(10, 4, 1, None, None, None) == 0     # every output is optional
    import imagepipe_amd as ipa
    assert ipa.deal_frames(10, 4, 1) == [1, 5, 9] and ipa.deal_frames(3, 8, 5) == []


def test_context_entry_points_without_a_gpu(L):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p(123)
    assert L.ipk_ctx_create(0, C.byref(h)) == -4 and h.value is None       # IPK_ERR_NO_DEVICE, and no dangling handle
    assert b"no HIP device" in L.ipk_last_error()
    assert L.ipk_ctx_create(0, None) == -2
    assert L.ipk_init_devices(None, 0) == -4
    assert L.ipk_device_set_size() == 0 and L.ipk_device_ctx(0) is None and L.ipk_ctx_current() is None
    assert L.ipk_ctx_device(None) == -1
    assert L.ipk_ctx_destroy(None) == 0
    bogus = C.c_void_p(0xDEAD0000)
    assert L.ipk_ctx_make_current(bogus) == -2 and L.ipk_ctx_destroy(bogus) == -2          # not a handle of this library: refused, never dereferenced
    assert L.ipk_ctx_make_current(None) == 0
    assert L.ipk_devices_sync() == -1                                                      # IPK_ERR_NOT_INIT
    from imagepipe_amd._lib import PipelineDesc
    d = PipelineDesc()
    srcs = (C.c_void_p * 1)(1); dsts = (C.c_void_p * 1)(1)
    for fn in (L.ipk_pipeline_run_batch_multi, L.ipk_host_pipeline_run_batch_multi):
        assert fn(C.byref(d), srcs, dsts, 1, 0, None) == -1
    assert L.ipk_pipeline_run_batch(C.byref(d), srcs, dsts, 1, 0, None, None) == -1
    assert b"no CPU fallback" in L.ipk_last_error()


# ---- descriptor versioning -----------------------------------------------------------------------------------------------------
def _desc(**kw):
    from imagepipe_amd._lib import PipelineDesc
    d = PipelineDesc()
    d.src_type = 0; d.width = 128; d.height = 64; d.cpp = 1; d.is_cfa = 1; d.cfa = b"RGGB"; d.allow_fused = 1; d.use_fastpath = 1
    for i in range(4):
        d.blacklevels[i] = 512.0; d.whitelevels[i] = 16383.0
    d.wb_coeffs[:] = [2.0, 1.0, 1.5, float("nan")]
    for k, v in kw.items():
        setattr(d, k, v)
    return d


def _sizes(L, d):
    a, b, c, e = C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_size_t()
    rc = L.ipk_pipeline_sizes(C.byref(d), C.byref(a), C.byref(b), C.byref(c), C.byref(e))
    return rc, (a.value, b.value, c.value, e.value)


def test_new_descriptors_state_their_size(L):
    from imagepipe_amd import _lib
    assert _lib.PipelineDesc().struct_size == C.sizeof(_lib.PipelineDesc) == L.ipk_abi_sizeof(1)
    assert _lib.FusedParams().struct_size == C.sizeof(_lib.FusedParams) == L.ipk_abi_sizeof(0)
    assert _lib.PipelineDesc.struct_size.offset == 0 and _lib.FusedParams.struct_size.offset == 0


def _round8(n):
    return (n + 7) // 8 * 8


def test_a_caller_compiled_against_the_first_layout_is_accepted(L):
    """an object that ends in front of cfa_width / cfa_height (the first published layout: 8 bytes short of the second) works, with shape-from-string
    semantics, and the library never reads the bytes behind it (they hold a shape that contradicts the string here: read, they would change the call)"""
    full = _desc()
    rc, want = _sizes(L, full)
    assert rc == 0
    old = _desc(cfa_width=7, cfa_height=3, schedule=99)       # poison behind the caller's end
    old.struct_size = _round8(L.ipk_abi_sizeof(17))           # sizeof of the first published layout: offsetof(cfa_width), rounded up to the struct's alignment
    assert old.struct_size == _round8(L.ipk_abi_sizeof(21)) - 8 < C.sizeof(type(old))
    rc, got = _sizes(L, old)
    assert rc == 0, L.ipk_last_error()
    assert got == want
    # hashes: the short descriptor and the full one with zero shape fields are the same pipeline
    h_old, h_full, h_poison = C.create_string_buffer(256), C.create_string_buffer(256), C.create_string_buffer(256)
    assert L.ipk_pipeline_hashes(C.byref(old), 0, 0, h_old) == 0 and L.ipk_pipeline_hashes(C.byref(full), 0, 0, h_full) == 0
    assert h_old.raw == h_full.raw
    # the same bytes at full size: the shape fields are now inside the object, are read, and name another pipeline (the demosaic hash onwards)
    poisoned = _desc(cfa_width=7, cfa_height=3)
    rc = L.ipk_pipeline_hashes(C.byref(poisoned), 0, 0, h_poison)
    assert rc == -2 or (h_poison.raw[:32] == h_full.raw[:32] and h_poison.raw[32:64] != h_full.raw[32:64])
    # a 16-letter pattern needs the shape: a first-layout caller states it in the string, as before the fields existed
    pat16 = b"RGBGRGBGBGRGBGRG"
    o16 = _desc(cfa=b"8x2:" + pat16); o16.struct_size = L.ipk_abi_sizeof(17)      # the bare minimum: no tail padding at all
    assert _sizes(L, o16)[0] == 0
    f16 = _desc(cfa=pat16, cfa_width=8, cfa_height=2)
    assert _sizes(L, f16)[0] == 0
    assert L.ipk_pipeline_hashes(C.byref(o16), 0, 0, h_old) == 0 and L.ipk_pipeline_hashes(C.byref(f16), 0, 0, h_full) == 0 and h_old.raw == h_full.raw


def test_a_caller_compiled_against_the_second_layout_is_accepted(L):
    """the second layout ends in front of `schedule`: its callers' objects carry cfa_width / cfa_height (read) and tail padding where the third layout's
    first field now lies (NOT read: a poisoned schedule there would be refused by the compute entry points, and is ignored here)"""
    pat16 = b"RGBGRGBGBGRGBGRG"
    second = _desc(cfa=pat16, cfa_width=8, cfa_height=2, schedule=99)
    second.struct_size = _round8(L.ipk_abi_sizeof(21))
    assert L.ipk_abi_sizeof(21) <= second.struct_size < C.sizeof(type(second))
    assert _sizes(L, second)[0] == 0, L.ipk_last_error()      # the shape fields were read (16 letters need them) ...
    third = _desc(cfa=pat16, cfa_width=8, cfa_height=2)
    h2, h3 = C.create_string_buffer(256), C.create_string_buffer(256)
    assert L.ipk_pipeline_hashes(C.byref(second), 0, 0, h2) == 0 and L.ipk_pipeline_hashes(C.byref(third), 0, 0, h3) == 0 and h2.raw == h3.raw
    from imagepipe_amd import _lib
    assert _lib.FusedParams.schedule.offset == L.ipk_abi_sizeof(20) and _lib.PipelineDesc.schedule.offset == L.ipk_abi_sizeof(21)


@pytest.mark.parametrize("bad,word", [(0, b"struct_size is 0"), (8, b"smaller than the first published layout"), (None, b"newer header")])
def test_impossible_sizes_are_refused(L, bad, word):
    d = _desc()
    d.struct_size = C.sizeof(type(d)) + 8 if bad is None else bad
    assert _sizes(L, d)[0] == -2 and word in L.ipk_last_error()
    hs = C.create_string_buffer(256)
    assert L.ipk_pipeline_hashes(C.byref(d), 0, 0, hs) == -2
    assert L.ipk_pipeline_takes_fastpath(C.byref(d), 1) == -2


def test_fused_params_versioning_without_a_gpu(L):
    """ipk_fused_params goes through the same gate; without a GPU the size check cannot be reached through a compute call (NOT_INIT comes first), so
    the layout facts are checked here and the behaviour in tests/test_gpu_multictx.py"""
    from imagepipe_amd import _lib
    assert L.ipk_abi_sizeof(16) == _lib.FusedParams.cfa_width.offset < L.ipk_abi_sizeof(20) == _lib.FusedParams.schedule.offset < C.sizeof(_lib.FusedParams)
    assert L.ipk_abi_sizeof(17) == _lib.PipelineDesc.cfa_width.offset < L.ipk_abi_sizeof(21) == _lib.PipelineDesc.schedule.offset < C.sizeof(_lib.PipelineDesc)
    # the end of every layout differs from the next one's by at least the struct's alignment, or a size could not tell them apart
    for t, a, b in ((_lib.FusedParams, 16, 20), (_lib.PipelineDesc, 17, 21)):
        assert _round8(L.ipk_abi_sizeof(a)) < _round8(L.ipk_abi_sizeof(b)) < C.sizeof(t)


def test_cpp_multi_context_host_fails_loudly_without_a_gpu():
    """tests/cpp/multi_test.cpp (the Rust drop-in's shape as a compiled host) builds against the header and, without a GPU, stops at ipk_init_devices with the
    library's error text -- it never computes anything on the CPU"""
    import os
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: tests/test_gpu_multictx.py runs it")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "cpp", "build", "multi_test")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(root, "tests", "cpp")])
    r = subprocess.run([exe, "2", "64", "32", "/dev/null", "0", "0"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3 and "no HIP device" in r.stderr and "MULTI_OK" not in r.stdout, r.stdout + r.stderr
