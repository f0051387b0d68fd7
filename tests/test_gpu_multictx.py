"""Several library contexts in ONE process (include/imagepipe_amd.h, "Several devices from ONE process"): the batch split a Rust drop-in can call.
The reference is one process whose callers loop Pipeline::run over frames (src/lib.rs:21-26, src/pipeline.rs:246-249); here frame i goes to device-set
member i mod N.  The GPU boxes have one GPU, so the set is two contexts on device 0 -- everything per-context (tables, CFA cache, scratch pool, task
queues, host lanes) is still exercised; only the second physical device is not.  Every result: bit-identical to the single-context run and to the oracle."""
import ctypes as C
import threading

import numpy as np
import pytest

import util
from util import assert_bits_equal

pytestmark = pytest.mark.gpu
XT = "GGRGGBGGBGGRBRGRBGGGBGGRGGRGGBRBGBRG"


@pytest.fixture(scope="module")
def L():
    import imagepipe_amd
    imagepipe_amd.init(0)
    return imagepipe_amd.lib()


def P(a):
    return a.ctypes.data_as(C.c_void_p)


def _pipe(ipa, w, h, cfa="RGGB", maxwidth=0, rotation=0):
    img = ipa.RawImage(width=w, height=h, data=None, cfa=cfa, blacklevels=[util.BLACK] * 4, whitelevels=[util.WHITE] * 4,
                       wb_coeffs=util.WB, cam_to_xyz_normalized=util.cam_matrix())
    pipe = ipa.Pipeline(img)
    pipe.globals.settings.maxwidth = maxwidth
    pipe.ops.transform.rotation = rotation
    return pipe


def _want(orc, raw, cfa="RGGB", maxwidth=0, out_type=0):
    od = orc.make_pipeline(raw, cfa=cfa, blacklevels=[util.BLACK] * 4, whitelevels=[util.WHITE] * 4, wb_coeffs=util.WB,
                           cam_to_xyz_normalized=util.cam_matrix(), maxwidth=maxwidth)
    return [orc.pipeline_run, orc.pipeline_output_8bit, orc.pipeline_output_16bit][out_type](od)


def test_two_contexts_on_one_gpu_from_two_threads(L, orc):
    """two contexts, two host threads, each looping host pipelines (fused Bayer, fused X-Trans, the staged scaled path) at the same time: every
    result equals the default context's and the oracle's"""
    import imagepipe_amd as ipa
    cases = [("RGGB", 0, 0), (XT, 0, 0), ("GBRG", 90, 0), ("RGGB", 0, 1), ("RGBE", 0, 0)]
    h, w = 120, 384
    frames = [util.noise_u16(util.SEED + 600 + i, h, w) for i in range(len(cases))]
    descs, shapes, wants, base = [], [], [], []
    for (cfa, mw, ot), raw in zip(cases, frames):
        p = _pipe(ipa, w, h, cfa, mw)
        descs.append(p.desc())
        _, (fw, fh) = p.sizes()
        shapes.append((fh, fw, 3))
        wants.append(_want(orc, raw, cfa, mw, ot))
        out = np.empty(fw * fh * 3, np.float32 if ot == 0 else np.uint8)
        assert L.ipk_host_pipeline_run(C.byref(descs[-1]), P(raw), P(out), ot, None) == 0, L.ipk_last_error()      # default context
        base.append(out)
    ctxs = [ipa.Context(0), ipa.Context(0)]
    assert ctxs[0].handle != ctxs[1].handle and ctxs[0].device == ctxs[1].device == 0
    errs, outs = [], {}
    start = threading.Barrier(2)

    def work(k):
        try:
            assert L.ipk_ctx_make_current(ctxs[k].handle) == 0
            assert L.ipk_ctx_current() == ctxs[k].handle
            start.wait()
            for rep in range(6):
                for i, ((cfa, mw, ot), raw) in enumerate(zip(cases, frames)):
                    out = np.empty(base[i].size, base[i].dtype)
                    rc = L.ipk_host_pipeline_run(C.byref(descs[i]), P(raw), P(out), ot, None)
                    assert rc == 0, L.ipk_last_error()
                    outs[(k, rep, i)] = out
        except BaseException as e:      # noqa: a failure in a thread must reach the test
            errs.append(repr(e))
    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    [t.start() for t in th]; [t.join() for t in th]
    assert not errs, errs
    assert L.ipk_ctx_current() != ctxs[0].handle and L.ipk_ctx_current() != ctxs[1].handle       # this thread never left the default context
    for (k, rep, i), out in outs.items():
        assert np.array_equal(out.view(np.uint8), base[i].view(np.uint8)), (k, rep, i)
        if cases[i][2] == 0:
            assert_bits_equal(out.reshape(shapes[i]), wants[i], "ctx %d rep %d case %d" % (k, rep, i))
        else:
            assert np.array_equal(out.reshape(shapes[i]), wants[i])
    for c in ctxs:
        c.destroy()
    # a destroyed context is refused, loudly, wherever it turns up
    assert L.ipk_ctx_make_current(ctxs[0].handle) == -1 and b"destroyed" in L.ipk_last_error()
    assert L.ipk_ctx_device(ctxs[0].handle) == -1
    assert L.ipk_is_initialized() == 1                                                            # the default context is untouched


def test_device_pointers_under_two_contexts_concurrently(L, orc):
    """the device-pointer entry points under `with ctx:` on two threads -- each context has its own scratch pool, CFA tables and task queues; frames
    large enough (and a batch) that the row-walking kernel DRAWS tasks from the context's queue"""
    import torch
    import imagepipe_amd as ipa
    h, w, n = 256, 1024, 6
    frames = [util.noise_u16(util.SEED + 700 + i, h, w) for i in range(n)]
    wants = [_want(orc, f) for f in frames]
    ctxs = [ipa.Context(0), ipa.Context(0)]
    errs, res = [], {}
    start = threading.Barrier(2)

    def work(k):
        try:
            with ctxs[k]:
                st = torch.cuda.Stream()
                srcs = [torch.from_numpy(f.view(np.int16)).cuda() for f in frames]
                outs = [torch.empty(h * w * 3, dtype=torch.float32, device="cuda") for _ in range(n)]
                plan = ipa.FusedPlan(width=w, height=h, is_float=False, black0=util.BLACK, white0=util.WHITE, cfa="RGGB", wb_coeffs=util.WB,
                                     cam_to_xyz_normalized=util.cam_matrix())
                batch = ipa.FusedBatchPlan(plan, srcs, outs)
                torch.cuda.synchronize()
                start.wait()
                for rep in range(10):
                    batch.run(st.cuda_stream)                      # one persistent launch for the six frames, tasks drawn from THIS context's queue
                    for s, o in zip(srcs[:2], outs[:2]):
                        plan.run(s, o, st.cuda_stream)
                st.synchronize()
                res[k] = [o.cpu().numpy() for o in outs]
                # the staged path as well (scratch pool): X-Trans preview through ipk_pipeline_run
                p = _pipe(ipa, w, h, XT, 200)
                p.globals.image.data = srcs[0]
                with torch.cuda.stream(st):
                    got = p.run().numpy()
                res[(k, "xt")] = got
        except BaseException as e:      # noqa
            errs.append(repr(e))
    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    [t.start() for t in th]; [t.join() for t in th]
    assert not errs, errs
    want_xt = _want(orc, frames[0], XT, 200)
    for k in range(2):
        for i in range(n):
            assert_bits_equal(res[k][i].reshape(h, w, 3), wants[i], "ctx %d frame %d" % (k, i))
        assert_bits_equal(res[(k, "xt")], want_xt, "ctx %d X-Trans preview" % k)
    for c in ctxs:
        c.destroy()


@pytest.mark.parametrize("cfa,maxwidth,out_type,n", [("RGGB", 0, 0, 7), ("RGGB", 0, 1, 4), (XT, 120, 0, 3), ("RGGB", 0, 2, 1)])
def test_host_batch_over_a_device_set(L, orc, cfa, maxwidth, out_type, n):
    """ipk_host_pipeline_run_batch_multi: host frames dealt over the set {device 0, device 0}, one host thread per member; every frame equals the
    oracle and what ipk_host_pipeline_run_batch on one context gives"""
    import imagepipe_amd as ipa
    members = ipa.init_devices([0, 0])
    assert len(members) == 2 == L.ipk_device_set_size() and members[0].handle != members[1].handle
    h, w = 100, 320
    frames = [util.noise_u16(util.SEED + 800 + i, h, w) for i in range(n)]
    p = _pipe(ipa, w, h, cfa, maxwidth)
    d = p.desc()
    _, (fw, fh) = p.sizes()
    dt = {0: np.float32, 1: np.uint8, 2: np.uint16}[out_type]
    out_bytes = fw * fh * 3 * np.dtype(dt).itemsize
    sp = [L.ipk_host_alloc(h * w * 2) for _ in range(n)]
    dp = [L.ipk_host_alloc(out_bytes) for _ in range(n)]
    assert all(sp) and all(dp)
    for q, f in zip(sp, frames):
        C.memmove(q, f.ctypes.data, f.nbytes)
    outs = [np.frombuffer((C.c_char * out_bytes).from_address(q), dtype=dt) for q in dp]
    srcs = (C.c_void_p * n)(*sp); dsts = (C.c_void_p * n)(*dp)
    used = C.c_int(-1)
    assert L.ipk_host_pipeline_run_batch_multi(C.byref(d), srcs, dsts, n, out_type, C.byref(used)) == 0, L.ipk_last_error()
    assert used.value == (1 if (maxwidth == 0) else 0)
    for i in range(n):
        want = _want(orc, frames[i], cfa, maxwidth, out_type)
        if out_type == 0:
            assert_bits_equal(outs[i].reshape(fh, fw, 3), want, "multi frame %d" % i)
        else:
            assert np.array_equal(outs[i].reshape(fh, fw, 3), want), i
    # errors of a member reach the caller with the member named
    bad = (C.c_void_p * n)(*([None] + sp[1:]))
    assert L.ipk_host_pipeline_run_batch_multi(C.byref(d), bad, dsts, n, out_type, None) < 0
    del outs
    for q in sp + dp:
        L.ipk_host_free(q)
    assert L.ipk_init_devices(None, 0) == 0 and L.ipk_device_set_size() >= 1                      # replaces the set: one member per visible device


def test_device_batch_over_a_device_set(L, orc):
    """ipk_pipeline_run_batch_multi + ipk_devices_sync: device frames, frame i on member i mod 2, launched from one host thread; and
    ipk_pipeline_run_batch on one context (one persistent launch) gives the same"""
    import torch
    import imagepipe_amd as ipa
    members = ipa.init_devices([0, 0])
    h, w, n = 200, 512, 9
    frames = [util.noise_u16(util.SEED + 900 + i, h, w) for i in range(n)]
    p = _pipe(ipa, w, h)
    d = p.desc()
    srcs_t, outs_t = [], []
    for i, f in enumerate(frames):
        with members[i % 2]:                                        # allocate on the member's device (here both are device 0)
            srcs_t.append(torch.from_numpy(f.view(np.int16)).cuda())
            outs_t.append(torch.zeros(h * w * 3, dtype=torch.float32, device="cuda"))
    torch.cuda.synchronize()
    srcs = (C.c_void_p * n)(*[t.data_ptr() for t in srcs_t]); dsts = (C.c_void_p * n)(*[t.data_ptr() for t in outs_t])
    used = C.c_int(-1)
    before = L.ipk_ctx_current()
    assert L.ipk_pipeline_run_batch_multi(C.byref(d), srcs, dsts, n, 0, C.byref(used)) == 0, L.ipk_last_error()
    assert L.ipk_devices_sync() == 0
    assert L.ipk_ctx_current() == before and used.value == 1
    for i in range(n):
        assert_bits_equal(outs_t[i].cpu().numpy().reshape(h, w, 3), _want(orc, frames[i]), "multi device frame %d" % i)
        outs_t[i].zero_()
    torch.cuda.synchronize()
    st = torch.cuda.current_stream().cuda_stream
    assert L.ipk_pipeline_run_batch(C.byref(d), srcs, dsts, n, 0, C.byref(used), st) == 0 and used.value == 1
    torch.cuda.synchronize()
    for i in (0, n - 1):
        assert_bits_equal(outs_t[i].cpu().numpy().reshape(h, w, 3), _want(orc, frames[i]), "one-context batch frame %d" % i)
    # a descriptor that is not one fused launch per frame (a size limit): the batch falls back to n pipeline runs, same results as the oracle
    p2 = _pipe(ipa, w, h, "RGGB", 128)
    d2 = p2.desc()
    _, (fw, fh) = p2.sizes()
    small = [torch.zeros(fw * fh * 3, dtype=torch.float32, device="cuda") for _ in range(3)]
    dsts2 = (C.c_void_p * 3)(*[t.data_ptr() for t in small])
    assert L.ipk_pipeline_run_batch_multi(C.byref(d2), srcs, dsts2, 3, 0, C.byref(used)) == 0 and used.value == 0
    assert L.ipk_devices_sync() == 0
    for i in range(3):
        assert_bits_equal(small[i].cpu().numpy().reshape(fh, fw, 3), _want(orc, frames[i], "RGGB", 128), "multi scaled frame %d" % i)
    assert L.ipk_init_devices(None, 0) == 0


def test_cache_belongs_to_its_context(L, orc):
    """a PipelineCache holds device buffers of the context that made it: another context is refused instead of handed foreign memory"""
    import torch
    import imagepipe_amd as ipa
    h, w = 64, 256
    raw = util.noise_u16(util.SEED + 950, h, w)
    p = _pipe(ipa, w, h)
    p.globals.image.data = torch.from_numpy(raw.view(np.int16)).cuda()
    cache = ipa.PipelineCache(1 << 26)
    got = p.run(cache).numpy()
    assert_bits_equal(got, _want(orc, raw), "cached run, default context")
    other = ipa.Context(0)
    with other:
        with pytest.raises(ipa.IpkError, match="another context"):
            p.run(cache)
    other.destroy()
    assert_bits_equal(p.run(cache).numpy(), _want(orc, raw), "cached run again")
    cache.close()


def test_fused_params_first_layout_object_is_accepted(L, orc):
    """descriptor versioning on the compute path: an ipk_fused_params that ends in front of cfa_width / cfa_height (struct_size 8 short) runs with
    shape-from-string semantics -- the bytes behind it (poisoned here) are never read; impossible sizes are refused"""
    from imagepipe_amd._lib import FusedParams
    h, w = 50, 300
    raw = util.noise_u16(util.SEED + 960, h, w)
    p = FusedParams()
    p.src_type = 0; p.owidth = w; p.width = w; p.height = h
    p.black0 = util.BLACK; p.white0 = util.WHITE; p.cfa = b"GBRG"
    p.wb_coeffs[:] = list(util.WB); p.cam_to_xyz_normalized[:] = [float(v) for v in util.cam_matrix().ravel()]
    p.npoints = 1; p.points[0] = 0.5; p.points[1] = 0.6
    want = _want(orc, raw, "GBRG")
    out = np.empty((h, w, 3), np.float32)
    assert L.ipk_host_raw_to_srgb(C.byref(p), P(raw), P(out)) == 0, L.ipk_last_error()
    assert_bits_equal(out, want, "full-size descriptor")
    p.cfa_width = 7; p.cfa_height = 3                               # poison behind the first layout's end
    assert L.ipk_host_raw_to_srgb(C.byref(p), P(raw), P(out)) == -2
    p.struct_size = (L.ipk_abi_sizeof(16) + 7) // 8 * 8             # sizeof of the first published layout
    out[:] = 0
    assert L.ipk_host_raw_to_srgb(C.byref(p), P(raw), P(out)) == 0, L.ipk_last_error()
    assert_bits_equal(out, want, "first-layout descriptor")
    p.cfa_width = 0; p.cfa_height = 0; p.schedule = 99             # the second layout ends in front of `schedule`
    p.struct_size = C.sizeof(FusedParams)
    assert L.ipk_host_raw_to_srgb(C.byref(p), P(raw), P(out)) == -2 and b"schedule" in L.ipk_last_error()
    p.struct_size = (L.ipk_abi_sizeof(20) + 7) // 8 * 8
    out[:] = 0
    assert L.ipk_host_raw_to_srgb(C.byref(p), P(raw), P(out)) == 0, L.ipk_last_error()
    assert_bits_equal(out, want, "second-layout descriptor")
    p.schedule = 0
    for bad in (0, 16, C.sizeof(FusedParams) + 8):
        p.struct_size = bad
        assert L.ipk_host_raw_to_srgb(C.byref(p), P(raw), P(out)) == -2, bad


@pytest.mark.parametrize("devices,n,w,h", [(["0", "0"], 7, 512, 200), (["0"], 3, 300, 97), (["0", "0", "0"], 5, 1024, 64)])
def test_multi_context_batch_from_a_plain_cpp_host(orc, devices, n, w, h):
    """tests/cpp/multi_test.cpp: the Rust drop-in's shape as a compiled C++ program -- ipk_init_devices, the shoot developed through the single-context loop,
    ipk_host_pipeline_run_batch_multi (f32 and 8-bit), the caller's own worker threads under ipk_ctx_make_current, and device-resident frames through
    ipk_pipeline_run_batch_multi + ipk_devices_sync: all bit-identical inside the program; its frame 0 against the oracle here"""
    import os
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "cpp", "build", "multi_test")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(root, "tests", "cpp")])
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "f0.f32")
        r = subprocess.run([exe, str(n), str(w), str(h), out] + devices, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "MULTI_OK %d members" % len(devices) in r.stdout, r.stdout + r.stderr
        got = np.fromfile(out, np.float32).reshape(h, w, 3)
        raw = np.fromfile(out + ".u16", np.uint16).reshape(h, w)
    assert_bits_equal(got, _want(orc, raw), "multi_test frame 0 vs oracle")
