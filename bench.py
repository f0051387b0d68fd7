#!/usr/bin/env python
"""bench.py -- megapixels/s of the full raw->sRGB pipe on a 100 MP f32 Bayer frame (BASELINE.json's metric).

  python bench.py --gpus N --steps K --warmup W [--config c3|c2|c4|c5|c5b] [--batch B --width X --height Y]
  N > 1: one rank per GPU.  Either the caller starts the ranks (python -m torch.distributed.run --nnodes=1 --nproc-per-node N
  --master-addr 127.0.0.1 ... bench.py --gpus N ...) or, invoked plainly, bench.py starts them itself the same way (relaunch_as_ranks).
  A line whose n_gpus differs from --gpus is never printed: fewer visible GPUs than N, or a WORLD_SIZE that is not N, is a non-zero exit.

One step = one pass of the hot path (Pipeline::run: gofloat + demosaic + tolab + basecurve + fromlab + gamma) over one batch of
synthetic frames, input and output resident in HBM.  Frames are independent (src/pipeline.rs:246-249), so frame i of a batch goes to
rank i mod N with no data-path collective; value = all frames' pixels / max-rank time.  Prints ONE JSON line on rank 0.

  --config c3 (default)  BASELINE.json configs[2]: one 10000x10000 RGGB f32 frame PER GPU per step (weak scaling) -- the headline
  --config c2            configs[1]: 6000x4000 RGGB f32, one frame per GPU per step
  --config c4            configs[3]: a batch of 64 x 24 MP frames sharded frame i -> rank i mod N (strong scaling; = --batch 64
                         --width 6000 --height 4000), with a second figure that includes the all-gather of the f32 results
  --config c5 / c5b      configs[4]: 8640x5760 X-Trans -> 2160x1440 (scaled demosaic + point-wise chain) / the same at full size
The default run also reports, as extra objects of the same line: the cold-clock time, the median, the measured device-copy
ceiling, two more data kinds (smooth, photo), the 64 x 24 MP batch (configs[3]) and the VALU-issue model of the kernel; with N > 1 also
`band_mode`: ONE frame row-sharded over the N GPUs on the library's RCCL transport (halo exchange, band kernel, all-gather), measured in child
processes so that it cannot cost the line.
"""
import argparse
import ctypes
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); the measured device-copy ceiling is reported beside it
XTRANS = "GGRGGBGGBGGRBRGRBGGGBGGRGGRGGBRBGBRG"
CONFIGS = {            # name: (width, height, cfa, maxwidth, frames per step [None = one per GPU], BASELINE.json configs[] index)
    "c3": (10000, 10000, "RGGB", 0, None, 2),
    "c2": (6000, 4000, "RGGB", 0, None, 1),
    "c4": (6000, 4000, "RGGB", 0, 64, 3),
    "c5": (8640, 5760, XTRANS, 2160, None, 4),
    "c5b": (8640, 5760, XTRANS, 0, None, 4),
}


CURVES = {"default": [(0.5, 0.6)], "none": [], "user5": [(0.1, 0.07), (0.3, 0.27), (0.5, 0.6), (0.7, 0.82), (0.9, 0.95)]}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c3")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--batch", type=int, default=None, help="frames per step, sharded frame i -> rank i mod N (default: one frame per GPU)")
    ap.add_argument("--data", choices=["noise", "smooth", "photo", "flat", "white"], default="noise",
                    help="noise: uniform 14-bit values (worst case, the headline); smooth: diagonal gradient over the full range; "
                         "photo: low-frequency mid-tone scene with shot-like noise and ~2 %% blown highlights; flat / white: development extremes (nothing / everything saturated)")
    ap.add_argument("--src", choices=["f32", "u16"], default="f32")
    ap.add_argument("--out", choices=["f32", "u8", "u16"], default="f32")
    ap.add_argument("--curve", choices=sorted(CURVES), default="default",
                    help="OpBaseCurve.points (src/ops/curves.rs:12-49): default = the raw default [(0.5, 0.6)] (3 knots, the compiled-in form); none = no points "
                         "(the op is a no-op unless --exposure is set); user5 = five user points (7 knots: the binary search of SplineFunc::interpolate)")
    ap.add_argument("--exposure", type=float, default=0.0, help="OpBaseCurve.exposure")
    ap.add_argument("--linear", action="store_true", help="PipelineSettings.linear: no OpGamma (src/ops/gamma.rs:17); u16 output forces it as output_16bit does")
    ap.add_argument("--kernel-stats", type=int, default=0, metavar="N",
                    help="additionally time N single launches with an event pair each and report min / median / p95 / max / stddev (roofline.launch_stats)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target wall time of the CPU baseline leg")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true", help="keep the committed PMC traffic figure instead of measuring it in two short child runs under rocprofv3 (at most 45 s each)")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements (cold clock, copy ceiling, other data kinds, 64-frame batch)")
    ap.add_argument("--prewarm-ms", type=float, default=250.0,
                    help="untimed launches before the W warmup steps, to bring the shader clock to its loaded steady state (0 = off); "
                         "the same workload without it is reported as config.cold_ms")
    ap.add_argument("--band", action="store_true",
                    help="additionally time ONE frame row-sharded over the N GPUs with the halo exchange (extra 'band_mode' object)")
    ap.add_argument("--schedule", choices=["auto", "split"], default="auto",
                    help="ipk_fused_params.schedule: split = every wave gets two pieces half a frame apart (frames with blown regions; results identical)")
    ap.add_argument("--no-box-state", action="store_true", help="skip the shader-clock / socket-power leg (config.shader_clock_GHz ...): a second of extra launches, "
                                                               "unwanted when a profiler averages over every launch of the run")
    ap.add_argument("--host-boundary", action="store_true", help="additionally run the host-buffers-in / host-buffers-out leg (`host_boundary`; part of the default run)")
    ap.add_argument("--single-process", action="store_true",
                    help="N GPUs from ONE process, the shape a drop-in behind the reference's Pipeline::run has (no torch.distributed, no RCCL): one ipk_ctx per "
                         "device (ipk_init_devices), frame i -> device i mod N (ipk_pipeline_run_batch_multi); prints the same line with a `scale` object")
    ap.add_argument("--devices", default=None, help="--single-process: the device ordinals, comma separated (default 0..N-1; an ordinal may repeat -- "
                                                    "development: two contexts on the one GPU of a 1-GPU box)")
    ap.add_argument("--sp-child", default=None, metavar="DEVICES", help=argparse.SUPPRESS)
    ap.add_argument("--band-child", nargs=4, metavar=("RANK", "WORLD", "LOCAL_RANK", "IDHEX"), default=None, help=argparse.SUPPRESS)
    return ap.parse_args()


def synth_frame(torch, h, w, kind, seed):
    """14-bit sensor values (black 512, white 16383) as int32, generated on the device."""
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    if kind == "noise":       # uniform in [0, 16383]: worst case for lookup-table locality
        return torch.randint(0, 16384, (h, w), generator=g, device="cuda", dtype=torch.int32)
    rr = torch.arange(h, device="cuda", dtype=torch.int32)[:, None]
    cc = torch.arange(w, device="cuda", dtype=torch.int32)[None, :]
    if kind == "flat":        # development: mid-grey + noise, nothing saturates (no out-of-table Lab ratio anywhere)
        return torch.randint(6000, 7000, (h, w), generator=g, device="cuda", dtype=torch.int32)
    if kind == "white":       # development: everything blown (every Lab ratio out of the table)
        return torch.full((h, w), 16383, device="cuda", dtype=torch.int32)
    if kind == "photo":       # what a sensor usually delivers: exposure well below clipping, a few saturated patches
        y = rr.to(torch.float32) / h; x = cc.to(torch.float32) / w
        scene = 0.28 + 0.18 * torch.sin(6.0 * x + 2.0 * y) * torch.cos(5.0 * y) + 0.10 * torch.sin(23.0 * x) * torch.sin(19.0 * y)
        blown = ((x - 0.8) ** 2 + (y - 0.2) ** 2 < 0.006) | ((x - 0.3) ** 2 + (y - 0.7) ** 2 < 0.002)
        scene = torch.where(blown, torch.full_like(scene, 1.2), scene)
        n = torch.randint(-40, 41, (h, w), generator=g, device="cuda", dtype=torch.int32)
        return torch.clamp((512 + scene * 15871).to(torch.int32) + n, min=0, max=16383)
    base = ((rr + cc) % 4096) * 4
    n = torch.randint(0, 64, (h, w), generator=g, device="cuda", dtype=torch.int32)
    return torch.clamp(base + n, max=16383)


def glibc_version():
    try:
        libc = ctypes.CDLL(None)
        libc.gnu_get_libc_version.restype = ctypes.c_char_p
        return libc.gnu_get_libc_version().decode()
    except Exception:
        return None


def relaunch_as_ranks(args):
    """`python bench.py --gpus N` (N > 1) without a launcher's environment: start the N ranks under torch.distributed.run, one per GPU,
    and hand their exit code on.  Refuses (exit 2) when the node shows fewer than N GPUs -- IPK_BENCH_SHARE_GPU=1 (development: all ranks
    on GPU 0 over gloo, the library's host transport) lifts that."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()
    if have < args.gpus and os.environ.get("IPK_BENCH_SHARE_GPU") != "1":
        sys.stderr.write("bench.py: --gpus %d but this node shows %d GPU(s); refusing to print a line for fewer ranks than asked\n" % (args.gpus, have))
        sys.exit(2)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, cwd=os.getcwd()))


class Ctx:
    """process-wide plumbing: torch, the process group, barriers and max-over-ranks reductions"""

    def __init__(self, args):
        import torch
        self.torch = torch
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != args.gpus:                     # never a line whose n_gpus is not what was asked for
            sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d\n" % (args.gpus, self.world))
            raise SystemExit(2)
        self.dist = None
        self.share = False
        self._comm = None
        self.failed = []
        if self.world > 1:
            import torch.distributed as dist
            # IPK_BENCH_SHARE_GPU=1 (development only): all ranks on one GPU over gloo, to exercise the N > 1 control flow on a 1-GPU box
            share = os.environ.get("IPK_BENCH_SHARE_GPU") == "1"
            self.share = share
            if share:
                self.local_rank = 0
            elif torch.cuda.device_count() < self.world:
                sys.stderr.write("bench.py: %d ranks but %d visible GPU(s)\n" % (self.world, torch.cuda.device_count()))
                raise SystemExit(2)
            torch.cuda.set_device(self.local_rank)
            if share:
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
            else:
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=torch.device("cuda", self.local_rank))
            self.dist = dist
        self.red_dev = "cpu" if (self.dist is not None and self.dist.get_backend() == "gloo") else "cuda"

    def fail(self, leg, msg):
        """a secondary measurement that failed: loud (stderr, "failed_legs" of the line) but never silent and never fatal to the headline"""
        self.failed.append(leg)
        sys.stderr.write("bench.py: rank %d: leg %s FAILED: %s\n" % (self.rank, leg, msg))

    def comm(self):
        """the LIBRARY's communicator over the ranks (ipk_comm: RCCL when the ranks have a GPU each, its host transport when they share one)"""
        if self._comm is None:
            from imagepipe_amd import parallel as par
            self._comm = par.Comm()
        return self._comm

    def comm_info(self):
        """what the library's transport saw: (rank, ranks, transport) from ipk_comm_info"""
        from imagepipe_amd import _lib
        r, n, t = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _lib.check(_lib.load().ipk_comm_info(self.comm().handle, ctypes.byref(r), ctypes.byref(n), ctypes.byref(t)), "ipk_comm_info")
        return {"ranks": n.value, "transport": "rccl" if t.value == 0 else "host"}

    def barrier(self):
        self.torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def max_over_ranks(self, vals):
        if self.dist is None:
            return [float(v) for v in vals]
        t = self.torch.tensor(list(vals), device=self.red_dev, dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return [float(v) for v in t]


def timed(ctx, step, steps, warmup, prewarm_ms=0.0):
    """W untimed warm-ups, then EXACTLY `steps` steps between barrier + synchronize on both sides.  Returns (wall seconds for the
    K steps, max over ranks; mean HIP-event milliseconds per step over that region and the median of a second, per-step-evented pass of K
    steps, on the launch stream, max over ranks)."""
    torch = ctx.torch
    ctx.barrier()
    if prewarm_ms > 0:
        # The MI355X raises its shader clock over the first tens of milliseconds of sustained load (the same kernel takes ~15 % longer
        # in the first 20 launches after idle).  These launches are untimed; config.cold_ms reports the run without them.
        t_pw = time.perf_counter()
        while (time.perf_counter() - t_pw) * 1e3 < prewarm_ms:
            for _ in range(4):
                step()
            torch.cuda.synchronize()
    for _ in range(warmup):
        step()
    ctx.barrier()
    ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev_a.record()
    for i in range(steps):
        step()
    ev_b.record()
    ctx.barrier()
    elapsed = time.perf_counter() - t0
    mean_ms = ev_a.elapsed_time(ev_b) / steps
    # the median comes from a second pass of K steps with an event between every two (outside the timed region: each event is a marker
    # in the queue that costs the GPU a few microseconds of idle time, which matters for the 0.1 ms configurations)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    for i in range(steps):
        evs[i].record()
        step()
    evs[steps].record()
    ctx.barrier()
    per = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(steps))
    median_ms = per[steps // 2] if steps % 2 else 0.5 * (per[steps // 2 - 1] + per[steps // 2])
    elapsed, mean_ms, median_ms = ctx.max_over_ranks([elapsed, mean_ms, median_ms])
    return elapsed, mean_ms, median_ms


class FusedBatch:
    """B frames of W x H through the fused raw->sRGB launch, frame i on rank i mod N; every frame keeps its own output buffer"""

    def __init__(self, ctx, ipa, util, W, H, B, cfa, src_kind, out_kind, data, seed0, points=((0.5, 0.6),), exposure=0.0, linear=False, schedule=0):
        torch = ctx.torch
        self.ctx, self.W, self.H, self.B = ctx, W, H, B
        self.is_float = src_kind == "f32"
        self.out_type = {"f32": ipa.OUT_F32, "u8": ipa.OUT_U8, "u16": ipa.OUT_U16}[out_kind]
        out_dt = {"f32": torch.float32, "u8": torch.uint8, "u16": torch.int16}[out_kind]
        self.mine = [i for i in range(B) if i % ctx.world == ctx.rank]
        self.srcs, self.dsts = [], []
        for i in self.mine:
            ints = synth_frame(torch, H, W, data, seed0 + i)
            self.srcs.append(ints.to(torch.float32).reshape(-1).contiguous() if self.is_float else ints.to(torch.int16).reshape(-1).contiguous())
            del ints
            self.dsts.append(torch.empty(H * W * 3, dtype=out_dt, device="cuda"))
        self.cm = util.cam_matrix()
        self.black, self.white, self.wb = util.BLACK, util.WHITE, util.WB
        self.plan = ipa.FusedPlan(width=W, height=H, is_float=self.is_float, black0=util.BLACK, white0=util.WHITE, cfa=cfa, wb_coeffs=util.WB,
                                  cam_to_xyz_normalized=self.cm, out_type=self.out_type, points=tuple(points), exposure=exposure, linear=linear, schedule=schedule)
        self.points, self.exposure, self.linear = [tuple(p) for p in points], exposure, linear
        self.stream = torch.cuda.current_stream().cuda_stream
        self.in_b = 4.0 if self.is_float else 2.0
        self.out_b = {"f32": 12.0, "u8": 3.0, "u16": 6.0}[out_kind]
        self.launches_per_step = max(1, len(self.mine))        # frames per step on this rank: kernel_ms is the time per FRAME
        self.batch = ipa.FusedBatchPlan(self.plan, self.srcs, self.dsts) if len(self.mine) > 1 and os.environ.get("IPK_BENCH_NO_BATCH_LAUNCH") != "1" else None

    def step(self):
        if self.batch is not None:                      # several frames on this rank: ipk_raw_to_srgb_batch, one persistent launch per 64 frames
            self.batch.run(self.stream)
            return
        for s, d in zip(self.srcs, self.dsts):
            self.plan.run(s, d, self.stream)

    def alg_bytes_per_launch(self):
        return (self.in_b + self.out_b) * self.H * self.W


def oracle_check(ctx, util, wl, rows=None):
    """whole-frame (or first `rows` rows) bit-for-bit comparison of the rank's first frame with the CPU oracle"""
    import numpy as np
    import oracle
    torch = ctx.torch
    H, W = wl.H, wl.W
    src, dst = wl.srcs[0], wl.dsts[0]
    wl.plan.run(src, dst, wl.stream); torch.cuda.synchronize()

    def compare(n):
        part = src[: min(H, n + 2) * W].cpu().numpy().reshape(-1, W)
        desc = oracle.make_pipeline(part if wl.is_float else part.view(np.uint16), cfa="RGGB", source_kind=1 if wl.is_float else 0,
                                    blacklevels=[util.BLACK] * 4, whitelevels=[util.WHITE] * 4, wb_coeffs=util.WB, cam_to_xyz_normalized=wl.cm,
                                    points=wl.points, exposure=wl.exposure, linear=wl.linear)
        got = dst[: n * W * 3].cpu()
        if wl.out_b == 12.0:
            want = torch.from_numpy(oracle.pipeline_run(desc)[:n].reshape(-1))
            if not torch.equal(got.view(torch.int32), want.view(torch.int32)):
                util.assert_bits_equal(got.numpy().reshape(n, W, 3), want.numpy().reshape(n, W, 3), "bench parity check")
        else:                                              # the quantised outputs: output_8bit / output_16bit (src/pipeline.rs:404-421, :451-468)
            q = (oracle.pipeline_output_8bit if wl.out_b == 3.0 else oracle.pipeline_output_16bit)(desc)[:n].reshape(-1)
            g = got.numpy() if wl.out_b == 3.0 else got.numpy().view(np.uint16)
            if not np.array_equal(g, q):
                bad = np.flatnonzero(g != q)
                raise AssertionError("bench parity check: %d of %d quantised samples differ, first at %d: got %d want %d" % (bad.size, q.size, bad[0], g[bad[0]], q[bad[0]]))
    if rows is not None:
        compare(rows)
        return "first %d rows of the rank's first frame bit-identical to the CPU oracle" % rows
    try:
        compare(H)
        return "all %d rows (%d output samples) bit-identical to the CPU oracle" % (H, H * W * 3)
    except MemoryError:
        compare(12)
        return "first 12 rows bit-identical to the CPU oracle (no host memory for the whole frame)"


def copy_ceiling(ctx, nbytes):
    """device-to-device copy of `nbytes` (read + write counted) with the library's own 16-bytes-per-lane copy kernel (ipk_copy_probe): the
    practical HBM ceiling next to the 8 TB/s spec peak (MI355X_MICROARCH.md quotes 6.29 TB/s for such a copy); torch's copy_ beside it"""
    torch = ctx.torch
    import imagepipe_amd as ipa
    L = ipa.lib()
    a = torch.empty(nbytes // 4, dtype=torch.float32, device="cuda").fill_(1.0)
    b = torch.empty_like(a)
    st = torch.cuda.current_stream().cuda_stream

    def run(fn):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        return 2.0 * nbytes / (e0.elapsed_time(e1) / 20 * 1e-3) / 1e9

    def probe():
        rc = L.ipk_copy_probe(a.data_ptr(), b.data_ptr(), nbytes, st)
        assert rc == 0, rc
    own, tch = run(probe), run(lambda: b.copy_(a))
    del b
    # the fused path's own read : write mix (4 B in, 12 B out per pixel) as a flat, contiguous, nontemporal kernel without arithmetic (ipk_mix_probe):
    # what the memory system gives ANY kernel that writes three f32 per sample read
    n_in = (nbytes // 4) // 4096 * 4096                      # source BYTES of the probe (a quarter of the copy's: the mix writes three times what it reads)
    c = torch.empty(n_in * 3 // 4, dtype=torch.float32, device="cuda")

    def mix():
        rc = L.ipk_mix_probe(a.data_ptr(), c.data_ptr(), n_in, st)
        assert rc == 0, rc
    for _ in range(5):
        mix()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        mix()
    e1.record(); torch.cuda.synchronize()
    mixed = 4.0 * n_in / (e0.elapsed_time(e1) / 20 * 1e-3) / 1e9
    del a, c
    return own, tch, mixed


def ceiling_leg(ctx, wl, steps, warmup):
    """The fused kernel's own memory skeleton timed in this session (ipk_stream_probe: the same persistent launch, task walk, row loads, OpGoFloat,
    demosaic, LDS staging and nontemporal f32 stores, no point-wise stages): the time the kernel would take if its colour arithmetic were free.
    Same frame, same stream, its own short clock pre-warm.  Returns (mean ms, median ms) or None when the probe has no variant for the workload."""
    torch = ctx.torch
    if not wl.srcs:
        return None
    out = torch.empty(wl.H * wl.W * 3, dtype=torch.float32, device="cuda")
    try:
        wl.plan.probe(wl.srcs[0], out, wl.stream)
    except Exception as e:
        sys.stderr.write("bench.py: no stream probe for this workload: %s\n" % e)
        return None
    _, m, md = timed(ctx, lambda: wl.plan.probe(wl.srcs[0], out, wl.stream), steps, warmup, 120.0)
    del out
    return m, md


def launch_stats(ctx, wl, n):
    """n single launches, an event pair around each, in the steady state: 60 untimed launches run straight in front of them with no synchronisation in
    between (the events are created beforehand).  min / median / p95 / max / stddev of the launch duration in ms -- and, beside it, what an idle gap costs:
    after a synchronise + 50 ms of sleep the shader clock has dropped and takes ~20 launches (~12 ms) to come back, which is what the 0.6 ms outliers of
    earlier rounds were (tools/outlier_probe.py: one run of 19 consecutive slow launches behind the pause, none elsewhere in 3000)."""
    torch = ctx.torch
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    idle = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for _ in range(60):
        wl.step()
    for a, b in evs:
        a.record(); wl.step(); b.record()
    torch.cuda.synchronize()
    time.sleep(0.05)
    for a, b in idle:
        a.record(); wl.step(); b.record()
    torch.cuda.synchronize()
    per = sorted(a.elapsed_time(b) / wl.launches_per_step for a, b in evs)
    after_idle = [a.elapsed_time(b) / wl.launches_per_step for a, b in idle]
    mean = sum(per) / n
    sd = (sum((x - mean) ** 2 for x in per) / n) ** 0.5
    return {"launches": n, "min_ms": round(per[0], 4), "median_ms": round(per[n // 2], 4), "p95_ms": round(per[min(n - 1, int(0.95 * n))], 4), "max_ms": round(per[-1], 4),
            "mean_ms": round(mean, 4), "stddev_ms": round(sd, 4), "stddev_frac": round(sd / mean, 4), "over_1p2x_median": sum(1 for x in per if x > 1.2 * per[n // 2]),
            "after_50ms_idle": {"first_ms": round(after_idle[0], 4), "max_ms": round(max(after_idle), 4), "mean_of_20_ms": round(sum(after_idle) / 20, 4)}}


def live_traffic(timeout_s=45):
    """HBM bytes per launch of the headline kernel, measured NOW: two child runs of this same script under `rocprofv3 --kernel-trace --pmc FETCH_SIZE`
    and `... WRITE_SIZE` (separate passes, kernel trace only -- MI355X_MICROARCH.md's recipe), a few steps each; FETCH_SIZE doubled per the guide's
    gfx950 note (units of 1 KiB).  Returns (bytes per launch, detail) or (None, reason) -- the committed figure then stays in the line, labelled as such."""
    import csv as _csv, shutil, subprocess, tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    vals = {}
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="ipk_pmc_", dir="/tmp")
        try:
            cmd = [exe, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable, os.path.abspath(__file__),
                   "--no-cpu-baseline", "--no-check", "--no-extras", "--steps", "5", "--warmup", "1", "--prewarm-ms", "0"]
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            fs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not fs:
                return None, "%s pass failed (rc %s)" % (ctr, r.returncode)
            got = [float(row["Counter_Value"]) for row in _csv.DictReader(open(fs[0]))
                   if "k_fused_bayer<" in row.get("Kernel_Name", "") and row.get("Counter_Name") == ctr]
            if not got:
                return None, "%s: no rows for the kernel" % ctr
            vals[ctr] = sum(got) / len(got)
        except Exception as e:
            return None, "%s pass: %r" % (ctr, e)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    total = vals["FETCH_SIZE"] * 2048.0 + vals["WRITE_SIZE"] * 1024.0
    return total, {"fetch_bytes": round(vals["FETCH_SIZE"] * 2048.0), "write_bytes": round(vals["WRITE_SIZE"] * 1024.0),
                   "method": "two child runs of this script under rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes), "
                             "per-launch mean; FETCH_SIZE x 2 per MI355X_MICROARCH.md's gfx950 note"}


def _hwmon_for(torch, dev):
    """the amdgpu hwmon directory of HIP device `dev` (matched by PCI bus id; a box with one GPU: its only card), or None"""
    cands = []
    for d in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
        hm = glob.glob(os.path.join(d, "hwmon", "hwmon*"))
        if not hm or not (os.path.exists(os.path.join(hm[0], "power1_average")) or os.path.exists(os.path.join(hm[0], "power1_input"))):
            continue
        slot = None
        try:
            for line in open(os.path.join(d, "uevent")):
                if line.startswith("PCI_SLOT_NAME="):
                    slot = line.strip().split("=", 1)[1].lower()
        except OSError:
            pass
        cands.append((slot, hm[0]))
    if not cands:
        return None
    try:
        pr = torch.cuda.get_device_properties(dev)
        want = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        for slot, hm in cands:
            if slot and slot.startswith(want):
                return hm
    except Exception:
        pass
    return cands[0][1] if len(cands) == 1 else None


def clock_power_leg(ctx, wl, seconds=1.0):
    """The box's state under THIS workload, so that a slower box and a slower kernel can be told apart (the kernel is clock / power bound, DESIGN.md
    section 4): the same step back to back for ~`seconds` in the pre-warmed steady state, while
      - a one-wave kernel on a second stream (ipk_clock_probe) counts shader-clock cycles (s_memtime) against the fixed 100 MHz reference (s_memrealtime)
        over 20 ms spans BESIDE the running launches: shader_clock_GHz, measured on the device itself;
      - a host thread reads the amdgpu hwmon files of the device (socket power, sclk) every 10 ms.
    Outside the timed region (a sampling thread and a co-resident wave have no business inside it); same launches, same clock regime."""
    import threading
    torch = ctx.torch
    import imagepipe_amd as ipa
    L = ipa.lib()
    out = {}
    hm = _hwmon_for(torch, torch.cuda.current_device())
    pfile = None
    if hm:
        for n in ("power1_average", "power1_input"):
            if os.path.exists(os.path.join(hm, n)):
                pfile = os.path.join(hm, n)
                break
    ffile = os.path.join(hm, "freq1_input") if hm and os.path.exists(os.path.join(hm, "freq1_input")) else None
    samples, stop = [], threading.Event()

    def poll():
        while not stop.is_set():
            try:
                pw = float(open(pfile).read()) / 1e6 if pfile else None
                fq = float(open(ffile).read()) / 1e9 if ffile else None
                samples.append((pw, fq))
            except Exception:
                pass
            stop.wait(0.01)
    side = torch.cuda.Stream()
    res = torch.zeros(2 * 64, dtype=torch.int64, device="cuda")
    n_probe = 0
    for _ in range(30):
        wl.step()
    th = threading.Thread(target=poll)
    th.start()
    t0 = time.perf_counter()
    k = 0
    while time.perf_counter() - t0 < seconds:
        if n_probe < 64 and k % 2 == 0:
            rc = L.ipk_clock_probe(res.data_ptr() + 16 * n_probe, 20000, side.cuda_stream)
            n_probe += 1 if rc == 0 else 0
        for _ in range(40):
            wl.step()
        torch.cuda.synchronize()
        k += 1
    stop.set(); th.join()
    torch.cuda.synchronize()
    r = res.cpu().numpy().reshape(-1, 2)[:n_probe]
    ghz = sorted(float(c) / float(t) / 10.0 for c, t in r if t > 0)
    if ghz:
        out["shader_clock_GHz"] = round(ghz[len(ghz) // 2], 3)
        out["shader_clock_GHz_min_max"] = [round(ghz[0], 3), round(ghz[-1], 3)]
    pw = sorted(p for p, _ in samples if p is not None)
    fq = sorted(f for _, f in samples if f is not None)
    if pw:
        busy = pw[len(pw) // 5:]                                   # the lowest fifth: ramp-up and the gaps at the synchronisations
        out["socket_power_W"] = round(sum(busy) / len(busy), 1)
        out["socket_power_W_max"] = round(pw[-1], 1)
    if fq:
        out["sclk_sysfs_GHz"] = round(fq[len(fq) // 2], 3)
    out["clock_power_method"] = ("%d ipk_clock_probe spans of 20 ms (s_memtime / s_memrealtime) on a second stream beside %.1f s of back-to-back launches of the timed "
                                 "workload; %d hwmon samples (%s) at 10 ms" % (len(ghz), seconds, len(samples), os.path.basename(pfile) if pfile else "no power file"))
    return out


def host_boundary_leg(ctx, ipa, util, n=6, check=True):
    """The boundary a drop-in actually crosses (Rust Vecs in, Vecs out): BASELINE.json configs[1]'s 24 MP frame as u16 sensor data in HOST
    memory -> f32 / 8-bit sRGB in HOST memory through ipk_host_pipeline_run (one frame, synchronous: upload + kernel + download in turn) and
    ipk_host_pipeline_run_batch (three streams over two device slots: per-frame cost tends to the slowest of the three).  Page-locked buffers
    (ipk_host_alloc).  pcie_floor_ms = the bytes of the busier direction / 63 GB/s (PCIe 5.0 x16, one direction, full duplex assumed); frac = floor / measured.
    link_measured: the same two transfers as bare hipMemcpyAsync calls, alone and both at once -- on these boxes the directions do NOT add up (57 GB/s each way
    alone, ~61-64 GB/s combined), so frac_of_measured_link (one upload + one download at once / the batch's time per frame) is the honest yardstick: ~1.1-1.2,
    the deeper queue of the batch gets slightly more out of the link than one pair of copies does."""
    import numpy as np
    L = ipa.lib()
    W, H = 6000, 4000
    PCIE_GBS = 63.0
    img = ipa.RawImage(width=W, height=H, data=None, cfa="RGGB", blacklevels=[util.BLACK] * 4, whitelevels=[util.WHITE] * 4,
                       wb_coeffs=util.WB, cam_to_xyz_normalized=util.cam_matrix())
    d = ipa.Pipeline(img).desc()
    in_bytes = W * H * 2
    frames = [util.noise_u16(util.SEED + 40 + i, H, W) for i in range(2)]
    sp = [L.ipk_host_alloc(in_bytes) for _ in range(n)]
    out = {"frame": [W, H], "src": "u16 host buffer (page-locked)", "frames": n, "pcie_GBps_assumed": PCIE_GBS}
    dp = []
    try:
        if not all(sp):
            raise RuntimeError("ipk_host_alloc failed")
        for i, q in enumerate(sp):
            ctypes.memmove(q, frames[i % 2].ctypes.data, in_bytes)
        torch = ctx.torch
        s_up, s_dn = torch.cuda.Stream(), torch.cuda.Stream()
        d_up = torch.empty(in_bytes, dtype=torch.uint8, device="cuda")

        def link(ob, host_dst):
            """what the link itself gives these two transfers (hipMemcpyAsync, page-locked): each alone, and both at once on two streams"""
            d_dn = torch.empty(ob, dtype=torch.uint8, device="cuda")
            up = lambda: L.ipk_memcpy_h2d(d_up.data_ptr(), sp[0], in_bytes, s_up.cuda_stream)
            dn = lambda: L.ipk_memcpy_d2h(host_dst, d_dn.data_ptr(), ob, s_dn.cuda_stream)

            def tm(fn, reps=4):
                fn(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    fn()
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / reps * 1e3
            a, b, c = tm(up), tm(dn), tm(lambda: (up(), dn()))
            del d_dn
            return {"up_alone_ms": round(a, 3), "down_alone_ms": round(b, 3), "both_at_once_ms": round(c, 3), "up_GBps": round(in_bytes / a / 1e6, 1), "down_GBps": round(ob / b / 1e6, 1),
                    "combined_GBps_both_at_once": round((in_bytes + ob) / c / 1e6, 1)}
        for name, ot, esz in (("to_u8", ipa.OUT_U8, 1), ("to_f32", ipa.OUT_F32, 4)):
            ob = W * H * 3 * esz
            dp = [L.ipk_host_alloc(ob) for _ in range(n)]
            if not all(dp):
                raise RuntimeError("ipk_host_alloc failed")
            srcs = (ctypes.c_void_p * n)(*sp); dsts = (ctypes.c_void_p * n)(*dp)
            used = ctypes.c_int(0)

            def one():
                rc = L.ipk_host_pipeline_run(ctypes.byref(d), sp[0], dp[0], ot, ctypes.byref(used))
                assert rc == 0, L.ipk_last_error()

            def batch():
                rc = L.ipk_host_pipeline_run_batch(ctypes.byref(d), srcs, dsts, n, ot, ctypes.byref(used))
                assert rc == 0, L.ipk_last_error()
            one(); batch()                                          # warm: lanes, device slots
            t0 = time.perf_counter()
            for _ in range(3):
                one()
            t_one = (time.perf_counter() - t0) / 3 * 1e3
            t0 = time.perf_counter()
            for _ in range(2):
                batch()
            t_b = (time.perf_counter() - t0) / (2 * n) * 1e3
            floor_serial = (in_bytes + ob) / (PCIE_GBS * 1e9) * 1e3
            floor_overlap = max(in_bytes, ob) / (PCIE_GBS * 1e9) * 1e3
            e = {"ms_per_frame_single": round(t_one, 3), "ms_per_frame_batched": round(t_b, 3), "MP_per_s_batched": round(W * H / 1e6 / (t_b * 1e-3), 1),
                 "pcie_bytes_per_frame": {"up": in_bytes, "down": ob},
                 "pcie_floor_ms": {"single (up + down in turn)": round(floor_serial, 3), "batched (directions overlap)": round(floor_overlap, 3)},
                 "frac_of_pcie_floor": {"single": round(floor_serial / t_one, 4), "batched": round(floor_overlap / t_b, 4)},
                 "used_fused": bool(used.value)}
            # the floor above assumes a full-duplex link; THIS box's link, measured now: the two directions share most of one direction's rate
            lk = link(ob, dp[0])
            e["link_measured"] = lk
            e["moved_GBps_batched"] = round((in_bytes + ob) / t_b / 1e6, 1)
            e["frac_of_measured_link"] = round(lk["both_at_once_ms"] / t_b, 4) if t_b > 0 else None
            if check and name == "to_u8":
                import oracle
                od = oracle.make_pipeline(frames[(n - 1) % 2], cfa="RGGB", blacklevels=[util.BLACK] * 4, whitelevels=[util.WHITE] * 4, wb_coeffs=util.WB,
                                          cam_to_xyz_normalized=util.cam_matrix())
                want = oracle.pipeline_output_8bit(od).reshape(-1)
                got = np.frombuffer((ctypes.c_char * ob).from_address(dp[n - 1]), dtype=np.uint8)
                if not np.array_equal(got, want):
                    raise AssertionError("host boundary: the batch's last 8-bit frame differs from the oracle's output_8bit")
                e["parity_check"] = "last frame of the batch: all %d samples equal the oracle's output_8bit" % want.size
            out[name] = e
            for q in dp:
                L.ipk_host_free(q)
            dp = []
        out["device_resident_ms_per_frame"] = "roofline.kernel_ms of --config c2 (0.116 ms, profiles/): the PCIe-inclusive figures above are never `value`"
    finally:
        for q in sp + dp:
            if q:
                L.ipk_host_free(q)
    return out


def valu_model(kernel_ms, data):
    """The VALU-issue account of the fused kernel from the tracked file profiles/r*_valu_model.json (tools/evidence_r03.sh):
    dynamic instruction counts per launch by class (rocprofv3 PMC), priced two ways with tools/ubench2.hip's figures -- `interleaved`: at the cost
    of one instruction of a loop with the kernel's kind of mix, where half-rate instructions hide behind full-rate ones (the issue time if the
    kernel interleaved as well); `additive`: every class at the cost of a loop of nothing else (an upper bound) -- plus the model-free wave-cycle
    ratio and the stall split.  frac = interleaved issue time / this run's kernel time."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_valu_model.json")))
    if not files:
        return None
    try:
        m = json.load(open(files[-1]))
        if data not in m:                                     # round-2 layout: one model, additive pricing only
            pred = m["predicted_ms"]
            return {"bound": "valu-issue", "issue_ms": {"additive": pred}, "measured_ms": round(kernel_ms, 4), "frac": round(pred / kernel_ms, 4),
                    "source": os.path.relpath(files[-1], ROOT)}
        d = m[data]
        out = {"bound": "valu-issue", "valu_per_pixel": d.get("valu_per_pixel"), "valu_insts_per_launch": d.get("valu_total_per_launch"),
               "issue_ms": {"interleaved": d.get("issue_ms_interleaved"), "additive": d.get("issue_ms_additive")}, "measured_ms": round(kernel_ms, 4),
               "frac": round(d["issue_ms_interleaved"] / kernel_ms, 4), "frac_additive": round(d["issue_ms_additive"] / kernel_ms, 4),
               "wave_cycles_per_valu": d.get("wave_cycles_per_valu"), "stall_split": d.get("stall_split"),
               "note": "frac: the time the VALU needs to issue the kernel's instructions if half-rate ones hide behind full-rate ones as in a pure-VALU loop of the "
                       "same mix, over the measured time; frac_additive prices every class alone (upper bound); wave_cycles_per_valu.ratio is model-free",
               "source": os.path.relpath(files[-1], ROOT)}
        return out
    except Exception as e:      # a malformed evidence file must not take the bench line down
        return {"error": repr(e)}


def main():
    args = parse()
    if args.band_child:
        return band_child(args)
    if args.sp_child is not None:
        return sp_child(args)
    if args.single_process:
        return main_single_process(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return relaunch_as_ranks(args)
    ctx = Ctx(args)
    torch = ctx.torch
    import imagepipe_amd as ipa
    import util
    ipa.init(ctx.local_rank)
    rank, world = ctx.rank, ctx.world

    cW, cH, cfa, maxw, cB, cidx = CONFIGS[args.config]
    # IPK_BENCH_DEV_SMALL=1 (development / tests only): the default run's whole control flow -- extras, the batch leg, the gather -- on
    # frames small enough for ranks that share one GPU; the numbers of such a run mean nothing
    dev_small = os.environ.get("IPK_BENCH_DEV_SMALL") == "1"
    if dev_small and args.config == "c3":
        cW, cH = 2048, 1024
    W = args.width or cW
    H = args.height or cH
    B = args.batch if args.batch is not None else cB
    weak = B is None
    if weak:
        B = world
    if args.config in ("c5", "c5b") and (args.width is None and args.height is None):
        return main_xtrans(args, ctx, ipa, util, W, H, cfa, maxw)

    # output_8bit forces linear = false, output_16bit linear = true (src/pipeline.rs:405, :452); Pipeline::run (f32) takes the setting
    linear = {"f32": args.linear, "u8": False, "u16": True}[args.out]
    curve_kw = dict(points=CURVES[args.curve], exposure=args.exposure, linear=linear, schedule={"auto": 0, "split": 1}[args.schedule])
    wl = FusedBatch(ctx, ipa, util, W, H, B, cfa, args.src, args.out, args.data, util.SEED + 2, **curve_kw)

    # ---- correctness check against the CPU oracle (outside the timed region) ----
    checked = None
    if not args.no_check and rank == 0 and cfa == "RGGB" and wl.mine:
        checked = oracle_check(ctx, util, wl, rows=None if B <= world else 64)

    extras = (not args.no_extras) and args.config == "c3" and args.width is None and args.height is None and args.batch is None
    cold_ms = None
    if extras or ((not args.no_extras) and args.config == "c2" and args.width is None and args.height is None and args.batch is None):   # the single 24 MP frame too
        # the same K steps straight after idle (rank 0 has just spent seconds in the parity check), no clock pre-warm
        ctx.barrier(); time.sleep(0.5)
        _, cold_ms, _ = timed(ctx, wl.step, args.steps, 0, 0.0)

    elapsed, mean_ms, median_ms = timed(ctx, wl.step, args.steps, args.warmup, args.prewarm_ms)
    kernel_ms = mean_ms / wl.launches_per_step                  # HIP events on the launch stream: average launch duration over the timed region
    mp_total = args.steps * B * (H * W) / 1e6
    value = mp_total / elapsed
    alg_bytes = wl.alg_bytes_per_launch()
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9

    result = {
        "metric": "megapixels/sec full raw->sRGB pipe, 100 MP f32 frame",
        "value": round(value, 1), "unit": "MP/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak" if weak else "strong", "vs_baseline": None,
        "value_basis": "all frames' pixels / wall time of the K timed steps (barrier + synchronize on both sides, max over ranks), i.e. the MEAN step; "
                       "the median step is roofline.kernel_ms_median, the run without clock pre-warm config.cold_ms",
        "value_from_median_step": round(B * (H * W) / 1e6 / (median_ms * 1e-3), 1),
        "dtype": "f32", "data": "synthetic (%s, 14-bit %s sensor values, torch Philox seed 0x%X+frame)" % (args.data, cfa if len(cfa) == 4 else "X-Trans", util.SEED + 2),
        "config": {"workload": "%dx%d (%.0f MP) synthetic %s %s mosaic -> fused gofloat+demosaic+tolab+basecurve+fromlab+gamma -> %s RGB, %s"
                               % (W, H, H * W / 1e6, cfa if len(cfa) == 4 else "X-Trans", args.src, args.out,
                                  "one frame per GPU per step" if weak else "%d frames per step, frame i on rank i mod N" % B),
                   "baseline_config": "BASELINE.json configs[%d]" % cidx,
                   "frame": [W, H], "src": args.src, "out": args.out, "frames_per_step": B, "prewarm_ms": args.prewarm_ms,
                   "schedule": args.schedule, "curve": args.curve, "curve_points": [list(p) for p in CURVES[args.curve]], "exposure": args.exposure, "linear": linear,
                   "sharding": "independent frames, no data-path collective", "host_glibc": glibc_version(),
                   "host_cbrtf_matches_device": ipa.lib().ipk_host_libm_matches(None) == 1},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                     "kernel": "k_fused_bayer", "kernel_ms": round(kernel_ms, 4), "kernel_ms_median": round(median_ms / wl.launches_per_step, 4),
                     "algorithmic_bytes_per_launch": alg_bytes},
    }
    if world > 1:
        # what a reader of an N > 1 line must not miss: `value` is weak scaling of independent frames (one frame per GPU per step, nothing exchanged), so it
        # grows ~N x by construction and proves nothing about a node; the judgeable object is `scale` (BASELINE.json configs[3]: the SAME 64 x 24 MP batch on
        # N GPUs against one GPU, compute-only and with the results gathered, each beside its xGMI expectation)
        result["value_basis"] = (("WEAK scaling, no data-path collective: every rank runs its own %dx%d frame per step, value = N frames' pixels / wall time of the K timed "
                                  "steps (barrier + synchronize on both sides, max over ranks) -- ~N x the one-GPU value by construction.  " % (W, H) if weak else
                                  "STRONG scaling of %d independent %dx%d frames per step, frame i on rank i mod N, nothing exchanged: value = their pixels / wall time of the K "
                                  "timed steps (barrier + synchronize on both sides, max over ranks), results left on the GPUs that computed them.  " % (B, W, H)) +
                                 "The multi-GPU claim is the `scale` object (one fixed batch: compute_only / gather_to_root_u8 / all_gather_f32, each measured against one GPU "
                                 "in this run and beside expected_ms from the xGMI link arithmetic), not this number")
    if cold_ms is not None:
        result["config"]["cold_ms"] = round(cold_ms / wl.launches_per_step, 4)
    if cold_ms is not None and not dev_small and not args.no_box_state:
        # the box's state under this workload, straight behind the timed region (the clock is where the timed steps left it)
        try:
            result["config"].update(clock_power_leg(ctx, wl))
            if "shader_clock_GHz" in result["config"]:
                # the box-independent figure: shader cycles one launch takes.  The kernel is clock / power bound, so kernel_ms moves with the clock the
                # box's firmware grants under the 1400 W cap (2.13-2.25 GHz on the boxes of one day) and this product far less (0.99-1.02 M; the clock is probed behind the timed region, not inside it)
                result["roofline"]["kernel_Mcycles"] = round(kernel_ms * result["config"]["shader_clock_GHz"], 4)
        except Exception as e:
            ctx.fail("clock_power", repr(e))
    if checked:
        result["parity_check"] = checked
    # HBM traffic per launch: not measurable from inside this process (PMC counters need rocprofv3); taken from the
    # committed rocprofv3 passes of this same command (tools/profile.sh -> profiles/<round>_counters.json: separate
    # --pmc FETCH_SIZE / WRITE_SIZE runs, FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 note) when the workload matches.
    headline = (W, H, args.src, args.out, args.data, weak) == (10000, 10000, "f32", "f32", "noise", True) and not dev_small
    if rank == 0 and headline:
        profs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_counters.json")))
        if profs:
            try:
                pj = json.load(open(profs[-1]))
                if "hbm_traffic_bytes_per_launch" in pj:
                    result["roofline"]["traffic"] = round(pj["hbm_traffic_bytes_per_launch"])
                    result["roofline"]["traffic_source"] = os.path.relpath(profs[-1], ROOT)
            except Exception:
                pass
        vm = valu_model(kernel_ms, args.data)
        if vm:
            result["roofline_valu"] = vm
        if extras and world == 1 and not args.no_live_traffic and os.environ.get("IPK_BENCH_NO_LIVE_PMC") != "1":
            # measured in THIS run when rocprofv3 is there (it is on the GPU boxes); the committed figure above stays only if the passes fail
            tb, det = live_traffic()
            if tb is not None:
                result["roofline"]["traffic"] = round(tb)
                result["roofline"]["traffic_source"] = "two child runs of this command under rocprofv3 (5 steps, 1 warm-up, no clock pre-warm), started by this run"
                result["roofline"]["traffic_detail"] = det
                result["roofline"]["traffic_over_algorithmic"] = round(tb / alg_bytes, 4)
            else:
                result["roofline"]["traffic_live_failed"] = det

    if args.kernel_stats > 0 or extras:
        result["roofline"]["launch_stats"] = launch_stats(ctx, wl, args.kernel_stats if args.kernel_stats > 0 else (40 if dev_small else 500))
    if (not args.no_extras) and args.out == "f32" and cfa == "RGGB":
        # the ceiling for THIS access pattern and read:write mix (4 B read + 12 B nontemporal write per pixel on the kernel's own strip / row walk),
        # measured in this session by the kernel's memory skeleton as a launch of its own; the 1:1 copy ceiling below is a different mix
        ce = ceiling_leg(ctx, wl, args.steps, args.warmup)
        if ce is not None:
            result["roofline"]["ceiling_ms"] = round(ce[0], 4)
            result["roofline"]["ceiling_ms_median"] = round(ce[1], 4)
            result["roofline"]["ceiling_GBps"] = round(alg_bytes / (ce[0] * 1e-3) / 1e9, 1)
            result["roofline"]["ceiling_frac_of_peak"] = round(alg_bytes / (ce[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            result["roofline"]["frac_of_ceiling"] = round(ce[0] / kernel_ms, 4)
            result["roofline"]["ceiling_kernel"] = "ipk_stream_probe = k_fused_bayer<..., 4, ...>: the fused kernel's launch, loads, OpGoFloat, demosaic, staging and stores without OpToLab..OpGamma"
    if extras:
        own, tch, mixed = copy_ceiling(ctx, (12 if dev_small else 1200) * 1000 * 1000)     # every rank (keeps the ranks in step)
        result["roofline"]["copy_ceiling_GBps"] = round(max(own, tch), 1)
        result["roofline"]["copy_ceiling_detail"] = {"ipk_copy_probe_GBps": round(own, 1), "torch_copy_GBps": round(tch, 1),
                                                     "guide_float4_copy_GBps": 6290.0, "bytes": (12 if dev_small else 1200) * 1000 * 1000}
        result["roofline"]["frac_of_copy_ceiling"] = round(achieved / result["roofline"]["copy_ceiling_GBps"], 4)
        result["roofline"]["mix_ceiling_GBps"] = round(mixed, 1)          # a flat kernel with the fused path's 1 : 3 read : write mix and no arithmetic (ipk_mix_probe)
        result["roofline"]["mix_ceiling_frac_of_peak"] = round(mixed / HBM_PEAK_GBS, 4)
        result["roofline"]["frac_of_mix_ceiling"] = round(achieved / mixed, 4)
        other = {}
        for kind in ("smooth", "photo"):
            w2 = FusedBatch(ctx, ipa, util, W, H, world, cfa, args.src, args.out, kind, util.SEED + 2)
            _, m2, md2 = timed(ctx, w2.step, max(5, args.steps // 2), 2, 120.0)     # generating the frame let the clock drop: its own short pre-warm
            other[kind] = {"kernel_ms": round(m2, 4), "kernel_ms_median": round(md2, 4), "frac": round(alg_bytes / (m2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            del w2
            if kind == "photo":
                # the same frame under IPK_SCHED_SPLIT (a caller's choice for frames with blown regions; bit-identical results)
                w3 = FusedBatch(ctx, ipa, util, W, H, world, cfa, args.src, args.out, kind, util.SEED + 2, schedule=1)
                _, m3, md3 = timed(ctx, w3.step, max(5, args.steps // 2), 2, 120.0)
                other[kind]["schedule_split"] = {"kernel_ms": round(m3, 4), "kernel_ms_median": round(md3, 4), "frac": round(alg_bytes / (m3 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                del w3
        result["other_data"] = other
        # BASELINE.json configs[3]: 64 x 24 MP frames, frame i -> rank i mod N, compute-only and with the all-gather of the results
        del wl
        torch.cuda.empty_cache()
        bw, bh, bn = (600, 400, 8) if dev_small else (6000, 4000, 64)
        result["batch_64x24MP"] = batch_mode(ctx, ipa, util, bw, bh, bn, "f32", "f32", args.data, steps=5, warmup=1, gather=world > 1)
        if "scale" in result["batch_64x24MP"]:
            result["scale"] = result["batch_64x24MP"].pop("scale")     # first-class in the N > 1 line
        if world > 1:
            # the same batch over the same N devices from ONE process (rank 0's child; the ranks wait): the shape a drop-in has
            torch.cuda.empty_cache()
            sp = sp_children(ctx, args)
            if rank == 0 and "scale" in result:
                result["scale"]["single_process"] = sp
    if world == 1 and ((extras and not dev_small) or args.host_boundary):
        try:
            result["host_boundary"] = host_boundary_leg(ctx, ipa, util, check=not args.no_check)
        except Exception as e:
            ctx.fail("host_boundary", repr(e))

    if extras and world > 1:
        # ONE frame row-sharded over the N GPUs on the library's RCCL transport, in child processes (so that it cannot cost this line)
        torch.cuda.empty_cache()
        result["band_mode"] = band_children(ctx, args, W, H)
        ctx.barrier()

    if args.config == "c4" or (args.batch is not None and args.batch > world):
        result["with_gather"] = gather_leg(ctx, wl, steps=max(2, args.steps // 4)) if world > 1 else {
            "note": "one GPU: every result is already resident on it, the gather is a no-op"}
        if world > 1:
            result["scale"] = scale_object(ctx, ipa, util, wl, {"ms_per_step": round(elapsed / args.steps * 1e3, 3), "with_gather": result["with_gather"]},
                                           args.src, args.out, args.data, max(2, args.steps // 2), 1)

    # ---- optional: one frame sharded by row bands over the ranks, 1-row halo exchange (P2P) before every launch ----
    if args.band and weak and not extras:
        result["band_mode"] = band_leg(ctx, ipa, util, wl, args)

    # ---- CPU baseline: the oracle's reference-shaped pipeline (unfused, one task per row) on this host ----
    if rank == 0 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(util, args.data, args.cpu_seconds)
    if world > 1:
        try:
            result["config"]["ipk_comm"] = ctx.comm_info()      # what the library's own transport saw (ipk_comm_info): rank count and transport
            if result["config"]["ipk_comm"]["ranks"] != world:
                ctx.fail("ipk_comm", "the library's communicator has %d ranks, the job %d" % (result["config"]["ipk_comm"]["ranks"], world))
        except Exception as e:
            result["config"]["ipk_comm"] = {"ok": False, "error": repr(e)}
            ctx.fail("ipk_comm", repr(e))
    if world > 1:
        result["scale_is_the_claim"] = "scale" in result          # value_basis points at `scale`: say whether this run carries one
        if "scale" not in result:
            result["value_basis"] += " -- THIS run carries no `scale` object (it is emitted by the default run, --config c4 or --batch > N)"
    result["failed_legs"] = ctx.failed
    assert result["n_gpus"] == args.gpus, (result["n_gpus"], args.gpus)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if ctx.dist is not None:
        ctx.dist.barrier()                  # the other ranks wait here while rank 0 runs the CPU baseline leg
        ctx.dist.destroy_process_group()


class MultiBatch:
    """B frames of W x H, frame i resident on device-set member i mod N, run from THIS process through ipk_pipeline_run_batch_multi (one internal stream
    per member, launches issued from one host thread) + ipk_devices_sync -- the batch split of DESIGN.md section 6 as a drop-in would call it."""

    def __init__(self, torch, ipa, util, members, W, H, B, data, seed0, out_type=None):
        self.torch, self.ipa, self.members, self.W, self.H, self.B = torch, ipa, members, W, H, B
        self.L = ipa.lib()
        out_type = ipa.OUT_F32 if out_type is None else out_type
        self.out_type = out_type
        odt = {ipa.OUT_F32: torch.float32, ipa.OUT_U8: torch.uint8, ipa.OUT_U16: torch.int16}[out_type]
        img = ipa.RawImage(width=W, height=H, data=None, cfa="RGGB", is_float=True, blacklevels=[util.BLACK] * 4, whitelevels=[util.WHITE] * 4,
                           wb_coeffs=util.WB, cam_to_xyz_normalized=util.cam_matrix())
        self.desc = ipa.Pipeline(img).desc()
        n = len(members)
        self.srcs, self.dsts = [], []
        for i in range(B):
            dev = "cuda:%d" % members[i % n].device
            with torch.cuda.device(members[0].device):
                f = synth_frame(torch, H, W, data, seed0 + i).to(torch.float32).reshape(-1)
            self.srcs.append(f.to(dev).contiguous())
            del f
            self.dsts.append(torch.empty(H * W * 3, dtype=odt, device=dev))
        for m in members:
            torch.cuda.synchronize(m.device)
        self._s = (ctypes.c_void_p * max(B, 1))(*[t.data_ptr() for t in self.srcs])
        self._d = (ctypes.c_void_p * max(B, 1))(*[t.data_ptr() for t in self.dsts])
        self.used = ctypes.c_int(0)

    def step(self):
        rc = self.L.ipk_pipeline_run_batch_multi(ctypes.byref(self.desc), self._s, self._d, self.B, self.out_type, ctypes.byref(self.used))
        if rc < 0:
            raise RuntimeError("ipk_pipeline_run_batch_multi: " + self.L.ipk_last_error().decode())

    def sync(self):
        rc = self.L.ipk_devices_sync()
        if rc < 0:
            raise RuntimeError("ipk_devices_sync: " + self.L.ipk_last_error().decode())

    def time(self, steps, warmup, prewarm_ms=0.0):
        t_pw = time.perf_counter()
        while (time.perf_counter() - t_pw) * 1e3 < prewarm_ms:
            self.step(); self.sync()
        for _ in range(warmup):
            self.step()
        self.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        self.sync()
        return (time.perf_counter() - t0) / steps * 1e3


def single_process_scale(torch, ipa, util, members, W, H, B, data, steps, warmup, host_frames=16, check=True):
    """The `scale` object of the single-process mode: BASELINE.json configs[3]'s batch on the N members against the same batch on member 0 alone,
    measured in this process; device-resident (compute_only: results stay on the GPUs that computed them) and host-to-host (8-bit results into the
    caller's page-locked buffers, ipk_host_pipeline_run_batch_multi: one host thread, three streams and two device slots per member)."""
    import numpy as np
    L = ipa.lib()
    n = len(members)
    out = {"mode": "ONE process: one ipk_ctx per device (ipk_init_devices), frame i -> member i mod N, ipk_pipeline_run_batch_multi + ipk_devices_sync; "
                   "no torch.distributed, no RCCL, nothing exchanged between devices (src/pipeline.rs:246-249: frames are independent pipelines)",
           "devices": [m.device for m in members], "n_gpus": n, "distinct_devices": len({m.device for m in members}),
           "workload": "%d x %dx%d RGGB f32 frames (BASELINE.json configs[3])" % (B, W, H)}
    multi = MultiBatch(torch, ipa, util, members, W, H, B, data, util.SEED + 1000)
    tN = multi.time(steps, warmup, 200.0)
    out["fused_batch_launches"] = bool(multi.used.value)
    if check:
        import oracle
        i = B - 1                                                   # the last frame: computed on the last member that got one
        part = multi.srcs[i].cpu().numpy().reshape(H, W)
        desc = oracle.make_pipeline(part, cfa="RGGB", source_kind=1, blacklevels=[util.BLACK] * 4, whitelevels=[util.WHITE] * 4, wb_coeffs=util.WB,
                                    cam_to_xyz_normalized=util.cam_matrix())
        want = torch.from_numpy(oracle.pipeline_run(desc).reshape(-1))
        got = multi.dsts[i].cpu()
        if not torch.equal(got.view(torch.int32), want.view(torch.int32)):
            raise AssertionError("single-process batch: frame %d (device %d) differs from the CPU oracle" % (i, members[i % n].device))
        out["parity_check"] = "frame %d (computed on set member %d, device %d) bit-identical to the CPU oracle, every sample" % (i, i % n, members[i % n].device)
    del multi
    torch.cuda.empty_cache()
    solo = MultiBatch(torch, ipa, util, members[:1], W, H, B, data, util.SEED + 1000)
    solo_set = ipa.init_devices([members[0].device])                # the set shrinks to member 0 for the one-GPU reference ...
    t1 = solo.time(steps, warmup, 200.0)
    del solo
    torch.cuda.empty_cache()
    members2 = ipa.init_devices([m.device for m in members])        # ... and comes back
    out["n1_batch_ms"] = round(t1, 3)
    phys = len({m.device for m in members})                        # contexts that share a GPU share its time: the expectation counts physical devices
    out["compute_only"] = {"ms": round(tN, 3), "speedup": round(t1 / tN, 2), "expected_ms": round(t1 / phys, 3), "expected_speedup": float(phys),
                           "MP_per_s": round(B * W * H / 1e6 / (tN * 1e-3), 1)}
    # host buffers in, host buffers out (what a Rust caller of output_8bit over a shoot sees)
    try:
        img = ipa.RawImage(width=W, height=H, data=None, cfa="RGGB", blacklevels=[util.BLACK] * 4, whitelevels=[util.WHITE] * 4,
                           wb_coeffs=util.WB, cam_to_xyz_normalized=util.cam_matrix())
        d = ipa.Pipeline(img).desc()
        nb = host_frames
        in_b, out_b = W * H * 2, W * H * 3
        sp = [L.ipk_host_alloc(in_b) for _ in range(min(nb, 4))]
        dp = [L.ipk_host_alloc(out_b) for _ in range(nb)]
        if not all(sp) or not all(dp):
            raise RuntimeError("ipk_host_alloc failed")
        for k, q in enumerate(sp):
            f = util.noise_u16(util.SEED + 40 + k, H, W)
            ctypes.memmove(q, f.ctypes.data, in_b)
        srcs = (ctypes.c_void_p * nb)(*[sp[k % len(sp)] for k in range(nb)]); dsts = (ctypes.c_void_p * nb)(*dp)

        def run(fn):
            ts = []
            for rep in range(5):                                    # the first call builds the lanes; the median of the other four is reported
                t0 = time.perf_counter()
                rc = fn(ctypes.byref(d), srcs, dsts, nb, ipa.OUT_U8, None)
                if rc < 0:
                    raise RuntimeError(L.ipk_last_error().decode())
                ts.append((time.perf_counter() - t0) * 1e3)
            ts = sorted(ts[1:])
            return 0.5 * (ts[1] + ts[2])
        tNh = run(L.ipk_host_pipeline_run_batch_multi)
        first = np.frombuffer((ctypes.c_char * out_b).from_address(dp[0]), dtype=np.uint8).copy()
        with members2[0]:
            t1h = run(L.ipk_host_pipeline_run_batch)
        same = bool(np.array_equal(first, np.frombuffer((ctypes.c_char * out_b).from_address(dp[0]), dtype=np.uint8)))
        out["host_in_host_out_u8"] = {"frames": nb, "ms": round(tNh, 3), "n1_ms": round(t1h, 3), "speedup": round(t1h / tNh, 2),
                                      "ms_per_frame": round(tNh / nb, 3), "pcie_bytes_per_frame": {"up": in_b, "down": out_b},
                                      "same_bytes_as_one_context": same,
                                      "entry_point": "ipk_host_pipeline_run_batch_multi (u16 host frames -> 8-bit sRGB host frames, page-locked), against ipk_host_pipeline_run_batch on member 0"}
        for q in sp + dp:
            L.ipk_host_free(q)
    except Exception as e:
        out["host_in_host_out_u8"] = {"ok": False, "error": repr(e)}
    return out


def sp_child(args):
    """child process of the N > 1 default run: the single-process mode over the same N devices while the ranks wait (its own process: nothing it does
    can cost the parent its line)"""
    import torch
    import imagepipe_amd as ipa
    import util
    devs = [int(x) for x in args.sp_child.split(",")]
    members = ipa.init_devices(devs)
    small = os.environ.get("IPK_BENCH_DEV_SMALL") == "1"
    bw, bh, bn = (600, 400, 8) if small else (6000, 4000, 64)
    res = single_process_scale(torch, ipa, util, members, bw, bh, bn, args.data, max(3, args.steps // 4), 1, host_frames=4 if small else 16)
    print("SP_CHILD " + json.dumps(res), flush=True)


def sp_children(ctx, args):
    """rank 0 starts sp_child over the job's devices; every rank waits for it at the barrier that follows"""
    import subprocess
    import datetime
    res = None
    ctx.barrier()                                   # every rank's GPU is idle before the child measures on all of them
    if ctx.rank == 0:
        try:
            devs = ",".join(str(i) for i in range(ctx.world)) if not ctx.share else ",".join("0" for _ in range(ctx.world))
            cmd = [sys.executable, os.path.abspath(__file__), "--sp-child", devs, "--steps", str(args.steps), "--data", args.data]
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
            for line in r.stdout.splitlines():
                if line.startswith("SP_CHILD "):
                    res = json.loads(line[len("SP_CHILD "):])
            if res is None:
                res = {"ok": False, "error": "child rc=%s: %s" % (r.returncode, (r.stderr or r.stdout)[-400:])}
        except Exception as e:
            res = {"ok": False, "error": repr(e)}
        if res.get("ok") is False:
            ctx.fail("scale.single_process", res["error"])
    # The other ranks wait on the HOST (the job's key-value store), not in a collective: an RCCL barrier would sit on their GPUs as a spinning kernel for
    # as long as the child measures on those very GPUs.  Whatever happens to the store, every rank then meets at the ordinary barrier.
    try:
        store = ctx.dist.distributed_c10d._get_default_store()
        key = "ipk_sp_done_%d" % sp_children.calls
        sp_children.calls += 1
        if ctx.rank == 0:
            store.set(key, "1")
        else:
            store.wait([key], datetime.timedelta(seconds=900))
    except Exception as e:
        sys.stderr.write("bench.py: rank %d: host-side wait unavailable (%r); waiting in the barrier\n" % (ctx.rank, e))
    ctx.barrier()
    return res


sp_children.calls = 0


def main_single_process(args):
    """python bench.py --gpus N --single-process: the same metric and line from ONE process driving N devices (see MultiBatch).  A step is one 100 MP
    frame per device (weak scaling, as the N-rank line); the judgeable multi-GPU object is `scale` (the 64 x 24 MP batch on N devices against one)."""
    import torch
    import imagepipe_amd as ipa
    import util
    devs = [int(x) for x in args.devices.split(",")] if args.devices else list(range(args.gpus))
    if len(devs) != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but --devices names %d\n" % (args.gpus, len(devs)))
        raise SystemExit(2)
    if max(devs) >= torch.cuda.device_count():
        sys.stderr.write("bench.py: device %d asked for, %d visible\n" % (max(devs), torch.cuda.device_count()))
        raise SystemExit(2)
    members = ipa.init_devices(devs)
    small = os.environ.get("IPK_BENCH_DEV_SMALL") == "1"
    W, H = (2048, 1024) if small else (args.width or 10000, args.height or 10000)
    N = len(members)
    wl = MultiBatch(torch, ipa, util, members, W, H, N, args.data, util.SEED + 2)
    for _ in range(args.warmup):
        wl.step()
    wl.sync()
    ms = wl.time(args.steps, args.warmup, args.prewarm_ms)
    alg = 16.0 * W * H
    result = {
        "metric": "megapixels/sec full raw->sRGB pipe, 100 MP f32 frame", "value": round(N * W * H / 1e6 / (ms * 1e-3), 1), "unit": "MP/s",
        "n_gpus": N, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic (%s, 14-bit RGGB sensor values, torch Philox seed 0x%X+frame)" % (args.data, util.SEED + 2),
        "value_basis": "ONE process, one 100 MP frame per device per step launched from one host thread (ipk_pipeline_run_batch_multi), wall time of the K steps "
                       "between ipk_devices_sync on both sides; weak scaling, nothing exchanged -- the multi-GPU claim is `scale`",
        "config": {"workload": "%dx%d (%.0f MP) synthetic RGGB f32 mosaic -> fused gofloat+demosaic+tolab+basecurve+fromlab+gamma -> f32 RGB, one frame per device per step, single process"
                               % (W, H, W * H / 1e6), "baseline_config": "BASELINE.json configs[2]", "frame": [W, H], "devices": devs, "single_process": True,
                   "distinct_devices": len(set(devs)), "host_glibc": glibc_version()},
        "roofline": {"bound": "hbm", "achieved": round(alg * N / (ms * 1e-3) / 1e9 / len(set(devs)), 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(alg * N / (ms * 1e-3) / 1e9 / len(set(devs)) / HBM_PEAK_GBS, 4), "traffic": None, "kernel": "k_fused_bayer",
                     "kernel_ms": round(ms * len(set(devs)) / N, 4), "algorithmic_bytes_per_launch": alg,
                     "note": "per physical device: wall time per step x distinct devices / frames (host-side clock; the one-GPU line carries the HIP-event figure)"},
    }
    del wl
    torch.cuda.empty_cache()
    if not args.no_extras:
        bw, bh, bn = (600, 400, 8) if small else (6000, 4000, 64)
        try:
            result["scale"] = single_process_scale(torch, ipa, util, members, bw, bh, bn, args.data, max(3, args.steps // 4), 1,
                                                   host_frames=4 if small else 16, check=not args.no_check)
        except Exception as e:
            result["scale"] = {"ok": False, "error": repr(e)}
            result["failed_legs"] = ["scale"]
    result["scale_is_the_claim"] = "scale" in result and result["scale"].get("ok") is not False
    if not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(util, args.data, args.cpu_seconds)
    print(json.dumps(result), flush=True)


def gather_leg(ctx, wl, steps):
    """compute + all-gather of every frame's result to every rank (SURVEY.md 8e prices the f32 gather above the compute; all figures are
    reported).  Three transports of the same bytes, one collective per round of N frames, each enqueued behind its frame's kernel:
      torch_f32  torch.distributed all_gather_into_tensor (RCCL)
      ipk_f32    the library's own entry point, ipk_band_gather over its RCCL communicator: the round's N frames are the N bands of one buffer,
                 every rank's kernel writes its frame in place, ncclAllGather in place (ipk_comm.cpp)
      ipk_u8     the same with the 8-bit output (output_8bit): a quarter of the bytes
    A leg that fails carries "ok": false and its error, and the failure is repeated on stderr and in the line's "failed_legs"."""
    torch, dist = ctx.torch, ctx.dist
    import imagepipe_amd as ipa
    from imagepipe_amd import parallel as par
    out = {}
    rounds = (wl.B + ctx.world - 1) // ctx.world
    if rounds * ctx.world != wl.B:
        return {"ok": False, "error": "batch not a multiple of the rank count"}
    per = wl.H * wl.W * 3
    mp_per_step = wl.B * wl.H * wl.W / 1e6

    def leg(name, fn):
        try:
            out[name] = fn()
            out[name]["ok"] = True
        except Exception as e:
            out[name] = {"ok": False, "error": repr(e)}
            ctx.fail("with_gather." + name, repr(e))

    def torch_f32():
        big = [torch.empty(ctx.world * per, dtype=wl.dsts[0].dtype, device="cuda") for _ in range(rounds)]

        def step():
            for j, (s, d) in enumerate(zip(wl.srcs, wl.dsts)):
                wl.plan.run(s, d, wl.stream)
                dist.all_gather_into_tensor(big[j], d)          # enqueued behind frame j's kernel; frame j+1's launch follows on the same stream
        elapsed, _, _ = timed(ctx, step, steps, 1, 0.0)
        return {"value": round(steps * mp_per_step / elapsed, 1), "unit": "MP/s", "ms_per_step": round(elapsed / steps * 1e3, 3),
                "gathered_bytes_per_rank_per_step": per * wl.dsts[0].element_size() * wl.B,
                "collective": "torch.distributed all_gather_into_tensor of the results, one per round of N frames"}

    def ipk(out_type, dtype, label, root=-1, overlap=False):
        comm = ctx.comm()
        bands = [par.Band(k, k * wl.H, wl.H, k * wl.H, wl.H) for k in range(ctx.world)]     # round buffer = N frames stacked: rank k's frame is band k
        plan = wl.plan if out_type == wl.out_type else ipa.FusedPlan(width=wl.W, height=wl.H, is_float=wl.is_float, black0=wl.black, white0=wl.white, cfa="RGGB",
                                                                     wb_coeffs=wl.wb, cam_to_xyz_normalized=wl.cm, out_type=out_type)
        big = [torch.empty((ctx.world * wl.H, wl.W * 3), dtype=dtype, device="cuda") for _ in range(rounds)]
        mine = [b[ctx.rank * wl.H:(ctx.rank + 1) * wl.H].reshape(-1) for b in big]

        def step():
            for j, s in enumerate(wl.srcs):
                plan.run(s, mine[j], wl.stream)                 # the frame lands in its band of the round's buffer
                # overlap: the gather runs on the communicator's own stream behind this frame's kernel (ipk_band_gather_begin), so that frame j's
                # bytes move while frame j+1 is computed; otherwise it is enqueued on the compute stream and the next kernel waits for it
                comm.gather(big[j], bands, root=root, stream=wl.stream, overlap=overlap)
            if overlap:
                comm.wait(wl.stream)
        elapsed, _, _ = timed(ctx, step, steps, 1, 0.0)
        # every receiving rank now holds every frame of the last round: rank r's band must equal what rank r computed (checked through a checksum exchange)
        sums = [float(big[-1][k * wl.H:(k + 1) * wl.H].to(torch.float64).sum()) for k in range(ctx.world)]
        allsums = [None] * ctx.world
        dist.all_gather_object(allsums, sums)
        own = [allsums[k][k] for k in range(ctx.world)]         # what each rank computed for its own band
        receivers = range(ctx.world) if root < 0 else [root]
        same = all(allsums[r] == own for r in receivers)
        if not same:
            raise RuntimeError("gathered frames differ from what their ranks computed")
        esz = big[0].element_size()
        return {"value": round(steps * mp_per_step / elapsed, 1), "unit": "MP/s", "ms_per_step": round(elapsed / steps * 1e3, 3),
                "gathered_bytes_per_rank_per_step": per * esz * wl.B, "every_rank_holds_the_same_frames": same, "root": root, "overlapped": overlap,
                "collective": "ipk_band_gather%s (%s transport), in place, %s" % ("_begin + ipk_comm_wait" if overlap else "", ctx.comm_info()["transport"], label)}

    leg("torch_f32", torch_f32)
    if wl.out_type == ipa.OUT_F32:
        leg("ipk_f32", lambda: ipk(ipa.OUT_F32, torch.float32, "f32 results"))
    leg("ipk_u8", lambda: ipk(ipa.OUT_U8, torch.uint8, "8-bit results (output_8bit)"))
    leg("ipk_u8_to_root", lambda: ipk(ipa.OUT_U8, torch.uint8, "8-bit results to rank 0 only", root=0))
    if wl.out_type == ipa.OUT_F32:
        leg("ipk_f32_overlapped", lambda: ipk(ipa.OUT_F32, torch.float32, "f32 results, frame k's gather behind frame k+1's kernel", overlap=True))
    leg("ipk_u8_to_root_overlapped", lambda: ipk(ipa.OUT_U8, torch.uint8, "8-bit results to rank 0 only, overlapped", root=0, overlap=True))
    # headline of the object: the library's own f32 gather when it ran, torch's otherwise
    best = out.get("ipk_f32") if out.get("ipk_f32", {}).get("ok") else out.get("torch_f32")
    if best and best.get("ok"):
        out["value"], out["unit"], out["ms_per_step"] = best["value"], "MP/s", best["ms_per_step"]
    out["ok"] = all(v.get("ok") for v in out.values() if isinstance(v, dict))
    return out


def batch_mode(ctx, ipa, util, W, H, B, src_kind, out_kind, data, steps, warmup, gather):
    wl = FusedBatch(ctx, ipa, util, W, H, B, "RGGB", src_kind, out_kind, data, util.SEED + 1000)
    elapsed, mean_ms, median_ms = timed(ctx, wl.step, steps, warmup, 0.0)
    mp = steps * B * H * W / 1e6
    kernel_ms = mean_ms / wl.launches_per_step
    parity = None
    if ctx.rank == 0 and out_kind == "f32" and wl.mine:
        parity = batch_oracle_check(ctx, util, wl)            # untimed: the first and the last frame of the rank, as the timed launch left them
    out = {"config": "BASELINE.json configs[3]: %d x %dx%d RGGB %s frames per step, frame i on rank i mod N, no data-path collective" % (B, W, H, src_kind),
           "value": round(mp / elapsed, 1), "unit": "MP/s", "ms_per_step": round(elapsed / steps * 1e3, 3), "steps": steps, "scaling": "strong",
           "frames_on_rank0": len(wl.mine), "kernel_ms": round(kernel_ms, 4),
           "launch": "ipk_raw_to_srgb_batch: one persistent launch per 64 frames of a rank (kernel_ms = time per frame)" if wl.batch is not None else "one launch per frame",
           "frac": round(wl.alg_bytes_per_launch() / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    if parity:
        out["parity_check"] = parity
    if gather:
        out["with_gather"] = gather_leg(ctx, wl, steps=2)
        out["scale"] = scale_object(ctx, ipa, util, wl, out, src_kind, out_kind, data, steps, warmup)
    return out


XGMI_LINK_GBPS = 153.0      # MI355X_MICROARCH.md: 7 xGMI links per GPU, ~153 GB/s each way, fully connected mesh of 8


def scale_object(ctx, ipa, util, wl, batch, src_kind, out_kind, data, steps, warmup):
    """What the multi-GPU claim is, first-class: the 64-frame batch (BASELINE.json configs[3]) on N GPUs against THE SAME batch on one GPU, measured in
    this run (every rank runs the whole batch alone once; max over ranks), for three deliveries of the results -- left on the GPUs that computed them
    (compute_only: the north-star's >= 6x at 8 GPUs is THIS number; frames are independent pipelines, src/pipeline.rs:246-249), the 8-bit results
    gathered to rank 0, the f32 results all-gathered to every rank -- each with the time the xGMI mesh allows (bytes per link / 153 GB/s, every peer on
    its own link) as expected_ms, so that a run on an 8-GPU node judges itself."""
    class Solo:                                      # the whole batch on this rank alone
        torch = ctx.torch; world = 1; rank = 0
    solo = FusedBatch(Solo, ipa, util, wl.W, wl.H, wl.B, "RGGB", src_kind, out_kind, data, util.SEED + 1000)
    e1, _, _ = timed(ctx, solo.step, steps, warmup, 0.0)
    del solo
    ctx.torch.cuda.empty_cache()
    n, t1 = ctx.world, e1 / steps * 1e3
    tN = batch["ms_per_step"]
    g = batch.get("with_gather", {})
    frames_per_rank = wl.B / n
    px = wl.W * wl.H
    link_ms = lambda bytes_per_frame: frames_per_rank * bytes_per_frame / (XGMI_LINK_GBPS * 1e9) * 1e3     # one peer's frames over one link

    def entry(key, bytes_per_frame, compute_ms, overlapped):
        e = g.get(key) or {}
        comm = link_ms(bytes_per_frame)
        exp = max(compute_ms, comm) if overlapped else compute_ms + comm
        r = {"expected_ms": round(exp, 6), "expected_speedup": round(t1 / exp, 2), "link_ms": round(comm, 9), "measured_by": "with_gather." + key}
        if e.get("ok"):
            r.update({"ms": e["ms_per_step"], "speedup": round(t1 / e["ms_per_step"], 2), "transport": e["collective"]})
        else:
            r.update({"ms": None, "speedup": None, "error": e.get("error", "leg did not run")})
        return r
    ideal = t1 / n
    return {
        "claim": "compute_only: frames are independent pipelines (src/pipeline.rs:246-249), frame i runs on GPU i mod N and its result stays there -- the north-star's "
                 ">= 6x at 8 GPUs is this figure; delivering every result to one rank (8-bit) or to all ranks (f32) is bound by the xGMI links, reported beside it",
        "workload": "%d x %dx%d RGGB %s frames (BASELINE.json configs[3])" % (wl.B, wl.W, wl.H, src_kind), "n_gpus": n,
        "n1_batch_ms": round(t1, 3), "n1_measured": "in this run: every rank ran the whole batch alone (ipk_raw_to_srgb_batch), max over ranks",
        "compute_only": {"ms": tN, "speedup": round(t1 / tN, 2), "expected_ms": round(ideal, 3), "expected_speedup": float(n)},
        "gather_to_root_u8": entry("ipk_u8_to_root", 3.0 * px, ideal, False),
        "gather_to_root_u8_overlapped": entry("ipk_u8_to_root_overlapped", 3.0 * px, ideal, True),
        "all_gather_f32": entry("ipk_f32", 12.0 * px, ideal, False),
        "all_gather_f32_overlapped": entry("ipk_f32_overlapped", 12.0 * px, ideal, True),
        "xgmi_model": {"link_GBps": XGMI_LINK_GBPS, "links_per_gpu": 7,
                       "link_ms": "frames per rank x bytes per frame / link rate: every peer's frames arrive over that peer's own link (fully connected mesh), "
                                  "so a gather to one rank and an all-gather load each link alike; expected_ms = compute/N + link_ms, or their maximum when overlapped"},
    }


def batch_oracle_check(ctx, util, wl):
    """the rank's first and last frame exactly as wl.step() -- the timed path: ipk_raw_to_srgb_batch when the rank has several frames -- left them,
    every sample against the CPU oracle"""
    import numpy as np
    import oracle
    torch = ctx.torch
    for d in wl.dsts:
        d.zero_()
    wl.step(); torch.cuda.synchronize()
    idx = sorted({0, len(wl.srcs) - 1})
    for i in idx:
        part = wl.srcs[i].cpu().numpy().reshape(wl.H, wl.W)
        desc = oracle.make_pipeline(part if wl.is_float else part.view(np.uint16), cfa="RGGB", source_kind=1 if wl.is_float else 0,
                                    blacklevels=[util.BLACK] * 4, whitelevels=[util.WHITE] * 4, wb_coeffs=util.WB, cam_to_xyz_normalized=wl.cm)
        want = torch.from_numpy(oracle.pipeline_run(desc).reshape(-1))
        got = wl.dsts[i].cpu()
        if not torch.equal(got.view(torch.int32), want.view(torch.int32)):
            util.assert_bits_equal(got.numpy().reshape(wl.H, wl.W, 3), want.numpy().reshape(wl.H, wl.W, 3), "bench batch parity check, frame %d" % wl.mine[i])
    return "frames %s of rank 0 (%s) bit-identical to the CPU oracle, every sample" % (
        [wl.mine[i] for i in idx], "one ipk_raw_to_srgb_batch launch per 64 frames" if wl.batch is not None else "one launch per frame")


def band_leg(ctx, ipa, util, wl, args):
    """ONE frame row-sharded over the ranks through the C entry points: ipk_band_plan, ipk_band_exchange_halo (RCCL ncclSend/ncclRecv in
    place on the slab; host transport when the ranks share a GPU), the band form of the fused kernel."""
    from imagepipe_amd import parallel as par
    H, W = wl.H, wl.W
    comm = par.Comm()
    bands = par.band_plan(H, ctx.world, 2)
    band = bands[ctx.rank]
    slab, own = par.alloc_slab(band, W, wl.srcs[0].dtype, "cuda")
    own.copy_(wl.srcs[0].view(H, W)[band.out_row0: band.out_row0 + band.out_rows])
    plan_b = ipa.FusedPlan(width=W, height=H, is_float=wl.is_float, black0=util.BLACK, white0=util.WHITE, cfa="RGGB", wb_coeffs=util.WB,
                           cam_to_xyz_normalized=wl.cm, out_type=wl.out_type, band=(band.src_row0, band.src_rows, band.out_row0, band.out_rows))
    out_b = plan_b.new_output()

    def step_band():
        comm.exchange_halo(slab, bands, wl.stream)
        plan_b.run(slab.view(-1), out_b, wl.stream)
    eb, _, _ = timed(ctx, step_band, args.steps, args.warmup, 0.0)
    comm.close()
    return {"ms_per_frame": round(eb / args.steps * 1e3, 4), "value": round(args.steps * H * W / 1e6 / eb, 1), "unit": "MP/s",
            "scaling": "strong", "rows_per_rank": band.out_rows, "halo_bytes_per_neighbour": W * (4 if wl.is_float else 2),
            "transport": comm.transport, "gather": "none (bands stay on their GPUs)"}


def band_child(args):
    """Child process of band_children(): ONE frame row-sharded over the N GPUs through the C entry points on the library's RCCL transport
    (ipk_comm_init_rccl with the id the parents exchanged): ipk_band_plan, ipk_band_exchange_halo in place on the slab, the band form of the fused
    kernel writing into its rows of the frame, ipk_band_gather to every rank.  Rank 0 also computes the whole frame in one launch and compares.
    Its own process, so that nothing it does -- a hang, a crash inside RCCL -- can cost the parent its bench line."""
    import ctypes as C
    rank, world, local_rank = int(args.band_child[0]), int(args.band_child[1]), int(args.band_child[2])
    import torch
    torch.cuda.set_device(local_rank)
    import imagepipe_amd as ipa
    from imagepipe_amd import _lib, parallel as par
    import util
    ipa.init(local_rank)
    L = _lib.load()
    h = C.c_void_p()
    _lib.check(L.ipk_comm_init_rccl(bytes.fromhex(args.band_child[3]), rank, world, C.byref(h)), "ipk_comm_init_rccl")
    comm = par.Comm.__new__(par.Comm)
    comm.handle, comm.rank, comm.world, comm.transport, comm.group = h, rank, world, "rccl", None
    comm.selftest()
    W, H = args.width, args.height
    bands = par.band_plan(H, world, 2)
    band = bands[rank]
    src = synth_frame(torch, H, W, args.data, util.SEED + 77).to(torch.float32)          # the same frame on every rank (same seed)
    slab, own = par.alloc_slab(band, W, torch.float32, "cuda")
    own.copy_(src[band.out_row0: band.out_row0 + band.out_rows])
    kw = dict(width=W, height=H, is_float=True, black0=util.BLACK, white0=util.WHITE, cfa="RGGB", wb_coeffs=util.WB, cam_to_xyz_normalized=util.cam_matrix())
    plan_b = ipa.FusedPlan(band=(band.src_row0, band.src_rows, band.out_row0, band.out_rows), **kw)
    frame = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    out_b = frame[band.out_row0: band.out_row0 + band.out_rows].reshape(-1)
    st = torch.cuda.current_stream().cuda_stream

    def compute():
        comm.exchange_halo(slab, bands, st)
        if band.out_rows:
            plan_b.run(slab.view(-1), out_b, st)

    def timeit(fn, n, warm):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    t_compute = timeit(compute, args.steps, 3)
    t_gather = timeit(lambda: (compute(), comm.gather(frame, bands, root=-1, stream=st)), max(2, args.steps // 4), 1)
    same = None
    if rank == 0:
        whole = torch.empty(H * W * 3, dtype=torch.float32, device="cuda")
        ipa.FusedPlan(**kw).run(src.view(-1), whole, st)
        torch.cuda.synchronize()
        same = bool(torch.equal(frame.view(-1).view(torch.int32), whole.view(torch.int32)))
    comm.close()
    if rank == 0:
        print("BAND_CHILD " + json.dumps({
            "frame": [W, H], "ranks": world, "transport": "rccl (ipk_comm: grouped ncclSend/ncclRecv halo rows in place, " +
            ("ncclAllGather in place" if all(b.out_rows == bands[0].out_rows for b in bands) else "grouped ncclSend/ncclRecv gather") + ")",
            "ms_per_frame": round(t_compute, 4), "value": round(H * W / 1e6 / (t_compute * 1e-3), 1), "unit": "MP/s", "scaling": "strong",
            "ms_per_frame_with_all_gather_f32": round(t_gather, 4), "rows_per_rank": band.out_rows, "halo_bytes_per_neighbour": W * 4,
            "gathered_frame_bit_identical_to_one_launch": same}), flush=True)


def band_children(ctx, args, W, H):
    """N > 1: the banded single-frame mode measured in CHILD processes (one per rank, on the rank's GPU), see band_child(); a child that fails or
    does not finish in time costs only this sub-object."""
    import subprocess
    try:
        import imagepipe_amd as ipa
        from imagepipe_amd import _lib
        box = [None]
        if ctx.rank == 0:
            idb = ctypes.create_string_buffer(_lib.COMM_ID_BYTES)
            _lib.check(_lib.load().ipk_comm_unique_id(idb), "ipk_comm_unique_id")
            box = [idb.raw.hex()]
        ctx.dist.broadcast_object_list(box, src=0)
        cmd = [sys.executable, os.path.abspath(__file__), "--band-child", str(ctx.rank), str(ctx.world), str(ctx.local_rank), box[0],
               "--width", str(W), "--height", str(H), "--steps", str(max(4, args.steps // 2)), "--data", args.data]
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
            out, err, rc = r.stdout, r.stderr, r.returncode
        except subprocess.TimeoutExpired as e:
            out, err, rc = (e.stdout or b"").decode(errors="replace") if isinstance(e.stdout, bytes) else (e.stdout or ""), "timeout after 300 s", -9
        res = None
        for line in out.splitlines():
            if line.startswith("BAND_CHILD "):
                res = json.loads(line[len("BAND_CHILD "):])
        if res is None and ctx.rank != 0 and rc == 0:
            res = {"ok": True}                                  # only rank 0's child prints the figures
        elif res is None:
            res = {"ok": False, "error": "rank %d child rc=%s: %s" % (ctx.rank, rc, (err or out)[-400:])}
        else:
            res["ok"] = rc == 0 and (ctx.rank != 0 or bool(res.get("gathered_frame_bit_identical_to_one_launch")))
            if not res["ok"]:
                res["error"] = "child rc=%s, gathered frame identical to one launch: %s" % (rc, res.get("gathered_frame_bit_identical_to_one_launch"))
    except Exception as e:
        res = {"ok": False, "error": repr(e)}
    if not res["ok"]:
        ctx.fail("band_mode", res["error"])
    # every rank's child must have come through, not just rank 0's (all ranks reach this reduction, whatever happened above)
    if ctx.max_over_ranks([0.0 if res["ok"] else 1.0])[0] != 0.0 and res["ok"]:
        res["ok"] = False
        res["error"] = "another rank's child failed"
        ctx.fail("band_mode", res["error"])
    return res


def cpu_baseline(util, data, seconds):
    import numpy as np
    import oracle
    ch, cw = 4000, 6000                                   # a 24 MP sample of the same synthetic workload
    sample = util.noise_u16(util.SEED + 2, ch, cw).astype(np.float32) if data == "noise" else util.smooth_u16(util.SEED + 2, ch, cw).astype(np.float32)
    desc = oracle.make_pipeline(sample, cfa="RGGB", source_kind=1, blacklevels=[util.BLACK] * 4, whitelevels=[util.WHITE] * 4,
                                wb_coeffs=util.WB, cam_to_xyz_normalized=util.cam_matrix())
    oracle.pipeline_run(desc)                              # warm-up (tables, page faults)
    n = 0; t0 = time.perf_counter()
    while True:
        oracle.pipeline_run(desc); n += 1
        dt = time.perf_counter() - t0
        if dt >= seconds or n >= 50:
            break
    return {"value": round(n * ch * cw / 1e6 / dt, 1), "unit": "MP/s", "cores": oracle.max_threads(), "kind": "port",
            "sample": "%d x 24 MP (6000x4000) frames of the same synthetic RGGB f32 workload through the C restatement of the "
                      "reference's unfused per-op pipeline (one OpenMP task per row); the Rust reference cannot be built here" % n}


def main_xtrans(args, ctx, ipa, util, W, H, cfa, maxw):
    """configs[4]: 8640x5760 X-Trans -> 2160x1440 (c5: gofloat + scaled_demosaic in one pass, then the point-wise chain) or full size
    (c5b: the fused kernel in generic-CFA mode).  One frame per GPU per step.  The roofline object is the dominant kernel's:
    c5 times ipk_raw_scaled_demosaic alone for it (its algorithmic bytes: 4 W H in + 16 nW nH out), c5b the single fused launch."""
    torch = ctx.torch
    import ctypes as C
    ints = synth_frame(torch, H, W, args.data, util.SEED + 2 + ctx.rank)
    src = ints.to(torch.float32).reshape(-1).contiguous()
    del ints
    img = ipa.RawImage(width=W, height=H, data=src, cfa=cfa, is_float=True, blacklevels=[util.BLACK] * 4, whitelevels=[util.WHITE] * 4,
                       wb_coeffs=util.WB, cam_to_xyz_normalized=util.cam_matrix())
    pipe = ipa.Pipeline.new_from_source(img)
    pipe.globals.settings.maxwidth = maxw
    out = pipe.run()
    nW, nH = out.width, out.height

    def step():
        pipe.run(out=out.data)
    elapsed, mean_ms, median_ms = timed(ctx, step, args.steps, args.warmup, args.prewarm_ms)
    if maxw:
        dst4 = torch.empty(nW * nH * 4, dtype=torch.float32, device="cuda")
        L = ipa.lib()
        stream = torch.cuda.current_stream().cuda_stream

        def kstep():
            rc = L.ipk_raw_scaled_demosaic(src.data_ptr(), ipa.SRC_F32, W, 0, 0, W, H, C.c_float(util.BLACK), C.c_float(util.WHITE), cfa.encode(), nW, nH,
                                           dst4.data_ptr(), stream)
            assert rc == 0, rc
        _, kernel_ms, kmed = timed(ctx, kstep, args.steps, args.warmup, 0.0)
        kname, alg_bytes = "k_raw_scaled_demosaic_w8m", 4.0 * W * H + 16.0 * nW * nH
    else:
        kernel_ms, kmed, kname, alg_bytes = mean_ms, median_ms, "k_fused_bayer (generic-CFA mode)", 16.0 * W * H
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
    result = {
        "metric": "megapixels/sec full raw->sRGB pipe (input mosaic pixels)",
        "value": round(args.steps * ctx.world * H * W / 1e6 / elapsed, 1), "unit": "MP/s",
        "n_gpus": ctx.world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic (%s, 14-bit X-Trans sensor values, torch Philox seed 0x%X+rank)" % (args.data, util.SEED + 2),
        "config": {"workload": "%dx%d (%.1f MP) synthetic X-Trans 6x6 f32 mosaic -> %dx%d f32 RGB (%s), one frame per GPU per step"
                               % (W, H, H * W / 1e6, nW, nH, "gofloat+scaled_demosaic, then tolab+basecurve+fromlab+gamma in one pass" if maxw else "fused, generic-CFA mode"),
                   "baseline_config": "BASELINE.json configs[4]", "frame": [W, H], "out_frame": [nW, nH], "output_MP_per_s": round(args.steps * ctx.world * nW * nH / 1e6 / elapsed, 1),
                   "used_fused": bool(pipe.last_used_fused), "host_glibc": glibc_version()},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "traffic": None, "kernel": kname, "kernel_ms": round(kernel_ms, 4), "kernel_ms_median": round(kmed, 4), "algorithmic_bytes_per_launch": alg_bytes},
    }
    if ctx.rank == 0:
        print(json.dumps(result), flush=True)
    if ctx.dist is not None:
        ctx.dist.barrier()
        ctx.dist.destroy_process_group()


if __name__ == "__main__":
    main()
