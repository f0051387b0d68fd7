#!/usr/bin/env python
"""bench.py -- megapixels/s of the full raw->sRGB pipe on a 100 MP f32 Bayer frame (BASELINE.json's metric).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

One step = one pass of the hot path (Pipeline::run: gofloat + demosaic + tolab + basecurve + fromlab + gamma, fused)
over one 10000x10000 synthetic RGGB f32 frame per GPU, input and output resident in HBM.  Frames are independent, so
N GPUs each process their own frame with no data-path collective (weak scaling); value = all frames' pixels / max-rank time.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling
BYTES_PER_PX = 16.0            # algorithmic: 4 B mosaic sample in + 3 x 4 B RGB out


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--width", type=int, default=10000)
    ap.add_argument("--height", type=int, default=10000)
    ap.add_argument("--data", choices=["noise", "smooth", "photo", "flat", "white"], default="noise",
                    help="noise: uniform 14-bit values (worst case, the headline); smooth: diagonal gradient over the full range; "
                         "photo: low-frequency mid-tone scene with shot-like noise and ~2 %% blown highlights; flat / white: development extremes (nothing / everything saturated)")
    ap.add_argument("--src", choices=["f32", "u16"], default="f32")
    ap.add_argument("--out", choices=["f32", "u8", "u16"], default="f32")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target wall time of the CPU baseline leg")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--prewarm-ms", type=float, default=250.0,
                    help="untimed launches before the W warmup steps, to bring the shader clock to its loaded steady state (0 = off)")
    ap.add_argument("--band", action="store_true",
                    help="additionally time ONE frame row-sharded over the N GPUs with the RCCL halo exchange (reported as an extra "
                         "'band_mode' object; the headline value stays the batch-sharded one)")
    return ap.parse_args()


def synth_frame(torch, h, w, kind, seed):
    """14-bit sensor values (black 512, white 16383) as f32, generated on the device."""
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


def main():
    args = parse()
    import torch
    import imagepipe_amd as ipa
    import util

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dist = None
    if world > 1:
        import torch.distributed as dist
        # IPK_BENCH_SHARE_GPU=1 (development only): all ranks on one GPU over gloo, to exercise the N > 1 control flow on a 1-GPU box
        share = os.environ.get("IPK_BENCH_SHARE_GPU") == "1"
        if share:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    red_dev = "cpu" if (dist is not None and dist.get_backend() == "gloo") else "cuda"
    ipa.init(local_rank)

    H, W = args.height, args.width
    ints = synth_frame(torch, H, W, args.data, util.SEED + 2 + rank)
    is_float = args.src == "f32"
    src = ints.to(torch.float32).reshape(-1).contiguous() if is_float else ints.to(torch.int16).reshape(-1).contiguous()
    del ints
    out_type = {"f32": ipa.OUT_F32, "u8": ipa.OUT_U8, "u16": ipa.OUT_U16}[args.out]
    out_dt = {"f32": torch.float32, "u8": torch.uint8, "u16": torch.int16}[args.out]
    dst = torch.empty(H * W * 3, dtype=out_dt, device="cuda")
    cm = util.cam_matrix()
    plan = ipa.FusedPlan(width=W, height=H, is_float=is_float, black0=util.BLACK, white0=util.WHITE, cfa="RGGB", wb_coeffs=util.WB,
                         cam_to_xyz_normalized=cm, out_type=out_type)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        plan.run(src, dst, stream)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- correctness check against the CPU oracle (outside the timed region): the whole frame, every output sample ----
    checked = None
    if not args.no_check and rank == 0 and args.out == "f32":
        import numpy as np
        import oracle
        step(); torch.cuda.synchronize()

        def compare(rows):
            part = src[: min(H, rows + 2) * W].cpu().numpy().reshape(-1, W)
            desc = oracle.make_pipeline(part if is_float else part.view(np.uint16), cfa="RGGB", source_kind=1 if is_float else 0,
                                        blacklevels=[util.BLACK] * 4, whitelevels=[util.WHITE] * 4, wb_coeffs=util.WB, cam_to_xyz_normalized=cm)
            want = torch.from_numpy(oracle.pipeline_run(desc)[:rows].reshape(-1))
            got = dst[: rows * W * 3].cpu()
            if not torch.equal(got.view(torch.int32), want.view(torch.int32)):
                util.assert_bits_equal(got.numpy().reshape(rows, W, 3), want.numpy().reshape(rows, W, 3), "bench parity check")
        try:
            compare(H)
            checked = "all %d rows (%d output samples) bit-identical to the CPU oracle" % (H, H * W * 3)
        except MemoryError:
            compare(12)
            checked = "first 12 rows bit-identical to the CPU oracle (no host memory for the whole frame)"

    # every rank starts its clock pre-warm together (rank 0 has just spent seconds in the parity check; a rank that warmed up
    # early would sit idle at the barrier below and cool down again)
    barrier()
    # The MI355X raises its shader clock over the first tens of milliseconds of sustained load (measured: the same kernel
    # takes 0.78 ms in the first 20 launches after idle and 0.68 ms from ~50 launches on).  These launches are untimed.
    t_pw = time.perf_counter()
    while (time.perf_counter() - t_pw) * 1e3 < args.prewarm_ms:
        for _ in range(16):
            step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    barrier()
    ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps          # HIP events on the launch stream: avg kernel (+launch gap) per step
    if dist is not None:
        t = torch.tensor([elapsed, kernel_ms], device=red_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms = float(t[0]), float(t[1])

    mp_total = world * args.steps * (H * W) / 1e6
    value = mp_total / elapsed
    in_b = 4.0 if is_float else 2.0
    out_b = {"f32": 12.0, "u8": 3.0, "u16": 6.0}[args.out]
    alg_bytes = (in_b + out_b) * H * W
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9

    result = {
        "metric": "megapixels/sec full raw->sRGB pipe, 100 MP f32 frame",
        "value": round(value, 1), "unit": "MP/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic (%s, 14-bit RGGB sensor values, torch Philox seed 0x%X+rank)" % (args.data, util.SEED + 2),
        "config": {"workload": "%dx%d (%.0f MP) synthetic RGGB Bayer %s mosaic -> fused gofloat+demosaic+tolab+basecurve+fromlab+gamma -> %s RGB, one frame per GPU per step"
                               % (W, H, H * W / 1e6, args.src, args.out),
                   "frame": [W, H], "src": args.src, "out": args.out, "frames_per_step": world, "prewarm_ms": args.prewarm_ms, "sharding": "one independent frame per GPU, no collective"},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                     "kernel": "k_fused_bayer", "kernel_ms": round(kernel_ms, 4), "algorithmic_bytes_per_launch": alg_bytes},
    }
    if checked:
        result["parity_check"] = checked
    # HBM traffic per launch: not measurable from inside this process (PMC counters need rocprofv3); taken from the
    # committed rocprofv3 passes of this same command (tools/profile.sh -> profiles/<round>_counters.json: separate
    # --pmc FETCH_SIZE / WRITE_SIZE runs, FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 note) when the workload matches.
    if rank == 0 and (W, H, args.src, args.out, args.data) == (10000, 10000, "f32", "f32", "noise"):
        import glob
        profs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_counters.json")))
        if profs:
            try:
                pj = json.load(open(profs[-1]))
                if "hbm_traffic_bytes_per_launch" in pj:
                    result["roofline"]["traffic"] = round(pj["hbm_traffic_bytes_per_launch"])
                    result["roofline"]["traffic_source"] = os.path.relpath(profs[-1], ROOT)
            except Exception:
                pass

    # ---- optional: one frame sharded by row bands over the ranks, 1-row halo exchange (RCCL P2P) before every launch ----
    if args.band:
        from imagepipe_amd import parallel as par
        bands = par.band_plan(H, world, 2)
        band = bands[rank]
        slab, own = par.alloc_slab(band, W, src.dtype, "cuda")
        own.copy_(src.view(H, W)[band.out_row0: band.out_row0 + band.out_rows])
        plan_b = ipa.FusedPlan(width=W, height=H, is_float=is_float, black0=util.BLACK, white0=util.WHITE, cfa="RGGB", wb_coeffs=util.WB,
                               cam_to_xyz_normalized=cm, out_type=out_type, band=(band.src_row0, band.src_rows, band.out_row0, band.out_rows))
        out_b = plan_b.new_output()

        def step_band():
            if world > 1:
                par.exchange_halo_inplace(slab, band, bands)
            plan_b.run(slab.view(-1), out_b, stream)

        for _ in range(args.warmup):
            step_band()
        barrier()
        tb = time.perf_counter()
        for _ in range(args.steps):
            step_band()
        barrier()
        eb = time.perf_counter() - tb
        if dist is not None:
            t = torch.tensor([eb], device=red_dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            eb = float(t[0])
        result["band_mode"] = {"ms_per_frame": round(eb / args.steps * 1e3, 4), "value": round(args.steps * H * W / 1e6 / eb, 1), "unit": "MP/s",
                               "scaling": "strong", "rows_per_rank": band.out_rows, "halo_bytes_per_neighbour": W * (4 if is_float else 2),
                               "gather": "none (bands stay on their GPUs)"}

    # ---- CPU baseline: the oracle's reference-shaped pipeline (unfused, one task per row) on this host ----
    if rank == 0 and not args.no_cpu_baseline:
        import numpy as np
        import oracle
        ch, cw = 4000, 6000                                   # a 24 MP sample of the same synthetic workload
        sample = util.noise_u16(util.SEED + 2, ch, cw).astype(np.float32) if args.data == "noise" else util.smooth_u16(util.SEED + 2, ch, cw).astype(np.float32)
        desc = oracle.make_pipeline(sample, cfa="RGGB", source_kind=1, blacklevels=[util.BLACK] * 4, whitelevels=[util.WHITE] * 4,
                                    wb_coeffs=util.WB, cam_to_xyz_normalized=cm)
        oracle.pipeline_run(desc)                              # warm-up (tables, page faults)
        n = 0; t0 = time.perf_counter()
        while True:
            oracle.pipeline_run(desc); n += 1
            dt = time.perf_counter() - t0
            if dt >= args.cpu_seconds or n >= 50:
                break
        result["cpu_baseline"] = {"value": round(n * ch * cw / 1e6 / dt, 1), "unit": "MP/s", "cores": oracle.max_threads(), "kind": "port",
                                  "sample": "%d x 24 MP (6000x4000) frames of the same synthetic RGGB f32 workload through the C restatement of the "
                                            "reference's unfused per-op pipeline (one OpenMP task per row); the Rust reference cannot be built here" % n}
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()                      # the other ranks wait here while rank 0 runs the CPU baseline leg
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
