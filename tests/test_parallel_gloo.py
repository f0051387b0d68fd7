"""The N>1 path on CPU: world_size-2 (and 3) gloo process groups exercise the sharding logic of imagepipe_amd.parallel --
band plan, point-to-point halo exchange, output gather, batch sharding.  The band kernel is injected (`compute`); here it
is the CPU oracle run on the slab, so a wrong halo, phase or row bookkeeping shows up as a pixel mismatch against the
oracle's whole-frame result.  (On the GPU box the same code runs with backend "nccl" = RCCL and the fused HIP kernel.)"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, case, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
      try:
          import oracle as orc
          import util
          from imagepipe_amd import parallel as par
          # (the oracle, like the reference, refuses frames under 10x10 -- gofloat.rs:74-82 -- so slabs here are >= 10 rows)
          orc.set_num_threads(1)
          cfa, h, w, period = case
          raw = util.noise_u16(util.SEED + 50, h, w)
          kw = dict(cfa=cfa, blacklevels=[util.BLACK] * 4, whitelevels=[util.WHITE] * 4, wb_coeffs=util.WB, cam_to_xyz_normalized=util.cam_matrix())
          whole = orc.pipeline_run(orc.make_pipeline(raw, **kw))
          bands = par.band_plan(h, world, period)
          b = bands[rank]
          assert sum(x.out_rows for x in bands) == h and all(x.out_row0 % period == 0 for x in bands)

          def compute(slab, band, out):
              # the oracle on the slab; slab row 0 is image row band.src_row0 (one row before a multiple of the CFA period
              # when there is a top halo), so the window is padded to the frame's CFA phase
              s = slab.numpy().view(np.uint16)
              pad_top = band.src_row0 % period                   # dummy rows in front so that window row index == image row (mod period)
              # ... then cut them (and the halo rows) off the result
              win = np.concatenate([np.zeros((pad_top, s.shape[1]), np.uint16), s]) if pad_top else s
              res = orc.pipeline_run(orc.make_pipeline(win, **kw))
              off = pad_top + (band.out_row0 - band.src_row0)
              # rows adjacent to a dummy row / the slab edge are only valid when they are true frame edges or had a halo
              out.copy_(torch.from_numpy(res[off: off + band.out_rows].copy()))

          # the C entry points through the host transport (ipk_comm_init_host; gloo moves the bytes): ipk_band_plan,
          # ipk_host_band_exchange_halo on the host slab, ipk_host_band_gather in place on the host frame
          comm = par.Comm("host")
          own = torch.from_numpy(raw[b.out_row0: b.out_row0 + b.out_rows].view(np.int16).copy())
          out, full = par.process_frame_banded(comm, own, h, w, compute, period=period, gather="all")
          ok_band = np.array_equal(out.numpy().view(np.uint32), whole[b.out_row0: b.out_row0 + b.out_rows].view(np.uint32))
          ok_full = np.array_equal(full.numpy().view(np.uint32), whole.view(np.uint32))
          out2, root = par.process_frame_banded(comm, own, h, w, compute, period=period, gather="root")
          ok_root = (root is None) if rank else np.array_equal(root.numpy().view(np.uint32), whole.view(np.uint32))
          # the slab after the exchange is exactly the frame's rows [src_row0, src_row0 + src_rows)
          slab, own_view = par.alloc_slab(b, w, torch.int16, "cpu")
          own_view.copy_(own)
          comm.exchange_halo(slab, bands)
          ok_band = ok_band and np.array_equal(slab.numpy().view(np.uint16), raw[b.src_row0: b.src_row0 + b.src_rows])
          comm.close()
          frames = par.shard_frames(7, rank, world)
          q.put((rank, ok_band, ok_full, ok_root, frames, (b.src_row0, b.src_rows, b.out_row0, b.out_rows)))
      except Exception as e:                                   # never leave the parent waiting on the queue
        q.put((rank, False, False, False, repr(e), ()))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,case", [(2, ("RGGB", 40, 64, 2)), (2, ("GBRG", 37, 50, 2)), (3, ("RGGB", 50, 36, 2)),
                                        (2, ("GGRGGBGGBGGRBRGRBGGGBGGRGGRGGBRBGBRG", 48, 42, 6)), (3, ("BGGR", 36, 20, 2))])
def test_banded_frame_equals_whole_frame(world, case):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=300) for _ in range(world))
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    for rank, ok_band, ok_full, ok_root, frames, band in res:
        assert ok_band, ("band mismatch", rank, band)
        assert ok_full and ok_root, ("gather mismatch", rank)
        assert frames == list(range(rank, 7, world))


def test_band_plan_properties():
    from imagepipe_amd.parallel import band_plan
    for h in (10, 11, 97, 4000, 10000):
        for world in (1, 2, 3, 4, 8):
            for period in (2, 6):
                bands = band_plan(h, world, period)
                assert sum(b.out_rows for b in bands) == h
                r = 0
                for b in bands:
                    assert b.out_row0 == r and b.out_row0 % period == 0
                    if b.out_rows:
                        assert b.src_row0 == max(0, r - 1) and b.src_row0 + b.src_rows == min(h, r + b.out_rows + 1)
                    else:
                        assert b.src_rows == 0                  # more ranks than CFA periods: an empty band holds nothing
                    r += b.out_rows
                sizes = [b.out_rows for b in bands[:-1]]
                assert not sizes or max(sizes) - min(sizes) <= period


def test_band_plan_scaled_covers_every_window(orc):
    """ipk_band_plan_scaled: output rows partitioned exactly, and each band's source range contains the window rows
    [floor(skip*r), floor(skip*(r+1))] (scaling.rs:86-87, f32 arithmetic restated here) of every output row it owns"""
    from imagepipe_amd.parallel import band_plan_scaled
    for h, nh in ((5760, 1440), (4000, 1000), (4000, 999), (101, 50), (37, 2)):
        skip = np.float32(np.float32(h - 1) / np.float32(nh - 1))
        for world in (1, 2, 3, 8, 64):
            bands = band_plan_scaled(h, nh, world)
            assert sum(b.out_rows for b in bands) == nh
            r = 0
            for b in bands:
                assert b.out_row0 == r
                for row in range(b.out_row0, b.out_row0 + b.out_rows):
                    lo = min(h - 1, int(np.floor(np.float32(skip * np.float32(row)))))
                    hi = min(h - 1, int(np.floor(np.float32(skip * np.float32(row + 1)))))
                    assert b.src_row0 <= lo and hi < b.src_row0 + b.src_rows, (h, nh, world, row)
                r += b.out_rows
