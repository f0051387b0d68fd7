"""torchrun worker for tests/test_gpu_bench.py: ONE frame sharded by row bands over the ranks, driven through the C entry points
(ipk_band_plan / ipk_band_exchange_halo / ipk_band_gather / ipk_band_gather_begin / ipk_raw_scaled_demosaic_band).  The ranks share
GPU 0, so the communicator uses the host transport (RCCL refuses two ranks on one device) with gloo moving the bytes; kernels, slabs
and frames are real device memory.  Modes:
  full   : full-resolution path -- halo exchange in place on the slab, band form of the fused kernel writing straight into its rows
           of the frame, in-place all-gather (f32), then the same with 8-bit output gathered on the communicator's own stream while
           the next launch is already queued (gather_begin / wait); every rank compares both frames with the oracle
  scaled : OpDemosaic's scaled branch -- output-row bands, each rank uploads only the source rows its windows read, the band
           kernel, gather to root; rank 0 compares the RGBE frame with the oracle's scaled_demosaic
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import torch.distributed as dist

import imagepipe_amd as ipa
from imagepipe_amd import parallel, _lib
import oracle
import util


def main():
    cfa, H, W = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    mode = sys.argv[4] if len(sys.argv) > 4 else "full"
    period = 2 if len(cfa) == 4 else 6
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    ipa.init(0)
    comm = parallel.Comm("host")
    comm.selftest()                                                     # ring + ragged gathers + halo exchange on device buffers
    raw = util.noise_u16(util.SEED + 95, H, W)
    okw = dict(cfa=cfa, blacklevels=[util.BLACK] * 4, whitelevels=[util.WHITE] * 4, wb_coeffs=util.WB, cam_to_xyz_normalized=util.cam_matrix())
    if mode == "full":
        bands = parallel.band_plan(H, world, period)
        b = bands[rank]
        own = ipa.upload_u16(raw[b.out_row0: b.out_row0 + b.out_rows]).view(b.out_rows, W)
        kw = dict(width=W, height=H, is_float=False, black0=util.BLACK, white0=util.WHITE, cfa=cfa, wb_coeffs=util.WB, cam_to_xyz_normalized=util.cam_matrix())
        out, full = parallel.process_frame_banded(comm, own, H, W, parallel.fused_band_compute(kw), period=period, gather="all")
        torch.cuda.synchronize()
        want = oracle.pipeline_run(oracle.make_pipeline(raw, **okw))
        util.assert_bits_equal(full.cpu().numpy().reshape(H, W, 3), want, "banded frame, rank %d" % rank)
        util.assert_bits_equal(out.cpu().numpy(), want[b.out_row0: b.out_row0 + b.out_rows], "own band, rank %d" % rank)
        # 8-bit output (4x fewer bytes to gather), gathered on the communicator's stream behind the band kernel
        slab, ownv = parallel.alloc_slab(b, W, own.dtype, "cuda")
        ownv.copy_(own)
        comm.exchange_halo(slab, bands)
        frame8 = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
        if b.out_rows:
            plan8 = ipa.FusedPlan(**dict(kw, out_type=ipa.OUT_U8, band=(b.src_row0, b.src_rows, b.out_row0, b.out_rows)))
            plan8.run(slab.reshape(-1), frame8[b.out_row0: b.out_row0 + b.out_rows].reshape(-1))
        comm.gather(frame8, bands, overlap=True)                         # ipk_band_gather_begin
        comm.wait()                                                      # ipk_comm_wait: this stream continues after the gather
        torch.cuda.synchronize()
        want8 = oracle.pipeline_output_8bit(oracle.make_pipeline(raw, **okw))
        assert np.array_equal(frame8.cpu().numpy(), want8), "8-bit banded frame, rank %d" % rank
    else:
        nW, nH = int(sys.argv[5]), int(sys.argv[6])
        bands = parallel.band_plan_scaled(H, nH, world)
        b = bands[rank]
        L = _lib.load()
        slab = ipa.upload_u16(raw[b.src_row0: b.src_row0 + b.src_rows])   # only the rows this band's windows read
        frame = torch.zeros((nH, nW, 4), dtype=torch.float32, device="cuda")
        cb = _lib.Band(b.out_row0, b.out_rows, b.src_row0, b.src_rows)
        if b.out_rows:
            _lib.check(L.ipk_raw_scaled_demosaic_band(slab.data_ptr(), ipa.SRC_U16, W, 0, W, H, C.c_float(util.BLACK), C.c_float(util.WHITE), cfa.encode(),
                                                      nW, nH, C.byref(cb), frame[b.out_row0: b.out_row0 + b.out_rows].data_ptr(),
                                                      torch.cuda.current_stream().cuda_stream), "ipk_raw_scaled_demosaic_band")
        comm.gather(frame, bands, root=0)
        torch.cuda.synchronize()
        if rank == 0:
            branch, want = oracle.demosaic_run(cfa, oracle.gofloat_cfa(raw, 0, 0, W, H, util.BLACK, util.WHITE), nW, nH)
            assert branch == 2, branch
            util.assert_bits_equal(frame.cpu().numpy(), want, "scaled demosaic gathered from %d bands" % world)
    dist.barrier()
    comm.close()
    if rank == 0:
        print("BANDED_OK mode=%s world=%d bands=%s" % (mode, world, [(x.out_row0, x.out_rows, x.src_row0, x.src_rows) for x in bands]))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
