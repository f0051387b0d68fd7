"""torchrun worker for tests/test_gpu_bench.py: one frame sharded by row bands over the ranks (all on GPU 0, gloo), real fused kernel
in band form, halo exchange, all-gather; every rank compares the gathered frame with the oracle."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import torch.distributed as dist

import imagepipe_amd as ipa
from imagepipe_amd import parallel
import oracle
import util


def main():
    cfa, H, W = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    period = 2 if len(cfa) == 4 else 6
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    ipa.init(0)
    raw = util.noise_u16(util.SEED + 95, H, W)
    bands = parallel.band_plan(H, world, period)
    b = bands[rank]
    own = ipa.upload_u16(raw[b.out_row0: b.out_row0 + b.out_rows]).view(b.out_rows, W)
    kw = dict(width=W, height=H, is_float=False, black0=util.BLACK, white0=util.WHITE, cfa=cfa, wb_coeffs=util.WB, cam_to_xyz_normalized=util.cam_matrix())
    out, full = parallel.process_frame_banded(own, H, W, parallel.fused_band_compute(kw), period=period, gather="all")
    torch.cuda.synchronize()
    want = oracle.pipeline_run(oracle.make_pipeline(raw, cfa=cfa, blacklevels=[util.BLACK] * 4, whitelevels=[util.WHITE] * 4, wb_coeffs=util.WB,
                                                    cam_to_xyz_normalized=util.cam_matrix()))
    util.assert_bits_equal(full.cpu().numpy().reshape(H, W, 3), want, "banded frame, rank %d" % rank)
    util.assert_bits_equal(out.cpu().numpy(), want[b.out_row0: b.out_row0 + b.out_rows], "own band, rank %d" % rank)
    dist.barrier()
    if rank == 0:
        print("BANDED_OK world=%d bands=%s" % (world, [(x.out_row0, x.out_rows) for x in bands]))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
