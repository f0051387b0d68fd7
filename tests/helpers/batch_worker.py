"""torchrun worker for tests/test_gpu_bench.py: BASELINE.json configs[3] in small -- a batch of frames sharded over the ranks
(frame i -> rank i % N, no data-path collective), every rank runs the fused kernel on its frames and checks them against the oracle;
a final all_gather of the processed frame ids shows that the batch was covered exactly once."""
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
    n_frames, H, W = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    ipa.init(0)
    mine = parallel.shard_frames(n_frames, rank, world)
    plan = ipa.FusedPlan(width=W, height=H, is_float=False, black0=util.BLACK, white0=util.WHITE, cfa="RGGB", wb_coeffs=util.WB,
                         cam_to_xyz_normalized=util.cam_matrix())
    for i in mine:
        raw = util.noise_u16(util.SEED + 200 + i, H, W)
        out = plan.new_output()
        plan.run(ipa.upload_u16(raw), out)
        torch.cuda.synchronize()
        want = oracle.pipeline_run(oracle.make_pipeline(raw, cfa="RGGB", blacklevels=[util.BLACK] * 4, whitelevels=[util.WHITE] * 4, wb_coeffs=util.WB,
                                                        cam_to_xyz_normalized=util.cam_matrix()))
        util.assert_bits_equal(out.cpu().numpy().reshape(H, W, 3), want, "frame %d on rank %d" % (i, rank))
    ids = torch.full((n_frames,), -1, dtype=torch.int64)
    ids[: len(mine)] = torch.tensor(mine, dtype=torch.int64)
    gathered = [torch.empty_like(ids) for _ in range(world)]
    dist.all_gather(gathered, ids)
    seen = sorted(int(v) for t in gathered for v in t if v >= 0)
    assert seen == list(range(n_frames)), seen
    dist.barrier()
    if rank == 0:
        print("BATCH_OK frames=%d world=%d" % (n_frames, world))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
