"""Multi-GPU sharding of the raw->sRGB path: one process per GPU, torch.distributed (RCCL over xGMI on the GPU box,
gloo in the CPU tests).

The reference has no distributed code; what shards is the data:
  * a BATCH of frames: frame i -> rank i % world.  No exchange at all (each Pipeline owns its ops and globals,
    src/pipeline.rs:246-249).  This is what bench.py scales with.
  * ONE large frame: contiguous row bands aligned to the CFA period.  demosaic::full taps +-1 row
    (src/ops/demosaic.rs:70-74), so neighbouring bands exchange one mosaic row each way (point-to-point, a few tens
    of KB) before the fused kernel runs in its band form; frame-edge bands get no halo (edge taps are skipped, not
    mirrored, demosaic.rs:103-104).  Gathering the output bands is optional and the expensive part.

Nothing here computes pixels: `compute` is the band kernel (imagepipe_amd.FusedPlan on a GPU; the tests inject the CPU
oracle to check the sharding logic under gloo).
"""
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_frames(n_frames: int, rank: int, world: int) -> List[int]:
    """Batch sharding: the frames rank `rank` owns."""
    return list(range(rank, n_frames, world))


@dataclass
class Band:
    rank: int
    out_row0: int          # first image row this rank produces
    out_rows: int
    src_row0: int          # first image row its slab must hold (band + halos that exist)
    src_rows: int

    @property
    def has_top_halo(self):
        return self.src_row0 < self.out_row0

    @property
    def has_bottom_halo(self):
        return self.src_row0 + self.src_rows > self.out_row0 + self.out_rows


def band_plan(height: int, world: int, period: int = 2) -> List[Band]:
    """Row bands of near-equal size whose boundaries are multiples of the CFA period (2 Bayer, 6 X-Trans), so every
    band starts at the same CFA phase as the frame; each band's slab carries the 1-row halos that lie inside the frame."""
    units = height // period
    bands, r = [], 0
    for k in range(world):
        n_units = units // world + (1 if k < units % world else 0)
        r1 = height if k == world - 1 else r + n_units * period
        rows = r1 - r
        s0, s1 = max(0, r - 1), min(height, r1 + 1)
        bands.append(Band(k, r, rows, s0, s1 - s0))
        r = r1
    return bands


def exchange_halo(band_rows: torch.Tensor, band: Band, bands: Sequence[Band], group=None) -> torch.Tensor:
    """`band_rows`: this rank's own mosaic rows [out_rows, W] (any dtype, on the device the process group handles).
    Sends its first row up and its last row down, receives the neighbours' edge rows, and returns the slab
    [src_rows, W] = (halo above if any) + band + (halo below if any).  Empty bands (more ranks than rows) are skipped."""
    rank, world = band.rank, len(bands)
    up = next((b for b in reversed(bands[:rank]) if b.out_rows > 0), None)
    down = next((b for b in bands[rank + 1:] if b.out_rows > 0), None)
    if band.out_rows == 0:
        return band_rows
    ops, top, bottom = [], None, None
    first, last = band_rows[0].contiguous(), band_rows[-1].contiguous()
    if up is not None:
        top = torch.empty_like(first)
        ops += [dist.P2POp(dist.isend, first, up.rank, group), dist.P2POp(dist.irecv, top, up.rank, group)]
    if down is not None:
        bottom = torch.empty_like(last)
        ops += [dist.P2POp(dist.isend, last, down.rank, group), dist.P2POp(dist.irecv, bottom, down.rank, group)]
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    parts = ([top[None]] if top is not None else []) + [band_rows] + ([bottom[None]] if bottom is not None else [])
    slab = torch.cat(parts, dim=0)
    assert slab.shape[0] == band.src_rows, (slab.shape, band)
    return slab


def alloc_slab(band: Band, width: int, dtype, device) -> Tuple[torch.Tensor, torch.Tensor]:
    """Persistent slab for one band: returns (slab [src_rows, W], own [out_rows, W]) where `own` is the view of the rows this
    rank owns.  Fill `own` once per frame, then exchange_halo_inplace() receives the neighbours' edge rows straight into the
    slab's first/last row -- no per-frame concatenation of a band that is hundreds of MB."""
    slab = torch.empty((band.src_rows, width), dtype=dtype, device=device)
    top = band.out_row0 - band.src_row0
    return slab, slab[top: top + band.out_rows]


def exchange_halo_inplace(slab: torch.Tensor, band: Band, bands: Sequence[Band], group=None) -> None:
    """Halo exchange on a slab from alloc_slab(): sends the band's first/last own row, receives into the halo rows."""
    if band.out_rows == 0:
        return
    rank = band.rank
    up = next((b for b in reversed(bands[:rank]) if b.out_rows > 0), None)
    down = next((b for b in bands[rank + 1:] if b.out_rows > 0), None)
    top = band.out_row0 - band.src_row0
    ops = []
    if up is not None:
        ops += [dist.P2POp(dist.isend, slab[top], up.rank, group), dist.P2POp(dist.irecv, slab[0], up.rank, group)]
    if down is not None:
        ops += [dist.P2POp(dist.isend, slab[top + band.out_rows - 1], down.rank, group),
                dist.P2POp(dist.irecv, slab[band.src_rows - 1], down.rank, group)]
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()


def gather_bands(out_band: torch.Tensor, bands: Sequence[Band], width: int, to: str = "all", group=None) -> Optional[torch.Tensor]:
    """Reassembles the [rows, W, 3] output bands.  to="all": every rank gets the frame (all_gather over xGMI);
    to="root": only rank 0 (gather); bands may differ in height, so they are padded to the tallest."""
    world = len(bands)
    maxrows = max(b.out_rows for b in bands)
    pad = torch.zeros((maxrows, width, 3), dtype=out_band.dtype, device=out_band.device)
    pad[: out_band.shape[0]] = out_band
    rank = dist.get_rank(group)
    if to == "all":
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad, group=group)
    else:
        parts = [torch.empty_like(pad) for _ in range(world)] if rank == 0 else None
        dist.gather(pad, parts, dst=0, group=group)
        if rank != 0:
            return None
    return torch.cat([p[: b.out_rows] for p, b in zip(parts, bands)], dim=0)


def process_frame_banded(own_rows: torch.Tensor, height: int, width: int, compute: Callable[[torch.Tensor, Band], torch.Tensor],
                         period: int = 2, gather: Optional[str] = None, group=None):
    """One frame sharded by rows over the process group.  `own_rows` = this rank's band of the (cropped) mosaic.
    compute(slab, band) -> [band.out_rows, W, 3] runs the band form of the fused kernel on the slab.
    Returns (out_band, full_frame_or_None)."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    bands = band_plan(height, world, period)
    band = bands[rank]
    assert own_rows.shape[0] == band.out_rows and own_rows.shape[1] == width, (own_rows.shape, band)
    slab = exchange_halo(own_rows, band, bands, group)
    out = compute(slab, band) if band.out_rows > 0 else torch.empty((0, width, 3), dtype=torch.float32, device=own_rows.device)
    full = gather_bands(out, bands, width, gather, group) if gather else None
    return out, full


def fused_band_compute(plan_kwargs: dict):
    """compute() for process_frame_banded on a GPU: the fused kernel in band form (ipk_raw_to_srgb with band_* set)."""
    import imagepipe_amd as ipa

    def compute(slab: torch.Tensor, band: Band) -> torch.Tensor:
        kw = dict(plan_kwargs)
        kw["band"] = (band.src_row0, band.src_rows, band.out_row0, band.out_rows)
        plan = ipa.FusedPlan(**kw)
        out = plan.new_output()
        plan.run(slab.reshape(-1).contiguous(), out)
        return out.view(band.out_rows, plan.width, 3)
    return compute
