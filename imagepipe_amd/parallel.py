"""Multi-GPU sharding of the raw->sRGB path, as a thin harness over the C ABI (include/imagepipe_amd.h, "Multi-GPU" section;
imagepipe_amd/csrc/ipk_comm.cpp).  One process per GPU.

The reference has no distributed code; what shards is the data:
  * a BATCH of frames: frame i -> rank i % world.  No exchange at all (each Pipeline owns its ops and globals,
    src/pipeline.rs:246-249).  This is what bench.py scales with.
  * ONE large frame, full resolution: contiguous row bands aligned to the CFA period (ipk_band_plan).  demosaic::full taps +-1 row
    (src/ops/demosaic.rs:70-74), so neighbouring bands exchange one mosaic row each way (ipk_band_exchange_halo: grouped
    ncclSend/ncclRecv in place on the slab) before the fused kernel runs in its band form; frame-edge bands get no halo (edge taps
    are skipped, not mirrored, demosaic.rs:103-104).
  * ONE large frame through OpDemosaic's scaled branch: output-row bands (ipk_band_plan_scaled), each rank reads the source rows its
    windows need (src/scaling.rs:84-94); nothing is exchanged, ipk_raw_scaled_demosaic_band computes the band.
  Gathering the output (ipk_band_gather, in place) is optional and the expensive part.

Everything that moves or computes lives in libimagepipe_amd.so; this module only (a) forms the communicator -- RCCL, with the
unique id handed round by torch.distributed, or the host transport with torch.distributed (gloo) as the caller-side messaging, which
is how ranks that SHARE one GPU are tested (RCCL refuses two ranks per device) and how the CPU tests run -- and (b) offers the
per-frame driver the tests use.  `compute` is the band kernel (imagepipe_amd.FusedPlan on a GPU; the CPU tests inject the oracle).
"""
import ctypes as C
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist

from . import _lib


def shard_frames(n_frames: int, rank: int, world: int) -> List[int]:
    """Batch sharding: the frames rank `rank` owns."""
    return list(range(rank, n_frames, world))


@dataclass
class Band:
    rank: int
    out_row0: int          # first image row this rank produces
    out_rows: int
    src_row0: int          # first image row its slab must hold (band + halos that exist / the rows its windows read)
    src_rows: int

    @property
    def has_top_halo(self):
        return self.src_row0 < self.out_row0

    @property
    def has_bottom_halo(self):
        return self.src_row0 + self.src_rows > self.out_row0 + self.out_rows


def _to_c(bands: Sequence[Band]):
    arr = (_lib.Band * len(bands))()
    for i, b in enumerate(bands):
        arr[i].out_row0, arr[i].out_rows, arr[i].src_row0, arr[i].src_rows = b.out_row0, b.out_rows, b.src_row0, b.src_rows
    return arr


def band_plan(height: int, world: int, period: int = 2) -> List[Band]:
    """ipk_band_plan: row bands of near-equal size whose boundaries are multiples of the CFA period (2 Bayer, 6 X-Trans), so every
    band starts at the same CFA phase as the frame; each band's slab carries the 1-row halos that lie inside the frame."""
    arr = (_lib.Band * world)()
    _lib.check(_lib.load().ipk_band_plan(height, world, period, arr), "ipk_band_plan")
    return [Band(k, arr[k].out_row0, arr[k].out_rows, arr[k].src_row0, arr[k].src_rows) for k in range(world)]


def band_plan_scaled(height: int, nheight: int, world: int) -> List[Band]:
    """ipk_band_plan_scaled: output-row bands of scaled_demosaic and the source rows each one reads (scaling.rs:84-94)"""
    arr = (_lib.Band * world)()
    _lib.check(_lib.load().ipk_band_plan_scaled(height, nheight, world, arr), "ipk_band_plan_scaled")
    return [Band(k, arr[k].out_row0, arr[k].out_rows, arr[k].src_row0, arr[k].src_rows) for k in range(world)]


class Comm:
    """ipk_comm over the default torch.distributed group.  transport="rccl": ncclCommInitRank with the id broadcast from rank 0
    (one GPU per rank); transport="host": the C side calls back into torch.distributed point-to-point on CPU byte tensors (any
    backend that moves CPU tensors, gloo in the tests) -- ranks may share a GPU, or have none."""

    def __init__(self, transport: Optional[str] = None, group=None):
        L = _lib.load()
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        if transport is None:
            transport = "rccl" if dist.get_backend(group) == "nccl" else "host"
        self.transport = transport
        h = C.c_void_p()
        if transport == "rccl":
            idb = C.create_string_buffer(_lib.COMM_ID_BYTES)
            if self.rank == 0:
                _lib.check(L.ipk_comm_unique_id(idb), "ipk_comm_unique_id")
            box = [idb.raw]
            dist.broadcast_object_list(box, src=0, group=group)
            _lib.check(L.ipk_comm_init_rccl(box[0], self.rank, self.world, C.byref(h)), "ipk_comm_init_rccl")
        else:
            def exchange(_ctx, send_peer, send, send_bytes, recv_peer, recv, recv_bytes):
                try:
                    reqs = []
                    if recv_peer >= 0 and recv_bytes:
                        rt = torch.from_numpy(np.ctypeslib.as_array(C.cast(recv, C.POINTER(C.c_uint8)), shape=(recv_bytes,)))
                        reqs.append(dist.irecv(rt, src=recv_peer, group=group))
                    if send_peer >= 0 and send_bytes:
                        st = torch.from_numpy(np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_uint8)), shape=(send_bytes,)))
                        reqs.append(dist.isend(st, dst=send_peer, group=group))
                    for r in reqs:
                        r.wait()
                    return 0
                except Exception:          # an exception must not unwind through the C frame
                    import traceback
                    traceback.print_exc()
                    return 1
            self._cb = _lib.EXCHANGE_FN(exchange)                      # keep the trampoline alive as long as the communicator
            _lib.check(L.ipk_comm_init_host(self.rank, self.world, self._cb, None, C.byref(h)), "ipk_comm_init_host")
        self.handle = h

    def close(self):
        if self.handle:
            _lib.load().ipk_comm_free(self.handle)
            self.handle = None

    def selftest(self):
        _lib.check(_lib.load().ipk_comm_selftest(self.handle), "ipk_comm_selftest")

    # -- halo exchange, in place on a slab from alloc_slab() ------------------------------------------------------------------
    def exchange_halo(self, slab: torch.Tensor, bands: Sequence[Band], stream=None) -> None:
        """the band's first / last own row goes to the neighbour above / below, their edge rows arrive in the slab's halo rows"""
        assert slab.is_contiguous() and slab.dim() == 2
        row_bytes = slab.shape[1] * slab.element_size()
        L = _lib.load()
        if slab.is_cuda:
            st = stream if stream is not None else torch.cuda.current_stream().cuda_stream
            _lib.check(L.ipk_band_exchange_halo(self.handle, slab.data_ptr(), row_bytes, _to_c(bands), st), "ipk_band_exchange_halo")
        else:
            _lib.check(L.ipk_host_band_exchange_halo(self.handle, slab.data_ptr(), row_bytes, _to_c(bands)), "ipk_host_band_exchange_halo")

    # -- gather, in place: every rank's band already sits at its rows of `frame` -----------------------------------------------
    def gather(self, frame: torch.Tensor, bands: Sequence[Band], root: int = -1, stream=None, overlap: bool = False) -> None:
        """frame: [out_height, row elements...] contiguous; root = -1: all ranks receive all bands"""
        assert frame.is_contiguous()
        row_bytes = frame[0].numel() * frame.element_size()
        L = _lib.load()
        if not frame.is_cuda:
            _lib.check(L.ipk_host_band_gather(self.handle, frame.data_ptr(), row_bytes, _to_c(bands), root), "ipk_host_band_gather")
            return
        st = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        if overlap:
            _lib.check(L.ipk_band_gather_begin(self.handle, frame.data_ptr(), row_bytes, _to_c(bands), root, st), "ipk_band_gather_begin")
        else:
            _lib.check(L.ipk_band_gather(self.handle, frame.data_ptr(), row_bytes, _to_c(bands), root, st), "ipk_band_gather")

    def wait(self, stream=None):
        st = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        _lib.check(_lib.load().ipk_comm_wait(self.handle, st), "ipk_comm_wait")


def alloc_slab(band: Band, width: int, dtype, device):
    """Persistent slab for one band: returns (slab [src_rows, W], own [out_rows, W]) where `own` is the view of the rows this
    rank owns.  Fill `own` once per frame, then Comm.exchange_halo() receives the neighbours' edge rows straight into the
    slab's first/last row -- no per-frame concatenation of a band that is hundreds of MB."""
    slab = torch.empty((band.src_rows, width), dtype=dtype, device=device)
    top = band.out_row0 - band.src_row0
    return slab, slab[top: top + band.out_rows]


def process_frame_banded(comm: Comm, own_rows: torch.Tensor, height: int, width: int, compute: Callable[[torch.Tensor, Band, torch.Tensor], None],
                         period: int = 2, gather: Optional[str] = None, out_dtype=torch.float32, out_channels: int = 3):
    """One full-resolution frame sharded by rows over the communicator.  `own_rows` = this rank's band of the (cropped) mosaic.
    compute(slab, band, out) writes the band's [out_rows, W, channels] result into `out`, which is the band's view of the full
    frame buffer when a gather follows (so the gather is in place) and a private buffer otherwise.
    Returns (out_band, full_frame_or_None); gather = None | "all" | "root"."""
    bands = band_plan(height, comm.world, period)
    band = bands[comm.rank]
    assert own_rows.shape[0] == band.out_rows and own_rows.shape[1] == width, (own_rows.shape, band)
    slab, own = alloc_slab(band, width, own_rows.dtype, own_rows.device)
    own.copy_(own_rows)
    comm.exchange_halo(slab, bands)
    full = None
    if gather:
        full = torch.empty((height, width, out_channels), dtype=out_dtype, device=own_rows.device)
        out = full[band.out_row0: band.out_row0 + band.out_rows]
    else:
        out = torch.empty((band.out_rows, width, out_channels), dtype=out_dtype, device=own_rows.device)
    if band.out_rows:
        compute(slab, band, out)
    if gather:
        comm.gather(full, bands, root=-1 if gather == "all" else 0)
        if gather == "root" and comm.rank != 0:
            full = None
    return out, full


def fused_band_compute(plan_kwargs: dict):
    """compute() for process_frame_banded on a GPU: the fused kernel in band form (ipk_raw_to_srgb with band_* set), writing
    straight into the band's rows of the destination frame."""
    import imagepipe_amd as ipa
    plans = {}

    def compute(slab: torch.Tensor, band: Band, out: torch.Tensor) -> None:
        key = (band.src_row0, band.src_rows, band.out_row0, band.out_rows)
        if key not in plans:                                           # one prepared descriptor per band shape, reused frame after frame
            plans[key] = ipa.FusedPlan(**dict(plan_kwargs, band=key))
        plans[key].run(slab.reshape(-1), out.reshape(-1))
    return compute
