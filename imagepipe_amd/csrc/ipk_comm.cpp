// ipk_comm.cpp -- multi-GPU entry points of the C ABI (include/imagepipe_amd.h, "Multi-GPU" section): band plans, the halo
// exchange and the in-place gather of ONE frame sharded by row bands across the GPUs of a node.
//
// MI355X-first: the 8 GPUs of a node are a fully connected xGMI mesh of point-to-point links (no switch), so
//   * the halo is one grouped ncclSend/ncclRecv pair per neighbour (one mosaic row each way, tens of KB: latency only), posted
//     in place on the slab the band kernel reads -- no staging copy, no concatenation of a band that is hundreds of MB;
//   * the gather sends every band to every peer at once (7 peers = 7 distinct links), each straight into its rows of the
//     destination frame; ncclAllGather is used when the bands are equal (RCCL then picks its own mesh algorithm), a group of
//     ncclSend/ncclRecv when they are ragged;
//   * the gather of frame k can run on the communicator's own stream while the kernel of frame k+1 runs (gather_begin / wait).
// RCCL is loaded at run time (dlopen) so that the library still loads on a host without it, and so that a process that already
// holds an RCCL (PyTorch ships one) shares that copy instead of mapping a second one.
// The host transport moves the same bytes through caller-supplied messaging (and host staging): for ranks that share one GPU,
// where RCCL refuses to form a communicator, and for callers with a fabric of their own.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/imagepipe_amd.h"
#include "ipk_host.hpp"
#include "ipk_internal.hpp"

using ipk::internal_fail;

namespace {

#define HIPCHK(expr)                                                                                         \
  do { hipError_t e_ = (expr);                                                                               \
       if (e_ != hipSuccess) return internal_fail(IPK_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); } while (0)

// ---- RCCL, resolved at run time ----------------------------------------------------------------------------------------
struct Rccl {
  void *handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

int rccl_load() {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (g_rccl.handle) return IPK_OK;
  // an RCCL the process already mapped (PyTorch's is loaded as "librccl.so") is reused; otherwise the ROCm installation's
  void *h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) return internal_fail(IPK_ERR_UNSUPPORTED, "RCCL not found (dlopen librccl.so.1): %s", dlerror());
#define SYM(field, name)                                                                                     \
  *reinterpret_cast<void **>(&g_rccl.field) = dlsym(h, name);                                                \
  if (!g_rccl.field) return internal_fail(IPK_ERR_UNSUPPORTED, "RCCL symbol %s missing", name)
  SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy");
  SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv"); SYM(AllGather, "ncclAllGather"); SYM(GroupStart, "ncclGroupStart");
  SYM(GroupEnd, "ncclGroupEnd"); SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
  g_rccl.handle = h;
  return IPK_OK;
}
#define NCCLCHK(expr)                                                                                        \
  do { ncclResult_t r_ = (expr);                                                                             \
       if (r_ != ncclSuccess) return internal_fail(IPK_ERR_HIP, "%s failed: %s", #expr, g_rccl.GetErrorString(r_)); } while (0)

static_assert(sizeof(ncclUniqueId) == IPK_COMM_ID_BYTES, "IPK_COMM_ID_BYTES must equal sizeof(ncclUniqueId)");

}  // namespace

struct ipk_comm {
  int rank = 0, nranks = 1, transport = 0;              // 0 RCCL, 1 host
  ncclComm_t nccl = nullptr;
  ipk_exchange_fn exchange = nullptr; void *ctx = nullptr;
  hipStream_t stream = nullptr;                         // gather_begin's stream
  hipEvent_t ev_in = nullptr, ev_out = nullptr; bool pending = false;
  std::vector<uint8_t> stage_a, stage_b;                // host transport staging
};

namespace {

bool bands_ok(const ipk_comm *c, const ipk_band *bands) { return c && bands; }
int neighbour_up(const ipk_band *b, int rank) { for (int k = rank - 1; k >= 0; --k) if (b[k].out_rows) return k; return -1; }
int neighbour_down(const ipk_band *b, int rank, int n) { for (int k = rank + 1; k < n; ++k) if (b[k].out_rows) return k; return -1; }

// the halo exchange's four row addresses on a slab
struct HaloRows { size_t send_up, recv_up, send_down, recv_down; };
HaloRows halo_rows(const ipk_band &b) {
  const size_t top = b.out_row0 - b.src_row0;
  return {top, 0, top + b.out_rows - 1, b.src_rows - 1};
}

}  // namespace

extern "C" {

// ---- band plans (host-only) ----------------------------------------------------------------------------------------------
int ipk_band_plan(size_t height, int nranks, int cfa_period, ipk_band *bands) {
  if (!bands || nranks < 1 || cfa_period < 1 || height < 1) return internal_fail(IPK_ERR_INVALID, "bad band_plan arguments");
  const size_t period = (size_t)cfa_period, units = height / period;
  size_t r = 0;
  for (int k = 0; k < nranks; ++k) {
    const size_t n_units = units / (size_t)nranks + ((size_t)k < units % (size_t)nranks ? 1 : 0);
    const size_t r1 = (k == nranks - 1) ? height : std::min(height, r + n_units * period);
    const size_t s0 = r > 0 ? r - 1 : 0, s1 = std::min(height, r1 + 1);
    bands[k].out_row0 = r; bands[k].out_rows = r1 - r;
    if (bands[k].out_rows) { bands[k].src_row0 = s0; bands[k].src_rows = s1 - s0; }
    else { bands[k].src_row0 = r; bands[k].src_rows = 0; }
    r = r1;
  }
  return IPK_OK;
}

int ipk_band_plan_scaled(size_t height, size_t nheight, int nranks, ipk_band *bands) {
  if (!bands || nranks < 1 || height < 1 || nheight < 2) return internal_fail(IPK_ERR_INVALID, "bad band_plan_scaled arguments");
  // scale_down_buffer's corners (scaling.rs:35-48): topleft (0,0), bottomleft (0, height-1) -> skip_y_y = (height-1)/(nheight-1) in f32
  const float skip = ((float)((int64_t)height - 1) - 0.0f) / ((float)(nheight - 1));
  size_t r = 0;
  for (int k = 0; k < nranks; ++k) {
    const size_t n = nheight / (size_t)nranks + ((size_t)k < nheight % (size_t)nranks ? 1 : 0);
    bands[k].out_row0 = r; bands[k].out_rows = n;
    if (n) {
      // scaling.rs:86-87: from_y = floor(skip*row), to_y = floor(skip*(row+1)), both `as usize` and clamped to height-1
      const size_t from = std::min(height - 1, ipk::f32_to_usize(std::floor(0.0f + skip * (float)r)));
      const size_t to = std::min(height - 1, ipk::f32_to_usize(std::floor(0.0f + skip * (float)(r + n))));
      bands[k].src_row0 = from; bands[k].src_rows = to - from + 1;
    } else { bands[k].src_row0 = 0; bands[k].src_rows = 0; }
    r += n;
  }
  return IPK_OK;
}

// ---- communicators -----------------------------------------------------------------------------------------------------------
int ipk_comm_unique_id(uint8_t *id128) {
  if (!id128) return internal_fail(IPK_ERR_INVALID, "null id");
  int rc = rccl_load(); if (rc) return rc;
  ncclUniqueId id;
  NCCLCHK(g_rccl.GetUniqueId(&id));
  std::memcpy(id128, &id, sizeof(id));
  return IPK_OK;
}

static int comm_common_init(ipk_comm *c) {
  HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  HIPCHK(hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&c->ev_out, hipEventDisableTiming));
  return IPK_OK;
}

int ipk_comm_init_rccl(const uint8_t *id128, int rank, int nranks, ipk_comm **out) {
  int rc = ipk::internal_require_init(); if (rc) return rc;
  if (!id128 || !out || nranks < 1 || rank < 0 || rank >= nranks) return internal_fail(IPK_ERR_INVALID, "bad comm_init_rccl arguments");
  rc = rccl_load(); if (rc) return rc;
  ipk_comm *c = new ipk_comm();
  c->rank = rank; c->nranks = nranks; c->transport = 0;
  ncclUniqueId id; std::memcpy(&id, id128, sizeof(id));
  ncclResult_t r = g_rccl.CommInitRank(&c->nccl, nranks, id, rank);
  if (r != ncclSuccess) { delete c; return internal_fail(IPK_ERR_HIP, "ncclCommInitRank failed: %s", g_rccl.GetErrorString(r)); }
  rc = comm_common_init(c);
  if (rc) { ipk_comm_free(c); return rc; }
  *out = c;
  return IPK_OK;
}

int ipk_comm_init_host(int rank, int nranks, ipk_exchange_fn exchange, void *ctx, ipk_comm **out) {
  if (!out || !exchange || nranks < 1 || rank < 0 || rank >= nranks) return internal_fail(IPK_ERR_INVALID, "bad comm_init_host arguments");
  ipk_comm *c = new ipk_comm();
  c->rank = rank; c->nranks = nranks; c->transport = 1; c->exchange = exchange; c->ctx = ctx;
  *out = c;                                              // device-side objects are created on first device use (the host-slab form needs no GPU)
  return IPK_OK;
}

int ipk_comm_free(ipk_comm *c) {
  if (!c) return IPK_OK;
  if (c->stream) (void)hipStreamSynchronize(c->stream);   // whatever was begun on it is drained before its events and the stream go
  if (c->nccl) (void)g_rccl.CommDestroy(c->nccl);
  if (c->ev_in) (void)hipEventDestroy(c->ev_in);
  if (c->ev_out) (void)hipEventDestroy(c->ev_out);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return IPK_OK;
}

int ipk_comm_info(const ipk_comm *c, int *rank, int *nranks, int *transport) {
  if (!c) return internal_fail(IPK_ERR_INVALID, "null communicator");
  if (rank) *rank = c->rank;
  if (nranks) *nranks = c->nranks;
  if (transport) *transport = c->transport;
  return IPK_OK;
}

// ---- halo exchange ------------------------------------------------------------------------------------------------------------
// host transport, host slab: everyone talks to its upper neighbour first, then to its lower one -- rank k's "down" meets rank k+1's
// "up", so the chain resolves from the top without deadlock
static int host_halo(ipk_comm *c, uint8_t *slab, size_t row_bytes, const ipk_band *bands) {
  const ipk_band &b = bands[c->rank];
  if (!b.out_rows) return IPK_OK;
  const int up = neighbour_up(bands, c->rank), down = neighbour_down(bands, c->rank, c->nranks);
  const HaloRows h = halo_rows(b);
  if (up >= 0 && c->exchange(c->ctx, up, slab + h.send_up * row_bytes, row_bytes, up, slab + h.recv_up * row_bytes, row_bytes))
    return internal_fail(IPK_ERR_HIP, "host transport: exchange with rank %d failed", up);
  if (down >= 0 && c->exchange(c->ctx, down, slab + h.send_down * row_bytes, row_bytes, down, slab + h.recv_down * row_bytes, row_bytes))
    return internal_fail(IPK_ERR_HIP, "host transport: exchange with rank %d failed", down);
  return IPK_OK;
}

int ipk_host_band_exchange_halo(ipk_comm *c, void *slab, size_t row_bytes, const ipk_band *bands) {
  if (!bands_ok(c, bands) || !row_bytes) return internal_fail(IPK_ERR_INVALID, "bad host_band_exchange_halo arguments");
  if (!bands[c->rank].out_rows) return IPK_OK;          // more ranks than CFA periods: an empty band holds and exchanges nothing
  if (!slab) return internal_fail(IPK_ERR_INVALID, "bad host_band_exchange_halo arguments");
  if (c->transport != 1) return internal_fail(IPK_ERR_INVALID, "host slabs need the host transport (RCCL moves device memory)");
  return host_halo(c, static_cast<uint8_t *>(slab), row_bytes, bands);
}

int ipk_band_exchange_halo(ipk_comm *c, void *slab, size_t row_bytes, const ipk_band *bands, void *stream) {
  int rc = ipk::internal_require_init(); if (rc) return rc;
  if (!bands_ok(c, bands) || !row_bytes) return internal_fail(IPK_ERR_INVALID, "bad band_exchange_halo arguments");
  const ipk_band &b = bands[c->rank];
  if (!b.out_rows) return IPK_OK;
  if (!slab) return internal_fail(IPK_ERR_INVALID, "bad band_exchange_halo arguments");
  const int up = neighbour_up(bands, c->rank), down = neighbour_down(bands, c->rank, c->nranks);
  const HaloRows h = halo_rows(b);
  uint8_t *s = static_cast<uint8_t *>(slab);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (c->transport == 0) {
    if (up < 0 && down < 0) return IPK_OK;
    NCCLCHK(g_rccl.GroupStart());
    if (up >= 0) {
      NCCLCHK(g_rccl.Send(s + h.send_up * row_bytes, row_bytes, ncclUint8, up, c->nccl, st));
      NCCLCHK(g_rccl.Recv(s + h.recv_up * row_bytes, row_bytes, ncclUint8, up, c->nccl, st));
    }
    if (down >= 0) {
      NCCLCHK(g_rccl.Send(s + h.send_down * row_bytes, row_bytes, ncclUint8, down, c->nccl, st));
      NCCLCHK(g_rccl.Recv(s + h.recv_down * row_bytes, row_bytes, ncclUint8, down, c->nccl, st));
    }
    NCCLCHK(g_rccl.GroupEnd());
    return IPK_OK;
  }
  // host transport on a device slab: the two own edge rows come down, travel through the caller's messaging, the halos go up
  c->stage_a.resize(4 * row_bytes);
  uint8_t *hs = c->stage_a.data();                      // [send_up | recv_up | send_down | recv_down]
  if (up >= 0) HIPCHK(hipMemcpyAsync(hs, s + h.send_up * row_bytes, row_bytes, hipMemcpyDeviceToHost, st));
  if (down >= 0) HIPCHK(hipMemcpyAsync(hs + 2 * row_bytes, s + h.send_down * row_bytes, row_bytes, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  if (up >= 0 && c->exchange(c->ctx, up, hs, row_bytes, up, hs + row_bytes, row_bytes)) return internal_fail(IPK_ERR_HIP, "host transport: exchange with rank %d failed", up);
  if (down >= 0 && c->exchange(c->ctx, down, hs + 2 * row_bytes, row_bytes, down, hs + 3 * row_bytes, row_bytes))
    return internal_fail(IPK_ERR_HIP, "host transport: exchange with rank %d failed", down);
  if (up >= 0) HIPCHK(hipMemcpyAsync(s + h.recv_up * row_bytes, hs + row_bytes, row_bytes, hipMemcpyHostToDevice, st));
  if (down >= 0) HIPCHK(hipMemcpyAsync(s + h.recv_down * row_bytes, hs + 3 * row_bytes, row_bytes, hipMemcpyHostToDevice, st));
  HIPCHK(hipStreamSynchronize(st));                     // the staging buffer is reused by the next call
  return IPK_OK;
}

// ---- gather -------------------------------------------------------------------------------------------------------------------
static int gather_on(ipk_comm *c, void *frame, size_t out_row_bytes, const ipk_band *bands, int root, hipStream_t st) {
  uint8_t *f = static_cast<uint8_t *>(frame);
  const int n = c->nranks, me = c->rank;
  if (n == 1) return IPK_OK;
  const ipk_band &mine = bands[me];
  if (c->transport == 0) {
    bool equal = root < 0;
    for (int k = 0; k < n && equal; ++k) equal = bands[k].out_rows == bands[0].out_rows && bands[k].out_row0 == (size_t)k * bands[0].out_rows;
    if (equal && bands[0].out_rows) {
      // equal bands lie rank after rank in the frame: the in-place ncclAllGather layout (sendbuff = recvbuff + rank * count)
      const size_t count = bands[0].out_rows * out_row_bytes;
      NCCLCHK(g_rccl.AllGather(f + (size_t)me * count, f, count, ncclUint8, c->nccl, st));
      return IPK_OK;
    }
    NCCLCHK(g_rccl.GroupStart());
    for (int k = 0; k < n; ++k) {
      if (k == me) continue;
      const bool i_send = mine.out_rows && (root < 0 || root == k);
      const bool i_recv = bands[k].out_rows && (root < 0 || root == me);
      if (i_send) NCCLCHK(g_rccl.Send(f + mine.out_row0 * out_row_bytes, mine.out_rows * out_row_bytes, ncclUint8, k, c->nccl, st));
      if (i_recv) NCCLCHK(g_rccl.Recv(f + bands[k].out_row0 * out_row_bytes, bands[k].out_rows * out_row_bytes, ncclUint8, k, c->nccl, st));
    }
    NCCLCHK(g_rccl.GroupEnd());
    return IPK_OK;
  }
  // host transport: own band down once, then n-1 shifted rounds (send to rank+s, receive from rank-s), each received band up
  const size_t my_bytes = mine.out_rows * out_row_bytes;
  c->stage_a.resize(std::max<size_t>(my_bytes, 1));
  if (my_bytes) HIPCHK(hipMemcpyAsync(c->stage_a.data(), f + mine.out_row0 * out_row_bytes, my_bytes, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  for (int s = 1; s < n; ++s) {
    const int to = (me + s) % n, from = (me - s + n) % n;
    const bool i_send = my_bytes && (root < 0 || root == to);
    const bool i_recv = bands[from].out_rows && (root < 0 || root == me);
    const size_t rb = i_recv ? bands[from].out_rows * out_row_bytes : 0;
    c->stage_b.resize(std::max<size_t>(rb, 1));
    if (!i_send && !i_recv) continue;
    if (c->exchange(c->ctx, i_send ? to : -1, c->stage_a.data(), i_send ? my_bytes : 0, i_recv ? from : -1, c->stage_b.data(), rb))
      return internal_fail(IPK_ERR_HIP, "host transport: gather round %d failed", s);
    if (i_recv) {
      HIPCHK(hipMemcpyAsync(f + bands[from].out_row0 * out_row_bytes, c->stage_b.data(), rb, hipMemcpyHostToDevice, st));
      HIPCHK(hipStreamSynchronize(st));
    }
  }
  return IPK_OK;
}

// host transport on a HOST frame: n-1 shifted rounds, bands move between the frames directly
int ipk_host_band_gather(ipk_comm *c, void *frame, size_t out_row_bytes, const ipk_band *bands, int root) {
  if (!bands_ok(c, bands) || !frame || !out_row_bytes || root >= c->nranks) return internal_fail(IPK_ERR_INVALID, "bad host_band_gather arguments");
  if (c->transport != 1) return internal_fail(IPK_ERR_INVALID, "host frames need the host transport (RCCL moves device memory)");
  uint8_t *f = static_cast<uint8_t *>(frame);
  const int n = c->nranks, me = c->rank;
  const ipk_band &mine = bands[me];
  for (int s = 1; s < n; ++s) {
    const int to = (me + s) % n, from = (me - s + n) % n;
    const bool i_send = mine.out_rows && (root < 0 || root == to);
    const bool i_recv = bands[from].out_rows && (root < 0 || root == me);
    if (!i_send && !i_recv) continue;
    if (c->exchange(c->ctx, i_send ? to : -1, f + mine.out_row0 * out_row_bytes, i_send ? mine.out_rows * out_row_bytes : 0,
                    i_recv ? from : -1, f + bands[from].out_row0 * out_row_bytes, i_recv ? bands[from].out_rows * out_row_bytes : 0))
      return internal_fail(IPK_ERR_HIP, "host transport: gather round %d failed", s);
  }
  return IPK_OK;
}

int ipk_band_gather(ipk_comm *c, void *frame, size_t out_row_bytes, const ipk_band *bands, int root, void *stream) {
  int rc = ipk::internal_require_init(); if (rc) return rc;
  if (!bands_ok(c, bands) || !frame || !out_row_bytes || root >= c->nranks) return internal_fail(IPK_ERR_INVALID, "bad band_gather arguments");
  return gather_on(c, frame, out_row_bytes, bands, root, reinterpret_cast<hipStream_t>(stream));
}

int ipk_band_gather_begin(ipk_comm *c, void *frame, size_t out_row_bytes, const ipk_band *bands, int root, void *after_stream) {
  int rc = ipk::internal_require_init(); if (rc) return rc;
  if (!bands_ok(c, bands) || !frame || !out_row_bytes || root >= c->nranks) return internal_fail(IPK_ERR_INVALID, "bad band_gather_begin arguments");
  if (!c->stream) { rc = comm_common_init(c); if (rc) return rc; }
  HIPCHK(hipEventRecord(c->ev_in, reinterpret_cast<hipStream_t>(after_stream)));
  HIPCHK(hipStreamWaitEvent(c->stream, c->ev_in, 0));
  rc = gather_on(c, frame, out_row_bytes, bands, root, c->stream);
  if (rc) return rc;
  HIPCHK(hipEventRecord(c->ev_out, c->stream));
  c->pending = true;
  return IPK_OK;
}

int ipk_comm_wait(ipk_comm *c, void *stream) {
  if (!c) return internal_fail(IPK_ERR_INVALID, "null communicator");
  if (!c->pending) return IPK_OK;       // no gather has ever been begun on this communicator: nothing to be ordered behind
  // Every caller gets the wait, however many streams ask (a download stream next to the compute stream): `pending` stays set once a gather has
  // begun -- ev_out always holds the LAST gather's completion, and waiting for an event that has already fired costs nothing.
  HIPCHK(hipStreamWaitEvent(reinterpret_cast<hipStream_t>(stream), c->ev_out, 0));
  return IPK_OK;
}

// ---- self-check of a transport ----------------------------------------------------------------------------------------------
}  // extern "C"
namespace {
struct DevBuf {                          // a device allocation that is returned on every path out of the self-check
  void *p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
};
}
extern "C" {
int ipk_comm_selftest(ipk_comm *c) {
  int rc = ipk::internal_require_init(); if (rc) return rc;
  if (!c) return internal_fail(IPK_ERR_INVALID, "null communicator");
  const int n = c->nranks, me = c->rank;
  const size_t row = 4096, rows_each = 3;
  if (c->transport == 0) {
    // ring step straight through RCCL: send to rank+1, receive from rank-1 (with one rank: a self send/recv inside one group), then a
    // 1 KB-per-rank ncclAllGather -- so that even a single-rank communicator drives the library end to end
    const int to = (me + 1) % n, from = (me - 1 + n) % n;
    std::vector<uint8_t> hs(row), hr(row, 0);
    for (size_t i = 0; i < row; ++i) hs[i] = (uint8_t)(me * 31 + i % 200);
    DevBuf bs, br, bg;
    HIPCHK(hipMalloc(&bs.p, row)); HIPCHK(hipMalloc(&br.p, row)); HIPCHK(hipMalloc(&bg.p, 1024 * (size_t)n));
    void *const ds = bs.p, *const dr = br.p, *const dg = bg.p;
    HIPCHK(hipMemcpy(ds, hs.data(), row, hipMemcpyHostToDevice));
    NCCLCHK(g_rccl.GroupStart());
    NCCLCHK(g_rccl.Send(ds, row, ncclUint8, to, c->nccl, nullptr));
    NCCLCHK(g_rccl.Recv(dr, row, ncclUint8, from, c->nccl, nullptr));
    NCCLCHK(g_rccl.GroupEnd());
    NCCLCHK(g_rccl.AllGather(ds, dg, 1024, ncclUint8, c->nccl, nullptr));
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(hr.data(), dr, row, hipMemcpyDeviceToHost));
    std::vector<uint8_t> hg(1024 * (size_t)n);
    HIPCHK(hipMemcpy(hg.data(), dg, hg.size(), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < row; ++i) if (hr[i] != (uint8_t)(from * 31 + i % 200)) return internal_fail(IPK_ERR_HIP, "comm selftest: RCCL ring step delivered wrong bytes");
    for (int k = 0; k < n; ++k) for (size_t i = 0; i < 1024; ++i)
      if (hg[(size_t)k * 1024 + i] != (uint8_t)(k * 31 + i % 200)) return internal_fail(IPK_ERR_HIP, "comm selftest: ncclAllGather delivered wrong bytes");
  }
  // a frame of n equal bands: every rank fills its band with a rank-dependent pattern, gathers, and checks all of it; then the
  // same with ragged bands (rank k holds k+1 rows) to take the grouped send/recv path; then a halo exchange on a 3-band slab
  for (int ragged = 0; ragged < 2; ++ragged) {
    std::vector<ipk_band> bands((size_t)n);
    size_t r = 0;
    for (int k = 0; k < n; ++k) { const size_t rr = ragged ? (size_t)k + 1 : rows_each; bands[(size_t)k] = {r, rr, r ? r - 1 : 0, rr + (r ? 1 : 0) + (k + 1 < n ? 1 : 0)}; r += rr; }
    const size_t total = r * row;
    std::vector<uint8_t> host(total, 0xEE), want(total);
    for (int k = 0; k < n; ++k)
      for (size_t i = 0; i < bands[(size_t)k].out_rows * row; ++i) want[bands[(size_t)k].out_row0 * row + i] = (uint8_t)(17 * k + 3 + (i * 7 + ragged) % 251);
    const ipk_band &b = bands[(size_t)me];
    std::memcpy(host.data() + b.out_row0 * row, want.data() + b.out_row0 * row, b.out_rows * row);
    DevBuf bdev;
    HIPCHK(hipMalloc(&bdev.p, total));
    void *const dev = bdev.p;
    HIPCHK(hipMemcpy(dev, host.data(), total, hipMemcpyHostToDevice));
    rc = ipk_band_gather(c, dev, row, bands.data(), -1, nullptr);
    if (!rc) { HIPCHK(hipDeviceSynchronize()); HIPCHK(hipMemcpy(host.data(), dev, total, hipMemcpyDeviceToHost)); }
    if (!rc && std::memcmp(host.data(), want.data(), total) != 0) rc = internal_fail(IPK_ERR_HIP, "comm selftest: gathered frame differs (ragged=%d, rank %d of %d)", ragged, me, n);
    // halo exchange: slab = [halo above?][own rows][halo below?] cut from the same pattern
    if (!rc) {
      std::vector<uint8_t> slab(b.src_rows * row, 0xDD);
      std::memcpy(slab.data() + (b.out_row0 - b.src_row0) * row, want.data() + b.out_row0 * row, b.out_rows * row);
      DevBuf bslab;
      HIPCHK(hipMalloc(&bslab.p, slab.size()));
      void *const ds = bslab.p;
      HIPCHK(hipMemcpy(ds, slab.data(), slab.size(), hipMemcpyHostToDevice));
      rc = ipk_band_exchange_halo(c, ds, row, bands.data(), nullptr);
      if (!rc) { HIPCHK(hipDeviceSynchronize()); HIPCHK(hipMemcpy(slab.data(), ds, slab.size(), hipMemcpyDeviceToHost)); }
      if (!rc && std::memcmp(slab.data(), want.data() + b.src_row0 * row, slab.size()) != 0)
        rc = internal_fail(IPK_ERR_HIP, "comm selftest: slab after the halo exchange differs (ragged=%d, rank %d of %d)", ragged, me, n);
    }
    if (rc) return rc;
  }
  return IPK_OK;
}

}  // extern "C"
