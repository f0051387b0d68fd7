// TEST DOUBLE, not RCCL: a `librccl.so.1` with NCCL's point-to-point and all-gather semantics for ranks that are THREADS of one process
// sharing one GPU -- the one configuration the real RCCL refuses ("Duplicate GPU detected").  tests/cpp/comm_test.cpp puts it first on
// LD_LIBRARY_PATH so that libimagepipe_amd.so's run-time loader (ipk_comm.cpp: dlopen "librccl.so.1") resolves to it, and then drives the
// library's RCCL transport -- grouped ncclSend/ncclRecv halo exchange, in-place ncclAllGather, ragged and rooted gathers, gather_begin /
// wait -- with 2, 3 and 4 ranks on the single-GPU test box.  What this checks is the library's side of the protocol (peers, offsets, counts,
// group pairing, stream ordering); the real RCCL is exercised by ipk_comm_selftest on a single-rank communicator (tests/test_gpu_bench.py)
// and by the driver's multi-GPU runs.
//
// Semantics kept: operations between ncclGroupStart/End are posted together and complete at the outermost ncclGroupEnd; a send matches
// the receive of its peer in FIFO order per (source, destination); data moves in the order of the streams the calls name (the mock drains
// the sender's stream before the bytes are read and the receiver's before the call returns, which is stronger than NCCL's enqueue-only
// contract and so cannot hide an ordering bug in the caller: anything the caller forgot to order still runs before or after, never during).
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <vector>

namespace {
struct Post { const void *src; size_t bytes; bool done; };
struct Group {
  int n = 0, joined = 0, gen = 0, arrived = 0;
  std::mutex mu; std::condition_variable cv;
  std::map<std::pair<int, int>, std::deque<Post *>> box;   // (from, to) -> posted sends
  std::vector<const void *> ag;                             // all-gather: every rank's send buffer
  void barrier(std::unique_lock<std::mutex> &lk) {
    const int g = gen;
    if (++arrived == n) { arrived = 0; ++gen; cv.notify_all(); }
    else cv.wait(lk, [&] { return gen != g; });
  }
};
std::mutex g_mu;
std::map<unsigned long long, Group *> g_groups;
unsigned long long g_next_id = 1;
}  // namespace

struct ncclComm { Group *g; int rank; };

namespace {
struct Op { bool send; void *buf; size_t bytes; int peer; ncclComm *c; hipStream_t st; };
thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;

size_t type_bytes(ncclDataType_t t) {
  switch (t) { case ncclInt8: case ncclUint8: return 1; case ncclFloat16: case ncclBfloat16: return 2; case ncclInt32: case ncclUint32: case ncclFloat32: return 4; default: return 8; }
}

ncclResult_t run_ops() {
  std::vector<Op> ops; ops.swap(t_ops);
  std::vector<Post *> mine;
  // 1. drain the streams the sends read from, then post every send (non-blocking)
  for (auto &o : ops) if (o.send && hipStreamSynchronize(o.st) != hipSuccess) return ncclUnhandledCudaError;
  for (auto &o : ops) if (o.send) {
    Post *p = new Post{o.buf, o.bytes, false};
    std::lock_guard<std::mutex> lk(o.c->g->mu);
    o.c->g->box[{o.c->rank, o.peer}].push_back(p); mine.push_back(p);
    o.c->g->cv.notify_all();
  }
  // 2. every receive waits for its peer's post, copies device to device on its own stream, and acknowledges
  for (auto &o : ops) if (!o.send) {
    Post *p = nullptr;
    {
      std::unique_lock<std::mutex> lk(o.c->g->mu);
      auto &q = o.c->g->box[{o.peer, o.c->rank}];
      o.c->g->cv.wait(lk, [&] { return !q.empty(); });
      p = q.front(); q.pop_front();
    }
    if (p->bytes != o.bytes) return ncclInvalidArgument;              // NCCL would hang or corrupt: a count mismatch is the caller's bug
    if (o.bytes && hipMemcpyAsync(o.buf, p->src, o.bytes, hipMemcpyDeviceToDevice, o.st) != hipSuccess) return ncclUnhandledCudaError;
    if (hipStreamSynchronize(o.st) != hipSuccess) return ncclUnhandledCudaError;
    { std::lock_guard<std::mutex> lk(o.c->g->mu); p->done = true; o.c->g->cv.notify_all(); }
  }
  // 3. a send completes when its bytes have been taken
  for (size_t i = 0, k = 0; i < ops.size(); ++i) if (ops[i].send) {
    Post *p = mine[k++];
    std::unique_lock<std::mutex> lk(ops[i].c->g->mu);
    ops[i].c->g->cv.wait(lk, [&] { return p->done; });
    delete p;
  }
  return ncclSuccess;
}
}  // namespace

extern "C" {
__attribute__((visibility("default"))) ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
  std::lock_guard<std::mutex> lk(g_mu);
  std::memset(id, 0, sizeof(*id));
  const unsigned long long v = g_next_id++;
  std::memcpy(id->internal, &v, sizeof(v));
  std::memcpy(id->internal + 8, "MOCKRCCL", 8);
  return ncclSuccess;
}
__attribute__((visibility("default"))) ncclResult_t ncclCommInitRank(ncclComm_t *out, int nranks, ncclUniqueId id, int rank) {
  unsigned long long v; std::memcpy(&v, id.internal, sizeof(v));
  Group *g;
  { std::lock_guard<std::mutex> lk(g_mu); auto &slot = g_groups[v]; if (!slot) { slot = new Group(); slot->n = nranks; slot->ag.resize((size_t)nranks); } g = slot; }
  if (g->n != nranks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  *out = new ncclComm{g, rank};
  std::unique_lock<std::mutex> lk(g->mu);
  g->barrier(lk);                                                      // like the real one: returns when every rank has joined
  return ncclSuccess;
}
__attribute__((visibility("default"))) ncclResult_t ncclCommDestroy(ncclComm_t c) { delete c; return ncclSuccess; }
__attribute__((visibility("default"))) const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : (r == ncclInvalidArgument ? "invalid argument (mock: count mismatch)" : "mock rccl error"); }
__attribute__((visibility("default"))) ncclResult_t ncclGroupStart() { ++t_depth; return ncclSuccess; }
__attribute__((visibility("default"))) ncclResult_t ncclGroupEnd() { if (--t_depth > 0) return ncclSuccess; return run_ops(); }
__attribute__((visibility("default"))) ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t st) {
  if (peer < 0 || peer >= c->g->n) return ncclInvalidArgument;
  t_ops.push_back({true, const_cast<void *>(buf), count * type_bytes(t), peer, c, st});
  return t_depth ? ncclSuccess : run_ops();
}
__attribute__((visibility("default"))) ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t st) {
  if (peer < 0 || peer >= c->g->n) return ncclInvalidArgument;
  t_ops.push_back({false, buf, count * type_bytes(t), peer, c, st});
  return t_depth ? ncclSuccess : run_ops();
}
__attribute__((visibility("default"))) ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t t, ncclComm_t c, hipStream_t st) {
  const size_t bytes = count * type_bytes(t);
  Group *g = c->g;
  if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
  std::unique_lock<std::mutex> lk(g->mu);
  g->ag[(size_t)c->rank] = send;
  g->barrier(lk);
  std::vector<const void *> srcs = g->ag;
  lk.unlock();
  for (int k = 0; k < g->n; ++k) {
    char *dst = static_cast<char *>(recv) + (size_t)k * bytes;
    if (dst != srcs[(size_t)k] && bytes && hipMemcpyAsync(dst, srcs[(size_t)k], bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return ncclUnhandledCudaError;
  }
  if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
  lk.lock();
  g->barrier(lk);                                                      // nobody's send buffer is reused before everyone has read it
  return ncclSuccess;
}
}  // extern "C"
