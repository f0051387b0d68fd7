// The multi-GPU entry points of the C ABI driven from a plain C++ host -- no Python, no torch: what the north_star's Rust caller does.
//   comm_test <nranks> <width> <height> [gpu|rccl]
// N "ranks" are threads of this process; the host transport's exchange callback (ipk_exchange_fn, MPI_Sendrecv semantics) is a
// mailbox in shared memory.  Without `gpu`: host slabs and frames (ipk_host_band_exchange_halo / ipk_host_band_gather) -- runs anywhere.
// With `gpu` (the ranks share device 0, which is why the transport is the host one: RCCL wants one GPU per rank): device slabs, the
// band form of the fused kernel writing into its rows of the frame, ipk_band_gather_begin / ipk_comm_wait in place -- and every rank's
// gathered frame must equal, bit for bit, what ONE ipk_raw_to_srgb launch computes for the whole frame.
// With `rccl`: the same on the library's RCCL transport (ipk_comm_unique_id / ipk_comm_init_rccl: grouped ncclSend/ncclRecv, in-place
// ncclAllGather) -- which needs one GPU per rank with the real RCCL, so the caller (tests/test_gpu_bench.py) puts the test double
// tests/cpp/mock_rccl.cpp first on LD_LIBRARY_PATH; plus a gather to one root.
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <thread>
#include <vector>
#include "imagepipe_amd.h"

namespace {
struct Mailbox {                                 // (from, to) -> queued messages
  std::mutex mu; std::condition_variable cv;
  std::map<std::pair<int, int>, std::vector<std::vector<unsigned char>>> q;
};
struct Ctx { Mailbox *mb; int rank; };
int exchange(void *vctx, int send_peer, const void *send, size_t send_bytes, int recv_peer, void *recv, size_t recv_bytes) {
  Ctx *c = static_cast<Ctx *>(vctx);
  if (send_peer >= 0 && send_bytes) {
    std::lock_guard<std::mutex> lk(c->mb->mu);
    const unsigned char *p = static_cast<const unsigned char *>(send);
    c->mb->q[{c->rank, send_peer}].emplace_back(p, p + send_bytes);
    c->mb->cv.notify_all();
  }
  if (recv_peer >= 0 && recv_bytes) {
    std::unique_lock<std::mutex> lk(c->mb->mu);
    auto &box = c->mb->q[{recv_peer, c->rank}];
    c->mb->cv.wait(lk, [&] { return !box.empty(); });
    if (box.front().size() != recv_bytes) return 1;
    std::memcpy(recv, box.front().data(), recv_bytes);
    box.erase(box.begin());
  }
  return 0;
}
#define CHECK(expr) do { int rc_ = (expr); if (rc_ < 0) { std::fprintf(stderr, "rank %d: %s -> %d: %s\n", rank, #expr, rc_, ipk_last_error()); ok = false; return; } } while (0)
}  // namespace

int main(int argc, char **argv) {
  if (argc < 4) { std::fprintf(stderr, "usage: comm_test nranks width height [gpu]\n"); return 2; }
  const int n = std::atoi(argv[1]);
  const size_t W = std::atol(argv[2]), H = std::atol(argv[3]);
  const bool rccl = argc > 4 && std::strcmp(argv[4], "rccl") == 0;
  const bool gpu = rccl || (argc > 4 && std::strcmp(argv[4], "gpu") == 0);
  std::vector<uint16_t> raw(W * H);
  uint64_t s = 0x9E3779B97F4A7C15ull;
  for (auto &v : raw) { s = s * 6364136223846793005ull + 1442695040888963407ull; v = (uint16_t)((s >> 33) % 16384); }
  std::vector<ipk_band> bands((size_t)n);
  if (ipk_band_plan(H, n, 2, bands.data()) != 0) { std::fprintf(stderr, "band plan: %s\n", ipk_last_error()); return 3; }

  ipk_fused_params p = IPK_FUSED_PARAMS_INIT;
  p.src_type = IPK_SRC_U16; p.owidth = W; p.width = W; p.height = H; p.black0 = 512.0f; p.white0 = 16383.0f;
  std::strcpy(p.cfa, "RGGB");
  const float wb[4] = {2.0f, 1.0f, 1.5f, 1.0f};
  std::memcpy(p.wb_coeffs, wb, sizeof(wb));
  float m43[12];
  ipk_const_matrix(2, m43);
  const float scale[3] = {1.10f, 1.05f, 1.20f};
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) p.cam_to_xyz_normalized[r * 4 + c] = m43[r * 4 + c] * scale[r];
  p.npoints = 1; p.points[0] = 0.5f; p.points[1] = 0.6f; p.out_type = IPK_OUT_F32;

  std::vector<float> whole;
  if (gpu) {
    if (ipk_init(0) != 0) { std::fprintf(stderr, "%s\n", ipk_last_error()); return 3; }
    void *dsrc = nullptr, *ddst = nullptr;
    whole.resize(W * H * 3);
    if (ipk_malloc(&dsrc, raw.size() * 2) || ipk_malloc(&ddst, whole.size() * 4) || ipk_memcpy_h2d(dsrc, raw.data(), raw.size() * 2, nullptr) ||
        ipk_raw_to_srgb(&p, dsrc, ddst, nullptr) || ipk_memcpy_d2h(whole.data(), ddst, whole.size() * 4, nullptr) || ipk_stream_sync(nullptr)) {
      std::fprintf(stderr, "whole frame: %s\n", ipk_last_error()); return 3;
    }
    ipk_free(dsrc); ipk_free(ddst);
  }

  uint8_t id[IPK_COMM_ID_BYTES];
  if (rccl && ipk_comm_unique_id(id) != 0) { std::fprintf(stderr, "unique id: %s\n", ipk_last_error()); return 3; }
  Mailbox mb;
  std::vector<int> results((size_t)n, 0);
  std::vector<std::thread> th;
  for (int rank = 0; rank < n; ++rank) th.emplace_back([&, rank] {
    bool ok = true;
    [&] {
      Ctx ctx{&mb, rank};
      ipk_comm *comm = nullptr;
      if (rccl) CHECK(ipk_comm_init_rccl(id, rank, n, &comm));
      else CHECK(ipk_comm_init_host(rank, n, exchange, &ctx, &comm));
      const ipk_band b = bands[(size_t)rank];
      const size_t top = b.out_row0 - b.src_row0;
      if (!gpu) {
        // host slab: own rows filled, halo rows poisoned; after the exchange it must equal the frame's rows [src_row0, +src_rows)
        std::vector<uint16_t> slab(b.src_rows * W, 0xDEAD);
        std::memcpy(slab.data() + top * W, raw.data() + b.out_row0 * W, b.out_rows * W * 2);
        CHECK(ipk_host_band_exchange_halo(comm, slab.data(), W * 2, bands.data()));
        if (std::memcmp(slab.data(), raw.data() + b.src_row0 * W, slab.size() * 2) != 0) { std::fprintf(stderr, "rank %d: slab differs\n", rank); ok = false; }
        // host gather in place: a byte image whose rows carry their owner's pattern
        std::vector<unsigned char> frame(H * W, 0xEE), want(H * W);
        for (size_t i = 0; i < want.size(); ++i) want[i] = (unsigned char)(raw[i] * 7 + 3);
        std::memcpy(frame.data() + b.out_row0 * W, want.data() + b.out_row0 * W, b.out_rows * W);
        CHECK(ipk_host_band_gather(comm, frame.data(), W, bands.data(), -1));
        if (frame != want) { std::fprintf(stderr, "rank %d: gathered host frame differs\n", rank); ok = false; }
      } else {
        CHECK(ipk_comm_selftest(comm));
        void *slab = nullptr, *frame = nullptr;
        CHECK(ipk_malloc(&slab, b.src_rows * W * 2 + 16));
        CHECK(ipk_malloc(&frame, H * W * 3 * 4));
        CHECK(ipk_memcpy_h2d(static_cast<uint16_t *>(slab) + top * W, raw.data() + b.out_row0 * W, b.out_rows * W * 2, nullptr));
        CHECK(ipk_band_exchange_halo(comm, slab, W * 2, bands.data(), nullptr));
        ipk_fused_params pb = p;
        pb.band_src_row0 = b.src_row0; pb.band_src_rows = b.src_rows; pb.band_out_row0 = b.out_row0; pb.band_out_rows = b.out_rows;
        if (b.out_rows) CHECK(ipk_raw_to_srgb(&pb, slab, static_cast<float *>(frame) + b.out_row0 * W * 3, nullptr));
        CHECK(ipk_band_gather_begin(comm, frame, W * 3 * 4, bands.data(), -1, nullptr));
        CHECK(ipk_comm_wait(comm, nullptr));
        std::vector<float> got(W * H * 3);
        CHECK(ipk_memcpy_d2h(got.data(), frame, got.size() * 4, nullptr));
        CHECK(ipk_stream_sync(nullptr));
        if (std::memcmp(got.data(), whole.data(), got.size() * 4) != 0) { std::fprintf(stderr, "rank %d: banded frame differs from the whole-frame launch\n", rank); ok = false; }
        // the same bands gathered to ONE root (the last rank): it ends up with the frame, nobody else's buffer is written
        std::vector<float> part(W * H * 3, -7.0f);
        std::memcpy(part.data() + b.out_row0 * W * 3, whole.data() + b.out_row0 * W * 3, b.out_rows * W * 3 * 4);
        std::vector<float> before = part;
        CHECK(ipk_memcpy_h2d(frame, part.data(), part.size() * 4, nullptr));
        CHECK(ipk_band_gather(comm, frame, W * 3 * 4, bands.data(), n - 1, nullptr));
        CHECK(ipk_memcpy_d2h(part.data(), frame, part.size() * 4, nullptr));
        CHECK(ipk_stream_sync(nullptr));
        if (std::memcmp(part.data(), rank == n - 1 ? whole.data() : before.data(), part.size() * 4) != 0) { std::fprintf(stderr, "rank %d: rooted gather wrong\n", rank); ok = false; }
        ipk_free(slab); ipk_free(frame);
      }
      ipk_comm_free(comm);
    }();
    results[(size_t)rank] = ok ? 1 : 0;
  });
  for (auto &t : th) t.join();
  for (int r = 0; r < n; ++r) if (!results[(size_t)r]) { std::fprintf(stderr, "rank %d failed\n", r); return 1; }
  std::printf("COMM_OK nranks=%d %zux%zu %s\n", n, W, H, rccl ? "rccl" : (gpu ? "gpu" : "host"));
  return 0;
}
