// A plain C++ host driving SEVERAL library contexts from ONE process -- the shape the Rust drop-in has ("Multi-GPU from the Rust side"):
//   multi_test <n_frames> <width> <height> <out.f32> <device> [<device> ...]
// The device list becomes the device set (ipk_init_devices; an ordinal may repeat: two contexts on one GPU).  The same shoot of synthetic u16 frames is
// developed four ways and every result must equal the first bit for bit:
//   1. ipk_host_pipeline_run, frame by frame, on the process's default context                       (the reference's loop over Pipeline::run)
//   2. ipk_host_pipeline_run_batch_multi: host buffers dealt frame i -> member i mod N               (8-bit and f32)
//   3. one host thread per member -- ipk_ctx_make_current, then the ordinary single-frame entry point -- the caller's own worker pool
//   4. frames resident on their member's device: ipk_malloc under that member, ipk_pipeline_run_batch_multi + ipk_devices_sync, copied back
// Frame 0's f32 result is written to <out.f32> for the Python test to compare with the oracle.  No Python, no torch.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <string>
#include <thread>
#include <vector>
#include "imagepipe_amd.h"

#define CHECK(expr) do { const int rc_ = (expr); if (rc_ < 0) { std::fprintf(stderr, "%s failed (%d): %s\n", #expr, rc_, ipk_last_error()); return 3; } } while (0)

int main(int argc, char **argv) {
  if (argc < 6) { std::fprintf(stderr, "usage: multi_test <n_frames> <width> <height> <out.f32> <device> [<device> ...]\n"); return 2; }
  const size_t n = (size_t)std::atol(argv[1]), W = (size_t)std::atol(argv[2]), H = (size_t)std::atol(argv[3]);
  std::vector<int> devs;
  for (int i = 5; i < argc; ++i) devs.push_back(std::atoi(argv[i]));
  CHECK(ipk_init_devices(devs.data(), (int)devs.size()));
  const int nd = ipk_device_set_size();
  if (nd != (int)devs.size() || !ipk_is_initialized()) { std::fprintf(stderr, "device set has %d members\n", nd); return 3; }
  for (int k = 0; k < nd; ++k) if (ipk_ctx_device(ipk_device_ctx(k)) != devs[(size_t)k]) { std::fprintf(stderr, "member %d is on the wrong device\n", k); return 3; }

  ipk_pipeline_desc d = IPK_PIPELINE_DESC_INIT;
  d.src_type = IPK_SRC_U16; d.width = W; d.height = H; d.cpp = 1; d.is_cfa = 1; std::strcpy(d.cfa, "RGGB");
  for (int i = 0; i < 4; ++i) { d.blacklevels[i] = 512.0f; d.whitelevels[i] = 16383.0f; }
  const float wb[4] = {2.0f, 1.0f, 1.5f, NAN};
  std::memcpy(d.wb_coeffs, wb, sizeof(wb));
  float m43[12];
  ipk_const_matrix(2, m43);
  const float scale[3] = {1.10f, 1.05f, 1.20f};
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) d.cam_to_xyz_normalized[r * 4 + c] = m43[r * 4 + c] * scale[r];
  d.npoints = 1; d.points[0] = 0.5f; d.points[1] = 0.6f; d.allow_fused = 1; d.use_fastpath = 1;
  size_t dw, dh, fw, fh;
  CHECK(ipk_pipeline_sizes(&d, &dw, &dh, &fw, &fh));
  const size_t in_b = W * H * 2, px3 = fw * fh * 3;

  // the shoot: page-locked buffers, visible to every device of the set
  std::vector<uint16_t *> raws(n);
  uint64_t s = 0x9E3779B97F4A7C15ull;
  for (size_t i = 0; i < n; ++i) {
    raws[i] = static_cast<uint16_t *>(ipk_host_alloc(in_b));
    if (!raws[i]) { std::fprintf(stderr, "ipk_host_alloc failed\n"); return 3; }
    for (size_t j = 0; j < W * H; ++j) { s = s * 6364136223846793005ull + 1442695040888963407ull; raws[i][j] = (uint16_t)((s >> 33) % 16384); }
  }
  auto host_f32 = [&]() { std::vector<float *> v(n); for (auto &p : v) p = static_cast<float *>(ipk_host_alloc(px3 * 4)); return v; };
  auto same = [&](const std::vector<float *> &a, const std::vector<float *> &b, const char *what) {
    for (size_t i = 0; i < n; ++i) if (std::memcmp(a[i], b[i], px3 * 4) != 0) { std::fprintf(stderr, "%s: frame %zu differs from the single-context result\n", what, i); return false; }
    return true;
  };

  // 1. the reference's loop, one context
  std::vector<float *> ref = host_f32();
  for (size_t i = 0; i < n; ++i) CHECK(ipk_host_pipeline_run(&d, raws[i], ref[i], IPK_OUT_F32, nullptr));

  // 2. the batch dealt over the set, f32 and 8-bit
  std::vector<float *> out2 = host_f32();
  std::vector<const void *> srcs(raws.begin(), raws.end());
  { std::vector<void *> dsts(out2.begin(), out2.end()); int fused = -1;
    CHECK(ipk_host_pipeline_run_batch_multi(&d, srcs.data(), dsts.data(), n, IPK_OUT_F32, &fused));
    if (!same(out2, ref, "ipk_host_pipeline_run_batch_multi") || fused != 1) return 4; }
  { std::vector<uint8_t *> o8(n), r8(n); std::vector<void *> dsts(n);
    for (size_t i = 0; i < n; ++i) { o8[i] = static_cast<uint8_t *>(ipk_host_alloc(px3)); r8[i] = static_cast<uint8_t *>(ipk_host_alloc(px3)); dsts[i] = o8[i]; }
    CHECK(ipk_host_pipeline_run_batch_multi(&d, srcs.data(), dsts.data(), n, IPK_OUT_U8, nullptr));
    for (size_t i = 0; i < n; ++i) {
      CHECK(ipk_host_pipeline_run(&d, raws[i], r8[i], IPK_OUT_U8, nullptr));
      if (std::memcmp(o8[i], r8[i], px3) != 0) { std::fprintf(stderr, "8-bit frame %zu differs\n", i); return 4; }
      ipk_host_free(o8[i]); ipk_host_free(r8[i]);
    } }

  // 3. the caller's own worker threads, one per member
  std::vector<float *> out3 = host_f32();
  { std::vector<int> rcs((size_t)nd, 0); std::vector<std::thread> th;
    for (int k = 0; k < nd; ++k) th.emplace_back([&, k]() {
      int rc = ipk_ctx_make_current(ipk_device_ctx(k));
      for (size_t i = (size_t)k; i < n && rc >= 0; i += (size_t)nd) rc = ipk_host_pipeline_run(&d, raws[i], out3[i], IPK_OUT_F32, nullptr);
      if (rc < 0) std::fprintf(stderr, "worker %d: %s\n", k, ipk_last_error());
      rcs[(size_t)k] = rc;
    });
    for (auto &t : th) t.join();
    for (int rc : rcs) if (rc < 0) return 5;
    if (!same(out3, ref, "worker threads on their own contexts")) return 5; }

  // 4. frames resident on their member's device
  std::vector<float *> out4 = host_f32();
  { std::vector<void *> sd(n, nullptr), dd(n, nullptr);
    for (size_t i = 0; i < n; ++i) {
      CHECK(ipk_ctx_make_current(ipk_device_ctx((int)(i % (size_t)nd))));
      CHECK(ipk_malloc(&sd[i], in_b)); CHECK(ipk_malloc(&dd[i], px3 * 4));
      CHECK(ipk_memcpy_h2d(sd[i], raws[i], in_b, nullptr)); CHECK(ipk_stream_sync(nullptr));
    }
    CHECK(ipk_ctx_make_current(nullptr));
    std::vector<const void *> cs(sd.begin(), sd.end());
    int fused = -1;
    CHECK(ipk_pipeline_run_batch_multi(&d, cs.data(), dd.data(), n, IPK_OUT_F32, &fused));
    CHECK(ipk_devices_sync());
    for (size_t i = 0; i < n; ++i) {
      CHECK(ipk_ctx_make_current(ipk_device_ctx((int)(i % (size_t)nd))));
      CHECK(ipk_memcpy_d2h(out4[i], dd[i], px3 * 4, nullptr)); CHECK(ipk_stream_sync(nullptr));
      CHECK(ipk_free(sd[i])); CHECK(ipk_free(dd[i]));
    }
    CHECK(ipk_ctx_make_current(nullptr));
    if (!same(out4, ref, "ipk_pipeline_run_batch_multi") || fused != 1) return 6; }

  // the dealing rule the entry points used
  for (int k = 0; k < nd; ++k) {
    size_t first, stride, count;
    CHECK(ipk_deal_frames(n, nd, k, &first, &stride, &count));
    if (first != (size_t)k || stride != (size_t)nd || count != (n > (size_t)k ? (n - (size_t)k + (size_t)nd - 1) / (size_t)nd : 0)) { std::fprintf(stderr, "ipk_deal_frames\n"); return 7; }
  }
  FILE *o = std::fopen(argv[4], "wb");
  if (!o || std::fwrite(ref[0], 4, px3, o) != px3) { std::fprintf(stderr, "cannot write %s\n", argv[4]); return 2; }
  std::fclose(o);
  { FILE *r = std::fopen((std::string(argv[4]) + ".u16").c_str(), "wb"); if (r) { std::fwrite(raws[0], 2, W * H, r); std::fclose(r); } }
  std::printf("MULTI_OK %d members, %zu frames of %zux%zu -> %zux%zu\n", nd, n, W, H, fw, fh);
  ipk_shutdown();
  return 0;
}
