// Development tool (round 6): what does the fused kernel's WALK cost the memory system, arithmetic aside?  A persistent launch like the kernel's (256 blocks
// of sixteen waves) moves an f32 mosaic to an f32 RGB frame -- 4 bytes read, 12 written per pixel, nontemporal 16-byte stores, row r + 2's loads in flight --
// with no arithmetic, no LDS and no halo, under different dealings of (strip, row segment) tasks to waves.  ipk_mix_probe (a flat, contiguous launch of the
// same traffic) reaches 0.70-0.72 of the HBM peak; ipk_stream_probe (the kernel's real skeleton) 0.61-0.65.
//   build: hipcc --offload-arch=gfx950 -O3 tools/walk_probe.hip -o tools/build/walk_probe      run (GPU box): tools/build/walk_probe [W H]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef float f4 __attribute__((ext_vector_type(4)));

struct Args {
  const float *src; float *dst;
  unsigned W, H, n_strips, n_segs, seg_rows, mode, n_tasks;
};

// V = 16-byte groups per lane per row (1: 256-pixel strips, 2: 512-pixel strips, 4: 1024-pixel strips)
template <int V>
__global__ __launch_bounds__(1024) void k_walk(Args a) {
  const unsigned lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  const unsigned n_waves = gridDim.x * 16u;
  const unsigned SW = 256u * V;
  for (unsigned k = 0;; ++k) {
    unsigned t;
    const unsigned g = blockIdx.x * 16u + wv;              // global wave index
    if (a.mode == 0) {                                     // the kernel's static dealing: groups of four neighbouring tasks a round of blocks apart
      t = ((wv / 4u) * gridDim.x + blockIdx.x) * 4u + (wv % 4u) + k * n_waves;
    } else if (a.mode == 1) {                              // plain: wave g takes task g (a block = sixteen neighbouring strips of one segment)
      t = g + k * n_waves;
    } else if (a.mode == 2) {                              // segment-major rounds: all waves work inside one advancing band of rows
      t = g + k * n_waves;
    } else if (a.mode == 3) {                              // spread 1: every wave of a block a whole round of blocks apart
      t = wv * gridDim.x + blockIdx.x + k * n_waves;
    } else if (a.mode == 4) {                              // spread 2
      t = ((wv / 2u) * gridDim.x + blockIdx.x) * 2u + (wv % 2u) + k * n_waves;
    } else if (a.mode == 5) {                              // spread 8
      t = ((wv / 8u) * gridDim.x + blockIdx.x) * 8u + (wv % 8u) + k * n_waves;
    } else if (a.mode == 6) {                              // the kernel's dealing over a STRIP-major numbering (neighbouring tasks = vertically adjacent segments)
      const unsigned u = ((wv / 4u) * gridDim.x + blockIdx.x) * 4u + (wv % 4u) + k * n_waves;
      t = u < a.n_tasks ? (u % a.n_segs) * a.n_strips + u / a.n_segs : u;
    } else if (a.mode == 8 || a.mode == 9) {               // block-synchronous: the block's waves take 16 (8: 8 + 8 a round of blocks apart) neighbouring strips of one
      const unsigned grp = a.mode == 8 ? 16u : 8u;         // segment and meet at a barrier every row: 48 (24) KB of every output row written together
      const unsigned ngrp = (a.n_strips + grp - 1) / grp;  // strip groups per segment
      const unsigned slot = (wv / grp) * gridDim.x + blockIdx.x + k * (gridDim.x * (16u / grp));      // which (segment, group)
      const unsigned seg = slot / ngrp, gi = slot % ngrp;
      const unsigned strip = min(gi * grp + (wv % grp), a.n_strips - 1);
      t = seg < a.n_segs ? seg * a.n_strips + strip : 0xFFFFFFFFu;
    } else {                                               // 7: XCD-contiguous: block b runs on XCD b % 8; XCD x gets the x-th eighth of the tasks
      const unsigned xcd = blockIdx.x & 7u, j = blockIdx.x >> 3, per = (a.n_tasks + 7u) / 8u;
      const unsigned u = (j * 16u + wv) + k * (n_waves / 8u);
      t = u < per ? xcd * per + u : 0xFFFFFFFFu;
    }
    t = __builtin_amdgcn_readfirstlane(t);
    if (a.mode >= 8 && a.mode <= 9) {
      // every wave of the block runs the same number of trips and rows (idle ones only meet the barriers)
      const unsigned slots = a.n_segs * ((a.n_strips + (a.mode == 8 ? 16u : 8u) - 1) / (a.mode == 8 ? 16u : 8u));
      if (blockIdx.x + k * (gridDim.x * (a.mode == 8 ? 1u : 2u)) >= slots) break;
      if (t >= a.n_tasks) { for (unsigned r = 0; r < a.seg_rows; ++r) __builtin_amdgcn_s_barrier(); continue; }
    } else
    if (t >= a.n_tasks) break;
    const unsigned strip = t % a.n_strips, seg = t / a.n_strips;
    const unsigned r0 = seg * a.seg_rows, r1 = min(a.H, r0 + a.seg_rows);
    const unsigned pc0 = min(strip * SW, a.W - SW);
    const float *sp = a.src + (size_t)r0 * a.W + pc0 + 4u * lane;
    float *dp = a.dst + ((size_t)r0 * a.W + pc0) * 3u + 4u * lane;
    f4 cur[V], nxt[V], nn[V];
    #pragma unroll
    for (int v = 0; v < V; ++v) { cur[v] = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(sp + 256 * v)); }
    #pragma unroll
    for (int v = 0; v < V; ++v) { nxt[v] = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(sp + (size_t)min(1u, r1 - r0 - 1) * a.W + 256 * v)); }
    for (unsigned r = r0; r < r0 + (a.mode >= 8 && a.mode <= 9 ? a.seg_rows : r1 - r0); ++r) {
      if (a.mode >= 8 && a.mode <= 9) { __builtin_amdgcn_s_barrier(); if (r >= r1) continue; }
      const unsigned ahead = min(r + 2u, r1 - 1u) - r0;
      #pragma unroll
      for (int v = 0; v < V; ++v) nn[v] = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(sp + (size_t)ahead * a.W + 256 * v));
      float *o = dp + (size_t)(r - r0) * a.W * 3u;
      #pragma unroll
      for (int v = 0; v < V; ++v) {
        __builtin_nontemporal_store(cur[v], reinterpret_cast<f4 *>(o + 768 * v));
        __builtin_nontemporal_store(cur[v], reinterpret_cast<f4 *>(o + 768 * v + 256));
        __builtin_nontemporal_store(cur[v], reinterpret_cast<f4 *>(o + 768 * v + 512));
      }
      #pragma unroll
      for (int v = 0; v < V; ++v) { cur[v] = nxt[v]; nxt[v] = nn[v]; }
    }
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char **argv) {
  const unsigned W = argc > 2 ? atoi(argv[1]) : 10000, H = argc > 2 ? atoi(argv[2]) : 10000;
  float *src, *dst;
  CK(hipMalloc(&src, (size_t)W * H * 4)); CK(hipMalloc(&dst, (size_t)W * H * 12));
  std::vector<float> h((size_t)W * H);
  unsigned s = 12345u;
  for (auto &v : h) { s = s * 1664525u + 1013904223u; v = (float)(s >> 18); }
  CK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double bytes = 16.0 * W * H;
  auto run = [&](const char *name, int V, unsigned mode, unsigned seg_rows) {
    Args a; a.src = src; a.dst = dst; a.W = W; a.H = H; a.mode = mode;
    const unsigned SW = 256u * V;
    a.n_strips = (W + SW - 1) / SW;
    a.seg_rows = seg_rows ? seg_rows : std::max(1u, (unsigned)(((unsigned long long)H * a.n_strips + 4095) / 4096));
    a.n_segs = (H + a.seg_rows - 1) / a.seg_rows;
    a.n_tasks = a.n_strips * a.n_segs;
    auto launch = [&]() { if (V == 1) hipLaunchKernelGGL(k_walk<1>, dim3(256), dim3(1024), 0, 0, a); else if (V == 2) hipLaunchKernelGGL(k_walk<2>, dim3(256), dim3(1024), 0, 0, a); else hipLaunchKernelGGL(k_walk<4>, dim3(256), dim3(1024), 0, 0, a); };
    for (int i = 0; i < 30; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
    printf("%-58s strips %3u x segs %4u of %3u rows (%5u tasks): %.4f ms %5.0f GB/s  %.3f of peak\n", name, a.n_strips, a.n_segs, a.seg_rows, a.n_tasks, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 8000);
  };
  for (int rep = 0; rep < 2; ++rep) {
    run("256-px strips, one task per wave, the kernel's dealing", 1, 0, 0);
    run("256-px strips, one task per wave, wave g = task g", 1, 1, 0);
    run("512-px strips, one task per wave, the kernel's dealing", 2, 0, 0);
    run("512-px strips, one task per wave, wave g = task g", 2, 1, 0);
    run("1024-px strips, one task per wave, the kernel's dealing", 4, 0, 0);
    run("256-px strips, 16 neighbours per block in lockstep (barrier per row)", 1, 8, 0);
    run("256-px strips, 2 x 8 neighbours per block in lockstep", 1, 9, 0);
    run("256-px strips, 16 neighbours in lockstep, 32-row tasks", 1, 8, 32);
    run("256-px strips, spread 1", 1, 3, 0);
    run("256-px strips, spread 2", 1, 4, 0);
    run("256-px strips, spread 8", 1, 5, 0);
    run("256-px strips, kernel's dealing over strip-major tasks", 1, 6, 0);
    run("256-px strips, one eighth of the frame per XCD", 1, 7, 0);
    run("512-px strips, spread 1", 2, 3, 0);
    run("512-px strips, one eighth of the frame per XCD", 2, 7, 0);
    run("256-px strips, 32-row tasks walked in order (band)", 1, 2, 32);
    run("256-px strips, 8-row tasks walked in order (band)", 1, 2, 8);
    run("512-px strips, 16-row tasks walked in order (band)", 2, 2, 16);
    run("512-px strips, 4-row tasks walked in order (band)", 2, 2, 4);
  }
  return 0;
}
