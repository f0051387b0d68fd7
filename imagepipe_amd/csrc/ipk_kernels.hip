// ipk_kernels.hip -- gfx950 kernels of the raw->sRGB hot path and their launchers.
//
// Staged kernels (one per reference op, so stage boundaries stay materialisable for the
// reference's per-op cache, src/pipeline.rs:364-372) and the fused raw->sRGB kernel.
// All of them are HBM-streaming kernels: coalesced row-major access, LDS only for the two
// 13-bit lookup tables (and the 48x48 CFA colour table), no MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstring>
#include <cmath>
#include <type_traits>
#include <atomic>
#include <mutex>
#include <utility>
#include <vector>
#include "ipk_device.hpp"
#include "ipk_launch.hpp"

using namespace ipkd;

// What was measured and not kept (instruction-class substitutions, LDS read ordering, unrolled row loops, load-first pipelining, queue shapes,
// ablation builds) lives in profiles/README.md "experiment log" with the commit that last carried each switch; this file holds the shipped form only.
//
// Wave-uniform tests of rare paths are marked unlikely, so that their code is laid out behind the loop and the common path runs through not-taken
// branches (a taken branch makes the wave refetch its instruction buffer).
#define IPK_RARE(x) __builtin_expect(!!(x), 0)
constexpr uint32_t kSpread = 4;        // static schedule: a block's waves start on groups of kSpread neighbouring tasks a round of blocks apart (fused_bayer_body)
constexpr uint32_t kMinTaskRows = 4;   // one task per wave: at least this many rows each (fused_task_grid)
#ifndef IPK_SPLIT_MIN_ROWS
#define IPK_SPLIT_MIN_ROWS 16u   // IPK_SCHED_SPLIT applies when a wave's contiguous piece has at least this many rows (24 MP, 23 rows: photo-like 0.0958 -> 0.0917 ms, noise 0.1154 -> 0.1196)
#endif
constexpr uint32_t kStealMin = 4;      // a takeover needs at least this many rows left behind the owner's current one

namespace ipk {

typedef LutPair LabTab;
typedef LutPair GamTab;
// A block's LDS copies of the two plain 8193-float tables, each in either form (all threads of the block; before its barrier).
// With the usual 1024 threads every thread issues ALL its global loads (8 or 16 per table) before the first LDS write, so the block pays
// the L2 latency once instead of once per loop iteration -- the prologue is a tenth of the point-wise chain's run time on a 3 MP preview.
template <typename T> struct TabRegs;
template <> struct TabRegs<float> {
  float a[kLutPairs / 1024];
  __device__ __forceinline__ void load(const float *__restrict__ plain, uint32_t t) {
    #pragma unroll
    for (int k = 0; k < kLutPairs / 1024; ++k) a[k] = plain[t + k * 1024];
  }
  __device__ __forceinline__ void store(float *__restrict__ lds, const float *__restrict__ plain, uint32_t t) const {
    #pragma unroll
    for (int k = 0; k < kLutPairs / 1024; ++k) lds[t + k * 1024] = a[k];
    if (t == 0) lds[kLutPairs] = plain[kLutPairs];
  }
};
template <> struct TabRegs<LutPair> {
  float a[kLutPairs / 1024], b[kLutPairs / 1024];
  __device__ __forceinline__ void load(const float *__restrict__ plain, uint32_t t) {
    #pragma unroll
    for (int k = 0; k < kLutPairs / 1024; ++k) { a[k] = plain[t + k * 1024]; b[k] = plain[t + k * 1024 + 1]; }
  }
  __device__ __forceinline__ void store(LutPair *__restrict__ lds, const float *__restrict__, uint32_t t) const {
    #pragma unroll
    for (int k = 0; k < kLutPairs / 1024; ++k) lds[t + k * 1024] = make_float2(a[k], b[k] - a[k]);
  }
};
__device__ __forceinline__ void fill_lds_table_any(float *__restrict__ lds, const float *__restrict__ plain) {
  for (int i = threadIdx.x; i < kLutPairs + 1; i += blockDim.x) lds[i] = plain[i];
}
__device__ __forceinline__ void fill_lds_table_any(LutPair *__restrict__ lds, const float *__restrict__ plain) {
  for (int i = threadIdx.x; i < kLutPairs; i += blockDim.x) { const float v1 = plain[i], v2 = plain[i + 1]; lds[i] = make_float2(v1, v2 - v1); }
}
template <typename LT, typename GT>
__device__ __forceinline__ void fill_lds_tables(LT *__restrict__ s_lab, const float *__restrict__ lab, GT *__restrict__ s_gam, const float *__restrict__ gam, bool want_gam = true) {
  if (blockDim.x == 1024) {
    TabRegs<LT> rl; TabRegs<GT> rg;
    rl.load(lab, threadIdx.x);
    if (want_gam) rg.load(gam, threadIdx.x);
    rl.store(s_lab, lab, threadIdx.x);
    if (want_gam) rg.store(s_gam, gam, threadIdx.x);
  } else {
    fill_lds_table_any(s_lab, lab);
    if (want_gam) fill_lds_table_any(s_gam, gam);
  }
}

// The same copies WITHOUT registers: global_load_lds_dwordx4 (gfx950) writes 16 bytes per lane straight into LDS at M0 + lane * 16, so a 1024-thread block
// moves a 64 KB pair table with four instructions per thread and a 32 KB plain one with two, and the loads can sit in flight behind whatever the wave
// does next -- the fused kernel issues them first thing, computes its first task's addresses, issues that task's row loads right behind them and meets the
// block's other waves at ONE barrier in front of its first table read (round 4: load 24 registers, write them to LDS, barrier, and only then the first
// row load: two memory round trips per launch where one will do; the form with the row loads ahead of a register-staged fill spilled).
// The pair image comes from the host (ipk_init builds {v[i], v[i+1] - v[i]} for the staged kernels anyway: the same f32 subtraction).
typedef const __attribute__((address_space(1))) void *ipk_gptr_t;
typedef __attribute__((address_space(3))) void *ipk_lptr_t;
__device__ __forceinline__ void lds_direct_16(const void *g_lane, void *lds_wave_base) {
  __builtin_amdgcn_global_load_lds((ipk_gptr_t)g_lane, (ipk_lptr_t)lds_wave_base, 16, 0, 0);
}
// all 1024 threads; `bytes` a multiple of 16 KB (64 KB pair table, 32 KB = the first 8192 floats of a plain one)
__device__ __forceinline__ void stage_lds_direct(void *lds, const void *g, uint32_t bytes) {
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  for (uint32_t off = wave * 1024u; off < bytes; off += 16u * 1024u)
    lds_direct_16(reinterpret_cast<const char *>(g) + off + lane * 16u, reinterpret_cast<char *>(lds) + off);
}
// The barrier that publishes an LDS-direct fill: the DMA into LDS is counted by vmcnt, which a workgroup barrier does not formally wait for (its fence
// orders lgkmcnt traffic), so the wait is stated here rather than left to what this compiler version happens to emit in front of s_barrier.
__device__ __forceinline__ void sync_after_lds_direct() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}
// (any other block size: the register-staged loop -- no launcher uses one, but a table silently left half-filled would be the worst kind of failure)
__device__ __forceinline__ void stage_table_direct(LutPair *s_tab, const float *plain, const LutPair *pairs) {
  if (blockDim.x == 1024) stage_lds_direct(s_tab, pairs, kLutPairs * 8u);
  else fill_lds_table_any(s_tab, plain);
}
__device__ __forceinline__ void stage_q8_direct(Q8Entry *s_tab, const Q8Entry *q8) {
  if (blockDim.x == 1024) stage_lds_direct(s_tab, q8, kLutPairs * 8u);
  else for (int i = threadIdx.x; i < kLutPairs; i += blockDim.x) s_tab[i] = q8[i];
}
__device__ __forceinline__ void stage_table_direct(float *s_tab, const float *plain, const LutPair *) {
  if (blockDim.x != 1024) { fill_lds_table_any(s_tab, plain); return; }
  stage_lds_direct(s_tab, plain, kLutPairs * 4u);
  if (threadIdx.x == 0) s_tab[kLutPairs] = plain[kLutPairs];            // the 8193rd entry: the plain device table ends there
}

// ------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------
static inline dim3 grid_rows(size_t width, size_t height, int bx, int cols_per_thread = 1) {
  size_t gx = (width + size_t(bx) * cols_per_thread - 1) / (size_t(bx) * cols_per_thread);
  size_t gy = height < 65535 ? height : 65535;
  return dim3((unsigned)gx, (unsigned)gy, 1);
}
// row-looping kernels that set up a per-block LDS table: a bounded number of blocks, each walking many rows
static inline dim3 grid_rows_few(size_t width, size_t height, int bx, unsigned total_blocks) {
  const size_t gx = (width + size_t(bx) - 1) / size_t(bx);
  size_t gy = total_blocks / gx; if (gy < 1) gy = 1; if (gy > height) gy = height; if (gy > 65535) gy = 65535;
  return dim3((unsigned)gx, (unsigned)gy, 1);
}
// Table-free streaming kernels (one element or four per thread) are launched FLAT -- as many blocks as the data needs, the grid-stride loop runs
// once -- where round 2 capped them at 16 blocks per CU: neighbouring blocks run at the same time and touch neighbouring addresses, which the
// memory system rewards (a plain copy: 5.0-5.2 TB/s through a capped grid-stride loop, 6.3-6.45 flat; tools/copy_probe.hip).  IPK_OPT_FLATGRID.
static inline unsigned flat_cap(unsigned) { return 0x7FFFFFFFu; }
// The staged kernels read every byte once and write every byte once: nontemporal accesses (IPK_OPT_NT) keep those streams from displacing each
// other in the L2 / Infinity Cache (plain copy 6.0 -> 6.45 TB/s, tools/copy_probe.hip).
typedef float ipk_f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_stream4(const float *p) {
  const ipk_f4v v = __builtin_nontemporal_load(reinterpret_cast<const ipk_f4v *>(p)); return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st_stream4(float *p, float4 v) {
  ipk_f4v q; q.x = v.x; q.y = v.y; q.z = v.z; q.w = v.w; __builtin_nontemporal_store(q, reinterpret_cast<ipk_f4v *>(p));
}
__device__ __forceinline__ float ld_stream(const float *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void st_stream(float *p, float v) { __builtin_nontemporal_store(v, p); }
template <typename U> __device__ __forceinline__ void st_stream_u(U *p, U v) { __builtin_nontemporal_store(v, p); }
static inline unsigned grid_1d(size_t n, int bx, unsigned cap) {
  size_t g = (n + bx - 1) / bx;
  if (g < 1) g = 1;
  return (unsigned)(g < cap ? g : cap);
}

// ------------------------------------------------------------------------------------------
// OpGoFloat (src/ops/gofloat.rs:84-201)
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void k_gofloat_cfa(const T *__restrict__ src, size_t owidth, size_t x, size_t y, uint32_t width, uint32_t height,
                              float min0, float range0, float *__restrict__ dst) {
  const uint32_t col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= width) return;
  for (uint32_t row = blockIdx.y; row < height; row += gridDim.y) {
    const float v = (float)src[owidth * (row + y) + x + col];
    dst[(size_t)row * width + col] = rs_min((v - min0) / range0, 1.0f);      // gofloat.rs:126 / :162
  }
}
// Four columns per thread (one 16-byte store, one 8/16-byte load; dword alignment is all gfx950 needs for them), rows
// strided over gridDim.y: the op is pure streaming, 2 or 4 bytes in and 4 out per sample.
struct __attribute__((aligned(4))) GfF4 { float x, y, z, w; };
struct __attribute__((aligned(4))) GfU4 { uint16_t x, y, z, w; };
struct __attribute__((aligned(2))) GfU4s { uint16_t x, y, z, w; };
template <typename T, bool ALIGNED>
__global__ __launch_bounds__(256) void k_gofloat_cfa_v4(const T *__restrict__ src, size_t owidth, size_t x, size_t y, uint32_t width, uint32_t height,
                                                        float min0, float range0, float *__restrict__ dst) {
  const uint32_t col = 4u * (blockIdx.x * blockDim.x + threadIdx.x);
  if (col >= width) return;                              // width % 4 == 0: a thread's four columns are all inside
  for (uint32_t row = blockIdx.y; row < height; row += gridDim.y) {
    const T *p = src + owidth * (row + y) + x + col;
    float v0, v1, v2, v3;
    if (sizeof(T) == 4) { const GfF4 t = *reinterpret_cast<const GfF4 *>(p); v0 = t.x; v1 = t.y; v2 = t.z; v3 = t.w; }
    else if (ALIGNED) { const GfU4 t = *reinterpret_cast<const GfU4 *>(p); v0 = (float)t.x; v1 = (float)t.y; v2 = (float)t.z; v3 = (float)t.w; }
    else { const GfU4s t = *reinterpret_cast<const GfU4s *>(p); v0 = (float)t.x; v1 = (float)t.y; v2 = (float)t.z; v3 = (float)t.w; }
    GfF4 o;
    o.x = rs_min((v0 - min0) / range0, 1.0f); o.y = rs_min((v1 - min0) / range0, 1.0f);       // gofloat.rs:126 / :162
    o.z = rs_min((v2 - min0) / range0, 1.0f); o.w = rs_min((v3 - min0) / range0, 1.0f);
    *reinterpret_cast<GfF4 *>(dst + (size_t)row * width + col) = o;
  }
}
template <typename T>
__global__ void k_gofloat_mono(const T *__restrict__ src, size_t owidth, size_t x, size_t y, uint32_t width, uint32_t height,
                               float min0, float range0, float4 *__restrict__ dst) {
  const uint32_t col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= width) return;
  for (uint32_t row = blockIdx.y; row < height; row += gridDim.y) {
    const float v = rs_min(((float)src[owidth * (row + y) + x + col] - min0) / range0, 1.0f);   // gofloat.rs:100-104
    dst[(size_t)row * width + col] = make_float4(v, v, v, 0.0f);
  }
}
struct Levels4 { float mins[4]; float ranges[4]; };
template <typename T>
__global__ void k_gofloat_rgb(const T *__restrict__ src, size_t owidth, size_t x, size_t y, uint32_t width, uint32_t height,
                              Levels4 lv, float4 *__restrict__ dst) {
  const uint32_t col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= width) return;
  for (uint32_t row = blockIdx.y; row < height; row += gridDim.y) {
    const T *i = src + (owidth * (row + y) + x + col) * 3;                   // gofloat.rs:113-118
    float4 o;
    o.x = rs_min(((float)i[0] - lv.mins[0]) / lv.ranges[0], 1.0f);
    o.y = rs_min(((float)i[1] - lv.mins[1]) / lv.ranges[1], 1.0f);
    o.z = rs_min(((float)i[2] - lv.mins[2]) / lv.ranges[2], 1.0f);
    o.w = 0.0f;
    dst[(size_t)row * width + col] = o;
  }
}
// run_other, 8 bit: expand_srgb_gamma(input8bit(v)) -- only 256 distinct inputs, so the block first
// evaluates the reference expression for each of them (the SRGB_GAMMA_REVERSE table is read from
// global memory 256 times) and the pixels index that 1 KB LDS table.
__global__ void k_gofloat_other_u8(const uint8_t *__restrict__ src, size_t owidth, size_t x, size_t y, uint32_t width, uint32_t height,
                                   const LutPair *__restrict__ gamma_reverse, float4 *__restrict__ dst) {
  __shared__ float s_expand[256];
  for (int i = threadIdx.x; i < 256; i += blockDim.x) s_expand[i] = lut_interp(gamma_reverse, input8bit((uint8_t)i));
  __syncthreads();
  const uint32_t col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= width) return;
  for (uint32_t row = blockIdx.y; row < height; row += gridDim.y) {
    const uint8_t *i = src + (owidth * (row + y) + x + col) * 3;             // gofloat.rs:181-186
    dst[(size_t)row * width + col] = make_float4(s_expand[i[0]], s_expand[i[1]], s_expand[i[2]], 0.0f);
  }
}
__global__ void k_gofloat_other_u16(const uint16_t *__restrict__ src, size_t owidth, size_t x, size_t y, uint32_t width, uint32_t height,
                                    float4 *__restrict__ dst) {
  const uint32_t col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= width) return;
  for (uint32_t row = blockIdx.y; row < height; row += gridDim.y) {
    const uint16_t *i = src + (owidth * (row + y) + x + col) * 3;            // gofloat.rs:191-196
    dst[(size_t)row * width + col] = make_float4(input16bit(i[0]), input16bit(i[1]), input16bit(i[2]), 0.0f);
  }
}

template <typename T>
void launch_gofloat_cfa(const T *src, size_t owidth, size_t x, size_t y, size_t w, size_t h, float black0, float white0,
                        float *dst, hipStream_t s) {
  if ((w & 3) == 0 && (reinterpret_cast<uintptr_t>(dst) & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 3) == 0) {
    const dim3 grid((unsigned)((w / 4 + 255) / 256), (unsigned)(h < 4096 ? h : 4096), 1);
    // u16: the 8-byte load wants the first sample of every row on a dword boundary
    if (sizeof(T) == 4 || ((owidth & 1) == 0 && (x & 1) == 0))
      hipLaunchKernelGGL((k_gofloat_cfa_v4<T, true>), grid, dim3(256), 0, s, src, owidth, x, y, (uint32_t)w, (uint32_t)h, black0, white0 - black0, dst);
    else
      hipLaunchKernelGGL((k_gofloat_cfa_v4<T, false>), grid, dim3(256), 0, s, src, owidth, x, y, (uint32_t)w, (uint32_t)h, black0, white0 - black0, dst);
    return;
  }
  hipLaunchKernelGGL(k_gofloat_cfa<T>, grid_rows(w, h, 256), dim3(256), 0, s, src, owidth, x, y, (uint32_t)w, (uint32_t)h,
                     black0, white0 - black0, dst);
}
template <typename T>
void launch_gofloat_mono(const T *src, size_t owidth, size_t x, size_t y, size_t w, size_t h, float black0, float white0,
                         float *dst4, hipStream_t s) {
  hipLaunchKernelGGL(k_gofloat_mono<T>, grid_rows(w, h, 256), dim3(256), 0, s, src, owidth, x, y, (uint32_t)w, (uint32_t)h,
                     black0, white0 - black0, reinterpret_cast<float4 *>(dst4));
}
template <typename T>
void launch_gofloat_rgb(const T *src, size_t owidth, size_t x, size_t y, size_t w, size_t h, const float *black4, const float *white4,
                        float *dst4, hipStream_t s) {
  Levels4 lv;
  for (int i = 0; i < 4; ++i) { lv.mins[i] = black4[i]; lv.ranges[i] = white4[i] - black4[i]; }   // gofloat.rs:86-89
  hipLaunchKernelGGL(k_gofloat_rgb<T>, grid_rows(w, h, 256), dim3(256), 0, s, src, owidth, x, y, (uint32_t)w, (uint32_t)h,
                     lv, reinterpret_cast<float4 *>(dst4));
}
template void launch_gofloat_cfa<uint16_t>(const uint16_t *, size_t, size_t, size_t, size_t, size_t, float, float, float *, hipStream_t);
template void launch_gofloat_cfa<float>(const float *, size_t, size_t, size_t, size_t, size_t, float, float, float *, hipStream_t);
template void launch_gofloat_mono<uint16_t>(const uint16_t *, size_t, size_t, size_t, size_t, size_t, float, float, float *, hipStream_t);
template void launch_gofloat_mono<float>(const float *, size_t, size_t, size_t, size_t, size_t, float, float, float *, hipStream_t);
template void launch_gofloat_rgb<uint16_t>(const uint16_t *, size_t, size_t, size_t, size_t, size_t, const float *, const float *, float *, hipStream_t);
template void launch_gofloat_rgb<float>(const float *, size_t, size_t, size_t, size_t, size_t, const float *, const float *, float *, hipStream_t);

void launch_gofloat_other_u8(const uint8_t *src, size_t owidth, size_t x, size_t y, size_t w, size_t h,
                             const void *gamma_reverse_pairs, float *dst4, hipStream_t s) {
  hipLaunchKernelGGL(k_gofloat_other_u8, grid_rows(w, h, 256), dim3(256), 0, s, src, owidth, x, y, (uint32_t)w, (uint32_t)h,
                     reinterpret_cast<const LutPair *>(gamma_reverse_pairs), reinterpret_cast<float4 *>(dst4));
}
void launch_gofloat_other_u16(const uint16_t *src, size_t owidth, size_t x, size_t y, size_t w, size_t h, float *dst4, hipStream_t s) {
  hipLaunchKernelGGL(k_gofloat_other_u16, grid_rows(w, h, 256), dim3(256), 0, s, src, owidth, x, y, (uint32_t)w, (uint32_t)h,
                     reinterpret_cast<float4 *>(dst4));
}

// ------------------------------------------------------------------------------------------
// demosaic::full, any CFA (src/ops/demosaic.rs:67-119)
// `lookups` = the 48x48 table of nine 3-bit tap colours built on the host exactly as
// demosaic.rs:77-90 does; staged in LDS (9 KB).
// ------------------------------------------------------------------------------------------
struct DemosaicAcc {
  float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;     // sums[0..3]; bucket 4 (discard) is never read
  float n0 = 0.0f, n1 = 0.0f, n2 = 0.0f, n3 = 0.0f;     // counts
  __device__ __forceinline__ void add(uint32_t color, float v) {   // demosaic.rs:105-106
    if (color == 0) { s0 += v; n0 += 1.0f; }
    else if (color == 1) { s1 += v; n1 += 1.0f; }
    else if (color == 2) { s2 += v; n2 += 1.0f; }
    else if (color == 3) { s3 += v; n3 += 1.0f; }
  }
  __device__ __forceinline__ float4 finish() const {               // demosaic.rs:110-114; untouched channels stay 0.0
    return make_float4(n0 > 0.0f ? s0 / n0 : 0.0f, n1 > 0.0f ? s1 / n1 : 0.0f,
                       n2 > 0.0f ? s2 / n2 : 0.0f, n3 > 0.0f ? s3 / n3 : 0.0f);
  }
};

__global__ void k_demosaic_full(const float *__restrict__ src, uint32_t width, uint32_t img_height, uint32_t src_row0,
                                uint32_t out_row0, uint32_t out_rows, const uint32_t *__restrict__ lookups,
                                float4 *__restrict__ dst) {
  __shared__ uint32_t s_lookups[48 * 48];
  for (int i = threadIdx.x; i < 48 * 48; i += blockDim.x) s_lookups[i] = lookups[i];
  __syncthreads();
  const uint32_t col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= width) return;
  for (uint32_t orow = blockIdx.y; orow < out_rows; orow += gridDim.y) {
    const uint32_t row = out_row0 + orow;                                   // image row
    const uint32_t colors = s_lookups[(row % 48) * 48 + (col % 48)];        // demosaic.rs:95
    DemosaicAcc acc;
    #pragma unroll
    for (int i = 0; i < 9; ++i) {                                           // tap order of demosaic.rs:70-74
      const int dy = i / 3 - 1, dx = i % 3 - 1;
      const int64_t r = (int64_t)row + dy, c = (int64_t)col + dx;
      if (r >= 0 && r < (int64_t)img_height && c >= 0 && c < (int64_t)width)
        acc.add((colors >> (3 * i)) & 7u, src[(size_t)(r - src_row0) * width + (size_t)c]);
    }
    dst[(size_t)orow * width + col] = acc.finish();
  }
}
void launch_demosaic_full(const float *src, size_t width, size_t img_height, size_t src_row0, size_t out_row0, size_t out_rows,
                          const uint32_t *lookups_dev, float *dst4, hipStream_t s) {
  hipLaunchKernelGGL(k_demosaic_full, grid_rows(width, out_rows, 256), dim3(256), 0, s, src, (uint32_t)width, (uint32_t)img_height,
                     (uint32_t)src_row0, (uint32_t)out_row0, (uint32_t)out_rows, lookups_dev, reinterpret_cast<float4 *>(dst4));
}

// ------------------------------------------------------------------------------------------
// scaling::transform_buffer<T> (src/scaling.rs:51-130)
// One thread per destination pixel; the window walk keeps the reference's y-outer / x-inner /
// component-innermost accumulation order.
// ------------------------------------------------------------------------------------------
struct TransformArgs {
  uint32_t width, height, nwidth, nheight, components;
  float tlx, tly;                 // topleft as f32
  float skip_x_x, skip_x_y, skip_y_x, skip_y_y;
  float inv_skip_x_x, inv_skip_y_y;   // RN(1/skip) for cdiv_fast
  int fast_x, fast_y;                 // host validated the multiply/fma division for that divisor
  int has_cfa;
  // optional fused OpGoFloat (CFA branch, gofloat.rs:122-130/158-166): the source is the raw sensor frame
  int norm; float min0, range0; uint64_t src_pitch, src_x, src_y;
  int norm_fast; float inv_range0;    // fused OpGoFloat: the host validated cdiv_fast for range0 (else IEEE division)
  int norm_light;                     // f32 sources: 2^-70 <= |black| <= 2^70, so a nonzero v - black is at least half an ulp of black (>= 2^-95):
                                      // no dividend can be tiny, and the per-row guard shrinks to one comparison on the row's minimum
  // output-row band [out_r0, out_r1) of the fused gofloat + scaled_demosaic kernels (multi-GPU sharding of one frame):
  // dst row 0 = output row out_r0; the source slab's first row is folded into src_y (which may wrap: pointer arithmetic mod 2^64)
  uint32_t out_r0, out_r1;
  // k_raw_scaled_demosaic_w8m: a one-dimensional launch of xcd_gx x xcd_gy blocks (xcd_gy a multiple of 8) laid out so that each XCD works on one
  // contiguous eighth of the block rows (see the kernel); 0 = the plain two-dimensional grid
  uint32_t xcd_gx, xcd_gy, xcd_group;
  // k_transform_buffer: 1 when the transform is scale_down_buffer's (corner (0, 0), no cross terms, both skips >= 1, frame sides below 2^24): sample offsets and
  // centres are then sums of terms that are zero or multiples of 2^-24, so a nonzero difference lies in [2^-24, 2^25] -- inside the multiply-fma
  // division's proven zone without tb_div's per-tap exponent test and its divergent IEEE fallback
  int plain_axis;
};
// (v - center) / skip  (scaling.rs:104-105): cdiv_fast when the host validated the divisor and the dividend is in the proven
// zone, the IEEE division otherwise (zero or negative skips of degenerate / rotated transforms, absurd centres)
__device__ __forceinline__ float tb_div(float d, float skip, float inv, int fast) {
  if (fast && !cdiv_guard(d)) return cdiv_fast(d, skip, inv);
  return d / skip;
}
template <typename T> struct PixCast;
template <> struct PixCast<float>    { static __device__ __forceinline__ float to(float v) { return v; }    static __device__ __forceinline__ float from(float f) { return f; } };
template <> struct PixCast<uint8_t>  { static __device__ __forceinline__ float to(uint8_t v) { return (float)v; }  static __device__ __forceinline__ uint8_t from(float f) { float c = rs_min(rs_max(f, 0.0f), 255.0f); return (uint8_t)f32_as_u32_sat(c); } };
template <> struct PixCast<uint16_t> { static __device__ __forceinline__ float to(uint16_t v) { return (float)v; } static __device__ __forceinline__ uint16_t from(float f) { float c = rs_min(rs_max(f, 0.0f), 65535.0f); return (uint16_t)f32_as_u32_sat(c); } };

template <typename T>
__global__ void k_transform_buffer(const T *__restrict__ src, TransformArgs a, const uint8_t *__restrict__ cfa48, T *__restrict__ dst) {
  __shared__ uint8_t s_cfa[48 * 48];
  if (a.has_cfa) {
    for (int i = threadIdx.x; i < 48 * 48; i += blockDim.x) s_cfa[i] = cfa48[i];
    __syncthreads();
  }
  const uint32_t col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= a.nwidth) return;
  constexpr bool RGB3 = sizeof(T) <= 2;                                     // the integer instantiations are the raster fast path's
  const bool plain = a.plain_axis != 0;
  for (uint32_t row = blockIdx.y; row < a.nheight; row += gridDim.y) {
    // per-row values (scaling.rs:77-82)
    const float from_x_r = a.tlx + a.skip_y_x * (float)row;
    const float to_x_r = a.tlx + a.skip_y_x * (float)(row + 1);
    const float from_y_r = a.tly + a.skip_y_y * (float)row;
    const float to_y_r = a.tly + a.skip_y_y * (float)(row + 1);
    const float center_x_r = a.tlx + (a.skip_y_x * (float)row) + (a.skip_y_x / 2.0f) - 0.5f;
    const float center_y_r = a.tly + (a.skip_y_y * (float)row) + (a.skip_y_y / 2.0f) - 0.5f;
    // per-column window (scaling.rs:84-89)
    const uint32_t from_x = min(a.width - 1, f32_as_u32_sat(floorf(from_x_r + (a.skip_x_x * (float)col))));
    const uint32_t to_x = min(a.width - 1, f32_as_u32_sat(floorf(to_x_r + (a.skip_x_x * (float)(col + 1)))));
    const uint32_t from_y = min(a.height - 1, f32_as_u32_sat(floorf(from_y_r + (a.skip_x_y * (float)col))));
    const uint32_t to_y = min(a.height - 1, f32_as_u32_sat(floorf(to_y_r + (a.skip_x_y * (float)(col + 1)))));
    const float center_x = center_x_r + (a.skip_x_x * (float)col) + (a.skip_x_x / 2.0f);
    const float center_y = center_y_r + (a.skip_x_y * (float)col) + (a.skip_x_y / 2.0f);

    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f, n0 = 0.0f, n1 = 0.0f, n2 = 0.0f, n3 = 0.0f;
    // The thumbnail path of output_8bit / output_16bit (scale_down_srgb / 16, src/scaling.rs:162-182): a raster, three samples per pixel, windows of at most
    // eight columns (scale <= 7).  One window row = eight pixel loads issued together (one access per pixel: 4 bytes for RGB8, 8 for RGB16; the last pixel
    // of the frame reads the bytes in FRONT of it and shifts) -- the tap-by-tap loop below waits out one memory round trip per tap.  Columns outside the
    // lane's window carry the weight 1 - dx dx = -inf, which the clamp turns into a factor of 0: 0 * sample and + 0 change nothing in sums that start at +0.0.
    if (RGB3 && plain && a.components == 3u && !a.has_cfa && (size_t)a.width * a.height > 1 && __builtin_amdgcn_ballot_w64(to_x - from_x >= 8u) == 0) {   // (a 1x1 source has no byte in front of its last pixel: tap loop)
      float ax[8];
      #pragma unroll
      for (uint32_t k = 0; k < 8; ++k) {
        const float dxv = (float)(from_x + k) - center_x;
        const float delta_x = a.fast_x ? cdiv_fast(dxv, a.skip_x_x, a.inv_skip_x_x) : dxv / a.skip_x_x;
        ax[k] = (from_x + k <= to_x) ? 1.0f - (delta_x * delta_x) : -__builtin_inff();
      }
      const size_t last_px = (size_t)a.width * a.height - 1;
      for (uint32_t y = from_y; y <= to_y; ++y) {
        const float dyv = (float)y - center_y;
        const float delta_y = a.fast_y ? cdiv_fast(dyv, a.skip_y_y, a.inv_skip_y_y) : dyv / a.skip_y_y;
        const float dy2 = delta_y * delta_y;
        float c0[8], c1[8], c2[8];
        #pragma unroll
        for (uint32_t k = 0; k < 8; ++k) {
          const size_t pi = (size_t)y * a.width + min(from_x + k, a.width - 1);
          const bool tail = pi == last_px && last_px > 0;
          if (sizeof(T) == 1) {
            struct __attribute__((packed, aligned(1))) W4 { uint32_t w; };
            uint32_t w = reinterpret_cast<const W4 *>(reinterpret_cast<const uint8_t *>(src) + pi * 3 - (tail ? 1 : 0))->w;
            w = tail ? w >> 8 : w;
            c0[k] = (float)(w & 0xFFu); c1[k] = (float)((w >> 8) & 0xFFu); c2[k] = (float)((w >> 16) & 0xFFu);
          } else {
            struct __attribute__((packed, aligned(2))) W8 { uint32_t lo, hi; };
            const W8 w = *reinterpret_cast<const W8 *>(reinterpret_cast<const uint16_t *>(src) + pi * 3 - (tail ? 1 : 0));
            const unsigned long long q = (((unsigned long long)w.hi << 32) | w.lo) >> (tail ? 16 : 0);
            c0[k] = (float)((uint32_t)q & 0xFFFFu); c1[k] = (float)((uint32_t)(q >> 16) & 0xFFFFu); c2[k] = (float)((uint32_t)(q >> 32) & 0xFFFFu);
          }
        }
        #pragma unroll
        for (uint32_t k = 0; k < 8; ++k) {
          float factor = ax[k] - dy2;                                         // 1.0 - dx dx - dy dy, left to right (scaling.rs:106)
          factor = (factor < 0.0f) ? 0.0f : factor;
          s0 += c0[k] * factor; n0 += factor; s1 += c1[k] * factor; n1 += factor; s2 += c2[k] * factor; n2 += factor;
        }
        if (y == 0xFFFFFFFFu) break;
      }
      T *o = dst + ((size_t)row * a.nwidth + col) * 3;
      o[0] = (n0 > 0.0f) ? PixCast<T>::from(s0 / n0) : PixCast<T>::from(0.0f);
      o[1] = (n1 > 0.0f) ? PixCast<T>::from(s1 / n1) : PixCast<T>::from(0.0f);
      o[2] = (n2 > 0.0f) ? PixCast<T>::from(s2 / n2) : PixCast<T>::from(0.0f);
      continue;
    }
    const uint32_t xm0 = from_x % 48;                                       // cfa.color_at(y, x) = pattern[y%48][x%48], kept incrementally
    uint32_t ym48 = (from_y % 48) * 48;
    for (uint32_t y = from_y; y <= to_y; ++y) {
      const float dyv = (float)y - center_y;
      const float delta_y = plain ? (a.fast_y ? cdiv_fast(dyv, a.skip_y_y, a.inv_skip_y_y) : dyv / a.skip_y_y) : tb_div(dyv, a.skip_y_y, a.inv_skip_y_y, a.fast_y);
      const float dy2 = delta_y * delta_y;
      uint32_t xm = xm0;
      for (uint32_t x = from_x; x <= to_x; ++x) {
        const float dxv = (float)x - center_x;
        const float delta_x = plain ? (a.fast_x ? cdiv_fast(dxv, a.skip_x_x, a.inv_skip_x_x) : dxv / a.skip_x_x) : tb_div(dxv, a.skip_x_x, a.inv_skip_x_x, a.fast_x);
        float factor = 1.0f - (delta_x * delta_x) - dy2;                    // scaling.rs:106
        factor = (factor < 0.0f) ? 0.0f : factor;
        if (RGB3 && plain && a.components == 3u && !a.has_cfa) {
          // a raster pixel's three samples in one access (scale_down_srgb / scale_down_srgb16: the thumbnail path of output_8bit / output_16bit) instead
          // of three one-sample loads; the frame's very last pixel keeps them (the wide load would read past the buffer)
          const size_t pi = (size_t)y * a.width + x;
          float c0, c1, c2;
          if (pi + 1 < (size_t)a.width * a.height) {
            if (sizeof(T) == 1) {
              struct __attribute__((packed, aligned(1))) W4 { uint32_t w; };
              const uint32_t w = reinterpret_cast<const W4 *>(reinterpret_cast<const uint8_t *>(src) + pi * 3)->w;
              c0 = (float)(w & 0xFFu); c1 = (float)((w >> 8) & 0xFFu); c2 = (float)((w >> 16) & 0xFFu);
            } else {
              struct __attribute__((packed, aligned(2))) W8 { uint32_t lo, hi; };
              const W8 w = *reinterpret_cast<const W8 *>(reinterpret_cast<const uint16_t *>(src) + pi * 3);
              c0 = (float)(w.lo & 0xFFFFu); c1 = (float)(w.lo >> 16); c2 = (float)(w.hi & 0xFFFFu);
            }
          } else { const T *p = src + pi * 3; c0 = PixCast<T>::to(p[0]); c1 = PixCast<T>::to(p[1]); c2 = PixCast<T>::to(p[2]); }
          s0 += c0 * factor; n0 += factor; s1 += c1 * factor; n1 += factor; s2 += c2 * factor; n2 += factor;
        } else
        if (a.has_cfa) {
          const uint32_t c = s_cfa[ym48 + xm];                              // cfa.color_at(y, x)
          xm = (xm == 47) ? 0 : xm + 1;
          const float t = PixCast<T>::to(src[(size_t)y * a.width + x]) * factor;
          if (c == 0) { s0 += t; n0 += factor; }
          else if (c == 1) { s1 += t; n1 += factor; }
          else if (c == 2) { s2 += t; n2 += factor; }
          else { s3 += t; n3 += factor; }
        } else {
          const T *p = src + ((size_t)y * a.width + x) * a.components;
          s0 += PixCast<T>::to(p[0]) * factor; n0 += factor;
          if (a.components > 1) { s1 += PixCast<T>::to(p[1]) * factor; n1 += factor; }
          if (a.components > 2) { s2 += PixCast<T>::to(p[2]) * factor; n2 += factor; }
          if (a.components > 3) { s3 += PixCast<T>::to(p[3]) * factor; n3 += factor; }
        }
        if (x == 0xFFFFFFFFu) break;
      }
      ym48 = (ym48 == 47 * 48) ? 0 : ym48 + 48;
      if (y == 0xFFFFFFFFu) break;
    }
    // scaling.rs:122-126: untouched components keep the zero fill
    T *o = dst + ((size_t)row * a.nwidth + col) * a.components;
    o[0] = (n0 > 0.0f) ? PixCast<T>::from(s0 / n0) : PixCast<T>::from(0.0f);
    if (a.components > 1) o[1] = (n1 > 0.0f) ? PixCast<T>::from(s1 / n1) : PixCast<T>::from(0.0f);
    if (a.components > 2) o[2] = (n2 > 0.0f) ? PixCast<T>::from(s2 / n2) : PixCast<T>::from(0.0f);
    if (a.components > 3) o[3] = (n3 > 0.0f) ? PixCast<T>::from(s3 / n3) : PixCast<T>::from(0.0f);
  }
}
// Host check of a runtime divisor for cdiv_fast (same obligations as ipk_api.cpp's validate_cdiv_for_range): positive,
// ordinary magnitude, and the three-step quotient equal to the IEEE one for EVERY dividend mantissa (exhaustive per divisor,
// memoised: ipk_host.hpp cdiv_mantissa_exhaustive_ok).  The dividends here are pixel offsets, far inside the exponent zone.
static int cdiv_host_ok(float c) { return cdiv_mantissa_exhaustive_ok(c) ? 1 : 0; }

template <typename T>
void launch_transform_buffer(const T *src, size_t width, size_t height, int64_t tlx, int64_t tly, int64_t trx, int64_t try_,
                             int64_t blx, int64_t bly, size_t nwidth, size_t nheight, size_t components,
                             const uint8_t *cfa48_dev, T *dst, hipStream_t s) {
  TransformArgs a{};
  a.width = (uint32_t)width; a.height = (uint32_t)height; a.nwidth = (uint32_t)nwidth; a.nheight = (uint32_t)nheight;
  a.components = (uint32_t)components;
  a.tlx = (float)tlx; a.tly = (float)tly;
  // scaling.rs:68-71
  a.skip_x_x = ((float)trx - (float)tlx) / ((float)(nwidth - 1));
  a.skip_x_y = ((float)try_ - (float)tly) / ((float)(nwidth - 1));
  a.skip_y_x = ((float)blx - (float)tlx) / ((float)(nheight - 1));
  a.skip_y_y = ((float)bly - (float)tly) / ((float)(nheight - 1));
  a.inv_skip_x_x = 1.0f / a.skip_x_x; a.inv_skip_y_y = 1.0f / a.skip_y_y;
  a.fast_x = cdiv_host_ok(a.skip_x_x); a.fast_y = cdiv_host_ok(a.skip_y_y);
  a.has_cfa = cfa48_dev != nullptr;
  a.norm = 0; a.min0 = 0.0f; a.range0 = 1.0f; a.src_pitch = width; a.src_x = 0; a.src_y = 0; a.norm_fast = 0; a.inv_range0 = 1.0f;
  a.plain_axis = (tlx == 0 && tly == 0 && a.skip_x_y == 0.0f && a.skip_y_x == 0.0f && a.skip_x_x >= 1.0f && a.skip_y_y >= 1.0f &&
                  a.skip_x_x <= 0x1p24f && a.skip_y_y <= 0x1p24f && width < (size_t(1) << 24) && height < (size_t(1) << 24)) ? 1 : 0;
  hipLaunchKernelGGL(k_transform_buffer<T>, grid_rows_few(nwidth, nheight, 128, 8192), dim3(128), 0, s, src, a, cfa48_dev, dst);
}

// OpGoFloat::run_other + scaling::scale_down_opbuf in one pass over an RGB8 / RGB16 raster: dst4 = scale_down_opbuf(run_other(src))
// (src/ops/gofloat.rs:171-201 + src/scaling.rs:147-160), what OpDemosaic::run does with a raster source under a size limit
// (demosaic.rs:44-46) -- the full-size 4-channel f32 buffer (16 B/px) never exists.  One thread per output pixel, the window
// walked y-outer / x-inner like the reference; the three colour components share their weights (the reference keeps one
// count per component, all fed the same factors), E is 0 * factor summed = +0.0 over a positive count = +0.0.
// RGB8 samples go through the 256-entry expand_srgb_gamma(input8bit(i)) table built per block; RGB16 samples through
// input16bit = v / 65535 as the proven three-step division (dividends are the integers 0..65535).
template <typename SrcT>
__global__ void k_raster_scale_down(const SrcT *__restrict__ src, TransformArgs a, const LutPair *__restrict__ gamma_reverse, float4 *__restrict__ dst) {
  __shared__ float s_expand[sizeof(SrcT) == 1 ? 256 : 4];
  if (sizeof(SrcT) == 1) {
    for (int i = threadIdx.x; i < 256; i += blockDim.x) s_expand[i] = lut_interp(gamma_reverse, input8bit((uint8_t)i));
    __syncthreads();
  }
  const uint32_t col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= a.nwidth) return;
  for (uint32_t row = blockIdx.y; row < a.nheight; row += gridDim.y) {
    const float from_x_r = a.tlx + a.skip_y_x * (float)row;
    const float to_x_r = a.tlx + a.skip_y_x * (float)(row + 1);
    const float from_y_r = a.tly + a.skip_y_y * (float)row;
    const float to_y_r = a.tly + a.skip_y_y * (float)(row + 1);
    const float center_x_r = a.tlx + (a.skip_y_x * (float)row) + (a.skip_y_x / 2.0f) - 0.5f;
    const float center_y_r = a.tly + (a.skip_y_y * (float)row) + (a.skip_y_y / 2.0f) - 0.5f;
    const uint32_t from_x = min(a.width - 1, f32_as_u32_sat(floorf(from_x_r + (a.skip_x_x * (float)col))));
    const uint32_t to_x = min(a.width - 1, f32_as_u32_sat(floorf(to_x_r + (a.skip_x_x * (float)(col + 1)))));
    const uint32_t from_y = min(a.height - 1, f32_as_u32_sat(floorf(from_y_r + (a.skip_x_y * (float)col))));
    const uint32_t to_y = min(a.height - 1, f32_as_u32_sat(floorf(to_y_r + (a.skip_x_y * (float)(col + 1)))));
    const float center_x = center_x_r + (a.skip_x_x * (float)col) + (a.skip_x_x / 2.0f);
    const float center_y = center_y_r + (a.skip_x_y * (float)col) + (a.skip_x_y / 2.0f);
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, n = 0.0f;
    // (round 5: the eight-loads-per-window-row form of k_transform_buffer's raster path measured here: 51.8 -> 62.7 us (u8) / 58.2 -> 71.0 (u16) with all eight
    // columns evaluated, 68 us with the columns cut to the wave's widest window -- a tap costs three table reads or nine instructions here, not three
    // multiply-adds, and the out-of-window ones are not free.  Not kept.)
    for (uint32_t y = from_y; y <= to_y; ++y) {
      const float delta_y = tb_div((float)y - center_y, a.skip_y_y, a.inv_skip_y_y, a.fast_y);
      const float dy2 = delta_y * delta_y;
      const SrcT *rowp = src + ((size_t)(y + a.src_y) * a.src_pitch + a.src_x) * 3;
      for (uint32_t x = from_x; x <= to_x; ++x) {
        const float delta_x = tb_div((float)x - center_x, a.skip_x_x, a.inv_skip_x_x, a.fast_x);
        float factor = 1.0f - (delta_x * delta_x) - dy2;                    // scaling.rs:106
        factor = (factor < 0.0f) ? 0.0f : factor;
        const SrcT *p = rowp + (size_t)x * 3;
        float v0, v1, v2;
        if (sizeof(SrcT) == 1) { v0 = s_expand[p[0]]; v1 = s_expand[p[1]]; v2 = s_expand[p[2]]; }
        else {
          constexpr float c = 65535.0f, rc = 1.0f / 65535.0f;
          const float x0 = (float)p[0], x1 = (float)p[1], x2 = (float)p[2];
          const float q0 = x0 * rc, q1 = x1 * rc, q2 = x2 * rc;
          v0 = __builtin_fmaf(__builtin_fmaf(-q0, c, x0), rc, q0); v1 = __builtin_fmaf(__builtin_fmaf(-q1, c, x1), rc, q1); v2 = __builtin_fmaf(__builtin_fmaf(-q2, c, x2), rc, q2);
        }
        s0 += v0 * factor; s1 += v1 * factor; s2 += v2 * factor; n += factor;
        if (x == 0xFFFFFFFFu) break;
      }
      if (y == 0xFFFFFFFFu) break;
    }
    float4 o = make_float4(0.0f, 0.0f, 0.0f, 0.0f);                          // scaling.rs:122-126: no weight, zero fill
    if (n > 0.0f) { o.x = s0 / n; o.y = s1 / n; o.z = s2 / n; }
    dst[(size_t)row * a.nwidth + col] = o;
  }
}
static int cdiv_host_ok(float c);
void launch_raster_scale_down(const void *src, int src_is_u16, size_t owidth, size_t x, size_t y, size_t width, size_t height,
                              size_t nwidth, size_t nheight, const void *gamma_reverse_pairs, float *dst4, hipStream_t s) {
  TransformArgs a{};
  a.width = (uint32_t)width; a.height = (uint32_t)height; a.nwidth = (uint32_t)nwidth; a.nheight = (uint32_t)nheight; a.components = 4;
  a.tlx = 0.0f; a.tly = 0.0f;                                               // scale_down_buffer's corners (scaling.rs:35-48)
  a.skip_x_x = ((float)((int64_t)width - 1) - 0.0f) / ((float)(nwidth - 1));
  a.skip_x_y = (0.0f - 0.0f) / ((float)(nwidth - 1));
  a.skip_y_x = (0.0f - 0.0f) / ((float)(nheight - 1));
  a.skip_y_y = ((float)((int64_t)height - 1) - 0.0f) / ((float)(nheight - 1));
  a.inv_skip_x_x = 1.0f / a.skip_x_x; a.inv_skip_y_y = 1.0f / a.skip_y_y;
  a.fast_x = cdiv_host_ok(a.skip_x_x); a.fast_y = cdiv_host_ok(a.skip_y_y);
  a.src_pitch = owidth; a.src_x = x; a.src_y = y;
  const LutPair *gr = reinterpret_cast<const LutPair *>(gamma_reverse_pairs);
  if (src_is_u16) hipLaunchKernelGGL(k_raster_scale_down<uint16_t>, grid_rows_few(nwidth, nheight, 128, 8192), dim3(128), 0, s, static_cast<const uint16_t *>(src), a, gr, reinterpret_cast<float4 *>(dst4));
  else hipLaunchKernelGGL(k_raster_scale_down<uint8_t>, grid_rows_few(nwidth, nheight, 128, 8192), dim3(128), 0, s, static_cast<const uint8_t *>(src), a, gr, reinterpret_cast<float4 *>(dst4));
}

// OpGoFloat (CFA branch) + scaling::scaled_demosaic in one pass over the raw sensor frame: dst4 = scaled_demosaic(gofloat(src)).
// T = uint16_t or float sensor samples; writes f32 RGBE.  (src/ops/gofloat.rs:122-130,158-166 + src/scaling.rs:132-145)
template <typename T>
__global__ void k_raw_scaled_demosaic(const T *__restrict__ src, TransformArgs a, const uint8_t *__restrict__ cfa48, float *__restrict__ dst) {
  __shared__ uint8_t s_cfa[48 * 48];
  for (int i = threadIdx.x; i < 48 * 48; i += blockDim.x) s_cfa[i] = cfa48[i];
  __syncthreads();
  const uint32_t col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= a.nwidth) return;
  for (uint32_t row = a.out_r0 + blockIdx.y; row < a.out_r1; row += gridDim.y) {
    const float from_x_r = a.tlx + a.skip_y_x * (float)row;
    const float to_x_r = a.tlx + a.skip_y_x * (float)(row + 1);
    const float from_y_r = a.tly + a.skip_y_y * (float)row;
    const float to_y_r = a.tly + a.skip_y_y * (float)(row + 1);
    const float center_x_r = a.tlx + (a.skip_y_x * (float)row) + (a.skip_y_x / 2.0f) - 0.5f;
    const float center_y_r = a.tly + (a.skip_y_y * (float)row) + (a.skip_y_y / 2.0f) - 0.5f;
    const uint32_t from_x = min(a.width - 1, f32_as_u32_sat(floorf(from_x_r + (a.skip_x_x * (float)col))));
    const uint32_t to_x = min(a.width - 1, f32_as_u32_sat(floorf(to_x_r + (a.skip_x_x * (float)(col + 1)))));
    const uint32_t from_y = min(a.height - 1, f32_as_u32_sat(floorf(from_y_r + (a.skip_x_y * (float)col))));
    const uint32_t to_y = min(a.height - 1, f32_as_u32_sat(floorf(to_y_r + (a.skip_x_y * (float)(col + 1)))));
    const float center_x = center_x_r + (a.skip_x_x * (float)col) + (a.skip_x_x / 2.0f);
    const float center_y = center_y_r + (a.skip_x_y * (float)col) + (a.skip_x_y / 2.0f);
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f, n0 = 0.0f, n1 = 0.0f, n2 = 0.0f, n3 = 0.0f;
    const uint32_t xm0 = from_x % 48;
    uint32_t ym48 = (from_y % 48) * 48;
    for (uint32_t y = from_y; y <= to_y; ++y) {
      const float delta_y = tb_div((float)y - center_y, a.skip_y_y, a.inv_skip_y_y, a.fast_y);
      const float dy2 = delta_y * delta_y;
      uint32_t xm = xm0;
      const T *rowp = src + (size_t)((uint64_t)y + a.src_y) * a.src_pitch + a.src_x;
      for (uint32_t x = from_x; x <= to_x; ++x) {
        const float delta_x = tb_div((float)x - center_x, a.skip_x_x, a.inv_skip_x_x, a.fast_x);
        float factor = 1.0f - (delta_x * delta_x) - dy2;
        factor = (factor < 0.0f) ? 0.0f : factor;
        const uint32_t c = s_cfa[ym48 + xm];
        xm = (xm == 47) ? 0 : xm + 1;
        const float sv = rs_min(((float)rowp[x] - a.min0) / a.range0, 1.0f);       // gofloat.rs:126
        const float t = sv * factor;
        if (c == 0) { s0 += t; n0 += factor; }
        else if (c == 1) { s1 += t; n1 += factor; }
        else if (c == 2) { s2 += t; n2 += factor; }
        else { s3 += t; n3 += factor; }
        if (x == 0xFFFFFFFFu) break;
      }
      ym48 = (ym48 == 47 * 48) ? 0 : ym48 + 48;
      if (y == 0xFFFFFFFFu) break;
    }
    float4 o;
    o.x = (n0 > 0.0f) ? s0 / n0 : 0.0f; o.y = (n1 > 0.0f) ? s1 / n1 : 0.0f; o.z = (n2 > 0.0f) ? s2 / n2 : 0.0f; o.w = (n3 > 0.0f) ? s3 / n3 : 0.0f;
    reinterpret_cast<float4 *>(dst)[(size_t)(row - a.out_r0) * a.nwidth + col] = o;
  }
}
// The same for windows of at most 8 x 8 source samples (every scale up to 7, i.e. all previews larger than 1/7 size).
// One lane per output pixel as before, but a source row of the window arrives as ONE 8-sample vector load per lane
// (element-aligned dwordx4 pairs / one dwordx4 for u16) instead of one dependent 4-byte load per tap -- the per-tap loads
// at a 16-byte lane stride were what held the kernel (21 L1 accesses per load instruction) -- the column part of the weight
// (1 - dx*dx; scaling.rs:104-106 evaluates (1 - dx*dx) - dy*dy) is computed once per pixel, the colours of the 8 columns come
// from one 16-bit LDS read per row, and the colour bins are filled by selects (a non-matching bin adds +0.0, which cannot
// change a sum that started at +0.0).  Same taps, same order (y outer, x inner), same arithmetic per tap.
template <typename T> struct Row8;
struct __attribute__((packed, aligned(4))) Row8F4 { float x, y, z, w; };
struct __attribute__((packed, aligned(2))) Row8U4 { uint32_t x, y, z, w; };
// load(): eight samples as floats.  issue() / expand() / touch(): the same in two steps for software-pipelined loops -- issue() only
// loads (raw registers), touch() makes the raw registers a use (the s_waitcnt lands there), expand() converts.
template <> struct Row8<float> {
  struct Raw { Row8F4 lo, hi; };
  static __device__ __forceinline__ Raw issue(const float *p) { return Raw{*reinterpret_cast<const Row8F4 *>(p), *reinterpret_cast<const Row8F4 *>(p + 4)}; }
  static __device__ __forceinline__ void touch(const Raw &r) {
    asm volatile("" ::"v"(r.lo.x), "v"(r.lo.y), "v"(r.lo.z), "v"(r.lo.w), "v"(r.hi.x), "v"(r.hi.y), "v"(r.hi.z), "v"(r.hi.w));
  }
  static __device__ __forceinline__ void expand(const Raw &r, float d[8]) {
    d[0] = r.lo.x; d[1] = r.lo.y; d[2] = r.lo.z; d[3] = r.lo.w; d[4] = r.hi.x; d[5] = r.hi.y; d[6] = r.hi.z; d[7] = r.hi.w;
  }
  static __device__ __forceinline__ void load(const float *p, float d[8]) { expand(issue(p), d); }
};
template <> struct Row8<uint16_t> {
  typedef Row8U4 Raw;
  static __device__ __forceinline__ Raw issue(const uint16_t *p) { return *reinterpret_cast<const Row8U4 *>(p); }
  static __device__ __forceinline__ void touch(const Raw &r) { asm volatile("" ::"v"(r.x), "v"(r.y), "v"(r.z), "v"(r.w)); }
  static __device__ __forceinline__ void expand(const Raw &r, float d[8]) {
    d[0] = (float)(r.x & 0xFFFFu); d[1] = (float)(r.x >> 16); d[2] = (float)(r.y & 0xFFFFu); d[3] = (float)(r.y >> 16);
    d[4] = (float)(r.z & 0xFFFFu); d[5] = (float)(r.z >> 16); d[6] = (float)(r.w & 0xFFFFu); d[7] = (float)(r.w >> 16);
  }
  static __device__ __forceinline__ void load(const uint16_t *p, float d[8]) { expand(issue(p), d); }
};
template <typename T>
__global__ __launch_bounds__(256) void k_raw_scaled_demosaic_w8(const T *__restrict__ src, TransformArgs a, const uint8_t *__restrict__ cfa48,
                                                               float *__restrict__ dst) {
  // colours of the 8 columns starting at (y % 48, x % 48), 2 bits each: built once per block, 9 entries per thread, so a
  // block keeps working on rows (gridDim.y is a few blocks per CU, not one per row)
  __shared__ uint16_t s_bits[48 * 48];
  for (int i = threadIdx.x; i < 48 * 48; i += blockDim.x) {
    const int y = i / 48, x = i % 48;
    uint32_t bits = 0;
    #pragma unroll
    for (int k = 0; k < 8; ++k) { const int xx = x + k; bits |= (uint32_t)(cfa48[y * 48 + (xx >= 48 ? xx - 48 : xx)] & 3u) << (2 * k); }
    s_bits[i] = (uint16_t)bits;
  }
  __syncthreads();
  const uint32_t col_raw = blockIdx.x * blockDim.x + threadIdx.x;
  const bool lane_in = col_raw < a.nwidth;
  const uint32_t col = min(col_raw, a.nwidth - 1);       // lanes past the row shadow its last pixel: every lane stays active
  // column window (scaling.rs:84-89 with skip_x_y = skip_y_x = 0, tlx = tly = 0 as scale_down_buffer sets them); the host
  // launches this kernel only when no window is wider or taller than 8
  const uint32_t from_x = min(a.width - 1, f32_as_u32_sat(floorf(a.tlx + (a.skip_x_x * (float)col))));
  const uint32_t to_x = min(a.width - 1, f32_as_u32_sat(floorf(a.tlx + (a.skip_x_x * (float)(col + 1)))));
  const float center_x = a.tlx + (a.skip_y_x / 2.0f) - 0.5f + (a.skip_x_x * (float)col) + (a.skip_x_x / 2.0f);
  const uint32_t nx = to_x - from_x + 1;
  // the 8-sample load must stay inside the source row; the few pixels whose window touches the last columns start it
  // earlier and shift their taps (kshift)
  const uint32_t lx = min(from_x, a.width - 8);
  const uint32_t kshift = from_x - lx;
  float ax[8];                                           // 1 - dx*dx of sample lx + k (meaningful for kshift <= k < kshift + nx)
  #pragma unroll
  for (uint32_t k = 0; k < 8; ++k) {
    const float delta_x = tb_div((float)(lx + k) - center_x, a.skip_x_x, a.inv_skip_x_x, a.fast_x);
    ax[k] = 1.0f - (delta_x * delta_x);
  }
  const uint32_t xm0 = lx % 48;
  uint32_t kend = 0;                                     // one past the last sample index any lane of the wave uses
  #pragma unroll
  for (uint32_t k = 0; k < 8; ++k) if (__builtin_amdgcn_ballot_w64(kshift + nx > k) != 0) kend = k + 1;
  for (uint32_t row = a.out_r0 + blockIdx.y; row < a.out_r1; row += gridDim.y) {
    const uint32_t from_y = min(a.height - 1, f32_as_u32_sat(floorf(a.tly + a.skip_y_y * (float)row)));
    const uint32_t to_y = min(a.height - 1, f32_as_u32_sat(floorf(a.tly + a.skip_y_y * (float)(row + 1))));
    const float center_y = a.tly + (a.skip_y_y * (float)row) + (a.skip_y_y / 2.0f) - 0.5f + (a.skip_x_y * (float)col) + (a.skip_x_y / 2.0f);
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f, n0 = 0.0f, n1 = 0.0f, n2 = 0.0f, n3 = 0.0f;
    for (uint32_t y = from_y; y <= to_y; ++y) {         // wave-uniform bounds
      const float delta_y = tb_div((float)y - center_y, a.skip_y_y, a.inv_skip_y_y, a.fast_y);
      const float dy2 = delta_y * delta_y;
      float d[8];
      Row8<T>::load(src + (size_t)((uint64_t)y + a.src_y) * a.src_pitch + a.src_x + lx, d);
      const uint32_t bits = s_bits[(y % 48) * 48 + xm0];
      #pragma unroll
      for (int k = 0; k < 8; ++k) d[k] = d[k] - a.min0;
      bool fastdiv = a.norm_fast != 0;
      if (sizeof(T) == 4 && fastdiv) {
        bool g = false;
        #pragma unroll
        for (int k = 0; k < 8; ++k) g |= cdiv_guard(d[k]);
        fastdiv = __builtin_amdgcn_ballot_w64(g) == 0;
      }
      float q[8];                                         // gofloat.rs:126; one uniform branch, not a select per tap
      if (fastdiv) {
        #pragma unroll
        for (int k = 0; k < 8; ++k) q[k] = cdiv_fast(d[k], a.range0, a.inv_range0);
      } else {
        #pragma unroll
        for (int k = 0; k < 8; ++k) q[k] = d[k] / a.range0;
      }
      #pragma unroll
      for (uint32_t k = 0; k < 8; ++k) {
        if (k < kend) {                                   // wave-uniform
          const bool in = (k - kshift) < nx;              // unsigned: kshift <= k < kshift + nx
          float factor = ax[k] - dy2;                     // scaling.rs:106
          factor = (factor < 0.0f) ? 0.0f : factor;
          factor = in ? factor : 0.0f;
          float t = rs_min(q[k], 1.0f) * factor;
          t = in ? t : 0.0f;                              // a sample outside the window contributes nothing, whatever it holds
          const uint32_t c = (bits >> (2 * k)) & 3u;
          s0 += (c == 0) ? t : 0.0f; n0 += (c == 0) ? factor : 0.0f;
          s1 += (c == 1) ? t : 0.0f; n1 += (c == 1) ? factor : 0.0f;
          s2 += (c == 2) ? t : 0.0f; n2 += (c == 2) ? factor : 0.0f;
          if (a.components > 3) { s3 += (c == 3) ? t : 0.0f; n3 += (c == 3) ? factor : 0.0f; }
        }
      }
    }
    float4 o;
    o.x = (n0 > 0.0f) ? s0 / n0 : 0.0f; o.y = (n1 > 0.0f) ? s1 / n1 : 0.0f; o.z = (n2 > 0.0f) ? s2 / n2 : 0.0f; o.w = (n3 > 0.0f) ? s3 / n3 : 0.0f;
    if (lane_in) reinterpret_cast<float4 *>(dst)[(size_t)(row - a.out_r0) * a.nwidth + col] = o;
  }
}
// The same again for filters with a short period (pw x ph cells, pw * ph <= kW8MaxCells: Bayer, X-Trans, 12x12 ...): the
// colour bins are filled by fused multiply-adds with {0,1} weights instead of selects.  Per pattern cell (row phase, column
// phase of the lane's first sample) LDS holds the one-hot colour of each of the 8 window columns as four floats; a tap then costs
// s_c = fma(t, m_c, s_c) and n_c = fma(factor, m_c, n_c) per colour -- exact: t * 1 and t * 0 are exact, x + (+-0) = x, and a sum
// that starts at +0.0 stays +0.0 under + (+-0) -- for every FINITE t.  Samples outside the window get factor 0 through a
// column weight of -inf (max(-inf - dy*dy, 0) = 0) instead of two selects.  A row of the window whose dividends leave the
// proven zone of the fast normalisation (f32 sources: NaN, inf, absurd magnitudes; the same wave-uniform test as above) takes
// the select form, which never multiplies such a sample.  ~150 instead of ~265 instructions per window row.
constexpr uint32_t kW8MaxCells = 144;
// A cell's record is 8 float4 (128 bytes = one full cycle of the 32 LDS banks): with that stride the few DIFFERENT cells the lanes of a wave
// read in one ds_read_b128 (x phases 0 / 2 / 4 of a 6-wide pattern at scale 4) all fall on the same banks -- measured 15 conflict cycles
// per LDS instruction, the LDS busy for 61 of the kernel's 72 us (profiles/r02_c5_pmc.json).  One float4 of padding per cell moves
// neighbouring cells 16 bytes apart modulo 128: SQ_LDS_BANK_CONFLICT 21.6 M -> 0 cycles per launch, 0.072 -> 0.063 ms.
constexpr uint32_t kW8CellF4 = 9;
// KU = the window columns every lane of every wave has (floor(skip) + 1, capped at 5): their taps are straight-line code; further columns (a
// window whose phase wraps, the shifted loads at the right frame edge) sit behind one wave-uniform branch.
// The one-hot cell records of k_raw_scaled_demosaic_w8m, one (cell, column) pair per thread and step (a block lives for two or three output rows: eight
// dependent byte loads per cell in front of its barrier were a sixth of its life) and no division: thread t takes column k = t & 7 of the pattern cells
// x = (t >> 3) & 15, + 16, ... in the rows t >> 7, t >> 7 + 2, ...; tiles up to 48 wide are legal ("24x2:", "48x1:"), so x walks on and x + k wraps at the
// 48-column period of cfa48.  Out of line on purpose: inlined, its loop state pushed the row loop over the seven-waves-per-SIMD register budget (6 spills,
// 48 -> 76 us at 50 MP).
__device__ __attribute__((noinline)) static void w8m_fill_cells(float4 *__restrict__ cells, const uint8_t *__restrict__ cfa48, uint32_t pw, uint32_t ph) {
  const uint32_t k = threadIdx.x & 7u;
  for (uint32_t y = threadIdx.x >> 7; y < ph; y += blockDim.x >> 7)
    for (uint32_t x = (threadIdx.x >> 3) & 15u; x < pw; x += 16u) {
      const uint32_t xk = x + k;
      const uint32_t c = cfa48[y * 48 + (xk >= 48u ? xk - 48u : xk)] & 3u;
      cells[(y * pw + x) * kW8CellF4 + k] = make_float4(c == 0u ? 1.0f : 0.0f, c == 1u ? 1.0f : 0.0f, c == 2u ? 1.0f : 0.0f, c == 3u ? 1.0f : 0.0f);
    }
}
// C4 = the filter has a fourth colour (RGBE ...): three-colour filters (Bayer, X-Trans) skip its two accumulations per tap -- as a runtime flag they
// were two fused multiply-adds and two selects per tap, a quarter of the tap's arithmetic.
template <typename T, uint32_t KU, bool C4>
// (seven waves per SIMD asked for: with the two register sets of the unrolled row loop hipcc otherwise takes 73 VGPRs -- six waves per SIMD, 3 % slower;
// the four-colour variants, two more accumulators, get the six: at seven they spill)
__global__ __launch_bounds__(256, C4 ? 6 : 7) void k_raw_scaled_demosaic_w8m(const T *__restrict__ src, TransformArgs a, const uint8_t *__restrict__ cfa48,
                                                                uint32_t pw, uint32_t ph, float *__restrict__ dst) {
  extern __shared__ __attribute__((aligned(16))) float s_m[];          // [ph][pw] cells of kW8CellF4 float4: [8 columns][4 colours] one-hot weights + padding
  // tiles up to 16 wide (every shipped filter) in line -- thread t: column k = t & 7 of cell x = (t >> 3) & 15, rows t >> 7, + 2, ...; x + k < 48 needs no wrap;
  // wider ones through the general routine
  if (pw <= 16u) {
    const uint32_t k = threadIdx.x & 7u, x = (threadIdx.x >> 3) & 15u;
    if (x < pw)
      for (uint32_t y = threadIdx.x >> 7; y < ph; y += blockDim.x >> 7) {
        const uint32_t c = cfa48[y * 48 + x + k] & 3u;
        reinterpret_cast<float4 *>(s_m)[(y * pw + x) * kW8CellF4 + k] = make_float4(c == 0u ? 1.0f : 0.0f, c == 1u ? 1.0f : 0.0f, c == 2u ? 1.0f : 0.0f, c == 3u ? 1.0f : 0.0f);
      }
  } else w8m_fill_cells(reinterpret_cast<float4 *>(s_m), cfa48, pw, ph);
  __syncthreads();
  // Which block takes which columns and rows.  Output row r reads the source rows floor(skip r) .. floor(skip (r + 1)), so rows r and r + 1 share
  // one (five rows for every four at scale 4: 1.22x the frame in HBM fetches, profiles/r02_c5_pmc.json).  The hardware deals consecutive block ids
  // to the eight XCDs in turn, each with its own L2; with a plain gx x gy grid (gx = 9 for a 2160-pixel row) the blocks of neighbouring rows
  // always land on different XCDs and both fetch the shared row.  Launched one-dimensionally (IPK_OPT_W8M_XCD), block id b belongs to XCD b % 8 and
  // is the (b / 8)-th block there; runs of `xcd_group` neighbouring block rows go to one XCD, the next run to the next XCD, so that inside a run
  // the second reader of a shared source row finds it in that XCD's L2.  Measured (50 MP X-Trans -> 2160x1440, FETCH_SIZE x 2 + WRITE_SIZE over
  // the algorithmic 248.8 MB; kernel time on one box): plain grid 1.22x, 0.0595 ms; runs of 1 / 2 / 4 / 8 / 16 / 74 (an eighth of the frame per XCD)
  // rows: 1.20x 0.0594 | 1.11x 0.0593 | 1.06x 0.0596 | 1.04x 0.0600 | -- 0.0636 | 1.01x 0.059-0.067 (box to box).  Long runs save traffic the
  // kernel is not waiting for -- it is bound by VALU issue (46.6 M instructions per launch = 48 us of its 57) -- and cost time on some boxes;
  // runs of 2 take the traffic to 1.11x at no cost.
  uint32_t blk_x = blockIdx.x, blk_y = blockIdx.y, rows_step = gridDim.y;
  if (a.xcd_gx != 0) {
    const uint32_t xcd = blockIdx.x & 7u, j = blockIdx.x >> 3;
    const uint32_t per_group = a.xcd_group * a.xcd_gx, gi = j / per_group, rem = j - gi * per_group;
    blk_x = rem % a.xcd_gx;
    blk_y = (gi * 8u + xcd) * a.xcd_group + rem / a.xcd_gx;
    rows_step = a.xcd_gy;
  }
  const uint32_t col_raw = blk_x * blockDim.x + threadIdx.x;
  const bool lane_in = col_raw < a.nwidth;
  const uint32_t col = min(col_raw, a.nwidth - 1);
  const uint32_t from_x = min(a.width - 1, f32_as_u32_sat(floorf(a.tlx + (a.skip_x_x * (float)col))));
  const uint32_t to_x = min(a.width - 1, f32_as_u32_sat(floorf(a.tlx + (a.skip_x_x * (float)(col + 1)))));
  const float center_x = a.tlx + (a.skip_y_x / 2.0f) - 0.5f + (a.skip_x_x * (float)col) + (a.skip_x_x / 2.0f);
  const uint32_t nx = to_x - from_x + 1;
  const uint32_t lx = min(from_x, a.width - 8);
  const uint32_t kshift = from_x - lx;
  float axm[8];                                          // 1 - dx*dx of sample lx + k, -inf outside the lane's window
  {
    // (x - center_x) / skip_x_x (scaling.rs:104).  As for the rows below: tl = (0, 0) and 1 <= skip <= 7 (launcher), so the sample's column and the centre
    // are sums of terms that are zero or at least 2^-25 in magnitude, a nonzero difference is at least 2^-49 and at most the frame width -- inside the
    // multiply-fma division's proven zone without tb_div's exponent test (round 4: eight divergent tests per block; a block lives for two or three rows)
    #pragma unroll
    for (uint32_t k = 0; k < 8; ++k) {
      const float dx = (float)(lx + k) - center_x;
      const float delta_x = a.fast_x ? cdiv_fast(dx, a.skip_x_x, a.inv_skip_x_x) : dx / a.skip_x_x;
      axm[k] = ((k - kshift) < nx) ? 1.0f - (delta_x * delta_x) : -__builtin_inff();
    }
  }
  const uint32_t xm = lx % pw;
  uint32_t kend = 0;
  #pragma unroll
  for (uint32_t k = 0; k < 8; ++k) if (__builtin_amdgcn_ballot_w64(kshift + nx > k) != 0) kend = k + 1;
  // The window rows this block walks -- output row after output row (stride gridDim.y), from_y..to_y inside each (scaling.rs:93) -- form
  // ONE stream, software-pipelined: the loads of the row after next are issued at the end of an iteration, the next row's are awaited
  // after the current row's arithmetic and BEFORE the output row's store (gfx9 counts loads and stores in one vmcnt, and a wait with a
  // younger store in flight becomes vmcnt(0)).  Before, every window row paid its full HBM latency in front of its arithmetic (load,
  // s_waitcnt vmcnt(0), compute): the kernel ran at 3.1 TB/s; a frame is only ~18 window rows per wave, so occupancy could not hide it.
  // (Tried, round 2: a block walking a RUN of consecutive output rows instead of rows gridDim.y apart -- the last window row of output row r is
  // the first of row r + 1, so inside a run it could be loaded once: 4 R + 1 loads for R rows instead of 5 R.  Measured on one box, 50 MP -> 2160x1440:
  // strided 0.0567 ms; runs without the reuse 0.0607; runs with it 0.066 (whole-number rounds of blocks kept in all three).  Neighbouring rows
  // processed at the same time by different blocks is the better order for the memory system, and the load that is not issued saves nothing.)
  struct Cur { uint32_t row, y, ty; bool valid; };
  auto ywin = [&](uint32_t r, uint32_t &fy, uint32_t &ty) {
    fy = min(a.height - 1, f32_as_u32_sat(floorf(a.tly + a.skip_y_y * (float)r)));
    ty = min(a.height - 1, f32_as_u32_sat(floorf(a.tly + a.skip_y_y * (float)(r + 1))));
  };
  auto next_of = [&](const Cur &c) -> Cur {
    if (c.y < c.ty) return Cur{c.row, c.y + 1, c.ty, true};
    const uint32_t nr = c.row + rows_step;
    if (nr >= a.out_r1) return Cur{c.row, c.y, c.ty, false};            // past the end: re-reads a valid row, never consumed
    uint32_t f, tt; ywin(nr, f, tt);
    return Cur{nr, f, tt, true};
  };
  auto rowptr = [&](uint32_t y) { return src + (size_t)((uint64_t)y + a.src_y) * a.src_pitch + a.src_x + lx; };
  auto centre = [&](uint32_t r) { return a.tly + (a.skip_y_y * (float)r) + (a.skip_y_y / 2.0f) - 0.5f + (a.skip_x_y * (float)col) + (a.skip_x_y / 2.0f); };
  Cur c0; c0.row = a.out_r0 + blk_y; c0.valid = true;
  if (c0.row >= a.out_r1) return;
  ywin(c0.row, c0.y, c0.ty);
  typename Row8<T>::Raw cur = Row8<T>::issue(rowptr(c0.y));
  Cur c1 = next_of(c0);
  typename Row8<T>::Raw nxt = Row8<T>::issue(rowptr(c1.y));
  float center_y = centre(c0.row);
  float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f, n0 = 0.0f, n1 = 0.0f, n2 = 0.0f, n3 = 0.0f;
  // one window row: `cur` holds its samples, `nxt` the next row's (in flight); the row after next is then loaded INTO `cur`.  The loop below calls it
  // with the two register sets in alternating roles, so the rows never move between registers (the rolling form copied ten registers per window row).
  auto window_row = [&](typename Row8<T>::Raw &cur, typename Row8<T>::Raw &nxt) -> bool {
    const uint32_t y = c0.y;
    {
      // (y - center_y) / skip_y_y (scaling.rs:105).  This kernel only runs for tl = (0, 0) and 1 <= skip <= 7 (launcher): y and the
      // centre are then zero or at least 2^-25 in magnitude, so a nonzero difference is at least 2^-49 and at most the frame height --
      // inside the three-step division's proven zone without the exponent test tb_div carries for rotated transforms
      const float dyn = (float)y - center_y;
      const float delta_y = a.fast_y ? cdiv_fast(dyn, a.skip_y_y, a.inv_skip_y_y) : dyn / a.skip_y_y;
      const float dy2 = delta_y * delta_y;
      float d[8];
      Row8<T>::expand(cur, d);
      const uint32_t cell = (y % ph) * pw + xm;
      #pragma unroll
      for (int k = 0; k < 8; ++k) d[k] = d[k] - a.min0;
      bool fastdiv = a.norm_fast != 0;
      if (sizeof(T) == 4 && fastdiv) {
        // Dividends outside the division's proven zone.  Huge positive ones, +inf and NaN clip to 1.0 whatever the division does; the
        // weights form multiplies every sample by 0 or 1, so -inf (kept by .min(1.0)) would turn the other colours' sums into NaN: one
        // comparison on the row's minimum covers the negative side.  Tiny nonzero dividends need the per-sample exponent test -- unless
        // the black level rules them out (norm_light: a nonzero v - black is then at least half an ulp of black; 45 -> 10 instructions
        // per window row, a quarter of this kernel's arithmetic).
        // d < -2^100 (or -inf) as an unsigned comparison of the bit patterns: negative floats order by magnitude, and the only other
        // patterns above 0xF1800000 are sign-bit NaNs, which may take the literal path too (4 integer max instead of a float min tree
        // whose every operand the compiler canonicalises first)
        const uint32_t umax = max(max(max(__float_as_uint(d[0]), __float_as_uint(d[1])), max(__float_as_uint(d[2]), __float_as_uint(d[3]))),
                                  max(max(__float_as_uint(d[4]), __float_as_uint(d[5])), max(__float_as_uint(d[6]), __float_as_uint(d[7]))));
        bool g = umax > 0xF1800000u;
        if (!a.norm_light) {
          #pragma unroll
          for (int k = 0; k < 8; ++k) g |= cdiv_guard(d[k]);
        }
        fastdiv = __builtin_amdgcn_ballot_w64(g) == 0;
      }
      if (fastdiv) {
        const float4 *m = reinterpret_cast<const float4 *>(s_m) + cell * kW8CellF4;
        auto tap = [&](uint32_t k) {
          // gofloat.rs:126.  cdiv_fast without its div_fixup: that only repairs zero / inf / NaN dividends, and here 0 gives 0
          // either way while +inf and NaN give NaN, which .min(1.0) turns into the same 1.0 as inf.min(1.0)
          const float q0 = d[k] * a.inv_range0;
          const float q = __builtin_fmaf(__builtin_fmaf(-q0, a.range0, d[k]), a.inv_range0, q0);
          const float factor = fmaxf(axm[k] - dy2, 0.0f);                   // scaling.rs:106-107; 0 outside the window
          const float t = rs_min(q, 1.0f) * factor;
          const float4 mk = m[k];
          s0 = __builtin_fmaf(t, mk.x, s0); n0 = __builtin_fmaf(factor, mk.x, n0);
          s1 = __builtin_fmaf(t, mk.y, s1); n1 = __builtin_fmaf(factor, mk.y, n1);
          s2 = __builtin_fmaf(t, mk.z, s2); n2 = __builtin_fmaf(factor, mk.z, n2);
          if (C4) { s3 = __builtin_fmaf(t, mk.w, s3); n3 = __builtin_fmaf(factor, mk.w, n3); }
        };
        #pragma unroll
        for (uint32_t k = 0; k < KU; ++k) tap(k);          // every lane's window has these columns (or weight 0 for them)
        if (kend > KU) {                                   // wave-uniform, rare: a wrapped window phase, the frame's right edge
          #pragma unroll
          for (uint32_t k = KU; k < 8; ++k)
            if (k < kend) tap(k);
        }
      } else {
        const float4 *m = reinterpret_cast<const float4 *>(s_m) + cell * kW8CellF4;
        #pragma unroll
        for (uint32_t k = 0; k < 8; ++k) {
          if (k < kend) {
            const bool in = (k - kshift) < nx;
            const float q = d[k] / a.range0;
            float factor = axm[k] - dy2;                  // -inf outside the window: the selects below zero it either way
            factor = (factor < 0.0f) ? 0.0f : factor;
            factor = in ? factor : 0.0f;
            float t = rs_min(q, 1.0f) * factor;
            t = in ? t : 0.0f;
            const float4 mk = m[k];                       // the column's colour, read back from its one-hot record
            s0 += (mk.x != 0.0f) ? t : 0.0f; n0 += (mk.x != 0.0f) ? factor : 0.0f;
            s1 += (mk.y != 0.0f) ? t : 0.0f; n1 += (mk.y != 0.0f) ? factor : 0.0f;
            s2 += (mk.z != 0.0f) ? t : 0.0f; n2 += (mk.z != 0.0f) ? factor : 0.0f;
            if (C4) { s3 += (mk.w != 0.0f) ? t : 0.0f; n3 += (mk.w != 0.0f) ? factor : 0.0f; }
          }
        }
      }
    }
    const bool row_end = c0.y == c0.ty;                     // wave-uniform
    Row8<T>::touch(nxt);                                    // the wait for the next row's samples: nothing younger is in flight
    __builtin_amdgcn_sched_barrier(0);
    if (row_end) {
      float4 o;
      o.x = (n0 > 0.0f) ? s0 / n0 : 0.0f; o.y = (n1 > 0.0f) ? s1 / n1 : 0.0f; o.z = (n2 > 0.0f) ? s2 / n2 : 0.0f; o.w = (n3 > 0.0f) ? s3 / n3 : 0.0f;
      // lanes past the row shadow its last pixel (same value to the same address): the store needs no predicate
      (void)lane_in;
      reinterpret_cast<float4 *>(dst)[(size_t)(c0.row - a.out_r0) * a.nwidth + col] = o;
      s0 = s1 = s2 = s3 = n0 = n1 = n2 = n3 = 0.0f;
    }
    __builtin_amdgcn_sched_barrier(0);
    if (!c1.valid) return false;
    const Cur c2 = next_of(c1);
    cur = Row8<T>::issue(rowptr(c2.y));
    if (row_end) center_y = centre(c1.row);
    c0 = c1; c1 = c2;
    return true;
  };
  for (;;) {
    if (!window_row(cur, nxt)) break;
    if (!window_row(nxt, cur)) break;
  }
}
template <typename T>
void launch_raw_scaled_demosaic(const T *src, size_t owidth, size_t x, size_t y, size_t width, size_t height, float black0, float white0,
                                int norm_fast, int has_fourth_colour, size_t nwidth, size_t nheight, const uint8_t *cfa48_dev, int pw, int ph, float *dst4, hipStream_t s,
                                size_t band_src_row0, size_t band_out_row0, size_t band_out_rows) {
  TransformArgs a{};
  a.width = (uint32_t)width; a.height = (uint32_t)height; a.nwidth = (uint32_t)nwidth; a.nheight = (uint32_t)nheight; a.components = 4;
  a.tlx = 0.0f; a.tly = 0.0f;                                               // scale_down_buffer's corners (scaling.rs:35-48)
  a.skip_x_x = ((float)((int64_t)width - 1) - 0.0f) / ((float)(nwidth - 1));
  a.skip_x_y = (0.0f - 0.0f) / ((float)(nwidth - 1));
  a.skip_y_x = (0.0f - 0.0f) / ((float)(nheight - 1));
  a.skip_y_y = ((float)((int64_t)height - 1) - 0.0f) / ((float)(nheight - 1));
  a.inv_skip_x_x = 1.0f / a.skip_x_x; a.inv_skip_y_y = 1.0f / a.skip_y_y;
  a.fast_x = cdiv_host_ok(a.skip_x_x); a.fast_y = cdiv_host_ok(a.skip_y_y);
  a.has_cfa = 1; a.norm = 1; a.min0 = black0; a.range0 = white0 - black0; a.src_pitch = owidth; a.src_x = x; a.src_y = y;
  a.norm_fast = norm_fast; a.inv_range0 = 1.0f / a.range0;
  a.norm_light = (std::fabs(black0) >= 0x1p-70f && std::fabs(black0) <= 0x1p70f) ? 1 : 0;
  // whole frame, or the output-row band [band_out_row0, +band_out_rows) of a frame sharded across GPUs: `src` then points at sensor
  // row y + band_src_row0 (the slab's first row), which is folded into src_y; window bounds still clamp against the full height
  a.out_r0 = 0; a.out_r1 = (uint32_t)nheight;
  if (band_out_rows) { a.out_r0 = (uint32_t)band_out_row0; a.out_r1 = (uint32_t)(band_out_row0 + band_out_rows); a.src_y = (uint64_t)0 - (uint64_t)band_src_row0; }
  const size_t out_rows = a.out_r1 - a.out_r0;
  a.xcd_gx = 0; a.xcd_gy = 0; a.xcd_group = 0;
  a.components = has_fourth_colour ? 4 : 3;               // the w8 kernel skips the fourth bin for three-colour filters (it stays 0.0)
  // windows of at most 8 x 8 samples: floor(skip*(c+1)) - floor(skip*c) + 1 <= ceil(skip) + 1
  if (a.skip_x_x >= 1.0f && a.skip_x_x <= 7.0f && a.skip_y_y >= 1.0f && a.skip_y_y <= 7.0f && width >= 8 &&
      (reinterpret_cast<uintptr_t>(dst4) & 15) == 0) {
    const unsigned gx = (unsigned)((nwidth + 255) / 256);
    // 21 blocks per CU = 3 full rounds of the 7 resident ones.  Round 4 sweep (50 MP X-Trans -> 2160x1440, one box, ms): 1620 / 1792 / 2688 / 3584 / 5376 /
    // 7168 blocks 0.064 / 0.059 / 0.058 / 0.0565 / 0.0563 / 0.0565 with the plain grid, 0.064 / 0.061 / 0.060 / 0.057 / 0.059 / 0.058 with runs of two
    // block rows per XCD; equal row counts per block (every block the same 3, 4, 6 or 8 rows) 0.064 / 0.064 / 0.061 / 0.058 / 0.057 / 0.059: fewer, longer-lived
    // blocks lose although they run the per-block set-up less often -- the kernel is not bound by its instruction count alone.
    // (round 5, after the kernel lost a quarter of its instructions: 3584 / 4096 / 5376 / 6144 / 8192 blocks and 7 or 8 resident per CU all within 2 %)
    const unsigned total_blocks = 5376u;
    const unsigned want = std::max(1u, total_blocks / gx);
    const dim3 grid2(gx, (unsigned)std::min<size_t>(out_rows, want), 1);
    const dim3 grid = grid2;
    if (pw > 0 && ph > 0 && (uint32_t)(pw * ph) <= kW8MaxCells && 48 % pw == 0 && 48 % ph == 0) {
      const size_t lds = (size_t)pw * ph * kW8CellF4 * 4 * sizeof(float);
      dim3 grid = grid2;
      a.xcd_gx = 0; a.xcd_gy = 0;
      {
        uint32_t group = 4;     // block rows per XCD in a run (round 5: 2 / 4 / 8 the same time, 16 slower; 4 takes the HBM traffic from 1.11x to 1.06x)
        if (group > 0 && grid2.y >= 8 * group) {
          a.xcd_gx = grid2.x; a.xcd_group = group; a.xcd_gy = grid2.y / (8 * group) * (8 * group);
          grid = dim3(a.xcd_gx * a.xcd_gy, 1, 1);
        }
      }
      if (has_fourth_colour) {
        if (a.skip_x_x >= 4.0f) hipLaunchKernelGGL((k_raw_scaled_demosaic_w8m<T, 5, true>), grid, dim3(256), lds, s, src, a, cfa48_dev, (uint32_t)pw, (uint32_t)ph, dst4);
        else if (a.skip_x_x >= 2.0f) hipLaunchKernelGGL((k_raw_scaled_demosaic_w8m<T, 3, true>), grid, dim3(256), lds, s, src, a, cfa48_dev, (uint32_t)pw, (uint32_t)ph, dst4);
        else hipLaunchKernelGGL((k_raw_scaled_demosaic_w8m<T, 2, true>), grid, dim3(256), lds, s, src, a, cfa48_dev, (uint32_t)pw, (uint32_t)ph, dst4);
      } else {
        if (a.skip_x_x >= 4.0f) hipLaunchKernelGGL((k_raw_scaled_demosaic_w8m<T, 5, false>), grid, dim3(256), lds, s, src, a, cfa48_dev, (uint32_t)pw, (uint32_t)ph, dst4);
        else if (a.skip_x_x >= 2.0f) hipLaunchKernelGGL((k_raw_scaled_demosaic_w8m<T, 3, false>), grid, dim3(256), lds, s, src, a, cfa48_dev, (uint32_t)pw, (uint32_t)ph, dst4);
        else hipLaunchKernelGGL((k_raw_scaled_demosaic_w8m<T, 2, false>), grid, dim3(256), lds, s, src, a, cfa48_dev, (uint32_t)pw, (uint32_t)ph, dst4);
      }
      return;
    }
    hipLaunchKernelGGL(k_raw_scaled_demosaic_w8<T>, grid, dim3(256), 0, s, src, a, cfa48_dev, dst4);
    return;
  }
  hipLaunchKernelGGL(k_raw_scaled_demosaic<T>, grid_rows_few(nwidth, out_rows, 128, 8192), dim3(128), 0, s, src, a, cfa48_dev, dst4);
}
template void launch_raw_scaled_demosaic<uint16_t>(const uint16_t *, size_t, size_t, size_t, size_t, size_t, float, float, int, int, size_t, size_t, const uint8_t *, int, int, float *, hipStream_t, size_t, size_t, size_t);
template void launch_raw_scaled_demosaic<float>(const float *, size_t, size_t, size_t, size_t, size_t, float, float, int, int, size_t, size_t, const uint8_t *, int, int, float *, hipStream_t, size_t, size_t, size_t);
template void launch_transform_buffer<float>(const float *, size_t, size_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, size_t, size_t, size_t, const uint8_t *, float *, hipStream_t);
template void launch_transform_buffer<uint8_t>(const uint8_t *, size_t, size_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, size_t, size_t, size_t, const uint8_t *, uint8_t *, hipStream_t);
template void launch_transform_buffer<uint16_t>(const uint16_t *, size_t, size_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, size_t, size_t, size_t, const uint8_t *, uint16_t *, hipStream_t);

// ------------------------------------------------------------------------------------------
// Point-wise stage kernels: one thread per pixel, 12/16-byte per-lane accesses that are
// contiguous across the wave.
// ------------------------------------------------------------------------------------------
struct __attribute__((packed, aligned(4))) f3 { float x, y, z; };

// OpToLab::run (src/ops/colorspaces.rs:103-111)
__global__ __launch_bounds__(1024) void k_tolab(const float4 *__restrict__ src, size_t n, ToLabParams p,
                                                const LutPair *__restrict__ lab_pairs, f3 *__restrict__ dst) {
  __shared__ LutPair s_lab[kLutPairs];
  load_lut_pairs(s_lab, lab_pairs);
  __syncthreads();
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = src[i];
    f3 o;
    camera_to_lab(s_lab, p, v.x, v.y, v.z, v.w, o.x, o.y, o.z);
    dst[i] = o;
  }
}
// OpBaseCurve::run (src/ops/curves.rs:44-48): channel 0 only, the rest is the clone
__global__ void k_basecurve(const f3 *__restrict__ src, size_t n, SplineDev sp, f3 *__restrict__ dst) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float *pi = reinterpret_cast<const float *>(src + i);
    float *po = reinterpret_cast<float *>(dst + i);
    const float l = ld_stream(pi), ca = ld_stream(pi + 1), cb = ld_stream(pi + 2);
    st_stream(po, spline_interpolate(sp, l)); st_stream(po + 1, ca); st_stream(po + 2, cb);
  }
}
// OpFromLab::run (src/ops/colorspaces.rs:128-136)
__global__ void k_fromlab(const f3 *__restrict__ src, size_t n, Mat9 m, f3 *__restrict__ dst) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float *pi = reinterpret_cast<const float *>(src + i);
    float *po = reinterpret_cast<float *>(dst + i);
    const float l = ld_stream(pi), ca = ld_stream(pi + 1), cb = ld_stream(pi + 2);
    f3 o;
    lab_to_rgb(m, l, ca, cb, o.x, o.y, o.z);
    st_stream(po, o.x); st_stream(po + 1, o.y); st_stream(po + 2, o.z);
  }
}
// OpGamma::run (src/ops/gamma.rs:20-24): every sample
__global__ __launch_bounds__(1024) void k_gamma(const float *__restrict__ src, size_t n, const LutPair *__restrict__ gam_pairs,
                                                float *__restrict__ dst) {
  __shared__ LutPair s_gam[kLutPairs];
  load_lut_pairs(s_gam, gam_pairs);
  __syncthreads();
  // four samples per thread (16-byte accesses) while whole groups remain, then the tail one by one
  const size_t n4 = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0 ? n / 4 : 0;
  // Four 16-byte groups per thread and iteration, all four loads issued before the first lookup, and the four ADJACENT: a block covers 64 KB of
  // contiguous input per step, blocks side by side, and the launch holds sixteen blocks per CU slot so that the dispatcher keeps the chip on one
  // compact, advancing window of the buffer (4096 blocks at 100 MP).  Measured at 100 MP on one box: round 3's form (two groups a grid width apart,
  // 512 blocks) 0.470 ms = 5.1 TB/s; four groups a grid width apart 0.446-0.470 whatever the grid; adjacent groups 0.403 with 512 blocks, 0.383 with
  // 4096-8192 (6.3 TB/s, what a plain copy reaches), 0.446 with 32768: the table fill of a short-lived block starts to show.  The groups are finished
  // one after the other (scheduling barriers): with all sixteen lookups interleaved the kernel needs 68 registers, and two blocks per CU want 64.
  auto one = [&](size_t k, const float4 &v) {
    st_stream4(dst + 4 * k, make_float4(gamma_sample(s_gam, v.x), gamma_sample(s_gam, v.y), gamma_sample(s_gam, v.z), gamma_sample(s_gam, v.w)));
    __builtin_amdgcn_sched_barrier(0);
  };
  const size_t bs = 4 * (size_t)blockDim.x, gs = bs * gridDim.x;
  size_t j = (size_t)blockIdx.x * bs + threadIdx.x;
  for (; j + 3 * blockDim.x < n4; j += gs) {
    const float4 v0 = ld_stream4(src + 4 * j), v1 = ld_stream4(src + 4 * (j + blockDim.x)), v2 = ld_stream4(src + 4 * (j + 2 * blockDim.x)), v3 = ld_stream4(src + 4 * (j + 3 * blockDim.x));
    __builtin_amdgcn_sched_barrier(0);
    one(j, v0); one(j + blockDim.x, v1); one(j + 2 * blockDim.x, v2); one(j + 3 * blockDim.x, v3);
  }
  if (j < n4) {                                            // the block whose span holds the end of the buffer
    #pragma unroll
    for (int u = 0; u < 4; ++u) if (j + u * blockDim.x < n4) one(j + u * blockDim.x, ld_stream4(src + 4 * (j + u * blockDim.x)));
  }
  for (size_t i = 4 * n4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = gamma_sample(s_gam, src[i]);
}
// rotate_buffer (src/ops/transform.rs:130-141): strided gather of 3-channel pixels
// T = float: rotate_buffer on the OpBuffer (transform.rs:130-141).  T = uint8_t / uint16_t: the same permutation applied to the
// quantised image -- output8bit / output16bit act per sample, so quantise-then-rotate equals the reference's rotate-then-
// quantise and moves 4x / 2x fewer bytes through the permutation.
template <typename T> struct __attribute__((packed)) Px3 { T x, y, z; };
template <typename T>
__global__ void k_rotate(const Px3<T> *__restrict__ src, uint32_t owidth, uint32_t oheight, int64_t base_offset, int64_t x_step,
                         int64_t y_step, Px3<T> *__restrict__ dst) {
  const uint32_t col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= owidth) return;
  for (uint32_t row = blockIdx.y; row < oheight; row += gridDim.y) {
    const int64_t offset = base_offset + y_step * (int64_t)row + x_step * (int64_t)col;   // in pixels
    dst[(size_t)row * owidth + col] = src[offset];
  }
}
// output8bit / output16bit loops (src/pipeline.rs:408-414, :455-461)
__global__ void k_output8(const float *__restrict__ src, size_t n, uint8_t *__restrict__ dst) {
  const size_t n4 = ((reinterpret_cast<uintptr_t>(src) & 15) | (reinterpret_cast<uintptr_t>(dst) & 3)) == 0 ? n / 4 : 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = ld_stream4(src + 4 * i);
    st_stream_u(reinterpret_cast<uint32_t *>(dst) + i, output8bit_x4(v.x, v.y, v.z, v.w));
  }
  for (size_t i = 4 * n4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = output8bit(src[i]);
}
__global__ void k_output16(const float *__restrict__ src, size_t n, uint16_t *__restrict__ dst) {
  const size_t n4 = ((reinterpret_cast<uintptr_t>(src) & 15) | (reinterpret_cast<uintptr_t>(dst) & 7)) == 0 ? n / 4 : 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = ld_stream4(src + 4 * i);
    st_stream_u(reinterpret_cast<uint32_t *>(dst) + 2 * i, output16bit_x2(v.x, v.y));
    st_stream_u(reinterpret_cast<uint32_t *>(dst) + 2 * i + 1, output16bit_x2(v.z, v.w));
  }
  for (size_t i = 4 * n4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = output16bit(src[i]);
}

static ToLabParams make_tolab(const float *mul4, const float *cm12) {
  ToLabParams p;
  for (int i = 0; i < 4; ++i) p.mul[i] = mul4[i];
  for (int i = 0; i < 12; ++i) p.cm[i] = cm12[i];
  return p;
}
// spline_interpolate_3a's precondition (ipk_device.hpp): finite coefficients, knot ordinates that are not -0.0
static bool spline3_arith_ok(const SplineDev &d) {
  for (int i = 0; i < 3; ++i) {
    if (!std::isfinite(d.px[i]) || !std::isfinite(d.py[i]) || (d.py[i] == 0.0f && std::signbit(d.py[i]))) return false;
    if (i < 2 && !(std::isfinite(d.c1[i]) && std::isfinite(d.c2[i]) && std::isfinite(d.c3[i]))) return false;
  }
  return true;
}
// spline_interpolate_grid's preconditions (ipk_device.hpp): four or more knots, strictly increasing abscissae, finite coefficients, knot ordinates that
// are not -0.0, and no grid cell with two knots (x_0 aside) under the very arithmetic the kernels use for the cell
static bool spline_grid_ok(const SplineDev &d, float &scale) {
  if (d.npoints < 4 || d.npoints > kSplineMaxKnots || d.nseg != d.npoints - 1) return false;
  for (int i = 0; i < d.npoints; ++i) {
    if (!std::isfinite(d.px[i]) || !std::isfinite(d.py[i]) || (d.py[i] == 0.0f && std::signbit(d.py[i]))) return false;
    if (i > 0 && !(d.px[i] > d.px[i - 1])) return false;
    if (i < d.nseg && !(std::isfinite(d.c1[i]) && std::isfinite(d.c2[i]) && std::isfinite(d.c3[i]))) return false;
  }
  const float span = d.px[d.npoints - 1] - d.px[0];
  scale = (float)kGridCells / span;
  if (!(span > 0.0f) || !std::isfinite(scale) || !(scale > 0.0f)) return false;
  uint32_t prev = 0; bool have = false;
  for (int k = 1; k < d.nseg; ++k) {
    const uint32_t ck = spline_grid_cell(d.px[k], d.px[0], scale);
    if (have && ck <= prev) return false;
    prev = ck; have = true;
  }
  return true;
}
static SplineDev make_spline(const SplineHost &h) {
  SplineDev d;
  d.npoints = h.npoints; d.nseg = h.nseg;
  for (int i = 0; i < kSplineMaxKnots; ++i) { d.px[i] = h.px[i]; d.py[i] = h.py[i]; d.c1[i] = h.c1[i]; d.c2[i] = h.c2[i]; d.c3[i] = h.c3[i]; }
  d.grid_ok = 0; d.grid_scale = 0.0f; d.grid_xl = 0.0f; d.grid_yl = 0.0f;
  float scale;
  if (spline_grid_ok(d, scale)) { d.grid_ok = 1; d.grid_scale = scale; d.grid_xl = d.px[d.npoints - 1]; d.grid_yl = d.py[d.npoints - 1]; }
  return d;
}

void launch_tolab(const float *src4, size_t npix, const float *mul4, const float *cm12, const void *lab_pairs, float *dst3,
                  int num_cus, hipStream_t s) {
  hipLaunchKernelGGL(k_tolab, dim3(grid_1d(npix, 1024, (unsigned)num_cus * 2)), dim3(1024), 0, s,
                     reinterpret_cast<const float4 *>(src4), npix, make_tolab(mul4, cm12),
                     reinterpret_cast<const LutPair *>(lab_pairs), reinterpret_cast<f3 *>(dst3));
}
void launch_basecurve(const float *src3, size_t npix, const SplineHost &sp, float *dst3, int num_cus, hipStream_t s) {
  hipLaunchKernelGGL(k_basecurve, dim3(grid_1d(npix, 256, flat_cap((unsigned)num_cus * 16))), dim3(256), 0, s,
                     reinterpret_cast<const f3 *>(src3), npix, make_spline(sp), reinterpret_cast<f3 *>(dst3));
}
void launch_fromlab(const float *src3, size_t npix, const float *m9, float *dst3, int num_cus, hipStream_t s) {
  Mat9 m; for (int i = 0; i < 9; ++i) m.m[i] = m9[i];
  hipLaunchKernelGGL(k_fromlab, dim3(grid_1d(npix, 256, flat_cap((unsigned)num_cus * 16))), dim3(256), 0, s,
                     reinterpret_cast<const f3 *>(src3), npix, m, reinterpret_cast<f3 *>(dst3));
}
void launch_gamma(const float *src, size_t n, const void *gam_pairs, float *dst, int num_cus, hipStream_t s) {
  // sixteen blocks per CU slot (two are resident at a time), each walking 64 KB-wide steps: see k_gamma
  hipLaunchKernelGGL(k_gamma, dim3(grid_1d((n / 4 + 3) / 4, 1024, (unsigned)num_cus * 16)), dim3(1024), 0, s, src, n,
                     reinterpret_cast<const LutPair *>(gam_pairs), dst);
}
// The transposing orientations (Rotate90/270, Transpose, Transverse: |y_step| == 1, |x_step| == source pitch): consecutive
// output ROWS are consecutive source pixels, so a 32 x 32 tile is read along the output rows (coalesced in the source),
// turned through LDS, and written along the output columns (coalesced in the destination).
template <typename T>
__global__ __launch_bounds__(256) void k_rotate_transposed(const Px3<T> *__restrict__ src, uint32_t owidth, uint32_t oheight, int64_t base_offset,
                                                          int64_t x_step, int64_t y_step, Px3<T> *__restrict__ dst) {
  __shared__ T tile[32][33 * 3];
  const uint32_t C0 = blockIdx.x * 32, R0 = blockIdx.y * 32;
  const uint32_t a = threadIdx.x & 31u, b = threadIdx.x >> 5;          // 32 x 8 threads
  #pragma unroll
  for (uint32_t i = 0; i < 4; ++i) {
    const uint32_t lr = a, lc = b + 8 * i;                              // lanes run along the output rows
    const uint32_t r = R0 + lr, c = C0 + lc;
    if (r < oheight && c < owidth) {
      const Px3<T> v = src[base_offset + y_step * (int64_t)r + x_step * (int64_t)c];
      tile[lc][3 * lr] = v.x; tile[lc][3 * lr + 1] = v.y; tile[lc][3 * lr + 2] = v.z;
    }
  }
  __syncthreads();
  #pragma unroll
  for (uint32_t i = 0; i < 4; ++i) {
    const uint32_t lc = a, lr = b + 8 * i;                              // lanes run along the output columns
    const uint32_t r = R0 + lr, c = C0 + lc;
    if (r < oheight && c < owidth) dst[(size_t)r * owidth + c] = Px3<T>{tile[lc][3 * lr], tile[lc][3 * lr + 1], tile[lc][3 * lr + 2]};
  }
}
template <typename T>
void launch_rotate(const T *src3, size_t owidth, size_t oheight, int64_t base_offset_px, int64_t x_step_px, int64_t y_step_px,
                   T *dst3, hipStream_t s) {
  if ((y_step_px == 1 || y_step_px == -1) && x_step_px != 1 && x_step_px != -1 && (oheight + 31) / 32 <= 65535) {
    hipLaunchKernelGGL(k_rotate_transposed<T>, dim3((unsigned)((owidth + 31) / 32), (unsigned)((oheight + 31) / 32), 1), dim3(256), 0, s,
                       reinterpret_cast<const Px3<T> *>(src3), (uint32_t)owidth, (uint32_t)oheight, base_offset_px, x_step_px, y_step_px,
                       reinterpret_cast<Px3<T> *>(dst3));
    return;
  }
  hipLaunchKernelGGL(k_rotate<T>, grid_rows(owidth, oheight, 256), dim3(256), 0, s, reinterpret_cast<const Px3<T> *>(src3),
                     (uint32_t)owidth, (uint32_t)oheight, base_offset_px, x_step_px, y_step_px, reinterpret_cast<Px3<T> *>(dst3));
}
template void launch_rotate<float>(const float *, size_t, size_t, int64_t, int64_t, int64_t, float *, hipStream_t);
template void launch_rotate<uint8_t>(const uint8_t *, size_t, size_t, int64_t, int64_t, int64_t, uint8_t *, hipStream_t);
template void launch_rotate<uint16_t>(const uint16_t *, size_t, size_t, int64_t, int64_t, int64_t, uint16_t *, hipStream_t);
// The same permutation on a 1-channel image (the sensor mosaic, u16 or f32), source addressed through its own pitch and crop
// window: `base`, `x_step`, `y_step` in source elements.  Feeds the fused kernel's rotated-space variants.
template <typename T>
__global__ void k_rotate1(const T *__restrict__ src, uint32_t owidth, uint32_t oheight, int64_t base_offset, int64_t x_step, int64_t y_step,
                          T *__restrict__ dst) {
  const uint32_t col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= owidth) return;
  for (uint32_t row = blockIdx.y; row < oheight; row += gridDim.y) dst[(size_t)row * owidth + col] = src[base_offset + y_step * (int64_t)row + x_step * (int64_t)col];
}
// flips and 180 degrees (|x_step| == 1): 16 bytes per lane, reversed inside the lane when the walk descends
template <typename T>
__global__ void k_rotate1_rows(const T *__restrict__ src, uint32_t owidth, uint32_t oheight, int64_t base_offset, int64_t x_step, int64_t y_step,
                               T *__restrict__ dst) {
  constexpr uint32_t NV = 16 / sizeof(T);
  struct __attribute__((packed, aligned(sizeof(T)))) Vec { T v[NV]; };
  const uint32_t c0 = (blockIdx.x * blockDim.x + threadIdx.x) * NV;
  if (c0 >= owidth) return;
  for (uint32_t row = blockIdx.y; row < oheight; row += gridDim.y) {
    const int64_t o0 = base_offset + y_step * (int64_t)row + x_step * (int64_t)c0;
    T *o = dst + (size_t)row * owidth + c0;
    if (c0 + NV <= owidth) {
      Vec v = *reinterpret_cast<const Vec *>(src + (x_step > 0 ? o0 : o0 - (int64_t)(NV - 1)));
      if (x_step < 0) {
        #pragma unroll
        for (uint32_t k = 0; k < NV / 2; ++k) { const T t = v.v[k]; v.v[k] = v.v[NV - 1 - k]; v.v[NV - 1 - k] = t; }
      }
      *reinterpret_cast<Vec *>(o) = v;
    } else {
      for (uint32_t k = 0; c0 + k < owidth; ++k) o[k] = src[o0 + x_step * (int64_t)k];
    }
  }
}
// Transposing orientations of the 1-channel mosaic: dst[r][c] = src[base + y_step r + x_step c] with y_step = +-1 (consecutive OUTPUT rows are consecutive
// source elements).  Tiles of 64 x 64 elements through LDS; a full tile moves 16 bytes per lane on BOTH sides -- a lane loads the 8 (u16) / 4 (f32)
// source-contiguous elements of one output column, and stores that many consecutive columns of one output row, gathered element by element from the tile.
// Round 5 (tools/transpose_probe.hip, 100 MP u16): one element per lane and access, the form of rounds 1-4, 2.65 TB/s -- its limit was the texture
// addresser's 64 lanes x 2 bytes per instruction, not the walk across 64 source rows as round 1 concluded from tile-shape sweeps that all kept
// narrow accesses; 16 bytes per lane: 4.9 TB/s (a plain copy of the same bytes 5.3); tiles of 64 x 128 ... 128 x 128 4.8-5.1, 256 x 256 3.7.
// Frame-edge tiles keep the element-wise form.
template <typename T>
__global__ __launch_bounds__(256) void k_rotate1_transposed(const T *__restrict__ src, uint32_t owidth, uint32_t oheight, int64_t base_offset,
                                                           int64_t x_step, int64_t y_step, T *__restrict__ dst) {
  constexpr uint32_t TW = 64, NV = 16 / sizeof(T), PITCH = TW + 2;
  struct __attribute__((packed, aligned(sizeof(T)))) Vec { T v[NV]; };
  __shared__ T tile[TW * PITCH];                                        // tile[c_local][r_local]
  const uint32_t C0 = blockIdx.x * TW, R0 = blockIdx.y * TW;
  if (C0 + TW <= owidth && R0 + TW <= oheight) {                        // block-uniform
    constexpr uint32_t VPR = TW / NV;                                   // vectors per tile row
    for (uint32_t v = threadIdx.x; v < TW * VPR; v += 256) {
      const uint32_t lc = v / VPR, r8 = (v % VPR) * NV;                 // output column lc, output rows r8 .. r8 + NV - 1
      const int64_t o = base_offset + x_step * (int64_t)(C0 + lc) + y_step * (int64_t)(R0 + r8);
      const Vec x = *reinterpret_cast<const Vec *>(src + (y_step > 0 ? o : o - (int64_t)(NV - 1)));
      #pragma unroll
      for (uint32_t k = 0; k < NV; ++k) tile[lc * PITCH + r8 + (y_step > 0 ? k : NV - 1 - k)] = x.v[k];
    }
    __syncthreads();
    for (uint32_t v = threadIdx.x; v < TW * VPR; v += 256) {
      const uint32_t lr = v / VPR, c8 = (v % VPR) * NV;
      Vec x;
      #pragma unroll
      for (uint32_t k = 0; k < NV; ++k) x.v[k] = tile[(c8 + k) * PITCH + lr];
      *reinterpret_cast<Vec *>(dst + (size_t)(R0 + lr) * owidth + C0 + c8) = x;
    }
    return;
  }
  constexpr uint32_t RPP = 256 / TW;                                    // tile rows covered per pass of the 256 threads
  const uint32_t a = threadIdx.x % TW, b = threadIdx.x / TW;
  for (uint32_t i = 0; i < TW / RPP; ++i) {
    const uint32_t lr = a, lc = b + RPP * i;                            // lanes run along the output rows = consecutive source elements
    const uint32_t r = R0 + lr, c = C0 + lc;
    if (r < oheight && c < owidth) tile[lc * PITCH + lr] = src[base_offset + y_step * (int64_t)r + x_step * (int64_t)c];
  }
  __syncthreads();
  for (uint32_t i = 0; i < TW / RPP; ++i) {
    const uint32_t lc = a, lr = b + RPP * i;                            // lanes run along the output columns
    const uint32_t r = R0 + lr, c = C0 + lc;
    if (r < oheight && c < owidth) dst[(size_t)r * owidth + c] = tile[lc * PITCH + lr];
  }
}
template <typename T>
void launch_rotate1(const T *src, size_t owidth, size_t oheight, int64_t base_offset, int64_t x_step, int64_t y_step, T *dst, hipStream_t s) {
  constexpr size_t TW = 64;
  if ((y_step == 1 || y_step == -1) && x_step != 1 && x_step != -1 && (oheight + TW - 1) / TW <= 65535) {
    hipLaunchKernelGGL(k_rotate1_transposed<T>, dim3((unsigned)((owidth + TW - 1) / TW), (unsigned)((oheight + TW - 1) / TW), 1), dim3(256), 0, s,
                       src, (uint32_t)owidth, (uint32_t)oheight, base_offset, x_step, y_step, dst);
    return;
  }
  if (x_step == 1 || x_step == -1) {
    constexpr size_t NV = 16 / sizeof(T);
    hipLaunchKernelGGL(k_rotate1_rows<T>, grid_rows((owidth + NV - 1) / NV, oheight, 256), dim3(256), 0, s, src, (uint32_t)owidth, (uint32_t)oheight, base_offset, x_step, y_step, dst);
    return;
  }
  hipLaunchKernelGGL(k_rotate1<T>, grid_rows(owidth, oheight, 256), dim3(256), 0, s, src, (uint32_t)owidth, (uint32_t)oheight, base_offset, x_step, y_step, dst);
}
template void launch_rotate1<float>(const float *, size_t, size_t, int64_t, int64_t, int64_t, float *, hipStream_t);
template void launch_rotate1<uint16_t>(const uint16_t *, size_t, size_t, int64_t, int64_t, int64_t, uint16_t *, hipStream_t);
void launch_output8(const float *src, size_t n, uint8_t *dst, int num_cus, hipStream_t s) {
  hipLaunchKernelGGL(k_output8, dim3(grid_1d((n + 3) / 4, 256, flat_cap((unsigned)num_cus * 16))), dim3(256), 0, s, src, n, dst);
}
// The sample-depth changes of the raster fast path (src/pipeline.rs:381-402, :428-449): v * 257 and (v + 128) / 257, sixteen samples per lane -- 16-byte
// loads and stores (round 5: one sample per lane and access ran at 3.8 / 4.5 TB/s of its 3 bytes per sample; see k_rotate1_transposed for the same finding)
struct __attribute__((packed, aligned(1))) ChanB16 { uint32_t w[4]; };
struct __attribute__((packed, aligned(2))) ChanH8 { uint32_t w[4]; };
__global__ __launch_bounds__(256) void k_chan_8_to_16(const uint8_t *__restrict__ src, size_t n, uint16_t *__restrict__ dst) {
  const size_t i0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16u;
  if (i0 + 16u <= n) {
    const ChanB16 v = *reinterpret_cast<const ChanB16 *>(src + i0);
    ChanH8 lo, hi;
    #pragma unroll
    for (int k = 0; k < 4; ++k) {                          // b * 257 = the byte twice: bytes (0,0,1,1) and (2,2,3,3) of each input dword
      const uint32_t a = __builtin_amdgcn_perm(v.w[k], v.w[k], 0x01010000u), b = __builtin_amdgcn_perm(v.w[k], v.w[k], 0x03030202u);
      if (k < 2) { lo.w[2 * k] = a; lo.w[2 * k + 1] = b; } else { hi.w[2 * (k - 2)] = a; hi.w[2 * (k - 2) + 1] = b; }
    }
    *reinterpret_cast<ChanH8 *>(dst + i0) = lo; *reinterpret_cast<ChanH8 *>(dst + i0 + 8) = hi;
  } else {
    for (size_t i = i0; i < n; ++i) dst[i] = (uint16_t)(src[i] * 257u);
  }
}
__global__ __launch_bounds__(256) void k_chan_16_to_8(const uint16_t *__restrict__ src, size_t n, uint8_t *__restrict__ dst) {
  const size_t i0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16u;
  if (i0 + 16u <= n) {
    const ChanH8 lo = *reinterpret_cast<const ChanH8 *>(src + i0), hi = *reinterpret_cast<const ChanH8 *>(src + i0 + 8);
    ChanB16 o;
    #pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t w0 = k < 2 ? lo.w[2 * k] : hi.w[2 * (k - 2)], w1 = k < 2 ? lo.w[2 * k + 1] : hi.w[2 * (k - 2) + 1];
      const uint32_t q0 = ((w0 & 0xFFFFu) + 128u) / 257u, q1 = ((w0 >> 16) + 128u) / 257u, q2 = ((w1 & 0xFFFFu) + 128u) / 257u, q3 = ((w1 >> 16) + 128u) / 257u;
      o.w[k] = q0 | (q1 << 8) | (q2 << 16) | (q3 << 24);
    }
    *reinterpret_cast<ChanB16 *>(dst + i0) = o;
  } else {
    for (size_t i = i0; i < n; ++i) dst[i] = (uint8_t)(((uint32_t)src[i] + 128u) / 257u);
  }
}
void launch_chan_8_to_16(const uint8_t *src, size_t n, uint16_t *dst, int, hipStream_t s) {
  hipLaunchKernelGGL(k_chan_8_to_16, dim3((unsigned)((n + 4095) / 4096)), dim3(256), 0, s, src, n, dst);
}
void launch_chan_16_to_8(const uint16_t *src, size_t n, uint8_t *dst, int, hipStream_t s) {
  hipLaunchKernelGGL(k_chan_16_to_8, dim3((unsigned)((n + 4095) / 4096)), dim3(256), 0, s, src, n, dst);
}
void launch_output16(const float *src, size_t n, uint16_t *dst, int num_cus, hipStream_t s) {
  hipLaunchKernelGGL(k_output16, dim3(grid_1d((n + 3) / 4, 256, flat_cap((unsigned)num_cus * 16))), dim3(256), 0, s, src, n, dst);
}

// ------------------------------------------------------------------------------------------
// Fused raw -> sRGB for every colour filter without a fourth colour (RGGB phases: role formulas; others: generic-CFA mode):
//   OpGoFloat::run_raw (CFA branch) + demosaic::full + OpToLab + OpBaseCurve + OpFromLab + OpGamma
//   [+ output8bit/output16bit], one pass: 4 (or 2) bytes in, 12 (or 3/6) bytes out per pixel.
//
// Structure (MI355X-first, not a restatement of the reference's row-parallel loops):
//   * one 1024-thread workgroup per CU (4 waves per SIMD: measured optimum), PERSISTENT: it stays for the whole launch; both 13-bit tables
//     live in LDS (the Lab table as {v, dv} pairs where there is room) next to one staging buffer per wave and, in generic-CFA mode, the
//     pattern-cell records;
//   * no barriers after the table load: every WAVE owns a strip of 256 columns (4 consecutive pixels per lane => 16-byte
//     loads contiguous across the wave) and walks down a segment of rows, keeping a 3-row window of normalised samples
//     in registers; horizontal neighbours come from the adjacent lane by DPP wave shifts, the two strip-edge columns
//     from one 2-lane halo load per row; output rows go through the wave's LDS staging buffer so that the stores are
//     lane-contiguous 16-byte stores;
//   * each mosaic sample is therefore read from HBM once (plus 3 halo rows per segment and 2 halo columns per strip) and
//     normalised once;
//   * tasks = strip x row segment of ~32 rows; a wave takes the task of its own index first and DRAWS further ones from the launch stream's
//     queue (one atomic each), which evens out frames whose saturated regions make some tasks longer and leaves no wave waiting for the rest
//     of its block (fused_bayer_body, fused_task_grid);
//   * parameters that are the same for practically every raw file are template flags (CMN, PXG), because a
//     runtime-uniform flag is a scalar branch per row.
// ------------------------------------------------------------------------------------------
constexpr int kQueueStride = 32;    // 32-bit words between a stream's queue counter and its arrival counter: one 128-byte line each
struct FusedArgs {
  const void *src;            // element (row 0 of the slab, sensor column x) -- see row_off
  void *dst;                  // first output row (image row out_r0)
  uint32_t W, H;              // cropped image size
  uint64_t owidth;            // source row pitch (elements)
  uint32_t row_off;           // image row held by slab row 0
  uint32_t out_r0, out_r1;    // output rows [out_r0, out_r1)
  uint32_t n_frames;          // batch launches (k_fused_bayer_batch): frames behind the BatchPtrs argument
  uint32_t *task_ctr;         // the launch stream's task queue: tasks n_waves + *task_ctr, ... are still to be drawn; kQueueStride words behind it the
                              // arrival counter of the leaving waves.  Both zero when a launch starts and when it ends.  Null = no queue: one task per
                              // wave (task_counters_for, fused_task_grid)
  float min0, range0;         // blacklevels[0], whitelevels[0]-blacklevels[0]
  float inv_range0;           // RN(1/range0) for the 4-instruction division
  int exact_norm;             // 1: normalise with true divisions (host could not validate the fast form for range0)
  int fast_ok;                // 1: parameters are finite and ordinary, pointwise4_fast may run (else literal path only)
  int xoff, yoff;             // Bayer phase: color_at(r,c) = RGGB[(r+yoff)&1][(c+xoff)&1]
  ToLabParams tolab;
  Mat9 rgbm;                  // XYZ_D65_33
  int has_curve, linear;
  uint32_t n_strips, n_segs;  // task grid: strips x row segments
  uint32_t lc_base, lc_rem;   // lane-columns per strip: base (+1 for the first lc_rem strips)
  const float *lab_table;
  const float *gam_table;     // SRGB_GAMMA_TRANSFORM, 8193 plain floats
  const LutPair *lab_pairs, *gam_pairs;   // the two tables as 8192 {v, dv} pairs (the LDS image of the pair form)
  const Q8Entry *gam_q8;      // gamma + output8bit as 8192 {k, threshold} steps (ipk_device.hpp Q8Entry): the 8-bit-output variants' LDS image
  const float *gen_cells;     // generic-CFA mode: gen_pw*gen_ph cells of 36 floats (ipk_host.hpp Cfa::gen_cells), else null
  uint32_t gen_pw, gen_ph;    // pattern width / height (both divide 48)
  int gen_check;              // generic-CFA mode, u16 sources: 1 when the host could not show that every normalised sample is ordinary
  int px_guard;               // 0: u16 source with host-checked levels and parameters -> the variant without per-pixel input guards
  int ori;                    // ROT variants: the orientation (ipk_orientation) whose rotated space this launch works in
  int roles[4];               // ROT variants: demosaic role (0 R, 1 G on the R row, 2 G on the B row, 3 B) of the rotated-space pixel with
                              // parities (row & 1, column & 1) -> roles[2 * (row & 1) + (column & 1)]
  SplineDev spline;
};

// one image row as a lane sees it: its 4 samples and the neighbours left/right of them
struct RowWin { float l, v0, v1, v2, v3, r; };

struct __attribute__((packed, aligned(4))) f4u { float x, y, z, w; };
struct __attribute__((packed, aligned(1))) u32u { uint32_t v; };
struct __attribute__((packed, aligned(2))) u64u { uint32_t a, b; };
struct __attribute__((packed, aligned(4))) us4 { uint16_t x, y, z, w; };
struct __attribute__((packed, aligned(2))) us4u { uint16_t x, y, z, w; };
struct __attribute__((packed, aligned(4))) u3w { uint32_t x, y, z; };

template <typename T> struct RawRow { float v0, v1, v2, v3, h; };

__device__ __forceinline__ float dpp_wave_shr1(float old, float v) {   // lane i <- lane i-1 ; lane 0 keeps `old`
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_wave_shl1(float old, float v) {   // lane i <- lane i+1 ; lane 63 keeps `old`
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}

// demosaic::full for one pixel with explicit tap validity: the frame-edge path (demosaic.rs:99-114).
// PR/PC = parity of (row+yoff)/(col+xoff) in the RGGB tile, so every tap's colour is a compile-time constant;
// taps in the reference's order, sums start at 0.0, same-colour neighbours discarded (demosaic.rs:87).
template <int PR, int PC>
__device__ __forceinline__ float4 demosaic_edge_px(const float t[9], uint32_t valid_mask) {
  constexpr int center_code = (PR << 1) | PC;          // 0:R 1:G(r-row) 2:G(b-row) 3:B
  constexpr int center_color = center_code == 0 ? 0 : (center_code == 3 ? 2 : 1);
  float s[3] = {0.0f, 0.0f, 0.0f}, n[3] = {0.0f, 0.0f, 0.0f};
  #pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int dy = i / 3 - 1, dx = i % 3 - 1;
    const int tcode = (((PR + dy) & 1) << 1) | ((PC + dx) & 1);
    const int color = tcode == 0 ? 0 : (tcode == 3 ? 2 : 1);
    if ((color != center_color) || i == 4) {
      const bool ok = (valid_mask >> i) & 1u;
      s[color] = ok ? s[color] + t[i] : s[color];        // static index after unrolling
      n[color] = ok ? n[color] + 1.0f : n[color];
    }
  }
  // demosaic.rs:110-114; untouched channels stay 0.0
  return make_float4(n[0] > 0.0f ? s[0] / n[0] : 0.0f, n[1] > 0.0f ? s[1] / n[1] : 0.0f, n[2] > 0.0f ? s[2] / n[2] : 0.0f, 0.0f);
}
__device__ __forceinline__ float4 demosaic_edge_dispatch(const float t[9], uint32_t m, int pr, int pc) {
  if (pr == 0) return pc == 0 ? demosaic_edge_px<0, 0>(t, m) : demosaic_edge_px<0, 1>(t, m);
  return pc == 0 ? demosaic_edge_px<1, 0>(t, m) : demosaic_edge_px<1, 1>(t, m);
}

// Interior pixel, all nine taps valid: the four tile roles written out.  Sums start from 0.0 and
// add taps in the reference's order; /2 and /4 are exact as *0.5 / *0.25; count 1 is sum/1.0.
// Z = true keeps the reference's `sums[c] = 0.0; sums[c] += tap` literally (demosaic.rs:99-108).  Z = false starts each sum from its first
// tap: `0.0 + t` differs from `t` only for t = -0.0 (gives +0.0), so the two forms can differ only in the SIGN of an exactly-zero channel.
// The staged OpDemosaic (its output is the RGBE OpBuffer itself) uses Z = true; the fused kernel feeds the channels into
// camera_to_lab, where a zero's sign cannot reach a result: r*mul keeps a zero a zero, the matrix sums are +-0 or dominated by their
// nonzero terms, and the Lab lookup maps +0 and -0 to the same value (pointwise4_fast's `oor` test sends -0 down the linear branch,
// which yields table[0] bit for bit) -- checked by the parity tests with -0.0 samples and a zero black level (tests/test_gpu_fused.py).
template <int ROLE, bool Z = true>   // 0: R site, 1: G on the R row, 2: G on the B row, 3: B site
__device__ __forceinline__ float4 demosaic_inner_px(float nw, float n, float ne, float w, float c, float e, float sw, float s, float se) {
  const float zero = Z ? 0.0f : -0.0f;                       // -0.0 + t == t for every t (including -0.0 and NaN): the compiler folds it away
  const float own = zero + c;
  const float cross = ((((zero + n) + w) + e) + s) * 0.25f;
  const float diag = ((((zero + nw) + ne) + sw) + se) * 0.25f;
  const float horiz = ((zero + w) + e) * 0.5f;
  const float vert = ((zero + n) + s) * 0.5f;
  if (ROLE == 0) return make_float4(own, cross, diag, 0.0f);
  if (ROLE == 1) return make_float4(horiz, own, vert, 0.0f);
  if (ROLE == 2) return make_float4(vert, own, horiz, 0.0f);
  return make_float4(diag, cross, own, 0.0f);
}

// ---- rotated space (ROT variants) --------------------------------------------------------------------------------------
// The launch works on the mosaic already permuted into the output orientation (rotate_buffer's permutation, transform.rs:87-144,
// applied to the 1-channel sensor data instead of the 3-channel result), so that the output needs no permutation pass.  The
// demosaic must still add its taps in the reference's order, which is defined in the ORIGINAL orientation: the neighbour the
// reference calls (dy, dx) sits at a mapped offset in rotated space.  With (transpose, flip_x, flip_y) = to_flips(orientation):
// not transposed: (dy', dx') = (flip_y ? -dy : dy, flip_x ? -dx : dx); transposed: (dy', dx') = (flip_x ? -dx : dx, flip_y ? -dy : dy).
template <int ORI> struct OriFlips {   // ipk_orientation -> (transpose, flip_x, flip_y): Normal fff, HFlip ftf, Rot180 ftt, VFlip fft, Transpose tff, Rot90 tft, Transverse ttt, Rot270 ttf
  static constexpr bool t = ORI >= 4, fx = ORI == 1 || ORI == 2 || ORI == 6 || ORI == 7, fy = ORI == 2 || ORI == 3 || ORI == 5 || ORI == 6;
  static constexpr int ry(int y, int x) { return t ? (fx ? 2 - x : x) : (fy ? 2 - y : y); }   // rotated-space window row of the original tap (y, x)
  static constexpr int rx(int y, int x) { return t ? (fy ? 2 - y : y) : (fx ? 2 - x : x); }
};
// the 3 x 3 window of rotated-space pixel j (rows pw / cw / nw, columns j .. j+2) renamed into the original orientation's tap order
template <int ORI>
__device__ __forceinline__ void ori_window(const float pw[6], const float cw[6], const float nw[6], int j, float t[9]) {
  const float *rows[3] = {pw, cw, nw};
  #pragma unroll
  for (int y = 0; y < 3; ++y) {
    #pragma unroll
    for (int x = 0; x < 3; ++x) t[3 * y + x] = rows[OriFlips<ORI>::ry(y, x)][j + OriFlips<ORI>::rx(y, x)];
  }
}
template <int ORI>
__device__ __forceinline__ uint32_t ori_mask(uint32_t m_rot) {   // validity bits of the rotated-space window -> bits in the original tap order
  uint32_t m = 0;
  #pragma unroll
  for (int y = 0; y < 3; ++y) {
    #pragma unroll
    for (int x = 0; x < 3; ++x) m |= ((m_rot >> (3 * OriFlips<ORI>::ry(y, x) + OriFlips<ORI>::rx(y, x))) & 1u) << (3 * y + x);
  }
  return m;
}
template <int ROLE, int ORI>
__device__ __forceinline__ float4 demosaic_inner_ori(const float pw[6], const float cw[6], const float nw[6], int j) {
  float t[9];
  ori_window<ORI>(pw, cw, nw, j, t);
  return demosaic_inner_px<ROLE>(t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], t[8]);
}
// the four interior pixels of a lane in rotated space.  Along a rotated-space row of a transposing orientation the sensor column is
// fixed and the sensor row alternates (odd pixels: the even pixels' role with its row bit flipped), otherwise the other way round.
template <int RE, int ORI>
__device__ __forceinline__ void demosaic_rot_row4(const float pw[6], const float cw[6], const float nw[6], float4 px[4]) {
  constexpr int RO = RE ^ (OriFlips<ORI>::t ? 2 : 1);
  px[0] = demosaic_inner_ori<RE, ORI>(pw, cw, nw, 0); px[1] = demosaic_inner_ori<RO, ORI>(pw, cw, nw, 1);
  px[2] = demosaic_inner_ori<RE, ORI>(pw, cw, nw, 2); px[3] = demosaic_inner_ori<RO, ORI>(pw, cw, nw, 3);
}
template <int ORI>
__device__ __forceinline__ void demosaic_rot_row(int role_e, const float pw[6], const float cw[6], const float nw[6], float4 px[4]) {
  if (role_e == 0) demosaic_rot_row4<0, ORI>(pw, cw, nw, px);
  else if (role_e == 1) demosaic_rot_row4<1, ORI>(pw, cw, nw, px);
  else if (role_e == 2) demosaic_rot_row4<2, ORI>(pw, cw, nw, px);
  else demosaic_rot_row4<3, ORI>(pw, cw, nw, px);
}
// wave-uniform dispatch over the seven non-Normal orientations
__device__ __forceinline__ void demosaic_rot_row_dyn(int ori, int role_e, const float pw[6], const float cw[6], const float nw[6], float4 px[4]) {
  switch (ori) {
    case 1: demosaic_rot_row<1>(role_e, pw, cw, nw, px); break;
    case 2: demosaic_rot_row<2>(role_e, pw, cw, nw, px); break;
    case 3: demosaic_rot_row<3>(role_e, pw, cw, nw, px); break;
    case 4: demosaic_rot_row<4>(role_e, pw, cw, nw, px); break;
    case 5: demosaic_rot_row<5>(role_e, pw, cw, nw, px); break;
    case 6: demosaic_rot_row<6>(role_e, pw, cw, nw, px); break;
    default: demosaic_rot_row<7>(role_e, pw, cw, nw, px); break;
  }
}
__device__ __forceinline__ uint32_t ori_window_dyn(int ori, const float pw[6], const float cw[6], const float nw[6], int j, float t[9], uint32_t m_rot) {
  switch (ori) {
    case 1: ori_window<1>(pw, cw, nw, j, t); return ori_mask<1>(m_rot);
    case 2: ori_window<2>(pw, cw, nw, j, t); return ori_mask<2>(m_rot);
    case 3: ori_window<3>(pw, cw, nw, j, t); return ori_mask<3>(m_rot);
    case 4: ori_window<4>(pw, cw, nw, j, t); return ori_mask<4>(m_rot);
    case 5: ori_window<5>(pw, cw, nw, j, t); return ori_mask<5>(m_rot);
    case 6: ori_window<6>(pw, cw, nw, j, t); return ori_mask<6>(m_rot);
    default: ori_window<7>(pw, cw, nw, j, t); return ori_mask<7>(m_rot);
  }
}
__device__ __forceinline__ float4 demosaic_inner_role(int role, const float t[9]) {   // role is wave-uniform
  if (role == 0) return demosaic_inner_px<0>(t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], t[8]);
  if (role == 1) return demosaic_inner_px<1>(t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], t[8]);
  if (role == 2) return demosaic_inner_px<2>(t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], t[8]);
  return demosaic_inner_px<3>(t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], t[8]);
}

// ---- generic-CFA mode (X-Trans and any other filter without a fourth colour) --------------------------------------
// Interior pixel from the pattern cell's record (see Cfa::gen_cells): masked sums in the reference's tap order, then the
// proven two-step division by the tap count.  `cell` points at 36 floats, 16-byte aligned, in LDS.
constexpr int kGenCellFloats = 36;
constexpr int kGenMaxCells = 144;                        // 12 x 12
__device__ __forceinline__ float4 demosaic_gen_px(const float *__restrict__ cell, const float t[9]) {
  float w[kGenCellFloats];
  #pragma unroll
  for (int q = 0; q < 9; ++q) {
    const float4 v = reinterpret_cast<const float4 *>(cell)[q];
    w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
  }
  float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
  #pragma unroll
  // the weights are 0 or 1, so t * w is exact and the fused form rounds the same sum once: s + t*w bit for bit, in one instruction instead of two
  for (int i = 0; i < 9; ++i) { s0 = __builtin_fmaf(t[i], w[i], s0); s1 = __builtin_fmaf(t[i], w[9 + i], s1); s2 = __builtin_fmaf(t[i], w[18 + i], s2); }
  return make_float4(__builtin_fmaf(s0, w[28], s0 * w[31]), __builtin_fmaf(s1, w[29], s1 * w[32]), __builtin_fmaf(s2, w[30], s2 * w[33]), 0.0f);
}
// The literal form (demosaic.rs:99-114) from the packed tap colours: frame-edge pixels (taps outside the image are
// skipped) and rows whose samples are outside the zone where the arithmetic form is proven.
__device__ __forceinline__ float4 demosaic_gen_literal_px(uint32_t lk, const float t[9], uint32_t valid_mask) {
  float s[3] = {0.0f, 0.0f, 0.0f}, n[3] = {0.0f, 0.0f, 0.0f};
  #pragma unroll
  for (int i = 0; i < 9; ++i) {
    const uint32_t code = (lk >> (3 * i)) & 7u;
    const bool ok = (valid_mask >> i) & 1u;
    #pragma unroll
    for (int k = 0; k < 3; ++k) {
      const bool hit = ok && code == (uint32_t)k;
      s[k] = hit ? s[k] + t[i] : s[k];
      n[k] = hit ? n[k] + 1.0f : n[k];
    }
  }
  return make_float4(n[0] > 0.0f ? s[0] / n[0] : 0.0f, n[1] > 0.0f ? s[1] / n[1] : 0.0f, n[2] > 0.0f ? s[2] / n[2] : 0.0f, 0.0f);
}
// An "ordinary" normalised sample: zero, or finite with 2^-20 <= |v| <= 2^20 (every real sensor value is: the smallest
// nonzero (v - black)/range of a 16-bit sensor is about 2^-16).  The generic-CFA demosaic needs its rows made of such
// samples: sums of nine stay inside the division's proven zone, products with 0.0 are +-0.0, never NaN.
// (Tried and measured: using the same row check to drop pointwise4_fast's per-pixel guards -- 18 slow-class instructions
// per pixel pair -- left the u16 kernels unchanged and made the f32 kernels 3.7 % slower; the guards stay.)
__device__ __forceinline__ bool gen_sample_bad(float v) {
  const float a = __builtin_fabsf(v);
  return !(a <= 0x1p20f) || (a < 0x1p-20f && v != 0.0f);
}

struct PixOut { float r, g, b; };

constexpr float kRcLabK = 1.0f / kLabK;


// ---- pixel-pair helpers -----------------------------------------------------------------------------------
// The lane's pixels are processed in pairs so that two independent dependency chains interleave.  Packed f32
// instructions (v_pk_mul/add/fma_f32) were tried for the pairs and measured slower than scalar code on MI355X
// (tools/ubench.hip: v_mul/v_add 2.75 cycles per wave64 instruction, v_pk_* 4.8, v_fma 4.1), so the pair type is a
// plain struct and the translation unit is built with -fno-slp-vectorize.
struct f2 { float x, y; };
__device__ __forceinline__ f2 F2(float a, float b) { return f2{a, b}; }
__device__ __forceinline__ f2 S2(float a) { return f2{a, a}; }
__device__ __forceinline__ f2 operator*(f2 a, f2 b) { return f2{a.x * b.x, a.y * b.y}; }
__device__ __forceinline__ f2 operator+(f2 a, f2 b) { return f2{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ f2 operator-(f2 a, f2 b) { return f2{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ f2 operator-(f2 a) { return f2{-a.x, -a.y}; }
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return f2{__builtin_fmaf(a.x, b.x, c.x), __builtin_fmaf(a.y, b.y, c.y)}; }
__device__ __forceinline__ f2 min2(f2 a, float m) { return F2(rs_min(a.x, m), rs_min(a.y, m)); }

// 1/c = rc_hi + rc_lo.  q = fma(x, rc_hi, x*rc_lo) equals x/c for EVERY finite x with 2^-100 <= |x| <= 2^100 for the
// constants 0.95047, 1.08883, 100, 255, 116, 500, 200 (exhaustive on-device proof, tests/test_gpu_selftest.py
// test_cdiv_two_step_variant); it is NOT exact for 24389/27, which keeps the three-step residual form.
constexpr float rc_hi(float c) { return 1.0f / c; }
constexpr float rc_lo(float c) { return (float)(1.0 / (double)c - (double)(1.0f / c)); }
__device__ __forceinline__ f2 cdiv2s(f2 x, float hi, float lo) { return fma2(x, S2(hi), x * S2(lo)); }
__device__ __forceinline__ f2 cdiv3s(f2 x, float c, float rc) {
  const f2 q0 = x * S2(rc);
  const f2 r = fma2(-q0, S2(c), x);
  return fma2(r, S2(rc), q0);
}
__device__ __forceinline__ float cdiv3s1(float x, float c, float rc) {
  const float q0 = x * rc;
  const float r = __builtin_fmaf(-q0, c, x);
  return __builtin_fmaf(r, rc, q0);
}

// XYZ_LAB_TRANSFORM.lookup's direct branch for v > 1 (color_conversions.rs:103-104,123): cbrtf; the short form when the
// whole wave's out-of-table values are below 2
__device__ __forceinline__ float lab_cbrt(float v, bool hi) {
  if (__builtin_expect(__builtin_amdgcn_ballot_w64(hi && v >= 2.0f) == 0, 1)) return cbrtf_glibc_1to2(v);
  return cbrtf_glibc_sel(v);
}

// OpToLab -> OpBaseCurve -> OpFromLab -> OpGamma for the FOUR pixels of a lane, fast form.  Bit-identical to the literal
// form (pointwise_exact) whenever it returns false; returns true ("bad") for a lane whose inputs leave the zone where
// that equivalence is proven, and the caller then recomputes the lane's pixels literally.
//
// What differs from the literal evaluation, and why the bits are the same:
//  * x / c -> cdiv2s / cdiv3s (multiply + fma forms), proven exact on 2^-100 <= |x| <= 2^100 finite; dividends are
//    kept in that zone by (a) the channel sanity check below (finite inputs, host-validated finite parameters, so no
//    inf/NaN reaches a division), (b) explicit exponent guards on x and z (arbitrary sums) and on cl when a curve is
//    present, (c) the structure of the other dividends: l = 116*fy-16, a+127, b+127, cl+16, 116*f-16, A*255-127 are
//    differences against a constant >= 16, hence 0 or >= 2^-21 in magnitude, and a zero dividend gives +0 in both forms
//    (these differences are never -0; x and z can be -0 but the table lookup maps +-0 to the same value);
//  * the E channel term e*cm[3] is dropped: E is +0.0 for the RGGB tiles this kernel accepts and cm[3] is finite
//    (host-checked), so the term is +-0 and only changes (-0)+(+0), whose sign the lookup ignores as well;
//  * lookup(): key and weight via v_cvt_u32 / v_fract (proven == pos - trunc(pos) on [0,8192]); the table branch
//    runs for every lane and out-of-table values (bit pattern above 1.0f; -0 is also caught and takes the linear
//    branch, which yields table[0] bit-for-bit) are patched afterwards behind wave-uniform branches;
//  * v.max(0).min(1) -> v_med3_f32 (proven equal on every f32 up to the sign of zero, which the lookup ignores).
// `par` = LDS copy of the uniform parameters (mul[0..3], cm[4..15], rgbm[16..24]): read through the LDS they end up in
// vector registers instead of competing with the wave's many 64-bit condition masks for scalar registers.
// Written for the lane's FOUR pixels (two pairs) stage by stage, so that each table stage issues its 12 LDS reads
// together and pays their latency once (the wave-uniform branches of the out-of-table patch keep the compiler from
// interleaving two separate two-pixel evaluations: same speed on most boxes, 22 % faster on one with slow LDS/clock).
//
// PXG = false drops the two input guards (the channel sanity check and the exponent guards on x and z; 18 slow-class
// instructions per pixel pair, 4.6 % of the u16 kernel).  Legal when every sample v is zero or 2^-31 <= |v|, v >= -2^20
// (clipped above at 1.0; NaN clips to 1.0 as in Rust's min), the multipliers are in [2^-4, 2^10] and the nonzero matrix
// entries in [2^-12, 2^20] (host-checked): samples of 2^-31 or more are multiples of 2^-54, Bayer demosaic averages of
// 2^-56, white-balanced channels are at least 2^-60, their matrix products at least 2^-72 and hence multiples of 2^-95,
// so a nonzero x or z has 2^-95 <= |.| <= 2^42 -- inside the proven zone [2^-100, 2^100] of the constant divisions.
// Where the sample bounds come from: u16 sources -- the host walks all 65 536 values; f32 sources (CMN variants only) --
// |black| >= range/64 makes a nonzero v - black at least |black| * 2^-25 >= range * 2^-31, and finish_row's single
// comparison flags a row with a dividend below -2^20 * range; a row window holding a flagged row takes the literal form.
// TOLAB_ONLY: stop behind OpToLab and hand back the Lab pixels (L, A, B in r, g, b) -- the staged ipk_tolab on the same arithmetic.
template <bool PXG, bool TOLAB_ONLY = false, int NP = 2, typename LT, typename GT>   // NP pixel pairs per lane (2: four pixels)
__device__ __forceinline__ bool pointwise4_fast(const FusedArgs &a, const float *__restrict__ par, const LT *__restrict__ s_lab,
                                                const GT *__restrict__ s_gam, const float *__restrict__ s_knots,
                                                const float4 px[2 * NP], PixOut o[2 * NP], const bool has_curve, const bool linear, const int cm = 0,
                                                const float *__restrict__ par_regs = nullptr, const float *__restrict__ s_grid = nullptr) {
  // cm: 0 = the caller's flags are runtime values (generic variants, the chains); 1 / 2 = a common-parameter variant (fused_bayer_body's CM) whose curve is
  // the 3-knot arithmetic form / the grid form
  bool bad = false;
  float v[6 * NP], f[6 * NP];
  f2 y[NP];
  // the multipliers and the camera matrix as SCALAR operands straight from the kernel arguments (s_load through the constant cache, re-read where the
  // scalar registers are short) instead of sixteen broadcast reads of the LDS copy per row into vector registers
  const float spar[16] = {a.tolab.mul[0], a.tolab.mul[1], a.tolab.mul[2], a.tolab.mul[3], a.tolab.cm[0], a.tolab.cm[1], a.tolab.cm[2], a.tolab.cm[3], a.tolab.cm[4],
                          a.tolab.cm[5], a.tolab.cm[6], a.tolab.cm[7], a.tolab.cm[8], a.tolab.cm[9], a.tolab.cm[10], a.tolab.cm[11]};
  const float *const par0 = spar;
  #pragma unroll
  for (int g = 0; g < NP; ++g) {
    const float4 &pa = px[2 * g], &pb = px[2 * g + 1];
    if (PXG) bad |= !(fminf(fminf(pa.x, pa.y), pa.z) >= -0x1p40f) | !(fminf(fminf(pb.x, pb.y), pb.z) >= -0x1p40f);
    const f2 r = min2(F2(pa.x, pb.x) * S2(par0[0]), 1.0f);
    // (round 5, again: the common-parameter variants without green's `* 1.0` and `.min(1.0)` -- the multiplier is exactly 1.0 after normalize_wbs and a
    // demosaiced green is at most 1.0 and never NaN -- two instructions per pixel: the f32 variants go to 128 VGPRs + 1-2 spills whether the values are
    // passed plainly or through an empty asm, time +-1 %; the u16 -> u8 variant keeps its registers: 0.4250 -> 0.4214 ms.  Not kept.)
    const f2 gc = min2(F2(pa.y, pb.y) * S2(par0[1]), 1.0f);
    const f2 b = min2(F2(pa.z, pb.z) * S2(par0[2]), 1.0f);
    const f2 x = r * S2(par0[4]) + gc * S2(par0[5]) + b * S2(par0[6]);
    y[g] = r * S2(par0[8]) + gc * S2(par0[9]) + b * S2(par0[10]);
    const f2 z = r * S2(par0[12]) + gc * S2(par0[13]) + b * S2(par0[14]);
    if (PXG) bad |= cdiv_guard(x.x) | cdiv_guard(x.y) | cdiv_guard(z.x) | cdiv_guard(z.y);
    const f2 xr = cdiv2s(x, rc_hi(kWhiteX), rc_lo(kWhiteX));
    const f2 zr = cdiv2s(z, rc_hi(kWhiteZ), rc_lo(kWhiteZ));
    v[6 * g] = xr.x; v[6 * g + 1] = xr.y; v[6 * g + 2] = y[g].x; v[6 * g + 3] = y[g].y; v[6 * g + 4] = zr.x; v[6 * g + 5] = zr.y;
  }
  {
    float pos[6 * NP]; LutPair e[6 * NP];
    #pragma unroll
    for (int k = 0; k < 6 * NP; ++k) pos[k] = v[k] * kLutMaxF;
    #pragma unroll
    for (int k = 0; k < 6 * NP; ++k) e[k] = lut_pair_at(s_lab, f32_as_u32_sat(pos[k]));
    #pragma unroll
    for (int k = 0; k < 6 * NP; ++k) f[k] = e[k].x + __builtin_amdgcn_fractf(pos[k]) * e[k].y;
  }
  // (Tried and measured, round 1: compacting the v > 1 lanes of all 12 slots through a per-wave LDS queue -- ballot + mbcnt
  // ranks, cbrtf on dense groups of 64, results scattered back -- instead of one cbrtf per slot with most lanes idle.  On
  // uniform noise, where every slot has a few such lanes, it removes 7 % of the VALU instructions and 6 % of the time
  // (0.633 -> 0.595 ms); on fully saturated regions it costs 16 %, and every hybrid that keeps the in-place form for dense
  // slots or rows pays ~3 % on all other data for its extra scalar bookkeeping: photo-like +3.6 %, gradient +3.3 %.  Not kept.
  // Also measured without effect (+-1 %): one v_max3 tree + a single branch in front of the 12 per-slot checks; one explicit
  // s_waitcnt lgkmcnt(0) per table stage instead of the compiler's one per consumer.)
  // (Round 2, the queue retried in two forms, all twelve slots counted first (ballot + s_bcnt1), packed by mbcnt rank into the wave's idle
  // staging buffer when they fit 128 entries, slot-wise otherwise: noise 0.599 -> 0.668 / 0.739 ms, photo 0.482 -> 0.527, nothing-saturated
  // 0.465 -> 0.513 / 0.483: the counting alone -- twelve more ballots and scalar adds per wave-row -- costs more than the dense evaluation saves.)
  // (Round 2, the instruction mix per 256-pixel wave-row, tools/pmc_mix.sh: 920 VALU + 76 scalar + 62 branch/wait + 46 LDS + 6 memory = 1110 at 2.2
  // wave-cycles each.  ONE s_waitcnt lgkmcnt(0) per table stage (compiler-visible builtin + scheduling barrier; the plain table's subtraction
  // moved behind it) instead of the compiler's one per consumer removes 19 of them: noise 0.586 -> 0.594 ms, photo 0.470 -> 0.473 -- the
  // progressive waits let the first lerps start while the last reads are still in flight, and a wait that is already satisfied costs little.)
  // (Round 2, again without gain: ONE wave-level test -- the OR of the twelve compare masks -- in front of the per-slot tests: noise 0.591 ->
  // 0.621 ms, photo 0.473 -> 0.480, smooth 0.572 -> 0.595.)
  // One slot = one table-stage value of all 64 lanes.  Per slot that stays in the table this costs a compare and a branch; a slot with lanes above
  // 1 adds one compare, the cube root and one select, and the negative / NaN lanes (their own compare: as a bit pattern they are exactly the
  // values above +inf's) the linear branch.  (Round 2 derived the third mask from the first two -- hipcc moves such a mask through a VGPR to
  // branch on it -- and copied each mask into vcc: 36 instructions per entered slot, 27 now; the uniform noise frame enters 8 of 12 per row.)
  // (Round 3, two slots per test -- the x, y or z ratios of a pixel pair under one unsigned maximum and one compare: with the two cube roots side by
  // side, two independent f64 chains in flight, noise 0.494 -> 0.533 ms, photo-like 0.388 -> 0.383 (a pair is evaluated whenever either slot needs
  // it); with the slots evaluated one by one inside the pair's branch noise 0.511 -> 0.521, photo-like 0.400 -> 0.395, smooth 0.446 -> 0.450.  Not kept.)
  #pragma unroll
  for (int k = 0; k < 6 * NP; ++k) {
    if (IPK_RARE(__builtin_amdgcn_ballot_w64(__float_as_uint(v[k]) > 0x3F800000u) != 0)) {   // some lane has v > 1, v < 0, -0 or NaN
      const bool hi = v[k] > 1.0f;
      // the cube root under the lanes' own mask: the same instructions are issued, but only the lanes above 1 -- a tenth to a third of them on
      // the noise frame -- switch the f64 data path.  The kernel is bound by the socket's power cap, so what the idle lanes do not burn comes back as clock.
      // In the common-parameter variants (cm != 0); the linear branch for negative ratios below is masked only where there are no per-pixel
      // guards (PXG == false): with the guards' registers live as well BOTH masked regions spill (X-Trans full resolution 0.295 -> 0.338 ms with 36
      // bytes of scratch), the cube root's alone does not (X-Trans noise 0.305 -> 0.288 ms).
      if (cm != 0) { if (__builtin_amdgcn_ballot_w64(hi) != 0) { if (hi) f[k] = lab_cbrt(v[k], true); } }
      else if (__builtin_amdgcn_ballot_w64(hi) != 0) { const float c = lab_cbrt(v[k], hi); f[k] = hi ? c : f[k]; }
      const bool lo = __float_as_uint(v[k]) > 0x7F800000u;            // negative (or -0), or NaN: out of the table and not above 1
      if (!PXG && cm != 0) { if (__builtin_amdgcn_ballot_w64(lo) != 0) { if (lo) { const float dv = kLabK * v[k] + 16.0f; f[k] = __builtin_fmaf(dv, rc_hi(116.0f), dv * rc_lo(116.0f)); } } }
      else if (__builtin_amdgcn_ballot_w64(lo) != 0)
      { const float dv = kLabK * v[k] + 16.0f; const float t = __builtin_fmaf(dv, rc_hi(116.0f), dv * rc_lo(116.0f)); f[k] = lo ? t : f[k]; }
    }
  }
  f2 rr[NP], gg[NP], bb[NP];
  f2 Lq[NP], Aq[NP], Bq[NP];
  // the XYZ -> sRGB matrix (par[16..24]): read where LDS_ORDER puts the curve's reads, used at the very end
  float pm[9];
  {
    #pragma unroll
    for (int i = 0; i < 9; ++i) pm[i] = TOLAB_ONLY ? 0.0f : a.rgbm.m[i];
    #pragma unroll
    for (int g = 0; g < NP; ++g) {
      const f2 fx = F2(f[6 * g], f[6 * g + 1]), fy = F2(f[6 * g + 2], f[6 * g + 3]), fz = F2(f[6 * g + 4], f[6 * g + 5]);
      const f2 l = S2(116.0f) * fy - S2(16.0f);
      const f2 a0 = S2(500.0f) * (fx - fy);
      const f2 b0 = S2(200.0f) * (fy - fz);
      f2 L = cdiv2s(l, rc_hi(100.0f), rc_lo(100.0f));
      Aq[g] = cdiv2s(a0 + S2(127.0f), rc_hi(255.0f), rc_lo(255.0f));
      Bq[g] = cdiv2s(b0 + S2(127.0f), rc_hi(255.0f), rc_lo(255.0f));
      if (!TOLAB_ONLY && has_curve) {
        if (cm == 1) L = F2(spline_interpolate_3a(a.spline, s_knots, L.x), spline_interpolate_3a(a.spline, s_knots, L.y));
        else if (s_grid != nullptr && (cm == 2 || a.spline.grid_ok)) {
          // a NaN argument is the one input on which the grid form and the literal search differ (y_0 against the first probe's knot, curves.rs:144-149):
          // such a lane is flagged and redone literally, so bit-equality does not rest on L never being NaN in a lane whose result is kept
          bad |= (L.x != L.x) | (L.y != L.y);
          L = F2(spline_interpolate_grid(a.spline, s_grid, L.x), spline_interpolate_grid(a.spline, s_grid, L.y));
        }
        else L = F2(spline_interpolate_sel(a.spline, s_knots, L.x), spline_interpolate_sel(a.spline, s_knots, L.y));
      }
      Lq[g] = L;
    }
  }
  #pragma unroll
  for (int g = 0; g < NP; ++g) {
    const f2 L = Lq[g], A = Aq[g], B = Bq[g];
    if (TOLAB_ONLY) {
      o[2 * g].r = L.x; o[2 * g].g = A.x; o[2 * g].b = B.x; o[2 * g + 1].r = L.y; o[2 * g + 1].g = A.y; o[2 * g + 1].b = B.y;
      continue;
    }
    const f2 cl = L * S2(100.0f);
    const f2 ca = (A * S2(255.0f)) - S2(127.0f);
    const f2 cb = (B * S2(255.0f)) - S2(127.0f);
    const f2 gy = cdiv2s(cl + S2(16.0f), rc_hi(116.0f), rc_lo(116.0f));
    const f2 gx = cdiv2s(ca, rc_hi(500.0f), rc_lo(500.0f)) + gy;
    const f2 gz = gy - cdiv2s(cb, rc_hi(200.0f), rc_lo(200.0f));
    const f2 gx3 = gx * gx * gx, gy3 = gy * gy * gy, gz3 = gz * gz * gz;
    // lab_to_xyz's linear branches (color_conversions.rs:183-187: f^3 <= e, L* <= k*e -- only the darkest tones) are evaluated for a
    // channel only when some lane of the wave takes one in this pixel pair: 6 + 9 + 6 instructions per pixel (a tenth of the kernel's
    // arithmetic) that bright and mid-tone frames never need.  NaN compares false and lands in the branch, as in the literal form.
    const bool xb0 = gx3.x > kLabE, xb1 = gx3.y > kLabE, zb0 = gz3.x > kLabE, zb1 = gz3.y > kLabE;
    const bool yb0 = cl.x > kLabK * kLabE, yb1 = cl.y > kLabK * kLabE;
    f2 xq = gx3, yq = gy3, zq = gz3;
    // (Round 3: these three branches under the mask of the lanes that take them, as the cube root is -- noise 0.486 -> 0.483 ms, photo-like 0.400 -> 0.400,
    // smooth 0.463 -> 0.466: a few f32 instructions on a few lanes save nothing measurable.  Not kept.)
    if (IPK_RARE(__builtin_amdgcn_ballot_w64(!(xb0 && xb1)) != 0)) {
      const f2 lx = cdiv3s(S2(116.0f) * gx - S2(16.0f), kLabK, kRcLabK);
      xq = F2(xb0 ? gx3.x : lx.x, xb1 ? gx3.y : lx.y);
    }
    if (IPK_RARE(__builtin_amdgcn_ballot_w64(!(yb0 && yb1)) != 0)) {
      const float ly0 = __builtin_amdgcn_div_fixupf(cdiv3s1(cl.x, kLabK, kRcLabK), kLabK, cl.x);
      const float ly1 = __builtin_amdgcn_div_fixupf(cdiv3s1(cl.y, kLabK, kRcLabK), kLabK, cl.y);
      if (has_curve) bad |= (!yb0 & cdiv_guard(cl.x)) | (!yb1 & cdiv_guard(cl.y));
      yq = F2(yb0 ? gy3.x : ly0, yb1 ? gy3.y : ly1);
    }
    if (IPK_RARE(__builtin_amdgcn_ballot_w64(!(zb0 && zb1)) != 0)) {
      const f2 lz = cdiv3s(S2(116.0f) * gz - S2(16.0f), kLabK, kRcLabK);
      zq = F2(zb0 ? gz3.x : lz.x, zb1 ? gz3.y : lz.y);
    }
    const f2 X = xq * S2(kWhiteX), Y = yq, Z = zq * S2(kWhiteZ);
    rr[g] = X * S2(pm[0]) + Y * S2(pm[1]) + Z * S2(pm[2]);
    gg[g] = X * S2(pm[3]) + Y * S2(pm[4]) + Z * S2(pm[5]);
    bb[g] = X * S2(pm[6]) + Y * S2(pm[7]) + Z * S2(pm[8]);
  }
  if (TOLAB_ONLY) return bad;
  constexpr bool Q8 = std::is_same<GT, Q8Entry>::value;        // the 8-bit-output variants: gamma + output8bit as one step lookup, results as integers in float BITS
  if constexpr (Q8) {
    #pragma unroll
    for (int g = 0; g < NP; ++g) {
      const float x6[6] = {rr[g].x, rr[g].y, gg[g].x, gg[g].y, bb[g].x, bb[g].y};
      uint32_t q[6];
      if (!linear) {
        float c[6]; Q8Entry e[6];
        #pragma unroll
        for (int k = 0; k < 6; ++k) c[k] = __builtin_amdgcn_fmed3f(x6[k], 0.0f, 1.0f);
        #pragma unroll
        for (int k = 0; k < 6; ++k) e[k] = reinterpret_cast<const Q8Entry *>(s_gam)[f32_as_u32_sat(c[k] * kLutMaxF)];
        #pragma unroll
        for (int k = 0; k < 6; ++k) q[k] = e[k].k + (c[k] >= e[k].t ? 1u : 0u);
      } else {
        #pragma unroll
        for (int k = 0; k < 6; ++k) q[k] = __builtin_amdgcn_cvt_pk_u8_f32(floorf(x6[k] * 256.0f), 0u, 0u);   // == output8bit (ipk_selftest_quant8 variant 1)
      }
      rr[g] = F2(__uint_as_float(q[0]), __uint_as_float(q[1])); gg[g] = F2(__uint_as_float(q[2]), __uint_as_float(q[3])); bb[g] = F2(__uint_as_float(q[4]), __uint_as_float(q[5]));
    }
  } else {
  if (!linear) {
    float pos[6 * NP]; LutPair e[6 * NP];
    #pragma unroll
    for (int g = 0; g < NP; ++g) {
      const float c[6] = {__builtin_amdgcn_fmed3f(rr[g].x, 0.0f, 1.0f), __builtin_amdgcn_fmed3f(rr[g].y, 0.0f, 1.0f), __builtin_amdgcn_fmed3f(gg[g].x, 0.0f, 1.0f),
                          __builtin_amdgcn_fmed3f(gg[g].y, 0.0f, 1.0f), __builtin_amdgcn_fmed3f(bb[g].x, 0.0f, 1.0f), __builtin_amdgcn_fmed3f(bb[g].y, 0.0f, 1.0f)};
      #pragma unroll
      for (int k = 0; k < 6; ++k) pos[6 * g + k] = c[k] * kLutMaxF;
    }
    #pragma unroll
    for (int k = 0; k < 6 * NP; ++k) e[k] = lut_pair_at(s_gam, f32_as_u32_sat(pos[k]));
    #pragma unroll
    for (int g = 0; g < NP; ++g) {
      float w[6];
      #pragma unroll
      for (int k = 0; k < 6; ++k) w[k] = __builtin_amdgcn_fractf(pos[6 * g + k]);
      const LutPair *p = e + 6 * g;
      rr[g] = F2(p[0].x, p[1].x) + F2(w[0], w[1]) * F2(p[0].y, p[1].y);
      gg[g] = F2(p[2].x, p[3].x) + F2(w[2], w[3]) * F2(p[2].y, p[3].y);
      bb[g] = F2(p[4].x, p[5].x) + F2(w[4], w[5]) * F2(p[4].y, p[5].y);
    }
  }
  }
  #pragma unroll
  for (int g = 0; g < NP; ++g) {
    o[2 * g].r = rr[g].x; o[2 * g].g = gg[g].x; o[2 * g].b = bb[g].x;
    o[2 * g + 1].r = rr[g].y; o[2 * g + 1].g = gg[g].y; o[2 * g + 1].b = bb[g].y;
  }
  return bad;
}

// The literal per-pixel evaluation (device functions of ipk_device.hpp: true divisions, the reference's control flow).
template <typename LT, typename GT>
__device__ __forceinline__ PixOut pointwise_exact(const FusedArgs &a, const LT *__restrict__ s_lab, const GT *__restrict__ s_gam,
                                                  const float *__restrict__ s_knots, const float4 &p) {
  float l, ca, cb;
  camera_to_lab(s_lab, a.tolab, p.x, p.y, p.z, p.w, l, ca, cb);
  if (a.has_curve) l = spline_interpolate_lds(s_knots, a.spline.npoints, a.spline.nseg, l);
  PixOut o;
  lab_to_rgb(a.rgbm, l, ca, cb, o.r, o.g, o.b);
  if (!a.linear) {
    if constexpr (std::is_same<GT, Q8Entry>::value) {      // the LDS holds the 8-bit step table: this rare path reads the pair table where it lives
      o.r = gamma_sample_plain(a.gam_pairs, o.r); o.g = gamma_sample_plain(a.gam_pairs, o.g); o.b = gamma_sample_plain(a.gam_pairs, o.b);
    } else { o.r = gamma_sample_plain(s_gam, o.r); o.g = gamma_sample_plain(s_gam, o.g); o.b = gamma_sample_plain(s_gam, o.b); }
  }
  if constexpr (std::is_same<GT, Q8Entry>::value) {        // ... and hands back what the fast form does: the quantised samples in float bits
    o.r = __uint_as_float((uint32_t)output8bit(o.r)); o.g = __uint_as_float((uint32_t)output8bit(o.g)); o.b = __uint_as_float((uint32_t)output8bit(o.b));
  }
  return o;
}

template <typename SrcT, bool VEC>
__device__ __forceinline__ void load_raw4(const SrcT *p, uint32_t nvalid, float &v0, float &v1, float &v2, float &v3);
template <>
__device__ __forceinline__ void load_raw4<float, true>(const float *p, uint32_t nvalid, float &v0, float &v1, float &v2, float &v3) {
  // (round 6: these row loads nontemporal -- the skeleton 0.297 -> 0.315 ms at 100 MP, 0.0666 -> 0.081 at 24 MP, the kernel +0...4 %: the halo rows and a
  // frame launched again are re-reads the caches serve; gpurun_out ab_r06_ntld.txt)
  if (nvalid == 4) { const f4u t = *reinterpret_cast<const f4u *>(p); v0 = t.x; v1 = t.y; v2 = t.z; v3 = t.w; }
  else { v0 = nvalid > 0 ? p[0] : 0.0f; v1 = nvalid > 1 ? p[1] : 0.0f; v2 = nvalid > 2 ? p[2] : 0.0f; v3 = 0.0f; }
}
template <>
__device__ __forceinline__ void load_raw4<uint16_t, true>(const uint16_t *p, uint32_t nvalid, float &v0, float &v1, float &v2, float &v3) {
  if (nvalid == 4) { const us4 t = *reinterpret_cast<const us4 *>(p); v0 = (float)t.x; v1 = (float)t.y; v2 = (float)t.z; v3 = (float)t.w; }
  else { v0 = nvalid > 0 ? (float)p[0] : 0.0f; v1 = nvalid > 1 ? (float)p[1] : 0.0f; v2 = nvalid > 2 ? (float)p[2] : 0.0f; v3 = 0.0f; }
}
template <>
__device__ __forceinline__ void load_raw4<uint16_t, false>(const uint16_t *p, uint32_t nvalid, float &v0, float &v1, float &v2, float &v3) {
  if (nvalid == 4) { const us4u t = *reinterpret_cast<const us4u *>(p); v0 = (float)t.x; v1 = (float)t.y; v2 = (float)t.z; v3 = (float)t.w; }
  else { v0 = nvalid > 0 ? (float)p[0] : 0.0f; v1 = nvalid > 1 ? (float)p[1] : 0.0f; v2 = nvalid > 2 ? (float)p[2] : 0.0f; v3 = 0.0f; }
}

template <int OUT> struct OutStore;
template <> struct OutStore<3> { static __device__ __forceinline__ void store(void *, size_t, uint32_t, const PixOut *, bool) {} };
template <> struct OutStore<0> {   // f32 RGB (Pipeline::run)
  typedef float elem;
  static __device__ __forceinline__ void store(void *dst, size_t pix, uint32_t nvalid, const PixOut o[4], bool) {
    float *p = reinterpret_cast<float *>(dst) + pix * 3;
    if (nvalid == 4) {
      f4u a{o[0].r, o[0].g, o[0].b, o[1].r}, b{o[1].g, o[1].b, o[2].r, o[2].g}, c{o[2].b, o[3].r, o[3].g, o[3].b};
      reinterpret_cast<f4u *>(p)[0] = a; reinterpret_cast<f4u *>(p)[1] = b; reinterpret_cast<f4u *>(p)[2] = c;
    } else {
      #pragma unroll
      for (int j = 0; j < 3; ++j)
        if ((uint32_t)j < nvalid) { p[3 * j] = o[j].r; p[3 * j + 1] = o[j].g; p[3 * j + 2] = o[j].b; }
    }
  }
};
template <> struct OutStore<1> {   // u8 (output_8bit)
  typedef uint8_t elem;
  static __device__ __forceinline__ void store(void *dst, size_t pix, uint32_t nvalid, const PixOut o[4], bool aligned) {
    uint8_t *p = reinterpret_cast<uint8_t *>(dst) + pix * 3;
    uint8_t q[12];
    #pragma unroll
    for (int j = 0; j < 4; ++j) { q[3 * j] = output8bit(o[j].r); q[3 * j + 1] = output8bit(o[j].g); q[3 * j + 2] = output8bit(o[j].b); }
    if (nvalid == 4 && aligned) {
      u3w w;
      w.x = q[0] | (q[1] << 8) | (q[2] << 16) | ((uint32_t)q[3] << 24);
      w.y = q[4] | (q[5] << 8) | (q[6] << 16) | ((uint32_t)q[7] << 24);
      w.z = q[8] | (q[9] << 8) | (q[10] << 16) | ((uint32_t)q[11] << 24);
      *reinterpret_cast<u3w *>(p) = w;
    } else {
      #pragma unroll
      for (int j = 0; j < 12; ++j)
        if ((uint32_t)j < nvalid * 3) p[j] = q[j];
    }
  }
};
template <> struct OutStore<2> {   // u16 (output_16bit)
  typedef uint16_t elem;
  static __device__ __forceinline__ void store(void *dst, size_t pix, uint32_t nvalid, const PixOut o[4], bool aligned) {
    uint16_t *p = reinterpret_cast<uint16_t *>(dst) + pix * 3;
    uint16_t q[12];
    #pragma unroll
    for (int j = 0; j < 4; ++j) { q[3 * j] = output16bit(o[j].r); q[3 * j + 1] = output16bit(o[j].g); q[3 * j + 2] = output16bit(o[j].b); }
    if (nvalid == 4 && aligned) {
      uint32_t *pw = reinterpret_cast<uint32_t *>(p);
      #pragma unroll
      for (int j = 0; j < 6; ++j) pw[j] = q[2 * j] | ((uint32_t)q[2 * j + 1] << 16);
    } else {
      #pragma unroll
      for (int j = 0; j < 12; ++j)
        if ((uint32_t)j < nvalid * 3) p[j] = q[j];
    }
  }
};

// Staged, wave-coalesced output for the FULL kernel.  Each lane deposits its 4 pixels (NDW dwords) at
// stg[NDW*lane ...]; the wave's row segment is then NDW*64 contiguous dwords, written back as three
// instructions of NDW/3 dwords per lane with lane-contiguous addresses.  The buffer is private to the wave
// (LDS operations of one wave execute in order), so no barrier is involved.  `pix` = index of the strip's
// first pixel in the output.
template <int OUT> struct OutStage;
template <> struct OutStage<3> {   // unused: OUT == 3 leaves the kernel before the point-wise stages
  static __device__ __forceinline__ void stage(uint32_t *, uint32_t, const PixOut *) {}
  static __device__ __forceinline__ void flush(const uint32_t *, uint32_t, void *, size_t) {}
};
template <> struct OutStage<0> {   // f32: 12 dwords per lane, 3 x dwordx4 stores
  static __device__ __forceinline__ void stage(uint32_t *stg, uint32_t lane, const PixOut o[4]) {
    float4 *p = reinterpret_cast<float4 *>(stg + 12 * lane);
    p[0] = make_float4(o[0].r, o[0].g, o[0].b, o[1].r);
    p[1] = make_float4(o[1].g, o[1].b, o[2].r, o[2].g);
    p[2] = make_float4(o[2].b, o[3].r, o[3].g, o[3].b);
  }
  static __device__ __forceinline__ void flush(const uint32_t *stg, uint32_t lane, void *dst, size_t pix) {
    f4u *g = reinterpret_cast<f4u *>(reinterpret_cast<float *>(dst) + pix * 3);          // element-aligned 16-byte stores
    const float4 *s = reinterpret_cast<const float4 *>(stg);
    const float4 q0 = s[lane], q1 = s[64 + lane], q2 = s[128 + lane];
    // the result is written once and not read again by this launch: nontemporal stores keep it from displacing the mosaic rows in the caches
    // (element-aligned 16-byte stores, as below)
    typedef float f4a __attribute__((ext_vector_type(4), aligned(4)));
    f4a *gn = reinterpret_cast<f4a *>(reinterpret_cast<float *>(dst) + pix * 3);
    f4a v0, v1, v2;
    v0.x = q0.x; v0.y = q0.y; v0.z = q0.z; v0.w = q0.w; v1.x = q1.x; v1.y = q1.y; v1.z = q1.z; v1.w = q1.w; v2.x = q2.x; v2.y = q2.y; v2.z = q2.z; v2.w = q2.w;
    { __builtin_nontemporal_store(v0, gn + lane); __builtin_nontemporal_store(v1, gn + 64 + lane); __builtin_nontemporal_store(v2, gn + 128 + lane); }
    (void)g;
  }
};
template <> struct OutStage<1> {   // u8: 3 dwords per lane, 3 x dword stores
  // the samples arrive already quantised (integers 0..255 in the floats' bits: pointwise4_fast / pointwise_exact with the Q8Entry table)
  static __device__ __forceinline__ void stage_bits(uint32_t *stg, uint32_t lane, const PixOut o[4]) {
    auto B = [](float f) { return __float_as_uint(f); };
    uint32_t *p = stg + 3 * lane;
    p[0] = (B(o[0].r) | (B(o[0].g) << 8)) | ((B(o[0].b) | (B(o[1].r) << 8)) << 16);
    p[1] = (B(o[1].g) | (B(o[1].b) << 8)) | ((B(o[2].r) | (B(o[2].g) << 8)) << 16);
    p[2] = (B(o[2].b) | (B(o[3].r) << 8)) | ((B(o[3].g) | (B(o[3].b) << 8)) << 16);
  }
  static __device__ __forceinline__ void stage(uint32_t *stg, uint32_t lane, const PixOut o[4]) {
    uint32_t *p = stg + 3 * lane;
    p[0] = output8bit_x4(o[0].r, o[0].g, o[0].b, o[1].r);
    p[1] = output8bit_x4(o[1].g, o[1].b, o[2].r, o[2].g);
    p[2] = output8bit_x4(o[2].b, o[3].r, o[3].g, o[3].b);
  }
  static __device__ __forceinline__ void flush(const uint32_t *stg, uint32_t lane, void *dst, size_t pix) {
    u32u *g = reinterpret_cast<u32u *>(reinterpret_cast<uint8_t *>(dst) + pix * 3);        // byte-aligned dword stores
    const uint32_t q0 = stg[lane], q1 = stg[64 + lane], q2 = stg[128 + lane];
    typedef uint32_t u32a __attribute__((aligned(1)));
    u32a *gn = reinterpret_cast<u32a *>(reinterpret_cast<uint8_t *>(dst) + pix * 3);
    __builtin_nontemporal_store(q0, gn + lane); __builtin_nontemporal_store(q1, gn + 64 + lane); __builtin_nontemporal_store(q2, gn + 128 + lane);
    (void)g;
  }
};
template <> struct OutStage<2> {   // u16: 6 dwords per lane, 3 x dwordx2 stores
  static __device__ __forceinline__ void stage(uint32_t *stg, uint32_t lane, const PixOut o[4]) {
    uint2 *p = reinterpret_cast<uint2 *>(stg + 6 * lane);
    p[0] = make_uint2(output16bit_x2(o[0].r, o[0].g), output16bit_x2(o[0].b, o[1].r));
    p[1] = make_uint2(output16bit_x2(o[1].g, o[1].b), output16bit_x2(o[2].r, o[2].g));
    p[2] = make_uint2(output16bit_x2(o[2].b, o[3].r), output16bit_x2(o[3].g, o[3].b));
  }
  static __device__ __forceinline__ void flush(const uint32_t *stg, uint32_t lane, void *dst, size_t pix) {
    u64u *g = reinterpret_cast<u64u *>(reinterpret_cast<uint16_t *>(dst) + pix * 3);      // 2-byte-aligned 8-byte stores
    const uint2 *s = reinterpret_cast<const uint2 *>(stg);
    const uint2 q0 = s[lane], q1 = s[64 + lane], q2 = s[128 + lane];
    typedef uint32_t u2a __attribute__((ext_vector_type(2), aligned(2)));
    u2a *gn = reinterpret_cast<u2a *>(reinterpret_cast<uint16_t *>(dst) + pix * 3);
    u2a v0, v1, v2; v0.x = q0.x; v0.y = q0.y; v1.x = q1.x; v1.y = q1.y; v2.x = q2.x; v2.y = q2.y;
    __builtin_nontemporal_store(v0, gn + lane); __builtin_nontemporal_store(v1, gn + 64 + lane); __builtin_nontemporal_store(v2, gn + 128 + lane);
    (void)g;
  }
};

// OUT == 3: "demosaic only" -- the kernel skips normalisation and the point-wise stages and writes demosaic::full's
// 4-channel RGBE pixels (16 dwords per lane, four lane-contiguous dwordx4 stores).  This is the staged OpDemosaic for
// Bayer sources, sharing the row-walking skeleton with the fused kernel.
struct RgbeStage {
  static __device__ __forceinline__ void stage(uint32_t *stg, uint32_t lane, const float4 px[4]) {
    float4 *p = reinterpret_cast<float4 *>(stg + 16 * lane);
    p[0] = px[0]; p[1] = px[1]; p[2] = px[2]; p[3] = px[3];
  }
  static __device__ __forceinline__ void flush(const uint32_t *stg, uint32_t lane, void *dst, size_t pix) {
    f4u *g = reinterpret_cast<f4u *>(reinterpret_cast<float *>(dst) + pix * 4);
    const float4 *s = reinterpret_cast<const float4 *>(stg);
    const float4 q0 = s[lane], q1 = s[64 + lane], q2 = s[128 + lane], q3 = s[192 + lane];
    // (plain stores here: the RGBE buffer is the next staged kernel's input; nontemporal ones made this kernel 2.5 % slower, 422 -> 432 us)
    g[lane] = f4u{q0.x, q0.y, q0.z, q0.w}; g[64 + lane] = f4u{q1.x, q1.y, q1.z, q1.w}; g[128 + lane] = f4u{q2.x, q2.y, q2.z, q2.w}; g[192 + lane] = f4u{q3.x, q3.y, q3.z, q3.w};
  }
  static __device__ __forceinline__ void store_direct(void *dst, size_t pix, uint32_t nvalid, const float4 px[4]) {
    float4 *g = reinterpret_cast<float4 *>(dst) + pix;
    #pragma unroll
    for (int j = 0; j < 4; ++j)
      if ((uint32_t)j < nvalid) g[j] = px[j];
  }
};

// FULL = (W >= 256): every strip is 64 lanes x 4 pixels wide, so the hot path has no per-lane size logic; strips start
// on multiples of 256 pixels except the last, which is shifted left to end at the frame's last column (any W, any
// alignment: gfx950 takes dword/dwordx2/dwordx4 global accesses at element alignment).
// Everything that is rare (frame-edge pixels, out-of-table Lab values, dividends outside cdiv_fast's proven
// zone, the exact-division redo) sits behind a WAVE-UNIFORM branch (`ballot != 0`), which keeps the common
// path straight-line code the scheduler can interleave across the lane's 4 pixels.
// Occupancy: one 1024-thread block per CU = 4 waves per SIMD, on purpose.  Measured (tools/ubench2.hip, and this kernel's
// u16->u8 variant, which fits two blocks in LDS): at 8 waves per SIMD the simple f32 ops lose their 2-cycle issue rate
// (v_mul 1.0 -> 1.4 ns per wave64 instruction) and the kernel ran 23 % slower (0.79 -> 0.98 ms at 100 MP).
// CMN = the common parameter set is compiled in: fast point-wise form allowed, a base curve of 2 or 3 knots, the validated fast
// normalisation, gamma on unless the output is 16-bit (output_16bit forces linear).  Runtime-uniform flags cost scalar
// branches in the row loop; with them folded away the f32 kernel is 4 % faster.  Anything else runs the CMN = false variant.
// ROT = the launch works in rotated space (see OriFlips): any of the seven non-Normal orientations, wave-uniform dispatch on a.ori.
// the frames of a batch launch: up to kBatchMax per launch, passed by value as the second kernel argument (1 KB of kernarg)
constexpr int kBatchMax = 64;
struct BatchPtrs { const void *src[kBatchMax]; void *dst[kBatchMax]; };
template <typename SrcT, bool VEC, int OUT, bool FULL, bool GEN, bool PXG, int CM, bool ROT, bool BATCH>
__device__ __forceinline__ void fused_bayer_body(const FusedArgs &a, const BatchPtrs *bp) {
  constexpr bool CMN = CM != 0;                          // 1: common parameters with the 3-knot curve compiled in; 2: with the grid form (four or more knots)
  // f32 sources can hold denormal/huge samples: guard the normalisation's dividends.  u16 samples minus a
  // host-validated black level cannot leave the proven zone.
  constexpr bool DEMO = OUT == 3;                        // demosaic only (staged OpDemosaic)
  // OUT == 4: the kernel's memory skeleton as an entry point of its own (ipk_stream_probe) -- the same launch, task walk, row loads, normalisation,
  // demosaic, LDS staging and nontemporal f32 stores, with the point-wise stages left out: the demosaiced R, G, B leave as the three output channels.
  // 4 (2) bytes in and 12 out per pixel on the fused kernel's own access pattern: the ceiling its time is compared with (bench.py roofline.ceiling_ms).
  constexpr bool SKEL = OUT == 4;
  constexpr int OUTS = SKEL ? 0 : OUT;                   // the output layout (OutStage / OutStore) the variant writes
  constexpr bool ZA = DEMO;                                    // literal `0.0 + tap` sums where the demosaic result itself is the output
  constexpr bool GUARD_NORM = sizeof(SrcT) == 4 && !DEMO;
  // LDS: Lab table as {v,dv} pairs (64 KB), gamma table plain (32 KB; pairs when the output is 8/16-bit), curve knots, and one staging
  // buffer per wave (3 KB for f32) that turns the lane-blocked output (12 values per lane) into lane-interleaved 16-byte stores.
  // (the generic-CFA variants also hold their cell records in LDS and keep both tables plain)
  typedef typename std::conditional<GEN, float, LabTab>::type LabT;
  constexpr bool Q8 = OUTS == 1 && FULL && !GEN && !DEMO && !SKEL;     // 8-bit output (full strips): OpGamma + output8bit as one step lookup (Q8Entry, ipk_device.hpp)
  typedef typename std::conditional<Q8, Q8Entry, typename std::conditional<GEN || OUTS == 0, float, GamTab>::type>::type GamT;   // f32 output: 48 KB of staging, no room for two pair tables
  __shared__ __attribute__((aligned(16))) LabT s_lab[DEMO ? 4 : kLutPairs + 4];
  __shared__ __attribute__((aligned(16))) GamT s_gam[DEMO ? 4 : kLutPairs + 4];
  // the two tables go straight into LDS (no registers, see stage_lds_direct) and are in flight from here to the block's one barrier, which every wave
  // reaches in front of its first row's table reads (arrive() below)
  if (!DEMO) {
    stage_table_direct(s_lab, a.lab_table, a.lab_pairs);
    if constexpr (Q8) stage_q8_direct(s_gam, a.gam_q8); else stage_table_direct(s_gam, a.gam_table, a.gam_pairs);
  }
  __shared__ __attribute__((aligned(16))) float s_knots[kKnotFloats];   // base-curve knots + the 3-knot form's segment records
  __shared__ float s_par[32];                            // mul[0..3], cm[4..15], rgbm[16..24]
  __shared__ __attribute__((aligned(16))) float s_cells[GEN ? kGenMaxCells * kGenCellFloats : 4];   // generic-CFA cell records
  if (GEN) for (uint32_t i = threadIdx.x; i < a.gen_pw * a.gen_ph * kGenCellFloats; i += blockDim.x) s_cells[i] = a.gen_cells[i];
  constexpr int STG = DEMO ? 1024 : (OUTS == 0 ? 768 : (OUTS == 1 ? 192 : 384));   // dwords of staging per wave: one output row segment
  __shared__ __attribute__((aligned(16))) uint32_t s_stage[FULL ? 16 * STG : 4];
  if (threadIdx.x < 4) s_par[threadIdx.x] = a.tolab.mul[threadIdx.x];
  else if (threadIdx.x < 16) s_par[threadIdx.x] = a.tolab.cm[threadIdx.x - 4];
  else if (threadIdx.x < 25) s_par[threadIdx.x] = a.rgbm.m[threadIdx.x - 16];
  if (threadIdx.x < kSplineMaxKnots) fill_knots(s_knots, a.spline, (int)threadIdx.x);
  // the grid form of a curve with four or more knots (the common-parameter variants are compiled for three)
  constexpr bool HAS_GRID = CM != 1 && !DEMO && !SKEL;
  __shared__ __attribute__((aligned(16))) float s_grid[HAS_GRID ? kGridFloats : 4];
  if (HAS_GRID && (CM == 2 || a.spline.grid_ok)) fill_grid(s_grid, a.spline, (int)threadIdx.x);
  // what the block's waves are working on, for takeovers once the queue is dry (IPK_OPT_STEAL): per wave slot one 64-bit descriptor -- low word: the
  // row its task ends in front of; high word: serial << 22 | frame << 16 | strip -- and the row it is at
  __shared__ unsigned long long s_tdesc[16];
  __shared__ uint32_t s_tcur[16];
  // every wave clears its OWN slot (one wave's LDS operations run in order, so its first publication below cannot be overtaken by the clearing; other
  // waves look at the slot only behind the barrier)
  if ((threadIdx.x & 63u) == 0) { s_tdesc[threadIdx.x >> 6] = 0ull; s_tcur[threadIdx.x >> 6] = 0u; }
  // THE block barrier: tables, parameters, curve and cell records are in LDS behind it.  Every wave passes it exactly once -- in front of its first
  // row, with that row's loads already in flight, or on its way out when it has no row to do.
  bool arrived = false;
  auto arrive = [&]() { if (!arrived) { sync_after_lds_direct(); arrived = true; } };

  const uint32_t lane = threadIdx.x & 63u;
  // (making the task index wave-uniform with readfirstlane moves the row/address arithmetic to the scalar unit: measured, no change)
  // (XCD-aware variant measured: blocks are dealt round-robin to the 8 XCDs, each with its own L2; giving every XCD one contiguous
  // run of tasks -- block b -> run b % 8, position b / 8 -- so that vertically adjacent row segments share an L2 left the HBM fetch
  // at 453 -> 452 MB per launch: the two halo rows a segment shares with its neighbour are read ~0.1 ms apart and do not survive in
  // a 4 MB L2 that streams 1.6 GB.  Time unchanged on uniform data, +4 % on a frame whose saturated region then lands on one XCD.)
  // One task = one strip x row segment of one frame.  The launch is PERSISTENT -- one 1024-thread block per CU, sixteen waves -- and a wave DRAWS
  // its next task from a device counter when it is done with the last one (lane 0's atomic, broadcast).  Round 2 gave every wave exactly one task
  // and oversubscribed the CUs two blocks deep: a CU then waited for the slowest of a block's sixteen waves before it could take the next block (the
  // LDS holds one), and every block staged the lookup tables again -- 13 % of the 100 MP frame's time (0.587 -> 0.510 ms).  A BATCH launch (the frames
  // of ipk_raw_to_srgb_batch, same shape and parameters, pointers in the second kernel argument) queues the tasks of all its frames the same way.
  // The counter is the launch STREAM's (launches of one stream never overlap) and every launch leaves it zero: the last wave to leave -- an arrival
  // counter next to it -- resets both.  A launch is therefore self-contained: a launch that failed to enqueue, a captured launch replayed from a
  // graph, a stream used by two host threads all find a zeroed queue (round 2 alternated two counters per stream and relied on every launch running
  // exactly once, in issue order).  a.task_ctr == null (no queue slot could be had for this stream): one task per wave, nothing drawn.
  const uint32_t per_frame = a.n_strips * a.n_segs;
  const uint32_t n_waves = gridDim.x * (blockDim.x >> 6);
  const uint32_t n_tasks = BATCH ? per_frame * a.n_frames : per_frame;
  // Every wave starts with the task of its own index; further tasks are drawn with one atomic each.  Atomics on one address cost ~6 ns apiece
  // device-wide (8 XCDs): a first draw by all 4096 waves at once would add 25 us to a launch -- hence the static first round; the 4096 failing
  // draws at the end are spread over the last tasks' run time.  (Measured with dummy atomics on the same cache line: one more per draw takes
  // the 100 MP frame from 0.53 to 0.96 ms, four more to 1.24 -- 11-17 ns apiece once they queue up; the 12 480 draws of that frame keep the
  // counter's line busy for a quarter of the launch.)  (A coherent load in front of the atomic, to see an empty queue without touching
  // it, made every draw cost ~40 ns instead: 0.81 ms for the 100 MP frame.)
  // (Tried, round 2: two levels -- chunks of 16 tasks per atomic, dealt to a block's waves through LDS tickets: 16x fewer atomics, but the tasks a block
  // is sitting on cannot go to another block's idle waves: 100 MP noise 0.526 -> 0.629 ms, photo 0.414 -> 0.463.  And eight queues, one per XCD on its
  // own cache line, each holding every eighth task: 0.535 -> 0.554, 64 x 24 MP 7.83 -> 8.36 ms.)
  // (Round 3, tools/wave_timeline.py -- per-wave time stamps: the waves of a 100 MP launch are resident for 87 % (noise) / 82 % (photo-like) of its
  // span; the sixteen waves of a CU share its VALU unevenly -- the same 32 rows take one wave 113 us and another 347 -- and the last tenth of the
  // launch runs on 15 % / 4 % of the waves; a task costs 1.3 us of draw and 4.5 us of priming.  What that tail is worth was then measured by
  // removing it: bands that get shorter towards the end of the frame (32 -> 16 -> 8 -> 4 rows, a wave's even share of what is left), eight queue
  // heads with stealing and an LDS mark for heads found dry, the draw issued one row (or one task) ahead, the first four row loads of a task in
  // flight together.  Residency rose to 94 %, and nothing got faster: 0.515 -> 0.522-0.530 ms on noise, 0.412 -> 0.419-0.430 photo-like, 24 MP
  // 0.140 -> 0.150, 64 x 24 MP 7.5 -> 7.8 ms (same box, tools/gss_sweep.sh, profiles/r03_band_sweep.txt).  The kernel is bound by VALU issue: the
  // waves that remain in the tail simply run faster, so the tail wastes about 5 %, and twice the tasks cost that much in draws and priming.
  // Units of 4 rows numbered strip after strip (chunks of vertically adjacent units) were worse still: a block's sixteen waves then read sixteen 1 KB
  // pieces 1.4 MB apart instead of 16 KB of one row, rows ran 6-10 % slower.  Kept from all of it: the self-resetting queue.)
  auto draw = [&]() -> uint32_t {
    uint32_t t = 0xFFFFFFFFu;
    if ((threadIdx.x & 63u) == 0) {
      t = n_waves + atomicAdd(a.task_ctr, 1u);
    }
    t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
    return t;
  };
  const bool queued = a.task_ctr != nullptr && n_tasks > n_waves;
  // Takeovers (IPK_OPT_STEAL).  The sixteen waves of a block are served unevenly by their SIMDs (oldest first: the same 32 rows take one wave 113 us and
  // another 347), so when the queue runs dry the slow ones have most of a task ahead of them: the median wave of a 100 MP launch used to leave 59-74 us
  // before the launch ended, and its last tenth ran on 4-15 % of the waves.  A wave that finds the queue dry now looks at what the other waves of its block
  // have left (LDS: s_tdesc / s_tcur), and takes over the lower half of the rows that the one with the most has not begun: one 64-bit compare-and-swap on
  // that wave's descriptor moves its end row up, and the owner, which re-reads its end row once per row, stops there.  The swap succeeds only on the
  // descriptor the taker looked at (same task, same end), so two takers cannot both get the same rows, and a descriptor that has moved on is simply looked
  // at again.  An owner that was already past the new end when it moved computes those rows too: the same values written twice.  Costs a wave two LDS
  // operations per row; nothing is drawn, staged or primed that was not before, except the priming of the taken-over half.
  const uint32_t wslot = threadIdx.x >> 6, nwb = blockDim.x >> 6;
  // (the descriptor holds 16 bits of strip and 6 of frame: wider launches simply run without takeovers)
  const bool steal_on = a.n_strips <= 0xFFFFu && (!BATCH || a.n_frames <= 64u);
  uint32_t tserial = 0;
  auto take_over = [&](uint32_t &frame, uint32_t &strip, uint32_t &r0, uint32_t &r1) -> bool {
    for (int attempt = 0; attempt < 4; ++attempt) {
      unsigned long long d = 0ull; uint32_t c = 0u;
      if (lane < nwb) {
        d = __hip_atomic_load(&s_tdesc[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        c = __hip_atomic_load(&s_tcur[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      const uint32_t e = (uint32_t)d;
      const uint32_t rem = e > c + 1u ? e - c - 1u : 0u;   // rows behind the one the owner is at
      uint32_t best = 0u, bl = 0u;
      for (uint32_t i = 0; i < 16u; ++i) { const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)rem, (int)i); if (x > best) { best = x; bl = i; } }
      if (best < kStealMin) return false;
      const uint32_t dlo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)d, (int)bl), dhi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(d >> 32), (int)bl);
      const uint32_t mid = dlo - best / 2u;
      const unsigned long long dv = ((unsigned long long)dhi << 32) | dlo, nv = ((unsigned long long)dhi << 32) | mid;
      uint32_t ok = 0u;
      if (lane == 0) ok = atomicCAS(&s_tdesc[bl], dv, nv) == dv ? 1u : 0u;
      if (__builtin_amdgcn_readfirstlane((int)ok) != 0) { strip = dhi & 0xFFFFu; frame = (dhi >> 16) & 63u; r0 = mid; r1 = dlo; return true; }
    }
    return false;
  };
  // (everything that decides the control flow here is made wave-uniform FOR THE COMPILER -- readfirstlane -- or the row counter, the row parity and the
  // addresses of the task loop below end up in vector registers behind exec-mask loops: the first form of the takeovers cost 6.6 M integer instructions)
  // Static schedule (IPK_OPT_SPREAD = G): a block's sixteen waves start on 16 / G groups of G neighbouring tasks, the groups a whole round of blocks
  // apart, so that every block's share is a sample of the whole frame -- takeovers only even out what is inside a block -- while neighbouring strips
  // still share a block (their halo columns, the rows they read at the same time).  24 MP photo-like frame (saturated patches): G = 16 (none) / 8 / 4 / 2 / 1
  // 0.102 / 0.099 / 0.093 / 0.095 / 0.098 ms, noise 0.1215 / 0.1185 / 0.1221 / 0.1193 / 0.1225; 48 MP photo-like 0.193 / 0.192 / 0.184 / 0.187 / 0.189
  // (G = 1 on noise: 0.231 -> 0.247: every strip's neighbours on other XCDs).  G = 4.
  const uint32_t gt0 = queued ? blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)
                              : (((threadIdx.x >> 6) / kSpread) * gridDim.x + blockIdx.x) * kSpread + ((threadIdx.x >> 6) % kSpread);
  // Without a queue (a.task_ctr == null, or nothing to draw) a wave walks the tasks gt0, gt0 + n_waves, ...: with at most one task per wave that is the
  // single task of old, and a launch that holds more tasks than waves (no queue slot for the stream and a 64-frame batch, a GPU with fewer CUs, a frame
  // wider than 16K strips) still runs every one of them.
  auto next_task = [&](uint32_t gt) -> uint32_t {
    if (gt >= n_tasks) return 0xFFFFFFFFu;
    if (queued) return draw();
    return (n_tasks - gt > n_waves) ? gt + n_waves : 0xFFFFFFFFu;
  };
  for (uint32_t gt = (uint32_t)__builtin_amdgcn_readfirstlane((int)gt0);; gt = next_task(gt)) {   // whole waves enter and leave together
    uint32_t frame = 0u, strip, r0, r1;
    if (gt < n_tasks) {
      frame = BATCH ? (uint32_t)__builtin_amdgcn_readfirstlane((int)(gt / per_frame)) : 0u;
      // the task index is wave-uniform, and the compiler is told so: row counter, row parity and strip parity then live in scalar registers and the
      // demosaic's role dispatch is scalar branches instead of exec-mask regions (noise 0.529 -> 0.521 ms)
      const uint32_t task = (uint32_t)__builtin_amdgcn_readfirstlane((int)(BATCH ? gt - frame * per_frame : gt));
      strip = task % a.n_strips;
      const uint32_t seg = task / a.n_strips;
      // rows: this segment's output rows [r0, r1)
      const uint32_t nrows = a.out_r1 - a.out_r0;
      r0 = a.out_r0 + (uint32_t)(((uint64_t)seg * nrows) / a.n_segs);
      r1 = a.out_r0 + (uint32_t)(((uint64_t)(seg + 1) * nrows) / a.n_segs);
    } else {
      arrive();
      if (!steal_on) break;
      if (lane == 0) __hip_atomic_store(&s_tdesc[wslot], (unsigned long long)(++tserial & 0x3FFu) << 54, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // nothing left here
      if (__builtin_amdgcn_readfirstlane(take_over(frame, strip, r0, r1) ? 1 : 0) == 0) break;
    }
    strip = (uint32_t)__builtin_amdgcn_readfirstlane((int)strip); r0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)r0); r1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)r1);
    if (BATCH) frame = (uint32_t)__builtin_amdgcn_readfirstlane((int)frame);
    const void *const frame_src = BATCH ? bp->src[frame] : a.src;
    void *const frame_dst = BATCH ? bp->dst[frame] : a.dst;

    // columns: this strip's lane-columns [lc0, lc0+nl), 4 pixels each.  FULL: every strip is 64 lanes wide and the
    // last one is shifted left to stay inside the frame; the columns it shares with its neighbour are computed by
    // both waves (identical values written twice), which keeps every lane active and every access unpredicated.
    const uint32_t lc0 = FULL ? 0u : strip * a.lc_base + min(strip, a.lc_rem);
    const uint32_t nl = FULL ? 64u : a.lc_base + (strip < a.lc_rem ? 1u : 0u);
    const bool lane_on = FULL ? true : lane < nl;
    const uint32_t pc0 = FULL ? min(strip * 256u, a.W - 256u) : 4u * lc0;   // first pixel column of the strip
    // lanes past the strip shadow its last lane: their loads stay in bounds and need no predicate
    const uint32_t col0 = pc0 + 4u * min(lane, nl - 1);
    const uint32_t nvalid = FULL ? 4u : min(4u, a.W - col0);
    // column parity of the lane's pixel j inside the RGGB tile is (j + xo) & 1: col0 - pc0 is a multiple of 4
    const uint32_t xo = ((uint32_t)a.xoff + pc0) & 1u;
    // generic-CFA mode: the lane's four pattern-cell columns (float offsets into a row of cell records) never change
    uint32_t cxo[4] = {0, 0, 0, 0};
    if (GEN) {
      #pragma unroll
      for (int j = 0; j < 4; ++j) cxo[j] = ((col0 + j) % a.gen_pw) * kGenCellFloats;
    }
    // generic-CFA mode checks its rows for ordinary samples (gen_sample_bad); u16 sources skip the check when the host did it
    // for all 65 536 values
    // (a compile-time false for the Bayer variants, so that they carry no trace of it)
    const bool gen_guard = GEN && (sizeof(SrcT) == 4 || DEMO || a.gen_check != 0);
    if (r0 >= r1) { arrive(); continue; }
    if (steal_on && lane == 0) {
      // this wave's task, for takers.  Takers read the descriptor first and the row second, and swap only against the descriptor they read; so the old
      // descriptor is withdrawn BEFORE the new task's row is published (one wave's LDS operations run in order): a taker that still holds the old
      // descriptor can then never pair it with the new task's row -- its swap fails -- and one that reads the new descriptor reads a row of the new task.
      __hip_atomic_store(&s_tdesc[wslot], (unsigned long long)(++tserial & 0x3FFu) << 54, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_store(&s_tcur[wslot], r0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_store(&s_tdesc[wslot], ((unsigned long long)(((++tserial & 0x3FFu) << 22) | (frame << 16) | strip) << 32) | r1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }

    // halo columns of the strip: lane 0 fetches column 4*lc0-1, the last lane column 4*(lc0+nl); every other lane
    // (and a halo that would fall outside the frame) re-reads its own first sample so the load needs no predicate
    const bool is_first = lane == 0, is_last = lane + 1 == nl;
    const bool single = FULL ? false : nl == 1;            // one lane carries both halos: second one via h2
    const int64_t hcol_want = is_first ? (int64_t)pc0 - 1 : (int64_t)pc0 + 4 * (int64_t)nl;
    const uint32_t hcol = ((is_first || is_last) && hcol_want >= 0 && hcol_want < (int64_t)a.W) ? (uint32_t)hcol_want : col0;
    const uint32_t h2col = (single && pc0 + 4u * nl < a.W) ? pc0 + 4u * nl : col0;

    const SrcT *src = reinterpret_cast<const SrcT *>(frame_src);
    const float min0 = a.min0, range0 = a.range0, inv_range0 = a.inv_range0;
    const bool exact_norm = CMN ? false : a.exact_norm != 0;
    const bool fast_ok = CMN ? true : a.fast_ok != 0, has_curve = CMN ? true : a.has_curve != 0, linear = CMN ? (OUTS == 2) : a.linear != 0;

    // One image row is fetched in two steps so that the global loads of row r+2 are in flight while row r is
    // being computed: issue_row() only loads, finish_row() normalises (OpGoFloat) and gathers the horizontal
    // neighbours (DPP wave shifts + the strip's two halo columns).
    struct RawRowT { float v0, v1, v2, v3, h, h2; };
    auto issue_row = [&](uint32_t row) -> RawRowT {
      const SrcT *rp = src + (uint64_t)(row - a.row_off) * a.owidth;
      RawRowT t;
      load_raw4<SrcT, VEC>(rp + col0, nvalid, t.v0, t.v1, t.v2, t.v3);
      t.h = (float)rp[hcol];
      t.h2 = single ? (float)rp[h2col] : 0.0f;
      return t;
    };
    // `flag` (generic-CFA mode): some sample of this row is outside the zone of the arithmetic demosaic form (wave-uniform)
    auto finish_row = [&](const RawRowT &t, bool &flag) -> RowWin {
      flag = false;
      // OpGoFloat: ((v - black) / range).min(1.0)  (gofloat.rs:126).  The division is cdiv_fast unless the host
      // could not validate it for this range, or a dividend of this wave is outside the proven zone.
      const float d0 = t.v0 - min0, d1 = t.v1 - min0, d2 = t.v2 - min0, d3 = t.v3 - min0, dh = t.h - min0, dh2 = t.h2 - min0;
      RowWin w;
      float h, h2 = 0.0f;
      if (DEMO) {                                          // input is an OpBuffer already: no OpGoFloat step
        w.v0 = t.v0; w.v1 = t.v1; w.v2 = t.v2; w.v3 = t.v3;
        w.l = dpp_wave_shr1(t.h, w.v3);
        const float rr0 = dpp_wave_shl1(t.h, w.v0);
        w.r = is_last ? (single ? t.h2 : t.h) : rr0;
        if (gen_guard) flag = __builtin_amdgcn_ballot_w64(gen_sample_bad(t.v0) | gen_sample_bad(t.v1) | gen_sample_bad(t.v2) | gen_sample_bad(t.v3) |
                                                          gen_sample_bad(t.h) | gen_sample_bad(t.h2)) != 0;
        return w;
      }
      bool redo = exact_norm;
      // a dividend outside cdiv_fast's zone either clips to 1.0 (huge positive: exact in both forms) or gives a sample the row
      // check below rejects, which then redoes the row with true divisions -- so the row check replaces this one when it runs
      if (GUARD_NORM && !gen_guard) {
        if (CMN) {
          // CMN implies |black| >= 2^-70 (host-checked), so a nonzero v - black is at least half an ulp of black: no tiny
          // dividends.  A huge positive one clips to 1.0 whatever the division does; inf and NaN are v_div_fixup's.  That
          // leaves dividends below -2^100, one comparison on the minimum of the six.
          // Without per-pixel guards (PXG == false; f32 only with CMN) the same comparison also keeps every sample above -2^20,
          // and the host has checked |black| >= range/64, which puts every nonzero sample at 2^-31 or more (see pointwise4_fast).
          const float lowest = PXG ? -0x1p100f : -0x1p20f * range0;
          redo = __builtin_amdgcn_ballot_w64(!(fminf(fminf(fminf(d0, d1), fminf(d2, d3)), fminf(dh, dh2)) >= lowest)) != 0;
          if (!PXG) flag = redo;
        } else {
          redo = redo || __builtin_amdgcn_ballot_w64(cdiv_guard(d0) | cdiv_guard(d1) | cdiv_guard(d2) | cdiv_guard(d3) |
                                                      cdiv_guard(dh) | cdiv_guard(dh2)) != 0;
        }
      }
      if (!redo) {
        // common-parameter variants: cdiv_fast without its v_div_fixup (round 5: -0.9 % on the u16 variants, nothing measurable on f32).  The fixup only repairs zero, infinite and NaN dividends: 0 gives 0 either way; +inf,
        // NaN and dividends whose first product overflows give NaN or +inf, which .min(1.0) turns into the same 1.0; -inf and everything below the row
        // guard's bound never get here (redo), and u16 samples minus a validated black level are ordinary numbers.  (Not in generic-CFA mode: its row check
        // looks at the NORMALISED samples and must still see a -inf there.)
        auto nf = [&](float d) { const float q0 = d * inv_range0; return __builtin_fmaf(__builtin_fmaf(-q0, range0, d), inv_range0, q0); };
        if (CMN && !GEN) {
          w.v0 = rs_min(nf(d0), 1.0f); w.v1 = rs_min(nf(d1), 1.0f); w.v2 = rs_min(nf(d2), 1.0f); w.v3 = rs_min(nf(d3), 1.0f);
          h = rs_min(nf(dh), 1.0f);
          if (single) h2 = rs_min(nf(dh2), 1.0f);
        } else {
        w.v0 = rs_min(cdiv_fast(d0, range0, inv_range0), 1.0f); w.v1 = rs_min(cdiv_fast(d1, range0, inv_range0), 1.0f);
        w.v2 = rs_min(cdiv_fast(d2, range0, inv_range0), 1.0f); w.v3 = rs_min(cdiv_fast(d3, range0, inv_range0), 1.0f);
        h = rs_min(cdiv_fast(dh, range0, inv_range0), 1.0f);
        if (single) h2 = rs_min(cdiv_fast(dh2, range0, inv_range0), 1.0f);
        }
      } else {
        w.v0 = rs_min(d0 / range0, 1.0f); w.v1 = rs_min(d1 / range0, 1.0f); w.v2 = rs_min(d2 / range0, 1.0f); w.v3 = rs_min(d3 / range0, 1.0f);
        h = rs_min(dh / range0, 1.0f);
        if (single) h2 = rs_min(dh2 / range0, 1.0f);
      }
      if (gen_guard) {
        flag = __builtin_amdgcn_ballot_w64(gen_sample_bad(w.v0) | gen_sample_bad(w.v1) | gen_sample_bad(w.v2) | gen_sample_bad(w.v3) |
                                           gen_sample_bad(h) | gen_sample_bad(h2)) != 0;
        if (flag && !redo) {                               // rare: the row's samples again, literally
          w.v0 = rs_min(d0 / range0, 1.0f); w.v1 = rs_min(d1 / range0, 1.0f); w.v2 = rs_min(d2 / range0, 1.0f); w.v3 = rs_min(d3 / range0, 1.0f);
          h = rs_min(dh / range0, 1.0f);
          if (single) h2 = rs_min(dh2 / range0, 1.0f);
        }
      }
      w.l = dpp_wave_shr1(h, w.v3);                        // lane 0 keeps its halo (left column)
      const float rr = dpp_wave_shl1(h, w.v0);             // lane 63 keeps its halo
      w.r = is_last ? (single ? h2 : h) : rr;
      return w;
    };

    const bool store_aligned = (OUTS == 0 || OUTS == 3) ? true : ((OUTS == 1) ? ((a.W & 3u) == 0) : ((a.W & 1u) == 0));
    const uint32_t Hm1 = a.H - 1, Wm1 = a.W - 1;
    const bool col_edge = lane_on && (col0 == 0 || col0 + 3 >= Wm1);

    RowWin P = {0, 0, 0, 0, 0, 0}, C, N = {0, 0, 0, 0, 0, 0};
    bool fP = false, fC = false, fN = false;
    // A chunk starts with the loads of its first four rows in flight together and awaits them in turn: one memory round trip instead of three
    // (4.7 us per chunk, tools/wave_timeline.py).  The row above the frame's first row does not exist: row 0 is loaded in its place, for a
    // window every pixel of which takes the edge path.
    RawRowT raw_next;
    {
      const RawRowT rp = issue_row(r0 > 0 ? r0 - 1 : 0u), rc = issue_row(r0), rn = issue_row(min(r0 + 1, Hm1));
      raw_next = issue_row(min(r0 + 2, Hm1));
      P = finish_row(rp, fP); C = finish_row(rc, fC); N = finish_row(rn, fN);
    }
    arrive();
    uint32_t ry = GEN ? r0 % a.gen_ph : 0u;                // pattern row of image row r
    // The prefetch is unconditional (row index clamped to the frame): a branch around a load makes the compiler's
    // s_waitcnt bookkeeping assume the shortest path and wait for the previous iteration's stores as well.
    // Software pipeline: while row r is computed its window (P, C, N = rows r-1, r, r+1) is already in registers; row r+2
    // is *finished* (the wait for its loads) after the arithmetic of row r and BEFORE that row's stores are issued, and row
    // r+3 is *issued* after them.  gfx9 counts loads and stores in one vmcnt and the compiler must assume they complete
    // out of order, so a wait for loads with younger stores in flight becomes vmcnt(0) and exposes the full store latency
    // every iteration (it was 24 % of the wave's time); here the only stores older than the awaited loads are a whole
    // iteration old.
    // (the ~20 register moves that rotate the row window per iteration would vanish in a 3x unrolled loop; the compiler refuses
    // `#pragma unroll 3` here -- wave-level ballots and barriers in the body -- and a hand-unrolled body triples the code for ~1 %)
    // One row: demosaic + point-wise stages + store of row r from the window (P, C, N) = rows r-1, r, r+1; then row r+2 is finished INTO P's
    // registers (P is dead by then) and row r+3 issued.  The caller passes the three windows in rotating roles -- (P, C, N), (C, N, P), (N, P, C) --
    // so the window never moves between registers (IPK_OPT_UNROLL3; the rolling form copies 18 registers per row).
    // demosaic::full for the lane's four pixels of row r from the window (P, C, N) = rows r-1, r, r+1 (frame-edge pixels included)
    auto compute_px = [&](const RowWin &P, const RowWin &C, const RowWin &N, const bool fP, const bool fC, const bool fN, const uint32_t r, float4 px[4]) {
      const int pr = (int)((r + (uint32_t)a.yoff) & 1u);
      const float pw[6] = {P.l, P.v0, P.v1, P.v2, P.v3, P.r};
      const float cw[6] = {C.l, C.v0, C.v1, C.v2, C.v3, C.r};
      const float nw[6] = {N.l, N.v0, N.v1, N.v2, N.v3, N.r};
      // interior formulas; role = (row parity, column parity) in the RGGB tile; the column parity of pixel j is
      // (j + xo) & 1 with a per-strip xo: wave-uniform branches only.
      const float *rowcells = s_cells + ry * a.gen_pw * kGenCellFloats;
      if (GEN) {
        // any filter without a fourth colour: masked sums + proven division, or the literal form for a row window that
        // holds a sample outside the proven zone
        const bool literal = fP | fC | fN;
        #pragma unroll
        for (int j = 0; j < 4; ++j) {
          float t[9] = {pw[j], pw[j + 1], pw[j + 2], cw[j], cw[j + 1], cw[j + 2], nw[j], nw[j + 1], nw[j + 2]};
          if (ROT) (void)ori_window_dyn(a.ori, pw, cw, nw, j, t, 0x1FFu);   // rotated space: the cell records are the sensor pixel's, its taps in their original order
          if (!literal) px[j] = demosaic_gen_px(rowcells + cxo[j], t);
          else px[j] = demosaic_gen_literal_px(__float_as_uint(rowcells[cxo[j] + 27]), t, 0x1FFu);
        }
      } else if (ROT) {
        // rotated space: the role of a pixel comes from the host's table over (row, column) parity, its taps renamed into the
        // original orientation's order; both choices are wave-uniform
        demosaic_rot_row_dyn(a.ori, a.roles[2 * (int)(r & 1u) + (int)xo], pw, cw, nw, px);
      } else if (pr == 0) {
        if (xo == 0) {
          px[0] = demosaic_inner_px<0, ZA>(pw[0], pw[1], pw[2], cw[0], cw[1], cw[2], nw[0], nw[1], nw[2]);
          px[1] = demosaic_inner_px<1, ZA>(pw[1], pw[2], pw[3], cw[1], cw[2], cw[3], nw[1], nw[2], nw[3]);
          px[2] = demosaic_inner_px<0, ZA>(pw[2], pw[3], pw[4], cw[2], cw[3], cw[4], nw[2], nw[3], nw[4]);
          px[3] = demosaic_inner_px<1, ZA>(pw[3], pw[4], pw[5], cw[3], cw[4], cw[5], nw[3], nw[4], nw[5]);
        } else {
          px[0] = demosaic_inner_px<1, ZA>(pw[0], pw[1], pw[2], cw[0], cw[1], cw[2], nw[0], nw[1], nw[2]);
          px[1] = demosaic_inner_px<0, ZA>(pw[1], pw[2], pw[3], cw[1], cw[2], cw[3], nw[1], nw[2], nw[3]);
          px[2] = demosaic_inner_px<1, ZA>(pw[2], pw[3], pw[4], cw[2], cw[3], cw[4], nw[2], nw[3], nw[4]);
          px[3] = demosaic_inner_px<0, ZA>(pw[3], pw[4], pw[5], cw[3], cw[4], cw[5], nw[3], nw[4], nw[5]);
        }
      } else {
        if (xo == 0) {
          px[0] = demosaic_inner_px<2, ZA>(pw[0], pw[1], pw[2], cw[0], cw[1], cw[2], nw[0], nw[1], nw[2]);
          px[1] = demosaic_inner_px<3, ZA>(pw[1], pw[2], pw[3], cw[1], cw[2], cw[3], nw[1], nw[2], nw[3]);
          px[2] = demosaic_inner_px<2, ZA>(pw[2], pw[3], pw[4], cw[2], cw[3], cw[4], nw[2], nw[3], nw[4]);
          px[3] = demosaic_inner_px<3, ZA>(pw[3], pw[4], pw[5], cw[3], cw[4], cw[5], nw[3], nw[4], nw[5]);
        } else {
          px[0] = demosaic_inner_px<3, ZA>(pw[0], pw[1], pw[2], cw[0], cw[1], cw[2], nw[0], nw[1], nw[2]);
          px[1] = demosaic_inner_px<2, ZA>(pw[1], pw[2], pw[3], cw[1], cw[2], cw[3], nw[1], nw[2], nw[3]);
          px[2] = demosaic_inner_px<3, ZA>(pw[2], pw[3], pw[4], cw[2], cw[3], cw[4], nw[2], nw[3], nw[4]);
          px[3] = demosaic_inner_px<2, ZA>(pw[3], pw[4], pw[5], cw[3], cw[4], cw[5], nw[3], nw[4], nw[5]);
        }
      }
      // frame-edge pixels: taps outside the image are skipped, not mirrored (demosaic.rs:103-104)
      const bool edge_lane = (r == 0) || (r == Hm1) || col_edge;
      if (IPK_RARE(__builtin_amdgcn_ballot_w64(edge_lane) != 0)) {
        if (edge_lane) {
          #pragma unroll
          for (int j = 0; j < 4; ++j) {
            const uint32_t c = col0 + j;
            if (c < a.W && (r == 0 || r == Hm1 || c == 0 || c == Wm1)) {
              const float t[9] = {pw[j], pw[j + 1], pw[j + 2], cw[j], cw[j + 1], cw[j + 2], nw[j], nw[j + 1], nw[j + 2]};
              uint32_t m = 0x1FFu;
              if (r == 0) m &= ~0x007u;
              if (r == Hm1) m &= ~0x1C0u;
              if (c == 0) m &= ~0x049u;
              if (c == Wm1) m &= ~0x124u;
              if (ROT) {
                float to[9];
                const uint32_t mo = ori_window_dyn(a.ori, pw, cw, nw, j, to, m);
                const int role = a.roles[2 * (int)(r & 1u) + (int)((j + xo) & 1u)];
                px[j] = GEN ? demosaic_gen_literal_px(__float_as_uint(rowcells[cxo[j] + 27]), to, mo) : demosaic_edge_dispatch(to, mo, role >> 1, role & 1);
              } else
              px[j] = GEN ? demosaic_gen_literal_px(__float_as_uint(rowcells[cxo[j] + 27]), t, m) : demosaic_edge_dispatch(t, m, pr, (int)((j + xo) & 1u));
            }
          }
        }
      }
    };
    auto row_step = [&](RowWin &P, RowWin &C, RowWin &N, bool &fP, bool &fC, bool &fN, const uint32_t r) {
      // The four waves of a SIMD are served oldest first: left alone, the same 23 rows take one wave 83 us and another 164 (24 MP frame, one task per
      // wave, tools/wave_timeline.py) and the launch ends on a quarter of its waves.  Where nothing is drawn from a queue a wave's issue priority
      // therefore falls as it advances through its task, row by row in a cycle of four: whichever of a SIMD's waves is a row behind outranks the
      // others until it has caught up (lifetimes 121-165 us, 24 MP frame 0.141 -> 0.133 ms).  With a queue the unevenness is harmless -- the fast
      // waves simply draw more tasks -- and waves in step with each other wait for their table reads and stores at the same time: 100 MP frame
      // 0.519 -> 0.542 ms with the priorities on, so they stay off there.
      if (!queued) switch ((r - r0) & 3u) {
        case 0: __builtin_amdgcn_s_setprio(3); break;
        case 1: __builtin_amdgcn_s_setprio(2); break;
        case 2: __builtin_amdgcn_s_setprio(1); break;
        default: __builtin_amdgcn_s_setprio(0); break;
      }
      float4 px[4];
      compute_px(P, C, N, fP, fC, fN, r, px);
      if (DEMO) {
        bool fNN;
        const RowWin NN = finish_row(raw_next, fNN);       // row r+2 (clamped past the frame: those rows are masked as edges)
        __builtin_amdgcn_sched_barrier(0);
        if (FULL) {
          uint32_t *stg = s_stage + (threadIdx.x >> 6) * STG;
          RgbeStage::stage(stg, lane, px);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          RgbeStage::flush(stg, lane, frame_dst, (size_t)(r - a.out_r0) * a.W + pc0);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        } else if (lane_on) {
          RgbeStage::store_direct(frame_dst, (size_t)(r - a.out_r0) * a.W + col0, nvalid, px);
        }
        __builtin_amdgcn_sched_barrier(0);
        raw_next = issue_row(min(r + 3, Hm1));
        P = NN; fP = fNN;
        if (GEN) ry = (ry + 1 == a.gen_ph) ? 0u : ry + 1;
        return;
      }
      PixOut o[4];
      if (SKEL) {
        #pragma unroll
        for (int j = 0; j < 4; ++j) { o[j].r = px[j].x; o[j].g = px[j].y; o[j].b = px[j].z; }
      } else {
      bool bad = !fast_ok || (!PXG && sizeof(SrcT) == 4 && (fP | fC | fN));   // f32 without per-pixel guards: a flagged row in the window
      if (fast_ok) bad |= pointwise4_fast<PXG>(a, s_par, s_lab, s_gam, s_knots, px, o, has_curve, linear, CM, nullptr, HAS_GRID ? s_grid : nullptr);
      if (IPK_RARE(__builtin_amdgcn_ballot_w64(bad) != 0)) {          // rare: an input outside the fast form's proven zone
        #pragma unroll
        for (int j = 0; j < 4; ++j) {
          const PixOut e = pointwise_exact(a, s_lab, s_gam, s_knots, px[j]);
          if (bad) o[j] = e;
        }
      }
      }
      bool fNN;
      const RowWin NN = finish_row(raw_next, fNN);         // row r+2: the wait for its loads sits before this row's stores
      __builtin_amdgcn_sched_barrier(0);
      if (FULL) {
        // Lane-blocked -> lane-interleaved through the wave's LDS staging buffer, then three stores per lane whose
        // addresses are contiguous across the wave (whole cache lines per instruction; a 48-byte lane stride would
        // touch every line of the 3 KB span with each of its stores).
        uint32_t *stg = s_stage + (threadIdx.x >> 6) * STG;
        // The compiler reasons about one lane: a lane never reads back what it staged, so without these wave-scope
        // fences it treats the staging writes as dead stores / reorders them past the reads.  Wavefront-scope fences
        // and the wave barrier emit no instructions (the hardware already runs one wave's LDS operations in order).
        if constexpr (Q8) OutStage<1>::stage_bits(stg, lane, o); else OutStage<OUTS>::stage(stg, lane, o);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        OutStage<OUTS>::flush(stg, lane, frame_dst, (size_t)(r - a.out_r0) * a.W + pc0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      } else {
        if (lane_on) OutStore<OUTS>::store(frame_dst, (size_t)(r - a.out_r0) * a.W + col0, nvalid, o, store_aligned);
      }
      __builtin_amdgcn_sched_barrier(0);
      raw_next = issue_row(min(r + 3, Hm1));
      P = NN; fP = fNN;
      if (GEN) ry = (ry + 1 == a.gen_ph) ? 0u : ry + 1;
    };
    // (Round 5, the staged demosaic -- OUT == 3 -- with TWO rows per round trip: a four-row window, the two rows behind it in flight together, both rows'
    // eight stores behind one wait for loads (gfx9 counts loads and stores in one vmcnt, so a wave otherwise has one row of stores in flight): 365.5 / 367.0
    // -> 364.4 / 364.1 us at 100 MP on one box, nothing; the same kernel measured 406 us on another box the same day -- these memory-bound kernels move
    // by 10 % from box to box, not with their row loop.)
    // (Round 6, TWO raw rows in flight per wave -- the row loop unrolled by two with the rows' registers in alternating roles, no spills: the skeleton 0.3256 ->
    // 0.3313 ms, the kernel 0.4559 / 0.3882 -> 0.4437 / 0.3864 ms noise / photo-like on one pass and 0.4510 / 0.3844 -> 0.4564 / 0.3828 on the next
    // (gpurun_out ab_r06_pf2.txt): more loads in flight buy nothing, the walk is not waiting for latency.  hipcc also drains vmcnt to 0 at the first wait
    // behind a loop's back edge whatever is younger, so half the intended overlap never happened.  What the walk costs the memory system was then measured
    // without any arithmetic (tools/walk_probe.hip, profiles/README.md): 256-pixel strips 0.595 of the HBM peak, 512- or 1024-pixel strips 0.646, a flat
    // launch of the same traffic 0.71; tasks walked as an advancing band of rows 0.50-0.56.)
    for (uint32_t r = r0, r1d = r1; r < r1d; ++r) {
      uint32_t e_now = r1;
      if (steal_on) {                                       // where this wave is, and where its task ends by now (asked for here, looked at behind the row)
        if (lane == 0) __hip_atomic_store(&s_tcur[wslot], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        e_now = (uint32_t)__hip_atomic_load(reinterpret_cast<uint32_t *>(&s_tdesc[wslot]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      row_step(P, C, N, fP, fC, fN, r);
      { const RowWin t = P; P = C; C = N; N = t; const bool ft = fP; fP = fC; fC = fN; fN = ft; }   // back to rolling order
      if (steal_on) r1d = (uint32_t)__builtin_amdgcn_readfirstlane((int)e_now);
    }
  }
  arrive();
  // the last wave to leave zeroes the queue for the stream's next launch (every other wave's draws have returned before it arrived here)
  if (queued && (threadIdx.x & 63u) == 0) {
    uint32_t *const arrived = a.task_ctr + kQueueStride;
    if (atomicAdd(arrived, 1u) == n_waves - 1u) {
      __hip_atomic_store(a.task_ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(arrived, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
template <typename SrcT, bool VEC, int OUT, bool FULL, bool GEN, bool PXG = true, int CM = 0, bool ROT = false>
__global__ __launch_bounds__(1024) void k_fused_bayer(FusedArgs a) { fused_bayer_body<SrcT, VEC, OUT, FULL, GEN, PXG, CM, ROT, false>(a, nullptr); }
// the persistent batch form; instantiated for the common parameter set only (launch_fused_bayer_batch)
template <typename SrcT, bool VEC, int OUT, bool PXG>
__global__ __launch_bounds__(1024) void k_fused_bayer_batch(FusedArgs a, BatchPtrs bp) { fused_bayer_body<SrcT, VEC, OUT, true, false, PXG, 1, false, true>(a, &bp); }

static void fused_task_grid(FusedArgs &a, int num_cus, unsigned &blocks, uint32_t frames = 1, uint32_t waves_per_block = 16);
static bool task_counters_for(TaskQueues *q, hipStream_t s, FusedArgs &a, std::unique_lock<std::mutex> &lk);
// after hipLaunchKernelGGL: an enqueue error becomes the launcher's return value (-4); nothing is left to undo, the heads were not touched
static int launch_status() { return hipGetLastError() == hipSuccess ? 0 : -4; }

// Staged demosaic::full for an RGGB-phase Bayer mosaic (OUT == 3 of the row-walking kernel).  Same band arguments as the
// generic kernel: src row 0 = image row src_row0, output rows [out_row0, out_row0+out_rows).
// gen_cells != null: generic-CFA mode (pattern gen_pw x gen_ph, no fourth colour) instead of the RGGB phase (xoff, yoff).
int launch_demosaic_bayer(const float *src, size_t width, size_t img_height, size_t src_row0, size_t out_row0, size_t out_rows,
                          int xoff, int yoff, const float *gen_cells, int gen_pw, int gen_ph, float *dst4, int num_cus, TaskQueues *queues, hipStream_t s) {
  FusedArgs a;
  std::memset(&a, 0, sizeof(a));
  a.src = src; a.dst = dst4; a.W = (uint32_t)width; a.H = (uint32_t)img_height; a.owidth = width;
  a.row_off = (uint32_t)src_row0; a.out_r0 = (uint32_t)out_row0; a.out_r1 = (uint32_t)(out_row0 + out_rows);
  a.xoff = xoff; a.yoff = yoff; a.range0 = 1.0f; a.inv_range0 = 1.0f;
  a.gen_cells = gen_cells; a.gen_pw = (uint32_t)gen_pw; a.gen_ph = (uint32_t)gen_ph;
  unsigned blocks;
  std::unique_lock<std::mutex> queue_lock;
  (void)task_counters_for(queues, s, a, queue_lock);
  // the demosaic-only variants are memory-bound and small (78 VGPRs, 67 KB of LDS): two blocks per CU are launched, of which one is resident at a time
  // (round 4: blocks of eight waves, three resident per CU -- six waves per SIMD instead of four -- measured slower, 402 -> 416 us at 100 MP)
  // (1 / 2 / 3 / 4 / 6 blocks per CU launched: 0.416 / 0.415 / 0.422 / 0.417 / 0.426 ms at 100 MP)
  fused_task_grid(a, gen_cells ? num_cus : 2 * (num_cus > 0 ? num_cus : 256), blocks);
  if (gen_cells) {
    if (a.W >= 256u) hipLaunchKernelGGL((k_fused_bayer<float, true, 3, true, true>), dim3(blocks), dim3(1024), 0, s, a);
    else hipLaunchKernelGGL((k_fused_bayer<float, true, 3, false, true>), dim3(blocks), dim3(1024), 0, s, a);
  } else {
    if (a.W >= 256u) hipLaunchKernelGGL((k_fused_bayer<float, true, 3, true, false>), dim3(blocks), dim3(1024), 0, s, a);
    else hipLaunchKernelGGL((k_fused_bayer<float, true, 3, false, false>), dim3(blocks), dim3(1024), 0, s, a);
  }
  return launch_status();
}

template <typename SrcT, bool VEC, int OUT>
static void launch_fused_t(const FusedArgs &a, unsigned grid, hipStream_t s) {
  const unsigned tpb = 1024;
  const bool common_but_curve = a.fast_ok && a.has_curve && !a.exact_norm && (a.linear != 0) == (OUT == 2) && a.W >= 256u &&
                                std::fabs(a.min0) >= 0x1p-70f && std::fabs(a.min0) <= 0x1p70f;
  const bool common = common_but_curve && a.spline.npoints == 3 && spline3_arith_ok(a.spline);
  // the same parameter set with a curve of four or more knots that qualifies for the grid form (a user's edited base curve): the Bayer variants only
  const bool common_grid = common_but_curve && a.spline.grid_ok != 0 && a.ori == 0 && !a.gen_cells;
  if (a.ori != 0) {                                      // rotated space: the common parameter set only (launch_fused_bayer checked)
    if (a.gen_cells) { hipLaunchKernelGGL((k_fused_bayer<SrcT, sizeof(SrcT) == 4, OUT, true, true, true, true, true>), dim3(grid), dim3(tpb), 0, s, a); return; }
    if (a.px_guard == 0) hipLaunchKernelGGL((k_fused_bayer<SrcT, sizeof(SrcT) == 4, OUT, true, false, false, true, true>), dim3(grid), dim3(tpb), 0, s, a);
    else hipLaunchKernelGGL((k_fused_bayer<SrcT, sizeof(SrcT) == 4, OUT, true, false, true, true, true>), dim3(grid), dim3(tpb), 0, s, a);
    return;
  }
  if (a.gen_cells) {                                     // generic-CFA mode: one load flavour per source type
    constexpr bool V = sizeof(SrcT) == 4;
    if (common) hipLaunchKernelGGL((k_fused_bayer<SrcT, V, OUT, true, true, true, true>), dim3(grid), dim3(tpb), 0, s, a);
    else if (a.W >= 256u) hipLaunchKernelGGL((k_fused_bayer<SrcT, V, OUT, true, true>), dim3(grid), dim3(tpb), 0, s, a);
    else hipLaunchKernelGGL((k_fused_bayer<SrcT, V, OUT, false, true>), dim3(grid), dim3(tpb), 0, s, a);
    return;
  }
  // u16 sources with ordinary levels and parameters (the common case for real sensors): no per-pixel input guards
  if constexpr (sizeof(SrcT) == 2) if (a.px_guard == 0 && a.W >= 256u) {
    if (common) hipLaunchKernelGGL((k_fused_bayer<SrcT, false, OUT, true, false, false, true>), dim3(grid), dim3(tpb), 0, s, a);
    else if (common_grid) hipLaunchKernelGGL((k_fused_bayer<SrcT, false, OUT, true, false, false, 2>), dim3(grid), dim3(tpb), 0, s, a);
    else hipLaunchKernelGGL((k_fused_bayer<SrcT, false, OUT, true, false, false, false>), dim3(grid), dim3(tpb), 0, s, a);
    return;
  }
  if (common_grid) {
    if constexpr (sizeof(SrcT) == 4) if (a.px_guard == 0) {
      hipLaunchKernelGGL((k_fused_bayer<SrcT, true, OUT, true, false, false, 2>), dim3(grid), dim3(tpb), 0, s, a);
      return;
    }
    hipLaunchKernelGGL((k_fused_bayer<SrcT, sizeof(SrcT) == 4, OUT, true, false, true, 2>), dim3(grid), dim3(tpb), 0, s, a);
    return;
  }
  if (common) {
    if constexpr (sizeof(SrcT) == 4) if (a.px_guard == 0) {   // f32: ordinary parameters and a black level of at least range/64
      hipLaunchKernelGGL((k_fused_bayer<SrcT, true, OUT, true, false, false, true>), dim3(grid), dim3(tpb), 0, s, a);
      return;
    }
    hipLaunchKernelGGL((k_fused_bayer<SrcT, sizeof(SrcT) == 4, OUT, true, false, true, true>), dim3(grid), dim3(tpb), 0, s, a);
    return;
  }
  if (a.W >= 256u) hipLaunchKernelGGL((k_fused_bayer<SrcT, VEC, OUT, true, false>), dim3(grid), dim3(tpb), 0, s, a);
  else hipLaunchKernelGGL((k_fused_bayer<SrcT, VEC, OUT, false, false>), dim3(grid), dim3(tpb), 0, s, a);
}

// The task queue of a stream (fused_bayer_body): a counter and an arrival counter, zero between launches -- a launch leaves them as it found
// them, so nothing is reset from the host and a launch that never ran (an error at enqueue, a graph that is never replayed) costs nothing.  Every
// stream handle gets its own slot from one device block allocated at ipk_init (nothing is allocated at launch time, which stream capture forbids);
// hipStreamPerThread is a different stream in every host thread and is keyed by the calling thread.  When the table is full the least recently used
// slot whose stream has drained is reused; when there is none the launch runs the static schedule (task_ctr = null) -- the queue is an optimisation
// and never fails a call.
// One TaskQueues object per library context (ipk_ctx): its device block lives on the context's device, so two contexts -- on two GPUs, or two on one --
// never share a head, and launches of different contexts are issued under different locks.
constexpr uint32_t kCtrSlots = 1024;
constexpr uint32_t kCtrSlotWords = 2 * kQueueStride;
struct StreamCtr { hipStream_t key; hipStream_t stream; uint32_t slot; uint64_t last_use; };
struct TaskQueues {
  std::mutex mu;
  uint32_t *block = nullptr;
  std::vector<StreamCtr> of;
  uint64_t clock = 0;
};
static std::atomic<bool> g_ctr_disabled{false};            // test hook (ipk_selftest_task_queue): every launch runs as if no queue slot could be had
TaskQueues *create_task_queues() {
  TaskQueues *q = new TaskQueues;
  if (hipMalloc(reinterpret_cast<void **>(&q->block), (size_t)kCtrSlots * kCtrSlotWords * sizeof(uint32_t)) != hipSuccess) { delete q; return nullptr; }
  if (hipMemset(q->block, 0, (size_t)kCtrSlots * kCtrSlotWords * sizeof(uint32_t)) != hipSuccess) { (void)hipFree(q->block); delete q; return nullptr; }
  return q;
}
void destroy_task_queues(TaskQueues *q) {
  if (!q) return;
  { std::lock_guard<std::mutex> lk(q->mu); if (q->block) (void)hipFree(q->block); q->block = nullptr; q->of.clear(); }
  delete q;
}
void selftest_task_queue(bool enabled) { g_ctr_disabled.store(!enabled); }
// `lk` is held by the caller until its kernel launch has been issued: two host threads launching on one stream must enqueue one after the other.
// Returns false (a.task_ctr = null: static schedule) when no slot can be had.
static bool task_counters_for(TaskQueues *q, hipStream_t s, FusedArgs &a, std::unique_lock<std::mutex> &lk) {
  a.task_ctr = nullptr;
  if (!q) return false;
  lk = std::unique_lock<std::mutex>(q->mu);
  if (!q->block || g_ctr_disabled.load()) return false;
  static thread_local char per_thread_key;
  const hipStream_t key = (s == hipStreamPerThread) ? reinterpret_cast<hipStream_t>(&per_thread_key) : s;
  StreamCtr *e = nullptr;
  for (auto &c : q->of) if (c.key == key) { e = &c; break; }
  if (!e) {
    if (q->of.size() < kCtrSlots) {
      q->of.push_back({key, s, (uint32_t)q->of.size(), 0});
      e = &q->of.back();
    } else {
      // reuse the least recently used slot whose stream has nothing in flight (a destroyed stream's handle answers with an error: also free).  Slots of
      // per-thread streams cannot be queried from this thread and are passed over -- one of them at the head of the order used to block reuse for
      // good -- and a busy candidate is passed over as well: the next few in age order are tried before the launch settles for the static schedule.
      hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
      if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return false; }
      StreamCtr *lru = nullptr;
      uint64_t older_than = 0;
      for (int attempt = 0; attempt < 8 && !lru; ++attempt) {
        StreamCtr *cand = nullptr;
        for (auto &c : q->of) if (c.key == c.stream && c.last_use > older_than && (!cand || c.last_use < cand->last_use)) cand = &c;
        if (!cand) break;
        older_than = cand->last_use;
        const hipError_t qs = hipStreamQuery(cand->stream);
        (void)hipGetLastError();
        if (qs != hipErrorNotReady) lru = cand;
      }
      if (!lru) return false;
      lru->key = key; lru->stream = s;
      e = lru;
    }
  }
  e->last_use = ++q->clock;
  a.task_ctr = q->block + (size_t)e->slot * kCtrSlotWords;
  return true;
}

// Task grid: strips of <= 64 lane-columns x row segments.  A task costs its rows plus about 2.5 rows' worth of set-up (two halo rows, the three-row
// priming of the walker), and the queue should hold several tasks per wave for the draws to even anything out:
//   a wave's even share of the work is at least 128 rows of a strip for one frame (above ~130 MP), 64 for a batch (40 for both until round 3): segments of ~32 rows, drawn from the queue.
//     100 MP frame (tools/task_rows_sweep.sh, one box): 24 / 32 / 48 rows 0.531 / 0.535 / 0.584 ms noise, 0.417 / 0.419 / 0.439 photo; 12 / 16 / 20 rows
//     within 2 % of 32 on another; a tall first task per wave followed by short ones (50-80 % of the rows, then 8-24-row tasks) 0.531-0.551: no better
//     than uniform; ONE task per wave (nothing to draw) 0.576 / 0.497; round 2's launch (two blocks per CU in turn, one task per wave) 0.587 / 0.472;
//   smaller frames, or no queue for this stream (task_ctr == null): one task per wave, at least 8 rows (24 MP: 23 rows each, 0.148 ms either way).
//   Round 3, with takeovers inside a block (IPK_OPT_STEAL: a finished wave takes the lower half of what the slowest wave of its block has left): the static
//   schedule lost its tail -- 24 MP 0.134 -> 0.118 ms noise, 0.118 -> 0.099 photo-like -- and is the better one up to a share of ~64 rows: 48 MP (47 rows)
//   static 0.224 / 0.189 ms against drawn 0.242 / 0.195; 100 MP (98 rows) static 0.484 / 0.401 against drawn 0.490 / 0.382 -- uniform noise no longer needs
//   the queue, a frame whose regions differ (photo-like: saturated patches) does, because takeovers stay inside a block and only the queue moves work
//   between blocks (tools/share_min_sweep.sh, one box).  (Also tried with the takeovers in place, 100 MP: a tall static first task per wave over the first
//   50 / 65 / 80 % of the rows and 32- or 16-row drawn tasks for the rest -- a third fewer primings and draws: noise 0.492 -> 0.490 / 0.490 / 0.485 ms,
//   photo-like 0.391 -> 0.384 / 0.398 / 0.404 -- and drawn tasks of 24 / 48 / 64 / 96 rows: 0.497 / 0.508 / 0.496 / 0.522 against 0.501 for 32,
//   photo-like 0.407 / 0.397 / 0.406 / 0.464 against 0.395.  The uniform 32-row grid stays.)
//   And with the static round spread over the blocks in groups of four tasks (IPK_OPT_SPREAD, fused_bayer_body) a block's share is a sample of the whole
//   frame and the static schedule needs no queue for ONE frame up to a share of ~128 rows: 72 MP static 0.349 / 0.277 ms against drawn 0.353 / 0.285,
//   100 MP 0.479 / 0.382 against 0.487 / 0.384, 196 MP 0.928 / 0.730 against 0.929 / 0.714.  A batch keeps the queue from a share of 64 rows: one task per
//   wave would be thousands of rows of one frame (64 x 24 MP: 7.31 ms drawn, 8.81 static).
static void fused_task_grid(FusedArgs &a, int num_cus, unsigned &blocks, uint32_t frames, uint32_t waves_per_block) {
  const uint32_t uni = 32u, share_min = frames == 1 ? 128u : 64u;
  const uint32_t grid = (uint32_t)(num_cus > 0 ? num_cus : 256);
  const uint32_t total_waves = grid * waves_per_block;
  const uint32_t w4 = (a.W + 3) / 4;
  a.n_strips = (w4 + 63) / 64;
  a.lc_base = w4 / a.n_strips; a.lc_rem = w4 % a.n_strips;   // balanced strips (generic kernel); FULL uses 64-lane strips
  const uint32_t nrows = a.out_r1 - a.out_r0;
  const uint64_t share = (uint64_t)nrows * a.n_strips * frames / total_waves;   // rows of one strip a wave gets from an even split
  if (share >= share_min && a.task_ctr != nullptr) a.n_segs = std::max(1u, nrows / uni);
  else {
    // one task per wave, at least IPK_MIN_TASK_ROWS rows each
    const uint32_t min_rows = kMinTaskRows;
    a.n_segs = std::min(nrows, std::min(std::max(1u, (uint32_t)(total_waves / ((uint64_t)a.n_strips * frames))), std::max(1u, nrows / min_rows)));
  }
  const uint64_t tasks = (uint64_t)a.n_strips * a.n_segs * frames;
  // A small frame has fewer tasks than the chip has waves.  Packed sixteen to a block they left CUs idle while the waves of the busy ones shared their
  // SIMDs four deep (a 3 MP preview: 1620 tasks on 102 of 256 CUs, 0.045 ms); the spread static round (IPK_OPT_SPREAD) deals groups of four tasks to as
  // many blocks as there are groups, so such a launch now fills all CUs with one or two groups each, and waves without a task take over halves.
  const bool drawn = share >= share_min && a.task_ctr != nullptr;
  const uint64_t per_block = !drawn ? (uint64_t)kSpread : waves_per_block;
  blocks = (unsigned)std::min<uint64_t>(grid, (tasks + per_block - 1) / per_block);
}

int launch_fused_bayer(const FusedLaunch &f, hipStream_t s) {
  FusedArgs a;
  a.src = f.src; a.dst = f.dst; a.W = (uint32_t)f.width; a.H = (uint32_t)f.height; a.owidth = f.owidth;
  a.row_off = (uint32_t)f.row_off; a.out_r0 = (uint32_t)f.out_r0; a.out_r1 = (uint32_t)f.out_r1;
  a.min0 = f.black0; a.range0 = f.white0 - f.black0;                       // gofloat.rs:86-89
  a.inv_range0 = 1.0f / a.range0;
  a.exact_norm = f.exact_norm;
  a.fast_ok = f.fast_ok;
  a.xoff = f.xoff; a.yoff = f.yoff;
  a.tolab = make_tolab(f.mul4, f.cm12);
  for (int i = 0; i < 9; ++i) a.rgbm.m[i] = f.rgbm9[i];
  a.has_curve = f.has_curve; a.linear = f.linear;
  if (f.has_curve) a.spline = make_spline(*f.spline); else { a.spline.npoints = 0; a.spline.nseg = 0; a.spline.grid_ok = 0; }
  if (f.has_curve && a.spline.npoints == 2) {             // 2 knots padded to (x0, x1, x1): same decisions through the 3-knot form (ipk_device.hpp)
    a.spline.npoints = 3; a.spline.nseg = 2;
    a.spline.px[2] = a.spline.px[1]; a.spline.py[2] = a.spline.py[1];
    a.spline.c1[1] = a.spline.c1[0]; a.spline.c2[1] = a.spline.c2[0]; a.spline.c3[1] = a.spline.c3[0];
    a.spline.c1[2] = a.spline.c1[1];
  }
  a.lab_table = reinterpret_cast<const float *>(f.lab_table);
  a.gam_table = reinterpret_cast<const float *>(f.gam_table);
  a.lab_pairs = reinterpret_cast<const LutPair *>(f.lab_pairs); a.gam_pairs = reinterpret_cast<const LutPair *>(f.gam_pairs);
  a.gam_q8 = reinterpret_cast<const Q8Entry *>(f.gam_q8);
  a.gen_cells = f.gen_cells; a.gen_pw = (uint32_t)f.gen_pw; a.gen_ph = (uint32_t)f.gen_ph; a.gen_check = f.gen_check; a.px_guard = f.px_guard;
  a.ori = f.ori;
  for (int i = 0; i < 4; ++i) a.roles[i] = f.roles[i];
  if (f.ori != 0) {
    // rotated space exists for the common parameter set (the CMN variants) only;
    // anything else: the caller permutes the output instead
    const bool common = a.fast_ok && a.has_curve && a.spline.npoints == 3 && spline3_arith_ok(a.spline) && !a.exact_norm && (a.linear != 0) == (f.out_type == 2) && a.W >= 256u &&
                        std::fabs(a.min0) >= 0x1p-70f && std::fabs(a.min0) <= 0x1p70f;
    if (!common || f.ori < 1 || f.ori > 7) return -2;
  }

  const bool vec = f.src_is_u16 ? f.src_aligned4 : true;
  if (f.batch_n > 0) {
    // the multi-frame form of the kernel exists for the common parameter set without per-pixel guards (launch_fused_t's first choices for
    // real sensors); 64 x 512x512 frames: 2.6 us per frame in one launch against 13 us with a launch each
    const bool common = a.fast_ok && a.has_curve && a.spline.npoints == 3 && spline3_arith_ok(a.spline) && !a.exact_norm && (a.linear != 0) == (f.out_type == 2) && a.W >= 256u &&
                        std::fabs(a.min0) >= 0x1p-70f && std::fabs(a.min0) <= 0x1p70f;
    const bool batchable = common && a.ori == 0 && !a.gen_cells && a.px_guard == 0 && f.batch_n > 1;
    for (int i0 = 0; i0 < f.batch_n; i0 += kBatchMax) {
      const int n = std::min(kBatchMax, f.batch_n - i0);
      if (batchable && n > 1) {
        BatchPtrs bp;
        for (int i = 0; i < n; ++i) { bp.src[i] = f.batch_src[i0 + i]; bp.dst[i] = f.batch_dst[i0 + i]; }
        for (int i = n; i < kBatchMax; ++i) { bp.src[i] = nullptr; bp.dst[i] = nullptr; }
        a.n_frames = (uint32_t)n;
        unsigned grid;
        std::unique_lock<std::mutex> queue_lock;
        (void)task_counters_for(f.queues, s, a, queue_lock);
        fused_task_grid(a, f.num_cus, grid, (uint32_t)n);
#define IPK_BATCH_LAUNCH(T, V, O) hipLaunchKernelGGL((k_fused_bayer_batch<T, V, O, false>), dim3(grid), dim3(1024), 0, s, a, bp)
        if (!f.src_is_u16) { if (f.out_type == 0) IPK_BATCH_LAUNCH(float, true, 0); else if (f.out_type == 1) IPK_BATCH_LAUNCH(float, true, 1); else IPK_BATCH_LAUNCH(float, true, 2); }
        else { if (f.out_type == 0) IPK_BATCH_LAUNCH(uint16_t, false, 0); else if (f.out_type == 1) IPK_BATCH_LAUNCH(uint16_t, false, 1); else IPK_BATCH_LAUNCH(uint16_t, false, 2); }
#undef IPK_BATCH_LAUNCH
        if (const int rc = launch_status()) return rc;
      } else {
        FusedLaunch one = f;
        one.batch_n = 0;
        for (int i = 0; i < n; ++i) { one.src = f.batch_src[i0 + i]; one.dst = f.batch_dst[i0 + i]; const int rc = launch_fused_bayer(one, s); if (rc) return rc; }
      }
    }
    return 0;
  }
  std::unique_lock<std::mutex> queue_lock;
  (void)task_counters_for(f.queues, s, a, queue_lock);
  unsigned blocks;
  fused_task_grid(a, f.num_cus, blocks);
  // IPK_SCHED_SPLIT (include/imagepipe_amd.h): the one-task-per-wave schedule with every segment cut in two, walked statically -- wave i takes tasks i and
  // i + n_waves, which lie half a frame apart, so a region that costs more (blown highlights: the cube-root path) is shared by twice as many waves, at one
  // more priming per wave.  Round 5 measured it as a build flag (100 MP: photo-like 0.3812 -> 0.3683 / 0.3791 -> 0.3709 ms, noise 0.4600 -> 0.4648 /
  // 0.464 -> 0.458, smooth 0.4298 -> 0.4515: which regions a wave pairs is the luck of the frame) -- hence a caller's choice, not a default.
  if (f.schedule == 1 && a.n_strips * a.n_segs <= (uint32_t)(f.num_cus > 0 ? f.num_cus : 256) * 16u && (a.out_r1 - a.out_r0) / a.n_segs >= IPK_SPLIT_MIN_ROWS) {
    a.n_segs *= 2u; a.task_ctr = nullptr;
  }
  if (f.out_type == 4) {                                  // ipk_stream_probe: the skeleton of the headline variants (Bayer phase, full strips, no guards)
    if (a.gen_cells || a.ori != 0 || a.W < 256u || a.exact_norm || std::fabs(a.min0) < 0x1p-70f || std::fabs(a.min0) > 0x1p70f) return -2;
    if (!f.src_is_u16) hipLaunchKernelGGL((k_fused_bayer<float, true, 4, true, false, false, true>), dim3(blocks), dim3(1024), 0, s, a);
    else hipLaunchKernelGGL((k_fused_bayer<uint16_t, false, 4, true, false, false, true>), dim3(blocks), dim3(1024), 0, s, a);
    return launch_status();
  }
  if (!f.src_is_u16) {
    if (f.out_type == 0) launch_fused_t<float, true, 0>(a, blocks, s);
    else if (f.out_type == 1) launch_fused_t<float, true, 1>(a, blocks, s);
    else launch_fused_t<float, true, 2>(a, blocks, s);
  } else if (vec) {
    if (f.out_type == 0) launch_fused_t<uint16_t, true, 0>(a, blocks, s);
    else if (f.out_type == 1) launch_fused_t<uint16_t, true, 1>(a, blocks, s);
    else launch_fused_t<uint16_t, true, 2>(a, blocks, s);
  } else {
    if (f.out_type == 0) launch_fused_t<uint16_t, false, 0>(a, blocks, s);
    else if (f.out_type == 1) launch_fused_t<uint16_t, false, 1>(a, blocks, s);
    else launch_fused_t<uint16_t, false, 2>(a, blocks, s);
  }
  return launch_status();
}


// ------------------------------------------------------------------------------------------
// OpToLab + OpBaseCurve + OpFromLab + OpGamma in one pass over a 4-channel OpBuffer (the ops between demosaic /
// rotatecrop and transform): 16 bytes in, 12 out per pixel, HBM-bound.  Same per-pixel code as the fused raw kernel
// (pointwise4_fast, literal redo behind a wave-uniform branch).  A wave takes 256 consecutive pixels per step; lane L
// owns pixels L, 64+L, 128+L, 192+L of the chunk, so every load (16 B per lane) and store (12 B per lane) is contiguous
// across the wave and nothing needs staging.
// ------------------------------------------------------------------------------------------
// TOLAB_ONLY: OpToLab alone (the staged ipk_tolab, for callers that memoise the Lab buffer): same loads, same fast form with the literal
// redo behind a wave-uniform branch, stops behind xyz_to_lab; only the Lab table lives in LDS (32 KB: several blocks per CU).
// TOLAB_ONLY takes TWO pixels per lane (128-pixel chunks): with half the live values the compiler keeps the kernel to 64 vector registers, so that two
// blocks (66 KB of LDS each) share a CU -- eight waves per SIMD instead of four cover the load latency this kernel otherwise sits in.
// NPV = pixel PAIRS per lane of the full chain: 2 (256-pixel chunks), or 1 for small frames -- a preview of 3 MP is three chunks per wave, and its time is the
// launch, the table fill and the first and last round trips rather than the pixels: with six shorter chunks per wave the four waves of a SIMD fill and drain
// their pipeline in half the time (config 5's 2160x1440: 21.0 -> 20.3 us; at 24 and 100 MP the long chunks win by 1-3 %, gpurun_out ab_r06_chainnp1.txt).
template <bool TOLAB_ONLY, int NPV = 2>
__device__ __forceinline__ void pointwise_chain_body(const FusedArgs &a, uint64_t npix) {
  constexpr int NP = TOLAB_ONLY ? 1 : NPV, PPL = 2 * NP;                // pixel pairs / pixels per lane
  constexpr uint64_t CH = 64u * PPL;                                   // pixels per wave step
  __shared__ __attribute__((aligned(16))) LabTab s_lab[kLutPairs + 4];
  __shared__ __attribute__((aligned(16))) GamTab s_gam[TOLAB_ONLY ? 4 : kLutPairs + 4];
  __shared__ __attribute__((aligned(16))) float s_knots[kKnotFloats];
  __shared__ float s_par[32];
  // the tables go straight into LDS (stage_lds_direct: no registers; 1024-thread blocks) and stay in flight while the wave loads its first pixels: the
  // block's barrier sits behind those loads, so a block pays ONE memory round trip before its first pixel instead of two (sixteen blocks per CU slot)
  stage_table_direct(s_lab, a.lab_table, a.lab_pairs);
  if (!TOLAB_ONLY) stage_table_direct(s_gam, a.gam_table, a.gam_pairs);
  if (threadIdx.x < 4) s_par[threadIdx.x] = a.tolab.mul[threadIdx.x];
  else if (threadIdx.x < 16) s_par[threadIdx.x] = a.tolab.cm[threadIdx.x - 4];
  else if (threadIdx.x < 25) s_par[threadIdx.x] = a.rgbm.m[threadIdx.x - 16];
  if (threadIdx.x < kSplineMaxKnots) fill_knots(s_knots, a.spline, (int)threadIdx.x);
  __shared__ __attribute__((aligned(16))) float s_grid[TOLAB_ONLY ? 4 : kGridFloats];
  if (!TOLAB_ONLY && a.spline.grid_ok) fill_grid(s_grid, a.spline, (int)threadIdx.x);
  bool arrived = false;
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const uint64_t nwaves = (uint64_t)gridDim.x * (blockDim.x >> 6);
  const uint64_t nchunks = (npix + CH - 1) / CH;
  const float4 *src = reinterpret_cast<const float4 *>(a.src);
  f3 *dst = reinterpret_cast<f3 *>(a.dst);
  // (round 5: the next chunk's pixels loaded during this one's arithmetic need 71 registers; held to the 64 that two resident blocks allow, the
  // spills double the kernel's time, 477 -> 882 us)
  // (round 6: the NEXT chunk's loads issued behind this chunk's -- the full chain has the 16 registers to spare at one block per CU: 27.7 / 191.6 / 661 us
  // at 3.1 / 24 / 100 MP against 27.7 / 190.9 / 651 without, gpurun_out ab_r06_chainpf.txt: the four waves of a SIMD already cover each other's loads)
  for (uint64_t chunk = wave; chunk < nchunks; chunk += nwaves) {
    const uint64_t base = chunk * CH + lane;
    float4 px[PPL];
    #pragma unroll
    for (int j = 0; j < PPL; ++j) {
      const uint64_t i = base + 64u * j;
      px[j] = ld_stream4(reinterpret_cast<const float *>(src + (i < npix ? i : npix - 1)));   // clamped, unpredicated: the tail lanes recompute the last pixel
    }
    if (!arrived) { sync_after_lds_direct(); arrived = true; }          // the block's one barrier, behind the first chunk's loads
    PixOut o[PPL];
    // the fast form drops the E term (e * cm[i][3]): legal while the fourth channel is +0.0, as every producer on this
    // path writes it (gofloat's RGB/mono/raster branches, demosaic of RGB filters); RGBE mosaics take the literal form
    uint32_t wbits = 0u;
    #pragma unroll
    for (int j = 0; j < PPL; ++j) wbits |= __float_as_uint(px[j].w);
    bool bad = a.fast_ok == 0 || wbits != 0u;
    if (a.fast_ok) {
      bad |= pointwise4_fast<true, TOLAB_ONLY, NP>(a, s_par, s_lab, s_gam, s_knots, px, o, a.has_curve != 0, a.linear != 0, 0, nullptr, TOLAB_ONLY ? nullptr : s_grid);
    }
    if (__builtin_amdgcn_ballot_w64(bad) != 0) {
      #pragma unroll
      for (int j = 0; j < PPL; ++j) {
        PixOut e;
        if (TOLAB_ONLY) camera_to_lab(s_lab, a.tolab, px[j].x, px[j].y, px[j].z, px[j].w, e.r, e.g, e.b);
        else e = pointwise_exact(a, s_lab, s_gam, s_knots, px[j]);
        if (bad) o[j] = e;
      }
    }
    #pragma unroll
    for (int j = 0; j < PPL; ++j) {
      const uint64_t i = base + 64u * j;
      if (i < npix) { float *po = reinterpret_cast<float *>(dst + i); st_stream(po, o[j].r); st_stream(po + 1, o[j].g); st_stream(po + 2, o[j].b); }
    }
  }
  if (!arrived) sync_after_lds_direct();                                // a wave without a chunk
}
template <bool TOLAB_ONLY>
__global__ __launch_bounds__(1024) void k_pointwise_chain(FusedArgs a, uint64_t npix) { pointwise_chain_body<TOLAB_ONLY>(a, npix); }
__global__ __launch_bounds__(1024) void k_pointwise_chain_small(FusedArgs a, uint64_t npix) { pointwise_chain_body<false, 1>(a, npix); }
// OpToLab alone: two resident blocks per CU (eight waves per SIMD) are the point of its two-pixel form, so the register budget is stated (64)
template <>
__global__ __launch_bounds__(1024, 8) void k_pointwise_chain<true>(FusedArgs a, uint64_t npix) { pointwise_chain_body<true>(a, npix); }
static FusedArgs chain_args(const FusedLaunch &f) {
  FusedArgs a;
  std::memset(&a, 0, sizeof(a));
  a.src = f.src; a.dst = f.dst;
  a.fast_ok = f.fast_ok;
  a.tolab = make_tolab(f.mul4, f.cm12);
  for (int i = 0; i < 9; ++i) a.rgbm.m[i] = f.rgbm9[i];
  a.has_curve = f.has_curve; a.linear = f.linear;
  if (f.has_curve) a.spline = make_spline(*f.spline); else { a.spline.npoints = 0; a.spline.nseg = 0; }
  a.lab_table = reinterpret_cast<const float *>(f.lab_table);
  a.gam_table = reinterpret_cast<const float *>(f.gam_table);
  a.lab_pairs = reinterpret_cast<const LutPair *>(f.lab_pairs); a.gam_pairs = reinterpret_cast<const LutPair *>(f.gam_pairs);
  return a;
}
int launch_pointwise_chain(const FusedLaunch &f, size_t npix, hipStream_t s) {
  FusedArgs a = chain_args(f);
  const unsigned cus = (unsigned)(f.num_cus > 0 ? f.num_cus : 256);
  if (npix < (size_t)cus * 16u * 256u * 6u) {                              // fewer than six long chunks per wave (6.3 MP on 256 CUs): the two-pixel form
    const size_t chunks = (npix + 127) / 128;
    hipLaunchKernelGGL(k_pointwise_chain_small, dim3((unsigned)std::max<size_t>(1, std::min<size_t>(cus, (chunks + 15) / 16))), dim3(1024), 0, s, a, (uint64_t)npix);
    return 0;
  }
  const size_t chunks = (npix + 255) / 256;
  const unsigned cap = (unsigned)(f.num_cus > 0 ? f.num_cus : 256);                      // (more blocks measured: 1 / 2 / 4 / 8 / 16 per CU 0.663 / 0.665 / 0.677 / 0.673 / 0.704 ms at 100 MP)
  const unsigned blocks = (unsigned)std::min<size_t>(cap, (chunks + 15) / 16);
  hipLaunchKernelGGL(k_pointwise_chain<false>, dim3(blocks ? blocks : 1), dim3(1024), 0, s, a, (uint64_t)npix);
  return 0;
}
// OpToLab::run on the fast form (the staged op): f.fast_ok / f.mul4 / f.cm12 / f.lab_table as for the chain
int launch_tolab_fast(const FusedLaunch &f, size_t npix, hipStream_t s) {
  FusedArgs a = chain_args(f);
  const size_t chunks = (npix + 127) / 128;                                               // two pixels per lane (pointwise_chain_body)
  // two 1024-thread blocks per CU are resident (66 KB of LDS, 63 registers each); sixteen per CU slot are launched, so that the chip works on one
  // compact, advancing window of the buffer: 1 / 2 / 4 / 8 / 16 / 32 / 64 per CU 0.555 / 0.525 / 0.516 / 0.499 / 0.497 / 0.506 / 0.539 ms at 100 MP
  const unsigned cap = (unsigned)(f.num_cus > 0 ? f.num_cus : 256) * 16u;
  const unsigned blocks = (unsigned)std::min<size_t>(cap, (chunks + 15) / 16);
  hipLaunchKernelGGL(k_pointwise_chain<true>, dim3(blocks ? blocks : 1), dim3(1024), 0, s, a, (uint64_t)npix);
  return 0;
}

// ------------------------------------------------------------------------------------------
// Raster sources in one pass: OpGoFloat::run_other (gofloat.rs:171-201) + OpToLab + OpBaseCurve + OpFromLab + OpGamma
// (+ output8bit / output16bit) straight from the RGB8 / RGB16 bytes to the output type -- 6 B/px for RGB8 -> u8 instead of
// the 62 B/px of the four staged kernels.  A wave takes 256 consecutive pixels per step and a lane four consecutive ones:
// 12 (24) source bytes per lane as element-aligned dword loads, contiguous across the wave; the last chunk is shifted left
// to end at the last pixel (its overlap is computed twice, identical values), so every lane is active; the output leaves
// through the wave's LDS staging buffer as in the raw kernel.  Same per-pixel code as everywhere else (pointwise4_fast,
// literal redo behind a wave-uniform branch).  npix >= 256.
// ------------------------------------------------------------------------------------------
struct __attribute__((packed, aligned(1))) rgb8x4 { uint32_t w[3]; };
struct __attribute__((packed, aligned(2))) rgb16x4 { uint32_t w[6]; };
struct Rgbe32 { float4 v; };            // SrcT tag: the source is a 4-channel f32 OpBuffer (demosaic's / gofloat's RGBE pixels), not raster bytes -- the staged
                                        // pipeline's tolab..gamma + quantisation in one pass when the caller wants 8 or 16 bits (ipk_pointwise_chain_out)
template <typename SrcT, int OUT>
__global__ __launch_bounds__(1024) void k_raster_chain(FusedArgs a, uint64_t npix, const LutPair *__restrict__ gamma_reverse) {
  constexpr bool RGBE = std::is_same<SrcT, Rgbe32>::value;
  constexpr bool Q8 = OUT == 1;                                         // OpGamma + output8bit as one step lookup (ipk_device.hpp Q8Entry)
  __shared__ __attribute__((aligned(16))) LabTab s_lab[kLutPairs + 4];
  __shared__ __attribute__((aligned(16))) typename std::conditional<Q8, Q8Entry, typename std::conditional<OUT == 0, float, GamTab>::type>::type s_gam[kLutPairs + 4];
  __shared__ __attribute__((aligned(16))) float s_knots[kKnotFloats];
  __shared__ float s_par[32];
  __shared__ float s_expand[sizeof(SrcT) == 1 ? 256 : 4];              // expand_srgb_gamma(input8bit(i)), the 256 possible RGB8 samples
  constexpr int STG = OUT == 0 ? 768 : (OUT == 1 ? 192 : 384);
  __shared__ __attribute__((aligned(16))) uint32_t s_stage[16 * STG];
  // the tables straight into LDS (stage_lds_direct: no registers), in flight behind the rest of the prologue
  stage_table_direct(s_lab, a.lab_table, a.lab_pairs);
  if constexpr (Q8) stage_q8_direct(s_gam, a.gam_q8); else stage_table_direct(s_gam, a.gam_table, a.gam_pairs);
  if (sizeof(SrcT) == 1) for (int i = threadIdx.x; i < 256; i += blockDim.x) s_expand[i] = lut_interp(gamma_reverse, input8bit((uint8_t)i));
  if (threadIdx.x < 4) s_par[threadIdx.x] = a.tolab.mul[threadIdx.x];
  else if (threadIdx.x < 16) s_par[threadIdx.x] = a.tolab.cm[threadIdx.x - 4];
  else if (threadIdx.x < 25) s_par[threadIdx.x] = a.rgbm.m[threadIdx.x - 16];
  if (threadIdx.x < kSplineMaxKnots) fill_knots(s_knots, a.spline, (int)threadIdx.x);
  __shared__ __attribute__((aligned(16))) float s_grid[kGridFloats];
  if (a.spline.grid_ok) fill_grid(s_grid, a.spline, (int)threadIdx.x);
  sync_after_lds_direct();
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const uint64_t nwaves = (uint64_t)gridDim.x * (blockDim.x >> 6);
  const uint64_t nchunks = (npix + 255) / 256;
  uint32_t *stg = s_stage + (threadIdx.x >> 6) * STG;
  const SrcT *src = reinterpret_cast<const SrcT *>(a.src);
  for (uint64_t chunk = wave; chunk < nchunks; chunk += nwaves) {
    const uint64_t base = min(chunk * 256, npix - 256);                // the last chunk ends at the last pixel
    float4 px[4];
    if constexpr (RGBE) {
      const float *p4 = reinterpret_cast<const float *>(a.src) + (base + 4u * lane) * 4;
      #pragma unroll
      for (int j = 0; j < 4; ++j) px[j] = ld_stream4(p4 + 4 * j);
    } else
    if (sizeof(SrcT) == 1) {
      const rgb8x4 t = *reinterpret_cast<const rgb8x4 *>(src + (base + 4u * lane) * 3);
      float e[12];
      #pragma unroll
      for (int k = 0; k < 12; ++k) e[k] = s_expand[(t.w[k >> 2] >> (8 * (k & 3))) & 0xFFu];
      #pragma unroll
      for (int j = 0; j < 4; ++j) px[j] = make_float4(e[3 * j], e[3 * j + 1], e[3 * j + 2], 0.0f);
    } else {
      const rgb16x4 t = *reinterpret_cast<const rgb16x4 *>(src + (base + 4u * lane) * 3);
      float e[12];
      #pragma unroll
      for (int k = 0; k < 12; ++k) e[k] = input16bit((uint16_t)((t.w[k >> 1] >> (16 * (k & 1))) & 0xFFFFu));
      #pragma unroll
      for (int j = 0; j < 4; ++j) px[j] = make_float4(e[3 * j], e[3 * j + 1], e[3 * j + 2], 0.0f);
    }
    PixOut o[4];
    bool bad = a.fast_ok == 0;
    if (RGBE) {                                                        // the fast form drops the E term: legal while the fourth channel is +0.0 (pointwise_chain_body)
      uint32_t wbits = 0u;
      #pragma unroll
      for (int j = 0; j < 4; ++j) wbits |= __float_as_uint(px[j].w);
      bad |= wbits != 0u;
    }
    if (a.fast_ok) bad |= pointwise4_fast<true>(a, s_par, s_lab, s_gam, s_knots, px, o, a.has_curve != 0, a.linear != 0, 0, nullptr, s_grid);
    if (__builtin_amdgcn_ballot_w64(bad) != 0) {
      #pragma unroll
      for (int j = 0; j < 4; ++j) {
        const PixOut e = pointwise_exact(a, s_lab, s_gam, s_knots, px[j]);
        if (bad) o[j] = e;
      }
    }
    if constexpr (Q8) OutStage<1>::stage_bits(stg, lane, o); else OutStage<OUT>::stage(stg, lane, o);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    OutStage<OUT>::flush(stg, lane, a.dst, (size_t)base);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}
template <typename SrcT>
static void launch_raster_t(const FusedArgs &a, size_t npix, int out_type, const void *gamma_reverse, unsigned blocks, hipStream_t s) {
  const LutPair *gr = reinterpret_cast<const LutPair *>(gamma_reverse);
  if (out_type == 0) hipLaunchKernelGGL((k_raster_chain<SrcT, 0>), dim3(blocks), dim3(1024), 0, s, a, (uint64_t)npix, gr);
  else if (out_type == 1) hipLaunchKernelGGL((k_raster_chain<SrcT, 1>), dim3(blocks), dim3(1024), 0, s, a, (uint64_t)npix, gr);
  else hipLaunchKernelGGL((k_raster_chain<SrcT, 2>), dim3(blocks), dim3(1024), 0, s, a, (uint64_t)npix, gr);
}
// OpToLab..OpGamma + output8bit / output16bit over a 4-channel f32 OpBuffer in one pass (f.out_type 1 or 2; npix >= 256)
int launch_chain_quantised(const FusedLaunch &f, size_t npix, hipStream_t s) {
  if (npix < 256 || (f.out_type != 1 && f.out_type != 2)) return -1;
  FusedArgs a = chain_args(f);
  a.gam_q8 = reinterpret_cast<const Q8Entry *>(f.gam_q8);
  const size_t chunks = (npix + 255) / 256;
  const unsigned cap = (unsigned)(f.num_cus > 0 ? f.num_cus : 256);
  const unsigned blocks = std::max(1u, (unsigned)std::min<size_t>(cap, (chunks + 15) / 16));
  if (f.out_type == 1) hipLaunchKernelGGL((k_raster_chain<Rgbe32, 1>), dim3(blocks), dim3(1024), 0, s, a, (uint64_t)npix, (const LutPair *)nullptr);
  else hipLaunchKernelGGL((k_raster_chain<Rgbe32, 2>), dim3(blocks), dim3(1024), 0, s, a, (uint64_t)npix, (const LutPair *)nullptr);
  return 0;
}
int launch_raster_chain(const FusedLaunch &f, size_t npix, int src_is_u16, const void *gamma_reverse_pairs, hipStream_t s) {
  if (npix < 256) return -1;
  FusedArgs a = chain_args(f);
  a.gam_q8 = reinterpret_cast<const Q8Entry *>(f.gam_q8);
  const size_t chunks = (npix + 255) / 256;
  const unsigned cap = (unsigned)(f.num_cus > 0 ? f.num_cus : 256);
  const unsigned blocks = std::max(1u, (unsigned)std::min<size_t>(cap, (chunks + 15) / 16));
  if (src_is_u16) launch_raster_t<uint16_t>(a, npix, f.out_type, gamma_reverse_pairs, blocks, s);
  else launch_raster_t<uint8_t>(a, npix, f.out_type, gamma_reverse_pairs, blocks, s);
  return 0;
}

// ------------------------------------------------------------------------------------------
// Self-test kernels: exhaustive on-device proofs of the arithmetic shortcuts the fused kernel uses
// (tests/test_gpu_selftest.py).  Each compares a shortcut with the plain IEEE expression over every
// f32 bit pattern and reports the mismatches.
// ------------------------------------------------------------------------------------------
struct SelftestOut { unsigned long long bad; unsigned int first_bad; unsigned int pad; };

__device__ __forceinline__ bool same_f32(float a, float b) {
  return (__float_as_uint(a) == __float_as_uint(b)) || (a != a && b != b);
}
// variant 0: cdiv_fast (with v_div_fixup); 1: the three arithmetic steps only; 2: two steps with a hi/lo reciprocal
// Only dividends with lo_bits <= |x| bits <= hi_bits (plus, if include_special, 0/inf/NaN) are checked.
__global__ void k_selftest_cdiv(float c, float rc, float rc_lo, int variant, unsigned lo_bits, unsigned hi_bits, int include_special,
                                SelftestOut *out) {
  unsigned long long bad = 0; unsigned first = 0xFFFFFFFFu;
  const unsigned long long total = 1ull << 32, stride = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const unsigned bits = (unsigned)i, mag = bits & 0x7FFFFFFFu;
    const bool special = mag == 0 || mag >= 0x7F800000u;
    if (special ? !include_special : (mag < lo_bits || mag > hi_bits)) continue;
    const float x = __uint_as_float(bits);
    float q;
    if (variant == 0) q = cdiv_fast(x, c, rc);
    else if (variant == 1) { const float q0 = x * rc; const float r = __builtin_fmaf(-q0, c, x); q = __builtin_fmaf(r, rc, q0); }
    else { const float t = x * rc_lo; q = __builtin_fmaf(x, rc, t); }
    if (!same_f32(q, x / c)) { ++bad; if (bits < first) first = bits; }
  }
  if (bad) { atomicAdd(&out->bad, bad); atomicMin(&out->first_bad, first); }
}
// pos - trunc(pos) versus v_fract_f32(pos) for every f32 pos in [0, 8192] (the lookup's weight)
__global__ void k_selftest_fract(SelftestOut *out) {
  unsigned long long bad = 0; unsigned first = 0xFFFFFFFFu;
  const unsigned hi = __float_as_uint(8192.0f);
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i <= hi; i += (unsigned long long)gridDim.x * blockDim.x) {
    const float pos = __uint_as_float((unsigned)i);
    if (!same_f32(pos - truncf(pos), __builtin_amdgcn_fractf(pos))) { ++bad; if ((unsigned)i < first) first = (unsigned)i; }
  }
  if (bad) { atomicAdd(&out->bad, bad); atomicMin(&out->first_bad, first); }
}
// clamp: v.max(0.0).min(1.0) versus v_med3_f32(v, 0, 1), every bit pattern (NaN -> 0; the sign of a zero result is
// ignored: the gamma lookup maps +-0 to the same value)
__global__ void k_selftest_clamp(SelftestOut *out) {
  unsigned long long bad = 0; unsigned first = 0xFFFFFFFFu;
  const unsigned long long total = 1ull << 32, stride = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const float v = __uint_as_float((unsigned)i);
    const float a = rs_min(rs_max(v, 0.0f), 1.0f), b = __builtin_amdgcn_fmed3f(v, 0.0f, 1.0f);
    if (!(a == b)) { ++bad; if ((unsigned)i < first) first = (unsigned)i; }
  }
  if (bad) { atomicAdd(&out->bad, bad); atomicMin(&out->first_bad, first); }
}
// output8bit: (v * 256).max(0).min(255) as u8 versus v_cvt_pk_u8_f32 of the product (variant 0), of its floor (variant 1), and
// versus the saturating v_cvt_u32_f32 + unsigned min (variant 2), every bit pattern
// The practical HBM ceiling the roofline figures are compared with: a plain copy, 16 bytes per lane, nothing else (bench.py's copy_ceiling;
// MI355X_MICROARCH.md measures 6.29 TB/s for such a copy against the 8 TB/s spec peak).  Measured forms (tools/copy_probe.hip, 1.2 GB, one box):
// a grid-stride loop over 2048 .. 16384 blocks 4.6-5.2 TB/s (hipMemcpyDtoD 5.26, torch's copy_ 4.7-5.3); ONE block per 1024 consecutive 16-byte
// elements, four per lane, 6.0 TB/s, with nontemporal loads and stores 6.45; one element per lane 6.28.  The flat forms win because neighbouring
// blocks -- which run at the same time -- touch neighbouring addresses; the probe is the best of them.
__global__ __launch_bounds__(256) void k_copy_probe(const ipk_f4v *__restrict__ src, ipk_f4v *__restrict__ dst, size_t n16) {
  const size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
  ipk_f4v v[4];
  #pragma unroll
  for (int u = 0; u < 4; ++u) if (base + u * 256 < n16) v[u] = __builtin_nontemporal_load(src + base + u * 256);
  #pragma unroll
  for (int u = 0; u < 4; ++u) if (base + u * 256 < n16) __builtin_nontemporal_store(v[u], dst + base + u * 256);
}
// The same for the fused kernel's READ : WRITE MIX -- 16 bytes read, 48 written per lane (4 : 12 bytes per pixel), flat launch, contiguous, nontemporal -- with
// no arithmetic at all: what the memory system gives ANY kernel that turns a 1-channel f32 mosaic into 3-channel f32 pixels (bench.py's mix_ceiling).
constexpr int kMixU = 1;     // one group per lane: 100 MP 0.273 ms = 5.86 TB/s; two 0.291; four 0.297 (same box)
__global__ __launch_bounds__(256) void k_mix_probe(const ipk_f4v *__restrict__ src, ipk_f4v *__restrict__ dst, size_t n16) {
  constexpr int U = kMixU;                                  // 16-byte groups read per lane
  const size_t base = (size_t)blockIdx.x * (256 * U) + threadIdx.x;
  ipk_f4v v[U];
  #pragma unroll
  for (int u = 0; u < U; ++u) if (base + u * 256 < n16) v[u] = __builtin_nontemporal_load(src + base + u * 256);
  #pragma unroll
  for (int u = 0; u < U; ++u) if (base + u * 256 < n16) {
    // the block's 256 U input groups become 768 U contiguous output groups, written as lane-contiguous sweeps of 256
    ipk_f4v *o = dst + (size_t)blockIdx.x * (768 * U) + u * 768 + threadIdx.x;
    __builtin_nontemporal_store(v[u], o); __builtin_nontemporal_store(v[u], o + 256); __builtin_nontemporal_store(v[u], o + 512);
  }
}
void launch_mix_probe(const void *src, void *dst, size_t src_bytes, hipStream_t s) {
  const size_t n16 = src_bytes / 16, per = 256 * kMixU;
  hipLaunchKernelGGL(k_mix_probe, dim3((unsigned)std::max<size_t>(1, (n16 + per - 1) / per)), dim3(256), 0, s, reinterpret_cast<const ipk_f4v *>(src), reinterpret_cast<ipk_f4v *>(dst), n16);
}
void launch_copy_probe(const void *src, void *dst, size_t bytes, int, hipStream_t s) {
  const size_t n16 = bytes / 16;
  hipLaunchKernelGGL(k_copy_probe, dim3((unsigned)std::max<size_t>(1, (n16 + 1023) / 1024)), dim3(256), 0, s, reinterpret_cast<const ipk_f4v *>(src), reinterpret_cast<ipk_f4v *>(dst), n16);
}
// Shader clock during whatever else the device runs: one wave spins for spin_ticks of the fixed 100 MHz reference counter (s_memrealtime) and reports
// how far the shader-clock counter (s_memtime) moved meanwhile.  Launched on a second stream beside the kernel under test (bench.py config.shader_clock_GHz).
__global__ __launch_bounds__(64) void k_clock_probe(unsigned long long *out2, unsigned long long spin_ticks) {
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long r1 = r0;
  while (r1 - r0 < spin_ticks) { __builtin_amdgcn_s_sleep(32); r1 = __builtin_amdgcn_s_memrealtime(); }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) { out2[0] = c1 - c0; out2[1] = r1 - r0; }
}
void launch_clock_probe(void *out2_dev, unsigned long long spin_ticks, hipStream_t s) {
  hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, s, reinterpret_cast<unsigned long long *>(out2_dev), spin_ticks);
}
// every f32 argument through the arithmetic 3-knot form against the literal search (curves.rs:126-157)
__global__ void k_selftest_spline3(SplineDev sp, SelftestOut *out) {
  __shared__ __attribute__((aligned(16))) float s_knots[kKnotFloats];
  __shared__ __attribute__((aligned(16))) float s_grid[kGridFloats];
  if (threadIdx.x < kSplineMaxKnots) fill_knots(s_knots, sp, (int)threadIdx.x);
  if (sp.grid_ok) fill_grid(s_grid, sp, (int)threadIdx.x);
  __syncthreads();
  unsigned long long bad = 0; unsigned first = 0xFFFFFFFFu;
  const unsigned long long total = 1ull << 32, stride = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const float v = __uint_as_float((unsigned)i);
    const float want = spline_interpolate_lds(s_knots, sp.npoints, sp.nseg, v), got = sp.grid_ok ? spline_interpolate_grid(sp, s_grid, v) : spline_interpolate_3a(sp, s_knots, v);
    // (the grid form answers a NaN argument with y_0, the literal search with its first probe's knot: no NaN reaches the curve in a lane whose result is
    //  used -- see spline_interpolate_grid -- so NaN arguments are left out of this comparison for grid curves)
    const bool same = __float_as_uint(want) == __float_as_uint(got) || (want != want && got != got) || (sp.grid_ok && v != v);
    if (!same) { ++bad; if ((unsigned)i < first) first = (unsigned)i; }
  }
  if (bad) { atomicAdd(&out->bad, bad); atomicMin(&out->first_bad, first); }
}
__global__ void k_selftest_quant8(SelftestOut *out, int variant) {
  unsigned long long bad = 0; unsigned first = 0xFFFFFFFFu;
  const unsigned long long total = 1ull << 32, stride = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const float v = __uint_as_float((unsigned)i);
    const uint32_t a = output8bit_literal(v);
    const float p = v * 256.0f;
    uint32_t b;
    if (variant == 0) b = __builtin_amdgcn_cvt_pk_u8_f32(p, 0u, 0u);
    else if (variant == 1) b = __builtin_amdgcn_cvt_pk_u8_f32(floorf(p), 0u, 0u);
    else b = min(f32_as_u32_sat(p), 255u);
    if (a != b) { ++bad; if ((unsigned)i < first) first = (unsigned)i; }
  }
  if (bad) { atomicAdd(&out->bad, bad); atomicMin(&out->first_bad, first); }
}
// The 8-bit step table (Q8Entry): one thread per gamma-table segment bisects, over f32 BIT PATTERNS (all operations are monotone in c >= 0), for the first
// sample of the segment, the first of the next one, and the first sample inside whose literal result -- OpGamma's step, then output8bit -- is one above
// the segment's first.  ipk_selftest_q8 then compares the step form with the literal one on every f32.
__device__ __forceinline__ uint32_t q8_literal(const LutPair *__restrict__ pairs, float c) { return (uint32_t)output8bit(lut_interp(pairs, c)); }
__global__ void k_build_q8(const LutPair *__restrict__ pairs, Q8Entry *__restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (uint32_t)kLutPairs) return;
  auto first_with_key = [&](uint32_t key) {                 // smallest bit pattern b in [0, bits(1.0)] with u32(c * 8191) >= key; bits(1.0) + 1 when there is none
    uint32_t lo = 0u, hi = 0x3F800001u;
    while (lo < hi) { const uint32_t mid = lo + (hi - lo) / 2u; if (f32_as_u32_sat(__uint_as_float(mid) * kLutMaxF) >= key) hi = mid; else lo = mid + 1u; }
    return lo;
  };
  const uint32_t b0 = first_with_key(i), b1 = first_with_key(i + 1u);       // the segment's samples: bit patterns [b0, b1)
  Q8Entry e; e.k = 0u; e.t = __builtin_inff();
  if (b0 < b1) {
    e.k = q8_literal(pairs, __uint_as_float(b0));
    if (q8_literal(pairs, __uint_as_float(b1 - 1u)) != e.k) {
      uint32_t lo = b0, hi = b1 - 1u;                       // the first pattern whose literal result is above k
      while (lo < hi) { const uint32_t mid = lo + (hi - lo) / 2u; if (q8_literal(pairs, __uint_as_float(mid)) > e.k) hi = mid; else lo = mid + 1u; }
      e.t = __uint_as_float(lo);
    }
  }
  out[i] = e;
}
void launch_build_q8(const void *gam_pairs, void *q8_out, hipStream_t s) {
  hipLaunchKernelGGL(k_build_q8, dim3(kLutPairs / 256), dim3(256), 0, s, reinterpret_cast<const LutPair *>(gam_pairs), reinterpret_cast<Q8Entry *>(q8_out));
}
// every f32 x: OpGamma's step on x followed by output8bit (the literal device forms) against clamp + step lookup (what the 8-bit kernels run)
__global__ void k_selftest_q8(const LutPair *__restrict__ pairs, const Q8Entry *__restrict__ q8, SelftestOut *out) {
  unsigned long long bad = 0; unsigned first = 0xFFFFFFFFu;
  const unsigned long long total = 1ull << 32, stride = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const float x = __uint_as_float((unsigned)i);
    const uint32_t a = (uint32_t)output8bit_literal(gamma_sample(pairs, x));
    const uint32_t b = q8_sample(q8, __builtin_amdgcn_fmed3f(x, 0.0f, 1.0f));
    if (a != b) { ++bad; if ((unsigned)i < first) first = (unsigned)i; }
  }
  if (bad) { atomicAdd(&out->bad, bad); atomicMin(&out->first_bad, first); }
}
// output16bit: the literal (v * 65535).round().max(0).min(65535) as u16 against the packed conversion form (output16bit_x2), every bit pattern, both halves
__global__ void k_selftest_quant16(SelftestOut *out) {
  unsigned long long bad = 0; unsigned first = 0xFFFFFFFFu;
  const unsigned long long total = 1ull << 32, stride = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const float v = __uint_as_float((unsigned)i);
    const uint32_t a = output16bit_literal(v);
    const uint32_t w = output16bit_x2(v, -v), w2 = output16bit_x2(0.25f, v);
    if ((w & 0xFFFFu) != a || (w2 >> 16) != a || (w2 & 0xFFFFu) != 16384u) { ++bad; if ((unsigned)i < first) first = (unsigned)i; }
  }
  if (bad) { atomicAdd(&out->bad, bad); atomicMin(&out->first_bad, first); }
}
int launch_selftest_quant16(void *out_dev, hipStream_t s) { hipLaunchKernelGGL(k_selftest_quant16, dim3(256 * 8), dim3(256), 0, s, reinterpret_cast<SelftestOut *>(out_dev)); return 0; }
int launch_selftest_q8(const void *gam_pairs, const void *q8, void *out_dev, hipStream_t s) {
  hipLaunchKernelGGL(k_selftest_q8, dim3(256 * 8), dim3(256), 0, s, reinterpret_cast<const LutPair *>(gam_pairs), reinterpret_cast<const Q8Entry *>(q8), reinterpret_cast<SelftestOut *>(out_dev));
  return 0;
}
// device cbrt variants on an array (the host compares with libm): 0 literal glibc port, 1 select form, 2 fast form for (1,2)
__global__ void k_selftest_cbrt(const float *__restrict__ in, float *__restrict__ out, size_t n, int variant) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float x = in[i];
    out[i] = variant == 0 ? cbrtf_glibc(x) : (variant == 1 ? cbrtf_glibc_sel(x) : cbrtf_glibc_1to2(x));
  }
}

int launch_selftest_cdiv(float c, int variant, unsigned lo_bits, unsigned hi_bits, int include_special, void *out_dev, hipStream_t s) {
  const float rc = 1.0f / c;
  const float rc_lo = (float)(1.0 / (double)c - (double)rc);
  hipLaunchKernelGGL(k_selftest_cdiv, dim3(256 * 8), dim3(256), 0, s, c, rc, rc_lo, variant, lo_bits, hi_bits, include_special,
                     reinterpret_cast<SelftestOut *>(out_dev));
  return 0;
}
// returns -2 when the curve is not one the arithmetic form is used for (not 2 or 3 knots, or spline3_arith_ok fails)
int launch_selftest_spline3(const SplineHost &h, void *out_dev, hipStream_t s) {
  SplineDev d = make_spline(h);
  if (d.npoints == 2) {                                   // padded like launch_fused_bayer does
    d.npoints = 3; d.nseg = 2; d.px[2] = d.px[1]; d.py[2] = d.py[1];
    d.c1[1] = d.c1[0]; d.c2[1] = d.c2[0]; d.c3[1] = d.c3[0]; d.c1[2] = d.c1[1];
  }
  if (!d.grid_ok && (d.npoints != 3 || !spline3_arith_ok(d))) return -2;
  hipLaunchKernelGGL(k_selftest_spline3, dim3(256 * 8), dim3(256), 0, s, d, reinterpret_cast<SelftestOut *>(out_dev));
  return 0;
}
int launch_selftest_fract(void *out_dev, hipStream_t s) { hipLaunchKernelGGL(k_selftest_fract, dim3(256 * 8), dim3(256), 0, s, reinterpret_cast<SelftestOut *>(out_dev)); return 0; }
int launch_selftest_clamp(void *out_dev, hipStream_t s) { hipLaunchKernelGGL(k_selftest_clamp, dim3(256 * 8), dim3(256), 0, s, reinterpret_cast<SelftestOut *>(out_dev)); return 0; }
int launch_selftest_quant8(void *out_dev, int variant, hipStream_t s) { hipLaunchKernelGGL(k_selftest_quant8, dim3(256 * 8), dim3(256), 0, s, reinterpret_cast<SelftestOut *>(out_dev), variant); return 0; }
int launch_selftest_cbrt(const float *in, float *out, size_t n, int variant, hipStream_t s) {
  hipLaunchKernelGGL(k_selftest_cbrt, dim3(256 * 8), dim3(256), 0, s, in, out, n, variant); return 0;
}

}  // namespace ipk
