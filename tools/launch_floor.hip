// Development tool: what a launch of the fused kernel's shape costs before its first row -- 256 blocks x 1024 threads with 146 KB of LDS: (a) nothing,
// (b) the two lookup tables copied into LDS (96 KB per block, LDS-direct loads) and the block barrier, (c) the same through registers.
// build: hipcc --offload-arch=gfx950 -O3 tools/launch_floor.hip -o tools/build/launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;
constexpr int kLds = 146 * 1024;
template <int MODE>
__global__ __launch_bounds__(1024) void k_floor(const float4 *tab, float *out) {
  extern __shared__ __attribute__((aligned(16))) float4 s[];
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  if (MODE == 1) {
    for (uint32_t off = wave * 64u; off < 6144u; off += 1024u) __builtin_amdgcn_global_load_lds((gptr_t)(tab + off + lane), (lptr_t)(s + off), 16, 0, 0);
  } else if (MODE == 2) {
    float4 r[6];
    #pragma unroll
    for (int k = 0; k < 6; ++k) r[k] = tab[threadIdx.x + 1024 * k];
    #pragma unroll
    for (int k = 0; k < 6; ++k) s[threadIdx.x + 1024 * k] = r[k];
  }
  __syncthreads();
  if (MODE != 0 && out && threadIdx.x == 0) out[blockIdx.x] = s[(blockIdx.x * 37u) % 6144u].x;
}
template <int MODE>
static float run(const float4 *tab, float *out, int blocks, int n) {
  hipFuncSetAttribute(reinterpret_cast<const void *>(k_floor<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k_floor<MODE>, dim3(blocks), dim3(1024), kLds, 0, tab, out);
  hipEventRecord(e0, 0);
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_floor<MODE>, dim3(blocks), dim3(1024), kLds, 0, tab, out);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / n;
}
int main() {
  float4 *tab; float *out;
  hipMalloc(reinterpret_cast<void **>(&tab), 6144 * 16); hipMemset(tab, 0, 6144 * 16); hipMalloc(reinterpret_cast<void **>(&out), 4096);
  for (int blocks : {256, 512}) {
    printf("%d blocks x 1024 threads, %d KB LDS, back-to-back launches: empty %.2f us | tables by LDS-direct loads + barrier %.2f us | through registers %.2f us\n",
           blocks, kLds / 1024, run<0>(tab, out, blocks, 2000), run<1>(tab, out, blocks, 2000), run<2>(tab, out, blocks, 2000));
  }
  return 0;
}
