// Development tool: how fast a 2-byte-element transpose (the mosaic permutation in front of the rotated-space fused kernel, launch_rotate1) can go on this
// chip, by tile shape and access width.  out[j][i] = in[i][j]; NI x NJ u16.  build: hipcc --offload-arch=gfx950 -O3 tools/transpose_probe.hip -o tools/build/transpose_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint16_t T;
struct __attribute__((packed, aligned(2))) V8 { T v[8]; };

// the shipped form: TW x TW tile, one element per lane and access
template <int TW>
__global__ __launch_bounds__(256) void k_elem(const T *__restrict__ src, uint32_t NI, uint32_t NJ, T *__restrict__ dst) {
  __shared__ T tile[TW][TW + 2];
  constexpr uint32_t RPP = 256 / TW;
  const uint32_t I0 = blockIdx.x * TW, J0 = blockIdx.y * TW, a = threadIdx.x % TW, b = threadIdx.x / TW;
  for (uint32_t p = 0; p < TW / RPP; ++p) { const uint32_t i = I0 + b + RPP * p, j = J0 + a; if (i < NI && j < NJ) tile[b + RPP * p][a] = src[(size_t)i * NJ + j]; }
  __syncthreads();
  for (uint32_t p = 0; p < TW / RPP; ++p) { const uint32_t j = J0 + b + RPP * p, i = I0 + a; if (i < NI && j < NJ) dst[(size_t)j * NI + i] = tile[a][b + RPP * p]; }
}
// TI source rows x TJ source columns per tile, 16-byte global accesses on both sides, element-wise LDS reads
template <int TI, int TJ, int THREADS, int PAD>
__global__ __launch_bounds__(THREADS) void k_vec(const T *__restrict__ src, uint32_t NI, uint32_t NJ, T *__restrict__ dst) {
  constexpr int PITCH = TJ + PAD;
  extern __shared__ __attribute__((aligned(16))) T lds[];
  const uint32_t I0 = blockIdx.x * TI, J0 = blockIdx.y * TJ;
  constexpr int VJ = TJ / 8, VI = TI / 8;
  for (int v = threadIdx.x; v < TI * VJ; v += THREADS) {
    const int i = v / VJ, j8 = (v % VJ) * 8;
    const V8 x = *reinterpret_cast<const V8 *>(src + (size_t)(I0 + i) * NJ + J0 + j8);
    #pragma unroll
    for (int k = 0; k < 8; ++k) lds[i * PITCH + j8 + k] = x.v[k];
  }
  __syncthreads();
  for (int v = threadIdx.x; v < TJ * VI; v += THREADS) {
    const int j = v / VI, i8 = (v % VI) * 8;
    V8 x;
    #pragma unroll
    for (int k = 0; k < 8; ++k) x.v[k] = lds[(i8 + k) * PITCH + j];
    *reinterpret_cast<V8 *>(dst + (size_t)(J0 + j) * NI + I0 + i8) = x;
  }
}
template <typename F> static float timeit(F f, int n) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) f();
  hipEventRecord(e0, 0); for (int i = 0; i < n; ++i) f(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / n;
}
template <int TI, int TJ, int THREADS, int PAD>
static void run_vec(const T *s, T *d, uint32_t NI, uint32_t NJ, const std::vector<T> &h) {
  const size_t lds = (size_t)TI * (TJ + PAD) * sizeof(T);
  hipFuncSetAttribute(reinterpret_cast<const void *>(k_vec<TI, TJ, THREADS, PAD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipMemset(d, 0, (size_t)NI * NJ * 2);
  auto f = [&]() { hipLaunchKernelGGL((k_vec<TI, TJ, THREADS, PAD>), dim3(NI / TI, NJ / TJ), dim3(THREADS), lds, 0, s, NI, NJ, d); };
  const float ms = timeit(f, 30);
  std::vector<T> o((size_t)NI * NJ); hipMemcpy(o.data(), d, o.size() * 2, hipMemcpyDeviceToHost);
  size_t bad = 0; for (size_t i = 0; i < NI; i += 97) for (size_t j = 0; j < NJ; j += 89) bad += o[j * NI + i] != h[i * NJ + j];
  printf("vec  %3d x %3d tile, %4d threads, pad %2d (%3zu KB LDS): %.4f ms = %.2f TB/s%s\n", TI, TJ, THREADS, PAD, lds / 1024, ms, 4.0 * NI * NJ / ms / 1e9, bad ? "  WRONG" : "");
}
int main() {
  const uint32_t NI = 10240, NJ = 9728;                 // ~100 MP, multiples of every tile size tried
  std::vector<T> h((size_t)NI * NJ); for (size_t i = 0; i < h.size(); ++i) h[i] = (T)(i * 2654435761u >> 13);
  T *s, *d; hipMalloc(reinterpret_cast<void **>(&s), h.size() * 2); hipMalloc(reinterpret_cast<void **>(&d), h.size() * 2);
  hipMemcpy(s, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  { auto f = [&]() { hipLaunchKernelGGL((k_elem<64>), dim3(NI / 64, NJ / 64), dim3(256), 0, 0, s, NI, NJ, d); };
    const float ms = timeit(f, 30); printf("elem  64 x  64 tile (shipped form): %.4f ms = %.2f TB/s\n", ms, 4.0 * NI * NJ / ms / 1e9); }
  { auto f = [&]() { hipMemcpyAsync(d, s, h.size() * 2, hipMemcpyDeviceToDevice, 0); }; const float ms = timeit(f, 30); printf("plain copy: %.4f ms = %.2f TB/s\n", ms, 4.0 * NI * NJ / ms / 1e9); }
  run_vec<64, 64, 256, 2>(s, d, NI, NJ, h);
  run_vec<64, 64, 256, 8>(s, d, NI, NJ, h);
  run_vec<64, 128, 256, 2>(s, d, NI, NJ, h);
  run_vec<128, 64, 256, 2>(s, d, NI, NJ, h);
  run_vec<128, 128, 512, 2>(s, d, NI, NJ, h);
  run_vec<128, 128, 1024, 2>(s, d, NI, NJ, h);
  run_vec<128, 256, 1024, 2>(s, d, NI, NJ, h);
  run_vec<256, 128, 1024, 2>(s, d, NI, NJ, h);
  run_vec<256, 256, 1024, 2>(s, d, NI, NJ, h);
  run_vec<256, 256, 1024, 10>(s, d, NI, NJ, h);
  return 0;
}
