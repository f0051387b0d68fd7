// Development tool: variants of a plain device-to-device copy on MI355X (which one is the fairest "copy ceiling" for bench.py).
//   hipcc --offload-arch=gfx950 -O3 -o tools/build/copy_probe tools/copy_probe.hip && tools/build/copy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4v __attribute__((ext_vector_type(4)));
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_loop(const f4v *__restrict__ s, f4v *__restrict__ d, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    f4v v[U];
    #pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(s + i + u * stride) : s[i + u * stride];
    #pragma unroll
    for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], d + i + u * stride); else d[i + u * stride] = v[u]; }
  }
  for (; i < n; i += stride) d[i] = s[i];
}
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_flat(const f4v *__restrict__ s, f4v *__restrict__ d, size_t n) {   // one block = U * 256 consecutive f4v
  const size_t base = (size_t)blockIdx.x * (U * 256) + threadIdx.x;
  f4v v[U];
  #pragma unroll
  for (int u = 0; u < U; ++u) if (base + u * 256 < n) v[u] = NT ? __builtin_nontemporal_load(s + base + u * 256) : s[base + u * 256];
  #pragma unroll
  for (int u = 0; u < U; ++u) if (base + u * 256 < n) { if (NT) __builtin_nontemporal_store(v[u], d + base + u * 256); else d[base + u * 256] = v[u]; }
}
int main() {
  const size_t bytes = 1200000000, n = bytes / 16;
  f4v *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMemset(a, 1, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char *name, auto launch) {
    for (int i = 0; i < 30; ++i) launch();
    hipDeviceSynchronize(); hipEventRecord(e0);
    for (int i = 0; i < 30; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 30;
    printf("%-28s %.4f ms  %.0f GB/s\n", name, ms, 2.0 * bytes / (ms * 1e-3) / 1e9);
  };
  for (int blocks : {256 * 8, 256 * 16, 256 * 32, 256 * 64}) {
    char nm[64];
    snprintf(nm, 64, "loop U1 blocks=%d", blocks); run(nm, [&] { hipLaunchKernelGGL((k_loop<1, false>), dim3(blocks), dim3(256), 0, 0, a, b, n); });
    snprintf(nm, 64, "loop U4 blocks=%d", blocks); run(nm, [&] { hipLaunchKernelGGL((k_loop<4, false>), dim3(blocks), dim3(256), 0, 0, a, b, n); });
    snprintf(nm, 64, "loop U4 nt blocks=%d", blocks); run(nm, [&] { hipLaunchKernelGGL((k_loop<4, true>), dim3(blocks), dim3(256), 0, 0, a, b, n); });
  }
  run("flat U1", [&] { hipLaunchKernelGGL((k_flat<1, false>), dim3((n + 255) / 256), dim3(256), 0, 0, a, b, n); });
  run("flat U4", [&] { hipLaunchKernelGGL((k_flat<4, false>), dim3((n + 1023) / 1024), dim3(256), 0, 0, a, b, n); });
  run("flat U4 nt", [&] { hipLaunchKernelGGL((k_flat<4, true>), dim3((n + 1023) / 1024), dim3(256), 0, 0, a, b, n); });
  run("flat U8", [&] { hipLaunchKernelGGL((k_flat<8, false>), dim3((n + 2047) / 2048), dim3(256), 0, 0, a, b, n); });
  run("flat U8 nt", [&] { hipLaunchKernelGGL((k_flat<8, true>), dim3((n + 2047) / 2048), dim3(256), 0, 0, a, b, n); });
  run("hipMemcpyDtoD", [&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); });
  return 0;
}
