// Development tool (round 6): is v_cvt_pknorm_u16_f32 output16bit (src/color_conversions.rs:327-330: (v * 65535.0).round().max(0).min(65535) as u16) on every f32?
//   build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/pknorm_probe.hip -o tools/build/pknorm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned long long *bad, unsigned *first) {
  unsigned long long b = 0; unsigned f = 0xFFFFFFFFu;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += (unsigned long long)gridDim.x * blockDim.x) {
    const float v = __uint_as_float((unsigned)i);
    float r = roundf(v * 65535.0f);
    r = (r > 0.0f) ? r : 0.0f;                 // .max(0): NaN -> 0
    r = (r < 65535.0f) ? r : 65535.0f;
    const unsigned ref = (unsigned)r;
    const u16x2 p = __builtin_amdgcn_cvt_pknorm_u16(v, 0.0f);
    if ((unsigned)p.x != ref) { ++b; if ((unsigned)i < f) f = (unsigned)i; }
  }
  if (b) { atomicAdd(bad, b); atomicMin(first, f); }
}
int main() {
  unsigned long long *bad, hb = 0; unsigned *first, hf = 0xFFFFFFFFu;
  hipMalloc(&bad, 8); hipMalloc(&first, 4); hipMemcpy(bad, &hb, 8, hipMemcpyHostToDevice); hipMemcpy(first, &hf, 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(2048), dim3(256), 0, 0, bad, first);
  hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&hf, first, 4, hipMemcpyDeviceToHost);
  printf("v_cvt_pknorm_u16_f32 vs output16bit: %llu mismatches of 2^32, first at bits 0x%08x\n", hb, hf);
  return 0;
}
