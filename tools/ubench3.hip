// Development tool (round 3): two questions about gfx950's issue model that decide what can still be done to the fused kernel.
//  (1) How does the VALU issue rate of a SIMD depend on the number of resident waves (1..8), for independent multiplies, for a dependent chain
//      and for the kernel's kind of mix?  (Is a parked wave lost issue bandwidth with four waves per SIMD, and would a fifth help?)
//  (2) v_mfma_f32_4x4x1_16b_f32 computes D[i] = A(lane 4*(l/4)+i) * B(lane l) + C[i] with one rounding: with C = -0.0 that is four individually
//      rounded products per lane on the MATRIX pipe.  Is it bit-identical to v_mul_f32 (denormals, specials)?  And what does a wave pay for one
//      between its VALU instructions?
// hipcc --offload-arch=gfx950 -O2 tools/ubench3.hip -o tools/build/ubench3 && tools/build/ubench3
#pragma clang diagnostic ignored "-Wunused-value"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <random>

#define ITERS 4096
typedef float v4f __attribute__((ext_vector_type(4)));

#define A1(I, X) asm volatile(I : "+v"(X) : "v"(k));
#define ALL8(I) A1(I, a0) A1(I, a1) A1(I, a2) A1(I, a3) A1(I, a4) A1(I, a5) A1(I, a6) A1(I, a7)
#define ALL16(I) ALL8(I) A1(I, b0) A1(I, b1) A1(I, b2) A1(I, b3) A1(I, b4) A1(I, b5) A1(I, b6) A1(I, b7)
#define PRO float a0 = c, a1 = c + 1, a2 = c + 2, a3 = c + 3, a4 = c + 4, a5 = c + 5, a6 = c + 6, a7 = c + 7; \
  float b0 = c * 2, b1 = c * 3, b2 = c * 4, b3 = c * 5, b4 = c * 6, b5 = c * 7, b6 = c * 8, b7 = c * 9; float k = c * 0.5f;
#define EPI out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7;

__global__ __launch_bounds__(256) void k_mul(float *out, float c) { PRO for (int i = 0; i < ITERS; ++i) { ALL16("v_mul_f32 %0, %0, %1") } EPI }
__global__ __launch_bounds__(256) void k_dep(float *out, float c) { PRO for (int i = 0; i < ITERS; ++i) {
  A1("v_mul_f32 %0, %0, %1", a0) A1("v_mul_f32 %0, %0, %1", a0) A1("v_mul_f32 %0, %0, %1", a0) A1("v_mul_f32 %0, %0, %1", a0)
  A1("v_mul_f32 %0, %0, %1", a0) A1("v_mul_f32 %0, %0, %1", a0) A1("v_mul_f32 %0, %0, %1", a0) A1("v_mul_f32 %0, %0, %1", a0)
  A1("v_mul_f32 %0, %0, %1", a0) A1("v_mul_f32 %0, %0, %1", a0) A1("v_mul_f32 %0, %0, %1", a0) A1("v_mul_f32 %0, %0, %1", a0)
  A1("v_mul_f32 %0, %0, %1", a0) A1("v_mul_f32 %0, %0, %1", a0) A1("v_mul_f32 %0, %0, %1", a0) A1("v_mul_f32 %0, %0, %1", a0) } EPI }
// the fused kernel's kind of mix: mul/add/fma interleaved with the half-rate class
__global__ __launch_bounds__(256) void k_mix(float *out, float c) { PRO for (int i = 0; i < ITERS; ++i) {
  A1("v_mul_f32 %0, %0, %1", a0) A1("v_add_f32 %0, %0, %1", a1) A1("v_min_f32 %0, %0, %1", a2) A1("v_mul_f32 %0, %0, %1", a3)
  A1("v_fmaak_f32 %0, %0, %1, 0x3f8ccccd", a4) A1("v_fract_f32 %0, %0", a5) A1("v_add_f32 %0, %0, %1", a6) A1("v_mul_f32 %0, %0, %1", a7)
  A1("v_cvt_u32_f32 %0, %0", b0) A1("v_mul_f32 %0, %0, %1", b1) A1("v_add_f32 %0, %0, %1", b2) A1("v_med3_f32 %0, %0, 0, 1.0", b3)
  A1("v_mul_f32 %0, %0, %1", b4) A1("v_sub_f32 %0, %0, %1", b5) A1("v_lshl_add_u32 %0, %0, 3, %1", b6) A1("v_mul_f32 %0, %0, %1", b7) } EPI }

// 16 VALU multiplies + NM MFMA 4x4x1 per iteration (the MFMA results feed a slow accumulate so that they are not dead)
template <int NM>
__global__ __launch_bounds__(256) void k_mul_mfma(float *out, float c) {
  PRO
  const v4f z = {-0.0f, -0.0f, -0.0f, -0.0f};
  float ma = c * 3.0f, mb = c * 5.0f;
  for (int i = 0; i < ITERS; ++i) {
    ALL8("v_mul_f32 %0, %0, %1")
    // the B operand is a value the loop has just produced, the products stay alive through an empty asm: nothing to hoist, nothing to drop
    if (NM >= 1) { const v4f d = __builtin_amdgcn_mfma_f32_4x4x1f32(ma, a0, z, 0, 0, 0); asm volatile("" :: "v"(d)); }
    if (NM >= 2) { const v4f d = __builtin_amdgcn_mfma_f32_4x4x1f32(mb, a4, z, 0, 0, 0); asm volatile("" :: "v"(d)); }
    A1("v_mul_f32 %0, %0, %1", b0) A1("v_mul_f32 %0, %0, %1", b1) A1("v_mul_f32 %0, %0, %1", b2) A1("v_mul_f32 %0, %0, %1", b3)
    if (NM >= 3) { const v4f d = __builtin_amdgcn_mfma_f32_4x4x1f32(ma, b0, z, 0, 0, 0); asm volatile("" :: "v"(d)); }
    if (NM >= 4) { const v4f d = __builtin_amdgcn_mfma_f32_4x4x1f32(mb, b2, z, 0, 0, 0); asm volatile("" :: "v"(d)); }
    A1("v_mul_f32 %0, %0, %1", b4) A1("v_mul_f32 %0, %0, %1", b5) A1("v_mul_f32 %0, %0, %1", b6) A1("v_mul_f32 %0, %0, %1", b7)
  }
  EPI
}
// the consumer form: the three useful products of each MFMA are ADDED (VALU) as the matrix rows of the kernel would: 16 mul + 4 mfma + 8 add
__global__ __launch_bounds__(256) void k_mul_mfma_use(float *out, float c) {
  PRO
  const v4f z = {-0.0f, -0.0f, -0.0f, -0.0f};
  float ma = c * 3.0f, mb = c * 5.0f;
  for (int i = 0; i < ITERS; ++i) {
    ALL8("v_mul_f32 %0, %0, %1")
    const v4f d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(ma, a0, z, 0, 0, 0);
    const v4f d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(mb, a4, z, 0, 0, 0);
    A1("v_mul_f32 %0, %0, %1", b0) A1("v_mul_f32 %0, %0, %1", b1) A1("v_mul_f32 %0, %0, %1", b2) A1("v_mul_f32 %0, %0, %1", b3)
    const v4f d2 = __builtin_amdgcn_mfma_f32_4x4x1f32(ma, b0, z, 0, 0, 0);
    const v4f d3 = __builtin_amdgcn_mfma_f32_4x4x1f32(mb, b2, z, 0, 0, 0);
    A1("v_mul_f32 %0, %0, %1", b4) A1("v_mul_f32 %0, %0, %1", b5) A1("v_mul_f32 %0, %0, %1", b6) A1("v_mul_f32 %0, %0, %1", b7)
    a1 = (d0.x + d0.y) + d0.z; a2 = (d1.x + d1.y) + d1.z; a5 = (d2.x + d2.y) + d2.z; a6 = (d3.x + d3.y) + d3.z;
  }
  EPI
}
// the same work on the VALU alone: 16 mul + 12 mul + 8 add
__global__ __launch_bounds__(256) void k_mul_valu_use(float *out, float c) {
  PRO
  float ma = c * 3.0f, mb = c * 5.0f, mc = c * 7.0f;
  for (int i = 0; i < ITERS; ++i) {
    ALL16("v_mul_f32 %0, %0, %1")
    float p0, p1, p2, q0, q1, q2, r0, r1, r2, s0, s1, s2;
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p0) : "v"(a0), "v"(ma)); asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p1) : "v"(a0), "v"(mb)); asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p2) : "v"(a0), "v"(mc));
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(q0) : "v"(a4), "v"(ma)); asm volatile("v_mul_f32 %0, %1, %2" : "=v"(q1) : "v"(a4), "v"(mb)); asm volatile("v_mul_f32 %0, %1, %2" : "=v"(q2) : "v"(a4), "v"(mc));
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r0) : "v"(b0), "v"(ma)); asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r1) : "v"(b0), "v"(mb)); asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r2) : "v"(b0), "v"(mc));
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(s0) : "v"(b2), "v"(ma)); asm volatile("v_mul_f32 %0, %1, %2" : "=v"(s1) : "v"(b2), "v"(mb)); asm volatile("v_mul_f32 %0, %1, %2" : "=v"(s2) : "v"(b2), "v"(mc));
    a1 = (p0 + p1) + p2; a2 = (q0 + q1) + q2; a5 = (r0 + r1) + r2; a6 = (s0 + s1) + s2;
  }
  EPI
}
// the same number of PRODUCTS as k_mul_mfma<4> delivers (16 + 12 useful), all on the VALU: 28 multiplies
__global__ __launch_bounds__(256) void k_mul28(float *out, float c) { PRO for (int i = 0; i < ITERS; ++i) {
  ALL16("v_mul_f32 %0, %0, %1") ALL8("v_mul_f32 %0, %0, %1") A1("v_mul_f32 %0, %0, %1", b0) A1("v_mul_f32 %0, %0, %1", b1) A1("v_mul_f32 %0, %0, %1", b2) A1("v_mul_f32 %0, %0, %1", b3) } EPI }

// exactness: D[i](lane l) against v_mul_f32(A(lane 4*(l/4)+i), B(lane l)) for the A, B of the buffers
__global__ void k_mfma_exact(const float *__restrict__ A, const float *__restrict__ B, size_t n, unsigned long long *bad, unsigned long long *bad_zero, int negzero_c) {
  unsigned long long nb = 0, nz = 0;
  for (size_t base = (size_t)blockIdx.x * blockDim.x; base < n; base += (size_t)gridDim.x * blockDim.x) {
    const size_t i = base + threadIdx.x;                 // n is a multiple of the block size: whole waves
    const float a = A[i], b = B[i];
    const float cz = negzero_c ? -0.0f : 0.0f;
    const v4f c = {cz, cz, cz, cz};
    const v4f d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    const int lane = threadIdx.x & 63;
    #pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float ar = __shfl(a, (lane & ~3) + r, 64);
      float ref; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(ref) : "v"(ar), "v"(b));
      const float got = r == 0 ? d.x : r == 1 ? d.y : r == 2 ? d.z : d.w;
      const uint32_t ur = __float_as_uint(ref), ug = __float_as_uint(got);
      const bool nan_both = (ref != ref) && (got != got);
      if (ur != ug && !nan_both) { if (((ur | ug) & 0x7FFFFFFFu) == 0) nz++; else nb++; }
    }
  }
  if (nb) atomicAdd(bad, nb);
  if (nz) atomicAdd(bad_zero, nz);
}

static float time_kernel(void (*fn)(float *, float), int blocks, float *out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), 0, 0, out, 1.0001f); hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), 0, 0, out, 1.0001f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
  float *out; hipMalloc(&out, 8192 * 256 * 4);
  printf("== (1) ns per VALU instruction per SIMD against waves per SIMD (256-thread blocks, one wave per SIMD each) ==\n");
  for (int w : {1, 2, 3, 4, 5, 6, 8}) {
    const int blocks = 256 * w;
    const double n = (double)ITERS * 16;
    const float m = time_kernel(k_mul, blocks, out), d = time_kernel(k_dep, blocks, out), x = time_kernel(k_mix, blocks, out);
    printf("waves %d: mul %.3f ns  dep-chain %.3f ns  mix %.3f ns   (per instruction per SIMD; per wave: mul %.2f ns dep %.2f ns mix %.2f ns)\n", w, m * 1e6 / (n * w), d * 1e6 / (n * w),
           x * 1e6 / (n * w), m * 1e6 / n, d * 1e6 / n, x * 1e6 / n);
  }
  printf("== (2) 16 v_mul + NM x v_mfma_f32_4x4x1 per iteration ==\n");
  for (int w : {1, 2, 4, 5, 6}) {
    const int blocks = 256 * w;
    const float t0 = time_kernel(k_mul_mfma<0>, blocks, out), t1 = time_kernel(k_mul_mfma<1>, blocks, out), t2 = time_kernel(k_mul_mfma<2>, blocks, out),
                t4 = time_kernel(k_mul_mfma<4>, blocks, out), t28 = time_kernel(k_mul28, blocks, out), t16 = time_kernel(k_mul, blocks, out),
                tu = time_kernel(k_mul_mfma_use, blocks, out), tv = time_kernel(k_mul_valu_use, blocks, out);
    printf("waves %d: 16 mul %.3f ms (plain loop %.3f) | +1 mfma %.3f | +2 mfma %.3f | +4 mfma %.3f | 28 mul %.3f ms | 16 mul + 4 mfma + 8 add %.3f | 28 mul + 8 add %.3f\n", w, t0, t16, t1, t2, t4, t28, tu, tv);
  }
  // exactness
  const size_t n = 1u << 24;
  std::vector<float> ha(n), hb(n);
  std::mt19937_64 rng(12345);
  auto rnd_bits = [&](int kind) -> float {
    uint32_t u = (uint32_t)rng();
    if (kind == 1) u = (u & 0x807FFFFFu) | ((uint32_t)(rng() % 40) << 23);              // tiny: exponent 0..39 (incl. denormals)
    if (kind == 2) u = (u & 0x807FFFFFu) | ((uint32_t)(100 + rng() % 60) << 23);        // mid
    if (kind == 3) { const uint32_t sp[8] = {0u, 0x80000000u, 0x7F800000u, 0xFF800000u, 0x7FC00000u, 0x00000001u, 0x807FFFFFu, 0x3F800000u}; u = sp[rng() % 8]; }
    float f; memcpy(&f, &u, 4); return f;
  };
  float *da, *db; unsigned long long *dbad; hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dbad, 16);
  printf("== (3) v_mfma_f32_4x4x1 against v_mul_f32, %zu lanes x 4 products per case ==\n", n);
  for (int ka = 0; ka < 4; ++ka) for (int kb = 0; kb < 4; ++kb) for (int nz = 0; nz < 2; ++nz) {
    for (size_t i = 0; i < n; ++i) { ha[i] = rnd_bits(ka); hb[i] = rnd_bits(kb); }
    hipMemcpy(da, ha.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), n * 4, hipMemcpyHostToDevice); hipMemset(dbad, 0, 16);
    hipLaunchKernelGGL(k_mfma_exact, dim3(1024), dim3(256), 0, 0, da, db, n, dbad, dbad + 1, nz);
    unsigned long long hbad[2]; hipMemcpy(hbad, dbad, 16, hipMemcpyDeviceToHost);
    printf("A kind %d x B kind %d, C = %s: %llu products differ, %llu differ only in the sign of zero\n", ka, kb, nz ? "-0.0" : "+0.0", hbad[0], hbad[1]);
  }
  return 0;
}
