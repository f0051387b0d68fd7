// Development tool: per-instruction VALU issue cost on gfx950 (cycles per wave64 instruction per SIMD).
// hipcc --offload-arch=gfx950 -O2 tools/ubench.hip -o tools/build/ubench && tools/build/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define ITERS 2048
#define UNROLL 16

#define DEF_KERNEL(NAME, DECL, ASM)                                                             \
__global__ __launch_bounds__(256) void k_##NAME(float *out, unsigned long long *cyc, float c) {   \
  DECL;                                                                                          \
  unsigned long long t0 = __builtin_readcyclecounter();                                          \
  for (int i = 0; i < ITERS; ++i) {                                                              \
    _Pragma("unroll") for (int u = 0; u < UNROLL / 8; ++u) { ASM }                               \
  }                                                                                              \
  unsigned long long t1 = __builtin_readcyclecounter();                                          \
  SINK;                                                                                          \
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;                                       \
}

// 8 independent f32 chains
#define DECL_F32 float a0 = c, a1 = c + 1, a2 = c + 2, a3 = c + 3, a4 = c + 4, a5 = c + 5, a6 = c + 6, a7 = c + 7
#define SINK out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7
#define OP2(INS) \
  asm volatile(INS " %0, %0, %1" : "+v"(a0) : "v"(c)); asm volatile(INS " %0, %0, %1" : "+v"(a1) : "v"(c)); \
  asm volatile(INS " %0, %0, %1" : "+v"(a2) : "v"(c)); asm volatile(INS " %0, %0, %1" : "+v"(a3) : "v"(c)); \
  asm volatile(INS " %0, %0, %1" : "+v"(a4) : "v"(c)); asm volatile(INS " %0, %0, %1" : "+v"(a5) : "v"(c)); \
  asm volatile(INS " %0, %0, %1" : "+v"(a6) : "v"(c)); asm volatile(INS " %0, %0, %1" : "+v"(a7) : "v"(c));
#define OP3(INS) \
  asm volatile(INS " %0, %0, %1, %1" : "+v"(a0) : "v"(c)); asm volatile(INS " %0, %0, %1, %1" : "+v"(a1) : "v"(c)); \
  asm volatile(INS " %0, %0, %1, %1" : "+v"(a2) : "v"(c)); asm volatile(INS " %0, %0, %1, %1" : "+v"(a3) : "v"(c)); \
  asm volatile(INS " %0, %0, %1, %1" : "+v"(a4) : "v"(c)); asm volatile(INS " %0, %0, %1, %1" : "+v"(a5) : "v"(c)); \
  asm volatile(INS " %0, %0, %1, %1" : "+v"(a6) : "v"(c)); asm volatile(INS " %0, %0, %1, %1" : "+v"(a7) : "v"(c));
#define OP1(INS) \
  asm volatile(INS " %0, %0" : "+v"(a0)); asm volatile(INS " %0, %0" : "+v"(a1)); asm volatile(INS " %0, %0" : "+v"(a2)); asm volatile(INS " %0, %0" : "+v"(a3)); \
  asm volatile(INS " %0, %0" : "+v"(a4)); asm volatile(INS " %0, %0" : "+v"(a5)); asm volatile(INS " %0, %0" : "+v"(a6)); asm volatile(INS " %0, %0" : "+v"(a7));

DEF_KERNEL(v_mul_f32, DECL_F32, OP2("v_mul_f32"))
DEF_KERNEL(v_add_f32, DECL_F32, OP2("v_add_f32"))
DEF_KERNEL(v_min_f32, DECL_F32, OP2("v_min_f32"))
DEF_KERNEL(v_fma_f32, DECL_F32, OP3("v_fma_f32"))
DEF_KERNEL(v_med3_f32, DECL_F32, OP3("v_med3_f32"))
DEF_KERNEL(v_div_fixup_f32, DECL_F32, OP3("v_div_fixup_f32"))
#define OPCND \
  asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a0) : "v"(c), "s"(msk)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a1) : "v"(c), "s"(msk)); \
  asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a2) : "v"(c), "s"(msk)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a3) : "v"(c), "s"(msk)); \
  asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a4) : "v"(c), "s"(msk)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a5) : "v"(c), "s"(msk)); \
  asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a6) : "v"(c), "s"(msk)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a7) : "v"(c), "s"(msk));
#define DECL_F32M DECL_F32; unsigned long long msk = __builtin_amdgcn_ballot_w64(c > 0.5f) ^ 0x5555555555555555ull
DEF_KERNEL(v_cndmask_b32, DECL_F32M, OPCND)
DEF_KERNEL(v_sub_f32, DECL_F32, OP2("v_sub_f32"))
DEF_KERNEL(v_max_f32, DECL_F32, OP2("v_max_f32"))
DEF_KERNEL(v_mov_b32, DECL_F32, OP1("v_mov_b32"))
DEF_KERNEL(v_cvt_f32_u32, DECL_F32, OP1("v_cvt_f32_u32"))
DEF_KERNEL(v_lshlrev_b32, DECL_F32, OP2("v_lshlrev_b32"))
DEF_KERNEL(v_lshl_add_u32, DECL_F32, OP3("v_lshl_add_u32"))
DEF_KERNEL(v_mbcnt_lo, DECL_F32, OP2("v_mbcnt_lo_u32_b32"))
DEF_KERNEL(v_min3_f32, DECL_F32, OP3("v_min3_f32"))
DEF_KERNEL(v_mad_f32_legacyfree, DECL_F32, OP3("v_mad_u32_u24"))
DEF_KERNEL(v_cvt_u32_f32, DECL_F32, OP1("v_cvt_u32_f32"))
DEF_KERNEL(v_trunc_f32, DECL_F32, OP1("v_trunc_f32"))
DEF_KERNEL(v_fract_f32, DECL_F32, OP1("v_fract_f32"))
DEF_KERNEL(v_rcp_f32, DECL_F32, OP1("v_rcp_f32"))
DEF_KERNEL(v_frexp_exp_i32_f32, DECL_F32, OP1("v_frexp_exp_i32_f32"))
DEF_KERNEL(v_frexp_mant_f32, DECL_F32, OP1("v_frexp_mant_f32"))
DEF_KERNEL(v_ldexp_f32, DECL_F32, OP2("v_ldexp_f32"))
DEF_KERNEL(v_and_b32, DECL_F32, OP2("v_and_b32"))
DEF_KERNEL(v_add_u32, DECL_F32, OP2("v_add_u32"))
#define OPDPP \
  asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0)); asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a1)); \
  asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a2)); asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a3)); \
  asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a4)); asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a5)); \
  asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a6)); asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a7));
DEF_KERNEL(v_mov_dpp, DECL_F32, OPDPP)
#define OPCMP(INS) \
  asm volatile(INS " vcc, %0, %1" :: "v"(a0), "v"(c) : "vcc"); asm volatile(INS " vcc, %0, %1" :: "v"(a1), "v"(c) : "vcc"); \
  asm volatile(INS " vcc, %0, %1" :: "v"(a2), "v"(c) : "vcc"); asm volatile(INS " vcc, %0, %1" :: "v"(a3), "v"(c) : "vcc"); \
  asm volatile(INS " vcc, %0, %1" :: "v"(a4), "v"(c) : "vcc"); asm volatile(INS " vcc, %0, %1" :: "v"(a5), "v"(c) : "vcc"); \
  asm volatile(INS " vcc, %0, %1" :: "v"(a6), "v"(c) : "vcc"); asm volatile(INS " vcc, %0, %1" :: "v"(a7), "v"(c) : "vcc");
DEF_KERNEL(v_cmp_gt_f32, DECL_F32, OPCMP("v_cmp_gt_f32"))
#undef SINK
#undef DECL_F32

// packed f32: 8 chains of float2
typedef float f2 __attribute__((ext_vector_type(2)));
#define DECL_PK f2 a0 = {c, c}, a1 = {c + 1, c}, a2 = {c + 2, c}, a3 = {c + 3, c}, a4 = {c + 4, c}, a5 = {c + 5, c}, a6 = {c + 6, c}, a7 = {c + 7, c}; f2 cc = {c, c}
#define SINK { f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7; out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y; }
#define OPPK2(INS) \
  asm volatile(INS " %0, %0, %1" : "+v"(a0) : "v"(cc)); asm volatile(INS " %0, %0, %1" : "+v"(a1) : "v"(cc)); \
  asm volatile(INS " %0, %0, %1" : "+v"(a2) : "v"(cc)); asm volatile(INS " %0, %0, %1" : "+v"(a3) : "v"(cc)); \
  asm volatile(INS " %0, %0, %1" : "+v"(a4) : "v"(cc)); asm volatile(INS " %0, %0, %1" : "+v"(a5) : "v"(cc)); \
  asm volatile(INS " %0, %0, %1" : "+v"(a6) : "v"(cc)); asm volatile(INS " %0, %0, %1" : "+v"(a7) : "v"(cc));
#define OPPK3(INS) \
  asm volatile(INS " %0, %0, %1, %1" : "+v"(a0) : "v"(cc)); asm volatile(INS " %0, %0, %1, %1" : "+v"(a1) : "v"(cc)); \
  asm volatile(INS " %0, %0, %1, %1" : "+v"(a2) : "v"(cc)); asm volatile(INS " %0, %0, %1, %1" : "+v"(a3) : "v"(cc)); \
  asm volatile(INS " %0, %0, %1, %1" : "+v"(a4) : "v"(cc)); asm volatile(INS " %0, %0, %1, %1" : "+v"(a5) : "v"(cc)); \
  asm volatile(INS " %0, %0, %1, %1" : "+v"(a6) : "v"(cc)); asm volatile(INS " %0, %0, %1, %1" : "+v"(a7) : "v"(cc));
DEF_KERNEL(v_pk_mul_f32, DECL_PK, OPPK2("v_pk_mul_f32"))
DEF_KERNEL(v_pk_add_f32, DECL_PK, OPPK2("v_pk_add_f32"))
DEF_KERNEL(v_pk_fma_f32, DECL_PK, OPPK3("v_pk_fma_f32"))
#undef SINK

// f64: 8 chains
#define DECL_F64 double a0 = c, a1 = c + 1, a2 = c + 2, a3 = c + 3, a4 = c + 4, a5 = c + 5, a6 = c + 6, a7 = c + 7; double cc = c
#define SINK out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)
DEF_KERNEL(v_mul_f64, DECL_F64, OPPK2("v_mul_f64"))
DEF_KERNEL(v_add_f64, DECL_F64, OPPK2("v_add_f64"))
DEF_KERNEL(v_fma_f64, DECL_F64, OPPK3("v_fma_f64"))
#define OPRCP64 \
  asm volatile("v_rcp_f64 %0, %0" : "+v"(a0)); asm volatile("v_rcp_f64 %0, %0" : "+v"(a1)); asm volatile("v_rcp_f64 %0, %0" : "+v"(a2)); asm volatile("v_rcp_f64 %0, %0" : "+v"(a3)); \
  asm volatile("v_rcp_f64 %0, %0" : "+v"(a4)); asm volatile("v_rcp_f64 %0, %0" : "+v"(a5)); asm volatile("v_rcp_f64 %0, %0" : "+v"(a6)); asm volatile("v_rcp_f64 %0, %0" : "+v"(a7));
DEF_KERNEL(v_rcp_f64, DECL_F64, OPRCP64)
#undef SINK

struct Entry { const char *name; void (*fn)(float *, unsigned long long *, float); };
#define E(NAME) {#NAME, k_##NAME}

int main() {
  std::vector<Entry> es = {E(v_mul_f32), E(v_add_f32), E(v_min_f32), E(v_fma_f32), E(v_med3_f32), E(v_div_fixup_f32), E(v_cndmask_b32), E(v_sub_f32), E(v_max_f32), E(v_mov_b32), E(v_cvt_f32_u32), E(v_lshlrev_b32), E(v_lshl_add_u32), E(v_mbcnt_lo), E(v_min3_f32), E(v_mad_f32_legacyfree),
                           E(v_cvt_u32_f32), E(v_trunc_f32), E(v_fract_f32), E(v_rcp_f32), E(v_frexp_exp_i32_f32), E(v_frexp_mant_f32), E(v_ldexp_f32),
                           E(v_and_b32), E(v_add_u32), E(v_mov_dpp), E(v_cmp_gt_f32), E(v_pk_mul_f32), E(v_pk_add_f32), E(v_pk_fma_f32),
                           E(v_mul_f64), E(v_add_f64), E(v_fma_f64), E(v_rcp_f64)};
  float *out; unsigned long long *cyc;
  hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&cyc, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wpsimd : {4}) {
    // grid: 256 CUs x wpsimd blocks of 256 threads (one wave per SIMD each)
    const int blocks = 256 * wpsimd;
    printf("--- %d wave(s) per SIMD ---\n", wpsimd);
    for (auto &e : es) {
      hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0001f);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0001f);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
      const double n = (double)ITERS * UNROLL;               // instructions per wave
      printf("%-22s wave-cycles/instr %.2f   (kernel %.3f ms; %.2f ns per instr per SIMD => %.2f cycles @2.4GHz per SIMD-instr)\n", e.name,
             (double)c / n, ms, ms * 1e6 / (n * wpsimd), ms * 1e6 / (n * wpsimd) * 2.4);
    }
  }
  return 0;
}
