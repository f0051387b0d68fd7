// Development tool: which VALU instructions issue at the fast (~2.75 cycle) rate on gfx950, and under what conditions.
// hipcc --offload-arch=gfx950 -O2 tools/ubench2.hip -o tools/build/ubench2 && tools/build/ubench2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITERS 4096

#define DEF_KERNEL(NAME, BODY, NINS)                                                              \
__global__ __launch_bounds__(256) void k_##NAME(float *out, float c) {                             \
  float a0 = c, a1 = c + 1, a2 = c + 2, a3 = c + 3, a4 = c + 4, a5 = c + 5, a6 = c + 6, a7 = c + 7; \
  float b0 = c * 2, b1 = c * 3, b2 = c * 4, b3 = c * 5, b4 = c * 6, b5 = c * 7, b6 = c * 8, b7 = c * 9; \
  float k = c * 0.5f; unsigned long long msk = __builtin_amdgcn_ballot_w64(c > 0.5f) ^ 0x5555555555555555ull; \
  for (int i = 0; i < ITERS; ++i) { BODY }                                                         \
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7; \
}                                                                                                   \
static const int n_##NAME = NINS;

#define A1(I, X) asm volatile(I : "+v"(X) : "v"(k), "s"(msk) : "vcc");
#define ALL8(I) A1(I, a0) A1(I, a1) A1(I, a2) A1(I, a3) A1(I, a4) A1(I, a5) A1(I, a6) A1(I, a7)
#define ALL16(I) ALL8(I) A1(I, b0) A1(I, b1) A1(I, b2) A1(I, b3) A1(I, b4) A1(I, b5) A1(I, b6) A1(I, b7)
#define ALT8(I, J) A1(I, a0) A1(J, a1) A1(I, a2) A1(J, a3) A1(I, a4) A1(J, a5) A1(I, a6) A1(J, a7)
#define ALT16(I, J) ALT8(I, J) A1(I, b0) A1(J, b1) A1(I, b2) A1(J, b3) A1(I, b4) A1(J, b5) A1(I, b6) A1(J, b7)
// non-destructive: d = op(x, k) where d != x  (pairs a_i <- b_i)
#define N1(I, D, X) asm volatile(I : "=v"(D) : "v"(X), "v"(k));
#define ND16(I) N1(I, a0, b0) N1(I, a1, b1) N1(I, a2, b2) N1(I, a3, b3) N1(I, a4, b4) N1(I, a5, b5) N1(I, a6, b6) N1(I, a7, b7) \
                N1(I, b0, a0) N1(I, b1, a1) N1(I, b2, a2) N1(I, b3, a3) N1(I, b4, a4) N1(I, b5, a5) N1(I, b6, a6) N1(I, b7, a7)

DEF_KERNEL(mul,        ALL16("v_mul_f32 %0, %0, %1"), 16)
DEF_KERNEL(mul_e64,    ALL16("v_mul_f32_e64 %0, %0, %1"), 16)
DEF_KERNEL(mul_nd,     ND16("v_mul_f32 %0, %1, %2"), 16)
DEF_KERNEL(mul_const,  ALL16("v_mul_f32 %0, 0x3f8ccccd, %0"), 16)
DEF_KERNEL(mul_inl,    ALL16("v_mul_f32 %0, 2.0, %0"), 16)
DEF_KERNEL(add,        ALL16("v_add_f32 %0, %0, %1"), 16)
DEF_KERNEL(fma,        ALL16("v_fma_f32 %0, %0, %1, %1"), 16)
DEF_KERNEL(fma_3reg,   ND16("v_fma_f32 %0, %1, %2, %2"), 16)
DEF_KERNEL(fmac,       ALL16("v_fmac_f32 %0, %1, %1"), 16)
DEF_KERNEL(fmac_lit,   ALL16("v_fmac_f32 %0, 0x3f8ccccd, %1"), 16)
DEF_KERNEL(fmaak,      ALL16("v_fmaak_f32 %0, %0, %1, 0x3f8ccccd"), 16)
DEF_KERNEL(fmamk,      ALL16("v_fmamk_f32 %0, %0, 0x3f8ccccd, %1"), 16)
DEF_KERNEL(min,        ALL16("v_min_f32 %0, %0, %1"), 16)
DEF_KERNEL(max,        ALL16("v_max_f32 %0, %0, %1"), 16)
DEF_KERNEL(cnd_e32,    ALL16("v_cndmask_b32 %0, %0, %1, vcc"), 16)
DEF_KERNEL(cnd_e64,    ALL16("v_cndmask_b32_e64 %0, %0, %1, %2"), 16)
DEF_KERNEL(or_b32,     ALL16("v_or_b32 %0, %0, %1"), 16)
DEF_KERNEL(xor_b32,    ALL16("v_xor_b32 %0, %0, %1"), 16)
DEF_KERNEL(sub_u32,    ALL16("v_sub_u32 %0, %0, %1"), 16)
DEF_KERNEL(lshr,       ALL16("v_lshrrev_b32 %0, 3, %0"), 16)
DEF_KERNEL(lshl,       ALL16("v_lshlrev_b32 %0, 3, %0"), 16)
DEF_KERNEL(lshl_add,   ALL16("v_lshl_add_u32 %0, %0, 3, %1"), 16)
DEF_KERNEL(add3,       ALL16("v_add3_u32 %0, %0, %1, %1"), 16)
DEF_KERNEL(and_or,     ALL16("v_and_or_b32 %0, %0, %1, %1"), 16)
DEF_KERNEL(bfe,        ALL16("v_bfe_u32 %0, %0, 3, 5"), 16)
DEF_KERNEL(mul_u24,    ALL16("v_mul_u32_u24 %0, %0, %1"), 16)
DEF_KERNEL(mad_u24,    ALL16("v_mad_u32_u24 %0, %0, %1, %1"), 16)
DEF_KERNEL(cvt_u32,    ALL16("v_cvt_u32_f32 %0, %0"), 16)
DEF_KERNEL(cvt_i32,    ALL16("v_cvt_i32_f32 %0, %0"), 16)
DEF_KERNEL(cvt_f32u,   ALL16("v_cvt_f32_u32 %0, %0"), 16)
DEF_KERNEL(fract,      ALL16("v_fract_f32 %0, %0"), 16)
DEF_KERNEL(floor,      ALL16("v_floor_f32 %0, %0"), 16)
DEF_KERNEL(med3,       ALL16("v_med3_f32 %0, %0, 0, 1.0"), 16)
DEF_KERNEL(cmp,        ALL16("v_cmp_gt_f32 vcc, %0, %1"), 16)
DEF_KERNEL(cmp_e64,    ALL16("v_cmp_gt_f32_e64 s[20:21], %0, %1"), 16)
DEF_KERNEL(mul_legacy, ALL16("v_mul_legacy_f32 %0, %0, %1"), 16)
DEF_KERNEL(subrev,     ALL16("v_subrev_f32 %0, %0, %1"), 16)
DEF_KERNEL(mul_neg,    ALL16("v_mul_f32_e64 %0, -%0, %1"), 16)
DEF_KERNEL(add_abs,    ALL16("v_add_f32_e64 %0, |%0|, %1"), 16)
DEF_KERNEL(mul_clamp,  ALL16("v_mul_f32_e64 %0, %0, %1 clamp"), 16)
DEF_KERNEL(pk_mul,     asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(double*)&a0) : "v"(*(double*)&b0)); asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(double*)&a2) : "v"(*(double*)&b0)); asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(double*)&a4) : "v"(*(double*)&b0)); asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(double*)&a6) : "v"(*(double*)&b0));, 4)
// mixes
DEF_KERNEL(mix_mul_fma,  ALT16("v_mul_f32 %0, %0, %1", "v_fma_f32 %0, %0, %1, %1"), 16)
DEF_KERNEL(mix_mul_min,  ALT16("v_mul_f32 %0, %0, %1", "v_min_f32 %0, %0, %1"), 16)
DEF_KERNEL(mix_mul_add,  ALT16("v_mul_f32 %0, %0, %1", "v_add_f32 %0, %0, %1"), 16)
DEF_KERNEL(mix_fma_min,  ALT16("v_fma_f32 %0, %0, %1, %1", "v_min_f32 %0, %0, %1"), 16)

#define SEQ4x4(I, J, K, L) A1(I, a0) A1(J, a1) A1(K, a2) A1(L, a3) A1(I, a4) A1(J, a5) A1(K, a6) A1(L, a7) A1(I, b0) A1(J, b1) A1(K, b2) A1(L, b3) A1(I, b4) A1(J, b5) A1(K, b6) A1(L, b7)
#define MUL "v_mul_f32 %0, %0, %1"
#define ADD "v_add_f32 %0, %0, %1"
#define MIN "v_min_f32 %0, %0, %1"
#define MAX "v_max_f32 %0, %0, %1"
#define CVT "v_cvt_u32_f32 %0, %0"
#define CVTF "v_cvt_f32_u32 %0, %0"
#define FRA "v_fract_f32 %0, %0"
#define CMP "v_cmp_gt_f32 vcc, %0, %1"
#define CND "v_cndmask_b32_e64 %0, %0, %1, %2"
#define LSHL "v_lshlrev_b32 %0, 3, %0"
#define FMA "v_fma_f32 %0, %0, %1, %1"
#define MED "v_med3_f32 %0, %0, 0, 1.0"
DEF_KERNEL(mix_min_max,  ALT16(MIN, MAX), 16)
DEF_KERNEL(mix_min_cvt,  ALT16(MIN, CVT), 16)
DEF_KERNEL(mix_min_cmp,  ALT16(MIN, CMP), 16)
DEF_KERNEL(mix_cvt_fra,  ALT16(CVT, FRA), 16)
DEF_KERNEL(mix_min_lshl, ALT16(MIN, LSHL), 16)
DEF_KERNEL(mix_cnd_min,  ALT16(CND, MIN), 16)
DEF_KERNEL(mix_cmp_cnd,  ALT16(CMP, CND), 16)
DEF_KERNEL(mix_cvt_cvtf, ALT16(CVT, CVTF), 16)
DEF_KERNEL(mix_med_fra,  ALT16(MED, FRA), 16)
DEF_KERNEL(seq_mmmn,     SEQ4x4(MUL, MUL, MUL, MIN), 16)
DEF_KERNEL(seq_mmnn,     SEQ4x4(MUL, ADD, MIN, MIN), 16)
DEF_KERNEL(seq_mncf,     SEQ4x4(MUL, MIN, CVT, FRA), 16)
DEF_KERNEL(seq_nnmm_blk, A1(MIN, a0) A1(MIN, a1) A1(MIN, a2) A1(MIN, a3) A1(MIN, a4) A1(MIN, a5) A1(MIN, a6) A1(MIN, a7) A1(MUL, b0) A1(MUL, b1) A1(MUL, b2) A1(MUL, b3) A1(MUL, b4) A1(MUL, b5) A1(MUL, b6) A1(MUL, b7), 16)
DEF_KERNEL(seq_fmul,     SEQ4x4(FMA, MUL, FMA, ADD), 16)
DEF_KERNEL(seq_real,     SEQ4x4(MUL, CVT, FRA, LSHL) SEQ4x4(MUL, ADD, MIN, MUL) SEQ4x4(FMA, MUL, ADD, CMP) SEQ4x4(CND, MUL, ADD, MED), 64)

// fully dependent realistic mix (one register)
#define SEQ4d(I, J, K, L) A1(I, a0) A1(J, a0) A1(K, a0) A1(L, a0)
DEF_KERNEL(dep_real,   SEQ4d(MUL, CVT, FRA, LSHL) SEQ4d(MUL, ADD, MIN, MUL) SEQ4d(FMA, MUL, ADD, CMP) SEQ4d(CND, MUL, ADD, MED) SEQ4d(MUL, CVT, FRA, LSHL) SEQ4d(MUL, ADD, MIN, MUL) SEQ4d(FMA, MUL, ADD, CMP) SEQ4d(CND, MUL, ADD, MED), 32)
// two independent dependent chains
#define SEQ4d2(I, J, K, L) A1(I, a0) A1(I, a1) A1(J, a0) A1(J, a1) A1(K, a0) A1(K, a1) A1(L, a0) A1(L, a1)
DEF_KERNEL(dep2_real,  SEQ4d2(MUL, CVT, FRA, LSHL) SEQ4d2(MUL, ADD, MIN, MUL) SEQ4d2(FMA, MUL, ADD, CMP) SEQ4d2(CND, MUL, ADD, MED), 32)
// two varying VGPR sources
#define V2(I, D, X, Y) asm volatile(I : "=v"(D) : "v"(X), "v"(Y));
#define VV16(I) V2(I, a0, b0, b1) V2(I, a1, b1, b2) V2(I, a2, b2, b3) V2(I, a3, b3, b4) V2(I, a4, b4, b5) V2(I, a5, b5, b6) V2(I, a6, b6, b7) V2(I, a7, b7, b0) \
                V2(I, b0, a0, a1) V2(I, b1, a1, a2) V2(I, b2, a2, a3) V2(I, b3, a3, a4) V2(I, b4, a4, a5) V2(I, b5, a5, a6) V2(I, b6, a6, a7) V2(I, b7, a7, a0)
DEF_KERNEL(mul_vv,     VV16("v_mul_f32 %0, %1, %2"), 16)
DEF_KERNEL(add_vv,     VV16("v_add_f32 %0, %1, %2"), 16)
DEF_KERNEL(min_vv,     VV16("v_min_f32 %0, %1, %2"), 16)
DEF_KERNEL(fma_vvv,    VV16("v_fma_f32 %0, %1, %2, %1"), 16)
// SGPR operand
#define S1(I, X) asm volatile(I : "+v"(X) : "s"(ks));
#define S16(I) S1(I, a0) S1(I, a1) S1(I, a2) S1(I, a3) S1(I, a4) S1(I, a5) S1(I, a6) S1(I, a7) S1(I, b0) S1(I, b1) S1(I, b2) S1(I, b3) S1(I, b4) S1(I, b5) S1(I, b6) S1(I, b7)
DEF_KERNEL(mul_sgpr,   float ks = __builtin_amdgcn_readfirstlane(k); S16("v_mul_f32 %0, %1, %0"), 16)
DEF_KERNEL(fma_sgpr,   float ks = __builtin_amdgcn_readfirstlane(k); S16("v_fma_f32 %0, %0, %1, %1"), 16)
DEF_KERNEL(fma_sgpr2,  float ks = __builtin_amdgcn_readfirstlane(k); S16("v_fma_f32 %0, %0, %1, %0"), 16)
// dependent chain: all 16 on the same register
#define DEP16(I) A1(I, a0) A1(I, a0) A1(I, a0) A1(I, a0) A1(I, a0) A1(I, a0) A1(I, a0) A1(I, a0) A1(I, a0) A1(I, a0) A1(I, a0) A1(I, a0) A1(I, a0) A1(I, a0) A1(I, a0) A1(I, a0)
DEF_KERNEL(dep_mul,    DEP16("v_mul_f32 %0, %0, %1"), 16)
DEF_KERNEL(dep_fma,    DEP16("v_fma_f32 %0, %0, %1, %1"), 16)
DEF_KERNEL(dep_min,    DEP16("v_min_f32 %0, %0, %1"), 16)

// ---- round 2 additions: candidates for replacing slow-class instructions, and the f64 conversions of the cbrtf branch ----
DEF_KERNEL(alignbit,   ALL16("v_alignbit_b32 %0, %0, %1, 30"), 16)
DEF_KERNEL(perm,       ALL16("v_perm_b32 %0, %0, %1, %1"), 16)
DEF_KERNEL(bfi,        ALL16("v_bfi_b32 %0, %0, %1, %1"), 16)
DEF_KERNEL(add_lshl,   ALL16("v_add_lshl_u32 %0, %0, %1, 2"), 16)
DEF_KERNEL(max3_u32,   ALL16("v_max3_u32 %0, %0, %1, %1"), 16)
DEF_KERNEL(max_u32,    ALL16("v_max_u32 %0, %0, %1"), 16)
DEF_KERNEL(cmp_u32,    ALL16("v_cmp_lt_u32 vcc, %0, %1"), 16)
DEF_KERNEL(ashr,       ALL16("v_ashrrev_i32 %0, 3, %0"), 16)
DEF_KERNEL(mul_lo,     ALL16("v_mul_lo_u32 %0, %0, %1"), 16)
DEF_KERNEL(rcp_f32,    ALL16("v_rcp_f32 %0, %0"), 16)
DEF_KERNEL(trunc,      ALL16("v_trunc_f32 %0, %0"), 16)
DEF_KERNEL(rndne,      ALL16("v_rndne_f32 %0, %0"), 16)
DEF_KERNEL(ldexp,      ALL16("v_ldexp_f32 %0, %0, 2"), 16)
DEF_KERNEL(cvt_pk_u8,  ALL16("v_cvt_pk_u8_f32 %0, %1, 0, %0"), 16)
// the same multiply / fma on DENORMAL operands (integers held as f32 bit patterns): is `as_float(key) * 4.0f` a full-rate shift?
#define DEN16 a0 = __int_as_float(1 + (int)c); a1 = __int_as_float(2 + (int)c); a2 = __int_as_float(3); a3 = __int_as_float(4); a4 = __int_as_float(5); a5 = __int_as_float(6); a6 = __int_as_float(7); a7 = __int_as_float(8); \
              b0 = __int_as_float(9); b1 = __int_as_float(10); b2 = __int_as_float(11); b3 = __int_as_float(12); b4 = __int_as_float(13); b5 = __int_as_float(14); b6 = __int_as_float(15); b7 = __int_as_float(16);
DEF_KERNEL(mul_denorm,   if (i == 0) { DEN16 } ALL16("v_mul_f32 %0, 1.0, %0"), 16)
DEF_KERNEL(fmaak_denorm, if (i == 0) { DEN16 } ALL16("v_fmaak_f32 %0, %0, %1, 0x00000040"), 16)
DEF_KERNEL(add_denorm,   if (i == 0) { DEN16 } ALL16("v_add_f32 %0, %0, %0"), 16)
// f32 <-> f64 conversions (alternating, 16 instructions), f64 rcp, f64 fma with distinct registers
#define DD8 double d0; double d1; double d2; double d3; double d4; double d5; double d6; double d7;
#define CVTP(X, D) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(D) : "v"(X)); asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(X) : "v"(D));
DEF_KERNEL(cvt_f64_pair, DD8 CVTP(a0, d0) CVTP(a1, d1) CVTP(a2, d2) CVTP(a3, d3) CVTP(a4, d4) CVTP(a5, d5) CVTP(a6, d6) CVTP(a7, d7), 16)
#define CVTU(X, D) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(D) : "v"(X)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(X) : "v"(*(float *)&D));
DEF_KERNEL(cvt_f64_up,   DD8 CVTU(a0, d0) CVTU(a1, d1) CVTU(a2, d2) CVTU(a3, d3) CVTU(a4, d4) CVTU(a5, d5) CVTU(a6, d6) CVTU(a7, d7), 16)

struct Entry { const char *name; void (*fn)(float *, float); int nins; };
#define E(NAME) {#NAME, k_##NAME, n_##NAME}

int main(int argc, char **argv) {
  std::vector<Entry> es = {E(mul), E(mul_e64), E(mul_nd), E(mul_const), E(mul_inl), E(add), E(fma), E(fma_3reg), E(fmac), E(fmac_lit), E(fmaak), E(fmamk), E(min), E(max),
                           E(cnd_e32), E(cnd_e64), E(or_b32), E(xor_b32), E(sub_u32), E(lshr), E(lshl), E(lshl_add), E(add3), E(and_or), E(bfe), E(mul_u24), E(mad_u24),
                           E(cvt_u32), E(cvt_i32), E(cvt_f32u), E(fract), E(floor), E(med3), E(cmp), E(cmp_e64), E(mul_legacy), E(subrev), E(mul_neg), E(add_abs), E(mul_clamp), E(pk_mul),
                           E(mix_mul_fma), E(mix_mul_min), E(mix_mul_add), E(mix_fma_min), E(dep_mul), E(dep_fma), E(dep_min), E(mix_min_max), E(mix_min_cvt), E(mix_min_cmp), E(mix_cvt_fra), E(mix_min_lshl), E(mix_cnd_min), E(mix_cmp_cnd), E(mix_cvt_cvtf), E(mix_med_fra), E(seq_mmmn), E(seq_mmnn), E(seq_mncf), E(seq_nnmm_blk), E(seq_fmul), E(seq_real), E(dep_real), E(dep2_real), E(mul_vv), E(add_vv), E(min_vv), E(fma_vvv), E(mul_sgpr), E(fma_sgpr), E(fma_sgpr2),
                           E(alignbit), E(perm), E(bfi), E(add_lshl), E(max3_u32), E(max_u32), E(cmp_u32), E(ashr), E(mul_lo), E(rcp_f32), E(trunc), E(rndne), E(ldexp), E(cvt_pk_u8),
                           E(mul_denorm), E(fmaak_denorm), E(add_denorm), E(cvt_f64_pair), E(cvt_f64_up)};
  float *out; hipMalloc(&out, 8192 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wpsimd : {4}) {
    const int blocks = 256 * wpsimd;                       // 256-thread blocks: one wave per SIMD each
    printf("--- %d wave(s) per SIMD ---\n", wpsimd);
    for (auto &e : es) {
      if (wpsimd != 4 && e.fn != k_mul && e.fn != k_fma && e.fn != k_min && e.fn != k_mix_mul_fma && e.fn != k_dep_mul && e.fn != k_dep_fma && e.fn != k_pk_mul) continue;
      hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, 1.0001f);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, 1.0001f);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double n = (double)ITERS * e.nins;               // instructions per wave
      printf("%-14s %.3f ms  %.2f ns per instr per SIMD\n", e.name, ms, ms * 1e6 / (n * wpsimd));
    }
  }
  return 0;
}
