// ipk_device.hpp -- per-pixel device functions shared by the staged and fused gfx950 kernels.
//
// Bit-exactness contract (SURVEY.md section 7 "hard parts"): every function evaluates the same
// IEEE binary32 operations in the same order as the reference's Rust.  The translation unit is
// compiled with -ffp-contract=off (no FMA contraction), without fast-math, with IEEE division
// (hipcc default) and f32 denormals enabled (gfx9 default).  Reference citations are file:line
// in pedrocr/imagepipe 0.5.0.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ipkd {

// One TransformLookup entry as the kernels hold it in LDS: {table[i], table[i+1]-table[i]}.
// The difference is the same f32 subtraction lookup() performs (src/color_conversions.rs:112),
// hoisted to table-build time, so one 8-byte ds_read_b64 feeds the whole interpolation.
typedef float2 LutPair;
constexpr int kLutPairs = 8192;          // keys 0..=8191 (pos = val*8191, val in [0,1])
constexpr float kLutMaxF = 8191.0f;

// D65 white (src/color_conversions.rs:7)
constexpr float kWhiteX = 0.95047f, kWhiteY = 1.000f, kWhiteZ = 1.08883f;
// e and k of the CIE Lab transfer (src/color_conversions.rs:121-122, :181-182), f32 divisions
constexpr float kLabE = 216.0f / 24389.0f;
constexpr float kLabK = 24389.0f / 27.0f;

// ---- Rust float semantics ------------------------------------------------------------------
// f32::min/max ignore a NaN operand == IEEE minNum/maxNum == fminf/fmaxf (v_min_f32/v_max_f32).
__device__ __forceinline__ float rs_min(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ float rs_max(float a, float b) { return fmaxf(a, b); }
// `f as usize` clamped to a 32-bit range (callers min() it with a dimension < 2^31 afterwards):
// v_cvt_u32_f32 saturates, maps NaN and negatives to 0.
__device__ __forceinline__ uint32_t f32_as_u32_sat(float f) { return __float2uint_rz(f); }

// ---- TransformLookup::lookup table branch (src/color_conversions.rs:106-112) ---------------
// Precondition: !(val < 0 || val > 1)  (NaN takes this branch too, as in the reference: key 0,
// a = NaN -> NaN).
// A table is either {v[i], v[i+1]-v[i]} pairs (one 8-byte read per lookup) or the plain 8193 floats (two adjacent
// 4-byte reads, ds_read2_b32, and the subtraction of :112 done per lookup: half the LDS footprint).
__device__ __forceinline__ LutPair lut_pair_at(const LutPair *__restrict__ tab, uint32_t key) { return tab[key]; }
__device__ __forceinline__ LutPair lut_pair_at(const float *__restrict__ tab, uint32_t key) {
  const float v1 = tab[key], v2 = tab[key + 1];
  return make_float2(v1, v2 - v1);
}
template <typename Tab>
__device__ __forceinline__ float lut_interp(const Tab *__restrict__ tab, float val) {
  const float pos = val * kLutMaxF;
  const uint32_t key = f32_as_u32_sat(pos);
  const float base = truncf(pos);
  const float a = pos - base;
  const LutPair p = lut_pair_at(tab, key);
  return p.x + a * p.y;
}

// ---- cbrtf, bit-identical to glibc 2.35 sysdeps/ieee754/flt-32/s_cbrtf.c -------------------
// Rust's f32::cbrt is the platform libm's cbrtf; XYZ_LAB_TRANSFORM calls it at run time for
// ratios > 1 (src/color_conversions.rs:103-104,123).  glibc's routine is not correctly rounded, so
// the kernel reproduces its arithmetic (frexp, a degree-2 polynomial and one Halley step in
// double, ldexp) rather than calling a different cbrt.  tests/test_gpu_selftest.py checks it against
// the host libm over every f32 in (1, 2^64].  Called only with x > 1 (finite or +inf).
__device__ __forceinline__ float cbrtf_glibc(float x) {
  if (__builtin_isinf(x)) return x;
  int xe;
  const float xm = frexpf(x, &xe);                       // xm in [0.5, 1)
  const double dxm = (double)xm;
  const float u = (float)(0.492659620528969547 + (0.697570460207922770 - 0.191502161678719066 * dxm) * dxm);
  const float t2 = u * u * u;
  const int rem = xe % 3;                                // xe >= 1 here
  const double factor = rem == 0 ? 1.0 : (rem == 1 ? 1.2599210498948731647672 : 1.5874010519681994747517);
  const float ym = (float)((double)u * ((double)t2 + 2.0 * dxm) / (2.0 * (double)t2 + dxm) * factor);
  return ldexpf(ym, xe / 3);
}

// Same arithmetic without control flow, for callers that evaluate it on every lane of a wave and select afterwards
// (any x: the result is only meaningful -- and only used -- for x > 1; +inf maps to +inf as in glibc).
__device__ __forceinline__ float cbrtf_glibc_sel(float x) {
  const int xe = __builtin_amdgcn_frexp_expf(x);
  const float xm = __builtin_amdgcn_frexp_mantf(x);
  const double dxm = (double)xm;
  const float u = (float)(0.492659620528969547 + (0.697570460207922770 - 0.191502161678719066 * dxm) * dxm);
  const float t2 = u * u * u;
  const int q = (xe * 21846) >> 16;                      // xe / 3 for 0 <= xe <= 128
  const int rem = xe - 3 * q;
  const double factor = rem == 0 ? 1.0 : (rem == 1 ? 1.2599210498948731647672 : 1.5874010519681994747517);
  const float ym = (float)((double)u * ((double)t2 + 2.0 * dxm) / (2.0 * (double)t2 + dxm) * factor);
  const float r = ldexpf(ym, q);
  return __builtin_isinf(x) ? x : r;
}

// The same routine for 1 < x < 2, where frexp gives xe = 1, xm = x/2: factor[2 + 1%3] = 2^(1/3), ldexp(ym, 0) = ym.  The
// f64 steps are cheaper than glibc's and NOT all bit-identical to them -- the polynomial is contracted into two fmas, the
// sums with an exact product (2*dxm, 2*t2) are single fmas, and the division is reciprocal + one Newton step + one
// multiply, not correctly rounded -- because the final rounding to f32 absorbs their last-bit differences: equality of the
// RESULT with the host libm's cbrtf for EVERY f32 in (1,2) is checked on the device (tests/test_gpu_selftest.py), which is
// a proof since the domain is enumerable.  (Measured on the way: dropping the Newton step fails for 627 096 inputs; the
// variants with the residual correction, with two Newton steps, or with the literal polynomial all pass as well.)
// IPK_OPT_CBRT_ASM: the same steps with (a) the argument's halving folded into the constants -- xm = x/2 is exact, so the polynomial in dx = 2 xm has
// its coefficients scaled by powers of two, 2 xm is dx itself and the denominator is taken twice (2 den = 4 t2 + dx; the reciprocal, its Newton
// step and the final products scale by exact powers of two, the last constant becomes 2^(1/3)/... see below) -- one instruction fewer; and (b) the
// polynomial's two fmas as explicit v_fma_f64: hipcc turns `fma(a, x, CONST)` into a register copy of CONST plus v_fmac_f64 (two-address form),
// two extra v_mov_b64 per evaluation.  Proven like the form above (tests/test_gpu_selftest.py, every f32 in (1,2)).
__device__ __forceinline__ double fma_f64_vvv(double a, double b, double c) {
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ double fma_f64_svv(double a, double b, double c) {   // a wave-uniform (the one scalar operand VOP3 may read)
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "s"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ float cbrtf_glibc_1to2(float x) {
  // with dx = (double)x = 2 dxm:  c2 dxm^2 = (c2/4) dx^2, c1 dxm = (c1/2) dx  (exact scalings)
  const double dx = (double)x;
  const float u = (float)fma_f64_vvv(fma_f64_svv(-0.191502161678719066 * 0.25, dx, 0.697570460207922770 * 0.5), dx, 0.492659620528969547);
  const float t2 = u * u * u;
  const double du = (double)u, dt2 = (double)t2;
  const double num = du * (dt2 + dx);                      // dt2 + 2 dxm: the exact sum, as the fma form gives it
  const double den2 = __builtin_fma(4.0, dt2, dx);          // 2 (2 dt2 + dxm), exactly twice the proven form's denominator
  double r = __builtin_amdgcn_rcp(den2);                    // = r/2 exactly (the reciprocal unit works on the mantissa)
  r = __builtin_fma(__builtin_fma(-den2, r, 1.0), r, r);
  return (float)((num * r) * (2.0 * 1.2599210498948731647672));
}

// XYZ_LAB_TRANSFORM.lookup (src/color_conversions.rs:102-114 with the closure of :120-124)
template <typename Tab>
__device__ __forceinline__ float lab_lookup(const Tab *__restrict__ lab, float v) {
  if (v < 0.0f || v > 1.0f) {
    if (v > kLabE) return cbrtf_glibc(v);                // only v > 1 reaches here
    return (kLabK * v + 16.0f) / 116.0f;                 // v < 0
  }
  return lut_interp(lab, v);
}

// xyz_to_lab (src/color_conversions.rs:156-169)
template <typename Tab>
__device__ __forceinline__ void xyz_to_lab(const Tab *__restrict__ lab, float x, float y, float z,
                                           float &ol, float &oa, float &ob) {
  const float xr = x / kWhiteX, yr = y / kWhiteY, zr = z / kWhiteZ;
  const float fx = lab_lookup(lab, xr);
  const float fy = lab_lookup(lab, yr);
  const float fz = lab_lookup(lab, zr);
  const float l = 116.0f * fy - 16.0f;
  const float a = 500.0f * (fx - fy);
  const float b = 200.0f * (fy - fz);
  ol = l / 100.0f; oa = (a + 127.0f) / 255.0f; ob = (b + 127.0f) / 255.0f;
}

// lab_to_xyz (src/color_conversions.rs:172-191)
__device__ __forceinline__ void lab_to_xyz(float l, float a, float b, float &ox, float &oy, float &oz) {
  const float cl = l * 100.0f;
  const float ca = (a * 255.0f) - 127.0f;
  const float cb = (b * 255.0f) - 127.0f;
  const float fy = (cl + 16.0f) / 116.0f;
  const float fx = ca / 500.0f + fy;
  const float fz = fy - (cb / 200.0f);
  const float fx3 = fx * fx * fx;
  const float xr = (fx3 > kLabE) ? fx3 : (116.0f * fx - 16.0f) / kLabK;
  const float yr = (cl > kLabK * kLabE) ? fy * fy * fy : cl / kLabK;
  const float fz3 = fz * fz * fz;
  const float zr = (fz3 > kLabE) ? fz3 : (116.0f * fz - 16.0f) / kLabK;
  ox = xr * kWhiteX; oy = yr * kWhiteY; oz = zr * kWhiteZ;
}

// camera_to_lab (src/color_conversions.rs:42-55); cm = [[f32;4];3] row-major
struct ToLabParams { float mul[4]; float cm[12]; };
template <typename Tab>
__device__ __forceinline__ void camera_to_lab(const Tab *__restrict__ lab, const ToLabParams &p,
                                              float p0, float p1, float p2, float p3,
                                              float &ol, float &oa, float &ob) {
  const float r = rs_min(p0 * p.mul[0], 1.0f);
  const float g = rs_min(p1 * p.mul[1], 1.0f);
  const float b = rs_min(p2 * p.mul[2], 1.0f);
  const float e = rs_min(p3 * p.mul[3], 1.0f);
  const float x = r * p.cm[0] + g * p.cm[1] + b * p.cm[2] + e * p.cm[3];
  const float y = r * p.cm[4] + g * p.cm[5] + b * p.cm[6] + e * p.cm[7];
  const float z = r * p.cm[8] + g * p.cm[9] + b * p.cm[10] + e * p.cm[11];
  xyz_to_lab(lab, x, y, z, ol, oa, ob);
}

// lab_to_rgb (src/color_conversions.rs:58-65); m = XYZ_D65_33 row-major
struct Mat9 { float m[9]; };
__device__ __forceinline__ void lab_to_rgb(const Mat9 &mm, float l, float a, float b, float &orr, float &og, float &ob) {
  float x, y, z;
  lab_to_xyz(l, a, b, x, y, z);
  orr = x * mm.m[0] + y * mm.m[1] + z * mm.m[2];
  og  = x * mm.m[3] + y * mm.m[4] + z * mm.m[5];
  ob  = x * mm.m[6] + y * mm.m[7] + z * mm.m[8];
}

// Table branch of lookup() on a plain copy of the table (v2 - v1 evaluated per lookup, as the reference does);
// the two reads are adjacent (ds_read2_b32).  Used where LDS space matters more than one subtraction.
__device__ __forceinline__ float lut_interp_plain(const float *__restrict__ tab, float val) {
  const float pos = val * kLutMaxF;
  const uint32_t key = f32_as_u32_sat(pos);
  const float base = truncf(pos);
  const float a = pos - base;
  const float v1 = tab[key], v2 = tab[key + 1];
  return v1 + a * (v2 - v1);
}
__device__ __forceinline__ float gamma_sample_plain(const float *__restrict__ gam, float v) {
  return lut_interp_plain(gam, rs_min(rs_max(v, 0.0f), 1.0f));
}
__device__ __forceinline__ float gamma_sample_plain(const LutPair *__restrict__ gam, float v) {   // the same step on a pair table
  return lut_interp(gam, rs_min(rs_max(v, 0.0f), 1.0f));
}

// OpGamma's per-sample step (src/ops/gamma.rs:22): apply_srgb_gamma(v.max(0).min(1)); the clamp makes
// the out-of-table branch of lookup unreachable.
__device__ __forceinline__ float gamma_sample(const LutPair *__restrict__ gam, float v) {
  return lut_interp(gam, rs_min(rs_max(v, 0.0f), 1.0f));
}

// OpGamma's step FOLLOWED BY output8bit (src/ops/gamma.rs:22, src/color_conversions.rs:323-326), as one lookup.  Inside one segment of the 13-bit gamma table
// the interpolated value rises by at most 12.92 / 8191 = 0.0016, i.e. by 0.40 of an 8-bit step: the quantised result of every clamped sample c of
// segment i is therefore k_i or k_i + 1, and -- every f32 operation on the way being monotone in c -- it is k_i + (c >= t_i) for ONE threshold t_i
// (+inf where the segment holds no step).  The table is built ON THE DEVICE from the pair table by bisection with the literal expression
// (k_build_q8), and the identity is then checked for every f32 bit pattern (ipk_selftest_q8): 3 instructions + one 8-byte read per sample instead of 7 + one.
struct __attribute__((aligned(8))) Q8Entry { uint32_t k; float t; };
__device__ __forceinline__ uint32_t q8_sample(const Q8Entry *__restrict__ tab, float c /* already clamped to [0, 1] */) {
  const Q8Entry e = tab[f32_as_u32_sat(c * kLutMaxF)];
  return e.k + (c >= e.t ? 1u : 0u);
}

// output8bit / output16bit (src/color_conversions.rs:323-330)
__device__ __forceinline__ uint8_t output8bit_literal(float v) {
  return (uint8_t)f32_as_u32_sat(rs_min(rs_max(v * 256.0f, 0.0f), 255.0f));
}
__device__ __forceinline__ uint8_t output8bit(float v) { return output8bit_literal(v); }
// four samples into one dword: v_cvt_pk_u8_f32 converts with saturation to [0, 255] (NaN -> 0) and writes one byte lane; it rounds
// to nearest, so it is fed floor(v * 256) -- equal to output8bit on every f32 (exhaustive: ipk_selftest_quant8 variant 1)
__device__ __forceinline__ uint32_t output8bit_x4(float a, float b, float c, float d) {
  uint32_t w = __builtin_amdgcn_cvt_pk_u8_f32(floorf(a * 256.0f), 0u, 0u);
  w = __builtin_amdgcn_cvt_pk_u8_f32(floorf(b * 256.0f), 1u, w);
  w = __builtin_amdgcn_cvt_pk_u8_f32(floorf(c * 256.0f), 2u, w);
  return __builtin_amdgcn_cvt_pk_u8_f32(floorf(d * 256.0f), 3u, w);
}
__device__ __forceinline__ uint16_t output16bit_literal(float v) {
  return (uint16_t)f32_as_u32_sat(rs_min(rs_max(roundf(v * 65535.0f), 0.0f), 65535.0f));
}
// Two samples into one dword.  Rust's f32::round rounds halves away from zero, which for p = v * 65535 >= 0 is floor(p + 0.5) -- the sum is exact wherever
// it matters (p < 2^17 carries at most seven fraction bits) -- and everything the clamp does is done by the conversions: v_cvt_u32_f32 sends negatives and
// NaN to 0 and saturates, v_cvt_pk_u16_u32 saturates at 65535.  mul, add, floor, cvt + half a pack per sample instead of roundf's six and two clamps;
// equal to the literal form on every f32 (exhaustive: ipk_selftest_quant16).  (v_cvt_pknorm_u16_f32 rounds halves to even: 32 768 inputs differ.)
__device__ __forceinline__ uint32_t output16bit_x2(float a, float b) {
  typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
  const u16x2 p = __builtin_amdgcn_cvt_pk_u16(f32_as_u32_sat(floorf(a * 65535.0f + 0.5f)), f32_as_u32_sat(floorf(b * 65535.0f + 0.5f)));
  return (uint32_t)p.x | ((uint32_t)p.y << 16);
}
__device__ __forceinline__ uint16_t output16bit(float v) { return (uint16_t)(output16bit_x2(v, 0.0f) & 0xFFFFu); }
// input8bit / input16bit (src/color_conversions.rs:313-320)
__device__ __forceinline__ float input8bit(uint8_t v) { return (float)v / 255.0f; }
__device__ __forceinline__ float input16bit(uint16_t v) { return (float)v / 65535.0f; }

// ---- SplineFunc::interpolate (src/ops/curves.rs:126-157) -----------------------------------
constexpr int kSplineMaxKnots = 66;
// LDS image of a curve (fill_knots): [px | py | c1 | c2 | c3] (kSplineMaxKnots floats each), 2 floats of padding, then one 8-float
// record {x_i, y_i, c1_i, c2_i, c3_i, -, -, -} per segment of the 3-knot form, 16-byte aligned: a per-lane segment choice is ONE address
// select and two reads (b128 + b32) instead of five address selects (v_cndmask is a half-rate instruction on gfx950)
constexpr int kKnotSegRec = 5 * kSplineMaxKnots + 2;
constexpr int kKnotFloats = kKnotSegRec + 16;
struct SplineDev {
  int npoints, nseg;
  float px[kSplineMaxKnots], py[kSplineMaxKnots], c1[kSplineMaxKnots], c2[kSplineMaxKnots], c3[kSplineMaxKnots];
  // grid form of a curve with four or more knots (spline_interpolate_grid): set by the host when the curve qualifies (make_spline)
  int grid_ok; float grid_scale, grid_xl, grid_yl;       // cells per unit of x; the last knot
};
__device__ __forceinline__ float spline_poly(float y, float c1, float c2, float c3, float diff) {
  // self.points[i].1 + self.c1s[i]*diff + self.c2s[i]*diff*diff + self.c3s[i]*diff*diff*diff  (:156)
  return y + c1 * diff + c2 * diff * diff + c3 * diff * diff * diff;
}
// Literal restatement, any knot count.  `s` lives in the kernel-argument segment (uniform loads for
// the ends, per-lane loads inside the search).
__device__ __forceinline__ float spline_interpolate(const SplineDev &s, float val) {
  const int np = s.npoints;
  if (np == 2) {
    // unrolled for the 2-knot curve (no user points): search range 0..=0
    if (val >= s.px[1]) return s.py[1];
    if (!(val > s.px[0])) return s.py[0];              // val <= first, or NaN (returns points[mid=0].1)
    return spline_poly(s.py[0], s.c1[0], s.c2[0], s.c3[0], val - s.px[0]);
  }
  if (np == 3) {
    // unrolled for the 3-knot curve (the default raw base curve, curves.rs:18): search range 0..=1,
    // first probe mid=0 always moves low to 1 (val > px[0] here), second probe mid=1 decides.
    if (val >= s.px[2]) return s.py[2];
    if (!(val > s.px[0])) return s.py[0];              // val <= first, or NaN (points[mid=0].1)
    const float x1 = s.px[1];
    if (x1 < val) return spline_poly(s.py[1], s.c1[1], s.c2[1], s.c3[1], val - x1);
    if (x1 > val) return spline_poly(s.py[0], s.c1[0], s.c2[0], s.c3[0], val - s.px[0]);
    return s.py[1];
  }
  if (val >= s.px[np - 1]) return s.py[np - 1];
  if (val <= s.px[0]) return s.py[0];
  int low = 0, high = s.nseg - 1;
  while (low <= high) {
    const int mid = (low + high) / 2;
    const float xhere = s.px[mid];
    if (xhere < val) low = mid + 1;
    else if (xhere > val) high = mid - 1;
    else return s.py[mid];                               // also NaN
  }
  const int i = high > 0 ? high : 0;
  return spline_poly(s.py[i], s.c1[i], s.c2[i], s.c3[i], val - s.px[i]);
}

// The same literal search with the knot arrays staged in LDS as [px | py | c1 | c2 | c3] (kSplineMaxKnots floats each):
// per-lane indexed reads then use the LDS counter (lgkmcnt) and leave the kernel's global-memory pipeline alone.
__device__ __forceinline__ float spline_interpolate_lds(const float *__restrict__ t, int np, int nseg, float val) {
  const float *px = t, *py = t + kSplineMaxKnots, *c1 = t + 2 * kSplineMaxKnots, *c2 = t + 3 * kSplineMaxKnots, *c3 = t + 4 * kSplineMaxKnots;
  if (val >= px[np - 1]) return py[np - 1];
  if (val <= px[0]) return py[0];
  int low = 0, high = nseg - 1;
  while (low <= high) {
    const int mid = (low + high) / 2;
    const float xhere = px[mid];
    if (xhere < val) low = mid + 1;
    else if (xhere > val) high = mid - 1;
    else return py[mid];
  }
  const int i = high > 0 ? high : 0;
  return spline_poly(py[i], c1[i], c2[i], c3[i], val - px[i]);
}

// The 3-knot curve (the raw default, curves.rs:18; or 2 knots padded by the host to (x0, x1, x1) / (y0, y1, y1)) for the common-parameter kernels, with
// two of the literal search's three special cases turned into arithmetic:
//   val <= x0 or NaN -> y0:  the argument is raised to x0 first (v_max_f32 returns the number when the other operand is NaN); segment 0 at a
//     difference of exactly 0 gives y0 + c1*0 + c2*0*0 + c3*0*0*0 = y0 -- every product is a zero, and adding zeros to y0 returns y0 bit for bit
//     unless y0 is -0.0 (which would come back as +0.0);
//   val == x1 -> y1:  the knot itself goes to segment 1 (`>=` instead of `>`), whose difference is then exactly 0: y1 by the same argument;
//   val >= x2 -> y2 stays a select (segment 1 at x2 is only approximately y2).
// Needs finite coefficients (0 * inf is NaN) and knot ordinates that are not -0.0: spline3_arith_ok(), checked by the host, which otherwise
// launches the generic variants (spline_interpolate_sel).  Equality with the literal search on EVERY f32 argument is checked on the device for the curves the tests use
// (ipk_selftest_spline3, tests/test_gpu_selftest.py).
__device__ __forceinline__ float spline_interpolate_3a(const SplineDev &s, const float *__restrict__ lds_knots, float val) {
  const float x0 = s.px[0], x1 = s.px[1], x2 = s.px[2];
  const float v1 = fmaxf(val, x0);
  const bool up = v1 >= x1;
  const float *rec = lds_knots + kKnotSegRec + (up ? 8 : 0);
  const float4 q = *reinterpret_cast<const float4 *>(rec);
  const float r = spline_poly(q.y, q.z, q.w, rec[4], v1 - q.x);
  return (val >= x2) ? s.py[2] : r;
}
__device__ __forceinline__ float spline_interpolate_sel(const SplineDev &s, const float *__restrict__ lds_knots, float val) {
  const int np = s.npoints;
  if (np == 3) {
    const float x0 = s.px[0], x1 = s.px[1], x2 = s.px[2];
    const bool up = x1 < val, down = x1 > val;
    // segment coefficients by per-lane index from the LDS copy (a select between two kernel-argument loads would
    // become a per-lane global load)
    const int i = up ? 1 : 0;
    const float bx = lds_knots[i], by = lds_knots[kSplineMaxKnots + i];
    const float k1 = lds_knots[2 * kSplineMaxKnots + i], k2 = lds_knots[3 * kSplineMaxKnots + i], k3 = lds_knots[4 * kSplineMaxKnots + i];
    float r = spline_poly(by, k1, k2, k3, val - bx);
    r = (!up && !down) ? s.py[1] : r;                  // exact knot hit
    r = !(val > x0) ? s.py[0] : r;                     // val <= first, or NaN
    r = (val >= x2) ? s.py[2] : r;                     // val >= end
    return r;
  }
  if (np == 2) {
    float r = spline_poly(s.py[0], s.c1[0], s.c2[0], s.c3[0], val - s.px[0]);
    r = !(val > s.px[0]) ? s.py[0] : r;
    r = (val >= s.px[1]) ? s.py[1] : r;
    return r;
  }
  return spline_interpolate_lds(lds_knots, np, s.nseg, val);
}

// ---- curves with four or more knots: the binary search as a grid lookup --------------------------------------------------
// SplineFunc::interpolate (curves.rs:126-157) for sorted knots decides: val >= x_last -> y_last; val <= x_0 -> y_0; val == x_k -> y_k; otherwise the
// segment i with x_i < val < x_{i+1}, y_i + c1_i d + c2_i d d + c3_i d d d with d = val - x_i.  With the argument raised to x_0 first and the knot
// itself counted into its own segment (d = 0: every product is a zero and y_k comes back bit for bit, as in spline_interpolate_3a) that is: segment
// i = max{k <= nseg-1 : x_k <= v1}, evaluated at d = v1 - x_i, and the x_last select.  The segment comes from a uniform grid over [x_0, x_last]:
// cell(v) = min(u32((v - x_0) * scale), 255) is monotone in v (each f32 step is), so every knot in an earlier cell is below v1 and every knot in a
// later one above it; the host admits a curve only when no cell holds two knots (besides x_0), which leaves ONE comparison -- against the knot in v1's
// own cell.  Per pixel: one 8-byte LDS read by cell {record offset, that knot's x}, one compare, the segment record (b128 + b32), the polynomial --
// where the literal search runs two to six dependent LDS probes inside a per-lane loop.  Needs what the 3-knot form needs: finite coefficients, knot
// ordinates that are not -0.0 (spline_grid_ok); NaN arguments differ (y_0 instead of the first probe's knot) and never reach the curve in a lane
// whose result is used: the point-wise stages flag non-finite inputs and redo them literally.  Equality with the literal search on EVERY f32 is
// checked on the device per curve (ipk_selftest_spline3).
constexpr int kGridCells = 256;
constexpr int kGridFloats = 2 * kGridCells + 8 * kSplineMaxKnots;     // cells {record byte offset, knot x}, then one 8-float record per segment
__device__ __host__ inline uint32_t spline_grid_cell(float v1, float x0, float scale) {
  const float t = (v1 - x0) * scale;                                   // >= 0: v1 >= x0
#if defined(__HIP_DEVICE_COMPILE__)
  const uint32_t c = __float2uint_rz(t);                               // saturating
#else
  const uint32_t c = !(t < 4294967296.0f) ? 0xFFFFFFFFu : (uint32_t)t;
#endif
  return c < (uint32_t)(kGridCells - 1) ? c : (uint32_t)(kGridCells - 1);
}
// called by the first kGridCells threads of a block (before its barrier); reads the curve from the kernel arguments
__device__ __forceinline__ void fill_grid(float *__restrict__ g, const SplineDev &s, int t) {
  if (t >= kGridCells) return;
  const float x0 = s.px[0];
  uint32_t base = 0u; float nx = __builtin_inff();
  for (int k = 1; k < s.nseg; ++k) {
    const uint32_t ck = spline_grid_cell(s.px[k], x0, s.grid_scale);
    if (ck < (uint32_t)t) base = (uint32_t)k;
    else if (ck == (uint32_t)t) nx = s.px[k];                          // at most one (host-checked)
  }
  g[2 * t] = __uint_as_float((2u * kGridCells + 8u * base) * 4u);
  g[2 * t + 1] = nx;
  if (t < s.nseg) {
    float *r = g + 2 * kGridCells + 8 * t;
    r[0] = s.px[t]; r[1] = s.py[t]; r[2] = s.c1[t]; r[3] = s.c2[t]; r[4] = s.c3[t]; r[5] = 0.0f; r[6] = 0.0f; r[7] = 0.0f;
  }
}
__device__ __forceinline__ float spline_interpolate_grid(const SplineDev &s, const float *__restrict__ g, float val) {
  const float x0 = s.px[0];
  const float v1 = fmaxf(val, x0);
  const uint32_t c = spline_grid_cell(v1, x0, s.grid_scale);
  const float2 ce = reinterpret_cast<const float2 *>(g)[c];
  const uint32_t off = __float_as_uint(ce.x) + (ce.y <= v1 ? 32u : 0u);
  const float *rec = reinterpret_cast<const float *>(reinterpret_cast<const char *>(g) + off);
  const float4 q = *reinterpret_cast<const float4 *>(rec);
  const float r = spline_poly(q.y, q.z, q.w, rec[4], v1 - q.x);
  return (val >= s.grid_xl) ? s.grid_yl : r;
}

// the LDS image of a curve: called by the first kKnotFloats threads of a block (before its barrier)
__device__ __forceinline__ void fill_knots(float *__restrict__ lds, const SplineDev &s, int i) {
  if (i < kSplineMaxKnots) {
    lds[i] = s.px[i]; lds[kSplineMaxKnots + i] = s.py[i]; lds[2 * kSplineMaxKnots + i] = s.c1[i];
    lds[3 * kSplineMaxKnots + i] = s.c2[i]; lds[4 * kSplineMaxKnots + i] = s.c3[i];
  }
  if (i < 2) {
    float *r = lds + kKnotSegRec + 8 * i;
    r[0] = s.px[i]; r[1] = s.py[i]; r[2] = s.c1[i]; r[3] = s.c2[i]; r[4] = s.c3[i]; r[5] = 0.0f; r[6] = 0.0f; r[7] = 0.0f;
  }
}

// ---- division by a positive constant, 4 instructions instead of the ~11 of an IEEE divide ---------
// q0 = x*rc ; r = fma(-q0, c, x) ; q1 = fma(r, rc, q0) ; v_div_fixup(q1, c, x)
// With rc = RN(1/c) the residual step returns the correctly rounded quotient (Markstein) whenever no
// intermediate under/overflows; v_div_fixup_f32 restores the IEEE results for x = +-0, +-inf, NaN.
// Checked EXHAUSTIVELY (all 2^31 positive f32 x, tests/test_gpu_selftest.py on the GPU and once on the
// host) for every constant used below: the only failures are nonzero |x| < 2^-104 (all c) and
// |x| > 2^126 (c < 1 only).  Call sites either prove their dividend is outside those zones (see the
// comment at each) or raise `bad` through cdiv_guard(), which makes the wave redo the pixel group with
// true divisions (pointwise_exact), so results stay bit-identical to `x / c` for every input.
__device__ __forceinline__ float cdiv_fast(float x, float c, float rc) {
  const float q0 = x * rc;
  const float r = __builtin_fmaf(-q0, c, x);
  const float q1 = __builtin_fmaf(r, rc, q0);
  return __builtin_amdgcn_div_fixupf(q1, c, x);
}
// true when x is a finite nonzero value outside [2^-100, 2^100] (zero, inf and NaN pass: the fixup
// handles them).  v_frexp_exp_i32_f32 returns 0 for 0/inf/NaN.
__device__ __forceinline__ bool cdiv_guard(float x) {
  const int e = __builtin_amdgcn_frexp_expf(x);
  return (unsigned)(e + 99) > 199u;
}
// ---- LDS table staging ---------------------------------------------------------------------
__device__ __forceinline__ void load_lut_pairs(LutPair *__restrict__ lds, const LutPair *__restrict__ g) {
  // 8192 pairs = 4096 float4; all threads of the block cooperate, 16 B per lane per step
  const float4 *src = reinterpret_cast<const float4 *>(g);
  float4 *dst = reinterpret_cast<float4 *>(lds);
  for (int i = threadIdx.x; i < kLutPairs / 2; i += blockDim.x) dst[i] = src[i];
}

}  // namespace ipkd
