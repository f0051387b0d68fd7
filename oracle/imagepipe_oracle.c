/*
 * imagepipe_oracle.c -- CPU restatement of pedrocr/imagepipe's per-pixel raw->sRGB path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the product library
 * (imagepipe_amd/libimagepipe_amd.so) never links, calls or falls back to anything here.
 *
 * What it is: a scalar C restatement (gcc, -O2 -ffp-contract=off -fno-fast-math, IEEE
 * binary32 arithmetic on SSE2) of the reference's Rust, function by function, in the same
 * operation order, with the Rust semantics that matter spelled out (saturating float->int
 * casts, f32::min/max NaN rules, left-to-right evaluation, no FMA contraction).  Every
 * function cites the reference file:line it follows (paths relative to the reference root).
 *
 * Pinning status:
 *   - The Rust reference cannot be compiled or run here (no rustc/cargo; rawloader and
 *     multicache path dependencies absent), so there is no oracle/_ref build.
 *   - PINNED by the reference's own known-answer tests, restated in tests/test_oracle_*.py:
 *     int<->float (color_conversions.rs:338-348), gamma LUT roundtrips (:391-402), Lab
 *     roundtrips (:421-611), spline (curves.rs:165-188), the nine orientation goldens
 *     (transform.rs:168-278), identity rescale (scaling.rs:189-203), rotatecrop sizes and
 *     first samples (rotatecrop.rs:181-312), whole-pipeline RGB roundtrips
 *     (tests/roundtrip_test.rs) and size negotiation (tests/maxsize_test.rs).
 *   - PARITY UNPINNED (no reference test or fixture exercises them; hand-derived known
 *     answers only): OpGoFloat::run_raw, demosaic::full, scaled_demosaic / transform_buffer
 *     at non-identity scale, and rawloader 0.37's CFA::new/color_at/shift semantics
 *     (rawloader is an absent path dependency, Cargo.toml:25-27; its behaviour is restated
 *     from its published source: pattern string row-major over width x height, R=0 G=1 B=2
 *     E=3 (M=1, Y=3), tiled to 48x48, color_at(row,col)=pattern[row%48][col%48]).
 *   - libm dependence: the three 8193-entry tables are built with this host's cbrtf/powf
 *     exactly as TransformLookup::new does (color_conversions.rs:87-100), and out-of-table
 *     Lab lookups call cbrtf directly (color_conversions.rs:103-104,123).  Rust's f32::cbrt
 *     and f32::powf call the platform libm, so this matches what the reference would compute
 *     on this host (glibc 2.35).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------ */
/* Rust numeric-cast semantics                                                            */
/* ------------------------------------------------------------------------------------ */

/* `f as usize`: saturating, NaN -> 0 (Rust reference, "as" casts since 1.45). */
static inline size_t f32_as_usize(float f) {
  if (!(f > 0.0f)) return 0;                 /* NaN, negatives, zero */
  if (f >= 18446744073709551616.0f) return SIZE_MAX;
  return (size_t)f;
}
/* `f as isize`: saturating, NaN -> 0. */
static inline int64_t f32_as_isize(float f) {
  if (f != f) return 0;
  if (f >= 9223372036854775808.0f) return INT64_MAX;
  if (f <= -9223372036854775808.0f) return INT64_MIN;
  return (int64_t)f;
}
static inline uint8_t f32_as_u8(float f) {
  if (!(f > 0.0f)) return 0;
  if (f >= 255.0f) return 255;
  return (uint8_t)f;
}
static inline uint16_t f32_as_u16(float f) {
  if (!(f > 0.0f)) return 0;
  if (f >= 65535.0f) return 65535;
  return (uint16_t)f;
}
/* f32::min / f32::max: a NaN operand is ignored (== fminf/fmaxf). */
static inline float rs_min(float a, float b) { return fminf(a, b); }
static inline float rs_max(float a, float b) { return fmaxf(a, b); }
/* f32::is_normal */
static inline int rs_is_normal(float v) { return fpclassify(v) == FP_NORMAL; }

static void orc_set_threads(int nthreads) {
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#else
  (void)nthreads;
#endif
}
ORC_API void orc_set_num_threads(int n) { orc_set_threads(n); }
ORC_API int orc_get_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ------------------------------------------------------------------------------------ */
/* Constants  (src/color_conversions.rs:1-39)                                             */
/* ------------------------------------------------------------------------------------ */

static const float SRGB_D65_33[3][3] = {
  {0.4124564f, 0.3575761f, 0.1804375f},
  {0.2126729f, 0.7151522f, 0.0721750f},
  {0.0193339f, 0.1191920f, 0.9503041f},
};
static const float WHITE_X = 0.95047f, WHITE_Y = 1.000f, WHITE_Z = 1.08883f;

/* src/color_conversions.rs:20-39 -- cofactor inverse evaluated in f32 */
static void inverse33(const float inm[3][3], float out[3][3]) {
  float invdet = 1.0f / (
    inm[0][0] * (inm[1][1] * inm[2][2] - inm[2][1] * inm[1][2]) -
    inm[0][1] * (inm[1][0] * inm[2][2] - inm[1][2] * inm[2][0]) +
    inm[0][2] * (inm[1][0] * inm[2][1] - inm[1][1] * inm[2][0]));
  out[0][0] =  (inm[1][1]*inm[2][2] - inm[2][1]*inm[1][2]) * invdet;
  out[0][1] = -(inm[0][1]*inm[2][2] - inm[0][2]*inm[2][1]) * invdet;
  out[0][2] =  (inm[0][1]*inm[1][2] - inm[0][2]*inm[1][1]) * invdet;
  out[1][0] = -(inm[1][0]*inm[2][2] - inm[1][2]*inm[2][0]) * invdet;
  out[1][1] =  (inm[0][0]*inm[2][2] - inm[0][2]*inm[2][0]) * invdet;
  out[1][2] = -(inm[0][0]*inm[1][2] - inm[1][0]*inm[0][2]) * invdet;
  out[2][0] =  (inm[1][0]*inm[2][1] - inm[2][0]*inm[1][1]) * invdet;
  out[2][1] = -(inm[0][0]*inm[2][1] - inm[2][0]*inm[0][1]) * invdet;
  out[2][2] =  (inm[0][0]*inm[1][1] - inm[1][0]*inm[0][1]) * invdet;
}

/* out9 = SRGB_D65_33 row-major */
ORC_API void orc_const_srgb_d65_33(float *out9) { memcpy(out9, SRGB_D65_33, sizeof(SRGB_D65_33)); }
/* out9 = XYZ_D65_33 = inverse(SRGB_D65_33) (color_conversions.rs:8) */
ORC_API void orc_const_xyz_d65_33(float *out9) {
  float o[3][3]; inverse33(SRGB_D65_33, o); memcpy(out9, o, sizeof(o));
}
/* out12 = SRGB_D65_43 : [[f32;4];3] (color_conversions.rs:12-16) */
ORC_API void orc_const_srgb_d65_43(float *out12) {
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) out12[i*4+j] = SRGB_D65_33[i][j]; out12[i*4+3] = 0.0f; }
}
ORC_API void orc_inverse33(const float *in9, float *out9) {
  float a[3][3], o[3][3]; memcpy(a, in9, sizeof(a)); inverse33(a, o); memcpy(out9, o, sizeof(o));
}

/* ------------------------------------------------------------------------------------ */
/* TransformLookup  (src/color_conversions.rs:80-141)                                     */
/* ------------------------------------------------------------------------------------ */

#define LUT_BITS 13
#define LUT_MAX ((1 << LUT_BITS) - 1)        /* 8191 */
#define LUT_LEN (LUT_MAX + 2)                /* 8193 entries, i in 0..=max+1 */

enum { ORC_LUT_XYZ_LAB = 0, ORC_LUT_SRGB_GAMMA_REVERSE = 1, ORC_LUT_SRGB_GAMMA = 2 };

/* color_conversions.rs:120-124 */
static float f_lab(float v) {
  float e = 216.0f / 24389.0f;
  float k = 24389.0f / 27.0f;
  if (v > e) return cbrtf(v);
  return (k * v + 16.0f) / 116.0f;
}
/* color_conversions.rs:126-132 */
static float f_gamma_reverse(float v) {
  if (v < 0.04045f) return v / 12.92f;
  return powf((v + 0.055f) / 1.055f, 2.4f);
}
/* color_conversions.rs:134-140 */
static float f_gamma(float v) {
  if (v < 0.0031308f) return v * 12.92f;
  return 1.055f * powf(v, 1.0f / 2.4f) - 0.055f;
}
typedef float (*transform_fn)(float);
static transform_fn lut_fn[3] = { f_lab, f_gamma_reverse, f_gamma };
static float lut_table[3][LUT_LEN];
static int lut_ready = 0;

/* color_conversions.rs:87-100 */
static void luts_init(void) {
  if (lut_ready) return;
  for (int t = 0; t < 3; t++)
    for (int i = 0; i <= LUT_MAX + 1; i++) {
      float v = (float)i / (float)LUT_MAX;
      lut_table[t][i] = lut_fn[t](v);
    }
  lut_ready = 1;
}
ORC_API void orc_luts_init(void) { luts_init(); }
ORC_API const float *orc_lut_table(int which) { luts_init(); return lut_table[which]; }
ORC_API int orc_lut_len(void) { return LUT_LEN; }

/* color_conversions.rs:102-114 */
static inline float lut_lookup(int which, float val) {
  if (val < 0.0f || val > 1.0f) {
    return lut_fn[which](val);
  } else {
    float pos = val * (float)LUT_MAX;
    size_t key = f32_as_usize(pos);
    float base = truncf(pos);
    float a = pos - base;
    float v1 = lut_table[which][key];
    float v2 = lut_table[which][key + 1];
    return v1 + a * (v2 - v1);
  }
}
ORC_API void orc_lookup(int which, const float *in, float *out, size_t n) {
  luts_init();
  #pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n; i++) out[i] = lut_lookup(which, in[i]);
}
static inline float expand_srgb_gamma(float v) { return lut_lookup(ORC_LUT_SRGB_GAMMA_REVERSE, v); } /* :145-147 */
static inline float apply_srgb_gamma(float v)  { return lut_lookup(ORC_LUT_SRGB_GAMMA, v); }         /* :151-153 */

/* ------------------------------------------------------------------------------------ */
/* 8/16-bit in/out  (src/color_conversions.rs:313-330)                                    */
/* ------------------------------------------------------------------------------------ */
static inline float input8bit(uint8_t v)   { return (float)v / 255.0f; }
static inline float input16bit(uint16_t v) { return (float)v / 65535.0f; }
static inline uint8_t output8bit(float v)  { return f32_as_u8(rs_min(rs_max(v * 256.0f, 0.0f), 255.0f)); }
static inline uint16_t output16bit(float v){ return f32_as_u16(rs_min(rs_max(roundf(v * 65535.0f), 0.0f), 65535.0f)); }

ORC_API void orc_input8bit(const uint8_t *in, float *out, size_t n)   { for (size_t i = 0; i < n; i++) out[i] = input8bit(in[i]); }
ORC_API void orc_input16bit(const uint16_t *in, float *out, size_t n) { for (size_t i = 0; i < n; i++) out[i] = input16bit(in[i]); }
/* also the serial quantise loops of Pipeline::output_8bit/16bit (src/pipeline.rs:408-414, :455-461) */
ORC_API void orc_output8bit(const float *in, uint8_t *out, size_t n)  { for (size_t i = 0; i < n; i++) out[i] = output8bit(in[i]); }
ORC_API void orc_output16bit(const float *in, uint16_t *out, size_t n){ for (size_t i = 0; i < n; i++) out[i] = output16bit(in[i]); }

/* ------------------------------------------------------------------------------------ */
/* Colour maths  (src/color_conversions.rs:42-65, :156-191)                               */
/* ------------------------------------------------------------------------------------ */

/* color_conversions.rs:156-169 */
static inline void xyz_to_lab(float x, float y, float z, float *ol, float *oa, float *ob) {
  float xr = x / WHITE_X, yr = y / WHITE_Y, zr = z / WHITE_Z;
  float fx = lut_lookup(ORC_LUT_XYZ_LAB, xr);
  float fy = lut_lookup(ORC_LUT_XYZ_LAB, yr);
  float fz = lut_lookup(ORC_LUT_XYZ_LAB, zr);
  float l = 116.0f * fy - 16.0f;
  float a = 500.0f * (fx - fy);
  float b = 200.0f * (fy - fz);
  *ol = l / 100.0f; *oa = (a + 127.0f) / 255.0f; *ob = (b + 127.0f) / 255.0f;
}
/* color_conversions.rs:172-191 */
static inline void lab_to_xyz(float l, float a, float b, float *ox, float *oy, float *oz) {
  float cl = l * 100.0f;
  float ca = (a * 255.0f) - 127.0f;
  float cb = (b * 255.0f) - 127.0f;
  float fy = (cl + 16.0f) / 116.0f;
  float fx = ca / 500.0f + fy;
  float fz = fy - (cb / 200.0f);
  float e = 216.0f / 24389.0f;
  float k = 24389.0f / 27.0f;
  float fx3 = fx * fx * fx;
  float xr = (fx3 > e) ? fx3 : (116.0f * fx - 16.0f) / k;
  float yr = (cl > k * e) ? fy * fy * fy : cl / k;
  float fz3 = fz * fz * fz;
  float zr = (fz3 > e) ? fz3 : (116.0f * fz - 16.0f) / k;
  *ox = xr * WHITE_X; *oy = yr * WHITE_Y; *oz = zr * WHITE_Z;
}
/* color_conversions.rs:42-55 ; cmatrix is [[f32;4];3] row-major */
static inline void camera_to_lab(const float mul[4], const float cm[12], const float *pixin,
                                 float *ol, float *oa, float *ob) {
  float r = rs_min(pixin[0] * mul[0], 1.0f);
  float g = rs_min(pixin[1] * mul[1], 1.0f);
  float b = rs_min(pixin[2] * mul[2], 1.0f);
  float e = rs_min(pixin[3] * mul[3], 1.0f);
  float x = r * cm[0] + g * cm[1] + b * cm[2]  + e * cm[3];
  float y = r * cm[4] + g * cm[5] + b * cm[6]  + e * cm[7];
  float z = r * cm[8] + g * cm[9] + b * cm[10] + e * cm[11];
  xyz_to_lab(x, y, z, ol, oa, ob);
}
/* color_conversions.rs:58-65 ; rgbmatrix is [[f32;3];3] row-major */
static inline void lab_to_rgb(const float m[9], const float *pixin, float *orr, float *og, float *ob) {
  float x, y, z;
  lab_to_xyz(pixin[0], pixin[1], pixin[2], &x, &y, &z);
  *orr = x * m[0] + y * m[1] + z * m[2];
  *og  = x * m[3] + y * m[4] + z * m[5];
  *ob  = x * m[6] + y * m[7] + z * m[8];
}

ORC_API void orc_xyz_to_lab(const float *xyz, float *lab, size_t n) {
  luts_init();
  #pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n; i++) xyz_to_lab(xyz[3*i], xyz[3*i+1], xyz[3*i+2], &lab[3*i], &lab[3*i+1], &lab[3*i+2]);
}
ORC_API void orc_lab_to_xyz(const float *lab, float *xyz, size_t n) {
  #pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n; i++) lab_to_xyz(lab[3*i], lab[3*i+1], lab[3*i+2], &xyz[3*i], &xyz[3*i+1], &xyz[3*i+2]);
}
ORC_API void orc_camera_to_lab(const float *mul4, const float *cm12, const float *in4, float *out3, size_t n) {
  luts_init();
  #pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n; i++) camera_to_lab(mul4, cm12, in4 + 4*i, &out3[3*i], &out3[3*i+1], &out3[3*i+2]);
}
ORC_API void orc_lab_to_rgb(const float *m9, const float *in3, float *out3, size_t n) {
  #pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n; i++) lab_to_rgb(m9, in3 + 3*i, &out3[3*i], &out3[3*i+1], &out3[3*i+2]);
}

/* ------------------------------------------------------------------------------------ */
/* rawloader 0.37 CFA (absent path dependency; restated, parity unpinned)                  */
/* call sites: src/ops/demosaic.rs:32-33,80,86 ; src/scaling.rs:110                        */
/* ------------------------------------------------------------------------------------ */
typedef struct { int width, height; int pattern[48][48]; } orc_cfa;

/* CFA::new(patname): row-major letters of one tile, tiled to 48x48.  4 -> 2x2, 36 -> 6x6, 144 -> 12x12; any other shape -- 16 letters
 * in particular, whose shape in rawloader (8 wide x 2 high per demosaic.rs:36-37, or dcraw's 2 x 8) cannot be verified here -- only with
 * the caller's statement of it as a prefix "WxH:" (W, H dividing 48), the same contract as the product's.  Returns 0 or -1. */
static int cfa_new(const char *pat, orc_cfa *c) {
  memset(c, 0, sizeof(*c));
  int w = 0, h = 0;
  const char *letters = pat, *colon = strchr(pat, ':');
  if (colon) {
    int v[2] = {0, 0}, k = 0, digits = 0;
    for (const char *q = pat; q < colon; q++) {
      if (*q >= '0' && *q <= '9') { v[k] = v[k] * 10 + (*q - '0'); if (++digits > 2) return -1; }
      else if (*q == 'x' && k == 0 && digits > 0) { k = 1; digits = 0; }
      else return -1;
    }
    if (k != 1 || digits == 0) return -1;
    w = v[0]; h = v[1]; letters = colon + 1;
    if (w < 1 || h < 1 || w > 48 || h > 48 || 48 % w || 48 % h) return -1;
  }
  size_t len = strlen(letters);
  if (w == 0) {
    switch (len) {
      case 0: c->width = 0; c->height = 0; return 0;
      case 4: c->width = 2; c->height = 2; break;
      case 36: c->width = 6; c->height = 6; break;
      case 16: return -1;                       /* shape not stated: refused like the product */
      case 144: c->width = 12; c->height = 12; break;
      default: return -1;
    }
  } else {
    if ((size_t)w * (size_t)h != len) return -1;
    c->width = w; c->height = h;
  }
  for (size_t i = 0; i < len; i++) {
    int v;
    switch (letters[i]) {
      case 'R': v = 0; break; case 'G': v = 1; break; case 'B': v = 2; break; case 'E': v = 3; break;
      case 'M': v = 1; break; case 'Y': v = 3; break;
      default: return -1;
    }
    c->pattern[i / c->width][i % c->width] = v;
  }
  for (int row = 0; row < 48; row++)
    for (int col = 0; col < 48; col++)
      c->pattern[row][col] = c->pattern[row % c->height][col % c->width];
  return 0;
}
static inline int cfa_color_at(const orc_cfa *c, size_t row, size_t col) {
  return c->pattern[(row + 48) % 48][(col + 48) % 48];
}
/* RawImage::cropped_cfa() == cfa.shift(crop_left, crop_top) (call site demosaic.rs:13):
 * new pattern[row][col] = old color_at(row + y, col + x). Writes the shifted name string. */
ORC_API int orc_cfa_shift(const char *pat, int x, int y, char *out) {
  orc_cfa c; if (cfa_new(pat, &c)) return -1;
  static const char names[4] = {'R', 'G', 'B', 'E'};
  int n = 0;
  if (!(c.width == c.height && (c.width == 2 || c.width == 6 || c.width == 12))) n = sprintf(out, "%dx%d:", c.width, c.height);
  for (int row = 0; row < c.height; row++)
    for (int col = 0; col < c.width; col++)
      out[n++] = names[cfa_color_at(&c, (size_t)(row + y), (size_t)(col + x))];
  out[n] = 0;
  return 0;
}
/* fills pattern48[48*48] (row-major ints) ; returns cfa.width or -1 */
ORC_API int orc_cfa_pattern(const char *pat, int *pattern48) {
  orc_cfa c; if (cfa_new(pat, &c)) return -1;
  for (int r = 0; r < 48; r++) for (int q = 0; q < 48; q++) pattern48[r*48+q] = c.pattern[r][q];
  return c.width;
}

/* ------------------------------------------------------------------------------------ */
/* OpGoFloat  (src/ops/gofloat.rs:74-202)                                                 */
/* ------------------------------------------------------------------------------------ */

/* gofloat.rs:74-82 ; owidth/oheight < 10 underflow in the reference (panic) -> return -1 */
ORC_API int orc_size_image(size_t crop_top, size_t crop_right, size_t crop_bottom, size_t crop_left,
                           size_t owidth, size_t oheight, size_t *out4) {
  if (owidth < 10 || oheight < 10) return -1;
  size_t x = crop_left < owidth - 10 ? crop_left : owidth - 10;
  size_t y = crop_top < oheight - 10 ? crop_top : oheight - 10;
  size_t cw = crop_left + crop_right, ch = crop_top + crop_bottom;
  size_t width = owidth - (cw < owidth - 10 ? cw : owidth - 10);
  size_t height = oheight - (ch < oheight - 10 ? ch : oheight - 10);
  out4[0] = x; out4[1] = y; out4[2] = width; out4[3] = height;
  return 0;
}

/* gofloat.rs:122-130 (Integer data, CFA/else branch): 1-channel output, levels index 0 only */
ORC_API void orc_gofloat_cfa_u16(const uint16_t *data, size_t owidth, size_t x, size_t y,
                                 size_t width, size_t height, float black0, float white0, float *out) {
  float min0 = black0, range0 = white0 - black0;           /* gofloat.rs:86-89 */
  #pragma omp parallel for schedule(dynamic, 1)
  for (size_t row = 0; row < height; row++) {
    const uint16_t *in = data + owidth * (row + y) + x;
    float *o = out + row * width;
    for (size_t c = 0; c < width; c++) o[c] = rs_min(((float)in[c] - min0) / range0, 1.0f);
  }
}
/* gofloat.rs:158-166 (Float data) */
ORC_API void orc_gofloat_cfa_f32(const float *data, size_t owidth, size_t x, size_t y,
                                 size_t width, size_t height, float black0, float white0, float *out) {
  float min0 = black0, range0 = white0 - black0;
  #pragma omp parallel for schedule(dynamic, 1)
  for (size_t row = 0; row < height; row++) {
    const float *in = data + owidth * (row + y) + x;
    float *o = out + row * width;
    for (size_t c = 0; c < width; c++) o[c] = rs_min((in[c] - min0) / range0, 1.0f);
  }
}
/* gofloat.rs:95-108 / :133-144 monochrome (cpp==1 && !is_cfa): 4-channel, monochrome=true */
ORC_API void orc_gofloat_mono_u16(const uint16_t *data, size_t owidth, size_t x, size_t y,
                                  size_t width, size_t height, float black0, float white0, float *out4) {
  float min0 = black0, range0 = white0 - black0;
  #pragma omp parallel for schedule(dynamic, 1)
  for (size_t row = 0; row < height; row++) {
    const uint16_t *in = data + owidth * (row + y) + x;
    float *o = out4 + row * width * 4;
    for (size_t c = 0; c < width; c++) {
      float val = rs_min(((float)in[c] - min0) / range0, 1.0f);
      o[4*c] = val; o[4*c+1] = val; o[4*c+2] = val; o[4*c+3] = 0.0f;
    }
  }
}
ORC_API void orc_gofloat_mono_f32(const float *data, size_t owidth, size_t x, size_t y,
                                  size_t width, size_t height, float black0, float white0, float *out4) {
  float min0 = black0, range0 = white0 - black0;
  #pragma omp parallel for schedule(dynamic, 1)
  for (size_t row = 0; row < height; row++) {
    const float *in = data + owidth * (row + y) + x;
    float *o = out4 + row * width * 4;
    for (size_t c = 0; c < width; c++) {
      float val = rs_min((in[c] - min0) / range0, 1.0f);
      o[4*c] = val; o[4*c+1] = val; o[4*c+2] = val; o[4*c+3] = 0.0f;
    }
  }
}
/* gofloat.rs:109-120 / :145-156 cpp==3: per-channel levels, 4-channel output */
ORC_API void orc_gofloat_rgb_u16(const uint16_t *data, size_t owidth, size_t x, size_t y,
                                 size_t width, size_t height, const float *black4, const float *white4, float *out4) {
  float mins[4], ranges[4];
  for (int i = 0; i < 4; i++) { mins[i] = black4[i]; ranges[i] = white4[i] - black4[i]; }
  #pragma omp parallel for schedule(dynamic, 1)
  for (size_t row = 0; row < height; row++) {
    const uint16_t *in = data + (owidth * (row + y) + x) * 3;
    float *o = out4 + row * width * 4;
    for (size_t c = 0; c < width; c++) {
      o[4*c]   = rs_min(((float)in[3*c]   - mins[0]) / ranges[0], 1.0f);
      o[4*c+1] = rs_min(((float)in[3*c+1] - mins[1]) / ranges[1], 1.0f);
      o[4*c+2] = rs_min(((float)in[3*c+2] - mins[2]) / ranges[2], 1.0f);
      o[4*c+3] = 0.0f;
    }
  }
}
ORC_API void orc_gofloat_rgb_f32(const float *data, size_t owidth, size_t x, size_t y,
                                 size_t width, size_t height, const float *black4, const float *white4, float *out4) {
  float mins[4], ranges[4];
  for (int i = 0; i < 4; i++) { mins[i] = black4[i]; ranges[i] = white4[i] - black4[i]; }
  #pragma omp parallel for schedule(dynamic, 1)
  for (size_t row = 0; row < height; row++) {
    const float *in = data + (owidth * (row + y) + x) * 3;
    float *o = out4 + row * width * 4;
    for (size_t c = 0; c < width; c++) {
      o[4*c]   = rs_min((in[3*c]   - mins[0]) / ranges[0], 1.0f);
      o[4*c+1] = rs_min((in[3*c+1] - mins[1]) / ranges[1], 1.0f);
      o[4*c+2] = rs_min((in[3*c+2] - mins[2]) / ranges[2], 1.0f);
      o[4*c+3] = 0.0f;
    }
  }
}
/* gofloat.rs:171-201 run_other: RGB8 through the sRGB-expand LUT, RGB16 / 65535 */
ORC_API void orc_gofloat_other_u8(const uint8_t *data, size_t owidth, size_t x, size_t y,
                                  size_t width, size_t height, float *out4) {
  luts_init();
  #pragma omp parallel for schedule(dynamic, 1)
  for (size_t row = 0; row < height; row++) {
    const uint8_t *in = data + (owidth * (row + y) + x) * 3;
    float *o = out4 + row * width * 4;
    for (size_t c = 0; c < width; c++) {
      o[4*c]   = expand_srgb_gamma(input8bit(in[3*c]));
      o[4*c+1] = expand_srgb_gamma(input8bit(in[3*c+1]));
      o[4*c+2] = expand_srgb_gamma(input8bit(in[3*c+2]));
      o[4*c+3] = 0.0f;
    }
  }
}
ORC_API void orc_gofloat_other_u16(const uint16_t *data, size_t owidth, size_t x, size_t y,
                                   size_t width, size_t height, float *out4) {
  #pragma omp parallel for schedule(dynamic, 1)
  for (size_t row = 0; row < height; row++) {
    const uint16_t *in = data + (owidth * (row + y) + x) * 3;
    float *o = out4 + row * width * 4;
    for (size_t c = 0; c < width; c++) {
      o[4*c]   = input16bit(in[3*c]);
      o[4*c+1] = input16bit(in[3*c+1]);
      o[4*c+2] = input16bit(in[3*c+2]);
      o[4*c+3] = 0.0f;
    }
  }
}

/* ------------------------------------------------------------------------------------ */
/* demosaic::full  (src/ops/demosaic.rs:67-119)                                           */
/* ------------------------------------------------------------------------------------ */
ORC_API int orc_demosaic_full(const char *cfa_pat, const float *src, size_t width, size_t height, float *out4) {
  orc_cfa cfa; if (cfa_new(cfa_pat, &cfa) || cfa.width == 0) return -1;
  static const int offsets3x3[9][2] = {       /* (dy, dx)  demosaic.rs:70-74 */
    {-1,-1}, {-1, 0}, {-1, 1},
    { 0,-1}, { 0, 0}, { 0, 1},
    { 1,-1}, { 1, 0}, { 1, 1},
  };
  /* demosaic.rs:77-90 */
  static int lookups[48][48][9];
  for (size_t row = 0; row < 48; row++)
    for (size_t col = 0; col < 48; col++) {
      int pixcolor = cfa_color_at(&cfa, row, col);
      for (int i = 0; i < 9; i++) {
        int dy = offsets3x3[i][0], dx = offsets3x3[i][1];
        size_t r2 = (size_t)(48 + dy) + row;
        size_t c2 = (size_t)(48 + dx) + col;
        int ocolor = cfa_color_at(&cfa, r2, c2);
        lookups[row][col][i] = (ocolor != pixcolor || (dx == 0 && dy == 0)) ? ocolor : 4;
      }
    }
  /* OpBuffer::new zero-fills (buffer.rs:31) */
  memset(out4, 0, width * height * 4 * sizeof(float));
  /* demosaic.rs:93-116 */
  #pragma omp parallel for schedule(dynamic, 1)
  for (size_t row = 0; row < height; row++) {
    float *line = out4 + row * width * 4;
    for (size_t col = 0; col < width; col++) {
      float *pix = line + col * 4;
      const int *colors = lookups[row % 48][col % 48];
      float sums[5] = {0, 0, 0, 0, 0};
      float counts[5] = {0, 0, 0, 0, 0};
      for (int i = 0; i < 9; i++) {
        int64_t r = (int64_t)row + offsets3x3[i][0];
        int64_t c = (int64_t)col + offsets3x3[i][1];
        if (r >= 0 && r < (int64_t)height && c >= 0 && c < (int64_t)width) {
          sums[colors[i]] += src[(size_t)r * width + (size_t)c];
          counts[colors[i]] += 1.0f;
        }
      }
      for (int c = 0; c < 4; c++)
        if (counts[c] > 0.0f) pix[c] = sums[c] / counts[c];
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------------------ */
/* scaling  (src/scaling.rs:8-160)                                                        */
/* ------------------------------------------------------------------------------------ */

/* scaling.rs:8-23 */
ORC_API void orc_calculate_scaling_total(size_t width, size_t height, size_t maxwidth, size_t maxheight,
                                         float *scale, size_t *nwidth, size_t *nheight) {
  if (maxwidth == 0 && maxheight == 0) { *scale = 1.0f; *nwidth = width; *nheight = height; return; }
  float xscale = (maxwidth == 0) ? 1.0f : (float)width / (float)maxwidth;
  float yscale = (maxheight == 0) ? 1.0f : (float)height / (float)maxheight;
  if (yscale <= 1.0f && xscale <= 1.0f) { *scale = 1.0f; *nwidth = width; *nheight = height; }
  else if (yscale > xscale) { *scale = yscale; *nwidth = f32_as_usize((float)width / yscale); *nheight = maxheight; }
  else { *scale = xscale; *nwidth = maxwidth; *nheight = f32_as_usize((float)height / xscale); }
}

/* scaling.rs:51-130, generic over T via macro.  tl/tr/bl are (x,y) isize pairs.
 * cfa_pat NULL => None.  The f32->T cast is Rust's saturating `as`. */
#define DEFINE_TRANSFORM_BUFFER(NAME, T, TO_F32, FROM_F32)                                          \
ORC_API int NAME(const T *src, size_t width, size_t height,                                         \
                 int64_t tlx, int64_t tly, int64_t trx, int64_t try_, int64_t blx, int64_t bly,     \
                 size_t nwidth, size_t nheight, size_t components, const char *cfa_pat, T *out) {    \
  orc_cfa cfa_s; const orc_cfa *cfa = NULL;                                                         \
  if (cfa_pat) { if (cfa_new(cfa_pat, &cfa_s)) return -1; cfa = &cfa_s; }                           \
  if (components > 4 || nwidth == 0 || nheight == 0) return -1;                                     \
  for (size_t i = 0; i < nwidth * nheight * components; i++) out[i] = FROM_F32(0.0f);               \
  float skip_x_x = ((float)trx - (float)tlx) / (float)(nwidth - 1);                                 \
  float skip_x_y = ((float)try_ - (float)tly) / (float)(nwidth - 1);                                \
  float skip_y_x = ((float)blx - (float)tlx) / (float)(nheight - 1);                                \
  float skip_y_y = ((float)bly - (float)tly) / (float)(nheight - 1);                                \
  _Pragma("omp parallel for schedule(dynamic, 1)")                                                  \
  for (size_t row = 0; row < nheight; row++) {                                                      \
    T *line = out + row * nwidth * components;                                                      \
    float from_x_r = (float)tlx + skip_y_x * (float)row;                                            \
    float to_x_r = (float)tlx + skip_y_x * (float)(row + 1);                                        \
    float from_y_r = (float)tly + skip_y_y * (float)row;                                            \
    float to_y_r = (float)tly + skip_y_y * (float)(row + 1);                                        \
    float center_x_r = ((float)tlx) + (skip_y_x * (float)row) + (skip_y_x / 2.0f) - 0.5f;           \
    float center_y_r = ((float)tly) + (skip_y_y * (float)row) + (skip_y_y / 2.0f) - 0.5f;           \
    for (size_t col = 0; col < nwidth; col++) {                                                     \
      size_t from_x = f32_as_usize(floorf(from_x_r + (skip_x_x * (float)col)));                     \
      if (from_x > width - 1) from_x = width - 1;                                                   \
      size_t to_x = f32_as_usize(floorf(to_x_r + (skip_x_x * (float)(col + 1))));                   \
      if (to_x > width - 1) to_x = width - 1;                                                       \
      size_t from_y = f32_as_usize(floorf(from_y_r + (skip_x_y * (float)col)));                     \
      if (from_y > height - 1) from_y = height - 1;                                                 \
      size_t to_y = f32_as_usize(floorf(to_y_r + (skip_x_y * (float)(col + 1))));                   \
      if (to_y > height - 1) to_y = height - 1;                                                     \
      float center_x = center_x_r + (skip_x_x * (float)col) + (skip_x_x / 2.0f);                    \
      float center_y = center_y_r + (skip_x_y * (float)col) + (skip_x_y / 2.0f);                    \
      float sums[4] = {0, 0, 0, 0};                                                                 \
      float counts[4] = {0, 0, 0, 0};                                                               \
      for (size_t y = from_y; y <= to_y; y++) {                                                     \
        for (size_t x = from_x; x <= to_x; x++) {                                                   \
          float delta_x = ((float)x - center_x) / skip_x_x;                                         \
          float delta_y = ((float)y - center_y) / skip_y_y;                                         \
          float factor = 1.0f - (delta_x * delta_x) - (delta_y * delta_y);                          \
          factor = (factor < 0.0f) ? 0.0f : factor;                                                 \
          if (cfa) {                                                                                \
            int c = cfa_color_at(cfa, y, x);                                                        \
            sums[c] += TO_F32(src[y * width + x]) * factor;                                         \
            counts[c] += factor;                                                                    \
          } else {                                                                                  \
            for (size_t c = 0; c < components; c++) {                                               \
              sums[c] += TO_F32(src[(y * width + x) * components + c]) * factor;                    \
              counts[c] += factor;                                                                  \
            }                                                                                       \
          }                                                                                         \
        }                                                                                           \
      }                                                                                             \
      for (size_t c = 0; c < components; c++)                                                       \
        if (counts[c] > 0.0f) line[col * components + c] = FROM_F32(sums[c] / counts[c]);           \
    }                                                                                               \
  }                                                                                                 \
  return 0;                                                                                         \
}
#define ID_F32(x) (x)
#define U_TO_F32(x) ((float)(x))
DEFINE_TRANSFORM_BUFFER(orc_transform_buffer_f32, float, ID_F32, ID_F32)
DEFINE_TRANSFORM_BUFFER(orc_transform_buffer_u8, uint8_t, U_TO_F32, f32_as_u8)
DEFINE_TRANSFORM_BUFFER(orc_transform_buffer_u16, uint16_t, U_TO_F32, f32_as_u16)

/* scaling.rs:35-48 scale_down_buffer: corners (0,0), (w-1,0), (0,h-1) */
/* scaling.rs:132-145 */
ORC_API int orc_scaled_demosaic(const char *cfa_pat, const float *src, size_t width, size_t height,
                                size_t nwidth, size_t nheight, float *out4) {
  return orc_transform_buffer_f32(src, width, height, 0, 0, (int64_t)width - 1, 0, 0, (int64_t)height - 1,
                                  nwidth, nheight, 4, cfa_pat, out4);
}
/* scaling.rs:147-160 */
ORC_API int orc_scale_down_opbuf(const float *src4, size_t width, size_t height,
                                 size_t nwidth, size_t nheight, float *out4) {
  return orc_transform_buffer_f32(src4, width, height, 0, 0, (int64_t)width - 1, 0, 0, (int64_t)height - 1,
                                  nwidth, nheight, 4, NULL, out4);
}
/* scaling.rs:162-182 */
ORC_API int orc_scale_down_srgb(const uint8_t *src, size_t width, size_t height, size_t nwidth, size_t nheight, uint8_t *out) {
  return orc_transform_buffer_u8(src, width, height, 0, 0, (int64_t)width - 1, 0, 0, (int64_t)height - 1, nwidth, nheight, 3, NULL, out);
}
ORC_API int orc_scale_down_srgb16(const uint16_t *src, size_t width, size_t height, size_t nwidth, size_t nheight, uint16_t *out) {
  return orc_transform_buffer_u16(src, width, height, 0, 0, (int64_t)width - 1, 0, 0, (int64_t)height - 1, nwidth, nheight, 3, NULL, out);
}

/* demosaic.rs:33-39 */
static float demosaic_minscale(int cfa_width) {
  switch (cfa_width) { case 2: return 2.0f; case 6: return 3.0f; case 8: return 2.0f; case 12: return 12.0f; default: return 2.0f; }
}
/* OpDemosaic::run dispatch (demosaic.rs:27-61).  in: buf (colors 1 or 4).  out4 sized nw*nh*4 where
 * (nw,nh) = (demosaic_width, demosaic_height) if scaling happens else (width,height).
 * Returns: 0 pass-through (out4 untouched, caller keeps input), 1 scale_down_opbuf,
 * 2 scaled_demosaic, 3 full, 4 full+scale_down_opbuf ; -1 error.  *ow,*oh = output size. */
ORC_API int orc_demosaic_run(const char *cfa_pat, const float *buf, size_t width, size_t height, size_t colors,
                             size_t nwidth, size_t nheight, float *out4, size_t *ow, size_t *oh) {
  float scale; size_t sw, sh;
  orc_calculate_scaling_total(width, height, nwidth, nheight, &scale, &sw, &sh);
  orc_cfa cfa; if (cfa_new(cfa_pat, &cfa)) return -1;
  float minscale = demosaic_minscale(cfa.width);
  if (scale <= 1.0f && colors == 4) { *ow = width; *oh = height; return 0; }
  else if (colors == 4) { *ow = nwidth; *oh = nheight; return orc_scale_down_opbuf(buf, width, height, nwidth, nheight, out4) ? -1 : 1; }
  else if (scale >= minscale) { *ow = nwidth; *oh = nheight; return orc_scaled_demosaic(cfa_pat, buf, width, height, nwidth, nheight, out4) ? -1 : 2; }
  else {
    if (scale > 1.0f) {
      float *full = (float *)malloc(width * height * 4 * sizeof(float));
      if (!full) return -1;
      if (orc_demosaic_full(cfa_pat, buf, width, height, full)) { free(full); return -1; }
      int rc = orc_scale_down_opbuf(full, width, height, nwidth, nheight, out4);
      free(full);
      *ow = nwidth; *oh = nheight;
      return rc ? -1 : 4;
    } else {
      *ow = width; *oh = height;
      return orc_demosaic_full(cfa_pat, buf, width, height, out4) ? -1 : 3;
    }
  }
}

/* ------------------------------------------------------------------------------------ */
/* OpToLab / OpFromLab  (src/ops/colorspaces.rs:12-27, :87-137)                           */
/* ------------------------------------------------------------------------------------ */

/* colorspaces.rs:12-27 */
ORC_API void orc_normalize_wbs(const float *vals, float *out) {
  float unity = vals[1];
  for (int i = 0; i < 4; i++) out[i] = (!rs_is_normal(vals[i])) ? 1.0f : vals[i] / unity;
}
/* colorspaces.rs:89-112 ; cam_to_xyz_normalized is [[f32;4];3] row-major */
ORC_API void orc_tolab(const float *src4, size_t width, size_t height, int monochrome,
                       const float *wb_coeffs, const float *cam_to_xyz_normalized, float *out3) {
  luts_init();
  float cm[12], mul[4];
  if (monochrome) { orc_const_srgb_d65_43(cm); mul[0] = mul[1] = mul[2] = mul[3] = 1.0f; }
  else { memcpy(cm, cam_to_xyz_normalized, sizeof(cm)); orc_normalize_wbs(wb_coeffs, mul); }
  #pragma omp parallel for schedule(dynamic, 1)
  for (size_t row = 0; row < height; row++) {
    const float *inb = src4 + row * width * 4;
    float *outb = out3 + row * width * 3;
    for (size_t c = 0; c < width; c++)
      camera_to_lab(mul, cm, inb + 4*c, &outb[3*c], &outb[3*c+1], &outb[3*c+2]);
  }
}
/* colorspaces.rs:127-137 ; mutate_lines_copying clones first (buffer.rs:42-50) */
ORC_API void orc_fromlab(const float *src3, size_t width, size_t height, float *out3) {
  float m[9]; orc_const_xyz_d65_33(m);
  memcpy(out3, src3, width * height * 3 * sizeof(float));
  #pragma omp parallel for schedule(dynamic, 1)
  for (size_t row = 0; row < height; row++) {
    float *line = out3 + row * width * 3;
    for (size_t c = 0; c < width; c++) {
      float r, g, b;
      lab_to_rgb(m, line + 3*c, &r, &g, &b);
      line[3*c] = r; line[3*c+1] = g; line[3*c+2] = b;
    }
  }
}


/* ------------------------------------------------------------------------------------ */
/* White-balance temperature helpers  (src/color_conversions.rs:277-310,                 */
/* src/ops/colorspaces.rs:59-85) -- host-side, once per edit, not per pixel              */
/* ------------------------------------------------------------------------------------ */
/* CIE 1931 2-degree standard observer colour-matching functions, 380..780 nm in 5 nm steps (public CIE data; the
   reference tabulates the same values at src/color_conversions.rs:193-275) */
static const double CIE_XBAR[81] = {
  0.001368, 0.002236, 0.004243, 0.007650, 0.014310, 0.023190, 0.043510, 0.077630, 0.134380,
  0.214770, 0.283900, 0.328500, 0.348280, 0.348060, 0.336200, 0.318700, 0.290800, 0.251100,
  0.195360, 0.142100, 0.095640, 0.057950, 0.032010, 0.014700, 0.004900, 0.002400, 0.009300,
  0.029100, 0.063270, 0.109600, 0.165500, 0.225750, 0.290400, 0.359700, 0.433450, 0.512050,
  0.594500, 0.678400, 0.762100, 0.842500, 0.916300, 0.978600, 1.026300, 1.056700, 1.062200,
  1.045600, 1.002600, 0.938400, 0.854450, 0.751400, 0.642400, 0.541900, 0.447900, 0.360800,
  0.283500, 0.218700, 0.164900, 0.121200, 0.087400, 0.063600, 0.046770, 0.032900, 0.022700,
  0.015840, 0.011359, 0.008111, 0.005790, 0.004109, 0.002899, 0.002049, 0.001440, 0.001000,
  0.000690, 0.000476, 0.000332, 0.000235, 0.000166, 0.000117, 0.000083, 0.000059, 0.000042,
};
static const double CIE_YBAR[81] = {
  0.000039, 0.000064, 0.000120, 0.000217, 0.000396, 0.000640, 0.001210, 0.002180, 0.004000,
  0.007300, 0.011600, 0.016840, 0.023000, 0.029800, 0.038000, 0.048000, 0.060000, 0.073900,
  0.090980, 0.112600, 0.139020, 0.169300, 0.208020, 0.258600, 0.323000, 0.407300, 0.503000,
  0.608200, 0.710000, 0.793200, 0.862000, 0.914850, 0.954000, 0.980300, 0.994950, 1.000000,
  0.995000, 0.978600, 0.952000, 0.915400, 0.870000, 0.816300, 0.757000, 0.694900, 0.631000,
  0.566800, 0.503000, 0.441200, 0.381000, 0.321000, 0.265000, 0.217000, 0.175000, 0.138200,
  0.107000, 0.081600, 0.061000, 0.044580, 0.032000, 0.023200, 0.017000, 0.011920, 0.008210,
  0.005723, 0.004102, 0.002929, 0.002091, 0.001484, 0.001047, 0.000740, 0.000520, 0.000361,
  0.000249, 0.000172, 0.000120, 0.000085, 0.000060, 0.000042, 0.000030, 0.000021, 0.000015,
};
static const double CIE_ZBAR[81] = {
  0.006450, 0.010550, 0.020050, 0.036210, 0.067850, 0.110200, 0.207400, 0.371300, 0.645600,
  1.039050, 1.385600, 1.622960, 1.747060, 1.782600, 1.772110, 1.744100, 1.669200, 1.528100,
  1.287640, 1.041900, 0.812950, 0.616200, 0.465180, 0.353300, 0.272000, 0.212300, 0.158200,
  0.111700, 0.078250, 0.057250, 0.042160, 0.029840, 0.020300, 0.013400, 0.008750, 0.005750,
  0.003900, 0.002750, 0.002100, 0.001800, 0.001650, 0.001400, 0.001100, 0.001000, 0.000800,
  0.000600, 0.000340, 0.000240, 0.000190, 0.000100, 0.000050, 0.000030, 0.000020, 0.000010,
  0.000000, 0.000000, 0.000000, 0.000000, 0.000000, 0.000000, 0.000000, 0.000000, 0.000000,
  0.000000, 0.000000, 0.000000, 0.000000, 0.000000, 0.000000, 0.000000, 0.000000, 0.000000,
  0.000000, 0.000000, 0.000000, 0.000000, 0.000000, 0.000000, 0.000000, 0.000000, 0.000000,
};

/* f64::powi(5) == compiler-rt/libgcc __powidf2: square-and-multiply */
static double powi5(double a) { double r = 1.0; int b = 5; for (;;) { if (b & 1) r *= a; b /= 2; if (b == 0) break; a *= a; } return r; }

/* color_conversions.rs:277-293 */
ORC_API void orc_temp_to_xyz(float temp, float *out3) {
  const double C1 = 3.7417717905326694e-16, C2 = 0.014387773457709927;
  double xyz[3] = {0.0, 0.0, 0.0};
  for (int i = 0; i < 81; i++) {
    double wavelength = (double)(380 + 5 * i) / 1.0e9;
    double power = C1 / (powi5(wavelength) * (exp(C2 / ((double)temp * wavelength)) - 1.0));
    xyz[0] += power * CIE_XBAR[i]; xyz[1] += power * CIE_YBAR[i]; xyz[2] += power * CIE_ZBAR[i];
  }
  double mx = fmax(fmax(xyz[0], xyz[1]), xyz[2]);
  out3[0] = (float)(xyz[0] / mx); out3[1] = (float)(xyz[1] / mx); out3[2] = (float)(xyz[2] / mx);
}
/* color_conversions.rs:295-310 */
ORC_API void orc_xyz_to_temp(const float *xyz, float *temp_tint) {
  float mn = 1000.0f, mx = 40000.0f, temp = 0.0f;
  float n[3] = {0.0f, 0.0f, 0.0f};
  while ((mx - mn) > 1.0f) {
    temp = (mx + mn) / 2.0f;
    orc_temp_to_xyz(temp, n);
    if ((n[2] / n[0]) > (xyz[2] / xyz[0])) mx = temp; else mn = temp;
  }
  temp_tint[0] = temp; temp_tint[1] = (n[1] / n[0]) / (xyz[1] / xyz[0]);
}
/* OpToLab::set_temp (colorspaces.rs:59-70); xyz_to_cam is [[f32;3];4] row-major; writes wb_coeffs[4] */
ORC_API void orc_tolab_set_temp(const float *xyz_to_cam12, float temp, float tint, float *wb4) {
  float t[3]; orc_temp_to_xyz(temp, t);
  float xyz[3] = {t[0], t[1] / tint, t[2]};
  float w[4];
  for (int i = 0; i < 4; i++) {
    w[i] = 0.0f;
    for (int j = 0; j < 3; j++) w[i] += xyz_to_cam12[i * 3 + j] * xyz[j];
    w[i] = 1.0f / w[i];
  }
  orc_normalize_wbs(w, wb4);
}
/* OpToLab::get_temp (colorspaces.rs:72-84); cam_to_xyz is [[f32;4];3] row-major */
ORC_API void orc_tolab_get_temp(const float *cam_to_xyz12, const float *wb4, float *temp_tint) {
  float xyz[3] = {0.0f, 0.0f, 0.0f};
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 4; j++) { float mul = wb4[j]; if (mul > 0.0f) xyz[i] += cam_to_xyz12[i * 4 + j] / mul; }
  orc_xyz_to_temp(xyz, temp_tint);
}

/* ------------------------------------------------------------------------------------ */
/* OpBaseCurve / SplineFunc  (src/ops/curves.rs:33-157)                                   */
/* ------------------------------------------------------------------------------------ */
#define ORC_SPLINE_MAXPTS 66
typedef struct {
  int npoints;                                /* knots after auto-added ends */
  float px[ORC_SPLINE_MAXPTS], py[ORC_SPLINE_MAXPTS];
  float c1s[ORC_SPLINE_MAXPTS], c2s[ORC_SPLINE_MAXPTS], c3s[ORC_SPLINE_MAXPTS];
  int nc1, nc3;
} orc_spline;

/* curves.rs:68-124 ; p = n (x,y) pairs */
static int spline_new(const float *p, int n, orc_spline *s) {
  if (n > ORC_SPLINE_MAXPTS - 2) return -1;
  int np = 0;
  if (n == 0 || (p[0] > 0.0f && p[1] > 0.0f)) { s->px[np] = 0.0f; s->py[np] = 0.0f; np++; }
  for (int i = 0; i < n; i++) { s->px[np] = p[2*i]; s->py[np] = p[2*i+1]; np++; }
  if (n == 0 || (p[2*(n-1)] < 1.0f && p[2*(n-1)+1] < 1.0f)) { s->px[np] = 1.0f; s->py[np] = 1.0f; np++; }
  s->npoints = np;
  if (np < 2) return -1;                       /* reference would index slopes[0] out of bounds (panic) */
  float dxs[ORC_SPLINE_MAXPTS], dys[ORC_SPLINE_MAXPTS], slopes[ORC_SPLINE_MAXPTS];
  int nd = np - 1;
  for (int i = 0; i < nd; i++) {
    float dx = s->px[i+1] - s->px[i];
    float dy = s->py[i+1] - s->py[i];
    dxs[i] = dx; dys[i] = dy; slopes[i] = dy / dx;
  }
  (void)dys;
  int nc1 = 0;
  s->c1s[nc1++] = slopes[0];
  for (int i = 0; i < nd - 1; i++) {
    float m = slopes[i], next = slopes[i+1];
    if (m * next <= 0.0f) s->c1s[nc1++] = 0.0f;
    else {
      float dx = dxs[i], dxnext = dxs[i+1];
      float common = dx + dxnext;
      s->c1s[nc1++] = 3.0f * common / ((common + dxnext) / m + (common + dx) / next);
    }
  }
  s->c1s[nc1++] = slopes[nd - 1];
  s->nc1 = nc1;
  int nc3 = 0;
  for (int i = 0; i < nc1 - 1; i++) {
    float c1 = s->c1s[i], slope = slopes[i];
    float invdx = 1.0f / dxs[i];
    float common = c1 + s->c1s[i+1] - slope - slope;
    s->c2s[nc3] = (slope - c1 - common) * invdx;
    s->c3s[nc3] = common * invdx * invdx;
    nc3++;
  }
  s->nc3 = nc3;
  return 0;
}
/* curves.rs:126-157 */
static inline float spline_interpolate(const orc_spline *s, float val) {
  float end = s->px[s->npoints - 1];
  if (val >= end) return s->py[s->npoints - 1];
  float first = s->px[0];
  if (val <= first) return s->py[0];
  int64_t low = 0, mid, high = (int64_t)s->nc3 - 1;
  while (low <= high) {
    mid = (low + high) / 2;
    float xhere = s->px[mid];
    if (xhere < val) low = mid + 1;
    else if (xhere > val) high = mid - 1;
    else return s->py[mid];                    /* also taken by NaN (both compares false) */
  }
  size_t i = (size_t)(high > 0 ? high : 0);
  float diff = val - s->px[i];
  return s->py[i] + s->c1s[i]*diff + s->c2s[i]*diff*diff + s->c3s[i]*diff*diff*diff;
}
/* Exposes SplineFunc for tests: returns knot count, fills arrays (each sized >= npts+2) */
ORC_API int orc_spline_new(const float *pts, int npts, float *px, float *py, float *c1s, float *c2s, float *c3s) {
  orc_spline s; if (spline_new(pts, npts, &s)) return -1;
  memcpy(px, s.px, s.npoints * sizeof(float)); memcpy(py, s.py, s.npoints * sizeof(float));
  memcpy(c1s, s.c1s, s.nc1 * sizeof(float)); memcpy(c2s, s.c2s, s.nc3 * sizeof(float)); memcpy(c3s, s.c3s, s.nc3 * sizeof(float));
  return s.npoints;
}
ORC_API int orc_spline_interpolate(const float *pts, int npts, const float *in, float *out, size_t n) {
  orc_spline s; if (spline_new(pts, npts, &s)) return -1;
  for (size_t i = 0; i < n; i++) out[i] = spline_interpolate(&s, in[i]);
  return 0;
}
/* curves.rs:33-49.  Returns 0 = no-op (caller keeps input; out3 untouched), 1 = ran, -1 error. */
ORC_API int orc_basecurve(const float *src3, size_t width, size_t height, float exposure,
                          const float *pts, int npts, float *out3) {
  if (npts == 0 && fabsf(exposure) < 0.001f) return 0;
  float fp[2 * ORC_SPLINE_MAXPTS];
  if (npts > ORC_SPLINE_MAXPTS - 2) return -1;
  float mulv = exp2f(exposure);
  for (int i = 0; i < npts; i++) { fp[2*i] = pts[2*i]; fp[2*i+1] = pts[2*i+1] * mulv; }
  orc_spline s; if (spline_new(fp, npts, &s)) return -1;
  memcpy(out3, src3, width * height * 3 * sizeof(float));     /* mutate_lines_copying */
  #pragma omp parallel for schedule(dynamic, 1)
  for (size_t row = 0; row < height; row++) {
    float *line = out3 + row * width * 3;
    for (size_t c = 0; c < width; c++) line[3*c] = spline_interpolate(&s, line[3*c]);
  }
  return 1;
}

/* ------------------------------------------------------------------------------------ */
/* OpGamma  (src/ops/gamma.rs:16-26)                                                      */
/* ------------------------------------------------------------------------------------ */
/* Returns 0 = no-op (linear), 1 = ran */
ORC_API int orc_gamma(const float *src, size_t width, size_t height, size_t colors, int linear, float *out) {
  if (linear) return 0;
  luts_init();
  memcpy(out, src, width * height * colors * sizeof(float));
  #pragma omp parallel for schedule(dynamic, 1)
  for (size_t row = 0; row < height; row++) {
    float *line = out + row * width * colors;
    for (size_t i = 0; i < width * colors; i++)
      line[i] = apply_srgb_gamma(rs_min(rs_max(line[i], 0.0f), 1.0f));
  }
  return 1;
}

/* ------------------------------------------------------------------------------------ */
/* OpTransform / rotate_buffer  (src/ops/transform.rs:56-144)                             */
/* rawloader Orientation::{to_flips,from_flips}: absent dependency, table derived from     */
/* and pinned by the nine goldens in transform.rs:168-278                                 */
/* ------------------------------------------------------------------------------------ */
enum { ORC_OR_NORMAL = 0, ORC_OR_HFLIP, ORC_OR_ROT180, ORC_OR_VFLIP, ORC_OR_TRANSPOSE,
       ORC_OR_ROT90, ORC_OR_TRANSVERSE, ORC_OR_ROT270, ORC_OR_UNKNOWN };
enum { ORC_ROT_NORMAL = 0, ORC_ROT_90, ORC_ROT_180, ORC_ROT_270 };

/* (transpose, flip_x, flip_y) */
ORC_API void orc_orientation_to_flips(int o, int *f3) {
  static const int t[9][3] = {
    {0,0,0}, /* Normal */   {0,1,0}, /* HorizontalFlip */ {0,1,1}, /* Rotate180 */
    {0,0,1}, /* VerticalFlip */ {1,0,0}, /* Transpose */  {1,0,1}, /* Rotate90 */
    {1,1,1}, /* Transverse */   {1,1,0}, /* Rotate270 */  {0,0,0}, /* Unknown */
  };
  f3[0] = t[o][0]; f3[1] = t[o][1]; f3[2] = t[o][2];
}
ORC_API int orc_orientation_from_flips(int transpose, int fx, int fy) {
  for (int o = 0; o < 8; o++) { int f[3]; orc_orientation_to_flips(o, f); if (f[0] == !!transpose && f[1] == !!fx && f[2] == !!fy) return o; }
  return ORC_OR_UNKNOWN;
}
/* transform.rs:24-36: img.orientation -> (rotation, fliph, flipv) */
ORC_API void orc_transform_new(int orientation, int *rot_fh_fv) {
  int r = ORC_ROT_NORMAL, fh = 0, fv = 0;
  switch (orientation) {
    case ORC_OR_NORMAL: case ORC_OR_UNKNOWN: break;
    case ORC_OR_VFLIP: fv = 1; break;
    case ORC_OR_HFLIP: fh = 1; break;
    case ORC_OR_ROT180: r = ORC_ROT_180; break;
    case ORC_OR_TRANSPOSE: r = ORC_ROT_90; fv = 1; break;
    case ORC_OR_ROT90: r = ORC_ROT_90; break;
    case ORC_OR_ROT270: r = ORC_ROT_270; break;
    case ORC_OR_TRANSVERSE: r = ORC_ROT_270; fh = 1; break;
  }
  rot_fh_fv[0] = r; rot_fh_fv[1] = fh; rot_fh_fv[2] = fv;
}
/* transform.rs:58-66: recompose the effective orientation */
ORC_API int orc_transform_orientation(int rotation, int fliph, int flipv) {
  static const int base[4] = { ORC_OR_NORMAL, ORC_OR_ROT90, ORC_OR_ROT180, ORC_OR_ROT270 };
  int f[3]; orc_orientation_to_flips(base[rotation], f);
  return orc_orientation_from_flips(f[0], f[1] ^ !!fliph, f[2] ^ !!flipv);
}
/* transform.rs:87-144.  out sized width*height*3; *ow,*oh receive the output dims. */
ORC_API int orc_rotate_buffer(const float *src3, size_t bwidth, size_t bheight, int orientation,
                              float *out3, size_t *ow, size_t *oh) {
  if (orientation == ORC_OR_NORMAL || orientation == ORC_OR_UNKNOWN) {
    memcpy(out3, src3, bwidth * bheight * 3 * sizeof(float)); *ow = bwidth; *oh = bheight; return 0;
  }
  int64_t width = (int64_t)bwidth, height = (int64_t)bheight;
  int f[3]; orc_orientation_to_flips(orientation, f);
  int transpose = f[0], flip_x = f[1], flip_y = f[2];
  int64_t base_offset = 0, x_step = 3, y_step = width * 3;
  if (flip_x) { x_step = -x_step; base_offset += (width - 1) * 3; }
  if (flip_y) { y_step = -y_step; base_offset += width * (height - 1) * 3; }
  if (transpose) {
    int64_t t = width; width = height; height = t;
    t = x_step; x_step = y_step; y_step = t;
  }
  *ow = (size_t)width; *oh = (size_t)height;
  #pragma omp parallel for schedule(dynamic, 1)
  for (int64_t row = 0; row < height; row++) {
    float *line = out3 + row * width * 3;
    int64_t line_offset = base_offset + y_step * row;
    for (int64_t col = 0; col < width; col++) {
      int64_t offset = line_offset + x_step * col;
      for (int c = 0; c < 3; c++) line[col * 3 + c] = src3[offset + c];
    }
  }
  return 0;
}
/* transform.rs:75-84 */
ORC_API void orc_transform_forward(int rotation, size_t w, size_t h, size_t *ow, size_t *oh) {
  if (rotation == ORC_ROT_90 || rotation == ORC_ROT_270) { *ow = h; *oh = w; } else { *ow = w; *oh = h; }
}

/* ------------------------------------------------------------------------------------ */
/* OpRotateCrop  (src/ops/rotatecrop.rs:39-164)                                           */
/* ------------------------------------------------------------------------------------ */
typedef struct {
  float crop_top, crop_right, crop_bottom, crop_left, rotation;
  float input_ratio;
  int has_output_size; size_t out_w, out_h;
} orc_rotatecrop;

static const float RC_EPSILON = 1.0f / 1000000.0f;         /* rotatecrop.rs:7 */
#define RC_FRAC_PI_2 1.57079632679489661923132169163975144f

static void rc_reset(orc_rotatecrop *op) { op->input_ratio = 1.0f; op->has_output_size = 0; op->out_w = op->out_h = 0; }
/* rotatecrop.rs:89-95 */
static int rc_noop(const orc_rotatecrop *op) {
  return fabsf(op->rotation) < RC_EPSILON && fabsf(op->crop_top) < RC_EPSILON && fabsf(op->crop_right) < RC_EPSILON &&
         fabsf(op->crop_bottom) < RC_EPSILON && fabsf(op->crop_left) < RC_EPSILON;
}
/* rotatecrop.rs:97-109 */
static void rc_rotate_point_reverse(const orc_rotatecrop *op, float x, float y, float width, float height,
                                    float swidth, float sheight, int64_t *ox, int64_t *oy) {
  if (op->rotation < RC_EPSILON) { *ox = f32_as_isize(x); *oy = f32_as_isize(y); }
  else {
    float angle = RC_FRAC_PI_2 * (op->rotation > 1.0f ? 1.0f : op->rotation);
    float sn = sinf(angle), cs = cosf(angle);
    float tx = x - (width / 2.0f), ty = y - (height / 2.0f);
    float nx = tx * cs + ty * sn + (swidth / 2.0f);
    float ny = -tx * sn + ty * cs + (sheight / 2.0f);
    *ox = f32_as_isize(nx); *oy = f32_as_isize(ny);
  }
}
/* rotatecrop.rs:111-163 */
static void rc_calc_size(const orc_rotatecrop *op, size_t owidth, size_t oheight, int reverse, size_t *rw, size_t *rh) {
  if (rc_noop(op)) { *rw = owidth; *rh = oheight; return; }
  float width = (float)owidth, height = (float)oheight;
  if (!(reverse || op->rotation < RC_EPSILON)) {
    float angle = RC_FRAC_PI_2 * (op->rotation > 1.0f ? 1.0f : op->rotation);
    float sn = sinf(angle), cs = cosf(angle);
    float w2 = width * cs + height * sn, h2 = width * sn + height * cs;
    width = w2; height = h2;
  }
  float nwidth, nheight;
  {
    float ratio = 1.0f - op->crop_left - op->crop_right;
    nwidth = reverse ? roundf(width / ratio) : roundf(width * ratio);
    if (ratio < RC_EPSILON || nwidth < 1.0f) { *rw = owidth; *rh = oheight; return; }
  }
  {
    float ratio = 1.0f - op->crop_top - op->crop_bottom;
    nheight = reverse ? roundf(height / ratio) : roundf(height * ratio);
    if (ratio < RC_EPSILON || nheight < 1.0f) { *rw = owidth; *rh = oheight; return; }
  }
  if (!(!reverse || op->rotation < RC_EPSILON)) {
    float angle = RC_FRAC_PI_2 * (op->rotation > 1.0f ? 1.0f : op->rotation);
    float sn = sinf(angle), cs = cosf(angle);
    float w2 = roundf(nheight / (sn + (cs / op->input_ratio)));
    float h2 = roundf(w2 / op->input_ratio);
    nwidth = w2; nheight = h2;
  }
  *rw = f32_as_usize(nwidth); *rh = f32_as_usize(nheight);
}
/* rotatecrop.rs:66-76 */
static void rc_transform_forward(orc_rotatecrop *op, size_t w, size_t h, size_t *ow, size_t *oh) {
  if (op->has_output_size) { *ow = op->out_w; *oh = op->out_h; }
  else { op->input_ratio = (float)w / (float)h; rc_calc_size(op, w, h, 0, ow, oh); }
}
/* rotatecrop.rs:78-82 */
static void rc_transform_reverse(orc_rotatecrop *op, size_t w, size_t h, size_t *ow, size_t *oh) {
  op->has_output_size = 1; op->out_w = w; op->out_h = h;
  rc_calc_size(op, w, h, 1, ow, oh);
}
static void rc_init(orc_rotatecrop *op, const float *p5) {
  op->crop_top = p5[0]; op->crop_right = p5[1]; op->crop_bottom = p5[2]; op->crop_left = p5[3]; op->rotation = p5[4];
  rc_reset(op);
}
/* params5 = crop_top, crop_right, crop_bottom, crop_left, rotation.
 * state3 (in/out) = input_ratio, has_output_size, out_w, out_h packed as doubles for ctypes ease */
ORC_API void orc_rotatecrop_calc_size(const float *params5, float input_ratio, size_t w, size_t h, int reverse, size_t *ow, size_t *oh) {
  orc_rotatecrop op; rc_init(&op, params5); op.input_ratio = input_ratio; rc_calc_size(&op, w, h, reverse, ow, oh);
}
/* rotatecrop.rs:39-64.  Returns 0 no-op / error-return-input, 1 ran (out sized by *ow x *oh x colors;
 * call once with out==NULL to get the size). */
ORC_API int orc_rotatecrop_run(const float *params5, const float *src, size_t width, size_t height, size_t colors,
                               float *out, size_t *ow, size_t *oh) {
  orc_rotatecrop op; rc_init(&op, params5);
  if (rc_noop(&op)) { *ow = width; *oh = height; return 0; }
  float swidth = (float)width, sheight = (float)height;
  size_t nwidth, nheight; rc_calc_size(&op, width, height, 0, &nwidth, &nheight);
  float fnwidth = (float)nwidth, fnheight = (float)nheight;
  float x = floorf(swidth * op.crop_left);
  if (x < 0.0f || x > swidth) { *ow = width; *oh = height; return 0; }
  float y = floorf(sheight * op.crop_top);
  if (y < 0.0f || y > sheight) { *ow = width; *oh = height; return 0; }
  int64_t tl[2], tr[2], bl[2];
  rc_rotate_point_reverse(&op, x, y, fnwidth, fnheight, swidth, sheight, &tl[0], &tl[1]);
  rc_rotate_point_reverse(&op, x + fnwidth - 1.0f, y, fnwidth, fnheight, swidth, sheight, &tr[0], &tr[1]);
  rc_rotate_point_reverse(&op, x, y + fnheight - 1.0f, fnwidth, fnheight, swidth, sheight, &bl[0], &bl[1]);
  *ow = nwidth; *oh = nheight;
  if (!out) return 1;
  return orc_transform_buffer_f32(src, width, height, tl[0], tl[1], tr[0], tr[1], bl[0], bl[1],
                                  nwidth, nheight, colors, NULL, out) ? -1 : 1;
}
/* Exposes the three corner points for boundary tests */
ORC_API int orc_rotatecrop_corners(const float *params5, size_t width, size_t height, int64_t *pts6, size_t *ow, size_t *oh) {
  orc_rotatecrop op; rc_init(&op, params5);
  if (rc_noop(&op)) return 0;
  float swidth = (float)width, sheight = (float)height;
  size_t nwidth, nheight; rc_calc_size(&op, width, height, 0, &nwidth, &nheight);
  float fnwidth = (float)nwidth, fnheight = (float)nheight;
  float x = floorf(swidth * op.crop_left); if (x < 0.0f || x > swidth) return 0;
  float y = floorf(sheight * op.crop_top); if (y < 0.0f || y > sheight) return 0;
  rc_rotate_point_reverse(&op, x, y, fnwidth, fnheight, swidth, sheight, &pts6[0], &pts6[1]);
  rc_rotate_point_reverse(&op, x + fnwidth - 1.0f, y, fnwidth, fnheight, swidth, sheight, &pts6[2], &pts6[3]);
  rc_rotate_point_reverse(&op, x, y + fnheight - 1.0f, fnwidth, fnheight, swidth, sheight, &pts6[4], &pts6[5]);
  *ow = nwidth; *oh = nheight;
  return 1;
}

/* Restatements of the two loop-nest size tests, run in C because they are ~50M and ~2.7M
 * iterations.  Return the number of failing cases (reference asserts zero).
 * rotatecrop.rs:274-294 roundtrip_transform */
ORC_API uint64_t orc_selftest_rotatecrop_roundtrip_transform(void) {
  uint64_t bad = 0;
  orc_rotatecrop op; float z[5] = {0, 0, 0, 0, 0}; rc_init(&op, z);
  for (int dim = 0; dim < 10000; dim += 89)
    for (int crop1 = 0; crop1 < 65535; crop1 += 97)
      for (int crop2 = 0; crop2 < 65535; crop2 += 101) {
        op.crop_top = input16bit((uint16_t)crop1); op.crop_right = input16bit((uint16_t)crop1);
        op.crop_bottom = input16bit((uint16_t)crop2); op.crop_left = input16bit((uint16_t)crop2);
        size_t iw, ih, rw, rh;
        rc_transform_reverse(&op, (size_t)dim, (size_t)dim, &iw, &ih);
        rc_transform_forward(&op, iw, ih, &rw, &rh);
        if (rw != (size_t)dim || rh != (size_t)dim) bad++;
      }
  return bad;
}
/* rotatecrop.rs:296-312 roundtrip_transform_rotation */
ORC_API uint64_t orc_selftest_rotatecrop_roundtrip_rotation(void) {
  uint64_t bad = 0;
  orc_rotatecrop op; float z[5] = {0, 0, 0, 0, 0}; rc_init(&op, z);
  for (int width = 0; width < 10000; width += 89)
    for (int height = 0; height < 10000; height += 97)
      for (int rotation = 0; rotation < 255; rotation++) {
        op.rotation = input8bit((uint8_t)rotation);
        size_t a, b, c, d, e, f;
        rc_transform_forward(&op, (size_t)width, (size_t)height, &a, &b);
        rc_transform_reverse(&op, a, b, &c, &d);
        rc_transform_forward(&op, c, d, &e, &f);
        if (e != a || f != b) bad++;
      }
  return bad;
}

/* ------------------------------------------------------------------------------------ */
/* Pipeline driver  (src/pipeline.rs:311-375, :404-421, :451-468)                         */
/* ------------------------------------------------------------------------------------ */
typedef struct {
  /* source */
  int source_kind;            /* 0 raw u16, 1 raw f32, 2 other rgb8, 3 other rgb16 */
  const void *data;
  size_t width, height;       /* RawImage.width/height or image dims */
  int cpp;                    /* raw: 1 or 3 */
  int is_cfa;                 /* raw: cfa.is_valid() */
  char cfa[160];              /* already cropped_cfa() */
  /* gofloat */
  size_t crop_top, crop_right, crop_bottom, crop_left;
  float blacklevels[4], whitelevels[4];
  /* rotatecrop */
  float rc[5];
  /* tolab */
  float cam_to_xyz_normalized[12];
  float wb_coeffs[4];
  /* basecurve */
  float exposure; int npoints; float points[2 * 64];
  /* transform */
  int rotation, fliph, flipv;
  /* settings */
  size_t maxwidth, maxheight;
  int linear;
  int use_fastpath;           /* PipelineSettings.use_fastpath (pipeline.rs:117) */
} orc_pipeline;

ORC_API size_t orc_pipeline_sizeof(void) { return sizeof(orc_pipeline); }

/* Size negotiation: pipeline.rs:314-338.  Fills demosaic_w/h and the final output w/h. */
ORC_API int orc_pipeline_sizes(const orc_pipeline *p, size_t *demosaic_w, size_t *demosaic_h, size_t *final_w, size_t *final_h) {
  orc_rotatecrop rc; rc_init(&rc, p->rc);
  size_t width = p->width, height = p->height, sz[4];
  /* forward: gofloat, demosaic(id), rotatecrop, tolab.. (id), transform */
  if (orc_size_image(p->crop_top, p->crop_right, p->crop_bottom, p->crop_left, width, height, sz)) return -1;
  width = sz[2]; height = sz[3];
  rc_transform_forward(&rc, width, height, &width, &height);
  orc_transform_forward(p->rotation, width, height, &width, &height);
  float scale; size_t nw, nh;
  orc_calculate_scaling_total(width, height, p->maxwidth, p->maxheight, &scale, &nw, &nh);
  width = nw; height = nh;
  /* reverse: transform, gamma.., rotatecrop, demosaic(id), gofloat(id) */
  orc_transform_forward(p->rotation, width, height, &width, &height);
  rc_transform_reverse(&rc, width, height, &width, &height);
  *demosaic_w = width; *demosaic_h = height;
  /* What run() then PRODUCES (the size output_8bit reports; tests/maxsize_test.rs asserts on it): the ops size their outputs
   * from the buffer they are handed, not from the negotiation, so after a rotatecrop the result can differ from the forward
   * fold by a pixel.  gofloat: the cropped frame; demosaic (demosaic.rs:27-61): the demosaic size when it scales, else its
   * input; rotatecrop (rotatecrop.rs:39-64): calc_size of its input unless it is a no-op / rejects its crops; transform:
   * swaps the sides for the transposing orientations. */
  size_t w = sz[2], h = sz[3];
  { float sc; size_t a, b; orc_calculate_scaling_total(w, h, *demosaic_w, *demosaic_h, &sc, &a, &b); if (sc > 1.0f) { w = *demosaic_w; h = *demosaic_h; } }
  { int64_t pts[6]; size_t ow, oh; if (orc_rotatecrop_corners(p->rc, w, h, pts, &ow, &oh)) { w = ow; h = oh; } }
  { int f[3]; orc_orientation_to_flips(orc_transform_orientation(p->rotation, p->fliph, p->flipv), f); if (f[0]) { size_t t = w; w = h; h = t; } }
  *final_w = w; *final_h = h;
  return 0;
}

/* Pipeline::run (pipeline.rs:311-375) with cache == None.  Returns a malloc'd 3-channel f32 buffer
 * (caller frees with orc_free) and its dims, or NULL. */
ORC_API float *orc_pipeline_run(const orc_pipeline *p, size_t *out_w, size_t *out_h) {
  luts_init();
  size_t dw, dh, fw, fh;
  if (orc_pipeline_sizes(p, &dw, &dh, &fw, &fh)) return NULL;
  size_t sz[4];
  orc_size_image(p->crop_top, p->crop_right, p->crop_bottom, p->crop_left, p->width, p->height, sz);
  size_t x = sz[0], y = sz[1], w = sz[2], h = sz[3];
  /* --- gofloat --- */
  float *buf; size_t colors; int monochrome = 0;
  if (p->source_kind <= 1) {
    if (p->cpp == 1 && !p->is_cfa) {
      colors = 4; monochrome = 1; buf = (float *)malloc(w * h * 4 * sizeof(float));
      if (p->source_kind == 0) orc_gofloat_mono_u16((const uint16_t *)p->data, p->width, x, y, w, h, p->blacklevels[0], p->whitelevels[0], buf);
      else orc_gofloat_mono_f32((const float *)p->data, p->width, x, y, w, h, p->blacklevels[0], p->whitelevels[0], buf);
    } else if (p->cpp == 3) {
      colors = 4; buf = (float *)malloc(w * h * 4 * sizeof(float));
      if (p->source_kind == 0) orc_gofloat_rgb_u16((const uint16_t *)p->data, p->width, x, y, w, h, p->blacklevels, p->whitelevels, buf);
      else orc_gofloat_rgb_f32((const float *)p->data, p->width, x, y, w, h, p->blacklevels, p->whitelevels, buf);
    } else {
      colors = 1; buf = (float *)malloc(w * h * sizeof(float));
      if (p->source_kind == 0) orc_gofloat_cfa_u16((const uint16_t *)p->data, p->width, x, y, w, h, p->blacklevels[0], p->whitelevels[0], buf);
      else orc_gofloat_cfa_f32((const float *)p->data, p->width, x, y, w, h, p->blacklevels[0], p->whitelevels[0], buf);
    }
  } else {
    colors = 4; buf = (float *)malloc(w * h * 4 * sizeof(float));
    if (p->source_kind == 2) orc_gofloat_other_u8((const uint8_t *)p->data, p->width, x, y, w, h, buf);
    else orc_gofloat_other_u16((const uint16_t *)p->data, p->width, x, y, w, h, buf);
  }
  /* --- demosaic --- */
  {
    float scale; size_t sw, sh;
    orc_calculate_scaling_total(w, h, dw, dh, &scale, &sw, &sh);
    size_t ow = w, oh = h;
    int will_scale = !(scale <= 1.0f);
    size_t aw = will_scale ? dw : w, ah = will_scale ? dh : h;
    float *out4 = (float *)malloc((aw * ah > w * h ? aw * ah : w * h) * 4 * sizeof(float));
    int rc = orc_demosaic_run(p->cfa, buf, w, h, colors, dw, dh, out4, &ow, &oh);
    if (rc < 0) { free(buf); free(out4); return NULL; }
    if (rc == 0) free(out4); else { free(buf); buf = out4; }
    w = ow; h = oh; colors = 4;
  }
  /* --- rotatecrop --- */
  {
    size_t ow, oh;
    int rc = orc_rotatecrop_run(p->rc, buf, w, h, colors, NULL, &ow, &oh);
    if (rc == 1) {
      float *o = (float *)malloc(ow * oh * colors * sizeof(float));
      if (orc_rotatecrop_run(p->rc, buf, w, h, colors, o, &ow, &oh) < 0) { free(buf); free(o); return NULL; }
      free(buf); buf = o; w = ow; h = oh;
    }
  }
  /* --- tolab --- */
  {
    float *o = (float *)malloc(w * h * 3 * sizeof(float));
    orc_tolab(buf, w, h, monochrome, p->wb_coeffs, p->cam_to_xyz_normalized, o);
    free(buf); buf = o; colors = 3;
  }
  /* --- basecurve --- */
  {
    float *o = (float *)malloc(w * h * 3 * sizeof(float));
    int rc = orc_basecurve(buf, w, h, p->exposure, p->points, p->npoints, o);
    if (rc == 1) { free(buf); buf = o; } else free(o);
    if (rc < 0) { free(buf); return NULL; }
  }
  /* --- fromlab --- */
  {
    float *o = (float *)malloc(w * h * 3 * sizeof(float));
    orc_fromlab(buf, w, h, o);
    free(buf); buf = o;
  }
  /* --- gamma --- */
  {
    float *o = (float *)malloc(w * h * 3 * sizeof(float));
    if (orc_gamma(buf, w, h, 3, p->linear, o) == 1) { free(buf); buf = o; } else free(o);
  }
  /* --- transform --- */
  {
    int orientation = orc_transform_orientation(p->rotation, p->fliph, p->flipv);
    if (!(orientation == ORC_OR_NORMAL || orientation == ORC_OR_UNKNOWN)) {
      float *o = (float *)malloc(w * h * 3 * sizeof(float));
      size_t ow, oh;
      orc_rotate_buffer(buf, w, h, orientation, o, &ow, &oh);
      free(buf); buf = o; w = ow; h = oh;
    }
  }
  *out_w = w; *out_h = h;
  return buf;
}
ORC_API void orc_free(void *p) { free(p); }

/* Pipeline::default_ops (pipeline.rs:286-288) for a raster source: the ops equal PipelineOps::new(Other), compared
 * bitwise as the reference's serialise-and-hash equality does (-0.0 != 0.0).  The reference also serialises
 * OpRotateCrop's negotiated state, which a previous slow-path run leaves behind; this stateless restatement
 * compares the user-visible fields only. */
static int bits_eq(float a, float b) { return memcmp(&a, &b, 4) == 0; }
ORC_API int orc_pipeline_default_ops_other(const orc_pipeline *p) {
  if (p->source_kind < 2) return 0;
  if (p->crop_top || p->crop_right || p->crop_bottom || p->crop_left || p->is_cfa || p->cfa[0]) return 0;
  for (int i = 0; i < 4; i++) if (!bits_eq(p->blacklevels[i], 0.0f) || !bits_eq(p->whitelevels[i], 0.0f)) return 0;
  for (int i = 0; i < 5; i++) if (!bits_eq(p->rc[i], 0.0f)) return 0;
  float m[12]; orc_const_srgb_d65_43(m);
  for (int i = 0; i < 12; i++) if (!bits_eq(p->cam_to_xyz_normalized[i], m[i])) return 0;
  const float wb[4] = {1.0f, 1.0f, 1.0f, 0.0f};
  for (int i = 0; i < 4; i++) if (!bits_eq(p->wb_coeffs[i], wb[i])) return 0;
  if (!bits_eq(p->exposure, 0.0f) || p->npoints != 0) return 0;
  if (p->rotation != 0 || p->fliph || p->flipv) return 0;
  return 1;
}
/* image 0.24 DynamicImage::to_rgb8 / to_rgb16 channel conversions (crate absent from /root/reference; its published
 * FromPrimitive impls): u8 -> u16 = c * 257, u16 -> u8 = (c + 128) / 257.  Parity unpinned. */
static uint16_t chan_8_to_16(uint8_t c) { return (uint16_t)(c * 257u); }
static uint8_t chan_16_to_8(uint16_t c) { return (uint8_t)(((uint32_t)c + 128u) / 257u); }

/* Pipeline::output_8bit: raster fast path (pipeline.rs:381-402), else the slow path (:404-421: linear=false, run, serial quantise) */
ORC_API uint8_t *orc_pipeline_output_8bit(orc_pipeline *p, size_t *out_w, size_t *out_h) {
  if (p->use_fastpath && orc_pipeline_default_ops_other(p)) {
    const size_t n = p->width * p->height * 3;
    uint8_t *rgb = (uint8_t *)malloc(n ? n : 1);
    if (p->source_kind == 2) memcpy(rgb, p->data, n);
    else for (size_t i = 0; i < n; i++) rgb[i] = chan_16_to_8(((const uint16_t *)p->data)[i]);
    float scale; size_t nw, nh;
    orc_calculate_scaling_total(p->width, p->height, p->maxwidth, p->maxheight, &scale, &nw, &nh);     /* scaling_size */
    *out_w = nw; *out_h = nh;
    if (nw == p->width && nh == p->height) return rgb;
    uint8_t *out = (uint8_t *)calloc(nw * nh * 3, 1);
    orc_scale_down_srgb(rgb, p->width, p->height, nw, nh, out);
    free(rgb);
    return out;
  }
  p->linear = 0;
  float *buf = orc_pipeline_run(p, out_w, out_h);
  if (!buf) return NULL;
  size_t n = *out_w * *out_h * 3;
  uint8_t *img = (uint8_t *)malloc(n);
  for (size_t i = 0; i < n; i++) img[i] = output8bit(buf[i]);
  free(buf);
  return img;
}
/* Pipeline::output_16bit: raster fast path (pipeline.rs:428-449), else the slow path (:451-468: linear=true) */
ORC_API uint16_t *orc_pipeline_output_16bit(orc_pipeline *p, size_t *out_w, size_t *out_h) {
  if (p->use_fastpath && orc_pipeline_default_ops_other(p)) {
    const size_t n = p->width * p->height * 3;
    uint16_t *rgb = (uint16_t *)malloc(n ? n * 2 : 2);
    if (p->source_kind == 3) memcpy(rgb, p->data, n * 2);
    else for (size_t i = 0; i < n; i++) rgb[i] = chan_8_to_16(((const uint8_t *)p->data)[i]);
    float scale; size_t nw, nh;
    orc_calculate_scaling_total(p->width, p->height, p->maxwidth, p->maxheight, &scale, &nw, &nh);
    *out_w = nw; *out_h = nh;
    if (nw == p->width && nh == p->height) return rgb;
    uint16_t *out = (uint16_t *)calloc(nw * nh * 3, 2);
    orc_scale_down_srgb16(rgb, p->width, p->height, nw, nh, out);
    free(rgb);
    return out;
  }
  p->linear = 1;
  float *buf = orc_pipeline_run(p, out_w, out_h);
  if (!buf) return NULL;
  size_t n = *out_w * *out_h * 3;
  uint16_t *img = (uint16_t *)malloc(n * 2);
  for (size_t i = 0; i < n; i++) img[i] = output16bit(buf[i]);
  free(buf);
  return img;
}
