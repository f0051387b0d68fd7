// ipk_host.hpp -- host-side (CPU, once-per-op) maths of the hot path: everything the reference
// computes outside its per-pixel loops and that decides kernel parameters or buffer sizes.
// Citations are file:line in the reference (pedrocr/imagepipe 0.5.0).
//
// Compiled with -ffp-contract=off: Rust evaluates f32 expressions left to right without FMA
// contraction, and these values (spline coefficients, matrices, sizes) feed bit-exact kernels.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstddef>
#include <cstring>
#include <limits>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace ipk {

// ---- Rust `as` casts ---------------------------------------------------------------------
inline size_t f32_to_usize(float f) {            // saturating, NaN -> 0
  if (!(f > 0.0f)) return 0;
  if (f >= 18446744073709551616.0f) return std::numeric_limits<size_t>::max();
  return static_cast<size_t>(f);
}
inline int64_t f32_to_isize(float f) {
  if (f != f) return 0;
  if (f >= 9223372036854775808.0f) return std::numeric_limits<int64_t>::max();
  if (f <= -9223372036854775808.0f) return std::numeric_limits<int64_t>::min();
  return static_cast<int64_t>(f);
}

// ---- colour constants (src/color_conversions.rs:1-39) -------------------------------------
struct Mat33 { float m[3][3]; };
inline Mat33 srgb_d65_33() {
  return Mat33{{{0.4124564f, 0.3575761f, 0.1804375f},
                {0.2126729f, 0.7151522f, 0.0721750f},
                {0.0193339f, 0.1191920f, 0.9503041f}}};
}
// cofactor inverse in f32 (src/color_conversions.rs:20-39); XYZ_D65_33 = inverse(SRGB_D65_33)
inline Mat33 inverse(const Mat33 &a) {
  const auto &i = a.m;
  const float invdet = 1.0f / (i[0][0] * (i[1][1] * i[2][2] - i[2][1] * i[1][2]) -
                               i[0][1] * (i[1][0] * i[2][2] - i[1][2] * i[2][0]) +
                               i[0][2] * (i[1][0] * i[2][1] - i[1][1] * i[2][0]));
  Mat33 o;
  o.m[0][0] =  (i[1][1] * i[2][2] - i[2][1] * i[1][2]) * invdet;
  o.m[0][1] = -(i[0][1] * i[2][2] - i[0][2] * i[2][1]) * invdet;
  o.m[0][2] =  (i[0][1] * i[1][2] - i[0][2] * i[1][1]) * invdet;
  o.m[1][0] = -(i[1][0] * i[2][2] - i[1][2] * i[2][0]) * invdet;
  o.m[1][1] =  (i[0][0] * i[2][2] - i[0][2] * i[2][0]) * invdet;
  o.m[1][2] = -(i[0][0] * i[1][2] - i[1][0] * i[0][2]) * invdet;
  o.m[2][0] =  (i[1][0] * i[2][1] - i[2][0] * i[1][1]) * invdet;
  o.m[2][1] = -(i[0][0] * i[2][1] - i[2][0] * i[0][1]) * invdet;
  o.m[2][2] =  (i[0][0] * i[1][1] - i[1][0] * i[0][1]) * invdet;
  return o;
}
// SRGB_D65_43 (src/color_conversions.rs:12-16), [[f32;4];3] row-major
inline void srgb_d65_43(float out[12]) {
  const Mat33 s = srgb_d65_33();
  for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) out[r * 4 + c] = s.m[r][c]; out[r * 4 + 3] = 0.0f; }
}

// ---- TransformLookup tables (src/color_conversions.rs:80-141) ------------------------------
constexpr int kLutMax = (1 << 13) - 1;   // 8191
constexpr int kLutLen = kLutMax + 2;     // 8193
enum LutId { kLutXyzLab = 0, kLutGammaReverse = 1, kLutGamma = 2 };

inline float lab_transform(float v) {            // :120-124
  const float e = 216.0f / 24389.0f, k = 24389.0f / 27.0f;
  return v > e ? std::cbrt(v) : (k * v + 16.0f) / 116.0f;
}
inline float gamma_reverse_transform(float v) {  // :126-132
  return v < 0.04045f ? v / 12.92f : std::pow((v + 0.055f) / 1.055f, 2.4f);
}
inline float gamma_transform(float v) {          // :134-140
  return v < 0.0031308f ? v * 12.92f : 1.055f * std::pow(v, 1.0f / 2.4f) - 0.055f;
}
inline std::vector<float> build_lut(LutId id) {  // TransformLookup::new :87-100
  std::vector<float> t(kLutLen);
  for (int i = 0; i <= kLutMax + 1; ++i) {
    const float v = static_cast<float>(i) / static_cast<float>(kLutMax);
    t[i] = id == kLutXyzLab ? lab_transform(v) : id == kLutGammaReverse ? gamma_reverse_transform(v) : gamma_transform(v);
  }
  return t;
}

// ---- rawloader CFA (absent dependency; restated -- see DESIGN.md "unpinned") ---------------
struct Cfa {
  int width = 0, height = 0;
  uint8_t pattern[48][48] = {};
  bool valid() const { return width > 0; }
  int color_at(size_t row, size_t col) const { return pattern[(row + 48) % 48][(col + 48) % 48]; }
  // CFA::new(patname).  Pattern strings: the letters R G B E (M = G, Y = E) of one tile, row-major.  The tile's shape is
  //   inferred for 4 (2 x 2), 36 (6 x 6) and 144 (12 x 12) letters, or
  //   stated by the caller as a prefix "WxH:" -- "2x8:RGBE..." -- which is the only way to pass 16 letters: rawloader 0.37 is absent from
  //   /root/reference, its tile shape for that length could not be verified (imagepipe's `8 => 2.0` minscale arm, demosaic.rs:36-37, says 8 wide;
  //   dcraw's filter tables are 2 wide x 8 high), and the caller has the answer in hand (its CFA object's width and height, demosaic.rs:33).
  // W and H must divide 48 (the reference tiles every pattern into 48 x 48, demosaic.rs:77-90 / color_at's `% 48`).
  static bool split_dims(const char *pat, int &w, int &h, const char *&letters) {
    w = h = 0; letters = pat;
    const char *colon = pat ? std::strchr(pat, ':') : nullptr;
    if (!colon) return true;
    int v[2] = {0, 0}, k = 0, digits = 0;
    for (const char *q = pat; q < colon; ++q) {
      if (*q >= '0' && *q <= '9') { v[k] = v[k] * 10 + (*q - '0'); if (++digits > 2) return false; }
      else if (*q == 'x' && k == 0 && digits > 0) { k = 1; digits = 0; }
      else return false;
    }
    if (k != 1 || digits == 0) return false;
    w = v[0]; h = v[1]; letters = colon + 1;
    return w >= 1 && h >= 1 && w <= 48 && h <= 48 && 48 % w == 0 && 48 % h == 0;
  }
  static bool unpinned_length(const char *pat) { return pat && !std::strchr(pat, ':') && std::strlen(pat) == 16; }
  static bool parse(const char *pat, Cfa &out) {
    out = Cfa();
    int w = 0, h = 0; const char *letters = pat;
    if (pat && !split_dims(pat, w, h, letters)) return false;
    const size_t len = letters ? std::strlen(letters) : 0;
    if (w == 0) {
      switch (len) {
        case 0: return true;
        case 4: out.width = 2; out.height = 2; break;
        case 36: out.width = 6; out.height = 6; break;
        case 16: return false;          // shape not stated: refused (IPK_ERR_UNSUPPORTED at the API) rather than guessed
        case 144: out.width = 12; out.height = 12; break;
        default: return false;
      }
    } else {
      if ((size_t)w * (size_t)h != len) return false;
      out.width = w; out.height = h;
    }
    for (size_t i = 0; i < len; ++i) {
      uint8_t v;
      switch (letters[i]) {
        case 'R': v = 0; break; case 'G': v = 1; break; case 'B': v = 2; break; case 'E': v = 3; break;
        case 'M': v = 1; break; case 'Y': v = 3; break;
        default: return false;
      }
      out.pattern[i / out.width][i % out.width] = v;
    }
    for (int r = 0; r < 48; ++r)
      for (int c = 0; c < 48; ++c) out.pattern[r][c] = out.pattern[r % out.height][c % out.width];
    return true;
  }
  // whether a plain string of width*height letters would be read back with this shape
  bool shape_is_inferred() const { return width == height && (width == 2 || width == 6 || width == 12); }
  // CFA::shift(x, y) as used by cropped_cfa()
  std::string shifted_name(int x, int y) const {
    static const char names[4] = {'R', 'G', 'B', 'E'};
    std::string s;
    if (!shape_is_inferred()) s = std::to_string(width) + "x" + std::to_string(height) + ":";
    for (int r = 0; r < height; ++r)
      for (int c = 0; c < width; ++c) s.push_back(names[color_at(size_t(r + y), size_t(c + x))]);
    return s;
  }
  // The per-pixel tap colours of demosaic::full (src/ops/demosaic.rs:77-90), 3 bits per tap packed
  // low-to-high in tap order (-1,-1),(-1,0),(-1,1),(0,-1),(0,0),(0,1),(1,-1),(1,0),(1,1); 4 = discard.
  void demosaic_lookups(uint32_t out[48 * 48]) const {
    static const int off[9][2] = {{-1,-1},{-1,0},{-1,1},{0,-1},{0,0},{0,1},{1,-1},{1,0},{1,1}};
    for (size_t row = 0; row < 48; ++row)
      for (size_t col = 0; col < 48; ++col) {
        const int pix = color_at(row, col);
        uint32_t w = 0;
        for (int i = 0; i < 9; ++i) {
          const int dy = off[i][0], dx = off[i][1];
          const int o = color_at(size_t(48 + dy) + row, size_t(48 + dx) + col);
          const uint32_t c = (o != pix || (dx == 0 && dy == 0)) ? uint32_t(o) : 4u;
          w |= c << (3 * i);
        }
        out[row * 48 + col] = w;
      }
  }
  // demosaic::full restated as arithmetic for the row-walking kernel's generic-CFA mode (any filter without an E/fourth
  // colour): per pattern cell 36 floats = tap weights {0,1} for R, G, B (9 each, the reference's tap order), the packed
  // tap-colour word of demosaic_lookups(), and RN(1/count) as a hi/lo pair per colour (count = contributing taps of an
  // interior pixel; 1 when a colour has none, so that 0/1 = 0 reproduces "stays 0.0").  An interior pixel is then
  //   sum_c = 0.0 + t0*w_c0 + ... + t8*w_c8   (a non-contributing tap adds +-0.0, which never changes a sum that starts at +0.0)
  //   out_c = fma(sum_c, rc_hi, sum_c*rc_lo)  (== sum_c / count for every finite sum in [2^-100, 2^100] and for 0:
  //                                           proven on all f32 for counts 1..9, tests/test_gpu_selftest.py)
  // Returns false when the pattern has a fourth colour.
  bool three_colour() const { for (int r = 0; r < height; ++r) for (int c = 0; c < width; ++c) if (pattern[r][c] > 2) return false; return valid(); }
  static constexpr int kGenCellFloats = 36;
  bool gen_cells(std::vector<float> &out) const {
    static const int off[9][2] = {{-1,-1},{-1,0},{-1,1},{0,-1},{0,0},{0,1},{1,-1},{1,0},{1,1}};
    out.assign(size_t(width) * height * kGenCellFloats, 0.0f);
    for (int y = 0; y < height; ++y)
      for (int x = 0; x < width; ++x) {
        float *cell = &out[(size_t(y) * width + x) * kGenCellFloats];
        const int pix = color_at(size_t(y), size_t(x));
        if (pix > 2) return false;
        uint32_t word = 0; int cnt[3] = {0, 0, 0};
        for (int i = 0; i < 9; ++i) {
          const int dy = off[i][0], dx = off[i][1];
          const int o = color_at(size_t(48 + dy + y), size_t(48 + dx + x));
          if (o > 2) return false;
          const int code = (o != pix || (dx == 0 && dy == 0)) ? o : 4;
          word |= uint32_t(code) << (3 * i);
          if (code < 3) { cell[code * 9 + i] = 1.0f; ++cnt[code]; }
        }
        std::memcpy(&cell[27], &word, 4);
        for (int c = 0; c < 3; ++c) {
          const float n = float(cnt[c] > 0 ? cnt[c] : 1);
          const float hi = 1.0f / n;
          cell[28 + c] = hi;
          cell[31 + c] = float(1.0 / double(n) - double(hi));
        }
      }
    return true;
  }
  // Is this one of the four phases of the RGGB Bayer tile?  color_at(r,c) == RGGB[(r+yoff)&1][(c+xoff)&1]
  bool bayer_phase(int &xoff, int &yoff) const {
    if (width != 2 || height != 2) return false;
    static const int rggb[2][2] = {{0, 1}, {1, 2}};
    for (yoff = 0; yoff < 2; ++yoff)
      for (xoff = 0; xoff < 2; ++xoff) {
        bool ok = true;
        for (int r = 0; r < 2 && ok; ++r)
          for (int c = 0; c < 2 && ok; ++c) ok = pattern[r][c] == rggb[(r + yoff) & 1][(c + xoff) & 1];
        if (ok) return true;
      }
    return false;
  }
};

// ---- OpGoFloat::size_image (src/ops/gofloat.rs:74-82) --------------------------------------
// ---- runtime divisors of the three-step division (ipk_device.hpp cdiv_fast) ------------------------------------------
// q0 = d*rc ; r = fma(-q0, c, d) ; q1 = fma(r, rc, q0) with rc = RN(1/c).  Scaling d or c by a power of two scales every
// intermediate exactly (no under/overflow inside the guarded zone 2^-100 <= |d| <= 2^100, 2^-60 <= c <= 2^60), so whether
// q1 == RN(d/c) depends on the two MANTISSAS only: sweeping all 2^23 dividend mantissas against c's mantissa is a proof
// for that divisor, not a sample (about 10 ms on one host core; memoised per divisor mantissa).
inline bool cdiv_mantissa_exhaustive_ok(float c) {
  if (!(c >= 0x1p-60f && c <= 0x1p60f)) return false;                       // also NaN, <= 0
  uint32_t cb; std::memcpy(&cb, &c, 4);
  const uint32_t cm = cb & 0x007FFFFFu;
  static std::mutex mu;
  static std::map<uint32_t, bool> memo;
  { std::lock_guard<std::mutex> lk(mu); auto it = memo.find(cm); if (it != memo.end()) return it->second; }
  const uint32_t c1b = cm | 0x3F800000u;                                    // c scaled into [1, 2)
  float c1; std::memcpy(&c1, &c1b, 4);
  const float rc = 1.0f / c1;
  bool ok = true;
  for (uint32_t m = 0; m < (1u << 23) && ok; ++m) {
    const uint32_t db = m | 0x3F800000u;
    float d; std::memcpy(&d, &db, 4);
    const float q0 = d * rc;
    const float r = std::fma(-q0, c1, d);
    ok = std::fma(r, rc, q0) == d / c1;
  }
  std::lock_guard<std::mutex> lk(mu);
  memo[cm] = ok;
  return ok;
}

struct Rect { size_t x, y, width, height; };
inline bool size_image(size_t crop_top, size_t crop_right, size_t crop_bottom, size_t crop_left,
                       size_t owidth, size_t oheight, Rect &r) {
  if (owidth < 10 || oheight < 10) return false;          // usize underflow in the reference
  r.x = std::min(crop_left, owidth - 10);
  r.y = std::min(crop_top, oheight - 10);
  r.width = owidth - std::min(crop_left + crop_right, owidth - 10);
  r.height = oheight - std::min(crop_top + crop_bottom, oheight - 10);
  return true;
}

// ---- scaling (src/scaling.rs:8-32) ---------------------------------------------------------
struct Scaling { float scale; size_t width, height; };
inline Scaling calculate_scaling_total(size_t width, size_t height, size_t maxwidth, size_t maxheight) {
  if (maxwidth == 0 && maxheight == 0) return {1.0f, width, height};
  const float xscale = maxwidth == 0 ? 1.0f : float(width) / float(maxwidth);
  const float yscale = maxheight == 0 ? 1.0f : float(height) / float(maxheight);
  if (yscale <= 1.0f && xscale <= 1.0f) return {1.0f, width, height};
  if (yscale > xscale) return {yscale, f32_to_usize(float(width) / yscale), maxheight};
  return {xscale, maxwidth, f32_to_usize(float(height) / xscale)};
}
inline float demosaic_minscale(int cfa_width) {            // src/ops/demosaic.rs:33-39
  switch (cfa_width) { case 2: return 2.0f; case 6: return 3.0f; case 8: return 2.0f; case 12: return 12.0f; default: return 2.0f; }
}

// ---- normalize_wbs (src/ops/colorspaces.rs:12-27) ------------------------------------------
inline void normalize_wbs(const float vals[4], float out[4]) {
  const float unity = vals[1];
  for (int i = 0; i < 4; ++i) out[i] = !std::isnormal(vals[i]) ? 1.0f : vals[i] / unity;
}


// ---- white-balance temperature helpers (src/color_conversions.rs:277-310, src/ops/colorspaces.rs:59-85) ----
// Host-side, once per edit.  Black-body spectrum against the CIE 1931 2-degree observer, f64 accumulation.
namespace cie {
// CIE 1931 2-degree standard observer colour-matching functions, 380..780 nm in 5 nm steps (public CIE data; the
// reference tabulates the same values at src/color_conversions.rs:193-275)
static constexpr double CIE_XBAR[81] = {
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
static constexpr double CIE_YBAR[81] = {
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
static constexpr double CIE_ZBAR[81] = {
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
}  // namespace cie
inline double powi5(double a) { double r = 1.0; for (int b = 5;;) { if (b & 1) r *= a; b /= 2; if (b == 0) break; a *= a; } return r; }   // f64::powi(5)
inline void temp_to_xyz(float temp, float out[3]) {
  constexpr double C1 = 3.7417717905326694e-16, C2 = 0.014387773457709927;
  double x = 0.0, y = 0.0, z = 0.0;
  for (int i = 0; i < 81; ++i) {
    const double wl = double(380 + 5 * i) / 1.0e9;
    const double power = C1 / (powi5(wl) * (std::exp(C2 / (double(temp) * wl)) - 1.0));
    x += power * cie::CIE_XBAR[i]; y += power * cie::CIE_YBAR[i]; z += power * cie::CIE_ZBAR[i];
  }
  const double mx = std::fmax(std::fmax(x, y), z);
  out[0] = float(x / mx); out[1] = float(y / mx); out[2] = float(z / mx);
}
inline void xyz_to_temp(const float xyz[3], float &temp, float &tint) {
  float mn = 1000.0f, mx = 40000.0f; temp = 0.0f;
  float n[3] = {0.0f, 0.0f, 0.0f};
  while ((mx - mn) > 1.0f) {
    temp = (mx + mn) / 2.0f;
    temp_to_xyz(temp, n);
    if ((n[2] / n[0]) > (xyz[2] / xyz[0])) mx = temp; else mn = temp;
  }
  tint = (n[1] / n[0]) / (xyz[1] / xyz[0]);
}
inline void tolab_set_temp(const float xyz_to_cam[12], float temp, float tint, float wb[4]) {
  float t[3]; temp_to_xyz(temp, t);
  const float xyz[3] = {t[0], t[1] / tint, t[2]};
  float w[4];
  for (int i = 0; i < 4; ++i) {
    w[i] = 0.0f;
    for (int j = 0; j < 3; ++j) w[i] += xyz_to_cam[i * 3 + j] * xyz[j];
    w[i] = 1.0f / w[i];
  }
  normalize_wbs(w, wb);
}
inline void tolab_get_temp(const float cam_to_xyz[12], const float wb[4], float &temp, float &tint) {
  float xyz[3] = {0.0f, 0.0f, 0.0f};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 4; ++j) { const float mul = wb[j]; if (mul > 0.0f) xyz[i] += cam_to_xyz[i * 4 + j] / mul; }
  xyz_to_temp(xyz, temp, tint);
}

// ---- SplineFunc::new (src/ops/curves.rs:68-124) --------------------------------------------
constexpr int kSplineMaxKnots = 66;
struct Spline {
  int npoints = 0, nseg = 0;                 // knots, segments (= c3s.len())
  float px[kSplineMaxKnots], py[kSplineMaxKnots], c1[kSplineMaxKnots], c2[kSplineMaxKnots], c3[kSplineMaxKnots];
  // p: n (x,y) pairs
  bool build(const float *p, int n) {
    if (n < 0 || n > kSplineMaxKnots - 2) return false;
    int np = 0;
    if (n == 0 || (p[0] > 0.0f && p[1] > 0.0f)) { px[np] = 0.0f; py[np] = 0.0f; ++np; }
    for (int i = 0; i < n; ++i) { px[np] = p[2 * i]; py[np] = p[2 * i + 1]; ++np; }
    if (n == 0 || (p[2 * (n - 1)] < 1.0f && p[2 * (n - 1) + 1] < 1.0f)) { px[np] = 1.0f; py[np] = 1.0f; ++np; }
    if (np < 2) return false;                 // reference panics indexing slopes[0]
    npoints = np;
    const int nd = np - 1;
    float dxs[kSplineMaxKnots], slopes[kSplineMaxKnots];
    for (int i = 0; i < nd; ++i) {
      const float dx = px[i + 1] - px[i], dy = py[i + 1] - py[i];
      dxs[i] = dx; slopes[i] = dy / dx;
    }
    int k = 0;
    c1[k++] = slopes[0];
    for (int i = 0; i < nd - 1; ++i) {
      const float m = slopes[i], next = slopes[i + 1];
      if (m * next <= 0.0f) c1[k++] = 0.0f;
      else {
        const float dx = dxs[i], dxnext = dxs[i + 1], common = dx + dxnext;
        c1[k++] = 3.0f * common / ((common + dxnext) / m + (common + dx) / next);
      }
    }
    c1[k++] = slopes[nd - 1];
    nseg = 0;
    for (int i = 0; i < k - 1; ++i) {
      const float a = c1[i], slope = slopes[i], invdx = 1.0f / dxs[i];
      const float common = a + c1[i + 1] - slope - slope;
      c2[nseg] = (slope - a - common) * invdx;
      c3[nseg] = common * invdx * invdx;
      ++nseg;
    }
    return true;
  }
};

// ---- Orientation flips (rawloader; table pinned by src/ops/transform.rs:168-278) -----------
inline void orientation_to_flips(int o, bool &transpose, bool &fx, bool &fy) {
  static const bool t[9][3] = {{0,0,0},{0,1,0},{0,1,1},{0,0,1},{1,0,0},{1,0,1},{1,1,1},{1,1,0},{0,0,0}};
  if (o < 0 || o > 8) o = 8;
  transpose = t[o][0]; fx = t[o][1]; fy = t[o][2];
}
inline int orientation_from_flips(bool transpose, bool fx, bool fy) {
  for (int o = 0; o < 8; ++o) { bool a, b, c; orientation_to_flips(o, a, b, c); if (a == transpose && b == fx && c == fy) return o; }
  return 8;
}
// OpTransform::run's recomposition (src/ops/transform.rs:58-66)
inline int transform_orientation(int rotation, bool fliph, bool flipv) {
  static const int base[4] = {0 /*Normal*/, 5 /*Rotate90*/, 2 /*Rotate180*/, 7 /*Rotate270*/};
  bool t, fx, fy; orientation_to_flips(base[rotation & 3], t, fx, fy);
  return orientation_from_flips(t, fx != fliph, fy != flipv);
}
inline void transform_forward(int rotation, size_t w, size_t h, size_t &ow, size_t &oh) {   // :75-84
  if (rotation == 1 || rotation == 3) { ow = h; oh = w; } else { ow = w; oh = h; }
}

// ---- OpRotateCrop sizing (src/ops/rotatecrop.rs:66-163) ------------------------------------
struct RotateCrop {
  float crop_top = 0, crop_right = 0, crop_bottom = 0, crop_left = 0, rotation = 0;
  float input_ratio = 1.0f;
  bool has_output = false; size_t out_w = 0, out_h = 0;
  static constexpr float kEps = 1.0f / 1000000.0f;
  static constexpr float kFracPi2 = 1.57079632679489661923132169163975144f;

  void reset() { input_ratio = 1.0f; has_output = false; out_w = out_h = 0; }
  bool noop() const {
    return std::fabs(rotation) < kEps && std::fabs(crop_top) < kEps && std::fabs(crop_right) < kEps &&
           std::fabs(crop_bottom) < kEps && std::fabs(crop_left) < kEps;
  }
  float angle() const { return kFracPi2 * (rotation > 1.0f ? 1.0f : rotation); }
  void calc_size(size_t owidth, size_t oheight, bool reverse, size_t &rw, size_t &rh) const {
    rw = owidth; rh = oheight;
    if (noop()) return;
    float width = float(owidth), height = float(oheight);
    if (!(reverse || rotation < kEps)) {
      const float sn = std::sin(angle()), cs = std::cos(angle());
      const float w2 = width * cs + height * sn, h2 = width * sn + height * cs;
      width = w2; height = h2;
    }
    float nwidth, nheight;
    {
      const float ratio = 1.0f - crop_left - crop_right;
      nwidth = reverse ? std::round(width / ratio) : std::round(width * ratio);
      if (ratio < kEps || nwidth < 1.0f) return;
    }
    {
      const float ratio = 1.0f - crop_top - crop_bottom;
      nheight = reverse ? std::round(height / ratio) : std::round(height * ratio);
      if (ratio < kEps || nheight < 1.0f) return;
    }
    if (!(!reverse || rotation < kEps)) {
      const float sn = std::sin(angle()), cs = std::cos(angle());
      const float w2 = std::round(nheight / (sn + (cs / input_ratio)));
      const float h2 = std::round(w2 / input_ratio);
      nwidth = w2; nheight = h2;
    }
    rw = f32_to_usize(nwidth); rh = f32_to_usize(nheight);
  }
  void transform_forward(size_t w, size_t h, size_t &ow, size_t &oh) {
    if (has_output) { ow = out_w; oh = out_h; }
    else { input_ratio = float(w) / float(h); calc_size(w, h, false, ow, oh); }
  }
  void transform_reverse(size_t w, size_t h, size_t &ow, size_t &oh) {
    has_output = true; out_w = w; out_h = h;
    calc_size(w, h, true, ow, oh);
  }
  void rotate_point_reverse(float x, float y, float width, float height, float swidth, float sheight,
                            int64_t &ox, int64_t &oy) const {                       // :97-109
    if (rotation < kEps) { ox = f32_to_isize(x); oy = f32_to_isize(y); return; }
    const float sn = std::sin(angle()), cs = std::cos(angle());
    const float tx = x - (width / 2.0f), ty = y - (height / 2.0f);
    const float nx = tx * cs + ty * sn + (swidth / 2.0f);
    const float ny = -tx * sn + ty * cs + (sheight / 2.0f);
    ox = f32_to_isize(nx); oy = f32_to_isize(ny);
  }
  // The three corner points OpRotateCrop::run hands to OpBuffer::transform (:39-64).
  // Returns false when the op returns its input unchanged.
  bool corners(size_t width, size_t height, int64_t pts[6], size_t &nw, size_t &nh) const {
    if (noop()) return false;
    const float swidth = float(width), sheight = float(height);
    calc_size(width, height, false, nw, nh);
    const float fnw = float(nw), fnh = float(nh);
    const float x = std::floor(swidth * crop_left);
    if (x < 0.0f || x > swidth) return false;
    const float y = std::floor(sheight * crop_top);
    if (y < 0.0f || y > sheight) return false;
    rotate_point_reverse(x, y, fnw, fnh, swidth, sheight, pts[0], pts[1]);
    rotate_point_reverse(x + fnw - 1.0f, y, fnw, fnh, swidth, sheight, pts[2], pts[3]);
    rotate_point_reverse(x, y + fnh - 1.0f, fnw, fnh, swidth, sheight, pts[4], pts[5]);
    return true;
  }
};

}  // namespace ipk
