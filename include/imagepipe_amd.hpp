// imagepipe_amd.hpp -- C++ mirror of the reference's operator surface over the C ABI (imagepipe_amd.h).
//
// The reference is Rust; with no Rust toolchain in this image the host side above the C ABI is C++.  The types keep
// the reference's names, fields and behaviour (citations: file:line in pedrocr/imagepipe 0.5.0):
//   OpBuffer (src/buffer.rs:5-32), ImageSource::{Raw,Other}, PipelineSettings/PipelineGlobals (src/pipeline.rs:110-151),
//   trait ImageOp (src/pipeline.rs:82-108), the eight ops (src/ops/*.rs), PipelineOps and Pipeline (src/pipeline.rs:153-470).
// Buffers live in HBM (DeviceArray); `run` returns either the SAME shared_ptr (the reference's "return the input Arc")
// or a fresh buffer.  Header-only; link with -limagepipe_amd.  Errors throw imagepipe::Error where the reference panics.
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <algorithm>
#include <stdexcept>
#include <tuple>
#include <string>
#include <utility>
#include <vector>

#include "imagepipe_amd.h"

namespace imagepipe {

struct Error : std::runtime_error { using std::runtime_error::runtime_error; };
inline int check(int rc, const char *what) {
  if (rc < 0) throw Error(std::string(what) + ": " + ipk_last_error());
  return rc;
}

// RAII device allocation
class DeviceArray {
 public:
  DeviceArray() = default;
  explicit DeviceArray(size_t bytes) : bytes_(bytes) { check(ipk_malloc(&p_, bytes), "ipk_malloc"); }
  DeviceArray(const void *host, size_t bytes) : DeviceArray(bytes) { check(ipk_memcpy_h2d(p_, host, bytes, nullptr), "h2d"); check(ipk_stream_sync(nullptr), "sync"); }
  DeviceArray(const DeviceArray &) = delete;
  DeviceArray &operator=(const DeviceArray &) = delete;
  DeviceArray(DeviceArray &&o) noexcept : p_(o.p_), bytes_(o.bytes_) { o.p_ = nullptr; o.bytes_ = 0; }
  DeviceArray &operator=(DeviceArray &&o) noexcept { if (this != &o) { if (p_) ipk_free(p_); p_ = o.p_; bytes_ = o.bytes_; o.p_ = nullptr; o.bytes_ = 0; } return *this; }
  ~DeviceArray() { if (p_) ipk_free(p_); }
  void *get() const { return p_; }
  size_t bytes() const { return bytes_; }
  void download(void *host) const { check(ipk_memcpy_d2h(host, p_, bytes_, nullptr), "d2h"); check(ipk_stream_sync(nullptr), "sync"); }
 private:
  void *p_ = nullptr; size_t bytes_ = 0;
};

// src/buffer.rs:5-32
struct OpBuffer {
  size_t width = 0, height = 0, colors = 0;
  bool monochrome = false;
  DeviceArray data;                                    // f32, row-major, channel-interleaved
  OpBuffer() = default;
  OpBuffer(size_t w, size_t h, size_t c, bool mono) : width(w), height(h), colors(c), monochrome(mono), data(w * h * c * sizeof(float)) {}
  static std::shared_ptr<OpBuffer> from_host(const std::vector<float> &v, size_t w, size_t h, size_t c, bool mono = false) {
    auto b = std::make_shared<OpBuffer>(); b->width = w; b->height = h; b->colors = c; b->monochrome = mono;
    b->data = DeviceArray(v.data(), v.size() * sizeof(float)); return b;
  }
  std::vector<float> to_host() const { std::vector<float> v(width * height * colors); data.download(v.data()); return v; }
  float *ptr() const { return static_cast<float *>(data.get()); }
};
using Buf = std::shared_ptr<OpBuffer>;

// The fields of rawloader::RawImage / image::DynamicImage the hot path reads
struct ImageSource {
  enum Kind { Raw, Other } kind = Raw;
  size_t width = 0, height = 0;
  // Raw
  int cpp = 1; bool is_float = false;
  std::string cfa;                                     // uncropped pattern ("" = not a CFA image)
  size_t crops[4] = {0, 0, 0, 0};                      // top, right, bottom, left
  float blacklevels[4] = {0, 0, 0, 0}, whitelevels[4] = {65535, 65535, 65535, 65535};
  float wb_coeffs[4] = {1.0f, 1.0f, 1.0f, NAN};
  bool has_neutralwb = false; float neutralwb[4] = {1.0f, 1.0f, 1.0f, NAN};   // RawImage::neutralwb() (rawloader, computed by the caller): the
                                                       // fallback OpToLab::new takes when the as-shot coefficients are not normal (colorspaces.rs:33-39)
  float cam_to_xyz_normalized[12] = {0.4124564f, 0.3575761f, 0.1804375f, 0, 0.2126729f, 0.7151522f, 0.0721750f, 0, 0.0193339f, 0.1191920f, 0.9503041f, 0};
  float cam_to_xyz[12] = {0.4124564f, 0.3575761f, 0.1804375f, 0, 0.2126729f, 0.7151522f, 0.0721750f, 0, 0.0193339f, 0.1191920f, 0.9503041f, 0};
  bool has_xyz_to_cam = false; float xyz_to_cam[12] = {};   // [[f32;3];4]; OpToLab::set_temp only (default XYZ_D65_34)
  int orientation = IPK_OR_NORMAL;
  // Other
  int bits = 8;
  DeviceArray data;                                    // sensor / raster samples on the device
  std::string cropped_cfa() const {
    if (cfa.empty()) return "";
    char out[200]; check(ipk_cfa_shift(cfa.c_str(), (int)crops[3], (int)crops[0], out), "ipk_cfa_shift"); return out;
  }
};

// src/pipeline.rs:110-151
struct PipelineSettings { size_t maxwidth = 0, maxheight = 0, demosaic_width = 0, demosaic_height = 0; bool linear = false, use_fastpath = true; };
struct PipelineGlobals { ImageSource image; PipelineSettings settings; };

// src/pipeline.rs:82-108
struct ImageOp {
  virtual ~ImageOp() = default;
  virtual const char *name() const = 0;
  virtual Buf run(const PipelineGlobals &pipeline, Buf buf) const = 0;
  virtual std::pair<size_t, size_t> transform_forward(size_t w, size_t h) { return {w, h}; }
  virtual std::pair<size_t, size_t> transform_reverse(size_t w, size_t h) { return {w, h}; }
  virtual void reset() {}
};

// src/ops/gofloat.rs
struct OpGoFloat : ImageOp {
  size_t crop_top = 0, crop_right = 0, crop_bottom = 0, crop_left = 0; bool is_cfa = false;
  float blacklevels[4] = {0, 0, 0, 0}, whitelevels[4] = {0, 0, 0, 0};
  explicit OpGoFloat(const ImageSource &img) {
    if (img.kind == ImageSource::Raw) {
      crop_top = img.crops[0]; crop_right = img.crops[1]; crop_bottom = img.crops[2]; crop_left = img.crops[3];
      is_cfa = !img.cfa.empty();
      std::memcpy(blacklevels, img.blacklevels, sizeof(blacklevels)); std::memcpy(whitelevels, img.whitelevels, sizeof(whitelevels));
    }
  }
  const char *name() const override { return "gofloat"; }
  void size_image(size_t ow, size_t oh, size_t out[4]) const { check(ipk_size_image(crop_top, crop_right, crop_bottom, crop_left, ow, oh, out), "size_image"); }
  std::pair<size_t, size_t> transform_forward(size_t w, size_t h) override { size_t o[4]; size_image(w, h, o); return {o[2], o[3]}; }
  Buf run(const PipelineGlobals &p, Buf) const override {
    const ImageSource &img = p.image;
    size_t o[4]; size_image(img.width, img.height, o);
    const size_t x = o[0], y = o[1], w = o[2], h = o[3];
    if (img.kind == ImageSource::Raw) {
      if (img.cpp == 1 && !is_cfa) {
        auto out = std::make_shared<OpBuffer>(w, h, 4, true);
        check(img.is_float ? ipk_gofloat_mono_f32((const float *)img.data.get(), img.width, x, y, w, h, blacklevels[0], whitelevels[0], out->ptr(), nullptr)
                           : ipk_gofloat_mono_u16((const uint16_t *)img.data.get(), img.width, x, y, w, h, blacklevels[0], whitelevels[0], out->ptr(), nullptr), "gofloat");
        return out;
      } else if (img.cpp == 3) {
        auto out = std::make_shared<OpBuffer>(w, h, 4, false);
        check(img.is_float ? ipk_gofloat_rgb_f32((const float *)img.data.get(), img.width, x, y, w, h, blacklevels, whitelevels, out->ptr(), nullptr)
                           : ipk_gofloat_rgb_u16((const uint16_t *)img.data.get(), img.width, x, y, w, h, blacklevels, whitelevels, out->ptr(), nullptr), "gofloat");
        return out;
      }
      auto out = std::make_shared<OpBuffer>(w, h, (size_t)img.cpp, false);
      check(img.is_float ? ipk_gofloat_cfa_f32((const float *)img.data.get(), img.width, x, y, w, h, blacklevels[0], whitelevels[0], out->ptr(), nullptr)
                         : ipk_gofloat_cfa_u16((const uint16_t *)img.data.get(), img.width, x, y, w, h, blacklevels[0], whitelevels[0], out->ptr(), nullptr), "gofloat");
      return out;
    }
    auto out = std::make_shared<OpBuffer>(w, h, 4, false);
    check(img.bits == 8 ? ipk_gofloat_other_u8((const uint8_t *)img.data.get(), img.width, x, y, w, h, out->ptr(), nullptr)
                        : ipk_gofloat_other_u16((const uint16_t *)img.data.get(), img.width, x, y, w, h, out->ptr(), nullptr), "gofloat");
    return out;
  }
};

// src/ops/demosaic.rs
struct OpDemosaic : ImageOp {
  std::string cfa;
  explicit OpDemosaic(const ImageSource &img) : cfa(img.kind == ImageSource::Raw ? img.cropped_cfa() : "") {}
  const char *name() const override { return "demosaic"; }
  Buf run(const PipelineGlobals &p, Buf buf) const override {
    const size_t nw = p.settings.demosaic_width, nh = p.settings.demosaic_height;
    auto out = std::make_shared<OpBuffer>();
    out->data = DeviceArray(std::max(buf->width * buf->height, nw * nh) * 4 * sizeof(float));
    size_t ow, oh;
    const int rc = check(ipk_demosaic_run(buf->ptr(), buf->width, buf->height, buf->colors, cfa.c_str(), nw, nh, out->ptr(), &ow, &oh, nullptr), "demosaic");
    if (rc == IPK_NOOP) return buf;
    out->width = ow; out->height = oh; out->colors = 4; out->monochrome = buf->monochrome;
    return out;
  }
};

// src/ops/rotatecrop.rs
struct OpRotateCrop : ImageOp {
  float crop_top = 0, crop_right = 0, crop_bottom = 0, crop_left = 0, rotation = 0;
  float input_ratio = 1.0f; bool has_output = false; std::pair<size_t, size_t> output_size{0, 0};
  const char *name() const override { return "rotatecrop"; }
  void reset() override { input_ratio = 1.0f; has_output = false; }
  std::pair<size_t, size_t> calc_size(size_t w, size_t h, bool reverse) const {
    const float p[5] = {crop_top, crop_right, crop_bottom, crop_left, rotation}; size_t ow, oh;
    check(ipk_rotatecrop_calc_size(p, input_ratio, w, h, reverse, &ow, &oh), "calc_size"); return {ow, oh};
  }
  std::pair<size_t, size_t> transform_forward(size_t w, size_t h) override {
    if (has_output) return output_size;
    input_ratio = float(w) / float(h); return calc_size(w, h, false);
  }
  std::pair<size_t, size_t> transform_reverse(size_t w, size_t h) override { has_output = true; output_size = {w, h}; return calc_size(w, h, true); }
  Buf run(const PipelineGlobals &, Buf buf) const override {
    const float p[5] = {crop_top, crop_right, crop_bottom, crop_left, rotation}; size_t ow, oh;
    if (check(ipk_rotatecrop(buf->ptr(), buf->width, buf->height, buf->colors, p, nullptr, &ow, &oh, nullptr), "rotatecrop") == IPK_NOOP) return buf;
    auto out = std::make_shared<OpBuffer>(ow, oh, buf->colors, buf->monochrome);
    check(ipk_rotatecrop(buf->ptr(), buf->width, buf->height, buf->colors, p, out->ptr(), &ow, &oh, nullptr), "rotatecrop");
    return out;
  }
};

// src/ops/colorspaces.rs
struct OpToLab : ImageOp {
  float cam_to_xyz[12]; float cam_to_xyz_normalized[12]; float xyz_to_cam[12]; float wb_coeffs[4];
  explicit OpToLab(const ImageSource &img) {
    check(ipk_const_matrix(3, xyz_to_cam), "const_matrix");
    if (img.kind == ImageSource::Raw) {
      std::memcpy(cam_to_xyz, img.cam_to_xyz, sizeof(cam_to_xyz));
      std::memcpy(cam_to_xyz_normalized, img.cam_to_xyz_normalized, sizeof(cam_to_xyz_normalized));
      if (img.has_xyz_to_cam) std::memcpy(xyz_to_cam, img.xyz_to_cam, sizeof(xyz_to_cam));
      // colorspaces.rs:33-41: normalize_wbs(wb_coeffs), or normalize_wbs(neutralwb()) when any of wb[0..2] is not normal.  neutralwb()
      // is rawloader's (camera metadata, absent here): the caller supplies it, and a raw source that needs it without one fails loudly
      const bool normal = std::isnormal(img.wb_coeffs[0]) && std::isnormal(img.wb_coeffs[1]) && std::isnormal(img.wb_coeffs[2]);
      if (!normal && !img.has_neutralwb) throw Error("OpToLab: as-shot wb_coeffs are not normal and the ImageSource carries no neutralwb");
      check(ipk_normalize_wbs(normal ? img.wb_coeffs : img.neutralwb, wb_coeffs), "normalize_wbs");
    } else {
      check(ipk_const_matrix(2, cam_to_xyz), "const_matrix"); std::memcpy(cam_to_xyz_normalized, cam_to_xyz, sizeof(cam_to_xyz));
      const float w[4] = {1, 1, 1, 0}; std::memcpy(wb_coeffs, w, sizeof(w));
    }
  }
  void set_temp(float temp, float tint) { check(ipk_tolab_set_temp(xyz_to_cam, temp, tint, wb_coeffs), "set_temp"); }           // :59-70
  std::pair<float, float> get_temp() const { float t, ti; check(ipk_tolab_get_temp(cam_to_xyz, wb_coeffs, &t, &ti), "get_temp"); return {t, ti}; }   // :72-84
  const char *name() const override { return "to_lab"; }
  Buf run(const PipelineGlobals &, Buf buf) const override {
    auto out = std::make_shared<OpBuffer>(buf->width, buf->height, 3, buf->monochrome);
    check(ipk_tolab(buf->ptr(), buf->width, buf->height, buf->monochrome, wb_coeffs, cam_to_xyz_normalized, out->ptr(), nullptr), "tolab");
    return out;
  }
};
struct OpFromLab : ImageOp {
  const char *name() const override { return "from_lab"; }
  Buf run(const PipelineGlobals &, Buf buf) const override {
    auto out = std::make_shared<OpBuffer>(buf->width, buf->height, 3, buf->monochrome);
    check(ipk_fromlab(buf->ptr(), buf->width, buf->height, out->ptr(), nullptr), "fromlab"); return out;
  }
};

// src/ops/curves.rs
struct OpBaseCurve : ImageOp {
  float exposure = 0.0f; std::vector<std::pair<float, float>> points;
  explicit OpBaseCurve(const ImageSource &img) { if (img.kind == ImageSource::Raw) points = {{0.50f, 0.60f}}; }
  const char *name() const override { return "basecurve"; }
  Buf run(const PipelineGlobals &, Buf buf) const override {
    std::vector<float> p; for (auto &q : points) { p.push_back(q.first); p.push_back(q.second); }
    if (p.empty()) p.resize(2);
    auto out = std::make_shared<OpBuffer>(buf->width, buf->height, 3, buf->monochrome);
    return check(ipk_basecurve(buf->ptr(), buf->width, buf->height, exposure, p.data(), (int)points.size(), out->ptr(), nullptr), "basecurve") == IPK_NOOP ? buf : out;
  }
};

// src/ops/gamma.rs
struct OpGamma : ImageOp {
  const char *name() const override { return "gamma"; }
  Buf run(const PipelineGlobals &p, Buf buf) const override {
    auto out = std::make_shared<OpBuffer>(buf->width, buf->height, buf->colors, buf->monochrome);
    return check(ipk_gamma(buf->ptr(), buf->width, buf->height, buf->colors, p.settings.linear, out->ptr(), nullptr), "gamma") == IPK_NOOP ? buf : out;
  }
};

// src/ops/transform.rs
struct OpTransform : ImageOp {
  int rotation = IPK_ROT_NORMAL; bool fliph = false, flipv = false;
  explicit OpTransform(const ImageSource &img) {
    if (img.kind != ImageSource::Raw) return;
    switch (img.orientation) {                            // transform.rs:25-36
      case IPK_OR_VFLIP: flipv = true; break; case IPK_OR_HFLIP: fliph = true; break;
      case IPK_OR_ROT180: rotation = IPK_ROT_180; break; case IPK_OR_TRANSPOSE: rotation = IPK_ROT_90; flipv = true; break;
      case IPK_OR_ROT90: rotation = IPK_ROT_90; break; case IPK_OR_ROT270: rotation = IPK_ROT_270; break;
      case IPK_OR_TRANSVERSE: rotation = IPK_ROT_270; fliph = true; break; default: break;
    }
  }
  const char *name() const override { return "transform"; }
  std::pair<size_t, size_t> transform_forward(size_t w, size_t h) override { return (rotation == IPK_ROT_90 || rotation == IPK_ROT_270) ? std::make_pair(h, w) : std::make_pair(w, h); }
  std::pair<size_t, size_t> transform_reverse(size_t w, size_t h) override { return transform_forward(w, h); }
  Buf run(const PipelineGlobals &, Buf buf) const override {
    auto out = std::make_shared<OpBuffer>(buf->width, buf->height, 3, buf->monochrome); size_t ow, oh;
    if (check(ipk_transform(buf->ptr(), buf->width, buf->height, rotation, fliph, flipv, out->ptr(), &ow, &oh, nullptr), "transform") == IPK_NOOP) return buf;
    out->width = ow; out->height = oh; return out;
  }
};

// src/pipeline.rs:153-179
struct PipelineOps {
  OpGoFloat gofloat; OpDemosaic demosaic; OpRotateCrop rotatecrop; OpToLab tolab; OpBaseCurve basecurve; OpFromLab fromlab; OpGamma gamma; OpTransform transform;
  explicit PipelineOps(const ImageSource &img) : gofloat(img), demosaic(img), tolab(img), basecurve(img), transform(img) {}
  std::vector<ImageOp *> all() { return {&gofloat, &demosaic, &rotatecrop, &tolab, &basecurve, &fromlab, &gamma, &transform}; }
};

// 8/16-bit results (src/pipeline.rs:21-41)
struct SRGBImage { size_t width, height; std::vector<uint8_t> data; };
struct SRGBImage16 { size_t width, height; std::vector<uint16_t> data; };

// PipelineCache = MultiCache<BufHash, OpBuffer> (src/pipeline.rs:43); Pipeline::new_cache(size) (:258-260)
class PipelineCache {
 public:
  explicit PipelineCache(size_t max_bytes) { check(ipk_cache_new(max_bytes, &c_), "cache_new"); }
  ~PipelineCache() { ipk_cache_free(c_); }
  PipelineCache(const PipelineCache &) = delete; PipelineCache &operator=(const PipelineCache &) = delete;
  ipk_cache *get() const { return c_; }
  bool contains(const std::array<uint8_t, 32> &key) const { return ipk_cache_contains(c_, key.data()) == 1; }
  size_t bytes() const { size_t b = 0; ipk_cache_stats(c_, &b, nullptr, nullptr, nullptr, nullptr); return b; }
 private:
  ipk_cache *c_ = nullptr;
};

// src/pipeline.rs:246-470
class Pipeline {
 public:
  PipelineGlobals globals; PipelineOps ops;
  bool allow_fused = true, last_used_fused = false;
  int schedule = IPK_SCHED_AUTO;                        // ipk_pipeline_desc.schedule: how a fused launch shares a frame's rows out (results unaffected)
  int last_ops_run = 0xFF;                 // bit i: op i executed in the last run (0 = served from the cache)
  uint64_t source_id = 0;                  // identifies the frame inside a shared PipelineCache (extension, see the C header)
  static PipelineCache new_cache(size_t size) { return PipelineCache(size); }
  static Pipeline new_from_source(ImageSource img, int device = 0) { check(ipk_init(device), "ipk_init"); return Pipeline(std::move(img)); }

  // Pipeline::run as the reference writes it: reset, negotiate sizes, then each op's run in order (pipeline.rs:311-375)
  Buf run_ops() {
    negotiate();
    Buf buf;
    for (ImageOp *op : ops.all()) buf = op->run(globals, buf);
    return buf;
  }
  // The same through the C driver, which fuses gofloat..gamma into one kernel when every op allows it
  Buf run(const PipelineCache *cache = nullptr) {
    size_t fw, fh; std::tie(fw, fh) = final_size();
    auto out = std::make_shared<OpBuffer>(fw, fh, 3, false);
    drive(cache, IPK_OUT_F32, out->ptr());
    check(ipk_stream_sync(nullptr), "sync"); return out;
  }
  SRGBImage output_8bit(const PipelineCache *cache = nullptr) {      // slow path, pipeline.rs:404-421
    size_t fw, fh; std::tie(fw, fh) = final_size();
    DeviceArray o(fw * fh * 3); drive(cache, IPK_OUT_U8, o.get());
    SRGBImage img{fw, fh, std::vector<uint8_t>(fw * fh * 3)}; o.download(img.data.data()); return img;
  }
  SRGBImage16 output_16bit(const PipelineCache *cache = nullptr) {   // pipeline.rs:451-468
    size_t fw, fh; std::tie(fw, fh) = final_size();
    DeviceArray o(fw * fh * 6); drive(cache, IPK_OUT_U16, o.get());
    SRGBImage16 img{fw, fh, std::vector<uint16_t>(fw * fh * 3)}; o.download(img.data.data()); return img;
  }
  // ophashes of pipeline.rs:342-361
  std::vector<std::array<uint8_t, 32>> hashes(int out_type = IPK_OUT_F32) const {
    ipk_pipeline_desc d = desc(); uint8_t raw[256];
    check(ipk_pipeline_hashes(&d, out_type, source_id, raw), "hashes");
    std::vector<std::array<uint8_t, 32>> v(8);
    for (int i = 0; i < 8; ++i) std::memcpy(v[i].data(), raw + 32 * i, 32);
    return v;
  }
  // size negotiation over the op objects (pipeline.rs:314-338)
  std::pair<size_t, size_t> negotiate() {
    for (ImageOp *op : ops.all()) op->reset();
    size_t w = globals.image.width, h = globals.image.height;
    for (ImageOp *op : ops.all()) std::tie(w, h) = op->transform_forward(w, h);
    float s; size_t nw, nh; ipk_calculate_scaling_total(w, h, globals.settings.maxwidth, globals.settings.maxheight, &s, &nw, &nh);
    w = nw; h = nh; const auto fin = std::make_pair(w, h);
    auto all = ops.all();
    for (auto it = all.rbegin(); it != all.rend(); ++it) std::tie(w, h) = (*it)->transform_reverse(w, h);
    globals.settings.demosaic_width = w; globals.settings.demosaic_height = h;
    return fin;
  }
  ipk_pipeline_desc desc() const {
    ipk_pipeline_desc d; std::memset(&d, 0, sizeof(d)); d.struct_size = (uint32_t)sizeof(d);
    const ImageSource &img = globals.image;
    d.src_type = img.kind == ImageSource::Raw ? (img.is_float ? IPK_SRC_F32 : IPK_SRC_U16) : (img.bits == 8 ? IPK_SRC_RGB8 : IPK_SRC_RGB16);
    d.width = img.width; d.height = img.height; d.cpp = img.kind == ImageSource::Raw ? img.cpp : 3; d.is_cfa = ops.gofloat.is_cfa;
    std::strncpy(d.cfa, ops.demosaic.cfa.c_str(), sizeof(d.cfa) - 1);
    d.crop_top = ops.gofloat.crop_top; d.crop_right = ops.gofloat.crop_right; d.crop_bottom = ops.gofloat.crop_bottom; d.crop_left = ops.gofloat.crop_left;
    std::memcpy(d.blacklevels, ops.gofloat.blacklevels, sizeof(d.blacklevels)); std::memcpy(d.whitelevels, ops.gofloat.whitelevels, sizeof(d.whitelevels));
    const float rc[5] = {ops.rotatecrop.crop_top, ops.rotatecrop.crop_right, ops.rotatecrop.crop_bottom, ops.rotatecrop.crop_left, ops.rotatecrop.rotation};
    std::memcpy(d.rotatecrop, rc, sizeof(rc));
    std::memcpy(d.cam_to_xyz_normalized, ops.tolab.cam_to_xyz_normalized, sizeof(d.cam_to_xyz_normalized)); std::memcpy(d.wb_coeffs, ops.tolab.wb_coeffs, sizeof(d.wb_coeffs));
    d.exposure = ops.basecurve.exposure; d.npoints = (int)ops.basecurve.points.size();
    for (size_t i = 0; i < ops.basecurve.points.size() && i < 64; ++i) { d.points[2 * i] = ops.basecurve.points[i].first; d.points[2 * i + 1] = ops.basecurve.points[i].second; }
    d.rotation = ops.transform.rotation; d.fliph = ops.transform.fliph; d.flipv = ops.transform.flipv;
    d.maxwidth = globals.settings.maxwidth; d.maxheight = globals.settings.maxheight; d.linear = globals.settings.linear; d.allow_fused = allow_fused; d.use_fastpath = globals.settings.use_fastpath;
    d.schedule = schedule;
    return d;
  }
 private:
  explicit Pipeline(ImageSource img) : globals{std::move(img), PipelineSettings()}, ops(globals.image) {}
  std::pair<size_t, size_t> final_size() const {
    ipk_pipeline_desc d = desc(); size_t dw, dh, fw, fh; check(ipk_pipeline_sizes(&d, &dw, &dh, &fw, &fh), "sizes"); return {fw, fh};
  }
  void drive(const PipelineCache *cache, int out_type, void *dst) {
    ipk_pipeline_desc d = desc(); int fused = 0;
    if (cache) check(ipk_pipeline_run_cached(&d, globals.image.data.get(), source_id, cache->get(), out_type, dst, &last_ops_run, &fused, nullptr), "pipeline_run_cached");
    else { check(ipk_pipeline_run(&d, globals.image.data.get(), dst, out_type, &fused, nullptr), "pipeline_run"); last_ops_run = 0xFF; }
    last_used_fused = fused != 0;
  }
};

// ---- several devices from ONE process (imagepipe_amd.h, "Several devices from ONE process") --------------------------------------------------
// The reference is one process whose callers loop Pipeline::run over the frames of a shoot (src/lib.rs:21-26, src/pipeline.rs:246-249).
// CurrentContext: scope guard -- inside it every call of this header (and of the C ABI) on THIS thread runs on the given context / device.
class CurrentContext {
 public:
  explicit CurrentContext(ipk_ctx *c) : prev_(ipk_ctx_current()) { check(ipk_ctx_make_current(c), "ipk_ctx_make_current"); }
  CurrentContext(const CurrentContext &) = delete;
  CurrentContext &operator=(const CurrentContext &) = delete;
  ~CurrentContext() { (void)ipk_ctx_make_current(prev_); }
 private:
  ipk_ctx *prev_;
};
// DeviceSet: one context per listed device ordinal (none listed: every visible device); frame i of a batch goes to member i % size().
class DeviceSet {
 public:
  explicit DeviceSet(const std::vector<int> &devices = {}) { check(ipk_init_devices(devices.empty() ? nullptr : devices.data(), (int)devices.size()), "ipk_init_devices"); }
  size_t size() const { return (size_t)ipk_device_set_size(); }
  ipk_ctx *member(size_t i) const { return ipk_device_ctx((int)i); }
  // a caller looping output_8bit / run over a shoot of same-shaped frames in HOST memory (page-locked: ipk_host_alloc); synchronous
  static void develop_host(const Pipeline &p, const std::vector<const void *> &raws, const std::vector<void *> &outs, int out_type);
  // frames resident on their member's device (frame i on member i % size()); enqueues on every device, sync() waits for all
  static void develop_device(const Pipeline &p, const std::vector<const void *> &raws, const std::vector<void *> &outs, int out_type);
  static void sync() { check(ipk_devices_sync(), "ipk_devices_sync"); }
};
inline void DeviceSet::develop_host(const Pipeline &p, const std::vector<const void *> &raws, const std::vector<void *> &outs, int out_type) {
  if (raws.size() != outs.size()) throw Error("develop_host: as many outputs as frames");
  ipk_pipeline_desc d = p.desc();
  check(ipk_host_pipeline_run_batch_multi(&d, raws.data(), outs.data(), raws.size(), out_type, nullptr), "ipk_host_pipeline_run_batch_multi");
}
inline void DeviceSet::develop_device(const Pipeline &p, const std::vector<const void *> &raws, const std::vector<void *> &outs, int out_type) {
  if (raws.size() != outs.size()) throw Error("develop_device: as many outputs as frames");
  ipk_pipeline_desc d = p.desc();
  check(ipk_pipeline_run_batch_multi(&d, raws.data(), outs.data(), raws.size(), out_type, nullptr), "ipk_pipeline_run_batch_multi");
}

}  // namespace imagepipe
