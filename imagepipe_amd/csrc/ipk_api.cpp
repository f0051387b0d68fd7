// ipk_api.cpp -- the C ABI (include/imagepipe_amd.h): context, host-side parameter preparation,
// the Pipeline::run driver and the host-pointer wrappers.  No device code here; kernels are
// reached through ipk_launch.hpp.  There is deliberately no CPU implementation of any pixel
// loop in this file: without a GPU every compute entry point fails with IPK_ERR_NO_DEVICE.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstddef>
#include <cstring>
#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/imagepipe_amd.h"
#include "ipk_hash.hpp"
#include "ipk_host.hpp"
#include "ipk_internal.hpp"
#include "ipk_launch.hpp"

namespace {

thread_local char g_err[512] = "";
int fail(int code, const char *fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
  return code;
}
#define HIPCHK(expr)                                                                               \
  do { hipError_t e_ = (expr);                                                                     \
       if (e_ != hipSuccess) return fail(IPK_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); } while (0)

struct DevCfa { uint32_t *lookups = nullptr; uint8_t *cfa48 = nullptr; float *gen_cells = nullptr; int gen_pw = 0, gen_ph = 0; };   // gen_cells: null for filters with a fourth colour

}  // namespace

namespace {
// The streams, events and device slots of the host-pointer pipeline driver.  One set per context, grown on demand and kept
// between calls (a fresh hipMalloc of two input and two output slots costs more than moving a frame), released with the context.
struct HostLanes {
  static constexpr int kSlots = 2;
  std::mutex mu;                                  // one host-pointer pipeline call at a time per context
  hipStream_t up = nullptr, run = nullptr, down = nullptr;
  hipEvent_t up_done[kSlots] = {}, run_done[kSlots] = {}, down_done[kSlots] = {};
  void *in[kSlots] = {}, *out[kSlots] = {};
  size_t in_cap = 0, out_cap = 0;
  bool made = false;
  int ensure(size_t in_bytes, size_t out_bytes);
  void release();
};
}  // namespace

// One library context: a device binding plus everything the entry points keep on that device -- the lookup tables, the CFA table cache, the
// stream-ordered scratch pool, the task-queue heads of the row-walking kernels, the host-pointer driver's lanes.  A process may hold several
// (one per GPU for a batch dealt frame i -> device i mod N from ONE process -- the reference is one process, src/lib.rs:21-26 -- or several on one
// GPU); every entry point works on the calling thread's CURRENT context (ipk_ctx_make_current), which defaults to the one ipk_init made.
// Contexts are never deallocated before process exit (ipk_ctx_destroy releases the device resources and marks the object dead), so a stale
// handle on another thread fails with IPK_ERR_NOT_INIT instead of touching freed memory.
struct ipk_ctx {
  bool ready = false;
  int libm_matches = -1; size_t libm_mismatches = 0;      // init-time comparison of the host's cbrtf with the device routine
  int device = -1;
  int num_cus = 0;
  void *lut_pairs[3] = {nullptr, nullptr, nullptr};      // device, 8192 x {v, dv}
  void *lut_plain[3] = {nullptr, nullptr, nullptr};      // device, 8193 floats
  void *lut_q8 = nullptr;                                // device, 8192 x {k, threshold}: OpGamma + output8bit as one step lookup
  std::map<std::string, DevCfa> cfa_cache;
  std::map<std::string, float *> rot_cells;              // generic-CFA cell records laid out for a rotated space (pattern, orientation, frame phase)
  // stream-ordered scratch pool for the staged pipeline's intermediate OpBuffers
  // last: the stream its most recent user enqueued on; clean: the device has been drained since the block came back
  struct Block { void *p; size_t bytes; bool busy; hipStream_t last; bool clean; };
  std::vector<Block> pool;
  std::mutex mu;
  ipk::TaskQueues *queues = nullptr;
  HostLanes lanes;
  hipStream_t multi_stream = nullptr;                    // the stream ipk_pipeline_run_batch_multi enqueues on for this context
};

namespace {
typedef ipk_ctx Context;
// host-side state every context shares (immutable once built)
struct HostTables {
  std::mutex mu;
  std::vector<float> lut_host[3];
  float xyz_d65_33[9];
} g_host;
std::mutex g_reg_mu;                                      // registry, default context, device set
std::vector<std::unique_ptr<Context>> g_registry;         // every context ever created (dead ones stay: see ipk_ctx)
std::atomic<Context *> g_default{nullptr};                             // ipk_init's context: current on every thread that has not chosen another
std::vector<Context *> g_devset;                          // ipk_init_devices: one context per listed device
thread_local Context *t_current = nullptr;
Context g_none;                                           // ready == false: what cx() answers before any context exists
Context &cx() {
  Context *c = t_current ? t_current : g_default.load(std::memory_order_acquire);
  return c ? *c : g_none;
}

void build_host_luts() {
  std::lock_guard<std::mutex> lk(g_host.mu);
  if (!g_host.lut_host[0].empty()) return;
  for (int i = 0; i < 3; ++i) g_host.lut_host[i] = ipk::build_lut(static_cast<ipk::LutId>(i));
  const ipk::Mat33 inv = ipk::inverse(ipk::srgb_d65_33());
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) g_host.xyz_d65_33[r * 3 + c] = inv.m[r][c];
}

// HIP's current device is per host thread: a caller on another thread than the context's creator (a Rayon worker), or one that switched
// contexts, must land on the context's GPU.  hipGetDevice is a thread-local read; asking every time (instead of remembering what this
// library set last) stays correct when the host application calls hipSetDevice itself between two calls.
int bind_device(int device) {
  int now = -1;
  if (hipGetDevice(&now) != hipSuccess || now != device) HIPCHK(hipSetDevice(device));
  return IPK_OK;
}
int require_init() {
  Context &c = cx();
  if (!c.ready) return fail(IPK_ERR_NOT_INIT, "no live context on this thread: ipk_init() / ipk_ctx_create() has not succeeded (no MI355X/HIP device bound) or the current context was destroyed; there is no CPU fallback");
  return bind_device(c.device);
}
#define REQUIRE_INIT() do { int rc_ = require_init(); if (rc_) return rc_; } while (0)
}  // namespace
namespace ipk {
int internal_fail(int code, const char *fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
  return code;
}
int internal_require_init() { return require_init(); }
int internal_device() { return cx().device; }
}  // namespace ipk
namespace {

hipStream_t S(void *stream) { return reinterpret_cast<hipStream_t>(stream); }

bool dims_ok(size_t w, size_t h) { return w >= 1 && h >= 1 && w < (1ull << 31) && h < (1ull << 31); }

// 16-letter patterns without a stated shape are refused, not guessed: rawloader's tile shape for them is unverified (ipk_host.hpp Cfa::parse)
int cfa_fail(const char *pat) {
  if (ipk::Cfa::unpinned_length(pat))
    return fail(IPK_ERR_UNSUPPORTED, "16-letter CFA pattern \"%s\": state the tile's shape (\"8x2:...\" / \"2x8:...\", or cfa_width / cfa_height); it is not guessed", pat);
  return fail(IPK_ERR_INVALID, "invalid CFA pattern \"%s\"", pat ? pat : "(null)");
}
// A descriptor's pattern in its CANONICAL spelling, with the shape its cfa_width / cfa_height fields state folded in (the fields are zeroed): plain
// letters when the tile's shape is the one the letter count implies (4 -> 2x2, 36 -> 6x6, 144 -> 12x12), "WxH:letters" otherwise.  A caller that
// fills the fields from its CFA object, one that writes a redundant "2x2:" prefix and one that passes the plain pattern therefore reach the same
// device tables and -- through hash_chain -- the same ipk_pipeline_hashes / cache keys, which stay the hash of the reference's plain pattern string
// (src/ops/demosaic.rs:13).  False: the fields contradict a prefix already in the string, or the result does not fit the buffer.
// Every descriptor that crosses the ABI is first TAKEN: struct_size bytes of the caller's object into a zeroed struct of this library's layout
// (include/imagepipe_amd.h, "Descriptor versioning"), so nothing is ever read past what the caller has, fields the caller's header did not know
// keep their zero defaults, and an object from a newer header is refused instead of half understood.  Then the CFA shape is folded (below).
#define IPK_FOLD_CFA(T, d) \
  T d##_folded; \
  if (d) { \
    { const int trc_ = take_desc((d), d##_folded); if (trc_) return trc_; } \
    if (!fold_cfa_shape(d##_folded)) \
      return (d##_folded.cfa_width == 0 && d##_folded.cfa_height == 0) ? fail(IPK_ERR_INVALID, "invalid CFA shape prefix (expected \"WxH:letters\" with W*H letters)") \
                                                                        : fail(IPK_ERR_INVALID, "cfa_width / cfa_height do not fit the pattern string"); \
    (d) = &d##_folded; \
  }
template <typename Desc>
static int take_desc(const Desc *in, Desc &out) {
  const size_t have = in->struct_size, oldest = offsetof(Desc, cfa_width);
  if (have == 0) return fail(IPK_ERR_INVALID, "descriptor.struct_size is 0: initialise the descriptor with IPK_*_INIT (struct_size = sizeof of the struct as the caller was compiled)");
  if (have < oldest) return fail(IPK_ERR_INVALID, "descriptor.struct_size %zu is smaller than the first published layout (%zu bytes)", have, oldest);
  if (have > sizeof(Desc)) return fail(IPK_ERR_INVALID, "descriptor.struct_size %zu is larger than this library's struct (%zu bytes): the caller was built against a newer header", have, sizeof(Desc));
  // Appended fields arrive in whole published layouts only: an object shorter than this library's struct is an object of the FIRST layout (its own
  // sizeof is `oldest` rounded up to the struct's alignment -- the bytes between are that compiler's tail padding, indeterminate, and must not land in
  // cfa_width), so exactly the fields of the newest whole layout the object covers are copied.  A later layout adds its boundary to this list.
  const size_t layouts[] = {oldest, offsetof(Desc, schedule), sizeof(Desc)};
  size_t take = oldest;
  for (size_t b : layouts) if (b <= have) take = b;
  std::memset(static_cast<void *>(&out), 0, sizeof(Desc));
  std::memcpy(static_cast<void *>(&out), in, take);
  out.struct_size = (uint32_t)sizeof(Desc);
  return IPK_OK;
}
template <typename Desc>
static bool fold_cfa_shape(Desc &d) {
  d.cfa[sizeof(d.cfa) - 1] = 0;
  if (d.cfa_width == 0 && d.cfa_height == 0 && !std::strchr(d.cfa, ':')) return true;      // plain letters, nothing stated: canonical already
  if ((d.cfa_width != 0 || d.cfa_height != 0) && (d.cfa_width < 1 || d.cfa_height < 1 || d.cfa_width > 48 || d.cfa_height > 48)) return false;
  int w = 0, h = 0; const char *letters = d.cfa;
  if (!ipk::Cfa::split_dims(d.cfa, w, h, letters)) return false;
  if (w != 0 && d.cfa_width != 0 && (w != d.cfa_width || h != d.cfa_height)) return false;
  if (w == 0) { w = d.cfa_width; h = d.cfa_height; }
  const size_t len = std::strlen(letters);
  const bool inferred = w == h && ((w == 2 && len == 4) || (w == 6 && len == 36) || (w == 12 && len == 144));
  char buf[sizeof(d.cfa) + 16];
  const int n = inferred ? std::snprintf(buf, sizeof(buf), "%s", letters) : std::snprintf(buf, sizeof(buf), "%dx%d:%s", w, h, letters);
  if (n < 0 || (size_t)n >= sizeof(d.cfa)) return false;
  std::memcpy(d.cfa, buf, (size_t)n + 1);
  d.cfa_width = 0; d.cfa_height = 0;
  return true;
}

// device-side tables for one CFA pattern string (uploaded once, cached)
int get_cfa(const char *pat, ipk::Cfa &cfa, DevCfa &dev) {
  if (!ipk::Cfa::parse(pat, cfa) || !cfa.valid()) return cfa_fail(pat);
  std::lock_guard<std::mutex> lk(cx().mu);
  auto it = cx().cfa_cache.find(pat);
  if (it != cx().cfa_cache.end()) { dev = it->second; return IPK_OK; }
  uint32_t lookups[48 * 48];
  cfa.demosaic_lookups(lookups);
  DevCfa d;
  HIPCHK(hipMalloc(reinterpret_cast<void **>(&d.lookups), sizeof(lookups)));
  HIPCHK(hipMalloc(reinterpret_cast<void **>(&d.cfa48), 48 * 48));
  HIPCHK(hipMemcpy(d.lookups, lookups, sizeof(lookups), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d.cfa48, &cfa.pattern[0][0], 48 * 48, hipMemcpyHostToDevice));
  std::vector<float> cells;
  if (cfa.gen_cells(cells)) {                                            // arithmetic form of demosaic::full for the row-walking kernel
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&d.gen_cells), cells.size() * sizeof(float)));
    HIPCHK(hipMemcpy(d.gen_cells, cells.data(), cells.size() * sizeof(float), hipMemcpyHostToDevice));
    d.gen_pw = cfa.width; d.gen_ph = cfa.height;
  }
  cx().cfa_cache[pat] = d;
  dev = d;
  return IPK_OK;
}

// scratch pool: buffers are handed out and returned in stream order -- a block goes back as soon as its last user is enqueued, and
// the next user on the SAME stream runs after it.  Nothing is enqueued for the hand-over (round 2 recorded an event per returned
// block; each such marker cost the queue ~5 us of idle time between two runs -- a tenth of a 2160x1440 preview).  A stream never
// takes a block another stream used last: it gets a new one, so after a few runs every stream of a multi-stream caller owns the blocks it
// cycles through.  Only when memory runs out is the device drained, after which every idle block may go anywhere.  No stream handle is
// ever touched except the caller's current one (a previous one may have been destroyed).
// hipStreamPerThread is ONE handle that names a different stream in every host thread: the pool's "same stream" test keys it by the calling
// thread (as the task queues do), so two threads on their per-thread streams never hand each other a block without an ordering between them.
// (A handle value that HIP reuses for a new stream after hipStreamDestroy while the old stream's work is still in flight cannot be told apart:
// callers synchronise a stream before destroying it -- INTEGRATION.md.)
static hipStream_t pool_key(hipStream_t s) {
  static thread_local char per_thread_key;
  return s == hipStreamPerThread ? reinterpret_cast<hipStream_t>(&per_thread_key) : s;
}
static int pool_pick(size_t bytes, hipStream_t stream) {
  int best = -1;
  for (size_t i = 0; i < cx().pool.size(); ++i) {
    const auto &b = cx().pool[i];
    if (!b.busy && b.bytes >= bytes && (b.clean || b.last == stream) && (best < 0 || b.bytes < cx().pool[best].bytes)) best = (int)i;
  }
  return best;
}
int pool_get(size_t bytes, void **out, hipStream_t stream) {
  stream = pool_key(stream);
  std::lock_guard<std::mutex> lk(cx().mu);
  int best = pool_pick(bytes, stream);
  if (best < 0) {
    bool foreign_fits = false;
    for (const auto &b : cx().pool) foreign_fits = foreign_fits || (!b.busy && b.bytes >= bytes);
    // nothing fits on any stream: the idle blocks are all too small for this frame size, so they go before a larger one is allocated
    // (a long-running process that moves between frame sizes keeps only what its current size needs; hipFree waits for their users)
    if (!foreign_fits)
      for (size_t i = cx().pool.size(); i-- > 0;)
        if (!cx().pool[i].busy) { (void)hipFree(cx().pool[i].p); cx().pool.erase(cx().pool.begin() + (long)i); }
    void *p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 1) == hipSuccess) {
      cx().pool.push_back({p, bytes, true, stream, false});
      *out = p;
      return IPK_OK;
    }
    (void)hipGetLastError();
    // out of memory: drain the device -- every idle block is then free of users -- and look again, dropping what is too small
    if (hipDeviceSynchronize() != hipSuccess) return fail(IPK_ERR_HIP, "hipDeviceSynchronize failed");
    for (auto &b : cx().pool) if (!b.busy) b.clean = true;
    best = pool_pick(bytes, stream);
    if (best < 0) {
      for (size_t i = cx().pool.size(); i-- > 0;)
        if (!cx().pool[i].busy) { (void)hipFree(cx().pool[i].p); cx().pool.erase(cx().pool.begin() + (long)i); }
      if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) return fail(IPK_ERR_NOMEM, "hipMalloc(%zu) failed", bytes);
      cx().pool.push_back({p, bytes, true, stream, false});
      *out = p;
      return IPK_OK;
    }
  }
  auto &b = cx().pool[best];
  b.busy = true; b.last = stream; b.clean = false; *out = b.p;
  return IPK_OK;
}
void pool_put(void *p, hipStream_t stream) {
  if (!p) return;
  stream = pool_key(stream);
  std::lock_guard<std::mutex> lk(cx().mu);
  for (auto &b : cx().pool) if (b.p == p) { b.last = stream; b.busy = false; return; }
}
struct Scratch {                       // RAII: returns its buffers to the pool
  hipStream_t stream;
  std::vector<void *> bufs;
  explicit Scratch(hipStream_t s) : stream(s) {}
  ~Scratch() { for (void *p : bufs) pool_put(p, stream); }
  int get(size_t bytes, void **out) { int rc = pool_get(bytes, out, stream); if (!rc) bufs.push_back(*out); return rc; }
  void release(void *p) { pool_put(p, stream); bufs.erase(std::remove(bufs.begin(), bufs.end(), p), bufs.end()); }
};


// Can OpGoFloat's `(v - black) / range` run as the 4-instruction cdiv_fast in the fused kernel?
// The kernel-side proof obligations (ipk_device.hpp): range positive and ordinary, the fast quotient equal to the
// true one on the dividends this source can produce, and (u16 sources, which the kernel does not guard) every
// nonzero dividend inside [2^-100, 2^100].  Anything else makes the kernel use true divisions.
// u16 sources: is every normalised sample min((v - black) / range, 1.0) "ordinary" (zero or inside [2^-20, 2^20], the
// device's gen_sample_bad)?  Then the kernels need no sample checks at all.
bool gen_levels_ok_u16(float black, float range) {
  static thread_local struct { uint32_t b, r; int ok; bool set; } memo = {0, 0, 0, false};
  uint32_t bb, rb; std::memcpy(&bb, &black, 4); std::memcpy(&rb, &range, 4);
  if (memo.set && memo.b == bb && memo.r == rb) return memo.ok != 0;
  bool ok = true;
  for (uint32_t v = 0; v < 65536 && ok; ++v) {
    const float q = std::fmin(((float)v - black) / range, 1.0f);
    const float a = std::fabs(q);
    ok = (q == 0.0f) || (a >= 0x1p-20f && a <= 0x1p20f);
  }
  memo = {bb, rb, ok ? 1 : 0, true};
  return ok;
}
bool validate_cdiv_for_range_uncached(float black, float range, bool src_is_u16);
bool validate_cdiv_for_range(float black, float range, bool src_is_u16) {
  // one-entry memo: a pipeline is launched many times with the same levels
  static thread_local struct { uint32_t b, r; int u16, ok; bool set; } memo = {0, 0, 0, 0, false};
  uint32_t bb, rb; std::memcpy(&bb, &black, 4); std::memcpy(&rb, &range, 4);
  if (memo.set && memo.b == bb && memo.r == rb && memo.u16 == (int)src_is_u16) return memo.ok != 0;
  const bool ok = validate_cdiv_for_range_uncached(black, range, src_is_u16);
  memo = {bb, rb, (int)src_is_u16, ok ? 1 : 0, true};
  return ok;
}
bool validate_cdiv_for_range_uncached(float black, float range, bool src_is_u16) {
  if (!(range >= 0x1p-60f && range <= 0x1p60f)) return false;             // also rejects NaN, <= 0
  const float rc = 1.0f / range;
  auto ok = [&](float d) {
    if (d == 0.0f || d != d || std::isinf(d)) return true;                // v_div_fixup_f32 supplies these
    const float ad = std::fabs(d);
    if (ad < 0x1p-100f || ad > 0x1p100f) return !src_is_u16;              // f32 kernels guard this zone themselves
    const float q0 = d * rc;
    const float r = std::fma(-q0, range, d);
    return std::fma(r, rc, q0) == d / range;
  };
  if (src_is_u16) {
    for (uint32_t v = 0; v < 65536; ++v) if (!ok((float)v - black)) return false;
    return true;
  }
  // f32 sources: any dividend can occur.  Every dividend mantissa against this divisor (a proof for the guarded exponent zone,
  // ipk_host.hpp), then a spread of exponents as a check of the scaling argument itself.
  if (!ipk::cdiv_mantissa_exhaustive_ok(range)) return false;
  uint64_t st = 0x9E3779B97F4A7C15ull;
  for (int i = 0; i < 4096; ++i) {
    st = st * 6364136223846793005ull + 1442695040888963407ull;
    uint32_t bits = (uint32_t)(st >> 32);
    bits = (bits & 0x807FFFFFu) | ((27u + (bits >> 23) % 200u) << 23);      // exponent in [2^-100, 2^100)
    float d; std::memcpy(&d, &bits, 4);
    if (!ok(d)) return false;
  }
  return true;
}

// One cached OpBuffer: device memory owned by the cache (and by whoever still holds the shared_ptr).
struct CBuf {
  void *p = nullptr; size_t w = 0, h = 0, colors = 0; int mono = 0;
  ~CBuf() { if (p) (void)hipFree(p); }
  size_t bytes() const { return w * h * colors * sizeof(float); }        // pipeline.rs:369
};
using CBufP = std::shared_ptr<CBuf>;
int cbuf_new(size_t w, size_t h, size_t colors, int mono, CBufP &out) {
  out = std::make_shared<CBuf>();
  out->w = w; out->h = h; out->colors = colors; out->mono = mono;
  if (hipMalloc(&out->p, std::max<size_t>(out->bytes(), 1)) != hipSuccess) { out->p = nullptr; return fail(IPK_ERR_NOMEM, "hipMalloc(%zu) failed", out->bytes()); }
  return IPK_OK;
}

}  // namespace

struct ipk_cache {
  ipk::LruByteCache<CBuf> lru;
  ipk_ctx *owner;                        // the context (device) whose memory the cached OpBuffers live in; null for a cache made before any context
  explicit ipk_cache(size_t bytes, ipk_ctx *o) : lru(bytes), owner(o) {}
};

namespace {
template <typename T>
int rotate_typed(const T *src3, size_t bwidth, size_t bheight, int orientation, T *dst3, size_t *out_width, size_t *out_height, void *stream) {
  REQUIRE_INIT();
  if (!src3 || !dst3 || !out_width || !out_height || !dims_ok(bwidth, bheight)) return fail(IPK_ERR_INVALID, "bad rotate arguments");
  bool transpose, flip_x, flip_y;
  ipk::orientation_to_flips(orientation, transpose, flip_x, flip_y);
  // transform.rs:102-128 in units of pixels
  int64_t width = (int64_t)bwidth, height = (int64_t)bheight;
  int64_t base = 0, x_step = 1, y_step = width;
  if (flip_x) { x_step = -x_step; base += width - 1; }
  if (flip_y) { y_step = -y_step; base += width * (height - 1); }
  if (transpose) { std::swap(width, height); std::swap(x_step, y_step); }
  *out_width = (size_t)width; *out_height = (size_t)height;
  ipk::launch_rotate<T>(src3, (size_t)width, (size_t)height, base, x_step, y_step, dst3, S(stream));
  HIPCHK(hipGetLastError());
  return IPK_OK;
}
}  // namespace

extern "C" {

// ------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------
// Everything a context keeps on its device.  `c` is not yet visible to any other thread; the caller has the device bound.
static int ctx_build(Context &c, int device) {
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  c.device = device;
  c.num_cus = prop.multiProcessorCount;
  build_host_luts();
  for (int t = 0; t < 3; ++t) {
    // {table[i], table[i+1]-table[i]}: the subtraction of lookup() (color_conversions.rs:112) hoisted
    std::vector<float> pairs(2 * 8192);
    for (int i = 0; i < 8192; ++i) { pairs[2 * i] = g_host.lut_host[t][i]; pairs[2 * i + 1] = g_host.lut_host[t][i + 1] - g_host.lut_host[t][i]; }
    HIPCHK(hipMalloc(&c.lut_pairs[t], pairs.size() * sizeof(float)));
    HIPCHK(hipMemcpy(c.lut_pairs[t], pairs.data(), pairs.size() * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMalloc(&c.lut_plain[t], ipk::kLutLen * sizeof(float)));
    HIPCHK(hipMemcpy(c.lut_plain[t], g_host.lut_host[t].data(), ipk::kLutLen * sizeof(float), hipMemcpyHostToDevice));
  }
  // OpGamma + output8bit as one step lookup (ipk_device.hpp Q8Entry): built on the device from THIS host's gamma table, and checked against the literal
  // composition on every f32 before the context goes live (a few milliseconds) -- the 8-bit kernels have no other form to fall back to, so a table that
  // failed the check (it cannot, by the slope argument; a broken powf could) stops ipk_init loudly instead of producing wrong bytes
  HIPCHK(hipMalloc(&c.lut_q8, 8192 * 8));
  ipk::launch_build_q8(c.lut_pairs[ipk::kLutGamma], c.lut_q8, nullptr);
  HIPCHK(hipGetLastError());
  {
    struct { unsigned long long bad; unsigned int first; unsigned int pad; } h = {0, 0xFFFFFFFFu, 0};
    void *dev = nullptr;
    HIPCHK(hipMalloc(&dev, sizeof(h)));
    HIPCHK(hipMemcpy(dev, &h, sizeof(h), hipMemcpyHostToDevice));
    ipk::launch_selftest_q8(c.lut_pairs[ipk::kLutGamma], c.lut_q8, dev, nullptr);
    const hipError_t e = hipMemcpy(&h, dev, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipFree(dev);
    if (e != hipSuccess) return fail(IPK_ERR_HIP, "the 8-bit step table could not be verified: %s", hipGetErrorString(e));
    if (h.bad != 0) return fail(IPK_ERR_UNSUPPORTED, "the 8-bit step table disagrees with OpGamma + output8bit on %llu inputs (first 0x%08x): this host's gamma table is not monotone?", h.bad, h.first);
  }
  // the row-walking kernels' task-queue heads, one block for all streams of this context (nothing is allocated at launch time, so launches can be captured)
  c.queues = ipk::create_task_queues();
  if (!c.queues) return fail(IPK_ERR_HIP, "task queue allocation failed");
  // The device's cbrtf reproduces glibc 2.35's routine; the lookup tables above and a reference built on THIS host use this host's libm.
  // If the two disagree (another libc, a newer glibc with a correctly rounded cbrtf) results stay deterministic but are no longer
  // bit-identical to that reference for Lab ratios above 1: checked here on 65 536 arguments spread over (1, 8) and reported through
  // ipk_host_libm_matches() -- never silently.
  {
    const size_t n = 65536;
    std::vector<float> in(n), out(n);
    for (size_t i = 0; i < n; ++i) { const uint32_t bits = 0x3F800001u + (uint32_t)((i * 0x017FFFFFull) / n); std::memcpy(&in[i], &bits, 4); }   // (1, 8)
    void *din = nullptr, *dout = nullptr;
    c.libm_matches = -1;
    if (hipMalloc(&din, n * 4) == hipSuccess && hipMalloc(&dout, n * 4) == hipSuccess &&
        hipMemcpy(din, in.data(), n * 4, hipMemcpyHostToDevice) == hipSuccess) {
      ipk::launch_selftest_cbrt(static_cast<const float *>(din), static_cast<float *>(dout), n, 1, nullptr);
      if (hipMemcpy(out.data(), dout, n * 4, hipMemcpyDeviceToHost) == hipSuccess) {
        size_t bad = 0;
        for (size_t i = 0; i < n; ++i) { const float h = cbrtf(in[i]); bad += std::memcmp(&h, &out[i], 4) != 0; }
        c.libm_matches = bad == 0 ? 1 : 0;
        c.libm_mismatches = bad;
      }
    }
    if (din) (void)hipFree(din);
    if (dout) (void)hipFree(dout);
  }
  c.ready = true;
  return IPK_OK;
}
// releases what ctx_build and later calls put on the device; the object itself stays in the registry (dead)
static void ctx_release(Context &c) {
  if (c.device < 0) return;
  int prev = -1;
  (void)hipGetDevice(&prev);
  if (hipSetDevice(c.device) != hipSuccess) { (void)hipGetLastError(); return; }
  c.ready = false;
  // only this context's work has to be over: its lanes, its batch stream and whatever its callers enqueued (their streams are theirs to drain
  // before destroying a context -- the device-wide wait below is the backstop, as it was for ipk_shutdown)
  (void)hipDeviceSynchronize();
  for (int t = 0; t < 3; ++t) { if (c.lut_pairs[t]) (void)hipFree(c.lut_pairs[t]); c.lut_pairs[t] = nullptr; if (c.lut_plain[t]) (void)hipFree(c.lut_plain[t]); c.lut_plain[t] = nullptr; }
  if (c.lut_q8) { (void)hipFree(c.lut_q8); c.lut_q8 = nullptr; }
  {
    std::lock_guard<std::mutex> lk(c.mu);
    for (auto &kv : c.cfa_cache) { (void)hipFree(kv.second.lookups); (void)hipFree(kv.second.cfa48); if (kv.second.gen_cells) (void)hipFree(kv.second.gen_cells); }
    c.cfa_cache.clear();
    for (auto &kv : c.rot_cells) (void)hipFree(kv.second);
    c.rot_cells.clear();
    for (auto &b : c.pool) (void)hipFree(b.p);
    c.pool.clear();
  }
  { std::lock_guard<std::mutex> lk(c.lanes.mu); c.lanes.release(); }
  if (c.multi_stream) { (void)hipStreamDestroy(c.multi_stream); c.multi_stream = nullptr; }
  ipk::destroy_task_queues(c.queues); c.queues = nullptr;
  c.num_cus = 0;
  if (prev >= 0 && prev != c.device) (void)hipSetDevice(prev);
}
static int ctx_create_locked(int device, Context **out) {      // g_reg_mu held
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); return fail(IPK_ERR_NO_DEVICE, "no HIP device visible"); }
  if (device < 0 || device >= n) return fail(IPK_ERR_INVALID, "device %d out of range (%d visible)", device, n);
  HIPCHK(hipSetDevice(device));
  std::unique_ptr<Context> c(new Context);
  const int rc = ctx_build(*c, device);
  if (rc) { ctx_release(*c); return rc; }
  *out = c.get();
  g_registry.push_back(std::move(c));
  return IPK_OK;
}
static bool ctx_known_locked(const Context *c) {
  for (const auto &p : g_registry) if (p.get() == c) return true;
  return false;
}

int ipk_init(int device) {
  std::lock_guard<std::mutex> lk(g_reg_mu);
  { Context *d0 = g_default.load(); if (d0 && d0->ready && d0->device == device) return IPK_OK; }
  Context *c = nullptr;
  const int rc = ctx_create_locked(device, &c);
  if (rc) return rc;
  // re-initialising on another device replaces the process default; the old default goes unless the device set still uses it
  Context *old = g_default.load();
  g_default = c;
  if (old && old->ready && std::find(g_devset.begin(), g_devset.end(), old) == g_devset.end()) { ctx_release(*old); (void)bind_device(device); }
  return IPK_OK;
}
int ipk_host_libm_matches(size_t *mismatches_of_65536) {
  if (!cx().ready) return fail(IPK_ERR_NOT_INIT, "ipk_init() has not succeeded");
  if (mismatches_of_65536) *mismatches_of_65536 = cx().libm_mismatches;
  return cx().libm_matches;
}

void ipk_shutdown(void) {
  std::lock_guard<std::mutex> lk(g_reg_mu);
  for (auto &c : g_registry) if (c->ready) ctx_release(*c);
  g_default = nullptr;
  g_devset.clear();
  t_current = nullptr;
}
int ipk_is_initialized(void) { return cx().ready ? 1 : 0; }

// ---- contexts (several devices from one process) ------------------------------------------------------------------------------
int ipk_ctx_create(int device, ipk_ctx **out) {
  if (!out) return fail(IPK_ERR_INVALID, "null output");
  *out = nullptr;
  std::lock_guard<std::mutex> lk(g_reg_mu);
  int prev = -1; (void)hipGetDevice(&prev);
  const int rc = ctx_create_locked(device, out);
  // creating a context does not move the calling thread: its current context (and that context's device) stay what they were
  if (prev >= 0) (void)hipSetDevice(prev);
  (void)hipGetLastError();
  return rc;
}
int ipk_ctx_destroy(ipk_ctx *ctx) {
  if (!ctx) return IPK_OK;
  std::lock_guard<std::mutex> lk(g_reg_mu);
  if (!ctx_known_locked(ctx)) return fail(IPK_ERR_INVALID, "not a context of this library");
  if (ctx->ready) ctx_release(*ctx);
  if (g_default.load() == ctx) g_default = nullptr;
  g_devset.erase(std::remove(g_devset.begin(), g_devset.end(), ctx), g_devset.end());
  if (t_current == ctx) t_current = nullptr;
  return IPK_OK;
}
int ipk_ctx_make_current(ipk_ctx *ctx) {
  if (!ctx) { t_current = nullptr; return cx().ready ? bind_device(cx().device) : IPK_OK; }
  { std::lock_guard<std::mutex> lk(g_reg_mu);
    if (!ctx_known_locked(ctx)) return fail(IPK_ERR_INVALID, "not a context of this library");
    if (!ctx->ready) return fail(IPK_ERR_NOT_INIT, "the context has been destroyed"); }
  t_current = ctx;
  return bind_device(ctx->device);
}
ipk_ctx *ipk_ctx_current(void) { Context &c = cx(); return c.ready ? &c : nullptr; }
int ipk_ctx_device(const ipk_ctx *ctx) { return (ctx && ctx->ready) ? ctx->device : -1; }

int ipk_init_devices(const int *devices, int n) {
  int visible = 0;
  if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0) { (void)hipGetLastError(); return fail(IPK_ERR_NO_DEVICE, "no HIP device visible"); }
  if (n < 0 || n > 64 || (n > 0 && !devices)) return fail(IPK_ERR_INVALID, "bad device list");
  std::vector<int> want;
  if (n == 0) for (int i = 0; i < visible; ++i) want.push_back(i);       // every visible device
  else want.assign(devices, devices + n);
  for (int d : want) if (d < 0 || d >= visible) return fail(IPK_ERR_INVALID, "device %d out of range (%d visible)", d, visible);
  std::lock_guard<std::mutex> lk(g_reg_mu);
  int prev = -1; (void)hipGetDevice(&prev);
  // the previous set goes (its contexts are released unless one is the process default)
  Context *const dflt = g_default.load();
  for (Context *c : g_devset) if (c != dflt && c->ready) ctx_release(*c);
  g_devset.clear();
  int rc = IPK_OK;
  for (size_t i = 0; i < want.size() && rc == IPK_OK; ++i) {
    Context *c = nullptr;
    // the process default serves its own device's first entry: a caller of ipk_init(0) + ipk_init_devices({0..7}) has eight contexts, not nine
    if (dflt && dflt->ready && dflt->device == want[i] && std::find(g_devset.begin(), g_devset.end(), dflt) == g_devset.end()) c = dflt;
    else rc = ctx_create_locked(want[i], &c);
    if (rc == IPK_OK) g_devset.push_back(c);
  }
  if (rc != IPK_OK) {
    for (Context *c : g_devset) if (c != dflt && c->ready) ctx_release(*c);
    g_devset.clear();
  } else if (!dflt || !dflt->ready) {
    g_default = g_devset[0];                                             // ipk_init_devices alone is a complete initialisation
  }
  { Context *d1 = g_default.load();
    if (d1 && d1->ready && !t_current) (void)hipSetDevice(d1->device);
    else if (prev >= 0) (void)hipSetDevice(prev); }
  (void)hipGetLastError();
  return rc;
}
int ipk_device_set_size(void) { std::lock_guard<std::mutex> lk(g_reg_mu); return (int)g_devset.size(); }
ipk_ctx *ipk_device_ctx(int index) {
  std::lock_guard<std::mutex> lk(g_reg_mu);
  return (index >= 0 && (size_t)index < g_devset.size()) ? g_devset[(size_t)index] : nullptr;
}
// The dealing rule of the multi-device batch entry points, as arithmetic a caller (and the CPU tests) can ask for: frame i of n goes to set
// member i mod n_devices, so member `index` gets frames index, index + n_devices, ... -- *count of them.  Frames are independent pipelines
// (src/pipeline.rs:246-249): nothing else is shared out.
int ipk_deal_frames(size_t n_frames, int n_devices, int index, size_t *first, size_t *stride, size_t *count) {
  if (n_devices < 1 || index < 0 || index >= n_devices) return fail(IPK_ERR_INVALID, "bad device index %d of %d", index, n_devices);
  const size_t nd = (size_t)n_devices, ix = (size_t)index;
  if (first) *first = ix;
  if (stride) *stride = nd;
  if (count) *count = n_frames > ix ? (n_frames - ix + nd - 1) / nd : 0;
  return IPK_OK;
}

// the layout this library was built with, for bindings to check theirs against (no GPU needed)
size_t ipk_abi_sizeof(int which) {
  switch (which) {
    case 0: return sizeof(ipk_fused_params);
    case 1: return sizeof(ipk_pipeline_desc);
    case 2: return sizeof(ipk_band);
    case 3: return sizeof(ipk_stage_time);
    case 16: return offsetof(ipk_fused_params, cfa_width);
    case 17: return offsetof(ipk_pipeline_desc, cfa_width);
    case 18: return offsetof(ipk_fused_params, band_src_row0);
    case 19: return offsetof(ipk_pipeline_desc, use_fastpath);
    case 20: return offsetof(ipk_fused_params, schedule);
    case 21: return offsetof(ipk_pipeline_desc, schedule);
    default: return 0;
  }
}
const char *ipk_last_error(void) { return g_err; }
int ipk_device_cus(void) { return cx().num_cus; }

int ipk_malloc(void **dptr, size_t bytes) { REQUIRE_INIT(); HIPCHK(hipMalloc(dptr, bytes ? bytes : 1)); return IPK_OK; }
int ipk_free(void *dptr) { REQUIRE_INIT(); HIPCHK(hipFree(dptr)); return IPK_OK; }
int ipk_memcpy_h2d(void *dst, const void *src, size_t bytes, void *stream) {
  REQUIRE_INIT(); HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, S(stream))); return IPK_OK;
}
int ipk_memcpy_d2h(void *dst, const void *src, size_t bytes, void *stream) {
  REQUIRE_INIT(); HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, S(stream))); return IPK_OK;
}
int ipk_stream_sync(void *stream) { REQUIRE_INIT(); HIPCHK(hipStreamSynchronize(S(stream))); return IPK_OK; }

int ipk_lut_table(int which, float *out8193) {
  if (which < 0 || which > 2 || !out8193) return fail(IPK_ERR_INVALID, "bad table id");
  build_host_luts();
  std::memcpy(out8193, g_host.lut_host[which].data(), ipk::kLutLen * sizeof(float));
  return IPK_OK;
}

// ------------------------------------------------------------------------------------------
// host-side maths
// ------------------------------------------------------------------------------------------
int ipk_size_image(size_t crop_top, size_t crop_right, size_t crop_bottom, size_t crop_left,
                   size_t owidth, size_t oheight, size_t *out4) {
  ipk::Rect r;
  if (!ipk::size_image(crop_top, crop_right, crop_bottom, crop_left, owidth, oheight, r))
    return fail(IPK_ERR_INVALID, "image %zux%zu is smaller than 10x10", owidth, oheight);
  out4[0] = r.x; out4[1] = r.y; out4[2] = r.width; out4[3] = r.height;
  return IPK_OK;
}
int ipk_calculate_scaling_total(size_t width, size_t height, size_t maxwidth, size_t maxheight,
                                float *scale, size_t *nwidth, size_t *nheight) {
  const ipk::Scaling s = ipk::calculate_scaling_total(width, height, maxwidth, maxheight);
  *scale = s.scale; *nwidth = s.width; *nheight = s.height;
  return IPK_OK;
}
int ipk_normalize_wbs(const float *vals4, float *out4) { ipk::normalize_wbs(vals4, out4); return IPK_OK; }
int ipk_const_matrix(int which, float *out12) {
  if (!out12) return fail(IPK_ERR_INVALID, "ipk_const_matrix: null output");
  const ipk::Mat33 s = ipk::srgb_d65_33(), x = ipk::inverse(s);
  switch (which) {
    case 0: for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) out12[r * 3 + c] = s.m[r][c]; return IPK_OK;      // SRGB_D65_33
    case 1: for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) out12[r * 3 + c] = x.m[r][c]; return IPK_OK;      // XYZ_D65_33
    case 2: ipk::srgb_d65_43(out12); return IPK_OK;                                                                   // SRGB_D65_43
    case 3: for (int r = 0; r < 4; ++r) for (int c = 0; c < 3; ++c) out12[r * 3 + c] = r < 3 ? x.m[r][c] : 0.0f; return IPK_OK;   // XYZ_D65_34
  }
  return fail(IPK_ERR_INVALID, "ipk_const_matrix: which must be 0..3");
}
int ipk_temp_to_xyz(float temp, float *out3) { ipk::temp_to_xyz(temp, out3); return IPK_OK; }
int ipk_xyz_to_temp(const float *xyz3, float *temp, float *tint) { ipk::xyz_to_temp(xyz3, *temp, *tint); return IPK_OK; }
int ipk_tolab_set_temp(const float *xyz_to_cam12, float temp, float tint, float *wb4) { ipk::tolab_set_temp(xyz_to_cam12, temp, tint, wb4); return IPK_OK; }
int ipk_tolab_get_temp(const float *cam_to_xyz12, const float *wb4, float *temp, float *tint) { ipk::tolab_get_temp(cam_to_xyz12, wb4, *temp, *tint); return IPK_OK; }
int ipk_spline_new(const float *pts, int npts, float *px, float *py, float *c1s, float *c2s, float *c3s) {
  ipk::Spline s;
  if (!s.build(pts, npts)) return fail(IPK_ERR_INVALID, "invalid curve (%d points)", npts);
  std::memcpy(px, s.px, s.npoints * sizeof(float)); std::memcpy(py, s.py, s.npoints * sizeof(float));
  std::memcpy(c1s, s.c1, s.npoints * sizeof(float));
  std::memcpy(c2s, s.c2, s.nseg * sizeof(float)); std::memcpy(c3s, s.c3, s.nseg * sizeof(float));
  return s.npoints;
}
int ipk_rotatecrop_calc_size(const float *p, float input_ratio, size_t width, size_t height, int reverse,
                             size_t *nwidth, size_t *nheight) {
  ipk::RotateCrop rc;
  rc.crop_top = p[0]; rc.crop_right = p[1]; rc.crop_bottom = p[2]; rc.crop_left = p[3]; rc.rotation = p[4];
  rc.input_ratio = input_ratio;
  rc.calc_size(width, height, reverse != 0, *nwidth, *nheight);
  return IPK_OK;
}
int ipk_cfa_shift(const char *pattern, int x, int y, char *out) {
  ipk::Cfa c;
  if (!ipk::Cfa::parse(pattern, c)) return cfa_fail(pattern);
  const std::string s = c.shifted_name(x, y);
  std::memcpy(out, s.c_str(), s.size() + 1);
  return IPK_OK;
}
int ipk_orientation_to_flips(int orientation, int *f) {
  bool t, x, y; ipk::orientation_to_flips(orientation, t, x, y); f[0] = t; f[1] = x; f[2] = y; return IPK_OK;
}
int ipk_orientation_from_flips(int transpose, int flip_x, int flip_y) { return ipk::orientation_from_flips(transpose, flip_x, flip_y); }
int ipk_transform_orientation(int rotation, int fliph, int flipv) {
  if (rotation < 0 || rotation > 3) return fail(IPK_ERR_INVALID, "bad rotation");
  return ipk::transform_orientation(rotation, fliph != 0, flipv != 0);
}

// ------------------------------------------------------------------------------------------
// stage kernels (device pointers)
// ------------------------------------------------------------------------------------------
#define GOFLOAT_ARGS_OK() do { REQUIRE_INIT(); if (!src || !dst || !dims_ok(width, height) || !dims_ok(owidth, 1)) return fail(IPK_ERR_INVALID, "bad gofloat arguments"); } while (0)

int ipk_gofloat_cfa_u16(const uint16_t *src, size_t owidth, size_t x, size_t y, size_t width, size_t height,
                        float black0, float white0, float *dst, void *stream) {
  GOFLOAT_ARGS_OK(); ipk::launch_gofloat_cfa<uint16_t>(src, owidth, x, y, width, height, black0, white0, dst, S(stream)); HIPCHK(hipGetLastError()); return IPK_OK;
}
int ipk_gofloat_cfa_f32(const float *src, size_t owidth, size_t x, size_t y, size_t width, size_t height,
                        float black0, float white0, float *dst, void *stream) {
  GOFLOAT_ARGS_OK(); ipk::launch_gofloat_cfa<float>(src, owidth, x, y, width, height, black0, white0, dst, S(stream)); HIPCHK(hipGetLastError()); return IPK_OK;
}
int ipk_gofloat_mono_u16(const uint16_t *src, size_t owidth, size_t x, size_t y, size_t width, size_t height,
                         float black0, float white0, float *dst, void *stream) {
  GOFLOAT_ARGS_OK(); ipk::launch_gofloat_mono<uint16_t>(src, owidth, x, y, width, height, black0, white0, dst, S(stream)); HIPCHK(hipGetLastError()); return IPK_OK;
}
int ipk_gofloat_mono_f32(const float *src, size_t owidth, size_t x, size_t y, size_t width, size_t height,
                         float black0, float white0, float *dst, void *stream) {
  GOFLOAT_ARGS_OK(); ipk::launch_gofloat_mono<float>(src, owidth, x, y, width, height, black0, white0, dst, S(stream)); HIPCHK(hipGetLastError()); return IPK_OK;
}
int ipk_gofloat_rgb_u16(const uint16_t *src, size_t owidth, size_t x, size_t y, size_t width, size_t height,
                        const float *black4, const float *white4, float *dst, void *stream) {
  GOFLOAT_ARGS_OK(); ipk::launch_gofloat_rgb<uint16_t>(src, owidth, x, y, width, height, black4, white4, dst, S(stream)); HIPCHK(hipGetLastError()); return IPK_OK;
}
int ipk_gofloat_rgb_f32(const float *src, size_t owidth, size_t x, size_t y, size_t width, size_t height,
                        const float *black4, const float *white4, float *dst, void *stream) {
  GOFLOAT_ARGS_OK(); ipk::launch_gofloat_rgb<float>(src, owidth, x, y, width, height, black4, white4, dst, S(stream)); HIPCHK(hipGetLastError()); return IPK_OK;
}
int ipk_gofloat_other_u8(const uint8_t *src, size_t owidth, size_t x, size_t y, size_t width, size_t height, float *dst, void *stream) {
  GOFLOAT_ARGS_OK(); ipk::launch_gofloat_other_u8(src, owidth, x, y, width, height, cx().lut_pairs[ipk::kLutGammaReverse], dst, S(stream)); HIPCHK(hipGetLastError()); return IPK_OK;
}
int ipk_gofloat_other_u16(const uint16_t *src, size_t owidth, size_t x, size_t y, size_t width, size_t height, float *dst, void *stream) {
  GOFLOAT_ARGS_OK(); ipk::launch_gofloat_other_u16(src, owidth, x, y, width, height, dst, S(stream)); HIPCHK(hipGetLastError()); return IPK_OK;
}

int ipk_demosaic_full_band(const float *src, size_t width, size_t img_height, size_t src_row0, size_t src_rows,
                           size_t out_row0, size_t out_rows, const char *cfa_pat, float *dst4, void *stream) {
  REQUIRE_INIT();
  if (!src || !dst4 || !dims_ok(width, img_height) || out_rows == 0 || out_row0 + out_rows > img_height)
    return fail(IPK_ERR_INVALID, "bad demosaic arguments");
  // the band must carry every in-image tap row of its output rows
  const size_t need0 = out_row0 > 0 ? out_row0 - 1 : 0;
  const size_t need1 = std::min(img_height, out_row0 + out_rows + 1);
  if (src_row0 > need0 || src_row0 + src_rows < need1) return fail(IPK_ERR_INVALID, "band rows [%zu,%zu) do not cover taps [%zu,%zu)", src_row0, src_row0 + src_rows, need0, need1);
  ipk::Cfa cfa; DevCfa dev;
  int rc = get_cfa(cfa_pat, cfa, dev); if (rc) return rc;
  int xoff, yoff;
  int lrc = 0;
  if (cfa.bayer_phase(xoff, yoff))       // the four RGGB phases: row-walking kernel (coalesced loads, register window, staged stores)
    lrc = ipk::launch_demosaic_bayer(src, width, img_height, src_row0, out_row0, out_rows, xoff, yoff, nullptr, 0, 0, dst4, cx().num_cus, cx().queues, S(stream));
  else if (dev.gen_cells)                // any other three-colour filter (X-Trans ...): same kernel, generic-CFA mode
    lrc = ipk::launch_demosaic_bayer(src, width, img_height, src_row0, out_row0, out_rows, 0, 0, dev.gen_cells, dev.gen_pw, dev.gen_ph, dst4, cx().num_cus, cx().queues, S(stream));
  else ipk::launch_demosaic_full(src, width, img_height, src_row0, out_row0, out_rows, dev.lookups, dst4, S(stream));
  if (lrc) return fail(IPK_ERR_HIP, "kernel launch failed (nothing was enqueued)");
  HIPCHK(hipGetLastError());
  return IPK_OK;
}
int ipk_demosaic_full(const float *src, size_t width, size_t height, const char *cfa, float *dst4, void *stream) {
  return ipk_demosaic_full_band(src, width, height, 0, height, 0, height, cfa, dst4, stream);
}

#define TRANSFORM_BODY(T)                                                                                    \
  REQUIRE_INIT();                                                                                            \
  if (!src || !dst || !dims_ok(width, height) || !dims_ok(nwidth, nheight) || components < 1 || components > 4) \
    return fail(IPK_ERR_INVALID, "bad transform_buffer arguments");                                          \
  const uint8_t *cfa48 = nullptr;                                                                            \
  if (cfa_pat) { ipk::Cfa cfa; DevCfa dev; int rc = get_cfa(cfa_pat, cfa, dev); if (rc) return rc; cfa48 = dev.cfa48; } \
  ipk::launch_transform_buffer<T>(src, width, height, tlx, tly, trx, try_, blx, bly, nwidth, nheight, components, cfa48, dst, S(stream)); \
  HIPCHK(hipGetLastError());                                                                                 \
  return IPK_OK;

int ipk_transform_buffer_f32(const float *src, size_t width, size_t height, int64_t tlx, int64_t tly, int64_t trx, int64_t try_,
                             int64_t blx, int64_t bly, size_t nwidth, size_t nheight, size_t components, const char *cfa_pat,
                             float *dst, void *stream) { TRANSFORM_BODY(float) }
int ipk_transform_buffer_u8(const uint8_t *src, size_t width, size_t height, int64_t tlx, int64_t tly, int64_t trx, int64_t try_,
                            int64_t blx, int64_t bly, size_t nwidth, size_t nheight, size_t components, const char *cfa_pat,
                            uint8_t *dst, void *stream) { TRANSFORM_BODY(uint8_t) }
int ipk_transform_buffer_u16(const uint16_t *src, size_t width, size_t height, int64_t tlx, int64_t tly, int64_t trx, int64_t try_,
                             int64_t blx, int64_t bly, size_t nwidth, size_t nheight, size_t components, const char *cfa_pat,
                             uint16_t *dst, void *stream) { TRANSFORM_BODY(uint16_t) }

// scale_down_buffer's corners (src/scaling.rs:35-48)
int ipk_scaled_demosaic(const float *src, size_t width, size_t height, const char *cfa, size_t nwidth, size_t nheight, float *dst4, void *stream) {
  if (!cfa) return fail(IPK_ERR_INVALID, "scaled_demosaic needs a CFA");
  return ipk_transform_buffer_f32(src, width, height, 0, 0, (int64_t)width - 1, 0, 0, (int64_t)height - 1, nwidth, nheight, 4, cfa, dst4, stream);
}
int ipk_scale_down_opbuf(const float *src4, size_t width, size_t height, size_t nwidth, size_t nheight, float *dst4, void *stream) {
  return ipk_transform_buffer_f32(src4, width, height, 0, 0, (int64_t)width - 1, 0, 0, (int64_t)height - 1, nwidth, nheight, 4, nullptr, dst4, stream);
}

// OpGoFloat (CFA branch) + scaled_demosaic in one pass over the raw frame (used by ipk_pipeline_run when OpDemosaic::run
// would take its scaled_demosaic branch: the 1-channel f32 intermediate never exists)
int ipk_raw_scaled_demosaic(const void *src, int src_type, size_t owidth, size_t x, size_t y, size_t width, size_t height,
                            float black0, float white0, const char *cfa_pat, size_t nwidth, size_t nheight, float *dst4, void *stream) {
  REQUIRE_INIT();
  if (!src || !dst4 || !cfa_pat || !dims_ok(width, height) || !dims_ok(nwidth, nheight) || (src_type != IPK_SRC_U16 && src_type != IPK_SRC_F32))
    return fail(IPK_ERR_INVALID, "bad raw_scaled_demosaic arguments");
  ipk::Cfa cfa; DevCfa dev; int rc = get_cfa(cfa_pat, cfa, dev); if (rc) return rc;
  const int norm_fast = validate_cdiv_for_range(black0, white0 - black0, src_type == IPK_SRC_U16) ? 1 : 0;
  if (src_type == IPK_SRC_U16) ipk::launch_raw_scaled_demosaic<uint16_t>(static_cast<const uint16_t *>(src), owidth, x, y, width, height, black0, white0, norm_fast, cfa.three_colour() ? 0 : 1, nwidth, nheight, dev.cfa48, (int)cfa.width, (int)cfa.height, dst4, S(stream), 0, 0, 0);
  else ipk::launch_raw_scaled_demosaic<float>(static_cast<const float *>(src), owidth, x, y, width, height, black0, white0, norm_fast, cfa.three_colour() ? 0 : 1, nwidth, nheight, dev.cfa48, (int)cfa.width, (int)cfa.height, dst4, S(stream), 0, 0, 0);
  HIPCHK(hipGetLastError());
  return IPK_OK;
}
// one output-row band of the same (ipk_band_plan_scaled): multi-GPU sharding of a frame that OpDemosaic scales
int ipk_raw_scaled_demosaic_band(const void *src, int src_type, size_t owidth, size_t x, size_t width, size_t height, float black0, float white0,
                                 const char *cfa_pat, size_t nwidth, size_t nheight, const ipk_band *band, float *dst4, void *stream) {
  REQUIRE_INIT();
  if (!src || !dst4 || !cfa_pat || !band || !dims_ok(width, height) || !dims_ok(nwidth, nheight) || (src_type != IPK_SRC_U16 && src_type != IPK_SRC_F32))
    return fail(IPK_ERR_INVALID, "bad raw_scaled_demosaic_band arguments");
  if (band->out_rows == 0) return IPK_OK;
  if (band->out_row0 + band->out_rows > nheight || band->src_row0 + band->src_rows > height) return fail(IPK_ERR_INVALID, "band outside the frame");
  if (nheight < 2) return fail(IPK_ERR_INVALID, "a banded scaled demosaic needs at least two output rows");
  {
    // the slab must hold every source row the band's windows read (scaling.rs:84-94; the same f32 expressions as ipk_band_plan_scaled): a slab
    // that does not -- a hand-built band, a plan made for another output height -- would make the kernel read outside it
    const float skip = ((float)((int64_t)height - 1) - 0.0f) / ((float)(nheight - 1));
    const size_t from = std::min(height - 1, ipk::f32_to_usize(std::floor(0.0f + skip * (float)band->out_row0)));
    const size_t to = std::min(height - 1, ipk::f32_to_usize(std::floor(0.0f + skip * (float)(band->out_row0 + band->out_rows))));
    if (band->src_row0 > from || band->src_row0 + band->src_rows <= to)
      return fail(IPK_ERR_INVALID, "band slab rows [%zu,%zu) do not cover the window rows [%zu,%zu]", band->src_row0, band->src_row0 + band->src_rows, from, to);
  }
  ipk::Cfa cfa; DevCfa dev; int rc = get_cfa(cfa_pat, cfa, dev); if (rc) return rc;
  const int norm_fast = validate_cdiv_for_range(black0, white0 - black0, src_type == IPK_SRC_U16) ? 1 : 0;
  if (src_type == IPK_SRC_U16) ipk::launch_raw_scaled_demosaic<uint16_t>(static_cast<const uint16_t *>(src), owidth, x, 0, width, height, black0, white0, norm_fast, cfa.three_colour() ? 0 : 1, nwidth, nheight, dev.cfa48, (int)cfa.width, (int)cfa.height, dst4, S(stream), band->src_row0, band->out_row0, band->out_rows);
  else ipk::launch_raw_scaled_demosaic<float>(static_cast<const float *>(src), owidth, x, 0, width, height, black0, white0, norm_fast, cfa.three_colour() ? 0 : 1, nwidth, nheight, dev.cfa48, (int)cfa.width, (int)cfa.height, dst4, S(stream), band->src_row0, band->out_row0, band->out_rows);
  HIPCHK(hipGetLastError());
  return IPK_OK;
}

// OpGoFloat::run_other + scale_down_opbuf in one pass over a raster (used by the pipeline drivers when OpDemosaic::run would take its
// scale_down_opbuf branch for a raster source: the full-size 4-channel f32 buffer never exists)
int ipk_raster_scale_down(const void *src, int src_type, size_t owidth, size_t x, size_t y, size_t width, size_t height,
                          size_t nwidth, size_t nheight, float *dst4, void *stream) {
  REQUIRE_INIT();
  if (!src || !dst4 || !dims_ok(width, height) || !dims_ok(nwidth, nheight) || (src_type != IPK_SRC_RGB8 && src_type != IPK_SRC_RGB16))
    return fail(IPK_ERR_INVALID, "bad raster_scale_down arguments");
  if (x + width > owidth) return fail(IPK_ERR_INVALID, "raster_scale_down: window wider than the source pitch");
  ipk::launch_raster_scale_down(src, src_type == IPK_SRC_RGB16, owidth, x, y, width, height, nwidth, nheight, cx().lut_pairs[ipk::kLutGammaReverse],
                                dst4, S(stream));
  HIPCHK(hipGetLastError());
  return IPK_OK;
}

int ipk_demosaic_run(const float *src, size_t width, size_t height, size_t colors, const char *cfa_pat,
                     size_t demosaic_width, size_t demosaic_height, float *dst4, size_t *out_width, size_t *out_height, void *stream) {
  REQUIRE_INIT();
  if (colors != 1 && colors != 4) return fail(IPK_ERR_INVALID, "demosaic expects 1 or 4 colours");
  const float scale = ipk::calculate_scaling_total(width, height, demosaic_width, demosaic_height).scale;
  ipk::Cfa cfa;
  if (!ipk::Cfa::parse(cfa_pat ? cfa_pat : "", cfa)) return cfa_fail(cfa_pat);
  const float minscale = ipk::demosaic_minscale(cfa.width);
  if (scale <= 1.0f && colors == 4) { *out_width = width; *out_height = height; return IPK_NOOP; }        // demosaic.rs:41-43
  if (colors == 4) {                                                                                       // :44-46
    *out_width = demosaic_width; *out_height = demosaic_height;
    return ipk_scale_down_opbuf(src, width, height, demosaic_width, demosaic_height, dst4, stream);
  }
  if (scale >= minscale) {                                                                                 // :47-50
    *out_width = demosaic_width; *out_height = demosaic_height;
    return ipk_scaled_demosaic(src, width, height, cfa_pat, demosaic_width, demosaic_height, dst4, stream);
  }
  if (scale > 1.0f) {                                                                                      // :54-56
    Scratch sc(S(stream)); void *full = nullptr;
    int rc = sc.get(width * height * 4 * sizeof(float), &full); if (rc) return rc;
    rc = ipk_demosaic_full(src, width, height, cfa_pat, static_cast<float *>(full), stream); if (rc) return rc;
    *out_width = demosaic_width; *out_height = demosaic_height;
    return ipk_scale_down_opbuf(static_cast<float *>(full), width, height, demosaic_width, demosaic_height, dst4, stream);
  }
  *out_width = width; *out_height = height;                                                                // :57-59
  return ipk_demosaic_full(src, width, height, cfa_pat, dst4, stream);
}

int ipk_rotatecrop(const float *src, size_t width, size_t height, size_t colors, const float *p, float *dst,
                   size_t *out_width, size_t *out_height, void *stream) {
  ipk::RotateCrop rc;
  rc.crop_top = p[0]; rc.crop_right = p[1]; rc.crop_bottom = p[2]; rc.crop_left = p[3]; rc.rotation = p[4];
  int64_t pts[6]; size_t nw, nh;
  if (!rc.corners(width, height, pts, nw, nh)) { *out_width = width; *out_height = height; return IPK_NOOP; }
  *out_width = nw; *out_height = nh;
  if (!dst) return IPK_OK;
  return ipk_transform_buffer_f32(src, width, height, pts[0], pts[1], pts[2], pts[3], pts[4], pts[5], nw, nh, colors, nullptr, dst, stream);
}

int ipk_tolab(const float *src4, size_t width, size_t height, int monochrome, const float *wb_coeffs,
              const float *cam_to_xyz_normalized, float *dst3, void *stream) {
  REQUIRE_INIT();
  if (!src4 || !dst3 || !dims_ok(width, height)) return fail(IPK_ERR_INVALID, "bad tolab arguments");
  float mul[4], cm[12];
  if (monochrome) { ipk::srgb_d65_43(cm); mul[0] = mul[1] = mul[2] = mul[3] = 1.0f; }                     // colorspaces.rs:90-101
  else { std::memcpy(cm, cam_to_xyz_normalized, sizeof(cm)); ipk::normalize_wbs(wb_coeffs, mul); }
  // the fused kernels' per-pixel form (proven multiply/fma divisions, literal redo behind a wave-uniform branch) when the parameters
  // are finite and ordinary; else the literal kernel.  Round 2: 0.67 -> see DESIGN.md section 5 (the staged path of caching callers)
  bool ok = true;
  for (int i = 0; i < 4; ++i) ok = ok && std::fabs(mul[i]) <= 0x1p20f;
  for (int i = 0; i < 12; ++i) ok = ok && std::fabs(cm[i]) <= 0x1p20f;
  if (ok && width * height >= 256) {
    ipk::FusedLaunch f;
    std::memset(&f, 0, sizeof(f));
    f.src = src4; f.dst = dst3; f.mul4 = mul; f.cm12 = cm; f.rgbm9 = g_host.xyz_d65_33; f.fast_ok = 1; f.has_curve = 0; f.linear = 1;
    f.lab_table = cx().lut_plain[ipk::kLutXyzLab]; f.gam_table = cx().lut_plain[ipk::kLutGamma]; f.lab_pairs = cx().lut_pairs[ipk::kLutXyzLab]; f.gam_pairs = cx().lut_pairs[ipk::kLutGamma]; f.num_cus = cx().num_cus;
    ipk::launch_tolab_fast(f, width * height, S(stream));
  } else {
    ipk::launch_tolab(src4, width * height, mul, cm, cx().lut_pairs[ipk::kLutXyzLab], dst3, cx().num_cus, S(stream));
  }
  HIPCHK(hipGetLastError());
  return IPK_OK;
}

static int build_curve(float exposure, const float *points, int npoints, ipk::Spline &sp) {
  if (npoints < 0 || npoints > 64) return fail(IPK_ERR_INVALID, "curve with %d points (max 64)", npoints);
  float fp[128];
  const float m = std::exp2(exposure);                                                                    // curves.rs:38-41
  for (int i = 0; i < npoints; ++i) { fp[2 * i] = points[2 * i]; fp[2 * i + 1] = points[2 * i + 1] * m; }
  if (!sp.build(fp, npoints)) return fail(IPK_ERR_INVALID, "degenerate curve");
  return IPK_OK;
}
static bool curve_is_noop(float exposure, int npoints) { return npoints == 0 && std::fabs(exposure) < 0.001f; }   // curves.rs:34-36

int ipk_basecurve(const float *src3, size_t width, size_t height, float exposure, const float *points, int npoints,
                  float *dst3, void *stream) {
  REQUIRE_INIT();
  if (curve_is_noop(exposure, npoints)) return IPK_NOOP;
  if (!src3 || !dst3 || !dims_ok(width, height)) return fail(IPK_ERR_INVALID, "bad basecurve arguments");
  ipk::Spline sp; int rc = build_curve(exposure, points, npoints, sp); if (rc) return rc;
  ipk::launch_basecurve(src3, width * height, sp, dst3, cx().num_cus, S(stream));
  HIPCHK(hipGetLastError());
  return IPK_OK;
}
int ipk_fromlab(const float *src3, size_t width, size_t height, float *dst3, void *stream) {
  REQUIRE_INIT();
  if (!src3 || !dst3 || !dims_ok(width, height)) return fail(IPK_ERR_INVALID, "bad fromlab arguments");
  ipk::launch_fromlab(src3, width * height, g_host.xyz_d65_33, dst3, cx().num_cus, S(stream));
  HIPCHK(hipGetLastError());
  return IPK_OK;
}
int ipk_gamma(const float *src, size_t width, size_t height, size_t colors, int linear, float *dst, void *stream) {
  REQUIRE_INIT();
  if (linear) return IPK_NOOP;                                                                             // gamma.rs:17-18
  if (!src || !dst || !dims_ok(width, height) || colors < 1) return fail(IPK_ERR_INVALID, "bad gamma arguments");
  ipk::launch_gamma(src, width * height * colors, cx().lut_pairs[ipk::kLutGamma], dst, cx().num_cus, S(stream));
  HIPCHK(hipGetLastError());
  return IPK_OK;
}
int ipk_rotate_buffer(const float *src3, size_t bwidth, size_t bheight, int orientation, float *dst3,
                      size_t *out_width, size_t *out_height, void *stream) {
  return rotate_typed<float>(src3, bwidth, bheight, orientation, dst3, out_width, out_height, stream);
}
int ipk_rotate_image_u8(const uint8_t *src3, size_t bwidth, size_t bheight, int orientation, uint8_t *dst3,
                        size_t *out_width, size_t *out_height, void *stream) {
  return rotate_typed<uint8_t>(src3, bwidth, bheight, orientation, dst3, out_width, out_height, stream);
}
int ipk_rotate_image_u16(const uint16_t *src3, size_t bwidth, size_t bheight, int orientation, uint16_t *dst3,
                         size_t *out_width, size_t *out_height, void *stream) {
  return rotate_typed<uint16_t>(src3, bwidth, bheight, orientation, dst3, out_width, out_height, stream);
}
int ipk_transform(const float *src3, size_t width, size_t height, int rotation, int fliph, int flipv, float *dst3,
                  size_t *out_width, size_t *out_height, void *stream) {
  if (rotation < 0 || rotation > 3) return fail(IPK_ERR_INVALID, "bad rotation");
  const int o = ipk::transform_orientation(rotation, fliph != 0, flipv != 0);
  if (o == IPK_OR_NORMAL || o == IPK_OR_UNKNOWN) { *out_width = width; *out_height = height; return IPK_NOOP; }   // transform.rs:68-69
  return ipk_rotate_buffer(src3, width, height, o, dst3, out_width, out_height, stream);
}
int ipk_output8bit(const float *src, size_t n, uint8_t *dst, void *stream) {
  REQUIRE_INIT(); if (!src || !dst) return fail(IPK_ERR_INVALID, "null buffer");
  ipk::launch_output8(src, n, dst, cx().num_cus, S(stream)); HIPCHK(hipGetLastError()); return IPK_OK;
}
int ipk_output16bit(const float *src, size_t n, uint16_t *dst, void *stream) {
  REQUIRE_INIT(); if (!src || !dst) return fail(IPK_ERR_INVALID, "null buffer");
  ipk::launch_output16(src, n, dst, cx().num_cus, S(stream)); HIPCHK(hipGetLastError()); return IPK_OK;
}

// ------------------------------------------------------------------------------------------
// fused raw -> sRGB
// ------------------------------------------------------------------------------------------
// Generic-CFA cell records for a launch in rotated space: the record of rotated-space pixel (y', x') is the record of the sensor
// pixel it came from (its taps stay in the sensor's order; the kernel renames the window).  The rotated pattern has the sensor
// pattern's dimensions swapped (transposing orientations) and a phase that depends on the frame size through the flips.
static int get_rot_cells(const char *pat, const ipk::Cfa &cfa, int ori, size_t width, size_t height, const float **out) {
  bool t, fx, fy;
  ipk::orientation_to_flips(ori, t, fx, fy);
  const int pw = cfa.width, ph = cfa.height;
  char key[160];
  snprintf(key, sizeof(key), "%s|%d|%d|%d", pat, ori, (int)(width % (size_t)pw), (int)(height % (size_t)ph));
  std::lock_guard<std::mutex> lk(cx().mu);
  auto it = cx().rot_cells.find(key);
  if (it != cx().rot_cells.end()) { *out = it->second; return IPK_OK; }
  std::vector<float> cells;
  if (!cfa.gen_cells(cells)) return fail(IPK_ERR_UNSUPPORTED, "CFA has a fourth colour");
  const int rpw = t ? ph : pw, rph = t ? pw : ph;          // rotated pattern: width follows the sensor rows when transposed
  std::vector<float> rot((size_t)rpw * rph * ipk::Cfa::kGenCellFloats);
  for (int y = 0; y < rph; ++y)
    for (int x = 0; x < rpw; ++x) {
      const int64_t ro = t ? (fy ? (int64_t)height - 1 - x : x) : (fy ? (int64_t)height - 1 - y : y);
      const int64_t co = t ? (fx ? (int64_t)width - 1 - y : y) : (fx ? (int64_t)width - 1 - x : x);
      const int sy = (int)(((ro % ph) + ph) % ph), sx = (int)(((co % pw) + pw) % pw);
      std::memcpy(&rot[((size_t)y * rpw + x) * ipk::Cfa::kGenCellFloats], &cells[((size_t)sy * pw + sx) * ipk::Cfa::kGenCellFloats], ipk::Cfa::kGenCellFloats * sizeof(float));
    }
  float *dev = nullptr;
  HIPCHK(hipMalloc(reinterpret_cast<void **>(&dev), rot.size() * sizeof(float)));
  HIPCHK(hipMemcpy(dev, rot.data(), rot.size() * sizeof(float), hipMemcpyHostToDevice));
  cx().rot_cells[key] = dev;
  *out = dev;
  return IPK_OK;
}

// ori = 0: the frame as it is.  ori = Rotate90 / Rotate270 (Bayer filters, whole frames): the mosaic is first permuted into the
// output orientation (1 channel: 2 or 4 bytes per pixel instead of the 12 of the result) and the kernel works in rotated space,
// so that dst receives OpTransform's output directly; IPK_ERR_UNSUPPORTED (nothing launched) when no such variant exists.
// nbatch > 0: the frames srcs[0..nbatch) -> dsts[0..nbatch), all with the geometry and parameters of *p (src / dst = the first pair)
static int fused_impl(const ipk_fused_params *p, const void *src, void *dst, void *stream, int ori,
                      size_t nbatch = 0, const void *const *srcs = nullptr, void *const *dsts = nullptr, bool probe = false) {
  REQUIRE_INIT();
  if (!p || !src || !dst) return fail(IPK_ERR_INVALID, "null argument");
  IPK_FOLD_CFA(ipk_fused_params, p)
  if (p->src_type != IPK_SRC_U16 && p->src_type != IPK_SRC_F32) return fail(IPK_ERR_INVALID, "fused path takes u16 or f32 CFA data");
  if (!dims_ok(p->width, p->height) || p->owidth < p->x + p->width) return fail(IPK_ERR_INVALID, "bad geometry");
  if (p->out_type < 0 || p->out_type > 2) return fail(IPK_ERR_INVALID, "bad out_type");
  ipk::Cfa cfa; DevCfa dev;
  { int rc = get_cfa(p->cfa, cfa, dev); if (rc) return rc; }
  int xoff = 0, yoff = 0;
  const bool bayer = cfa.bayer_phase(xoff, yoff);
  if (!bayer && !dev.gen_cells) return fail(IPK_ERR_UNSUPPORTED, "CFA \"%s\" has a fourth colour; run the staged ops", p->cfa);

  ipk::FusedLaunch f;
  const size_t esz = p->src_type == IPK_SRC_U16 ? 2 : 4;
  const bool band = p->band_out_rows != 0;
  f.row_off = band ? p->band_src_row0 : 0;
  f.out_r0 = band ? p->band_out_row0 : 0;
  f.out_r1 = band ? p->band_out_row0 + p->band_out_rows : p->height;
  if (f.out_r1 > p->height || f.out_r0 >= f.out_r1) return fail(IPK_ERR_INVALID, "band rows outside the frame");
  if (band) {
    const size_t need0 = f.out_r0 > 0 ? f.out_r0 - 1 : 0, need1 = std::min(p->height, f.out_r1 + 1);
    if (p->band_src_row0 > need0 || p->band_src_row0 + p->band_src_rows < need1)
      return fail(IPK_ERR_INVALID, "band source rows do not cover the 1-row halo");
  }
  // whole frame: src is sensor element (0,0) -> skip the top crop; band: src already starts at sensor row y+band_src_row0
  const char *base = static_cast<const char *>(src) + ((band ? 0 : p->y * p->owidth) + p->x) * esz;
  f.src = base; f.dst = dst;
  f.src_is_u16 = p->src_type == IPK_SRC_U16;
  f.src_aligned4 = (reinterpret_cast<uintptr_t>(base) % 4 == 0) && (p->owidth % 2 == 0);
  f.batch_n = 0; f.batch_src = nullptr; f.batch_dst = nullptr;
  std::vector<const void *> bases;
  if (nbatch) {
    const size_t off = static_cast<size_t>(base - static_cast<const char *>(src));
    for (size_t i = 0; i < nbatch; ++i) {
      if (!srcs[i] || !dsts[i]) return fail(IPK_ERR_INVALID, "null frame pointer in the batch");
      bases.push_back(static_cast<const char *>(srcs[i]) + off);
      f.src_aligned4 = f.src_aligned4 && reinterpret_cast<uintptr_t>(bases.back()) % 4 == 0;
    }
    f.batch_n = (int)nbatch; f.batch_src = bases.data(); f.batch_dst = dsts;
  }
  f.width = p->width; f.height = p->height; f.owidth = p->owidth;
  f.ori = 0; f.roles[0] = f.roles[1] = f.roles[2] = f.roles[3] = 0;
  Scratch rot(S(stream));
  if (ori != 0) {
    bool t, fx, fy;
    ipk::orientation_to_flips(ori, t, fx, fy);
    if (band || ori < 1 || ori > 7 || (t ? p->height : p->width) < 256)               // the rotated frame must be at least one 256-pixel strip wide
      return fail(IPK_ERR_UNSUPPORTED, "no rotated-space variant for this frame");
    // rotate_buffer's index walk (transform.rs:102-128) over the crop window of the pitched 1-channel source
    int64_t width = (int64_t)p->width, height = (int64_t)p->height, x_step = 1, y_step = (int64_t)p->owidth, off = 0;
    if (fx) { x_step = -x_step; off += width - 1; }
    if (fy) { y_step = -y_step; off += (int64_t)p->owidth * (height - 1); }
    if (t) { std::swap(width, height); std::swap(x_step, y_step); }
    void *m = nullptr;
    int rc = rot.get((size_t)width * (size_t)height * esz, &m); if (rc) return rc;
    if (f.src_is_u16) ipk::launch_rotate1<uint16_t>(reinterpret_cast<const uint16_t *>(base), (size_t)width, (size_t)height, off, x_step, y_step, static_cast<uint16_t *>(m), S(stream));
    else ipk::launch_rotate1<float>(reinterpret_cast<const float *>(base), (size_t)width, (size_t)height, off, x_step, y_step, static_cast<float *>(m), S(stream));
    HIPCHK(hipGetLastError());
    // the demosaic role of a rotated-space pixel = the role of the sensor pixel it came from: depends on the parities only
    for (int pr = 0; pr < 2; ++pr) for (int pc = 0; pc < 2; ++pc) {
      const int64_t ro = t ? (fy ? (int64_t)p->height - 1 - pc : pc) : (fy ? (int64_t)p->height - 1 - pr : pr);
      const int64_t co = t ? (fx ? (int64_t)p->width - 1 - pr : pr) : (fx ? (int64_t)p->width - 1 - pc : pc);
      f.roles[2 * pr + pc] = (int)((((ro + yoff) & 1) << 1) | ((co + xoff) & 1));
    }
    f.ori = ori;
    f.src = m; f.src_aligned4 = (width % 2 == 0);
    f.width = (size_t)width; f.height = (size_t)height; f.owidth = (size_t)width;
    f.out_r0 = 0; f.out_r1 = (size_t)height; f.row_off = 0;
    xoff = 0; yoff = 0;
  }
  f.black0 = p->black0; f.white0 = p->white0;
  f.exact_norm = validate_cdiv_for_range(p->black0, p->white0 - p->black0, f.src_is_u16) ? 0 : 1;
  f.xoff = xoff; f.yoff = yoff;
  f.gen_cells = bayer ? nullptr : dev.gen_cells; f.gen_pw = dev.gen_pw; f.gen_ph = dev.gen_ph;
  if (ori != 0 && !bayer) {                                // rotated space: the pattern's dimensions swap, the records follow the sensor pixels
    const float *rc_dev = nullptr;
    int rc = get_rot_cells(p->cfa, cfa, ori, p->width, p->height, &rc_dev); if (rc) return rc;
    bool tt, fxx, fyy; ipk::orientation_to_flips(ori, tt, fxx, fyy);
    f.gen_cells = rc_dev;
    if (tt) { f.gen_pw = dev.gen_ph; f.gen_ph = dev.gen_pw; }
  }
  // generic-CFA mode sums up to nine normalised samples and divides by a constant: u16 kernels check their samples only
  // when the levels allow one outside [2^-60, 2^60] (f32 kernels always check)
  f.gen_check = (!bayer && f.src_is_u16 && !gen_levels_ok_u16(p->black0, p->white0 - p->black0)) ? 1 : 0;
  float mul[4];
  ipk::normalize_wbs(p->wb_coeffs, mul);                                                                   // colorspaces.rs:100
  f.mul4 = mul; f.cm12 = p->cam_to_xyz_normalized; f.rgbm9 = g_host.xyz_d65_33;
  // the fast point-wise form assumes finite, ordinary parameters (see pointwise4_fast): |value| <= 2^20, no NaN/inf
  {
    auto sane = [](float v) { return std::fabs(v) <= 0x1p20f; };        // false for NaN and inf
    bool ok = true;
    for (int i = 0; i < 4; ++i) ok = ok && sane(mul[i]);
    for (int i = 0; i < 12; ++i) ok = ok && sane(p->cam_to_xyz_normalized[i]);
    f.fast_ok = ok ? 1 : 0;
    // when the samples (u16: all 65 536 walked here; f32: bounded below through the black level), the multipliers and the matrix
    // are ordinary (see pointwise4_fast in ipk_kernels.hip), the kernel variant without per-pixel input guards is legal
    const float range = p->white0 - p->black0;
    bool plain = ok && bayer && range > 0.0f;
    if (f.src_is_u16) plain = plain && gen_levels_ok_u16(p->black0, range);                       // every nonzero sample >= 2^-20 in magnitude
    else plain = plain && std::fabs(p->black0) >= range * 0x1p-6f && std::fabs(p->black0) <= 0x1p70f;   // every nonzero sample >= 2^-31
    for (int i = 0; i < 3; ++i) plain = plain && mul[i] >= 0x1p-4f && mul[i] <= 0x1p10f;
    for (int i = 0; i < 12; ++i) { const float c = std::fabs(p->cam_to_xyz_normalized[i]); plain = plain && (c == 0.0f || (c >= 0x1p-12f && c <= 0x1p20f)); }
    f.px_guard = plain ? 0 : 1;
  }
  ipk::Spline sp;
  f.has_curve = !curve_is_noop(p->exposure, p->npoints);
  if (f.has_curve) {
    int rc = build_curve(p->exposure, p->points, p->npoints, sp); if (rc) return rc;
    for (int i = 0; i < sp.npoints; ++i) if (!(std::fabs(sp.px[i]) <= 0x1p20f && std::fabs(sp.py[i]) <= 0x1p20f && std::fabs(sp.c1[i]) <= 0x1p40f)) f.fast_ok = 0;
    for (int i = 0; i < sp.nseg; ++i) if (!(std::fabs(sp.c2[i]) <= 0x1p40f && std::fabs(sp.c3[i]) <= 0x1p40f)) f.fast_ok = 0;
  }
  f.spline = &sp;
  f.linear = p->linear;
  f.out_type = probe ? 4 : p->out_type;
  f.lab_table = cx().lut_plain[ipk::kLutXyzLab]; f.gam_table = cx().lut_plain[ipk::kLutGamma]; f.lab_pairs = cx().lut_pairs[ipk::kLutXyzLab]; f.gam_pairs = cx().lut_pairs[ipk::kLutGamma];
  f.num_cus = cx().num_cus; f.queues = cx().queues; f.gam_q8 = cx().lut_q8;
  if (p->schedule != IPK_SCHED_AUTO && p->schedule != IPK_SCHED_SPLIT) return fail(IPK_ERR_INVALID, "bad schedule");
  f.schedule = p->schedule;
  { const int lrc = ipk::launch_fused_bayer(f, S(stream));
    if (lrc == -4) return fail(IPK_ERR_HIP, "kernel launch failed (nothing was enqueued; the stream's task queue is untouched)");
    if (lrc != 0) return fail(IPK_ERR_UNSUPPORTED, probe ? "the stream probe exists for Bayer frames of 256+ columns with validated levels" : "no rotated-space variant for these parameters"); }
  HIPCHK(hipGetLastError());
  return IPK_OK;
}
int ipk_raw_to_srgb(const ipk_fused_params *p, const void *src, void *dst, void *stream) { return fused_impl(p, src, dst, stream, 0); }
// Measurement aid: the fused kernel's memory skeleton (its launch, task walk, row loads, OpGoFloat, demosaic::full, LDS staging, nontemporal stores) without
// the point-wise stages -- dst receives the demosaiced R, G, B as width*rows*3 f32.  bench.py times it next to ipk_raw_to_srgb (roofline.ceiling_ms).
int ipk_stream_probe(const ipk_fused_params *p, const void *src, void *dst, void *stream) {
  IPK_FOLD_CFA(ipk_fused_params, p)
  if (p && p->band_out_rows != 0) return fail(IPK_ERR_INVALID, "the stream probe takes whole frames");
  return fused_impl(p, src, dst, stream, 0, 0, nullptr, nullptr, true);
}
// A batch of same-shaped frames through one descriptor: Pipeline::run over a shoot.  One persistent launch per 64 frames where the
// kernel has a batch variant (ordinary Bayer parameters), one launch per frame otherwise -- the results are the single-frame ones.
int ipk_raw_to_srgb_batch(const ipk_fused_params *p, const void *const *srcs, void *const *dsts, size_t n, void *stream) {
  if (!p || (n && (!srcs || !dsts))) return fail(IPK_ERR_INVALID, "null argument");
  if (n == 0) return IPK_OK;
  if (n > (size_t)1 << 20) return fail(IPK_ERR_INVALID, "batch too large");
  IPK_FOLD_CFA(ipk_fused_params, p)
  if (p->band_out_rows != 0) return fail(IPK_ERR_INVALID, "a batch takes whole frames, not bands");
  return fused_impl(p, srcs[0], dsts[0], stream, 0, n, srcs, dsts);
}
int ipk_raw_to_srgb_oriented(const ipk_fused_params *p, const void *src, int orientation, void *dst, size_t *out_width, size_t *out_height, void *stream) {
  if (!p || !out_width || !out_height) return fail(IPK_ERR_INVALID, "null argument");
  IPK_FOLD_CFA(ipk_fused_params, p)
  if (orientation == IPK_OR_NORMAL || orientation == IPK_OR_UNKNOWN) { *out_width = p->width; *out_height = p->height; return fused_impl(p, src, dst, stream, 0); }
  if (orientation < 0 || orientation > 8) return fail(IPK_ERR_INVALID, "bad orientation");
  { bool t, fx, fy; ipk::orientation_to_flips(orientation, t, fx, fy); *out_width = t ? p->height : p->width; *out_height = t ? p->width : p->height; }
  return fused_impl(p, src, dst, stream, orientation);
}

// OpToLab::run + OpBaseCurve::run + OpFromLab::run + OpGamma::run (colorspaces.rs:89-112, curves.rs:33-49, colorspaces.rs:127-137,
// gamma.rs:16-26) in one pass: what Pipeline::run computes between rotatecrop and transform when no cache needs the
// intermediate buffers.
namespace {
// what tolab..gamma need besides the pixels: normalised multipliers, the matrix, the curve, the tables -- and whether all of it
// is ordinary enough for the fast point-wise form (pointwise4_fast; otherwise the kernels evaluate the literal form)
struct PointwisePrep {
  float mul[4], cm[12];
  ipk::Spline sp;
  ipk::FusedLaunch f;
  int prepare(int monochrome, const float *wb_coeffs, const float *cam_to_xyz_normalized, float exposure, const float *points, int npoints, int linear) {
    if (npoints < 0 || npoints > 64 || (npoints > 0 && !points)) return fail(IPK_ERR_INVALID, "npoints out of range");
    if (monochrome) { ipk::srgb_d65_43(cm); mul[0] = mul[1] = mul[2] = mul[3] = 1.0f; }                     // colorspaces.rs:90-101
    else { std::memcpy(cm, cam_to_xyz_normalized, sizeof(cm)); ipk::normalize_wbs(wb_coeffs, mul); }
    std::memset(&f, 0, sizeof(f));
    f.mul4 = mul; f.cm12 = cm; f.rgbm9 = g_host.xyz_d65_33;
    auto sane = [](float v) { return std::fabs(v) <= 0x1p20f; };
    bool ok = true;
    for (int i = 0; i < 4; ++i) ok = ok && sane(mul[i]);
    for (int i = 0; i < 12; ++i) ok = ok && sane(cm[i]);
    f.fast_ok = ok ? 1 : 0;
    f.has_curve = !curve_is_noop(exposure, npoints);
    if (f.has_curve) {
      int rc = build_curve(exposure, points, npoints, sp); if (rc) return rc;
      for (int i = 0; i < sp.npoints; ++i) if (!(std::fabs(sp.px[i]) <= 0x1p20f && std::fabs(sp.py[i]) <= 0x1p20f && std::fabs(sp.c1[i]) <= 0x1p40f)) f.fast_ok = 0;
      for (int i = 0; i < sp.nseg; ++i) if (!(std::fabs(sp.c2[i]) <= 0x1p40f && std::fabs(sp.c3[i]) <= 0x1p40f)) f.fast_ok = 0;
    }
    f.spline = &sp;
    f.linear = linear;
    f.lab_table = cx().lut_plain[ipk::kLutXyzLab]; f.gam_table = cx().lut_plain[ipk::kLutGamma]; f.lab_pairs = cx().lut_pairs[ipk::kLutXyzLab]; f.gam_pairs = cx().lut_pairs[ipk::kLutGamma];
    f.num_cus = cx().num_cus; f.gam_q8 = cx().lut_q8;
    return IPK_OK;
  }
};
}  // namespace

int ipk_pointwise_chain(const float *src4, size_t width, size_t height, int monochrome, const float *wb_coeffs, const float *cam_to_xyz_normalized,
                        float exposure, const float *points, int npoints, int linear, float *dst3, void *stream) {
  REQUIRE_INIT();
  if (!src4 || !dst3 || !wb_coeffs || !cam_to_xyz_normalized || !dims_ok(width, height)) return fail(IPK_ERR_INVALID, "bad pointwise_chain arguments");
  PointwisePrep pp;
  int rc = pp.prepare(monochrome, wb_coeffs, cam_to_xyz_normalized, exposure, points, npoints, linear); if (rc) return rc;
  pp.f.src = src4; pp.f.dst = dst3;
  ipk::launch_pointwise_chain(pp.f, width * height, S(stream));
  HIPCHK(hipGetLastError());
  return IPK_OK;
}

// the same followed by output8bit / output16bit (src/pipeline.rs:408-414, :455-461) in one pass: what output_8bit / output_16bit compute behind a staged
// demosaic (a preview under a size limit, an X-Trans or four-colour frame) without the f32 image in between
int ipk_pointwise_chain_out(const float *src4, size_t width, size_t height, int monochrome, const float *wb_coeffs, const float *cam_to_xyz_normalized,
                            float exposure, const float *points, int npoints, int linear, int out_type, void *dst, void *stream) {
  REQUIRE_INIT();
  if (!src4 || !dst || !wb_coeffs || !cam_to_xyz_normalized || !dims_ok(width, height)) return fail(IPK_ERR_INVALID, "bad pointwise_chain_out arguments");
  if (out_type == IPK_OUT_F32) return ipk_pointwise_chain(src4, width, height, monochrome, wb_coeffs, cam_to_xyz_normalized, exposure, points, npoints, linear, static_cast<float *>(dst), stream);
  if (out_type != IPK_OUT_U8 && out_type != IPK_OUT_U16) return fail(IPK_ERR_INVALID, "bad out_type");
  if (width * height < 256) return fail(IPK_ERR_UNSUPPORTED, "pointwise_chain_out needs at least 256 pixels (use ipk_pointwise_chain + ipk_output8bit / 16bit)");
  PointwisePrep pp;
  int rc = pp.prepare(monochrome, wb_coeffs, cam_to_xyz_normalized, exposure, points, npoints, linear); if (rc) return rc;
  pp.f.src = src4; pp.f.dst = dst; pp.f.out_type = out_type;
  if (ipk::launch_chain_quantised(pp.f, width * height, S(stream)) != 0) return fail(IPK_ERR_UNSUPPORTED, "pointwise_chain_out: frame too small");
  HIPCHK(hipGetLastError());
  return IPK_OK;
}

int ipk_raster_to_srgb(const void *src, int src_type, size_t width, size_t height, const float *wb_coeffs, const float *cam_to_xyz_normalized,
                       float exposure, const float *points, int npoints, int linear, int out_type, void *dst, void *stream) {
  REQUIRE_INIT();
  if (!src || !dst || !wb_coeffs || !cam_to_xyz_normalized || !dims_ok(width, height)) return fail(IPK_ERR_INVALID, "bad raster_to_srgb arguments");
  if (src_type != IPK_SRC_RGB8 && src_type != IPK_SRC_RGB16) return fail(IPK_ERR_INVALID, "raster_to_srgb takes RGB8 or RGB16 sources");
  if (out_type < 0 || out_type > 2) return fail(IPK_ERR_INVALID, "bad out_type");
  if (width * height < 256) return fail(IPK_ERR_UNSUPPORTED, "raster_to_srgb needs at least 256 pixels (use the staged ops)");
  PointwisePrep pp;
  int rc = pp.prepare(0, wb_coeffs, cam_to_xyz_normalized, exposure, points, npoints, linear); if (rc) return rc;
  pp.f.src = src; pp.f.dst = dst; pp.f.out_type = out_type;
  if (ipk::launch_raster_chain(pp.f, width * height, src_type == IPK_SRC_RGB16, cx().lut_pairs[ipk::kLutGammaReverse], S(stream)) != 0)
    return fail(IPK_ERR_UNSUPPORTED, "raster_to_srgb: frame too small");
  HIPCHK(hipGetLastError());
  return IPK_OK;
}

// ------------------------------------------------------------------------------------------
// self-test hooks
// ------------------------------------------------------------------------------------------
static int selftest_collect(void *dev, uint64_t *n_bad, uint32_t *first_bad) {
  struct { unsigned long long bad; unsigned int first; unsigned int pad; } h;
  HIPCHK(hipMemcpy(&h, dev, sizeof(h), hipMemcpyDeviceToHost));
  (void)hipFree(dev);
  *n_bad = h.bad; if (first_bad) *first_bad = h.first;
  return IPK_OK;
}
static int selftest_alloc(void **dev) {
  struct { unsigned long long bad; unsigned int first; unsigned int pad; } h = {0, 0xFFFFFFFFu, 0};
  HIPCHK(hipMalloc(dev, sizeof(h)));
  HIPCHK(hipMemcpy(*dev, &h, sizeof(h), hipMemcpyHostToDevice));
  return IPK_OK;
}
int ipk_selftest_cdiv(float c, int variant, float lo, float hi, int include_special, uint64_t *n_bad, uint32_t *first_bad_bits) {
  REQUIRE_INIT();
  if (!(c > 0.0f) || variant < 0 || variant > 2 || !n_bad) return fail(IPK_ERR_INVALID, "bad selftest arguments");
  uint32_t lo_bits, hi_bits; std::memcpy(&lo_bits, &lo, 4); std::memcpy(&hi_bits, &hi, 4);
  void *dev; int rc = selftest_alloc(&dev); if (rc) return rc;
  ipk::launch_selftest_cdiv(c, variant, lo_bits, hi_bits, include_special, dev, nullptr);
  HIPCHK(hipGetLastError());
  return selftest_collect(dev, n_bad, first_bad_bits);
}
int ipk_selftest_lut_weight(uint64_t *n_bad, uint32_t *first_bad_bits) {
  REQUIRE_INIT();
  void *dev; int rc = selftest_alloc(&dev); if (rc) return rc;
  ipk::launch_selftest_fract(dev, nullptr); HIPCHK(hipGetLastError());
  return selftest_collect(dev, n_bad, first_bad_bits);
}
int ipk_selftest_clamp01(uint64_t *n_bad, uint32_t *first_bad_bits) {
  REQUIRE_INIT();
  void *dev; int rc = selftest_alloc(&dev); if (rc) return rc;
  ipk::launch_selftest_clamp(dev, nullptr); HIPCHK(hipGetLastError());
  return selftest_collect(dev, n_bad, first_bad_bits);
}
int ipk_copy_probe(const void *src, void *dst, size_t bytes, void *stream) {
  REQUIRE_INIT();
  if (!src || !dst || (bytes & 15) || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return fail(IPK_ERR_INVALID, "ipk_copy_probe wants 16-byte aligned buffers and a multiple of 16 bytes");
  ipk::launch_copy_probe(src, dst, bytes, cx().num_cus, S(stream)); HIPCHK(hipGetLastError());
  return IPK_OK;
}
int ipk_mix_probe(const void *src, void *dst, size_t src_bytes, void *stream) {
  REQUIRE_INIT();
  if (!src || !dst || (src_bytes & 4095) || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return fail(IPK_ERR_INVALID, "ipk_mix_probe wants 16-byte aligned buffers and a multiple of 4096 source bytes (whole blocks)");
  ipk::launch_mix_probe(src, dst, src_bytes, S(stream)); HIPCHK(hipGetLastError());
  return IPK_OK;
}
int ipk_clock_probe(uint64_t *out2_dev, uint32_t spin_us, void *stream) {
  REQUIRE_INIT();
  if (!out2_dev || spin_us == 0 || spin_us > 2000000u) return fail(IPK_ERR_INVALID, "ipk_clock_probe: null output or a spin outside (0, 2 s]");
  ipk::launch_clock_probe(out2_dev, (unsigned long long)spin_us * 100ull, S(stream)); HIPCHK(hipGetLastError());
  return IPK_OK;
}
int ipk_selftest_task_queue(int enabled) { REQUIRE_INIT(); ipk::selftest_task_queue(enabled != 0); return IPK_OK; }
int ipk_selftest_spline3(float exposure, const float *points, int npoints, uint64_t *n_bad, uint32_t *first_bad_bits) {
  REQUIRE_INIT();
  if (!n_bad || (npoints > 0 && !points)) return fail(IPK_ERR_INVALID, "bad selftest arguments");
  ipk::Spline sp; int rc = build_curve(exposure, points, npoints, sp); if (rc) return rc;
  void *dev; rc = selftest_alloc(&dev); if (rc) return rc;
  if (ipk::launch_selftest_spline3(sp, dev, nullptr) != 0) { (void)hipFree(dev); return fail(IPK_ERR_UNSUPPORTED, "the kernels keep the select form for this curve"); }
  HIPCHK(hipGetLastError());
  return selftest_collect(dev, n_bad, first_bad_bits);
}
int ipk_selftest_quant16(uint64_t *n_bad, uint32_t *first_bad_bits) {
  REQUIRE_INIT();
  if (!n_bad) return fail(IPK_ERR_INVALID, "bad selftest arguments");
  void *dev; int rc = selftest_alloc(&dev); if (rc) return rc;
  ipk::launch_selftest_quant16(dev, nullptr); HIPCHK(hipGetLastError());
  return selftest_collect(dev, n_bad, first_bad_bits);
}
int ipk_selftest_q8(uint64_t *n_bad, uint32_t *first_bad_bits) {
  REQUIRE_INIT();
  if (!n_bad) return fail(IPK_ERR_INVALID, "bad selftest arguments");
  void *dev; int rc = selftest_alloc(&dev); if (rc) return rc;
  ipk::launch_selftest_q8(cx().lut_pairs[ipk::kLutGamma], cx().lut_q8, dev, nullptr); HIPCHK(hipGetLastError());
  return selftest_collect(dev, n_bad, first_bad_bits);
}
int ipk_selftest_quant8(int variant, uint64_t *n_bad, uint32_t *first_bad_bits) {
  REQUIRE_INIT();
  void *dev; int rc = selftest_alloc(&dev); if (rc) return rc;
  ipk::launch_selftest_quant8(dev, variant, nullptr); HIPCHK(hipGetLastError());
  return selftest_collect(dev, n_bad, first_bad_bits);
}
int ipk_selftest_cbrtf(const float *in, float *out, size_t n, int variant, void *stream) {
  REQUIRE_INIT();
  if (!in || !out || variant < 0 || variant > 2) return fail(IPK_ERR_INVALID, "bad selftest arguments");
  ipk::launch_selftest_cbrt(in, out, n, variant, S(stream)); HIPCHK(hipGetLastError());
  return IPK_OK;
}

// ------------------------------------------------------------------------------------------
// Pipeline::run (src/pipeline.rs:311-375) for one source, cache == None
// ------------------------------------------------------------------------------------------
// rc_state (may be null): the OpRotateCrop as the two folds of the negotiation leave it (input_ratio, output size) -- the state
// its Serialize impl exposes to the hash chain (pipeline.rs:318-335, rotatecrop.rs:10-18)
static int pipeline_sizes_impl(const ipk_pipeline_desc *d, size_t *demosaic_w, size_t *demosaic_h, size_t *final_w, size_t *final_h,
                               ipk::RotateCrop *rc_state) {
  if (!d) return fail(IPK_ERR_INVALID, "null descriptor");
  if (d->rotation < 0 || d->rotation > 3) return fail(IPK_ERR_INVALID, "bad rotation");
  ipk::RotateCrop rc;                                                     // reset() state (pipeline.rs:314-316)
  rc.crop_top = d->rotatecrop[0]; rc.crop_right = d->rotatecrop[1]; rc.crop_bottom = d->rotatecrop[2];
  rc.crop_left = d->rotatecrop[3]; rc.rotation = d->rotatecrop[4];
  ipk::Rect r;
  if (!ipk::size_image(d->crop_top, d->crop_right, d->crop_bottom, d->crop_left, d->width, d->height, r))
    return fail(IPK_ERR_INVALID, "source smaller than 10x10");
  size_t w = r.width, h = r.height;
  rc.transform_forward(w, h, w, h);                                       // forward fold (pipeline.rs:318-324)
  ipk::transform_forward(d->rotation, w, h, w, h);
  const ipk::Scaling s = ipk::calculate_scaling_total(w, h, d->maxwidth, d->maxheight);   // :328-329
  w = s.width; h = s.height;
  ipk::transform_forward(d->rotation, w, h, w, h);                        // reverse fold (:331-335)
  rc.transform_reverse(w, h, w, h);
  if (rc_state) *rc_state = rc;
  *demosaic_w = w; *demosaic_h = h;
  // The size run() PRODUCES (what output_8bit reports and tests/maxsize_test.rs asserts on): every op sizes its output from
  // the buffer it is handed, not from the negotiation, so behind a rotatecrop the result can differ from the forward fold by a
  // pixel.  demosaic.rs:27-61: the demosaic size when it scales, else its input; rotatecrop.rs:39-64: calc_size of its input
  // unless it is a no-op or rejects its crops; transform.rs:56-73: sides swapped by the transposing orientations.
  size_t pw = r.width, ph = r.height;
  if (ipk::calculate_scaling_total(pw, ph, w, h).scale > 1.0f) { pw = w; ph = h; }
  {
    ipk::RotateCrop run_rc;
    run_rc.crop_top = d->rotatecrop[0]; run_rc.crop_right = d->rotatecrop[1]; run_rc.crop_bottom = d->rotatecrop[2];
    run_rc.crop_left = d->rotatecrop[3]; run_rc.rotation = d->rotatecrop[4];
    int64_t pts[6]; size_t nw, nh;
    if (run_rc.corners(pw, ph, pts, nw, nh)) { pw = nw; ph = nh; }
  }
  {
    bool transpose, fx, fy;
    ipk::orientation_to_flips(ipk::transform_orientation(d->rotation, d->fliph != 0, d->flipv != 0), transpose, fx, fy);
    if (transpose) std::swap(pw, ph);
  }
  *final_w = pw; *final_h = ph;
  return IPK_OK;
}
int ipk_pipeline_sizes(const ipk_pipeline_desc *d, size_t *demosaic_w, size_t *demosaic_h, size_t *final_w, size_t *final_h) {
  IPK_FOLD_CFA(ipk_pipeline_desc, d)
  return pipeline_sizes_impl(d, demosaic_w, demosaic_h, final_w, final_h, nullptr);
}

// Pipeline::default_ops (pipeline.rs:286-288) for a raster source: every user-visible op field equals PipelineOps::new(Other),
// bit for bit (the reference compares serialised bytes).  Its OpRotateCrop also serialises negotiated state that a previous
// slow-path run leaves behind; the descriptor is stateless, so that quirk is not reproduced.
static bool default_ops_other(const ipk_pipeline_desc *d) {
  auto same = [](float a, float b) { return std::memcmp(&a, &b, 4) == 0; };
  if (d->src_type != IPK_SRC_RGB8 && d->src_type != IPK_SRC_RGB16) return false;
  if (d->crop_top || d->crop_right || d->crop_bottom || d->crop_left || d->is_cfa || d->cfa[0]) return false;
  for (int i = 0; i < 4; ++i) if (!same(d->blacklevels[i], 0.0f) || !same(d->whitelevels[i], 0.0f)) return false;
  for (int i = 0; i < 5; ++i) if (!same(d->rotatecrop[i], 0.0f)) return false;
  float m[12]; ipk::srgb_d65_43(m);
  for (int i = 0; i < 12; ++i) if (!same(d->cam_to_xyz_normalized[i], m[i])) return false;
  const float wb[4] = {1.0f, 1.0f, 1.0f, 0.0f};
  for (int i = 0; i < 4; ++i) if (!same(d->wb_coeffs[i], wb[i])) return false;
  if (!same(d->exposure, 0.0f) || d->npoints != 0) return false;
  return d->rotation == 0 && !d->fliph && !d->flipv;
}
int ipk_pipeline_takes_fastpath(const ipk_pipeline_desc *d, int out_type) {
  if (!d) return fail(IPK_ERR_INVALID, "null descriptor");
  ipk_pipeline_desc taken;
  { const int rc = take_desc(d, taken); if (rc) return rc; }
  d = &taken;
  return (d->use_fastpath && (out_type == IPK_OUT_U8 || out_type == IPK_OUT_U16) && default_ops_other(d)) ? 1 : 0;
}
// output_8bit / output_16bit fast path (pipeline.rs:381-402, :428-449) on device buffers
static int run_fastpath(const ipk_pipeline_desc *d, const void *src, void *dst, int out_type, void *stream) {
  const size_t n = d->width * d->height * 3;
  const ipk::Scaling sc = ipk::calculate_scaling_total(d->width, d->height, d->maxwidth, d->maxheight);      // scaling_size
  const bool scaled = sc.width != d->width || sc.height != d->height;
  const bool want16 = out_type == IPK_OUT_U16, have16 = d->src_type == IPK_SRC_RGB16;
  const size_t esz = want16 ? 2 : 1;
  Scratch tmp(S(stream));
  const void *rgb = src;
  if (want16 != have16) {                                                  // to_rgb8 / to_rgb16
    void *conv = dst;
    if (scaled) { int rc = tmp.get(n * esz, &conv); if (rc) return rc; }
    if (want16) ipk::launch_chan_8_to_16(static_cast<const uint8_t *>(src), n, static_cast<uint16_t *>(conv), cx().num_cus, S(stream));
    else ipk::launch_chan_16_to_8(static_cast<const uint16_t *>(src), n, static_cast<uint8_t *>(conv), cx().num_cus, S(stream));
    HIPCHK(hipGetLastError());
    rgb = conv;
  }
  if (!scaled) {
    if (rgb != dst) HIPCHK(hipMemcpyAsync(dst, rgb, n * esz, hipMemcpyDeviceToDevice, S(stream)));
    return IPK_OK;
  }
  // scale_down_srgb / scale_down_srgb16 (scaling.rs:162-182)
  return want16
    ? ipk_transform_buffer_u16(static_cast<const uint16_t *>(rgb), d->width, d->height, 0, 0, (int64_t)d->width - 1, 0, 0, (int64_t)d->height - 1,
                               sc.width, sc.height, 3, nullptr, static_cast<uint16_t *>(dst), stream)
    : ipk_transform_buffer_u8(static_cast<const uint8_t *>(rgb), d->width, d->height, 0, 0, (int64_t)d->width - 1, 0, 0, (int64_t)d->height - 1,
                              sc.width, sc.height, 3, nullptr, static_cast<uint8_t *>(dst), stream);
}

// ---- do_timing! (src/pipeline.rs:68-80): per-stage times of the pipeline driver, hipEvents on the caller's stream ----------------
namespace {
struct TimingSession { bool active = false; std::vector<std::pair<std::string, hipEvent_t>> marks; };
thread_local TimingSession g_timing;
struct StageTimer {                    // marks are no-ops unless ipk_timing_begin() armed this thread
  hipStream_t s; bool on; const char *rest = nullptr;
  explicit StageTimer(hipStream_t st) : s(st), on(g_timing.active) { if (on) mark("(start)"); }
  void mark(const char *name) {
    if (!on) return;
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess || hipEventRecord(e, s) != hipSuccess) { on = false; return; }
    g_timing.marks.emplace_back(name, e);
  }
  ~StageTimer() { if (rest) mark(rest); }
};
}  // namespace
int ipk_timing_begin(void) {
  for (auto &m : g_timing.marks) (void)hipEventDestroy(m.second);
  g_timing.marks.clear();
  g_timing.active = true;
  return IPK_OK;
}
int ipk_timing_end(ipk_stage_time *out, int max_stages, int *n_stages) {
  // the session ends whatever the arguments are: a bad call must not leave the thread armed with its events alive
  g_timing.active = false;
  auto &mk = g_timing.marks;
  if (!n_stages || (max_stages > 0 && !out)) {
    for (auto &m : mk) (void)hipEventDestroy(m.second);
    mk.clear();
    return fail(IPK_ERR_INVALID, "null argument");
  }
  int n = 0, rc = IPK_OK;
  if (!mk.empty() && hipEventSynchronize(mk.back().second) != hipSuccess) rc = fail(IPK_ERR_HIP, "hipEventSynchronize failed");
  for (size_t i = 1; i < mk.size() && rc == IPK_OK; ++i) {
    if (mk[i].first == "(start)") continue;                       // a second pipeline call inside one session starts a new sequence
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, mk[i - 1].second, mk[i].second) != hipSuccess) { rc = fail(IPK_ERR_HIP, "hipEventElapsedTime failed"); break; }
    if (n < max_stages) { std::snprintf(out[n].name, sizeof(out[n].name), "%s", mk[i].first.c_str()); out[n].ms = ms; }
    ++n;
  }
  for (auto &m : mk) (void)hipEventDestroy(m.second);
  mk.clear();
  *n_stages = n;
  return rc;
}

int ipk_pipeline_run(const ipk_pipeline_desc *d, const void *src, void *dst, int out_type, int *used_fused, void *stream) {
  REQUIRE_INIT();
  if (!d || !src || !dst) return fail(IPK_ERR_INVALID, "null argument");
  if (out_type < 0 || out_type > 2) return fail(IPK_ERR_INVALID, "bad out_type");
  IPK_FOLD_CFA(ipk_pipeline_desc, d)
  if (ipk_pipeline_takes_fastpath(d, out_type) == 1) {
    if (used_fused) *used_fused = 0;
    if (d->width < 1 || d->height < 1) return fail(IPK_ERR_INVALID, "empty source");
    StageTimer tmf(S(stream)); tmf.rest = "fastpath";
    return run_fastpath(d, src, dst, out_type, stream);
  }
  size_t dw, dh, fw, fh;
  int rc = ipk_pipeline_sizes(d, &dw, &dh, &fw, &fh); if (rc) return rc;
  StageTimer tm(S(stream));
  // output_8bit forces linear=false (pipeline.rs:405), output_16bit linear=true (:452)
  const int linear = out_type == IPK_OUT_U8 ? 0 : (out_type == IPK_OUT_U16 ? 1 : d->linear);
  ipk::Rect r;
  ipk::size_image(d->crop_top, d->crop_right, d->crop_bottom, d->crop_left, d->width, d->height, r);
  const bool raw = d->src_type == IPK_SRC_U16 || d->src_type == IPK_SRC_F32;
  const bool cfa_branch = raw && !(d->cpp == 1 && !d->is_cfa) && d->cpp != 3;          // gofloat.rs:95,109,121
  const int orientation = ipk::transform_orientation(d->rotation, d->fliph != 0, d->flipv != 0);
  const bool transform_noop = orientation == IPK_OR_NORMAL || orientation == IPK_OR_UNKNOWN;
  ipk::RotateCrop rcop;
  rcop.crop_top = d->rotatecrop[0]; rcop.crop_right = d->rotatecrop[1]; rcop.crop_bottom = d->rotatecrop[2];
  rcop.crop_left = d->rotatecrop[3]; rcop.rotation = d->rotatecrop[4];

  // ---- fused path: legal when every op between gofloat and gamma is point-wise or demosaic::full ----
  if (used_fused) *used_fused = 0;
  if (d->allow_fused && cfa_branch && d->cpp == 1 && rcop.noop()) {
    const float scale = ipk::calculate_scaling_total(r.width, r.height, dw, dh).scale;
    ipk::Cfa cfa; int xo, yo;
    if (scale <= 1.0f && ipk::Cfa::parse(d->cfa, cfa) && (cfa.bayer_phase(xo, yo) || cfa.three_colour())) {
      ipk_fused_params fp;
      std::memset(&fp, 0, sizeof(fp));
      fp.struct_size = (uint32_t)sizeof(fp);
      fp.src_type = d->src_type; fp.owidth = d->width; fp.x = r.x; fp.y = r.y; fp.width = r.width; fp.height = r.height;
      fp.black0 = d->blacklevels[0]; fp.white0 = d->whitelevels[0];
      std::memcpy(fp.cfa, d->cfa, sizeof(fp.cfa));
      std::memcpy(fp.wb_coeffs, d->wb_coeffs, sizeof(fp.wb_coeffs));
      std::memcpy(fp.cam_to_xyz_normalized, d->cam_to_xyz_normalized, sizeof(fp.cam_to_xyz_normalized));
      fp.exposure = d->exposure; fp.npoints = d->npoints; std::memcpy(fp.points, d->points, sizeof(fp.points));
      fp.linear = linear; fp.out_type = out_type; fp.schedule = d->schedule;
      tm.rest = "fused gofloat+demosaic+to_lab+basecurve+from_lab+gamma(+transform)";
      if (transform_noop) {
        rc = ipk_raw_to_srgb(&fp, src, dst, stream);
        if (rc == IPK_OK && used_fused) *used_fused = 1;
        return rc;
      }
      // An orientation other than Normal (every portrait shot): OpTransform is the last op and a pure permutation of
      // pixels, so gofloat..gamma still run as the one fused launch, into a scratch buffer, and rotate_buffer (+ the
      // quantise loop) follows -- 2 or 3 launches instead of 7.
      size_t ow = 0, oh = 0;
      // Rotate90 / Rotate270 of a Bayer frame (the portrait shot): permute the 1-channel mosaic and run the fused kernel in rotated
      // space -- no pass over the 3-channel result at all
      rc = ipk_raw_to_srgb_oriented(&fp, src, orientation, dst, &ow, &oh, stream);
      if (rc == IPK_OK) {
        if (ow != fw || oh != fh) return fail(IPK_ERR_INVALID, "internal: produced %zux%zu, negotiated %zux%zu", ow, oh, fw, fh);
        if (used_fused) *used_fused = 1;
        return IPK_OK;
      }
      if (rc != IPK_ERR_UNSUPPORTED) return rc;
      Scratch sc2(S(stream));
      void *tmp = nullptr;
      // otherwise: the fused kernel quantises on the way out; the permutation then runs on the 3- or 6-byte pixels (it commutes with the
      // per-sample output8bit / output16bit)
      rc = sc2.get(r.width * r.height * 3 * (out_type == IPK_OUT_F32 ? sizeof(float) : (out_type == IPK_OUT_U8 ? 1 : 2)), &tmp); if (rc) return rc;
      rc = ipk_raw_to_srgb(&fp, src, tmp, stream); if (rc < 0) return rc;
      if (out_type == IPK_OUT_F32) rc = ipk_rotate_buffer(static_cast<const float *>(tmp), r.width, r.height, orientation, static_cast<float *>(dst), &ow, &oh, stream);
      else if (out_type == IPK_OUT_U8) rc = ipk_rotate_image_u8(static_cast<const uint8_t *>(tmp), r.width, r.height, orientation, static_cast<uint8_t *>(dst), &ow, &oh, stream);
      else rc = ipk_rotate_image_u16(static_cast<const uint16_t *>(tmp), r.width, r.height, orientation, static_cast<uint16_t *>(dst), &ow, &oh, stream);
      if (rc < 0) return rc;
      if (ow != fw || oh != fh) return fail(IPK_ERR_INVALID, "internal: produced %zux%zu, negotiated %zux%zu", ow, oh, fw, fh);
      if (used_fused) *used_fused = 1;
      return IPK_OK;
    }
  }

  // ---- raster sources, same idea: run_other + tolab..gamma (+ quantisation) as one launch when OpDemosaic (a 4-channel buffer at
  // scale <= 1: pass-through, demosaic.rs:39-44) and OpRotateCrop are no-ops ----
  if (d->allow_fused && !raw && rcop.noop() && r.x == 0 && r.y == 0 && r.width == d->width && r.height == d->height &&
      r.width * r.height >= 256 && ipk::calculate_scaling_total(r.width, r.height, dw, dh).scale <= 1.0f) {
    tm.rest = "fused gofloat+to_lab+basecurve+from_lab+gamma(+transform)";
    if (transform_noop) {
      rc = ipk_raster_to_srgb(src, d->src_type, r.width, r.height, d->wb_coeffs, d->cam_to_xyz_normalized, d->exposure, d->points, d->npoints,
                              linear, out_type, dst, stream);
      if (rc == IPK_OK && used_fused) *used_fused = 1;
      return rc;
    }
    // a non-Normal orientation: the same launch into a scratch image of the output type, then the permutation
    Scratch sc2(S(stream));
    void *tmp = nullptr;
    size_t ow = 0, oh = 0;
    rc = sc2.get(r.width * r.height * 3 * (out_type == IPK_OUT_F32 ? sizeof(float) : (out_type == IPK_OUT_U8 ? 1 : 2)), &tmp); if (rc) return rc;
    rc = ipk_raster_to_srgb(src, d->src_type, r.width, r.height, d->wb_coeffs, d->cam_to_xyz_normalized, d->exposure, d->points, d->npoints,
                            linear, out_type, tmp, stream);
    if (rc < 0) return rc;
    if (out_type == IPK_OUT_F32) rc = ipk_rotate_buffer(static_cast<const float *>(tmp), r.width, r.height, orientation, static_cast<float *>(dst), &ow, &oh, stream);
    else if (out_type == IPK_OUT_U8) rc = ipk_rotate_image_u8(static_cast<const uint8_t *>(tmp), r.width, r.height, orientation, static_cast<uint8_t *>(dst), &ow, &oh, stream);
    else rc = ipk_rotate_image_u16(static_cast<const uint16_t *>(tmp), r.width, r.height, orientation, static_cast<uint16_t *>(dst), &ow, &oh, stream);
    if (rc < 0) return rc;
    if (ow != fw || oh != fh) return fail(IPK_ERR_INVALID, "internal: produced %zux%zu, negotiated %zux%zu", ow, oh, fw, fh);
    if (used_fused) *used_fused = 1;
    return IPK_OK;
  }

  // ---- staged path: the eight ops in the reference's order (pipeline.rs:155-164) ----
  Scratch sc(S(stream));
  hipStream_t st = S(stream); (void)st;
  size_t w = r.width, h = r.height, colors;
  int monochrome = 0;
  void *buf = nullptr;
  // gofloat (+ demosaic when OpDemosaic::run would take its scaled_demosaic branch: one pass over the raw frame)
  bool demosaic_done = false;
  if (d->allow_fused && cfa_branch && d->cpp == 1) {
    ipk::Cfa cfa;
    const float scale = ipk::calculate_scaling_total(w, h, dw, dh).scale;
    if (ipk::Cfa::parse(d->cfa, cfa) && cfa.valid() && scale >= ipk::demosaic_minscale(cfa.width)) {          // demosaic.rs:47-50
      rc = sc.get(dw * dh * 4 * sizeof(float), &buf); if (rc) return rc;
      rc = ipk_raw_scaled_demosaic(src, d->src_type, d->width, r.x, r.y, w, h, d->blacklevels[0], d->whitelevels[0], d->cfa, dw, dh,
                                   static_cast<float *>(buf), stream);
      if (rc < 0) return rc;
      w = dw; h = dh; colors = 4; demosaic_done = true;
    }
  }
  if (d->allow_fused && !raw && !demosaic_done && ipk::calculate_scaling_total(w, h, dw, dh).scale > 1.0f) {
    // raster source under a size limit: run_other + OpDemosaic's scale_down_opbuf branch (demosaic.rs:44-46) in one pass
    rc = sc.get(dw * dh * 4 * sizeof(float), &buf); if (rc) return rc;
    rc = ipk_raster_scale_down(src, d->src_type, d->width, r.x, r.y, w, h, dw, dh, static_cast<float *>(buf), stream);
    if (rc < 0) return rc;
    w = dw; h = dh; colors = 4; demosaic_done = true;
  }
  if (demosaic_done) {
  } else if (raw) {
    if (d->cpp == 1 && !d->is_cfa) {
      colors = 4; monochrome = 1;
      rc = sc.get(w * h * 4 * sizeof(float), &buf); if (rc) return rc;
      rc = d->src_type == IPK_SRC_U16
        ? ipk_gofloat_mono_u16(static_cast<const uint16_t *>(src), d->width, r.x, r.y, w, h, d->blacklevels[0], d->whitelevels[0], static_cast<float *>(buf), stream)
        : ipk_gofloat_mono_f32(static_cast<const float *>(src), d->width, r.x, r.y, w, h, d->blacklevels[0], d->whitelevels[0], static_cast<float *>(buf), stream);
    } else if (d->cpp == 3) {
      colors = 4;
      rc = sc.get(w * h * 4 * sizeof(float), &buf); if (rc) return rc;
      rc = d->src_type == IPK_SRC_U16
        ? ipk_gofloat_rgb_u16(static_cast<const uint16_t *>(src), d->width, r.x, r.y, w, h, d->blacklevels, d->whitelevels, static_cast<float *>(buf), stream)
        : ipk_gofloat_rgb_f32(static_cast<const float *>(src), d->width, r.x, r.y, w, h, d->blacklevels, d->whitelevels, static_cast<float *>(buf), stream);
    } else {
      if (d->cpp != 1) return fail(IPK_ERR_UNSUPPORTED, "cpp=%d sources are not supported", d->cpp);
      colors = 1;
      rc = sc.get(w * h * sizeof(float), &buf); if (rc) return rc;
      rc = d->src_type == IPK_SRC_U16
        ? ipk_gofloat_cfa_u16(static_cast<const uint16_t *>(src), d->width, r.x, r.y, w, h, d->blacklevels[0], d->whitelevels[0], static_cast<float *>(buf), stream)
        : ipk_gofloat_cfa_f32(static_cast<const float *>(src), d->width, r.x, r.y, w, h, d->blacklevels[0], d->whitelevels[0], static_cast<float *>(buf), stream);
    }
  } else {
    colors = 4;
    rc = sc.get(w * h * 4 * sizeof(float), &buf); if (rc) return rc;
    rc = d->src_type == IPK_SRC_RGB8
      ? ipk_gofloat_other_u8(static_cast<const uint8_t *>(src), d->width, r.x, r.y, w, h, static_cast<float *>(buf), stream)
      : ipk_gofloat_other_u16(static_cast<const uint16_t *>(src), d->width, r.x, r.y, w, h, static_cast<float *>(buf), stream);
  }
  if (rc < 0) return rc;
  tm.mark(demosaic_done ? "gofloat+demosaic" : "gofloat");
  // demosaic
  if (!demosaic_done) {
    void *o = nullptr; size_t ow, oh;
    rc = sc.get(std::max(w * h, dw * dh) * 4 * sizeof(float), &o); if (rc) return rc;
    rc = ipk_demosaic_run(static_cast<const float *>(buf), w, h, colors, d->cfa, dw, dh, static_cast<float *>(o), &ow, &oh, stream);
    if (rc < 0) return rc;
    if (rc == IPK_NOOP) sc.release(o); else { sc.release(buf); buf = o; w = ow; h = oh; colors = 4; }
    tm.mark("demosaic");
  }
  // rotatecrop
  {
    size_t ow, oh;
    rc = ipk_rotatecrop(static_cast<const float *>(buf), w, h, colors, d->rotatecrop, nullptr, &ow, &oh, stream);
    if (rc < 0) return rc;
    if (rc != IPK_NOOP) {
      void *o = nullptr;
      rc = sc.get(ow * oh * colors * sizeof(float), &o); if (rc) return rc;
      rc = ipk_rotatecrop(static_cast<const float *>(buf), w, h, colors, d->rotatecrop, static_cast<float *>(o), &ow, &oh, stream);
      if (rc < 0) return rc;
      sc.release(buf); buf = o; w = ow; h = oh;
    }
    tm.mark("rotatecrop");
  }
  const size_t n3 = w * h * 3 * sizeof(float);
  const bool f32_out0 = out_type == IPK_OUT_F32;
  bool chained = false;
  if (d->allow_fused && colors == 4 && !f32_out0 && transform_noop && w * h >= 256) {
    // ... and with output8bit / output16bit in the same pass when the caller wants 8 or 16 bits and nothing follows: the f32 image never exists
    if (w != fw || h != fh) return fail(IPK_ERR_INVALID, "internal: produced %zux%zu, negotiated %zux%zu", w, h, fw, fh);
    rc = ipk_pointwise_chain_out(static_cast<const float *>(buf), w, h, monochrome, d->wb_coeffs, d->cam_to_xyz_normalized, d->exposure, d->points, d->npoints,
                                 linear, out_type, dst, stream);
    if (rc < 0) return rc;
    tm.rest = nullptr;
    tm.mark("to_lab+basecurve+from_lab+gamma+quantise");
    return IPK_OK;
  }
  if (d->allow_fused && colors == 4) {
    // tolab + basecurve + fromlab + gamma in one pass (no cache wants the three intermediates)
    void *o = nullptr;
    if (f32_out0 && transform_noop) o = dst; else { rc = sc.get(n3, &o); if (rc) return rc; }
    rc = ipk_pointwise_chain(static_cast<const float *>(buf), w, h, monochrome, d->wb_coeffs, d->cam_to_xyz_normalized, d->exposure, d->points, d->npoints,
                             linear, static_cast<float *>(o), stream);
    if (rc < 0) return rc;
    sc.release(buf); buf = o; colors = 3; chained = true;
    tm.mark("to_lab+basecurve+from_lab+gamma");
  }
  // tolab
  if (!chained) {
    void *o = nullptr;
    rc = sc.get(w * h * 3 * sizeof(float), &o); if (rc) return rc;
    rc = ipk_tolab(static_cast<const float *>(buf), w, h, monochrome, d->wb_coeffs, d->cam_to_xyz_normalized, static_cast<float *>(o), stream);
    if (rc < 0) return rc;
    sc.release(buf); buf = o; colors = 3;
    tm.mark("to_lab");
  }
  // basecurve
  if (!chained) {
    void *o = nullptr;
    rc = sc.get(n3, &o); if (rc) return rc;
    rc = ipk_basecurve(static_cast<const float *>(buf), w, h, d->exposure, d->points, d->npoints, static_cast<float *>(o), stream);
    if (rc < 0) return rc;
    if (rc == IPK_NOOP) sc.release(o); else { sc.release(buf); buf = o; }
    tm.mark("basecurve");
  }
  // the last f32 stage writes straight into dst when the caller wants f32
  const bool gamma_runs = !linear;
  const bool f32_out = out_type == IPK_OUT_F32;
  // fromlab
  if (!chained) {
    void *o = nullptr;
    const bool last = f32_out && !gamma_runs && transform_noop;
    if (last) o = dst; else { rc = sc.get(n3, &o); if (rc) return rc; }
    rc = ipk_fromlab(static_cast<const float *>(buf), w, h, static_cast<float *>(o), stream);
    if (rc < 0) return rc;
    sc.release(buf); buf = o;
    tm.mark("from_lab");
  }
  // gamma
  if (gamma_runs && !chained) {
    void *o = nullptr;
    const bool last = f32_out && transform_noop;
    if (last) o = dst; else { rc = sc.get(n3, &o); if (rc) return rc; }
    rc = ipk_gamma(static_cast<const float *>(buf), w, h, 3, 0, static_cast<float *>(o), stream);
    if (rc < 0) return rc;
    sc.release(buf); buf = o;
    tm.mark("gamma");
  }
  // transform -- for the 8- and 16-bit outputs after the quantise loop instead of before it: output8bit / output16bit act
  // per sample, so the permutation commutes with them and then moves 3 or 6 bytes per pixel instead of 12
  if (!transform_noop && !f32_out) {
    tm.rest = "quantise+transform";
    void *q = nullptr; size_t ow, oh;
    rc = sc.get(w * h * 3 * (out_type == IPK_OUT_U8 ? 1 : 2), &q); if (rc) return rc;
    rc = out_type == IPK_OUT_U8 ? ipk_output8bit(static_cast<const float *>(buf), w * h * 3, static_cast<uint8_t *>(q), stream)
                                : ipk_output16bit(static_cast<const float *>(buf), w * h * 3, static_cast<uint16_t *>(q), stream);
    if (rc < 0) return rc;
    rc = out_type == IPK_OUT_U8 ? ipk_rotate_image_u8(static_cast<const uint8_t *>(q), w, h, orientation, static_cast<uint8_t *>(dst), &ow, &oh, stream)
                                : ipk_rotate_image_u16(static_cast<const uint16_t *>(q), w, h, orientation, static_cast<uint16_t *>(dst), &ow, &oh, stream);
    if (rc < 0) return rc;
    if (ow != fw || oh != fh) return fail(IPK_ERR_INVALID, "internal: produced %zux%zu, negotiated %zux%zu", ow, oh, fw, fh);
    return IPK_OK;
  }
  if (!transform_noop) {
    void *o = nullptr; size_t ow, oh;
    if (f32_out) o = dst; else { rc = sc.get(n3, &o); if (rc) return rc; }
    rc = ipk_rotate_buffer(static_cast<const float *>(buf), w, h, orientation, static_cast<float *>(o), &ow, &oh, stream);
    if (rc < 0) return rc;
    sc.release(buf); buf = o; w = ow; h = oh;
    tm.mark("transform");
  }
  if (w != fw || h != fh) return fail(IPK_ERR_INVALID, "internal: produced %zux%zu, negotiated %zux%zu", w, h, fw, fh);
  // quantise (pipeline.rs:408-414 / :455-461)
  if (out_type != IPK_OUT_F32) tm.rest = "quantise";
  if (out_type == IPK_OUT_U8) rc = ipk_output8bit(static_cast<const float *>(buf), w * h * 3, static_cast<uint8_t *>(dst), stream);
  else if (out_type == IPK_OUT_U16) rc = ipk_output16bit(static_cast<const float *>(buf), w * h * 3, static_cast<uint16_t *>(dst), stream);
  else rc = IPK_OK;
  return rc < 0 ? rc : IPK_OK;
}

// ------------------------------------------------------------------------------------------
// Pipeline::run with a cache (src/pipeline.rs:341-372): hash chain + memoised device OpBuffers
// ------------------------------------------------------------------------------------------
namespace {
struct Negotiated {
  ipk::Rect r; size_t dw, dh, fw, fh; ipk::RotateCrop rc; int linear; int orientation; bool transform_noop;
};
int negotiate(const ipk_pipeline_desc *d, int out_type, Negotiated &n) {
  if (!d) return fail(IPK_ERR_INVALID, "null descriptor");
  if (out_type < 0 || out_type > 2) return fail(IPK_ERR_INVALID, "bad out_type");
  // one negotiation: the sizes and the rotatecrop state both come from pipeline_sizes_impl's folds (the reverse fold is seeded
  // with scaling_size of the forward result, pipeline.rs:328-335 -- not with the size run() produces)
  int rc = pipeline_sizes_impl(d, &n.dw, &n.dh, &n.fw, &n.fh, &n.rc); if (rc) return rc;
  ipk::size_image(d->crop_top, d->crop_right, d->crop_bottom, d->crop_left, d->width, d->height, n.r);
  n.linear = out_type == IPK_OUT_U8 ? 0 : (out_type == IPK_OUT_U16 ? 1 : (d->linear != 0));
  n.orientation = ipk::transform_orientation(d->rotation, d->fliph != 0, d->flipv != 0);
  n.transform_noop = n.orientation == IPK_OR_NORMAL || n.orientation == IPK_OR_UNKNOWN;
  return IPK_OK;
}
// ophashes[0..7] of pipeline.rs:342-361.  Field order follows the reference structs.
void hash_chain(const ipk_pipeline_desc *d, const Negotiated &n, uint64_t source_id, ipk::BufHash out[8]) {
  ipk::BufHasher h;
  // PipelineSettings (pipeline.rs:110-137)
  h.usize(d->maxwidth); h.usize(d->maxheight); h.usize(n.dw); h.usize(n.dh); h.boolean(n.linear != 0); h.boolean(d->use_fastpath != 0);
  // gofloat (gofloat.rs:4-12).  The reference hashes no identity of the image at all (a PipelineCache is only
  // valid for one source); source_id + the source geometry make one cache safe across several resident frames.
  h.str_raw("gofloat");
  h.usize(d->crop_top); h.usize(d->crop_right); h.usize(d->crop_bottom); h.usize(d->crop_left); h.boolean(d->is_cfa != 0);
  h.f32s(d->blacklevels, 4); h.f32s(d->whitelevels, 4);
  h.u64(source_id); h.u64(uint64_t(d->src_type)); h.usize(d->width); h.usize(d->height); h.u64(uint64_t(d->cpp));
  out[0] = h.result();
  h.str_raw("demosaic"); h.string(d->cfa);                                               // demosaic.rs:4-6
  out[1] = h.result();
  h.str_raw("rotatecrop");                                                               // rotatecrop.rs:10-18
  h.f32(n.rc.crop_top); h.f32(n.rc.crop_right); h.f32(n.rc.crop_bottom); h.f32(n.rc.crop_left); h.f32(n.rc.rotation);
  h.f32(n.rc.input_ratio); h.boolean(n.rc.has_output); if (n.rc.has_output) { h.usize(n.rc.out_w); h.usize(n.rc.out_h); }
  out[2] = h.result();
  h.str_raw("to_lab"); h.f32s(d->cam_to_xyz_normalized, 12); h.f32s(d->wb_coeffs, 4);   // colorspaces.rs:5-10 (the fields run() reads)
  out[3] = h.result();
  h.str_raw("basecurve"); h.f32(d->exposure); h.u64(uint64_t(d->npoints)); h.f32s(d->points, size_t(2 * std::max(0, d->npoints)));   // curves.rs:6-9
  out[4] = h.result();
  h.str_raw("from_lab"); out[5] = h.result();
  h.str_raw("gamma"); out[6] = h.result();
  h.str_raw("transform"); h.u64(uint64_t(d->rotation)); h.boolean(d->fliph != 0); h.boolean(d->flipv != 0);                          // transform.rs:15-19
  out[7] = h.result();
}
ipk::BufHash key_of(const uint8_t *k) { ipk::BufHash b; std::memcpy(b.data(), k, 32); return b; }
}  // namespace

int ipk_pipeline_hashes(const ipk_pipeline_desc *d, int out_type, uint64_t source_id, uint8_t *out256) {
  if (!out256) return fail(IPK_ERR_INVALID, "null output");
  IPK_FOLD_CFA(ipk_pipeline_desc, d)
  Negotiated n; int rc = negotiate(d, out_type, n); if (rc) return rc;
  if (d->npoints < 0 || d->npoints > 64) return fail(IPK_ERR_INVALID, "npoints out of range");
  ipk::BufHash hs[8]; hash_chain(d, n, source_id, hs);
  for (int i = 0; i < 8; ++i) std::memcpy(out256 + 32 * i, hs[i].data(), 32);
  return IPK_OK;
}

// the cached buffers' users run on the owner's device: drain THAT device (whatever context the calling thread has current) before they are freed
static void cache_drain(ipk_cache *c) {
  ipk_ctx *o = c->owner ? c->owner : ipk_ctx_current();
  if (!o || !o->ready) return;
  int prev = -1; (void)hipGetDevice(&prev);
  if (prev != o->device) (void)hipSetDevice(o->device);
  (void)hipDeviceSynchronize();
  if (prev >= 0 && prev != o->device) (void)hipSetDevice(prev);
}
int ipk_cache_new(size_t max_bytes, ipk_cache **out) {
  if (!out) return fail(IPK_ERR_INVALID, "null output");
  *out = new ipk_cache(max_bytes, ipk_ctx_current());
  return IPK_OK;
}
int ipk_cache_free(ipk_cache *c) {
  if (c) { cache_drain(c); delete c; }
  return IPK_OK;
}
int ipk_cache_clear(ipk_cache *c) {
  if (!c) return fail(IPK_ERR_INVALID, "null cache");
  cache_drain(c);
  c->lru.clear();
  return IPK_OK;
}
int ipk_cache_contains(const ipk_cache *c, const uint8_t *key32) {
  if (!c || !key32) return fail(IPK_ERR_INVALID, "null argument");
  return c->lru.contains(key_of(key32)) ? 1 : 0;
}
int ipk_cache_stats(const ipk_cache *c, size_t *bytes, size_t *entries, uint64_t *hits, uint64_t *misses, uint64_t *evictions) {
  if (!c) return fail(IPK_ERR_INVALID, "null cache");
  size_t b, e; uint64_t hi, mi, ev; c->lru.stats(b, e, hi, mi, ev);
  if (bytes) *bytes = b; if (entries) *entries = e; if (hits) *hits = hi; if (misses) *misses = mi; if (evictions) *evictions = ev;
  return IPK_OK;
}
int ipk_cache_get(ipk_cache *c, const uint8_t *key32, const float **data, size_t *width, size_t *height, size_t *colors, int *monochrome) {
  if (!c || !key32) return fail(IPK_ERR_INVALID, "null argument");
  CBufP b = c->lru.get(key_of(key32));
  if (!b) return IPK_NOOP;
  if (data) *data = static_cast<const float *>(b->p);
  if (width) *width = b->w; if (height) *height = b->h; if (colors) *colors = b->colors; if (monochrome) *monochrome = b->mono;
  return IPK_OK;
}
int ipk_selftest_cache_put(ipk_cache *c, const uint8_t *key32, size_t bytes) {   // LRU bookkeeping without device memory (CPU tests)
  if (!c || !key32) return fail(IPK_ERR_INVALID, "null argument");
  c->lru.put(key_of(key32), std::make_shared<CBuf>(), bytes);
  return IPK_OK;
}
int ipk_selftest_sha256(const void *data, size_t n, uint8_t *out32) {
  ipk::Sha256 s; s.update(data, n); const auto r = s.result(); std::memcpy(out32, r.data(), 32); return IPK_OK;
}

int ipk_pipeline_run_cached(const ipk_pipeline_desc *d, const void *src, uint64_t source_id, ipk_cache *cache, int out_type, void *dst,
                            int *ops_run, int *used_fused, void *stream) {
  REQUIRE_INIT();
  if (!d || !src || !dst || !cache) return fail(IPK_ERR_INVALID, "null argument");
  if (!cache->owner) cache->owner = ipk_ctx_current();                   // a cache made before ipk_init belongs to the first context that fills it
  if (cache->owner != ipk_ctx_current()) return fail(IPK_ERR_INVALID, "the cache belongs to another context (its buffers live on that context's device)");
  IPK_FOLD_CFA(ipk_pipeline_desc, d)
  if (d->npoints < 0 || d->npoints > 64) return fail(IPK_ERR_INVALID, "npoints out of range");
  if (ipk_pipeline_takes_fastpath(d, out_type) == 1) {                    // returns before the cache is consulted (pipeline.rs:381-402)
    if (ops_run) *ops_run = 0;
    if (used_fused) *used_fused = 0;
    return run_fastpath(d, src, dst, out_type, stream);
  }
  Negotiated n; int rc = negotiate(d, out_type, n); if (rc) return rc;
  ipk::BufHash hs[8]; hash_chain(d, n, source_id, hs);
  if (ops_run) *ops_run = 0;
  if (used_fused) *used_fused = 0;
  hipStream_t st = S(stream);

  // the latest op whose output is memoised (pipeline.rs:352-361: every hash is looked up, the last hit wins)
  CBufP buf; int startpos = 0;
  for (int i = 0; i < 8; ++i) { CBufP hit = cache->lru.get(hs[i]); if (hit) { buf = hit; startpos = i + 1; } }

  const bool raw = d->src_type == IPK_SRC_U16 || d->src_type == IPK_SRC_F32;
  const bool cfa_branch = raw && !(d->cpp == 1 && !d->is_cfa) && d->cpp != 3;
  const size_t w0 = n.r.width, h0 = n.r.height;
  int mask = 0;

  // Nothing memoised and the whole chain is one fused launch: cheaper on this machine than materialising seven
  // intermediates; only the final buffer enters the cache.
  if (startpos == 0 && d->allow_fused && cfa_branch && d->cpp == 1 && n.rc.noop()) {   // any orientation: ipk_pipeline_run folds OpTransform in
    const float scale = ipk::calculate_scaling_total(w0, h0, n.dw, n.dh).scale;
    ipk::Cfa cfa; int xo, yo;
    if (scale <= 1.0f && ipk::Cfa::parse(d->cfa, cfa) && (cfa.bayer_phase(xo, yo) || cfa.three_colour())) {
      ipk_pipeline_desc d2 = *d; d2.linear = n.linear;
      CBufP o; rc = cbuf_new(n.fw, n.fh, 3, 0, o); if (rc) return rc;
      int fused = 0;
      rc = ipk_pipeline_run(&d2, src, o->p, IPK_OUT_F32, &fused, stream); if (rc < 0) return rc;
      cache->lru.put(hs[7], o, o->bytes());
      buf = o; startpos = 8; mask = 0xFF;
      if (used_fused) *used_fused = fused;
    }
  }

  for (int i = startpos; i < 8; ++i) {
    CBufP o;
    switch (i) {
      case 0: {                                                          // gofloat
        if (d->allow_fused && cfa_branch && d->cpp == 1) {               // + demosaic in the same pass when it would scale
          ipk::Cfa cfa;
          const float scale = ipk::calculate_scaling_total(w0, h0, n.dw, n.dh).scale;
          if (ipk::Cfa::parse(d->cfa, cfa) && cfa.valid() && scale >= ipk::demosaic_minscale(cfa.width) && !cache->lru.contains(hs[1])) {
            rc = cbuf_new(n.dw, n.dh, 4, 0, o); if (rc) return rc;
            rc = ipk_raw_scaled_demosaic(src, d->src_type, d->width, n.r.x, n.r.y, w0, h0, d->blacklevels[0], d->whitelevels[0], d->cfa, n.dw, n.dh,
                                         static_cast<float *>(o->p), stream);
            if (rc < 0) return rc;
            mask |= 3; buf = o; cache->lru.put(hs[1], o, o->bytes()); i = 1;   // op 1's output; op 0's is never materialised
            continue;
          }
        }
        if (d->allow_fused && !raw && ipk::calculate_scaling_total(w0, h0, n.dw, n.dh).scale > 1.0f && !cache->lru.contains(hs[1])) {
          rc = cbuf_new(n.dw, n.dh, 4, 0, o); if (rc) return rc;                 // raster under a size limit: run_other + scale_down_opbuf in one pass
          rc = ipk_raster_scale_down(src, d->src_type, d->width, n.r.x, n.r.y, w0, h0, n.dw, n.dh, static_cast<float *>(o->p), stream);
          if (rc < 0) return rc;
          mask |= 3; buf = o; cache->lru.put(hs[1], o, o->bytes()); i = 1;
          continue;
        }
        if (raw) {
          if (d->cpp == 1 && !d->is_cfa) {
            rc = cbuf_new(w0, h0, 4, 1, o); if (rc) return rc;
            rc = d->src_type == IPK_SRC_U16
              ? ipk_gofloat_mono_u16(static_cast<const uint16_t *>(src), d->width, n.r.x, n.r.y, w0, h0, d->blacklevels[0], d->whitelevels[0], static_cast<float *>(o->p), stream)
              : ipk_gofloat_mono_f32(static_cast<const float *>(src), d->width, n.r.x, n.r.y, w0, h0, d->blacklevels[0], d->whitelevels[0], static_cast<float *>(o->p), stream);
          } else if (d->cpp == 3) {
            rc = cbuf_new(w0, h0, 4, 0, o); if (rc) return rc;
            rc = d->src_type == IPK_SRC_U16
              ? ipk_gofloat_rgb_u16(static_cast<const uint16_t *>(src), d->width, n.r.x, n.r.y, w0, h0, d->blacklevels, d->whitelevels, static_cast<float *>(o->p), stream)
              : ipk_gofloat_rgb_f32(static_cast<const float *>(src), d->width, n.r.x, n.r.y, w0, h0, d->blacklevels, d->whitelevels, static_cast<float *>(o->p), stream);
          } else {
            if (d->cpp != 1) return fail(IPK_ERR_UNSUPPORTED, "cpp=%d sources are not supported", d->cpp);
            rc = cbuf_new(w0, h0, 1, 0, o); if (rc) return rc;
            rc = d->src_type == IPK_SRC_U16
              ? ipk_gofloat_cfa_u16(static_cast<const uint16_t *>(src), d->width, n.r.x, n.r.y, w0, h0, d->blacklevels[0], d->whitelevels[0], static_cast<float *>(o->p), stream)
              : ipk_gofloat_cfa_f32(static_cast<const float *>(src), d->width, n.r.x, n.r.y, w0, h0, d->blacklevels[0], d->whitelevels[0], static_cast<float *>(o->p), stream);
          }
        } else {
          rc = cbuf_new(w0, h0, 4, 0, o); if (rc) return rc;
          rc = d->src_type == IPK_SRC_RGB8
            ? ipk_gofloat_other_u8(static_cast<const uint8_t *>(src), d->width, n.r.x, n.r.y, w0, h0, static_cast<float *>(o->p), stream)
            : ipk_gofloat_other_u16(static_cast<const uint16_t *>(src), d->width, n.r.x, n.r.y, w0, h0, static_cast<float *>(o->p), stream);
        }
        break;
      }
      case 1: {                                                          // demosaic
        size_t ow, oh;
        rc = cbuf_new(std::max(buf->w, n.dw), std::max(buf->h, n.dh), 4, buf->mono, o); if (rc) return rc;
        rc = ipk_demosaic_run(static_cast<const float *>(buf->p), buf->w, buf->h, buf->colors, d->cfa, n.dw, n.dh, static_cast<float *>(o->p), &ow, &oh, stream);
        if (rc >= 0 && rc != IPK_NOOP) { o->w = ow; o->h = oh; }
        break;
      }
      case 2: {                                                          // rotatecrop
        size_t ow, oh;
        rc = ipk_rotatecrop(static_cast<const float *>(buf->p), buf->w, buf->h, buf->colors, d->rotatecrop, nullptr, &ow, &oh, stream);
        if (rc < 0 || rc == IPK_NOOP) break;
        rc = cbuf_new(ow, oh, buf->colors, buf->mono, o); if (rc) return rc;
        rc = ipk_rotatecrop(static_cast<const float *>(buf->p), buf->w, buf->h, buf->colors, d->rotatecrop, static_cast<float *>(o->p), &ow, &oh, stream);
        break;
      }
      case 3:                                                            // to_lab
        rc = cbuf_new(buf->w, buf->h, 3, buf->mono, o); if (rc) return rc;
        rc = ipk_tolab(static_cast<const float *>(buf->p), buf->w, buf->h, buf->mono, d->wb_coeffs, d->cam_to_xyz_normalized, static_cast<float *>(o->p), stream);
        break;
      case 4:                                                            // basecurve
        if (curve_is_noop(d->exposure, d->npoints)) { rc = IPK_NOOP; break; }
        rc = cbuf_new(buf->w, buf->h, 3, buf->mono, o); if (rc) return rc;
        rc = ipk_basecurve(static_cast<const float *>(buf->p), buf->w, buf->h, d->exposure, d->points, d->npoints, static_cast<float *>(o->p), stream);
        break;
      case 5:                                                            // from_lab
        rc = cbuf_new(buf->w, buf->h, 3, buf->mono, o); if (rc) return rc;
        rc = ipk_fromlab(static_cast<const float *>(buf->p), buf->w, buf->h, static_cast<float *>(o->p), stream);
        break;
      case 6:                                                            // gamma
        if (n.linear) { rc = IPK_NOOP; break; }
        rc = cbuf_new(buf->w, buf->h, 3, buf->mono, o); if (rc) return rc;
        rc = ipk_gamma(static_cast<const float *>(buf->p), buf->w, buf->h, 3, 0, static_cast<float *>(o->p), stream);
        break;
      case 7: {                                                          // transform
        if (n.transform_noop) { rc = IPK_NOOP; break; }
        size_t ow, oh;
        rc = cbuf_new(buf->w, buf->h, 3, buf->mono, o); if (rc) return rc;
        rc = ipk_rotate_buffer(static_cast<const float *>(buf->p), buf->w, buf->h, n.orientation, static_cast<float *>(o->p), &ow, &oh, stream);
        if (rc >= 0) { o->w = ow; o->h = oh; }
        break;
      }
    }
    if (rc < 0) return rc;
    if (rc != IPK_NOOP) buf = o;                                         // a no-op hands its input on (the same Arc under a second key)
    mask |= 1 << i;
    cache->lru.put(hs[i], buf, buf->bytes());
  }
  if (ops_run) *ops_run = mask;
  if (buf->w != n.fw || buf->h != n.fh || buf->colors != 3)
    return fail(IPK_ERR_INVALID, "internal: produced %zux%zux%zu, negotiated %zux%zu", buf->w, buf->h, buf->colors, n.fw, n.fh);
  const size_t cnt = buf->w * buf->h * 3;
  if (out_type == IPK_OUT_U8) rc = ipk_output8bit(static_cast<const float *>(buf->p), cnt, static_cast<uint8_t *>(dst), stream);
  else if (out_type == IPK_OUT_U16) rc = ipk_output16bit(static_cast<const float *>(buf->p), cnt, static_cast<uint16_t *>(dst), stream);
  else { HIPCHK(hipMemcpyAsync(dst, buf->p, cnt * sizeof(float), hipMemcpyDeviceToDevice, st)); rc = IPK_OK; }
  if (rc < 0) return rc;
  // `buf` may be evicted (and its memory freed) as soon as we return; hipFree waits for the device, so the copy above is safe
  return IPK_OK;
}

// ------------------------------------------------------------------------------------------
// host-pointer forms
// ------------------------------------------------------------------------------------------
namespace {
struct DevBuf {
  void *p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  int alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1) == hipSuccess ? IPK_OK : fail(IPK_ERR_NOMEM, "hipMalloc(%zu) failed", bytes); }
  int upload(const void *h, size_t bytes) { int rc = alloc(bytes); if (rc) return rc; HIPCHK(hipMemcpy(p, h, bytes, hipMemcpyHostToDevice)); return IPK_OK; }
  int download(void *h, size_t bytes) { HIPCHK(hipMemcpy(h, p, bytes, hipMemcpyDeviceToHost)); return IPK_OK; }
};
size_t src_elem_size(int t) { return t == IPK_SRC_U16 ? 2 : t == IPK_SRC_F32 ? 4 : t == IPK_SRC_RGB8 ? 1 : 2; }
size_t out_elem_size(int t) { return t == IPK_OUT_F32 ? 4 : t == IPK_OUT_U8 ? 1 : 2; }
}  // namespace

#define HOST_TRY(expr) do { int rc_ = (expr); if (rc_ < 0) return rc_; } while (0)

int ipk_host_pipeline_run(const ipk_pipeline_desc *d, const void *src, void *dst, int out_type, int *used_fused) {
  if (!src || !dst) { REQUIRE_INIT(); return fail(IPK_ERR_INVALID, "null argument"); }
  return ipk_host_pipeline_run_batch(d, &src, &dst, 1, out_type, used_fused);
}
void *ipk_host_alloc(size_t bytes) {
  void *p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return p;
}
void ipk_host_free(void *p) { if (p) (void)hipHostFree(p); }

namespace {
int HostLanes::ensure(size_t in_bytes, size_t out_bytes) {
  if (!made) {
    for (hipStream_t *s : {&up, &run, &down}) HIPCHK(hipStreamCreateWithFlags(s, hipStreamNonBlocking));
    for (int i = 0; i < kSlots; ++i)
      for (hipEvent_t *e : {&up_done[i], &run_done[i], &down_done[i]}) HIPCHK(hipEventCreateWithFlags(e, hipEventDisableTiming));
    made = true;
  }
  if (in_bytes > in_cap) {
    for (int i = 0; i < kSlots; ++i) { if (in[i]) (void)hipFree(in[i]); in[i] = nullptr; }
    in_cap = 0;
    for (int i = 0; i < kSlots; ++i) if (hipMalloc(&in[i], in_bytes) != hipSuccess) return fail(IPK_ERR_NOMEM, "hipMalloc(%zu) failed", in_bytes);
    in_cap = in_bytes;
  }
  if (out_bytes > out_cap) {
    for (int i = 0; i < kSlots; ++i) { if (out[i]) (void)hipFree(out[i]); out[i] = nullptr; }
    out_cap = 0;
    for (int i = 0; i < kSlots; ++i) if (hipMalloc(&out[i], out_bytes) != hipSuccess) return fail(IPK_ERR_NOMEM, "hipMalloc(%zu) failed", out_bytes);
    out_cap = out_bytes;
  }
  return IPK_OK;
}
void HostLanes::release() {
  for (hipStream_t *s : {&up, &run, &down}) if (*s) { (void)hipStreamSynchronize(*s); (void)hipStreamDestroy(*s); *s = nullptr; }
  for (int i = 0; i < kSlots; ++i) {
    for (hipEvent_t *e : {&up_done[i], &run_done[i], &down_done[i]}) if (*e) { (void)hipEventDestroy(*e); *e = nullptr; }
    if (in[i]) (void)hipFree(in[i]); if (out[i]) (void)hipFree(out[i]);
    in[i] = out[i] = nullptr;
  }
  in_cap = out_cap = 0; made = false;
}
}  // namespace

// ------------------------------------------------------------------------------------------
// A caller looping Pipeline::run over the frames of a shoot (src/pipeline.rs:246-249: frames are independent), device pointers
// ------------------------------------------------------------------------------------------
namespace {
// Is Pipeline::run for this descriptor exactly one fused launch per frame with nothing in front of or behind it (the conditions of
// ipk_pipeline_run's first branch, with OpTransform a no-op)?  Then a batch of such frames is one persistent launch per 64 (ipk_raw_to_srgb_batch).
bool desc_is_one_fused_launch(const ipk_pipeline_desc *d, int out_type, ipk_fused_params &fp) {
  size_t dw, dh, fw, fh;
  if (pipeline_sizes_impl(d, &dw, &dh, &fw, &fh, nullptr) != IPK_OK) return false;
  if (ipk_pipeline_takes_fastpath(d, out_type) == 1) return false;
  ipk::Rect r;
  if (!ipk::size_image(d->crop_top, d->crop_right, d->crop_bottom, d->crop_left, d->width, d->height, r)) return false;
  const bool raw = d->src_type == IPK_SRC_U16 || d->src_type == IPK_SRC_F32;
  const bool cfa_branch = raw && !(d->cpp == 1 && !d->is_cfa) && d->cpp != 3;
  const int orientation = ipk::transform_orientation(d->rotation, d->fliph != 0, d->flipv != 0);
  ipk::RotateCrop rcop;
  rcop.crop_top = d->rotatecrop[0]; rcop.crop_right = d->rotatecrop[1]; rcop.crop_bottom = d->rotatecrop[2];
  rcop.crop_left = d->rotatecrop[3]; rcop.rotation = d->rotatecrop[4];
  if (!(d->allow_fused && cfa_branch && d->cpp == 1 && rcop.noop())) return false;
  if (!(orientation == IPK_OR_NORMAL || orientation == IPK_OR_UNKNOWN)) return false;
  if (ipk::calculate_scaling_total(r.width, r.height, dw, dh).scale > 1.0f) return false;
  ipk::Cfa cfa; int xo, yo;
  if (!ipk::Cfa::parse(d->cfa, cfa) || !(cfa.bayer_phase(xo, yo) || cfa.three_colour())) return false;
  std::memset(&fp, 0, sizeof(fp));
  fp.struct_size = (uint32_t)sizeof(fp);
  fp.src_type = d->src_type; fp.owidth = d->width; fp.x = r.x; fp.y = r.y; fp.width = r.width; fp.height = r.height;
  fp.black0 = d->blacklevels[0]; fp.white0 = d->whitelevels[0];
  std::memcpy(fp.cfa, d->cfa, sizeof(fp.cfa));
  std::memcpy(fp.wb_coeffs, d->wb_coeffs, sizeof(fp.wb_coeffs));
  std::memcpy(fp.cam_to_xyz_normalized, d->cam_to_xyz_normalized, sizeof(fp.cam_to_xyz_normalized));
  fp.exposure = d->exposure; fp.npoints = d->npoints; std::memcpy(fp.points, d->points, sizeof(fp.points));
  fp.linear = out_type == IPK_OUT_U8 ? 0 : (out_type == IPK_OUT_U16 ? 1 : d->linear);   // pipeline.rs:405, :452
  fp.out_type = out_type; fp.schedule = d->schedule;
  return true;
}
}  // namespace

int ipk_pipeline_run_batch(const ipk_pipeline_desc *d, const void *const *srcs, void *const *dsts, size_t n, int out_type, int *used_fused, void *stream) {
  REQUIRE_INIT();
  if (!d || (n && (!srcs || !dsts))) return fail(IPK_ERR_INVALID, "null argument");
  if (out_type < 0 || out_type > 2) return fail(IPK_ERR_INVALID, "bad out_type");
  IPK_FOLD_CFA(ipk_pipeline_desc, d)
  if (d->npoints < 0 || d->npoints > 64) return fail(IPK_ERR_INVALID, "npoints out of range");
  for (size_t i = 0; i < n; ++i) if (!srcs[i] || !dsts[i]) return fail(IPK_ERR_INVALID, "null frame pointer at index %zu", i);
  if (used_fused) *used_fused = 0;
  if (n == 0) return IPK_OK;
  ipk_fused_params fp;
  if (n > 1 && desc_is_one_fused_launch(d, out_type, fp)) {
    const int rc = ipk_raw_to_srgb_batch(&fp, srcs, dsts, n, stream);
    if (rc == IPK_OK && used_fused) *used_fused = 1;
    return rc;
  }
  for (size_t i = 0; i < n; ++i) { const int rc = ipk_pipeline_run(d, srcs[i], dsts[i], out_type, used_fused, stream); if (rc < 0) return rc; }
  return IPK_OK;
}

// ---- the same over the process's device set: frame i -> set member i mod N, from ONE host thread (launches are asynchronous) -------------
namespace {
void devset_snapshot(std::vector<Context *> &v) {
  std::lock_guard<std::mutex> lk(g_reg_mu);
  v = g_devset;
  if (v.empty() && cx().ready) v.push_back(&cx());      // no set: the current context alone
}
struct CurrentGuard {                                    // puts the calling thread's current context (and its device) back
  Context *saved = t_current;
  ~CurrentGuard() { t_current = saved; if (cx().ready) (void)hipSetDevice(cx().device); }
};
}  // namespace

int ipk_pipeline_run_batch_multi(const ipk_pipeline_desc *d, const void *const *srcs, void *const *dsts, size_t n, int out_type, int *used_fused) {
  REQUIRE_INIT();
  if (!d || (n && (!srcs || !dsts))) return fail(IPK_ERR_INVALID, "null argument");
  std::vector<Context *> set; devset_snapshot(set);
  const size_t nd = set.size();
  CurrentGuard guard;
  int fused_all = 1;
  for (size_t k = 0; k < nd; ++k) {
    std::vector<const void *> s; std::vector<void *> o;
    for (size_t i = k; i < n; i += nd) { s.push_back(srcs[i]); o.push_back(dsts[i]); }
    if (s.empty()) continue;
    if (!set[k]->ready) return fail(IPK_ERR_NOT_INIT, "device-set member %zu has been destroyed", k);
    t_current = set[k];
    { int rc = bind_device(set[k]->device); if (rc) return rc; }
    if (!set[k]->multi_stream) HIPCHK(hipStreamCreateWithFlags(&set[k]->multi_stream, hipStreamNonBlocking));
    int fused = 0;
    const int rc = ipk_pipeline_run_batch(d, s.data(), o.data(), s.size(), out_type, &fused, set[k]->multi_stream);
    if (rc < 0) return rc;
    fused_all = fused_all && fused;
  }
  if (used_fused) *used_fused = n ? fused_all : 0;
  return IPK_OK;
}
int ipk_devices_sync(void) {
  REQUIRE_INIT();
  std::vector<Context *> set; devset_snapshot(set);
  CurrentGuard guard;
  int rc = IPK_OK;
  for (Context *c : set) {
    if (!c->ready || !c->multi_stream) continue;
    if (hipSetDevice(c->device) != hipSuccess || hipStreamSynchronize(c->multi_stream) != hipSuccess) {
      rc = fail(IPK_ERR_HIP, "synchronising device %d failed: %s", c->device, hipGetErrorString(hipGetLastError()));
    }
  }
  return rc;
}

// Host buffers in, host buffers out, over the device set: one host thread per member runs ipk_host_pipeline_run_batch on the frames dealt
// to it (its own three streams and two device slots), so every GPU uploads, computes and downloads at the same time.  Synchronous at return,
// like every ipk_host_* entry point.  The buffers should come from ipk_host_alloc (page-locked, visible to every device).
int ipk_host_pipeline_run_batch_multi(const ipk_pipeline_desc *d, const void *const *srcs, void *const *dsts, size_t n, int out_type, int *used_fused) {
  REQUIRE_INIT();
  if (!d || (n && (!srcs || !dsts))) return fail(IPK_ERR_INVALID, "null argument");
  std::vector<Context *> set; devset_snapshot(set);
  const size_t nd = set.size();
  if (nd == 1) {
    CurrentGuard guard;
    t_current = set[0];
    return ipk_host_pipeline_run_batch(d, srcs, dsts, n, out_type, used_fused);
  }
  struct Share { int rc = IPK_OK; int fused = 0; size_t frames = 0; std::string err; };
  std::vector<Share> res(nd);
  auto work = [&](size_t k) {
    std::vector<const void *> s; std::vector<void *> o;
    for (size_t i = k; i < n; i += nd) { s.push_back(srcs[i]); o.push_back(dsts[i]); }
    res[k].frames = s.size();
    if (s.empty()) return;
    t_current = set[k];                                  // this worker thread's current context
    res[k].rc = ipk_host_pipeline_run_batch(d, s.data(), o.data(), s.size(), out_type, &res[k].fused);
    if (res[k].rc < 0) res[k].err = g_err;               // ipk_last_error() is per thread: hand the text to the caller's
    t_current = nullptr;
  };
  {
    std::vector<std::thread> th;
    for (size_t k = 1; k < nd; ++k) th.emplace_back(work, k);
    { CurrentGuard guard; work(0); }                     // the caller's thread takes the first share
    for (auto &t : th) t.join();
  }
  int fused_all = 1;
  for (size_t k = 0; k < nd; ++k) {
    if (res[k].rc < 0) return fail(res[k].rc, "device-set member %zu (device %d): %s", k, set[k]->device, res[k].err.c_str());
    if (res[k].frames) fused_all = fused_all && res[k].fused;
  }
  if (used_fused) *used_fused = n ? fused_all : 0;
  return IPK_OK;
}

int ipk_host_pipeline_run_batch(const ipk_pipeline_desc *d, const void *const *srcs, void *const *dsts, size_t n, int out_type, int *used_fused) {
  REQUIRE_INIT();
  if (!d || (n && (!srcs || !dsts))) return fail(IPK_ERR_INVALID, "null argument");
  IPK_FOLD_CFA(ipk_pipeline_desc, d)
  for (size_t i = 0; i < n; ++i) if (!srcs[i] || !dsts[i]) return fail(IPK_ERR_INVALID, "null frame pointer at index %zu", i);
  size_t dw, dh, fw, fh;
  HOST_TRY(ipk_pipeline_sizes(d, &dw, &dh, &fw, &fh));
  if (n == 0) return IPK_OK;
  const bool raw = d->src_type == IPK_SRC_U16 || d->src_type == IPK_SRC_F32;
  const size_t in_bytes = d->width * d->height * (raw ? (size_t)d->cpp : 3) * src_elem_size(d->src_type);
  const size_t out_bytes = fw * fh * 3 * out_elem_size(out_type);
  HostLanes &L = cx().lanes;
  std::lock_guard<std::mutex> lk(L.mu);
  HIPCHK(hipDeviceSynchronize());   // the lanes are non-blocking streams: nothing enqueued earlier (scratch-pool users on other streams) may still be running
  HOST_TRY(L.ensure(in_bytes, out_bytes));
  // one frame's enqueues; any failure stops the batch, and the lanes are drained in every case before the return (the
  // caller's buffers must not be touched afterwards)
  auto enqueue = [&](size_t i) -> int {
    const int s = (int)(i % HostLanes::kSlots);
    const bool reused = i >= (size_t)HostLanes::kSlots;
    // upload i into slot s once the kernels of frame i - kSlots have consumed it
    if (reused) HIPCHK(hipStreamWaitEvent(L.up, L.run_done[s], 0));
    HIPCHK(hipMemcpyAsync(L.in[s], srcs[i], in_bytes, hipMemcpyHostToDevice, L.up));
    HIPCHK(hipEventRecord(L.up_done[s], L.up));
    // compute once the upload has landed and the slot's previous output has left
    HIPCHK(hipStreamWaitEvent(L.run, L.up_done[s], 0));
    if (reused) HIPCHK(hipStreamWaitEvent(L.run, L.down_done[s], 0));
    const int rc = ipk_pipeline_run(d, L.in[s], L.out[s], out_type, used_fused, L.run);
    if (rc < 0) return rc;
    HIPCHK(hipEventRecord(L.run_done[s], L.run));
    // download
    HIPCHK(hipStreamWaitEvent(L.down, L.run_done[s], 0));
    HIPCHK(hipMemcpyAsync(dsts[i], L.out[s], out_bytes, hipMemcpyDeviceToHost, L.down));
    HIPCHK(hipEventRecord(L.down_done[s], L.down));
    return IPK_OK;
  };
  int rc = IPK_OK;
  for (size_t i = 0; i < n && rc == IPK_OK; ++i) rc = enqueue(i);
  const hipError_t e0 = hipStreamSynchronize(L.up), e1 = hipStreamSynchronize(L.run), e2 = hipStreamSynchronize(L.down);
  if (rc < 0) return rc;
  HIPCHK(e0); HIPCHK(e1); HIPCHK(e2);
  return IPK_OK;
}

int ipk_host_gofloat_cfa_u16(const uint16_t *src, size_t owidth, size_t oheight, size_t x, size_t y, size_t width, size_t height,
                             float black0, float white0, float *dst) {
  REQUIRE_INIT(); DevBuf a, b;
  HOST_TRY(a.upload(src, owidth * oheight * 2)); HOST_TRY(b.alloc(width * height * 4));
  HOST_TRY(ipk_gofloat_cfa_u16(static_cast<const uint16_t *>(a.p), owidth, x, y, width, height, black0, white0, static_cast<float *>(b.p), nullptr));
  return b.download(dst, width * height * 4);
}
int ipk_host_gofloat_cfa_f32(const float *src, size_t owidth, size_t oheight, size_t x, size_t y, size_t width, size_t height,
                             float black0, float white0, float *dst) {
  REQUIRE_INIT(); DevBuf a, b;
  HOST_TRY(a.upload(src, owidth * oheight * 4)); HOST_TRY(b.alloc(width * height * 4));
  HOST_TRY(ipk_gofloat_cfa_f32(static_cast<const float *>(a.p), owidth, x, y, width, height, black0, white0, static_cast<float *>(b.p), nullptr));
  return b.download(dst, width * height * 4);
}
int ipk_host_demosaic_full(const float *src, size_t width, size_t height, const char *cfa, float *dst4) {
  REQUIRE_INIT(); DevBuf a, b;
  HOST_TRY(a.upload(src, width * height * 4)); HOST_TRY(b.alloc(width * height * 16));
  HOST_TRY(ipk_demosaic_full(static_cast<const float *>(a.p), width, height, cfa, static_cast<float *>(b.p), nullptr));
  return b.download(dst4, width * height * 16);
}
int ipk_host_transform_buffer_f32(const float *src, size_t width, size_t height, int64_t tlx, int64_t tly, int64_t trx, int64_t try_,
                                  int64_t blx, int64_t bly, size_t nwidth, size_t nheight, size_t components, const char *cfa, float *dst) {
  REQUIRE_INIT(); DevBuf a, b;
  const size_t in_comps = cfa ? 1 : components;
  HOST_TRY(a.upload(src, width * height * in_comps * 4)); HOST_TRY(b.alloc(nwidth * nheight * components * 4));
  HOST_TRY(ipk_transform_buffer_f32(static_cast<const float *>(a.p), width, height, tlx, tly, trx, try_, blx, bly, nwidth, nheight, components, cfa,
                                    static_cast<float *>(b.p), nullptr));
  return b.download(dst, nwidth * nheight * components * 4);
}
int ipk_host_tolab(const float *src4, size_t width, size_t height, int monochrome, const float *wb, const float *cm, float *dst3) {
  REQUIRE_INIT(); DevBuf a, b;
  HOST_TRY(a.upload(src4, width * height * 16)); HOST_TRY(b.alloc(width * height * 12));
  HOST_TRY(ipk_tolab(static_cast<const float *>(a.p), width, height, monochrome, wb, cm, static_cast<float *>(b.p), nullptr));
  return b.download(dst3, width * height * 12);
}
int ipk_host_basecurve(const float *src3, size_t width, size_t height, float exposure, const float *points, int npoints, float *dst3) {
  REQUIRE_INIT();
  if (curve_is_noop(exposure, npoints)) return IPK_NOOP;
  DevBuf a, b;
  HOST_TRY(a.upload(src3, width * height * 12)); HOST_TRY(b.alloc(width * height * 12));
  HOST_TRY(ipk_basecurve(static_cast<const float *>(a.p), width, height, exposure, points, npoints, static_cast<float *>(b.p), nullptr));
  return b.download(dst3, width * height * 12);
}
int ipk_host_fromlab(const float *src3, size_t width, size_t height, float *dst3) {
  REQUIRE_INIT(); DevBuf a, b;
  HOST_TRY(a.upload(src3, width * height * 12)); HOST_TRY(b.alloc(width * height * 12));
  HOST_TRY(ipk_fromlab(static_cast<const float *>(a.p), width, height, static_cast<float *>(b.p), nullptr));
  return b.download(dst3, width * height * 12);
}
int ipk_host_gamma(const float *src, size_t width, size_t height, size_t colors, int linear, float *dst) {
  REQUIRE_INIT();
  if (linear) return IPK_NOOP;
  DevBuf a, b;
  const size_t bytes = width * height * colors * 4;
  HOST_TRY(a.upload(src, bytes)); HOST_TRY(b.alloc(bytes));
  HOST_TRY(ipk_gamma(static_cast<const float *>(a.p), width, height, colors, 0, static_cast<float *>(b.p), nullptr));
  return b.download(dst, bytes);
}
int ipk_host_rotate_buffer(const float *src3, size_t width, size_t height, int orientation, float *dst3, size_t *ow, size_t *oh) {
  REQUIRE_INIT(); DevBuf a, b;
  HOST_TRY(a.upload(src3, width * height * 12)); HOST_TRY(b.alloc(width * height * 12));
  HOST_TRY(ipk_rotate_buffer(static_cast<const float *>(a.p), width, height, orientation, static_cast<float *>(b.p), ow, oh, nullptr));
  return b.download(dst3, width * height * 12);
}
int ipk_host_output8bit(const float *src, size_t n, uint8_t *dst) {
  REQUIRE_INIT(); DevBuf a, b;
  HOST_TRY(a.upload(src, n * 4)); HOST_TRY(b.alloc(n));
  HOST_TRY(ipk_output8bit(static_cast<const float *>(a.p), n, static_cast<uint8_t *>(b.p), nullptr));
  return b.download(dst, n);
}
int ipk_host_output16bit(const float *src, size_t n, uint16_t *dst) {
  REQUIRE_INIT(); DevBuf a, b;
  HOST_TRY(a.upload(src, n * 4)); HOST_TRY(b.alloc(n * 2));
  HOST_TRY(ipk_output16bit(static_cast<const float *>(a.p), n, static_cast<uint16_t *>(b.p), nullptr));
  return b.download(dst, n * 2);
}
int ipk_host_raw_to_srgb(const ipk_fused_params *p, const void *src, void *dst) {
  REQUIRE_INIT();
  if (!p || !src || !dst) return fail(IPK_ERR_INVALID, "null argument");
  IPK_FOLD_CFA(ipk_fused_params, p)
  const size_t esz = p->src_type == IPK_SRC_U16 ? 2 : 4;
  const bool band = p->band_out_rows != 0;
  const size_t src_rows = band ? p->band_src_rows : p->y + p->height;
  const size_t out_rows = band ? p->band_out_rows : p->height;
  DevBuf a, b;
  HOST_TRY(a.upload(src, src_rows * p->owidth * esz));
  HOST_TRY(b.alloc(out_rows * p->width * 3 * out_elem_size(p->out_type)));
  HOST_TRY(ipk_raw_to_srgb(p, a.p, b.p, nullptr));
  return b.download(dst, out_rows * p->width * 3 * out_elem_size(p->out_type));
}

}  // extern "C"
