/*
 * imagepipe_amd.h -- C ABI of the MI355X (gfx950) raw->sRGB hot path.
 *
 * This is the drop-in boundary: every entry point replaces one function of the reference
 * (pedrocr/imagepipe 0.5.0) behind its own `ImageOp::run` / `Pipeline::run` surface.  Each
 * declaration cites the reference interface (file:line, relative to the reference root) it
 * stands in for.  INTEGRATION.md shows the Rust `extern "C"` block and the `impl ImageOp`
 * wrappers a maintainer would add.
 *
 * Conventions
 *   - Plain pointers and sizes only.  An OpBuffer (src/buffer.rs:5-11) crosses the boundary as
 *     (data, width, height, colors[, monochrome]): f32, row-major, channel-interleaved,
 *     data[(row*width+col)*colors+c].
 *   - `ipk_*` entry points take DEVICE pointers and enqueue on `stream` (a hipStream_t, NULL =
 *     default stream).  They return as soon as the work is enqueued; buffers stay in HBM between
 *     stages.  Inputs are never modified (the reference's Arc<OpBuffer> inputs are immutable).
 *   - `ipk_host_*` entry points take HOST pointers, are synchronous at return (Rayon-join
 *     semantics of the reference ops) and do the H2D/D2H copies themselves.
 *   - Return value: IPK_OK (0), IPK_NOOP (1, "the op returned its input Arc unchanged": the
 *     output buffer was not written) or a negative ipk_status.  There is no CPU fallback: without a
 *     GPU every compute entry point fails with IPK_ERR_NO_DEVICE.
 *   - Results are bit-identical to the reference CPU path on the same host libm (the 13-bit
 *     lookup tables are built with the host's cbrtf/powf exactly as TransformLookup::new does).
 */
#ifndef IMAGEPIPE_AMD_H
#define IMAGEPIPE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IPK_API __attribute__((visibility("default")))

typedef enum {
  IPK_OK = 0,
  IPK_NOOP = 1,
  IPK_ERR_NOT_INIT = -1,
  IPK_ERR_INVALID = -2,
  IPK_ERR_HIP = -3,
  IPK_ERR_NO_DEVICE = -4,
  IPK_ERR_UNSUPPORTED = -5,
  IPK_ERR_NOMEM = -6
} ipk_status;

/* rawloader::Orientation (call sites src/ops/transform.rs:25-36,58-66,106) */
typedef enum {
  IPK_OR_NORMAL = 0, IPK_OR_HFLIP = 1, IPK_OR_ROT180 = 2, IPK_OR_VFLIP = 3, IPK_OR_TRANSPOSE = 4,
  IPK_OR_ROT90 = 5, IPK_OR_TRANSVERSE = 6, IPK_OR_ROT270 = 7, IPK_OR_UNKNOWN = 8
} ipk_orientation;

/* imagepipe::Rotation (src/ops/transform.rs:7-12) */
typedef enum { IPK_ROT_NORMAL = 0, IPK_ROT_90 = 1, IPK_ROT_180 = 2, IPK_ROT_270 = 3 } ipk_rotation;

/* element type of a raw source: RawImageData::Integer / ::Float (src/ops/gofloat.rs:94,132) and
 * the 8/16-bit raster of ImageSource::Other (src/ops/gofloat.rs:171-201) */
typedef enum { IPK_SRC_U16 = 0, IPK_SRC_F32 = 1, IPK_SRC_RGB8 = 2, IPK_SRC_RGB16 = 3 } ipk_src_type;

/* what Pipeline::run / output_8bit / output_16bit hand back (src/pipeline.rs:311,377,424) */
typedef enum { IPK_OUT_F32 = 0, IPK_OUT_U8 = 1, IPK_OUT_U16 = 2 } ipk_out_type;

/* ---------------------------------------------------------------------------------------- */
/* Context                                                                                  */
/* ---------------------------------------------------------------------------------------- */

/* Selects the HIP device, builds the three 8193-entry TransformLookup tables on the host with
 * libm (src/color_conversions.rs:87-100,119-141 -- the reference's lazy_static initialisers) and
 * uploads them.  Idempotent.  Fails with IPK_ERR_NO_DEVICE when no GPU is visible. */
IPK_API int ipk_init(int device);
IPK_API void ipk_shutdown(void);
IPK_API int ipk_is_initialized(void);
/* sizeof / offsetof of the descriptor structs as this library was built, so that a binding can check its own layout before the first call:
 * which = 0 ipk_fused_params, 1 ipk_pipeline_desc, 2 ipk_band, 3 ipk_stage_time; 16 / 17 offsetof(cfa_width) in the first two (the end of the
 * first published layout), 18 offsetof(ipk_fused_params, band_src_row0), 19 offsetof(ipk_pipeline_desc, use_fastpath), 20 / 21 offsetof(schedule) in
 * the first two (the end of the second layout); anything else 0.  No GPU needed. */
IPK_API size_t ipk_abi_sizeof(int which);
/* Human-readable description of the last failure on this thread (never NULL). */
IPK_API const char *ipk_last_error(void);
/* 1 when this host's libm cbrtf equals the device's cube-root routine (a port of glibc 2.35's, src/color_conversions.rs:103-104,123 call
 * the platform's) on the 65 536 arguments of (1, 8) ipk_init compares; 0 when it does not -- results are then still deterministic but not
 * bit-identical to a reference built on THIS host for Lab ratios above 1 (the count goes to *mismatches_of_65536, may be NULL);
 * -1 if the check could not run. */
IPK_API int ipk_host_libm_matches(size_t *mismatches_of_65536);
/* Number of compute units of the selected device (0 before ipk_init). */
IPK_API int ipk_device_cus(void);

/* ---- Several devices from ONE process --------------------------------------------------------------------------------------------
 * The reference is one process whose callers loop Pipeline::run over frames on Rayon threads (src/lib.rs:21-26, src/pipeline.rs:246-249); a
 * drop-in therefore reaches eight GPUs from one process, not through eight processes and a communicator.  A CONTEXT (ipk_ctx) is one device
 * binding plus everything the library keeps on that device (lookup tables, CFA tables, scratch pool, task queues, host-pointer lanes).
 * Every entry point of this header works on the calling thread's CURRENT context: ipk_init's context on every thread that has not chosen
 * another, else what ipk_ctx_make_current set (HIP's own per-thread current-device model; the call also binds the HIP device).  Several
 * contexts may share one device (two independent pipelines on one GPU); contexts share nothing mutable, so threads that use different
 * contexts never contend.  Device pointers belong to the device of the context that was current when they were allocated (ipk_malloc,
 * ipk_cache_new ...): pass them only to calls made under a context of that device. */
typedef struct ipk_ctx ipk_ctx;
/* A new context on `device`.  Does not change the calling thread's current context.  ipk_ctx_destroy releases its device memory (after
 * draining the device); a destroyed handle stays a valid argument and fails with IPK_ERR_NOT_INIT wherever it is used. */
IPK_API int ipk_ctx_create(int device, ipk_ctx **out);
IPK_API int ipk_ctx_destroy(ipk_ctx *ctx);
/* ctx becomes the calling thread's current context (NULL: back to the process default, ipk_init's). */
IPK_API int ipk_ctx_make_current(ipk_ctx *ctx);
IPK_API ipk_ctx *ipk_ctx_current(void);                 /* NULL before any context exists */
IPK_API int ipk_ctx_device(const ipk_ctx *ctx);         /* HIP device ordinal, -1 for a destroyed context */
/* The process's DEVICE SET: one context per entry of devices[0..n) (n = 0: one per visible device; the same ordinal may appear more than
 * once -- two contexts on one GPU).  The batch entry points below deal frame i to member i mod N.  A context ipk_init made serves as the
 * member for its device; without one, member 0 becomes the process default, so ipk_init_devices alone is a complete initialisation.
 * Calling it again replaces the set.  ipk_shutdown destroys every context of the process. */
IPK_API int ipk_init_devices(const int *devices, int n);
IPK_API int ipk_device_set_size(void);
IPK_API ipk_ctx *ipk_device_ctx(int index);             /* member `index` of the set (NULL outside it): make it current to allocate on its device */
/* The dealing rule as arithmetic (no GPU needed): of n_frames frames, set member `index` of n_devices gets frames
 * *first, *first + *stride, ... -- *count of them (frame i -> member i mod n_devices). */
IPK_API int ipk_deal_frames(size_t n_frames, int n_devices, int index, size_t *first, size_t *stride, size_t *count);

/* Device memory / stream helpers for callers that do not bring their own allocator. */
IPK_API int ipk_malloc(void **dptr, size_t bytes);
IPK_API int ipk_free(void *dptr);
IPK_API int ipk_memcpy_h2d(void *dst, const void *src, size_t bytes, void *stream);
IPK_API int ipk_memcpy_d2h(void *dst, const void *src, size_t bytes, void *stream);
IPK_API int ipk_stream_sync(void *stream);

/* Host copies of the lookup tables the device uses (8193 floats each), for fixture checks.
 * which: 0 XYZ_LAB_TRANSFORM, 1 SRGB_GAMMA_REVERSE, 2 SRGB_GAMMA_TRANSFORM
 * (src/color_conversions.rs:120,126,134).  Works without a GPU. */
IPK_API int ipk_lut_table(int which, float *out8193);

/* ---------------------------------------------------------------------------------------- */
/* Host-side size / parameter maths (no GPU needed)                                          */
/* ---------------------------------------------------------------------------------------- */

/* OpGoFloat::size_image (src/ops/gofloat.rs:74-82): out4 = x, y, width, height.
 * IPK_ERR_INVALID when owidth or oheight < 10 (the reference underflows/panics). */
IPK_API int ipk_size_image(size_t crop_top, size_t crop_right, size_t crop_bottom, size_t crop_left,
                           size_t owidth, size_t oheight, size_t *out4);
/* scaling::calculate_scaling_total (src/scaling.rs:8-23) */
IPK_API int ipk_calculate_scaling_total(size_t width, size_t height, size_t maxwidth, size_t maxheight,
                                        float *scale, size_t *nwidth, size_t *nheight);
/* OpToLab's normalize_wbs (src/ops/colorspaces.rs:12-27) */
IPK_API int ipk_normalize_wbs(const float *vals4, float *out4);
/* The colour constants of src/color_conversions.rs:1-18: which = 0 SRGB_D65_33 (9 floats), 1 XYZ_D65_33 (9; the f32
 * cofactor inverse, :20-39), 2 SRGB_D65_43 (12, [[f32;4];3]), 3 XYZ_D65_34 (12, [[f32;3];4]) */
IPK_API int ipk_const_matrix(int which, float *out12);
/* temp_to_xyz / xyz_to_temp (src/color_conversions.rs:277-310) and OpToLab::set_temp / get_temp
 * (src/ops/colorspaces.rs:59-85): xyz_to_cam12 = [[f32;3];4], cam_to_xyz12 = [[f32;4];3], row-major */
IPK_API int ipk_temp_to_xyz(float temp, float *out3);
IPK_API int ipk_xyz_to_temp(const float *xyz3, float *temp, float *tint);
IPK_API int ipk_tolab_set_temp(const float *xyz_to_cam12, float temp, float tint, float *wb4);
IPK_API int ipk_tolab_get_temp(const float *cam_to_xyz12, const float *wb4, float *temp, float *tint);
/* SplineFunc::new (src/ops/curves.rs:68-124): pts = npts (x,y) pairs; arrays sized >= npts+2.
 * Returns the knot count (>= 2) or a negative status. */
IPK_API int ipk_spline_new(const float *pts, int npts, float *px, float *py, float *c1s, float *c2s, float *c3s);
/* OpRotateCrop::calc_size (src/ops/rotatecrop.rs:111-163); params5 = crop_top, crop_right,
 * crop_bottom, crop_left, rotation */
IPK_API int ipk_rotatecrop_calc_size(const float *params5, float input_ratio, size_t width, size_t height,
                                     int reverse, size_t *nwidth, size_t *nheight);
/* rawloader CFA::shift as used by RawImage::cropped_cfa() (call site src/ops/demosaic.rs:13).
 * Pattern strings, here and in every entry point that takes one: the letters R G B E (M = G, Y = E) of one tile, row-major.  The tile's shape
 * is inferred from 4 (2x2), 36 (6x6) or 144 (12x12) letters, or STATED by the caller in front of them as "WxH:" (W columns, H rows, both
 * dividing 48 -- the reference tiles every pattern into 48x48): "2x8:RGBGRGBG..." -- the caller's CFA object knows its width and height
 * (src/ops/demosaic.rs:33).  16 letters WITHOUT a stated shape return IPK_ERR_UNSUPPORTED: rawloader's shape for them (8 wide x 2 high as
 * imagepipe's minscale arm suggests, or dcraw's 2 x 8) cannot be verified without its source, and a wrong guess would pass every test.
 * The shifted pattern comes back in the same notation (with the prefix unless the shape is one the letter count implies). */
IPK_API int ipk_cfa_shift(const char *pattern, int x, int y, char *out /* >= strlen+1 */);
/* Orientation::to_flips / from_flips (call sites src/ops/transform.rs:58-66,106);
 * flips3 = transpose, flip_x, flip_y */
IPK_API int ipk_orientation_to_flips(int orientation, int *flips3);
IPK_API int ipk_orientation_from_flips(int transpose, int flip_x, int flip_y);
/* The orientation OpTransform::run composes from (rotation, fliph, flipv) (src/ops/transform.rs:58-66) */
IPK_API int ipk_transform_orientation(int rotation, int fliph, int flipv);

/* ---------------------------------------------------------------------------------------- */
/* Stage kernels, device pointers (one per reference op)               */
/* ---------------------------------------------------------------------------------------- */

/* OpGoFloat::run_raw, CFA / cpp==1 branch (src/ops/gofloat.rs:122-130 u16, :158-166 f32):
 * dst[row][col] = min((src[owidth*(row+y)+x+col] - black0) / (white0-black0), 1), 1 channel. */
IPK_API int ipk_gofloat_cfa_u16(const uint16_t *src, size_t owidth, size_t x, size_t y, size_t width, size_t height,
                                float black0, float white0, float *dst, void *stream);
IPK_API int ipk_gofloat_cfa_f32(const float *src, size_t owidth, size_t x, size_t y, size_t width, size_t height,
                                float black0, float white0, float *dst, void *stream);
/* monochrome branch (src/ops/gofloat.rs:95-108, :133-144): 4-channel (v,v,v,0) */
IPK_API int ipk_gofloat_mono_u16(const uint16_t *src, size_t owidth, size_t x, size_t y, size_t width, size_t height,
                                 float black0, float white0, float *dst4, void *stream);
IPK_API int ipk_gofloat_mono_f32(const float *src, size_t owidth, size_t x, size_t y, size_t width, size_t height,
                                 float black0, float white0, float *dst4, void *stream);
/* cpp==3 branch (src/ops/gofloat.rs:109-120, :145-156): per-channel levels, 4-channel out.
 * black4/white4 are HOST arrays. */
IPK_API int ipk_gofloat_rgb_u16(const uint16_t *src, size_t owidth, size_t x, size_t y, size_t width, size_t height,
                                const float *black4, const float *white4, float *dst4, void *stream);
IPK_API int ipk_gofloat_rgb_f32(const float *src, size_t owidth, size_t x, size_t y, size_t width, size_t height,
                                const float *black4, const float *white4, float *dst4, void *stream);
/* OpGoFloat::run_other (src/ops/gofloat.rs:171-201): RGB8 via expand_srgb_gamma(input8bit),
 * RGB16 via input16bit; 4-channel out */
IPK_API int ipk_gofloat_other_u8(const uint8_t *src, size_t owidth, size_t x, size_t y, size_t width, size_t height,
                                 float *dst4, void *stream);
IPK_API int ipk_gofloat_other_u16(const uint16_t *src, size_t owidth, size_t x, size_t y, size_t width, size_t height,
                                  float *dst4, void *stream);

/* demosaic::full (src/ops/demosaic.rs:67-119).  `cfa` is the (cropped) pattern string OpDemosaic
 * holds (src/ops/demosaic.rs:4-22).  1-channel in, 4-channel RGBE out. */
IPK_API int ipk_demosaic_full(const float *src, size_t width, size_t height, const char *cfa, float *dst4, void *stream);
/* Row-band form for frames sharded across GPUs: `src` holds image rows [src_row0, src_row0+src_rows)
 * (the band plus its 1-row halos where they exist), output rows [out_row0, out_row0+out_rows) are
 * written to dst4 (row 0 of dst4 = image row out_row0).  Taps outside [0,img_height) are skipped
 * exactly as at the frame edge (src/ops/demosaic.rs:103-104). */
IPK_API int ipk_demosaic_full_band(const float *src, size_t width, size_t img_height, size_t src_row0, size_t src_rows,
                                   size_t out_row0, size_t out_rows, const char *cfa, float *dst4, void *stream);

/* scaling::transform_buffer<T> (src/scaling.rs:51-130): corner points are (x,y) isize pairs;
 * cfa NULL = None.  components <= 4. */
IPK_API int ipk_transform_buffer_f32(const float *src, size_t width, size_t height,
                                     int64_t tlx, int64_t tly, int64_t trx, int64_t try_, int64_t blx, int64_t bly,
                                     size_t nwidth, size_t nheight, size_t components, const char *cfa, float *dst, void *stream);
IPK_API int ipk_transform_buffer_u8(const uint8_t *src, size_t width, size_t height,
                                    int64_t tlx, int64_t tly, int64_t trx, int64_t try_, int64_t blx, int64_t bly,
                                    size_t nwidth, size_t nheight, size_t components, const char *cfa, uint8_t *dst, void *stream);
IPK_API int ipk_transform_buffer_u16(const uint16_t *src, size_t width, size_t height,
                                     int64_t tlx, int64_t tly, int64_t trx, int64_t try_, int64_t blx, int64_t bly,
                                     size_t nwidth, size_t nheight, size_t components, const char *cfa, uint16_t *dst, void *stream);
/* scaling::scaled_demosaic (src/scaling.rs:132-145) and scale_down_opbuf (:147-160) */
IPK_API int ipk_scaled_demosaic(const float *src, size_t width, size_t height, const char *cfa,
                                size_t nwidth, size_t nheight, float *dst4, void *stream);
IPK_API int ipk_scale_down_opbuf(const float *src4, size_t width, size_t height,
                                 size_t nwidth, size_t nheight, float *dst4, void *stream);
/* OpGoFloat::run_raw (CFA branch, src/ops/gofloat.rs:122-130,158-166) followed by scaling::scaled_demosaic
 * (src/scaling.rs:132-145) in one pass over the raw sensor frame; src_type IPK_SRC_U16 or IPK_SRC_F32. */
IPK_API int ipk_raw_scaled_demosaic(const void *src, int src_type, size_t owidth, size_t x, size_t y, size_t width, size_t height,
                                    float black0, float white0, const char *cfa, size_t nwidth, size_t nheight, float *dst4, void *stream);
/* The raster-source counterpart: OpGoFloat::run_other (src/ops/gofloat.rs:171-201) + scaling::scale_down_opbuf (src/scaling.rs:147-160)
 * in one pass -- what OpGoFloat and OpDemosaic (demosaic.rs:44-46: a 4-channel buffer above the size limit) compute for an RGB8 /
 * RGB16 raster, without the full-size f32 buffer.  src: owidth-pitched RGB samples, window (x, y, width, height); dst4:
 * nwidth*nheight*4 f32 (E = 0). */
IPK_API int ipk_raster_scale_down(const void *src, int src_type, size_t owidth, size_t x, size_t y, size_t width, size_t height,
                                  size_t nwidth, size_t nheight, float *dst4, void *stream);
/* OpDemosaic::run dispatch (src/ops/demosaic.rs:27-61).  colors = 1 or 4.  dst4 must hold
 * max(width*height, demosaic_width*demosaic_height)*4 floats.  IPK_NOOP = pass-through.
 * out_width / out_height receive the result size. */
IPK_API int ipk_demosaic_run(const float *src, size_t width, size_t height, size_t colors, const char *cfa,
                             size_t demosaic_width, size_t demosaic_height, float *dst4,
                             size_t *out_width, size_t *out_height, void *stream);

/* OpRotateCrop::run (src/ops/rotatecrop.rs:39-64).  Call with dst == NULL to query the output
 * size; IPK_NOOP when the op is a no-op or rejects its crops (returns the input). */
IPK_API int ipk_rotatecrop(const float *src, size_t width, size_t height, size_t colors, const float *params5,
                           float *dst, size_t *out_width, size_t *out_height, void *stream);

/* OpToLab::run (src/ops/colorspaces.rs:89-112) = camera_to_lab per pixel
 * (src/color_conversions.rs:42-55,156-169).  wb_coeffs[4] and cam_to_xyz_normalized[12]
 * ([[f32;4];3] row-major) are HOST arrays; normalize_wbs is applied here as in run(). 4ch -> 3ch. */
IPK_API int ipk_tolab(const float *src4, size_t width, size_t height, int monochrome,
                      const float *wb_coeffs, const float *cam_to_xyz_normalized, float *dst3, void *stream);
/* OpBaseCurve::run (src/ops/curves.rs:33-49): points = npoints HOST (x,y) pairs.  IPK_NOOP when
 * the op early-outs (no points and |exposure| < 0.001). */
IPK_API int ipk_basecurve(const float *src3, size_t width, size_t height, float exposure,
                          const float *points, int npoints, float *dst3, void *stream);
/* OpFromLab::run (src/ops/colorspaces.rs:127-137) = lab_to_rgb with XYZ_D65_33
 * (src/color_conversions.rs:58-65,172-191) */
IPK_API int ipk_fromlab(const float *src3, size_t width, size_t height, float *dst3, void *stream);
/* OpGamma::run (src/ops/gamma.rs:16-26); IPK_NOOP when linear */
IPK_API int ipk_gamma(const float *src, size_t width, size_t height, size_t colors, int linear, float *dst, void *stream);
/* rotate_buffer (src/ops/transform.rs:87-144), 3-channel */
IPK_API int ipk_rotate_buffer(const float *src3, size_t width, size_t height, int orientation,
                              float *dst3, size_t *out_width, size_t *out_height, void *stream);
/* rotate_buffer's permutation (src/ops/transform.rs:87-144) applied to an already quantised 3-channel image: output8bit /
 * output16bit (src/pipeline.rs:408-414,455-461) act per sample, so quantise-then-rotate equals the reference's rotate-then-
 * quantise bit for bit while the permutation moves 4x / 2x fewer bytes.  ipk_pipeline_run orders the two this way for the
 * 8- and 16-bit outputs of non-Normal orientations. */
IPK_API int ipk_rotate_image_u8(const uint8_t *src3, size_t width, size_t height, int orientation, uint8_t *dst3,
                                size_t *out_width, size_t *out_height, void *stream);
IPK_API int ipk_rotate_image_u16(const uint16_t *src3, size_t width, size_t height, int orientation, uint16_t *dst3,
                                 size_t *out_width, size_t *out_height, void *stream);
/* OpTransform::run (src/ops/transform.rs:56-73); IPK_NOOP for Normal/Unknown */
IPK_API int ipk_transform(const float *src3, size_t width, size_t height, int rotation, int fliph, int flipv,
                          float *dst3, size_t *out_width, size_t *out_height, void *stream);
/* the quantise loops of Pipeline::output_8bit/16bit (src/pipeline.rs:408-414, :455-461;
 * output8bit/output16bit src/color_conversions.rs:323-330) */
IPK_API int ipk_output8bit(const float *src, size_t n, uint8_t *dst, void *stream);
IPK_API int ipk_output16bit(const float *src, size_t n, uint16_t *dst, void *stream);

/* ---------------------------------------------------------------------------------------- */
/* Fused raw -> sRGB (gofloat + demosaic::full + tolab + basecurve + fromlab + gamma [+ quantise]) */
/* ---------------------------------------------------------------------------------------- */

/* How a fused launch deals a single frame's row segments to its waves.  The reference's Rayon loops balance themselves by work stealing
 * (src/ops/demosaic.rs:93-116: one task per output row); a persistent GPU launch fixes each wave's rows up front, so a frame whose regions cost
 * differently -- blown highlights take the cube-root path of XYZ_LAB_TRANSFORM.lookup, src/color_conversions.rs:102-104 -- finishes on its
 * slowest waves.  IPK_SCHED_SPLIT gives every wave two pieces half a frame apart instead of one contiguous piece: a blown region is shared by
 * twice as many waves (photo-like 100 MP frame: -2...-3.4 %), at one more three-row priming per wave (uniform frames: +1...+5 %).  A caller
 * that knows its frames (a shoot with clipped skies) sets it; AUTO is the contiguous schedule.  Batches and frames above ~130 MP draw their
 * tasks from a queue and ignore the field.  Results are bit-identical under every schedule. */
typedef enum { IPK_SCHED_AUTO = 0, IPK_SCHED_SPLIT = 1 } ipk_schedule;

/* Everything the ops between OpGoFloat and OpGamma read, for a CFA raw source at full scale with
 * a no-op rotatecrop (the default Pipeline::run on a RawImage). */
typedef struct {
  uint32_t struct_size;            /* sizeof(ipk_fused_params) as the CALLER was compiled (IPK_FUSED_PARAMS_INIT sets it): see "Descriptor versioning" below */
  int src_type;                    /* IPK_SRC_U16 or IPK_SRC_F32 */
  size_t owidth;                   /* RawImage.width: row pitch of src in elements */
  size_t x, y, width, height;      /* OpGoFloat::size_image result */
  float black0, white0;            /* blacklevels[0], whitelevels[0] (gofloat.rs:126) */
  char cfa[160];                   /* cropped CFA pattern (OpDemosaic.cfa): letters, or "WxH:letters" (see ipk_cfa_shift) */
  float wb_coeffs[4];              /* OpToLab.wb_coeffs (normalised again inside, colorspaces.rs:100) */
  float cam_to_xyz_normalized[12]; /* OpToLab.cam_to_xyz_normalized, [[f32;4];3] row-major */
  float exposure;                  /* OpBaseCurve.exposure */
  int npoints;                     /* OpBaseCurve.points */
  float points[128];
  int linear;                      /* PipelineSettings.linear: skip OpGamma */
  int out_type;                    /* IPK_OUT_F32 (Pipeline::run), IPK_OUT_U8 / IPK_OUT_U16 (+ output8bit/16bit) */
  /* row band (multi-GPU sharding of one frame); all zero = whole frame.  `src` then points at
   * image row band_src_row0 (in cropped coordinates, i.e. row y+band_src_row0 of the sensor), and
   * only output rows [band_out_row0, band_out_row0+band_out_rows) are produced into dst. */
  size_t band_src_row0, band_src_rows, band_out_row0, band_out_rows;
  /* Descriptor versioning.  Fields added after a struct was first published are APPENDED, never inserted, and the leading struct_size says how many
   * bytes of it the caller's object really has: the library copies exactly that many into a zeroed struct of its own and never reads past them, so a
   * caller compiled against an older header (an object that ends in front of the fields below) keeps working with the defaults of the fields it
   * does not know -- 0, 0 below: "take the shape from the string".  A struct_size smaller than the first published layout (ipk_abi_sizeof(16) / (17):
   * everything in front of cfa_width), or LARGER than this library's struct (a caller built against a newer header: fields this library would
   * silently ignore), fails with IPK_ERR_INVALID; so does 0 (an object that was never initialised).  ipk_abi_sizeof(0) / (1) still let a binding
   * compare whole layouts at start-up (tests/test_rust_binding.py checks the generated Rust layout the same way). */
  int cfa_width, cfa_height;       /* the tile's shape as the caller's CFA object has it (cfa.width / cfa.height, src/ops/demosaic.rs:33); 0, 0 = take
                                      it from the string (a "WxH:" prefix, or the letter count 4 / 36 / 144).  16 letters need one of the two. */
  /* third layout */
  int schedule;                    /* ipk_schedule: how the launch shares a frame's rows out among the waves (results do not depend on it) */
  int reserved0;                   /* 0 */
} ipk_fused_params;
#define IPK_FUSED_PARAMS_INIT {(uint32_t)sizeof(ipk_fused_params)}      /* ipk_fused_params p = IPK_FUSED_PARAMS_INIT;  (everything else zero) */

/* src: device pointer to the sensor data (element (0,0) of the uncropped frame, or of the band's
 * first source row); dst: device pointer to width*rows*3 elements of out_type.
 * Any colour filter without a fourth colour is accepted (the four RGGB phases, X-Trans, 12x12, 16 letters with a stated shape ...; pattern strings: see ipk_cfa_shift); fails with
 * IPK_ERR_UNSUPPORTED for RGBE-style filters (callers then run the staged ops). */
IPK_API int ipk_raw_to_srgb(const ipk_fused_params *p, const void *src, void *dst, void *stream);
/* n frames of ONE shape and ONE parameter set (a caller looping Pipeline::run over a shoot, src/pipeline.rs:246-249: frames are independent;
 * BASELINE.json configs[3]): srcs[i] -> dsts[i], host arrays of device pointers, whole frames only.  Where the kernel has a batch variant
 * (Bayer filter, ordinary levels / multipliers / matrix, a 2- or 3-knot curve, width >= 256; any frame size: eight 100 MP frames in one launch run at 0.51 ms each) the frames run as ONE persistent launch per 64
 * -- the lookup tables are staged once per CU instead of once per frame and CU, and no CU idles between frames; otherwise one launch per
 * frame.  Either way every dsts[i] is bit-identical to ipk_raw_to_srgb(p, srcs[i], dsts[i]). */
IPK_API int ipk_raw_to_srgb_batch(const ipk_fused_params *p, const void *const *srcs, void *const *dsts, size_t n, void *stream);
/* ipk_raw_to_srgb followed by OpTransform (src/ops/transform.rs:56-73) for the given rawloader orientation, without a pass over the
 * 3-channel result: the 1-channel mosaic is permuted instead (rotate_buffer's index walk, :102-128) and the kernel works in
 * rotated space, adding the demosaic taps in the reference's order of the ORIGINAL orientation.  Whole frames (no band) of a
 * three-colour filter whose rotated width is at least 256 pixels, with the common parameter set (finite ordinary multipliers and
 * matrix, a 2- or 3-knot curve, validated levels).  dst receives out_width x out_height x 3 samples of p->out_type.
 * Normal / Unknown: ipk_raw_to_srgb.  Returns IPK_ERR_UNSUPPORTED (nothing written) otherwise: the caller then uses
 * ipk_raw_to_srgb + ipk_rotate_buffer / ipk_rotate_image_*, as ipk_pipeline_run does. */
IPK_API int ipk_raw_to_srgb_oriented(const ipk_fused_params *p, const void *src, int orientation, void *dst,
                                     size_t *out_width, size_t *out_height, void *stream);

/* OpToLab::run + OpBaseCurve::run + OpFromLab::run + OpGamma::run (src/ops/colorspaces.rs:89-112, src/ops/curves.rs:33-49,
 * src/ops/colorspaces.rs:127-137, src/ops/gamma.rs:16-26) in one pass over a 4-channel OpBuffer: the ops Pipeline::run applies
 * between rotatecrop and transform, for callers that do not need the three intermediate buffers (cache == None).
 * Arguments as for the four stage entry points; bit-identical to running them one after the other. */
IPK_API int ipk_pointwise_chain(const float *src4, size_t width, size_t height, int monochrome, const float *wb_coeffs,
                                const float *cam_to_xyz_normalized, float exposure, const float *points, int npoints, int linear,
                                float *dst3, void *stream);
/* The same followed by the quantise loop of output_8bit / output_16bit (src/pipeline.rs:408-414, :455-461) in ONE pass: dst receives width*height*3 samples
 * of out_type (IPK_OUT_U8 / IPK_OUT_U16; IPK_OUT_F32 = ipk_pointwise_chain).  What ipk_pipeline_run uses behind a staged demosaic when the caller wants
 * 8 or 16 bits and OpTransform is a no-op (a preview under a size limit, X-Trans, four-colour filters): bit-identical to ipk_pointwise_chain followed by
 * ipk_output8bit / ipk_output16bit.  At least 256 pixels (IPK_ERR_UNSUPPORTED below). */
IPK_API int ipk_pointwise_chain_out(const float *src4, size_t width, size_t height, int monochrome, const float *wb_coeffs,
                                    const float *cam_to_xyz_normalized, float exposure, const float *points, int npoints, int linear,
                                    int out_type, void *dst, void *stream);
/* The raster-source counterpart of ipk_raw_to_srgb: OpGoFloat::run_other (src/ops/gofloat.rs:171-201; RGB8 through
 * expand_srgb_gamma(input8bit(v)), RGB16 through input16bit(v)) + OpToLab + OpBaseCurve + OpFromLab + OpGamma (+ output8bit /
 * output16bit, src/pipeline.rs:408-414,455-461) in one pass from the width*height*3 source samples to width*height*3 outputs of
 * out_type.  What Pipeline::run / output_8bit / output_16bit compute for an ImageSource::Other whose ops are not the defaults
 * (the default ops take the integer fast path) when demosaic, rotatecrop and transform are no-ops; ipk_pipeline_run uses it
 * then (allow_fused).  At least 256 pixels. */
IPK_API int ipk_raster_to_srgb(const void *src, int src_type, size_t width, size_t height, const float *wb_coeffs,
                               const float *cam_to_xyz_normalized, float exposure, const float *points, int npoints, int linear,
                               int out_type, void *dst, void *stream);

/* ---------------------------------------------------------------------------------------- */
/* Pipeline driver: Pipeline::run / output_8bit / output_16bit for one source                */
/* (src/pipeline.rs:311-375, :377-422, :424-469), cache == None                              */
/* ---------------------------------------------------------------------------------------- */

/* The fields of ImageSource + PipelineOps + PipelineSettings the hot path reads. */
typedef struct {
  uint32_t struct_size;            /* sizeof(ipk_pipeline_desc) as the caller was compiled (IPK_PIPELINE_DESC_INIT); see ipk_fused_params, "Descriptor versioning" */
  int src_type;                    /* ipk_src_type */
  size_t width, height;            /* RawImage.width/height or raster dims */
  int cpp;                         /* RawImage.cpp (1 or 3); ignored for RGB8/RGB16 */
  int is_cfa;                      /* OpGoFloat.is_cfa */
  char cfa[160];                   /* OpDemosaic.cfa = cropped_cfa() ("" for Other): letters, or "WxH:letters" */
  size_t crop_top, crop_right, crop_bottom, crop_left;   /* OpGoFloat crops */
  float blacklevels[4], whitelevels[4];
  float rotatecrop[5];             /* OpRotateCrop: crop_top, crop_right, crop_bottom, crop_left, rotation */
  float cam_to_xyz_normalized[12];
  float wb_coeffs[4];
  float exposure; int npoints; float points[128];        /* OpBaseCurve */
  int rotation, fliph, flipv;      /* OpTransform */
  size_t maxwidth, maxheight;      /* PipelineSettings */
  int linear;
  int allow_fused;                 /* 1: use ipk_raw_to_srgb when legal (cache==None); 0: always staged */
  int use_fastpath;                /* PipelineSettings.use_fastpath (pipeline.rs:117; the reference defaults it to true) */
  /* later additions are appended (see ipk_fused_params) */
  int cfa_width, cfa_height;       /* as in ipk_fused_params: the tile's shape from the caller's CFA object, 0, 0 = from the string */
  /* third layout */
  int schedule;                    /* ipk_schedule, handed to the fused launch when the run is one */
  int reserved0;                   /* 0 */
  int reserved1, reserved2;        /* 0 (the struct's end moves past the second layout's tail padding, so that a size tells the layouts apart) */
} ipk_pipeline_desc;
#define IPK_PIPELINE_DESC_INIT {(uint32_t)sizeof(ipk_pipeline_desc)}

/* Size negotiation of Pipeline::run (src/pipeline.rs:314-338): demosaic_{w,h} as stored in the
 * settings, and the size of the buffer run() returns.  The ops size their outputs from the buffer
 * they are handed (demosaic.rs:60-90, rotatecrop.rs:41-57, transform.rs:46-60), so behind a
 * rotatecrop the returned buffer can be a pixel smaller than the forward fold of
 * transform_forward() the negotiation itself uses; final_{w,h} is what run() produces.
 * A side that scales to 0 is reported as 0 and ipk_pipeline_run rejects it.  No GPU needed. */
IPK_API int ipk_pipeline_sizes(const ipk_pipeline_desc *d, size_t *demosaic_w, size_t *demosaic_h,
                               size_t *final_w, size_t *final_h);
/* Runs the ops in the reference's fixed order (src/pipeline.rs:155-164,364-372) on a DEVICE source
 * buffer and writes the result (final_w*final_h*3 elements of out_type) to the DEVICE buffer dst.
 * out_type IPK_OUT_F32 = Pipeline::run with d->linear; IPK_OUT_U8 = output_8bit (forces
 * linear=false, pipeline.rs:405); IPK_OUT_U16 = output_16bit (forces linear=true, :452).
 * *used_fused (may be NULL) reports which path ran. */
IPK_API int ipk_pipeline_run(const ipk_pipeline_desc *d, const void *src, void *dst, int out_type,
                             int *used_fused, void *stream);
/* Raster fast path of output_8bit / output_16bit (pipeline.rs:381-402, :428-449): when d->use_fastpath is set, the
 * source is RGB8/RGB16, out_type is U8/U16 and the ops are the defaults for a raster source (Pipeline::default_ops,
 * :286-288, compared bitwise), ipk_pipeline_run / _cached skip the float pipeline: DynamicImage::to_rgb8/16 channel
 * conversion if the depths differ (image 0.24: c*257, (c+128)/257 -- crate absent, parity unpinned) and
 * scale_down_srgb / scale_down_srgb16 (scaling.rs:162-182) if a size limit applies.  Returns 1 / 0. */
IPK_API int ipk_pipeline_takes_fastpath(const ipk_pipeline_desc *d, int out_type);
/* do_timing! (src/pipeline.rs:68-80: the reference logs the wall time of every op of Pipeline::run): ipk_timing_begin arms the calling
 * thread, the following ipk_pipeline_run call(s) on it bracket every stage they enqueue with hipEvents on their stream, ipk_timing_end
 * waits for the last one and returns the stages in execution order under the reference's op names ("gofloat", "demosaic", "rotatecrop",
 * "to_lab", "basecurve", "from_lab", "gamma", "transform"; merged stages are joined with '+', the one-launch paths start with "fused",
 * the output loops are "quantise").  *n_stages receives the number of stages (may exceed max_stages; the surplus is dropped). */
typedef struct { char name[88]; float ms; } ipk_stage_time;
IPK_API int ipk_timing_begin(void);
IPK_API int ipk_timing_end(ipk_stage_time *out, int max_stages, int *n_stages);
/* Same with HOST source and destination buffers; synchronous. */
IPK_API int ipk_host_pipeline_run(const ipk_pipeline_desc *d, const void *src, void *dst, int out_type, int *used_fused);
/* A batch of n same-shaped HOST frames through one descriptor (a caller looping Pipeline::run / output_8bit over a shoot,
 * src/pipeline.rs:311-372, :404-421): three HIP streams (upload, compute, download) over two device slots, so frame i's kernels
 * run while frame i+1 crosses PCIe upwards and frame i-1 downwards; per-frame cost tends to max(upload, compute, download)
 * instead of their sum.  The copies only overlap for page-locked host memory (ipk_host_alloc below, or memory the caller has
 * registered); pageable buffers work but serialise.  Results are what n calls of ipk_host_pipeline_run give.  Synchronous at return. */
IPK_API int ipk_host_pipeline_run_batch(const ipk_pipeline_desc *d, const void *const *srcs, void *const *dsts, size_t n,
                                        int out_type, int *used_fused);
/* n same-shaped DEVICE frames through one descriptor on the current context (the device-pointer form of the above, enqueued on `stream`):
 * where Pipeline::run is exactly one fused launch per frame the batch is one persistent launch per 64 frames (ipk_raw_to_srgb_batch), otherwise
 * n calls of ipk_pipeline_run.  Results are what n calls of ipk_pipeline_run give. */
IPK_API int ipk_pipeline_run_batch(const ipk_pipeline_desc *d, const void *const *srcs, void *const *dsts, size_t n, int out_type,
                                   int *used_fused, void *stream);
/* The same over the DEVICE SET (ipk_init_devices) from one host thread: frame i runs on set member i mod N, so srcs[i] / dsts[i] must live on
 * that member's device.  Nothing is exchanged between devices and every result stays where it was computed (frames are independent pipelines,
 * src/pipeline.rs:246-249).  Enqueues on one internal stream per member and returns; ipk_devices_sync waits for all of them.  Without a device
 * set the current context takes every frame. */
IPK_API int ipk_pipeline_run_batch_multi(const ipk_pipeline_desc *d, const void *const *srcs, void *const *dsts, size_t n, int out_type,
                                         int *used_fused);
IPK_API int ipk_devices_sync(void);
/* HOST frames in, HOST results out, over the device set: the drop-in for a caller looping Pipeline::run / output_8bit over a shoot on a
 * multi-GPU node.  One host thread per member runs ipk_host_pipeline_run_batch on the frames dealt to it (frame i -> member i mod N), so all
 * GPUs upload, compute and download concurrently; the caller's Vecs are the "gather".  Synchronous at return; results are what n calls of
 * ipk_host_pipeline_run give.  Use ipk_host_alloc buffers (page-locked and visible to every device). */
IPK_API int ipk_host_pipeline_run_batch_multi(const ipk_pipeline_desc *d, const void *const *srcs, void *const *dsts, size_t n,
                                              int out_type, int *used_fused);
/* Page-locked host memory for the buffers handed to the ipk_host_* entry points (NULL on failure). */
IPK_API void *ipk_host_alloc(size_t bytes);
IPK_API void ipk_host_free(void *p);

/* ---------------------------------------------------------------------------------------- */
/* Stage-boundary caching contract of Pipeline::run(Some(cache)) (src/pipeline.rs:341-372;    */
/* BufHasher src/hasher.rs:12-48; PipelineCache = MultiCache<BufHash, OpBuffer>, :43,258-260)  */
/* ---------------------------------------------------------------------------------------- */

/* The eight chained op hashes (32 bytes each, op order gofloat..transform) Pipeline::run computes after
 * size negotiation: hash i covers PipelineSettings and ops 0..i, so editing op k changes hashes k..7 only.
 * out_type selects the `linear` the settings carry (output_8bit / output_16bit force it).  source_id is an
 * extension: the reference hashes nothing that identifies the image; a nonzero id lets one cache hold the
 * buffers of several resident frames.  Host-only. */
IPK_API int ipk_pipeline_hashes(const ipk_pipeline_desc *d, int out_type, uint64_t source_id, uint8_t *out256);

/* Pipeline::new_cache(size): a byte-budgeted LRU of device OpBuffers.  `get` refreshes recency; `put` evicts
 * least-recently-used entries until the newcomer fits.  Use one stream per cache. */
typedef struct ipk_cache ipk_cache;
IPK_API int ipk_cache_new(size_t max_bytes, ipk_cache **out);
IPK_API int ipk_cache_free(ipk_cache *cache);
IPK_API int ipk_cache_clear(ipk_cache *cache);
IPK_API int ipk_cache_contains(const ipk_cache *cache, const uint8_t *key32);                 /* 1 / 0 */
IPK_API int ipk_cache_stats(const ipk_cache *cache, size_t *bytes, size_t *entries, uint64_t *hits, uint64_t *misses, uint64_t *evictions);
/* Borrow a memoised buffer (IPK_NOOP when absent).  The pointer stays valid until the entry is evicted. */
IPK_API int ipk_cache_get(ipk_cache *cache, const uint8_t *key32, const float **data, size_t *width, size_t *height,
                          size_t *colors, int *monochrome);

/* Pipeline::run / output_8bit / output_16bit with Some(cache): resumes after the last op whose hash is in the
 * cache, stores every buffer it materialises under that op's hash, writes the final image to the DEVICE buffer
 * dst.  *ops_run (may be NULL) = bit i set when op i executed (0 = served from the cache).  With nothing
 * memoised and a fusable chain the whole run is one ipk_raw_to_srgb launch and only the final buffer is stored. */
IPK_API int ipk_pipeline_run_cached(const ipk_pipeline_desc *d, const void *src, uint64_t source_id, ipk_cache *cache,
                                    int out_type, void *dst, int *ops_run, int *used_fused, void *stream);

/* ---------------------------------------------------------------------------------------- */
/* Multi-GPU: one process per GPU; a frame batch shards with no exchange (frame i -> rank i mod N, */
/* src/pipeline.rs:246-249), ONE frame shards by row bands                */
/* ---------------------------------------------------------------------------------------- */

/* One rank's row band.  Full-resolution path (ipk_raw_to_srgb's band_* fields): output rows [out_row0, +out_rows) of the cropped
 * frame, and the source rows [src_row0, +src_rows) its slab must hold = the band plus the 1-row halos demosaic::full taps
 * (src/ops/demosaic.rs:70-74) where they lie inside the frame.  Scaled path: out_* count rows of the nwidth x nheight result and
 * src_* the source rows its windows read (src/scaling.rs:84-94). */
typedef struct { size_t out_row0, out_rows, src_row0, src_rows; } ipk_band;

/* Row bands of near-equal size whose boundaries are multiples of the CFA period (2 Bayer, 6 X-Trans, 12), so every band starts
 * at the frame's CFA phase.  bands[nranks].  Host-only. */
IPK_API int ipk_band_plan(size_t height, int nranks, int cfa_period, ipk_band *bands);
/* Output-row bands of scaling::scaled_demosaic (src/scaling.rs:132-145) from width x height to nwidth x nheight: rank k produces
 * output rows [out_row0, +out_rows) and needs source rows floor(skip*r0) .. floor(skip*r1) (:84-87, clamped to the frame) --
 * neighbouring ranks' source ranges overlap by a window, nothing is exchanged.  Host-only. */
IPK_API int ipk_band_plan_scaled(size_t height, size_t nheight, int nranks, ipk_band *bands);
/* ipk_raw_scaled_demosaic for one band of ipk_band_plan_scaled: `src` points at the slab's first row (sensor row y + band->src_row0,
 * column 0 of the sensor frame), dst4 receives band->out_rows x nwidth RGBE pixels.  Bit-identical to the same rows of the whole frame. */
IPK_API int ipk_raw_scaled_demosaic_band(const void *src, int src_type, size_t owidth, size_t x, size_t width, size_t height,
                                         float black0, float white0, const char *cfa, size_t nwidth, size_t nheight,
                                         const ipk_band *band, float *dst4, void *stream);

/* A communicator over the ranks that share one frame.  Two transports behind the same entry points:
 *   RCCL   (ipk_comm_init_rccl): ncclSend/ncclRecv/ncclAllGather on device buffers over xGMI -- the product path on a multi-GPU node;
 *   host   (ipk_comm_init_host): the caller's own messaging (MPI, gloo, a socket) moves the bytes, staged through host memory --
 *          for hosts that already have a fabric of their own, and for ranks that share ONE GPU (RCCL refuses two ranks per device).
 * Nothing here touches pixels: results are bit-identical to the unsharded run by construction of the band kernels. */
typedef struct ipk_comm ipk_comm;
#define IPK_COMM_ID_BYTES 128
/* ncclGetUniqueId on one rank; the caller hands the 128 bytes to every rank by its own means (it has to rendezvous anyway) */
IPK_API int ipk_comm_unique_id(uint8_t *id128);
/* ncclCommInitRank on the device ipk_init bound.  Collective over the nranks processes. */
IPK_API int ipk_comm_init_rccl(const uint8_t *id128, int rank, int nranks, ipk_comm **out);
/* Host transport.  exchange(ctx, send_peer, send, send_bytes, recv_peer, recv, recv_bytes) must send `send` to send_peer and
 * receive recv_bytes from recv_peer into `recv` (either peer may be -1 = none), returning 0 on success; it may block until both
 * complete (MPI_Sendrecv semantics).  Buffers are host memory. */
typedef int (*ipk_exchange_fn)(void *ctx, int send_peer, const void *send, size_t send_bytes, int recv_peer, void *recv, size_t recv_bytes);
IPK_API int ipk_comm_init_host(int rank, int nranks, ipk_exchange_fn exchange, void *ctx, ipk_comm **out);
IPK_API int ipk_comm_free(ipk_comm *comm);
/* transport: 0 RCCL, 1 host */
IPK_API int ipk_comm_info(const ipk_comm *comm, int *rank, int *nranks, int *transport);
/* Halo exchange of the full-resolution path, in place on the slab (device memory, row pitch row_bytes, rows = bands[rank].src_rows):
 * the band's first / last own row goes to the neighbour above / below, their edge rows arrive in the slab's halo rows.  RCCL: one
 * ncclGroup of ncclSend/ncclRecv on `stream` (asynchronous); host transport: synchronous.  Bands with no rows are skipped. */
IPK_API int ipk_band_exchange_halo(ipk_comm *comm, void *slab, size_t row_bytes, const ipk_band *bands, void *stream);
/* The same on a HOST slab (host transport only; what a caller does before uploading its band). */
IPK_API int ipk_host_band_exchange_halo(ipk_comm *comm, void *slab, size_t row_bytes, const ipk_band *bands);
/* Reassembles the result IN PLACE: `frame` is the full out_height x out_row_bytes image on every receiving rank, and each rank's
 * kernel has already written its own band at its rows (dst = frame + bands[rank].out_row0 * out_row_bytes).  root = -1: every rank
 * receives every band (ncclAllGather when the bands are equal, one group of ncclSend/ncclRecv straight into place otherwise: the xGMI
 * mesh is point-to-point, all peers at once); root >= 0: only that rank receives.  Any element type: sizes are bytes. */
IPK_API int ipk_band_gather(ipk_comm *comm, void *frame, size_t out_row_bytes, const ipk_band *bands, int root, void *stream);
/* The same on a HOST frame (host transport only). */
IPK_API int ipk_host_band_gather(ipk_comm *comm, void *frame, size_t out_row_bytes, const ipk_band *bands, int root);
/* The gather on the communicator's own stream, ordered after the work `after_stream` holds now: frame k's gather overlaps frame
 * k+1's kernel.  ipk_comm_wait makes `stream` wait for the gathers begun so far (no host block). */
IPK_API int ipk_band_gather_begin(ipk_comm *comm, void *frame, size_t out_row_bytes, const ipk_band *bands, int root, void *after_stream);
IPK_API int ipk_comm_wait(ipk_comm *comm, void *stream);
/* Transport self-check: a ring send/recv and an all-gather of a known pattern on device buffers, verified on the host.  Collective. */
IPK_API int ipk_comm_selftest(ipk_comm *comm);

/* ---------------------------------------------------------------------------------------- */
/* Host-pointer forms of the stage kernels: what a Rust `impl ImageOp::run` binds when it keeps   */
/* its OpBuffers in host Vec<f32>s.  Same arguments as the device forms, minus the stream.       */
/* ---------------------------------------------------------------------------------------- */
IPK_API int ipk_host_gofloat_cfa_u16(const uint16_t *src, size_t owidth, size_t oheight, size_t x, size_t y, size_t width, size_t height,
                                     float black0, float white0, float *dst);
IPK_API int ipk_host_gofloat_cfa_f32(const float *src, size_t owidth, size_t oheight, size_t x, size_t y, size_t width, size_t height,
                                     float black0, float white0, float *dst);
IPK_API int ipk_host_demosaic_full(const float *src, size_t width, size_t height, const char *cfa, float *dst4);
IPK_API int ipk_host_transform_buffer_f32(const float *src, size_t width, size_t height,
                                          int64_t tlx, int64_t tly, int64_t trx, int64_t try_, int64_t blx, int64_t bly,
                                          size_t nwidth, size_t nheight, size_t components, const char *cfa, float *dst);
IPK_API int ipk_host_tolab(const float *src4, size_t width, size_t height, int monochrome,
                           const float *wb_coeffs, const float *cam_to_xyz_normalized, float *dst3);
IPK_API int ipk_host_basecurve(const float *src3, size_t width, size_t height, float exposure,
                               const float *points, int npoints, float *dst3);
IPK_API int ipk_host_fromlab(const float *src3, size_t width, size_t height, float *dst3);
IPK_API int ipk_host_gamma(const float *src, size_t width, size_t height, size_t colors, int linear, float *dst);
IPK_API int ipk_host_rotate_buffer(const float *src3, size_t width, size_t height, int orientation,
                                   float *dst3, size_t *out_width, size_t *out_height);
IPK_API int ipk_host_output8bit(const float *src, size_t n, uint8_t *dst);
IPK_API int ipk_host_output16bit(const float *src, size_t n, uint16_t *dst);
IPK_API int ipk_host_raw_to_srgb(const ipk_fused_params *p, const void *src, void *dst);

/* ---------------------------------------------------------------------------------------- */
/* Self-test hooks: on-device, exhaustive checks that the arithmetic shortcuts of the fused     */
/* kernel reproduce the plain IEEE expressions of the reference (used by tests/, not by callers) */
/* ---------------------------------------------------------------------------------------- */
/* x / c (src/color_conversions.rs:158,168,177-179,184-187; src/ops/gofloat.rs:126) versus the kernel's
 * multiply/fma forms, over every f32 x with lo <= |x| <= hi (and 0/inf/NaN if include_special).
 * variant 0: cdiv_fast incl. v_div_fixup; 1: the three arithmetic steps; 2: two steps, hi/lo reciprocal. */
IPK_API int ipk_selftest_cdiv(float c, int variant, float lo, float hi, int include_special,
                              uint64_t *n_bad, uint32_t *first_bad_bits);
/* pos - pos.trunc() (src/color_conversions.rs:108-109) versus v_fract_f32 for every pos in [0, 8192] */
IPK_API int ipk_selftest_lut_weight(uint64_t *n_bad, uint32_t *first_bad_bits);
/* v.max(0.0).min(1.0) (src/ops/gamma.rs:22) versus v_med3_f32(v,0,1) for every f32 */
IPK_API int ipk_selftest_clamp01(uint64_t *n_bad, uint32_t *first_bad_bits);
/* Measurement aid, no counterpart in the reference: a plain device-to-device copy, 16 bytes per lane, as the practical HBM ceiling next to which
 * bench.py reports the kernels' achieved bandwidth (SURVEY.md 8d asks for the measured copy / triad ceiling beside the 8 TB/s spec peak). */
IPK_API int ipk_copy_probe(const void *src, void *dst, size_t bytes, void *stream);
/* The same for the fused path's read : write mix: src_bytes read, 3 * src_bytes written (4 : 12 bytes per pixel, f32 mosaic -> f32 RGB), flat launch,
 * contiguous, nontemporal, no arithmetic: the ceiling of ANY kernel with that traffic (bench.py roofline.mix_ceiling_GBps).  dst holds 3 * src_bytes;
 * src_bytes a multiple of 4096 (whole blocks of 256 lanes x 16 bytes), both pointers 16-byte aligned. */
IPK_API int ipk_mix_probe(const void *src, void *dst, size_t src_bytes, void *stream);
/* Measurement aid: the shader clock WHILE other work runs.  One wave on `stream` spins for spin_us microseconds of the fixed 100 MHz reference counter
 * (s_memrealtime) and writes {shader-clock cycles (s_memtime), reference ticks} over that span to out2_dev (two uint64 in device memory): clock in GHz =
 * out2[0] / out2[1] / 10.  bench.py launches it on a second stream beside the kernel under test (config.shader_clock_GHz), so that a slower box and a
 * slower kernel can be told apart. */
IPK_API int ipk_clock_probe(uint64_t *out2_dev, uint32_t spin_us, void *stream);
/* Measurement aid, no counterpart in the reference: the memory skeleton of ipk_raw_to_srgb as a launch of its own -- the same persistent launch and
 * task walk, the same row loads, OpGoFloat normalisation (src/ops/gofloat.rs:126), demosaic::full (src/ops/demosaic.rs:67-119), LDS staging and
 * nontemporal stores, WITHOUT OpToLab..OpGamma: dst receives the demosaiced R, G, B channels (the first three of demosaic::full's RGBE pixel) as
 * width*height*3 f32 samples, whatever p->out_type says.  4 (u16: 2) bytes in and 12 out per pixel on the fused kernel's own access pattern:
 * bench.py times it in the same session as the fused kernel and reports roofline.ceiling_ms / frac_of_ceiling from it.  Whole Bayer frames of at
 * least 256 columns with levels the fast normalisation is validated for; IPK_ERR_UNSUPPORTED otherwise. */
IPK_API int ipk_stream_probe(const ipk_fused_params *p, const void *src, void *dst, void *stream);
/* SplineFunc::interpolate (src/ops/curves.rs:126-157) on every f32 against the form the fused kernels use for a base curve of 2 or 3 knots (lower
 * clamp and exact knot hit as arithmetic, ipk_device.hpp spline_interpolate_3a); the curve is given like ipk_basecurve's.  IPK_ERR_UNSUPPORTED when
 * the kernels would not use that form for this curve (more knots, a knot ordinate of -0.0, non-finite coefficients). */
IPK_API int ipk_selftest_spline3(float exposure, const float *points, int npoints, uint64_t *n_bad, uint32_t *first_bad_bits);
/* output8bit (src/color_conversions.rs:323-326) on every f32 against a cheaper form: variant 0 v_cvt_pk_u8_f32(v*256), 1 the same of
 * floor(v*256), 2 min(saturating v_cvt_u32_f32(v*256), 255) -- the form the kernels use must report 0 mismatches */
IPK_API int ipk_selftest_quant8(int variant, uint64_t *n_bad, uint32_t *first_bad_bits);
/* OpGamma's step followed by output8bit (src/ops/gamma.rs:22, src/color_conversions.rs:323-326) on every f32, against the ONE-lookup form the 8-bit
 * output kernels run: clamp, then k_i + (c >= t_i) from a table of 8192 {k, threshold} steps built on the device from the gamma table (inside one of its
 * segments the quantised value changes at most once, and every operation on the way is monotone). */
IPK_API int ipk_selftest_q8(uint64_t *n_bad, uint32_t *first_bad_bits);
/* output16bit = (v*65535.0).round().max(0.0).min(65535.0) as u16 (src/color_conversions.rs:327-330; f32::round rounds halves away from zero) against the
 * form the kernels pack their 16-bit output with -- floor(v*65535 + 0.5) through the saturating v_cvt_u32_f32 and v_cvt_pk_u16_u32 -- on every f32. */
IPK_API int ipk_selftest_quant16(uint64_t *n_bad, uint32_t *first_bad_bits);
/* the device cbrtf routines (variant 0 literal glibc port, 1 select form, 2 form for 1<x<2) on a device array;
 * callers compare with the host libm's cbrtf (src/color_conversions.rs:123 -> f32::cbrt) */
IPK_API int ipk_selftest_cbrtf(const float *in, float *out, size_t n, int variant, void *stream);
/* test hook of the row-walking kernels' launch schedule: enabled = 0 makes every following launch run as if its stream could get no task-queue slot
 * (the queue-less static schedule: every wave walks the tasks of its index, a whole round of waves apart); 1 restores the queues.  The results
 * are the same bits either way -- which is what tests/test_gpu_fused.py checks for launches that hold more tasks than the chip has waves. */
IPK_API int ipk_selftest_task_queue(int enabled);
/* host-side test hooks of the caching contract: LRU bookkeeping without device memory; SHA-256 known answers */
IPK_API int ipk_selftest_cache_put(ipk_cache *cache, const uint8_t *key32, size_t bytes);
IPK_API int ipk_selftest_sha256(const void *data, size_t n, uint8_t *out32);

#ifdef __cplusplus
}
#endif
#endif /* IMAGEPIPE_AMD_H */
