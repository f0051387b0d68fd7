// ipk_launch.hpp -- host-callable launchers of the gfx950 kernels (defined in ipk_kernels.hip).
// Plain C++ declarations so the C-ABI layer (ipk_api.cpp) never sees device code.
#pragma once
#include <hip/hip_runtime_api.h>
#include <cstddef>
#include <cstdint>
#include "ipk_host.hpp"

namespace ipk {

typedef Spline SplineHost;

// the per-stream task-queue heads of the row-walking kernels: one device block per library context, allocated on the device that is current when
// create_task_queues() is called (ipk_ctx_create) and released by destroy_task_queues() (ipk_ctx_destroy); null = every launch runs its static schedule
struct TaskQueues;
TaskQueues *create_task_queues();
void destroy_task_queues(TaskQueues *q);

template <typename T>
void launch_gofloat_cfa(const T *src, size_t owidth, size_t x, size_t y, size_t w, size_t h, float black0, float white0,
                        float *dst, hipStream_t s);
template <typename T>
void launch_gofloat_mono(const T *src, size_t owidth, size_t x, size_t y, size_t w, size_t h, float black0, float white0,
                         float *dst4, hipStream_t s);
template <typename T>
void launch_gofloat_rgb(const T *src, size_t owidth, size_t x, size_t y, size_t w, size_t h, const float *black4, const float *white4,
                        float *dst4, hipStream_t s);
void launch_gofloat_other_u8(const uint8_t *src, size_t owidth, size_t x, size_t y, size_t w, size_t h,
                             const void *gamma_reverse_pairs, float *dst4, hipStream_t s);
void launch_gofloat_other_u16(const uint16_t *src, size_t owidth, size_t x, size_t y, size_t w, size_t h, float *dst4, hipStream_t s);

void launch_demosaic_full(const float *src, size_t width, size_t img_height, size_t src_row0, size_t out_row0, size_t out_rows,
                          const uint32_t *lookups_dev, float *dst4, hipStream_t s);

// returns 0, or -4 when the launch could not be enqueued (hipGetLastError after the launch)
int launch_demosaic_bayer(const float *src, size_t width, size_t img_height, size_t src_row0, size_t out_row0, size_t out_rows,
                          int xoff, int yoff, const float *gen_cells, int gen_pw, int gen_ph, float *dst4, int num_cus, TaskQueues *queues, hipStream_t s);
void selftest_task_queue(bool enabled);   // test hook: false = every following launch runs the queue-less static schedule

template <typename T>
void launch_transform_buffer(const T *src, size_t width, size_t height, int64_t tlx, int64_t tly, int64_t trx, int64_t try_,
                             int64_t blx, int64_t bly, size_t nwidth, size_t nheight, size_t components,
                             const uint8_t *cfa48_dev, T *dst, hipStream_t s);

// run_other + scale_down_opbuf in one pass over an RGB8 / RGB16 raster (gofloat.rs:171-201 + scaling.rs:147-160)
void launch_raster_scale_down(const void *src, int src_is_u16, size_t owidth, size_t x, size_t y, size_t width, size_t height,
                              size_t nwidth, size_t nheight, const void *gamma_reverse_pairs, float *dst4, hipStream_t s);
template <typename T>
void launch_raw_scaled_demosaic(const T *src, size_t owidth, size_t x, size_t y, size_t width, size_t height, float black0, float white0, int norm_fast, int has_fourth_colour,
                                size_t nwidth, size_t nheight, const uint8_t *cfa48_dev, int pattern_width, int pattern_height, float *dst4, hipStream_t s,
                                size_t band_src_row0 = 0, size_t band_out_row0 = 0, size_t band_out_rows = 0);   // band_out_rows == 0: the whole frame

void launch_tolab(const float *src4, size_t npix, const float *mul4, const float *cm12, const void *lab_pairs, float *dst3,
                  int num_cus, hipStream_t s);
void launch_basecurve(const float *src3, size_t npix, const SplineHost &sp, float *dst3, int num_cus, hipStream_t s);
void launch_fromlab(const float *src3, size_t npix, const float *m9, float *dst3, int num_cus, hipStream_t s);
void launch_gamma(const float *src, size_t n, const void *gam_pairs, float *dst, int num_cus, hipStream_t s);
template <typename T>   // float (OpBuffer) or uint8_t / uint16_t (the quantised image: same permutation, fewer bytes)
void launch_rotate(const T *src3, size_t owidth, size_t oheight, int64_t base_offset_px, int64_t x_step_px, int64_t y_step_px,
                   T *dst3, hipStream_t s);
void launch_output8(const float *src, size_t n, uint8_t *dst, int num_cus, hipStream_t s);
void launch_output16(const float *src, size_t n, uint16_t *dst, int num_cus, hipStream_t s);
// DynamicImage::to_rgb16 / to_rgb8 channel conversion of the raster fast path (image 0.24: c*257, (c+128)/257)
void launch_chan_8_to_16(const uint8_t *src, size_t n, uint16_t *dst, int num_cus, hipStream_t s);
void launch_chan_16_to_8(const uint16_t *src, size_t n, uint8_t *dst, int num_cus, hipStream_t s);

struct FusedLaunch {
  const void *src; void *dst;
  bool src_is_u16, src_aligned4;
  size_t width, height, owidth;
  size_t row_off, out_r0, out_r1;
  float black0, white0;
  int exact_norm;                // 1: gofloat's division must be a true division (see validate_cdiv)
  int fast_ok;                   // 1: matrix / multipliers / curve are finite and ordinary (fast point-wise form allowed)
  int xoff, yoff;
  const float *mul4, *cm12, *rgbm9;
  int has_curve, linear;
  const SplineHost *spline;
  int out_type;                  // 0 f32, 1 u8, 2 u16
  const void *lab_table, *gam_table;   // plain 8193-float tables (XYZ_LAB_TRANSFORM, SRGB gamma)
  const void *lab_pairs, *gam_pairs;   // the same two as 8192 x {v[i], v[i+1] - v[i]} (built at ipk_init): the LDS image of the pair form, copied without registers
  const void *gam_q8;                  // 8192 x {k, threshold}: OpGamma + output8bit as one step lookup (launch_build_q8), the 8-bit variants' LDS image
  int px_guard;                  // 0: u16 source whose levels and parameters the host found ordinary (kernel variant without per-pixel input guards)
  const float *gen_cells; int gen_pw, gen_ph, gen_check;   // generic-CFA mode (device cell records) or null: RGGB phase (xoff, yoff)
  int num_cus;
  int schedule;                  // ipk_schedule (launch_fused_bayer, single frames on the static schedule only)
  TaskQueues *queues;            // the launching context's task queues (launch_fused_bayer only; may be null)
  int ori;                       // 0, or the ipk_orientation (Rotate90 / Rotate270) in whose rotated space the launch works: src is the permuted mosaic
  int roles[4];                  // ori != 0: demosaic role of the rotated-space pixel with parities (row & 1, col & 1), index 2 * row parity + col parity
  // launch_fused_bayer only: batch_n > 0 = that many frames of this shape and these parameters (host arrays of device pointers, src already
  // offset like `src`); one persistent launch per 64 frames where a batch variant of the kernel exists, one launch per frame otherwise
  int batch_n; const void *const *batch_src; void *const *batch_dst;
};
// returns 0, -2 when f.ori != 0 and the parameters have no rotated-space variant (nothing is launched), -4 when a launch could not be enqueued
int launch_fused_bayer(const FusedLaunch &f, hipStream_t s);
// rotate_buffer's permutation on a 1-channel image through an arbitrary source pitch / window (steps in source elements)
template <typename T>
void launch_rotate1(const T *src, size_t owidth, size_t oheight, int64_t base_offset, int64_t x_step, int64_t y_step, T *dst, hipStream_t s);
// OpToLab..OpGamma in one pass over a 4-channel f32 buffer (src/dst, mul4, cm12, rgbm9, curve, linear, tables, fast_ok are read)
int launch_pointwise_chain(const FusedLaunch &f, size_t npix, hipStream_t s);
int launch_tolab_fast(const FusedLaunch &f, size_t npix, hipStream_t s);   // OpToLab alone on the chain's fast form
// OpToLab..OpGamma + output8bit / output16bit in one pass over a 4-channel f32 buffer (f.out_type 1 / 2, f.gam_q8 set; npix >= 256): -1 otherwise
int launch_chain_quantised(const FusedLaunch &f, size_t npix, hipStream_t s);
// run_other + OpToLab..OpGamma + quantisation in one pass over an RGB8 / RGB16 raster (npix >= 256; f.out_type selects the output)
int launch_raster_chain(const FusedLaunch &f, size_t npix, int src_is_u16, const void *gamma_reverse_pairs, hipStream_t s);

// exhaustive on-device checks of the arithmetic shortcuts (see ipk_kernels.hip "Self-test kernels")
int launch_selftest_cdiv(float c, int variant, unsigned lo_bits, unsigned hi_bits, int include_special, void *out_dev, hipStream_t s);
void launch_copy_probe(const void *src, void *dst, size_t bytes, int num_cus, hipStream_t s);
void launch_mix_probe(const void *src, void *dst, size_t src_bytes, hipStream_t s);
void launch_clock_probe(void *out2_dev, unsigned long long spin_ticks, hipStream_t s);   // out2 = {shader-clock cycles, 100 MHz reference ticks} over the spin   // 1 : 3 read : write, dst holds 3 * src_bytes
int launch_selftest_spline3(const SplineHost &h, void *out_dev, hipStream_t s);
int launch_selftest_fract(void *out_dev, hipStream_t s);
int launch_selftest_clamp(void *out_dev, hipStream_t s);
int launch_selftest_quant8(void *out_dev, int variant, hipStream_t s);
int launch_selftest_quant16(void *out_dev, hipStream_t s);
void launch_build_q8(const void *gam_pairs, void *q8_out, hipStream_t s);        // q8_out: 8192 x 8 bytes of device memory
int launch_selftest_q8(const void *gam_pairs, const void *q8, void *out_dev, hipStream_t s);
int launch_selftest_cbrt(const float *in, float *out, size_t n, int variant, hipStream_t s);

}  // namespace ipk
