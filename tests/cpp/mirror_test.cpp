// Exercises include/imagepipe_amd.hpp (the C++ mirror of Pipeline / ImageOp / OpBuffer) on one raw frame:
//   mirror_test <in.u16> <width> <height> <cfa> <maxwidth> <rotation> <out.f32>
// Runs Pipeline::run (C driver, fused when legal) and the op-by-op loop, requires them to agree bit for bit,
// and writes the result for the Python test to compare with the oracle.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <tuple>
#include <vector>
#include "imagepipe_amd.hpp"

int main(int argc, char **argv) {
  if (argc != 8) { std::fprintf(stderr, "usage\n"); return 2; }
  const size_t w = std::atol(argv[2]), h = std::atol(argv[3]);
  std::vector<uint16_t> raw(w * h);
  FILE *f = std::fopen(argv[1], "rb");
  if (!f || std::fread(raw.data(), 2, raw.size(), f) != raw.size()) { std::fprintf(stderr, "bad input\n"); return 2; }
  std::fclose(f);
  try {
    if (ipk_init(0) != 0) { std::fprintf(stderr, "%s\n", ipk_last_error()); return 3; }
    imagepipe::ImageSource img;
    img.kind = imagepipe::ImageSource::Raw; img.width = w; img.height = h; img.cfa = argv[4];
    for (int i = 0; i < 4; ++i) { img.blacklevels[i] = 512.0f; img.whitelevels[i] = 16383.0f; }
    const float wb[4] = {2.0f, 1.0f, 1.5f, NAN};
    std::copy(wb, wb + 4, img.wb_coeffs);
    const float scale[3] = {1.10f, 1.05f, 1.20f};
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) img.cam_to_xyz_normalized[r * 4 + c] *= scale[r];
    img.data = imagepipe::DeviceArray(raw.data(), raw.size() * 2);
    auto pipe = imagepipe::Pipeline::new_from_source(std::move(img));
    pipe.globals.settings.maxwidth = std::atol(argv[5]);
    pipe.ops.transform.rotation = std::atoi(argv[6]);
    auto a = pipe.run();
    auto b = pipe.run_ops();
    const std::vector<float> va = a->to_host(), vb = b->to_host();
    if (a->width != b->width || a->height != b->height || va.size() != vb.size() || std::memcmp(va.data(), vb.data(), va.size() * 4) != 0) {
      std::fprintf(stderr, "driver and op loop disagree\n"); return 4;
    }
    auto o8 = pipe.output_8bit();
    const bool fused_flag = pipe.last_used_fused;
    auto same = [](const std::vector<float> &x, const std::vector<float> &y) { return x.size() == y.size() && std::memcmp(x.data(), y.data(), x.size() * 4) == 0; };
    {  // Pipeline::run(Some(cache)): cold run, hit, edit of one op
      auto cache = imagepipe::Pipeline::new_cache(size_t(1) << 30);
      auto c1 = pipe.run(&cache); const int cold = pipe.last_ops_run;
      auto c2 = pipe.run(&cache); const int warm = pipe.last_ops_run;
      if (cold != 0xFF || warm != 0 || !same(c1->to_host(), va) || !same(c2->to_host(), va) || !cache.contains(pipe.hashes()[7]) || cache.bytes() == 0) {
        std::fprintf(stderr, "cached run disagrees (cold %x warm %x)\n", cold, warm); return 6;
      }
      pipe.ops.transform.fliph = !pipe.ops.transform.fliph;
      auto c3 = pipe.run(&cache); const int edited = pipe.last_ops_run;
      pipe.allow_fused = false;
      auto ref = pipe.run();
      pipe.allow_fused = true; pipe.ops.transform.fliph = !pipe.ops.transform.fliph;
      if (edited == 0 || !same(c3->to_host(), ref->to_host())) { std::fprintf(stderr, "cached run after edit disagrees (%x)\n", edited); return 7; }
    }
    {  // the same frame as a "shoot" of three, dealt over a device set of two contexts (both on device 0 here), host buffers in and out;
       // and one run under the second member as the current context -- all equal to the single-context results above
      imagepipe::DeviceSet set({0, 0});
      if (set.size() != 2) { std::fprintf(stderr, "device set size %zu\n", set.size()); return 8; }
      std::vector<const void *> ins; std::vector<void *> outs;
      for (int i = 0; i < 3; ++i) {
        void *in = ipk_host_alloc(raw.size() * 2), *out = ipk_host_alloc(o8.data.size());
        if (!in || !out) { std::fprintf(stderr, "ipk_host_alloc\n"); return 8; }
        std::memcpy(in, raw.data(), raw.size() * 2); ins.push_back(in); outs.push_back(out);
      }
      imagepipe::DeviceSet::develop_host(pipe, ins, outs, IPK_OUT_U8);
      for (int i = 0; i < 3; ++i) if (std::memcmp(outs[i], o8.data.data(), o8.data.size()) != 0) { std::fprintf(stderr, "device-set frame %d differs\n", i); return 8; }
      for (int i = 0; i < 3; ++i) { ipk_host_free(const_cast<void *>(ins[i])); ipk_host_free(outs[i]); }
      { imagepipe::CurrentContext cur(set.member(1));
        auto a2 = pipe.run();
        if (!same(a2->to_host(), va)) { std::fprintf(stderr, "run under the second context differs\n"); return 8; } }
      if (ipk_ctx_current() == set.member(1)) { std::fprintf(stderr, "the scope guard did not restore the context\n"); return 8; }
    }
    std::printf("%zu %zu %d %zu\n", a->width, a->height, fused_flag ? 1 : 0, o8.data.size());
    FILE *o = std::fopen(argv[7], "wb");
    std::fwrite(va.data(), 4, va.size(), o); std::fclose(o);
  } catch (const std::exception &e) { std::fprintf(stderr, "error: %s\n", e.what()); return 5; }
  return 0;
}
