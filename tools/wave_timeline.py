#!/usr/bin/env python
"""Development tool: where the 4096 persistent waves of the fused kernel spend one launch (needs a -DIPK_DEV_PROBE build:
tools/build_variant.sh probe -DIPK_DEV_PROBE=1; IPK_SO_OVERRIDE=.../libprobe.so python tools/wave_timeline.py [noise|photo] [W H]).
Per wave the kernel records its start and end (shader cycles and the 100 MHz wall clock), the cycles spent in task draws, tasks and rows done."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import imagepipe_amd as ipa, util, bench

kind = sys.argv[1] if len(sys.argv) > 1 else "noise"
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (10000, 10000)
ipa.init(0)
src = bench.synth_frame(torch, H, W, kind, util.SEED + 2).to(torch.float32).reshape(-1).contiguous()
dst = torch.empty(H * W * 3, dtype=torch.float32, device="cuda")
plan = ipa.FusedPlan(width=W, height=H, is_float=True, black0=util.BLACK, white0=util.WHITE, cfa="RGGB", wb_coeffs=util.WB,
                     cam_to_xyz_normalized=util.cam_matrix(), out_type=ipa.OUT_F32)
st = torch.cuda.current_stream().cuda_stream
for _ in range(30):
    plan.run(src, dst, st)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); plan.run(src, dst, st); e1.record(); torch.cuda.synchronize()
L = ctypes.CDLL(os.environ["IPK_SO_OVERRIDE"])
buf = np.zeros(4096 * 8, dtype=np.uint64)
rc = L.ipk_dev_probe_read(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.size)); assert rc == 0, rc
atomics = (buf.reshape(4096, 8)[:, 2] >> np.uint64(48)).astype(np.float64)
buf.reshape(4096, 8)[:, 2] &= np.uint64((1 << 48) - 1)
q = buf.reshape(4096, 8).astype(np.float64)
live = q[:, 3] > 0                     # waves that ran at least one task
atomics = atomics[live]; q = q[live]
n_live = int(live.sum())
w0, w1 = q[:, 6], q[:, 7]
t_begin, t_end = w0.min(), w1.max()
span_us = (t_end - t_begin) / 100.0
life = (w1 - w0) / 100.0
cyc = q[:, 1] - q[:, 0]
print("%s %dx%d: event %.1f us; first wave start -> last wave end %.1f us" % (kind, W, H, e0.elapsed_time(e1) * 1e3, span_us))
print("  lifetime us percentiles 0/10/50/90/100: %.1f %.1f %.1f %.1f %.1f" % tuple(np.percentile(life, [0, 10, 50, 90, 100])))
print("  wave start offsets us: min %.1f median %.1f max %.1f" % ((w0 - t_begin).min() / 100, np.median(w0 - t_begin) / 100, (w0 - t_begin).max() / 100))
print("  wave end before launch end us: min %.1f p10 %.1f median %.1f p90 %.1f max %.1f" % tuple(np.percentile((t_end - w1) / 100, [0, 10, 50, 90, 100])))
print("  wave lifetime us: mean %.1f (%.3f of span); shader clock during life %.2f GHz" % (life.mean(), life.mean() / span_us, (cyc / (life * 1e3)).mean()))
print("  tasks per wave: min %d mean %.2f max %d; rows per wave: min %d mean %.1f max %d" % (q[:, 3].min(), q[:, 3].mean(), q[:, 3].max(), q[:, 5].min(), q[:, 5].mean(), q[:, 5].max()))
print("  queue atomics per wave: mean %.2f max %d (total %d)" % (atomics.mean(), atomics.max(), atomics.sum()))
print("  draw wait per wave us (at its own clock): mean %.2f max %.2f; share of lifetime %.4f" % ((q[:, 2] / cyc * life).mean(), (q[:, 2] / cyc * life).max(), (q[:, 2] / cyc).mean()))
pr = q[:, 4] / cyc * life
print("  task priming (4 row loads, 3 awaited) per wave us: mean %.2f = %.2f per task; share of lifetime %.4f" % (pr.mean(), (pr / np.maximum(q[:, 3], 1)).mean(), (q[:, 4] / cyc).mean()))
ends = np.sort((w1 - t_begin) / 100.0)
for f in (0.5, 0.75, 0.9, 0.95):
    t = span_us * f
    print("  at %.0f %% of the span (%.0f us): %d of %d waves still resident" % (f * 100, t, int((ends > t).sum()), n_live))
per_row = life / np.maximum(q[:, 5] + 2.5 * q[:, 3], 1)
print("  us per row-equivalent (rows + 2.5 per task): mean %.3f" % per_row.mean())
if hasattr(L, "ipk_dev_probe2_read"):
    b2 = np.zeros(4096 * 4, dtype=np.uint64)
    rc = L.ipk_dev_probe2_read(b2.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(b2.size)); assert rc == 0, rc
    q2 = b2.reshape(4096, 4).astype(np.float64)[live]
    print("  share of the wave's cycles in the wait for the next row's loads (and older stores): mean %.4f p10 %.4f p90 %.4f; cycles per row: %.0f" % (
        (q2[:, 0] / cyc).mean(), np.percentile(q2[:, 0] / cyc, 10), np.percentile(q2[:, 0] / cyc, 90), (q2[:, 0] / np.maximum(q[:, 5], 1)).mean()))
    print("  share from normalising the next row to the last store's issue (finish_row, LDS staging round trip, three stores): mean %.4f; cycles per row: %.0f" % (
        (q2[:, 1] / cyc).mean(), (q2[:, 1] / np.maximum(q[:, 5], 1)).mean()))
    print("  cycles per row overall: %.0f" % (cyc / np.maximum(q[:, 5], 1)).mean())
