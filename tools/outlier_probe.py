#!/usr/bin/env python
"""Development tool: which launches of the headline kernel are the slow ones?  N back-to-back launches of the 100 MP frame, an event pair around each; prints the
distribution, the positions of the launches above 1.15x the median, the time between them, and the shader clock rocm-smi reports before / after."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import imagepipe_amd as ipa, util
ipa.init(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
W = H = 10000
g = torch.Generator(device="cuda"); g.manual_seed(5)
src = torch.randint(0, 16384, (H * W,), generator=g, device="cuda", dtype=torch.int32).to(torch.float32)
plan = ipa.FusedPlan(width=W, height=H, is_float=True, black0=util.BLACK, white0=util.WHITE, cfa="RGGB", wb_coeffs=util.WB, cam_to_xyz_normalized=util.cam_matrix())
out = plan.new_output()
st = torch.cuda.current_stream().cuda_stream
for _ in range(600): plan.run(src, out, st)
torch.cuda.synchronize()
evs = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
evs[0].record()
for i in range(N):
    plan.run(src, out, st); evs[i + 1].record()
torch.cuda.synchronize()
per = [evs[i].elapsed_time(evs[i + 1]) for i in range(N)]
t_at = [evs[0].elapsed_time(evs[i]) for i in range(N)]
s = sorted(per); med = s[N // 2]
print("launches %d  min %.4f  median %.4f  p95 %.4f  p99 %.4f  max %.4f  stddev %.4f ms" % (N, s[0], med, s[int(.95 * N)], s[int(.99 * N)], s[-1], (sum((x - sum(per) / N) ** 2 for x in per) / N) ** .5))
slow = [i for i in range(N) if per[i] > 1.15 * med]
print("above 1.15x median: %d launches" % len(slow))
gaps = [round(t_at[b] - t_at[a], 1) for a, b in zip(slow, slow[1:])]
print("ms between consecutive slow launches:", gaps[:60])
print("slow launch durations:", [round(per[i], 3) for i in slow[:60]])
# runs of consecutive slow launches
runs, cur = [], 1
for a, b in zip(slow, slow[1:]):
    if b == a + 1: cur += 1
    else: runs.append(cur); cur = 1
if slow: runs.append(cur)
print("run lengths of consecutive slow launches:", runs[:60])
# windowed mean to see slow drifts
win = 100
print("mean per window of %d launches:" % win, [round(sum(per[i:i + win]) / win, 4) for i in range(0, N - win + 1, win)])
