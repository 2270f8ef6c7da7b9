#!/usr/bin/env python3
"""tools/egg_census.py — per-wave census of k_egg: where a launch's time goes.

Step 1 (here, no GPU):   python tools/egg_census.py --build     -> build/ab/libsbx_eggstats.so (-DSBX_EGG_STATS)
Step 2 (on the GPU box): python tools/egg_census.py [--plain] [W H]

The census build makes every lane of k_egg write (start time, duration in 10 ns ticks of s_memrealtime, longest trace of the wave |
lanes with a shadow march << 8 | XCC << 16, HW_ID) instead of its colour.  From lane 0 of every wave: the histogram of wave
durations, the number of resident waves over time (the launch's occupancy timeline), the work in the last part of the launch.
Not a product path."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if "--build" in sys.argv:
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "ab_build.py"), "eggstats:kern_egg.hip:-DSBX_EGG_STATS",
                           "eggstats_plain:kern_egg.hip:-DSBX_EGG_STATS,-DEGG_HOT_FIRST=0"])
    sys.exit(0)

import numpy as np
import torch
import shaderbox_amd as sa

_lib = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--lib=")]          # --lib=name: build/ab/libsbx_<name>.so (a census build)
sa.LIB_PATH = os.path.join(ROOT, "build", "ab", "libsbx_%s.so" % (_lib[0] if _lib else "eggstats" + ("_plain" if "--plain" in sys.argv else "")))
print("# %s" % ("rows dealt bottom to top (EGG_HOT_FIRST=0)" if "--plain" in sys.argv else "hot-first dispatch (the shipped order)"))
args = [a for a in sys.argv[1:] if not a.startswith("--")]
W, H = (int(args[0]), int(args[1])) if len(args) >= 2 else (1920, 1080)
TW, TH = 16, 4
r = sa.Renderer()
for k in range(20):
    a = r.render("egg", W, H, 0.37)
    if k % 2 == 1:
        torch.cuda.synchronize()                     # (the dispatch order's table is adopted when the host sees it complete)
torch.cuda.synchronize()
if r.tile_order("egg")[0] > 0:
    print("# (this launch runs under the dispatch table: SBX_TILE_ORDER=0 in the environment for the kernel's own order)")
a = r.render("egg", W, H, 0.37).cpu().numpy().view(np.uint32).reshape(H, W, 4)
w0 = a[::TH, ::TW]                                   # lane 0 of every wave
t0 = w0[..., 0].astype(np.int64)
dur = w0[..., 1].astype(np.int64)
base = t0.min()
t0 = (t0 - base) & 0xffffffff
t1 = t0 + dur
steps = (w0[..., 2] & 0xff).astype(np.int64)
nsh = ((w0[..., 2] >> 8) & 0xff).astype(np.int64)
xcc = ((w0[..., 2] >> 16) & 0xf).astype(np.int64)
span = t1.max()
tick = 0.01                                           # us per tick (100 MHz)
print("k_egg %dx%d: %d waves, launch span (first wave start to last wave end) %.1f us" % (W, H, t0.size, span * tick))
print("wave duration us: mean %.2f  p50 %.2f  p90 %.2f  p99 %.2f  max %.2f ; sum %.0f us = %.2f slots busy on average of %d"
      % (dur.mean() * tick, np.percentile(dur, 50) * tick, np.percentile(dur, 90) * tick, np.percentile(dur, 99) * tick, dur.max() * tick,
         dur.sum() * tick, dur.sum() / span, 1024 * 7))
print("longest trace of a wave: mean %.1f, waves with >= 40 steps %d, with 80 steps %d; waves with a shadow march %d (mean %.1f lanes)"
      % (steps.mean(), (steps >= 40).sum(), (steps >= 80).sum(), (nsh > 0).sum(), nsh[nsh > 0].mean() if (nsh > 0).any() else 0))
# occupancy timeline in 20 bins
edges = np.linspace(0, span, 21)
print("time bin (us)      resident waves (avg)   waves started   heavy waves (>=40 steps or shadow) started")
for i in range(20):
    lo, hi = edges[i], edges[i + 1]
    overlap = np.clip(np.minimum(t1, hi) - np.maximum(t0, lo), 0, None).sum() / (hi - lo)
    started = ((t0 >= lo) & (t0 < hi))
    heavy = started & ((steps >= 40) | (nsh > 0))
    print("%7.1f - %7.1f   %10.0f   %10d   %10d" % (lo * tick, hi * tick, overlap, started.sum(), heavy.sum()))
# which rows do the last-finishing waves belong to
order = np.argsort(t1.ravel())[::-1][:400]
rows = (order // w0.shape[1]) * TH
print("the 400 last-finishing waves: tile rows (y) min %d median %d max %d; their mean duration %.1f us, mean start %.1f us"
      % (rows.min(), int(np.median(rows)), rows.max(), dur.ravel()[order].mean() * tick, t0.ravel()[order].mean() * tick))
# per 40-row band of the image: mean wave duration
print("band of rows: mean / max wave duration us, share of all wave time")
for y in range(0, w0.shape[0], max(1, w0.shape[0] // 18)):
    d = dur[y:y + max(1, w0.shape[0] // 18)]
    print("  rows %4d-%4d: %6.2f / %6.2f   %4.1f %%" % (y * TH, (y + max(1, w0.shape[0] // 18)) * TH - 1, d.mean() * tick, d.max() * tick,
                                                     100.0 * d.sum() / dur.sum()))
print("per XCC wave time share:", [round(float(dur[xcc == k].sum()) / dur.sum(), 3) for k in range(8)])
# coarse 2-D map of the MAX wave duration (us): 30 columns x 27 rows of cells (4 x 10 waves each)
gy, gx = dur.shape
cy, cx = 10, 4
print("max wave duration per cell of %d x %d waves (us); top of the image first" % (cx, cy))
for y in range(gy - cy, -1, -cy):
    print(" ".join("%3.0f" % (dur[y:y + cy, x:x + cx].max() * tick) for x in range(0, gx, cx)))
print("mean trace steps per cell")
for y in range(gy - cy, -1, -cy):
    print(" ".join("%3.0f" % (steps[y:y + cy, x:x + cx].mean()) for x in range(0, gx, cx)))
