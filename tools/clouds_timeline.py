#!/usr/bin/env python3
"""tools/clouds_timeline.py — occupancy timeline of ONE k_clouds launch (3840x2160): every wave writes its start and duration
(build: python tools/ab_build.py cltimes:kern_clouds.hip:-DSBX_CL_TIMES); how long the chip is full, how long the ramp and the tail are,
which rows the last waves belong to.  Run on the GPU box: python tools/clouds_timeline.py [libname]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import shaderbox_amd as sa

args = [a for a in sys.argv[1:] if not a.startswith("--")]
name = args[0] if args else "cltimes"
strip = [a for a in sys.argv[1:] if a.startswith("--rank=")]      # --rank=R/N: the launch of rank R of N (cyclic 8-row blocks, in place)
sa.LIB_PATH = os.path.join(ROOT, "build", "ab", "libsbx_%s.so" % name)
W, H, TW, TH = 3840, 2160, 32, 2
r = sa.Renderer()
if strip:
    from shaderbox_amd import shard
    rank, world = (int(v) for v in strip[0].split("=")[1].split("/"))
    frame = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    for k in range(31):
        r.render_rank_in_place("clouds", W, H, 0.37, 8, rank, world, frame)
        if k % 4 == 3:
            torch.cuda.synchronize()                  # (a dispatch table is adopted when the host sees it complete)
    torch.cuda.synchronize()
    rows = np.asarray(shard.rank_row_indices(H, 8, rank, world))
    a = frame.cpu().numpy().view(np.uint32).reshape(H, W, 4)[rows]
    name += " rank %d of %d (%d rows)" % (rank, world, len(rows))
else:
    for k in range(30):
        a = r.render("clouds", W, H, 0.37)
        if k % 4 == 3:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    a = r.render("clouds", W, H, 0.37).cpu().numpy().view(np.uint32).reshape(H, W, 4)
w0 = a[::TH, ::TW]
t0 = w0[..., 0].astype(np.int64)
dur = w0[..., 1].astype(np.int64)
mar = w0[..., 2].astype(np.int64)
t0 = (t0 - t0.min()) & 0xffffffff
t1 = t0 + dur
span = t1.max()
tick = 0.01
print("# %s" % name)
print("k_clouds %dx%d: %d waves, launch span %.1f us; wave duration us: mean %.1f p50 %.1f p99 %.1f max %.1f; marching waves %.1f %%"
      % (W, H, t0.size, span * tick, dur.mean() * tick, np.percentile(dur, 50) * tick, np.percentile(dur, 99) * tick, dur.max() * tick, 100 * mar.mean()))
print("sum of wave time %.0f us = %.0f slots busy on average of %d" % (dur.sum() * tick, dur.sum() / span, 1024 * 6))
edges = np.linspace(0, span, 41)
print("time bin (us)      resident waves   started   (rows of the waves started: min-max)")
for i in range(40):
    lo, hi = edges[i], edges[i + 1]
    res = np.clip(np.minimum(t1, hi) - np.maximum(t0, lo), 0, None).sum() / (hi - lo)
    st = (t0 >= lo) & (t0 < hi)
    rows = np.nonzero(st.any(axis=1))[0]
    print("%7.1f - %7.1f   %8.0f   %7d   %s" % (lo * tick, hi * tick, res, st.sum(), ("%d-%d" % (rows.min() * TH, rows.max() * TH + 1)) if len(rows) else "-"))
last = np.argsort(t1.ravel())[::-1][:300]
print("the 300 last-finishing waves: rows min %d median %d max %d; mean duration %.1f us, mean start %.1f us"
      % ((last // w0.shape[1]).min() * TH, int(np.median(last // w0.shape[1])) * TH, (last // w0.shape[1]).max() * TH,
         dur.ravel()[last].mean() * tick, t0.ravel()[last].mean() * tick))
print("rows band: mean / max wave duration us")
for y in range(0, w0.shape[0], w0.shape[0] // 18):
    d = dur[y:y + w0.shape[0] // 18]
    print("  rows %4d-%4d: %7.1f / %7.1f" % (y * TH, (y + w0.shape[0] // 18) * TH - 1, d.mean() * tick, d.max() * tick))
