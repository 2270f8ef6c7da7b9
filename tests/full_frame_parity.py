#!/usr/bin/env python3
"""One-off (run on the GPU box; minutes of host time on its 256 threads): EVERY pixel of every BASELINE.json frame — C2 EGG
1920x1080, C3 RAYTRACER 3840x2160, C4 CLOUDS 3840x2160, C5 ATMOSPHERE and PLANET 7680x4320, t = 0.37, mouse 0 — from the shipped
kernels against the CPU oracle, bit for bit (NaN == NaN).  The -m gpu tests compare evenly spread full rows of these frames;
this compares all of them.     python tests/full_frame_parity.py [app ...]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import shaderbox_amd
from oracle.oracle import APP_IDS, Oracle

CASES = [("egg", 1920, 1080), ("raytracer", 3840, 2160), ("clouds", 3840, 2160), ("atmosphere", 7680, 4320),
         ("planet", 7680, 4320), ("sdf_ao", 3840, 2160), ("vinyl", 3840, 2160), ("clouds_best", 3840, 2160),
         ("clouds_ue4", 3840, 2160), ("clouds_sky", 3840, 2160), ("vinyl_gpu", 3840, 2160),
         ("planet_atmosphere", 3840, 2160)]
want = set(sys.argv[1:])
R = shaderbox_amd.Renderer(0)
O = Oracle()
bad_total = 0
for app, W, H in CASES:
    if want and app not in want:
        continue
    gpu = R.render(app, W, H, 0.37).cpu().numpy()
    t0 = time.perf_counter()
    bad = 0
    worst = 0.0
    BAND = 270                                   # oracle rows per call (bounded host memory)
    for y0 in range(0, H, BAND):
        rows = list(range(y0, min(y0 + BAND, H)))
        ref = O.render_rows(APP_IDS[app], W, H, 0.37, rows)
        g = gpu[y0:y0 + len(rows)]
        both_nan = np.isnan(g) & np.isnan(ref)
        diff = (g.view(np.uint32) != ref.view(np.uint32)) & ~both_nan
        bad += int(diff.any(-1).sum())
        if diff.any():
            worst = max(worst, float(np.nanmax(np.abs(np.where(diff, g.astype(np.float64) - ref.astype(np.float64), 0.0)))))
    dt = time.perf_counter() - t0
    print("%-12s %5dx%-5d %9d pixels against the oracle (%.1f s of host time): %d differing, max |diff| %g"
          % (app, W, H, W * H, dt, bad, worst), flush=True)
    bad_total += bad
print("full-frame parity: %d differing pixels in total" % bad_total)
sys.exit(1 if bad_total else 0)
