#!/usr/bin/env python3
"""tools/clouds_census.py — wave-level census of the APP_CLOUDS 4K frame.

Step 1 (here, no GPU):   python tools/clouds_census.py --build     -> build/libsbx_stats.so (-DSBX_CL_STATS)
Step 2 (on the GPU box): PYTHONPATH=. python tools/clouds_census.py

The census build makes k_clouds write its counters instead of colours: lane 0 of every wave writes
(main steps, lit steps, sum of alive lanes, sum of lit lanes), lane 1 (hc_slow calls, insert passes, cells
inserted, light-sample re-lookups), lane 2 (main samples past the first stage, past the second, Lipschitz-skipped steps).  Not a product path."""
import os, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "shaderbox_amd", "lib", "libsbx.so")
STATS = os.path.join(ROOT, "build", "libsbx_stats.so")

if "--build" in sys.argv:
    import shaderbox_amd.build as B
    keep = os.path.join(ROOT, "build", "libsbx_keep.so")
    os.makedirs(os.path.dirname(STATS), exist_ok=True)
    shutil.copy(LIB, keep)
    B.FLAGS.insert(0, "-DSBX_CL_STATS")
    B.build(force=True)
    shutil.copy(LIB, STATS)
    B.FLAGS.pop(0)
    B.build(force=True)
    print("built", STATS)
    sys.exit(0)

import ctypes
import numpy as np
import torch
import shaderbox_amd as sa

shutil.copy(STATS, LIB)          # on the (throw-away) GPU box copy only
r = sa.Renderer()
W, H = 3840, 2160
a = r.render("clouds", W, H, 0.37).cpu().numpy().reshape(H, W, 4)
w0, w1, w2 = a[::2, 0::32], a[::2, 1::32], a[::2, 2::32]          # 32x2 tiles: lanes 0, 1 and 2 of every wave
m = w0[..., 0] > 0
print("waves %d, marching %.1f %%" % (m.size, 100 * m.mean()))
print("per marching wave: main steps %.1f, lit steps %.1f, light samples %.1f" %
      (w0[..., 0][m].mean(), w0[..., 1][m].mean(), 6 * w0[..., 1][m].mean()))
print("alive lanes per main step %.1f, lit lanes per lit step %.1f" %
      (w0[..., 2].sum() / w0[..., 0].sum(), w0[..., 3].sum() / w0[..., 1].sum()))
print("per marching wave: hc_slow calls %.1f, insert passes %.1f, cells inserted %.1f, light re-lookups %.1f" %
      (w1[..., 0][m].mean(), w1[..., 1][m].mean(), w1[..., 2][m].mean(), w1[..., 3][m].mean()))
print("per marching wave: main samples past the first stage %.1f, past the second %.1f, steps skipped by the Lipschitz bound %.1f" %
      (w2[..., 0][m].mean(), w2[..., 1][m].mean(), w2[..., 2][m].mean()))
