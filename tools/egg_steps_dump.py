#!/usr/bin/env python3
"""tools/egg_steps_dump.py — per-PIXEL trace step counts of k_egg (census build, tools/egg_census.py --build), saved as a uint8 map
for offline models of workgroup shapes (tools/egg_group_model.py).  Run on the GPU box:  python tools/egg_steps_dump.py [W H]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import shaderbox_amd as sa

sa.LIB_PATH = os.path.join(ROOT, "build", "ab", "libsbx_eggstats.so")
args = [a for a in sys.argv[1:] if not a.startswith("--")]
W, H = (int(args[0]), int(args[1])) if len(args) >= 2 else (1920, 1080)
r = sa.Renderer()
a = r.render("egg", W, H, 0.37).cpu().numpy().view(np.uint32).reshape(H, W, 4)
steps = (a[..., 3] & 0xff).astype(np.uint8)
shadow = ((a[..., 3] >> 8) & 0xff).astype(np.uint8)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "egg_steps_%dx%d.npz" % (W, H)), steps=steps, shadow=shadow)
print("steps: mean %.2f max %d; pixels with >= 24 steps %d, >= 40 %d, == 80 %d; shadow pixels %d"
      % (steps.mean(), steps.max(), (steps >= 24).sum(), (steps >= 40).sum(), (steps >= 80).sum(), (shadow > 0).sum()))
