#!/usr/bin/env python3
"""One-off soak (run on the GPU box, PYTHONPATH=.): EGG / SDF_AO / VINYL / PLANET default kernels (exact culling, skips)
against the plain ones (sbx_set_variant 1) on N random frames each.  tests/test_gpu_parity.py runs a short version."""
import sys, torch, numpy as np
sys.path.insert(0, ".")
import shaderbox_amd
R = shaderbox_amd.Renderer(0)
rng = np.random.default_rng(7)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for app in ("egg", "sdf_ao", "vinyl", "planet"):
    bad = 0
    for i in range(N):
        t = float(rng.uniform(0, 60)) if i % 3 else float(rng.uniform(0, 3))
        mouse = (float(rng.uniform(0, 640)), float(rng.uniform(0, 360))) if i % 2 else (0.0, 0.0)
        W, H = (640, 360) if i % 4 else (333, 187)
        if app == "planet": W, H = (320, 180) if i % 4 else (201, 113)
        R.set_variant(0); a = R.render(app, W, H, t, mouse=mouse).clone()
        R.set_variant(1); b = R.render(app, W, H, t, mouse=mouse)
        same = (a.view(torch.int32) == b.view(torch.int32)) | (torch.isnan(a) & torch.isnan(b))
        if not bool(same.all()):
            bad += 1; print("MISMATCH", app, i, t, mouse, int((~same).sum()))
    print(app, "frames", N, "mismatching", bad)
R.set_variant(0)
