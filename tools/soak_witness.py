#!/usr/bin/env python3
"""Run on the GPU box: the recorded-domain kernels (csrc/sbx_witness.h: EGG, SDF_AO, VINYL, VINYL_GPU, RAYTRACER) against the SAME
kernels with the IEEE forms only (sbx_set_variant 3) on N random (u_time, u_mouse) frames at 3840x2160 and at an odd size whose
centre column has fragCoord.x == u_res.x / 2 (a zero component in the primary direction: the record fires there) — every pixel,
bit for bit.     python tools/soak_witness.py [frames per app = 200] [app,app,...]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import shaderbox_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(2024)
R = shaderbox_amd.Renderer(0)
total_bad = 0
for app in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("egg", "raytracer", "sdf_ao", "vinyl", "vinyl_gpu")):
    bad = 0
    pixels = 0
    for i in range(n):
        W, H = ((3840, 2160), (1921, 1081), (2560, 1440))[i % 3]
        t = float(rng.uniform(0, 120)) if i % 4 else float(rng.uniform(0, 3))
        mouse = (float(rng.uniform(0, W)), float(rng.uniform(0, H))) if i % 2 else (0.0, 0.0)
        R.set_variant(0)
        a = R.render(app, W, H, t, mouse=mouse).clone()
        R.set_variant(3)
        b = R.render(app, W, H, t, mouse=mouse)
        same = (a.view(torch.int32) == b.view(torch.int32)) | (torch.isnan(a) & torch.isnan(b))
        pixels += W * H
        if not bool(same.all()):
            bad += 1
            print("MISMATCH %s: %d pixels; %dx%d t=%r mouse=%r" % (app, int((~same).any(-1).sum()), W, H, t, mouse))
    R.set_variant(0)
    print("soak %-10s %d frames (%.0f M pixels) against the IEEE-form kernel: %d frames with a differing pixel" % (app, n, pixels / 1e6, bad))
    total_bad += bad
print("witness soak: %d frames with differing pixels" % total_bad)
sys.exit(1 if total_bad else 0)
