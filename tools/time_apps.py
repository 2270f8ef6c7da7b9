#!/usr/bin/env python3
"""Kernel time (HIP events, median of 7) of every app at its BASELINE config size, on one GPU."""
import sys
import torch
sys.path.insert(0, ".")
import shaderbox_amd

R = shaderbox_amd.Renderer(0)
R.set_timing(True)
CASES = [("clouds", 3840, 2160), ("egg", 1920, 1080), ("egg", 3840, 2160), ("raytracer", 3840, 2160), ("atmosphere", 7680, 4320),
         ("planet", 7680, 4320), ("sdf_ao", 3840, 2160), ("vinyl", 3840, 2160),
         ("clouds_best", 3840, 2160), ("clouds_tex", 3840, 2160), ("clouds_ue4", 3840, 2160), ("clouds_sky", 3840, 2160),
         ("vinyl_gpu", 3840, 2160), ("planet_atmosphere", 7680, 4320)]
# APP_CLOUDS' USE_NOISE_TEX build samples two baked volumes: the 128^3 volume ddsvolgen bakes for the shape, a 64^3 one for the detail
R.set_noise_volumes(R.worley_volume(128), R.worley_volume(64))
for _ in range(30):                      # clocks up before the first case
    R.render("clouds", 3840, 2160, 0.37)
torch.cuda.synchronize()
for app, w, h in CASES:
    buf = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
    R.render(app, w, h, 0.37, out=buf); torch.cuda.synchronize()
    ms = []
    for _ in range(7):
        R.render(app, w, h, 0.37, out=buf); ms.append(R.last_kernel_ms())
    ms.sort()
    print("%-10s %5dx%-5d %8.3f ms  %9.1f Mpix/s" % (app, w, h, ms[3], w * h / ms[3] / 1e3))
