#!/usr/bin/env python3
"""One-off soak (run on the GPU box): every app's shipped kernel against the CPU oracle on random times, mouse positions and
odd frame sizes (the golden frames and the parity tests fix a handful of each).  EGG, SDF_AO, VINYL and PLANET additionally
against their plain form (sbx_set_variant 1: no culling / no skips) at a larger size.
    python tests/soak_apps.py [frames per app = 24] [seed = 1]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import shaderbox_amd
from oracle.oracle import APP_IDS, Oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
R = shaderbox_amd.Renderer(0)
O = Oracle()
APPS = ["egg", "sdf_ao", "vinyl", "raytracer", "atmosphere", "planet", "clouds", "clouds_best", "clouds_ue4", "clouds_tex", "clouds_sky",
        "vinyl_gpu", "planet_atmosphere"]
# APP_CLOUDS' USE_NOISE_TEX build: two small baked volumes (32^3 shape, 16^3 detail) bound on both sides
_v1, _v2 = R.worley_volume(32), R.worley_volume(16)
R.set_noise_volumes(_v1, _v2)
O.set_noise_volumes(_v1.cpu().numpy(), _v2.cpu().numpy())
SIZES = [(160, 90), (97, 61), (128, 128), (211, 40), (64, 150)]
total_bad = 0
for app in APPS:
    bad = worst = 0
    for i in range(n):
        W, H = SIZES[i % len(SIZES)]
        t = float(rng.uniform(0, 60)) if i % 3 else float(rng.uniform(0, 3))
        mouse = (float(rng.uniform(0, W)), float(rng.uniform(0, H))) if i % 2 else (0.0, 0.0)
        R.set_variant(0)
        aux = None
        if app == "clouds_tex" and i % 2:                     # the aux block's code paths: z-only suns of any sign and length
            aux = shaderbox_amd.clouds_defaults(R.lib)        # (the z-only light march), general suns, coverage, step counts
            aux.cld_coverage = float(rng.uniform(.2, .8))
            aux.cld_march_steps = int(rng.integers(10, 140))
            aux.illum_march_steps = int(rng.integers(0, 10))
            aux.cld_thick = float(rng.choice([rng.uniform(40, 300), -rng.uniform(10, 200)], p=[.85, .15]))
            if i % 4 == 1:
                aux.sun_dir[0], aux.sun_dir[1], aux.sun_dir[2] = 0.0, 0.0, float(rng.choice([-2.0, 1.0, .3, 0.0]))
            else:
                d = rng.standard_normal(3); d /= np.linalg.norm(d)
                aux.sun_dir[0], aux.sun_dir[1], aux.sun_dir[2] = [float(x) for x in d]
        g = R.render(app, W, H, t, mouse=mouse, aux=aux).cpu().numpy()
        ref = O.render(APP_IDS[app], W, H, t, mouse=mouse, aux=aux)
        both_nan = np.isnan(g) & np.isnan(ref)
        diff = ((g.view(np.uint32) != ref.view(np.uint32)) & ~both_nan).any(-1)
        if diff.any():
            bad += 1
            worst = max(worst, int(diff.sum()))
            print("MISMATCH %s frame %d: %d pixels; %dx%d t=%r mouse=%r" % (app, i, int(diff.sum()), W, H, t, mouse))
    plain = ""
    if app in ("egg", "sdf_ao", "vinyl", "vinyl_gpu", "raytracer", "planet"):      # (RAYTRACER: variant 1 = its IEEE roots / normalisations)
        pb = 0
        for i in range(max(4, n // 4)):
            t = float(rng.uniform(0, 60))
            mouse = (float(rng.uniform(0, 1280)), float(rng.uniform(0, 720))) if i % 2 else (0.0, 0.0)
            R.set_variant(0); a = R.render(app, 1280, 720, t, mouse=mouse).clone()
            R.set_variant(1); b = R.render(app, 1280, 720, t, mouse=mouse).clone()
            same = (a.view(torch.int32) == b.view(torch.int32)) | (torch.isnan(a) & torch.isnan(b))
            if app != "planet" and i % 2 == 0:          # the recorded-domain roots with their re-run forced (variant 2: sbx_witness.h)
                R.set_variant(2); c = R.render(app, 1280, 720, t, mouse=mouse)
                same &= (a.view(torch.int32) == c.view(torch.int32)) | (torch.isnan(a) & torch.isnan(c))
            if not bool(same.all()):
                pb += 1
                print("MISMATCH %s vs plain form: %d pixels; t=%r mouse=%r" % (app, int((~same).any(-1).sum()), t, mouse))
        R.set_variant(0)
        plain = "; %d 1280x720 frames against the plain form, %d differing" % (max(4, n // 4), pb)
        bad += pb
    print("soak %-12s %d frames against the oracle, %d with differing pixels%s" % (app, n, bad if not plain else bad - pb, plain))
    total_bad += bad
print("soak: %d apps, %d frames with differing pixels" % (len(APPS), total_bad))
sys.exit(1 if total_bad else 0)
