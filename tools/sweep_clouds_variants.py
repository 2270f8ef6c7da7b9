#!/usr/bin/env python3
"""One-off soak (run on the GPU box, PYTHONPATH=.): the shipped APP_CLOUDS kernel against the plain per-lane kernel
(sbx_set_variant 1) on N random frames (time, mouse, aux).  tests/test_gpu_parity.py runs a short version."""
import sys, torch, numpy as np
sys.path.insert(0, ".")
import shaderbox_amd
R = shaderbox_amd.Renderer(0)
rng = np.random.default_rng(123)
bad = 0
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for i in range(N):
    aux = shaderbox_amd.clouds_defaults(R.lib)
    t = float(rng.uniform(0, 50)) if i % 3 else float(rng.uniform(0, 3))
    mouse = (float(rng.uniform(0, 6.3)), 0.0) if i % 2 else (0.0, 0.0)
    aux.cld_coverage = float(rng.uniform(0.2, 0.9))
    aux.cld_march_steps = int(rng.integers(10, 160))
    aux.illum_march_steps = int(rng.integers(0, 9))
    aux.cld_thick = float(rng.uniform(40, 300))
    aux.sigma_scattering = float(rng.uniform(.02, .6))
    if i % 5 == 0:
        aux.wind_dir[0], aux.wind_dir[1], aux.wind_dir[2] = [float(x) for x in rng.uniform(-.3, .3, 3)]
    if i % 7 == 0:
        d = rng.standard_normal(3); d /= np.linalg.norm(d)
        aux.sun_dir[0], aux.sun_dir[1], aux.sun_dir[2] = [float(x) for x in d]
    W, H = (640, 360) if i % 4 else (333, 187)
    R.set_variant(0); a = R.render("clouds", W, H, t, mouse=mouse, aux=aux).clone()
    R.set_variant(1); b = R.render("clouds", W, H, t, mouse=mouse, aux=aux)
    same = (a.view(torch.int32) == b.view(torch.int32)) | (torch.isnan(a) & torch.isnan(b))
    if not bool(same.all()):
        bad += 1
        print("MISMATCH", i, t, mouse, aux.cld_coverage, aux.cld_march_steps, int((~same).sum()))
R.set_variant(0)
print("frames", N, "mismatching", bad)
