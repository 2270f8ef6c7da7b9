#!/usr/bin/env python3
"""One-off soak (run on the GPU box): the shipped k_clouds against the plain per-lane kernel (sbx_set_variant 1) on random
uniforms and aux blocks, with the ranges stretched over the borders of the kernel's special cases: sigma * dt across the REG
limit of 80, positions far from the origin (u_time to 3e6, wind to 50: across and beyond the Lipschitz skip's domain), z-only suns of any length and general suns (ZL on / off), wind with a y component (y-table key changes every
frame), tiny / huge / negative thickness, 1..300 march steps, 0..12 light steps, coverage 0..1.
    python tools/soak_clouds.py [frames=2000] [seed=1] [scale=1]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import shaderbox_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
scale = int(sys.argv[3]) if len(sys.argv) > 3 else 1
R = shaderbox_amd.Renderer(0)
bad = 0
small = 0
kinds = {"reg_zl": 0, "reg_yz": 0, "reg_gen": 0, "noreg": 0}
for i in range(n):
    aux = shaderbox_amd.clouds_defaults(R.lib)
    t = float(rng.uniform(0, 100)) if i % 3 else float(rng.uniform(0, 3))
    if i % 7 == 0:                       # far from the origin: either side of the Lipschitz skip's domain (|wind_off| ~ 2^17) and beyond
        t = float(rng.choice([rng.uniform(400, 900), 10.0 ** rng.uniform(3, 6.5), -1e9]))
    mouse = (float(rng.uniform(0, 6.3)), 0.0) if i % 2 else (0.0, 0.0)
    aux.cld_coverage = float(rng.choice([rng.uniform(0.3, 0.8), rng.uniform(0, 1), 0.0, 1.0], p=[.6, .3, .05, .05]))
    aux.cld_march_steps = int(rng.choice([rng.integers(20, 160), rng.integers(1, 300)]))
    aux.illum_march_steps = int(rng.integers(0, 13))
    aux.cld_thick = float(rng.choice([rng.uniform(40, 300), rng.uniform(.01, 5), rng.uniform(1e3, 1e5), -rng.uniform(10, 200)],
                                     p=[.7, .1, .15, .05]))
    aux.sigma_scattering = float(rng.choice([rng.uniform(.02, .6), rng.uniform(1, 200), 0.0], p=[.75, .2, .05]))
    if i % 4 == 0:
        aux.wind_dir[0], aux.wind_dir[1], aux.wind_dir[2] = [float(x) for x in rng.uniform(-.3, .3, 3)]
    if i % 28 == 0:
        aux.wind_dir[0], aux.wind_dir[1], aux.wind_dir[2] = [float(x) for x in rng.uniform(-50, 50, 3)]
    k = i % 6
    if k == 0:
        d = rng.standard_normal(3); d /= np.linalg.norm(d)
        aux.sun_dir[0], aux.sun_dir[1], aux.sun_dir[2] = [float(x) for x in d]
    elif k == 1:
        aux.sun_dir[0], aux.sun_dir[1], aux.sun_dir[2] = 0.0, 0.0, float(rng.choice([-2.0, 1.0, .3, -1e-3]))
    elif k == 2:                        # the y-z plane (light_march_yz)
        aux.sun_dir[0], aux.sun_dir[1], aux.sun_dir[2] = 0.0, float(rng.choice([rng.uniform(-1, 1), 1e-6, 3.0])), float(rng.uniform(-1, 1))
    dt = aux.cld_thick / max(aux.cld_march_steps, 1)
    if i % 5 == 0 and dt > 0:           # sigma * dt on both sides of exp_small_'s bound (.94 sigma dt <= .2049, i.e. .218)
        aux.sigma_scattering = float(rng.uniform(.12, .25)) / dt
    small += int(aux.sigma_scattering >= 0 and dt >= 0 and .94 * aux.sigma_scattering * dt <= .2049)
    reg = abs(aux.sigma_scattering * dt) <= 80
    zl = aux.sun_dir[0] * dt == 0 and aux.sun_dir[1] * dt == 0
    yz = not zl and aux.sun_dir[0] * dt == 0
    kinds["noreg" if not reg else ("reg_zl" if zl else ("reg_yz" if yz else "reg_gen"))] += 1
    W, H = [(320, 180), (333, 187), (160, 284)][i % 3]
    W, H = W * scale, H * scale
    R.set_variant(0); a = R.render("clouds", W, H, t, mouse=mouse, aux=aux).clone()
    R.set_variant(1); b = R.render("clouds", W, H, t, mouse=mouse, aux=aux)
    same = (a.view(torch.int32) == b.view(torch.int32)) | (torch.isnan(a) & torch.isnan(b))
    if not bool(same.all()):
        bad += 1
        print("MISMATCH frame %d: %d pixels; t=%g mouse=%s cov=%g steps=%d lsteps=%d thick=%g sigma=%g sun=%s wind=%s"
              % (i, int((~same).any(-1).sum()), t, mouse, aux.cld_coverage, aux.cld_march_steps, aux.illum_march_steps,
                 aux.cld_thick, aux.sigma_scattering, list(aux.sun_dir), list(aux.wind_dir)))
print("soak: %d frames (%s; %d of them inside exp_small_'s domain), %d with differing pixels" % (n, kinds, small, bad))
