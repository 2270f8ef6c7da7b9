#!/usr/bin/env python3
"""Run on the GPU box: the random-frame sweep of tests/test_gpu_parity.py::test_clouds_random_sweep_default_equals_perlane — the
default APP_CLOUDS kernel against the plain per-lane kernel (sbx_set_variant 1) on N random (time, mouse, aux) frames — for the
shipped library and for every A/B library given (or, with --all, every build/ab/libsbx_v_*.so of `tools/ab_build.py
--all-variants`).  One process per library (two libsbx in one process register kernels of the same name).

    python tools/sweep_clouds_variants.py [--frames 120] [--all] [name ...]"""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one_planet(path, N):
    """the same for APP_PLANET (libraries named libsbx_v_pl_*): default kernel == plain kernel on random (time, mouse, size) frames,
    NaN == NaN (app_planet.h:270-273 makes NaN pixels)"""
    import numpy as np
    import torch
    import shaderbox_amd
    if path != "base":
        shaderbox_amd.LIB_PATH = path
    R = shaderbox_amd.Renderer(0)
    rng = np.random.default_rng(321)
    bad = 0
    for i in range(N):
        t = float(rng.uniform(0, 60)) if i % 3 else float(rng.uniform(0, 3))
        mouse = (float(rng.uniform(0, 600)), float(rng.uniform(0, 300))) if i % 2 else (0.0, 0.0)
        W, H = [(640, 360), (333, 187), (1280, 200)][i % 3]
        app = "planet_atmosphere" if i % 9 == 8 else "planet"
        R.set_variant(0); a = R.render(app, W, H, t, mouse=mouse).clone()
        R.set_variant(1); b = R.render(app, W, H, t, mouse=mouse)
        same = (a.view(torch.int32) == b.view(torch.int32)) | (torch.isnan(a) & torch.isnan(b))
        if not bool(same.all()):
            bad += 1
    R.set_variant(0)
    print("%-44s frames %d, with a differing pixel: %d%s" % (os.path.basename(path) + " (planet)", N, bad, "" if bad == 0 else "   <-- DIFFERENT"))


def one_raytracer(path, N):
    """APP_RAYTRACER (libraries named libsbx_v_rt_*): the default kernel == the witness's test edge (2) == the IEEE kernel with the six
    planes from the scene block (3) on random (time, mouse, size) frames"""
    import numpy as np
    import torch
    import shaderbox_amd
    if path != "base":
        shaderbox_amd.LIB_PATH = path
    R = shaderbox_amd.Renderer(0)
    rng = np.random.default_rng(55)
    bad = 0
    for i in range(N):
        t = float(rng.uniform(0, 100))
        mouse = (float(rng.uniform(1, 900)), float(rng.uniform(1, 500))) if i % 2 else (0.0, 0.0)
        W, H = [(640, 360), (333, 187), (1280, 720)][i % 3]
        fr = []
        for v in (0, 2, 3):
            R.set_variant(v)
            fr.append(R.render("raytracer", W, H, t, mouse=mouse).clone())
        same = (fr[0].view(torch.int32) == fr[1].view(torch.int32)) & (fr[0].view(torch.int32) == fr[2].view(torch.int32))
        if not bool(same.all()):
            bad += 1
    R.set_variant(0)
    print("%-44s frames %d, with a differing pixel: %d%s" % (os.path.basename(path) + " (raytracer)", N, bad, "" if bad == 0 else "   <-- DIFFERENT"))


def one_egg(path, N):
    """APP_EGG (libraries named libsbx_v_egg_*): the default kernel == the witness's test edge (2) == the un-culled IEEE kernel (1) on
    random (time, mouse, size) frames — for EGG_COOP=1 that is the survivor queue and the finisher kernel against the plain union"""
    import numpy as np
    import torch
    import shaderbox_amd
    if path != "base":
        shaderbox_amd.LIB_PATH = path
    R = shaderbox_amd.Renderer(0)
    rng = np.random.default_rng(77)
    bad = 0
    for i in range(N):
        t = float(rng.uniform(0, 100))
        mouse = (float(rng.uniform(1, 900)), float(rng.uniform(1, 500))) if i % 2 else (0.0, 0.0)
        W, H = [(1920, 1080), (640, 360), (333, 187), (1280, 720)][i % 4]
        fr = []
        for v in (0, 2, 1):
            R.set_variant(v)
            fr.append(R.render("egg", W, H, t, mouse=mouse).clone())
        same = (fr[0].view(torch.int32) == fr[1].view(torch.int32)) & (fr[0].view(torch.int32) == fr[2].view(torch.int32))
        if not bool(same.all()) or R.fault_status() != 0:
            bad += 1
    R.set_variant(0)
    print("%-44s frames %d, with a differing pixel: %d%s" % (os.path.basename(path) + " (egg)", N, bad, "" if bad == 0 else "   <-- DIFFERENT"))


def one(path, N):
    import numpy as np
    import torch
    import shaderbox_amd
    if "libsbx_v_egg_" in os.path.basename(path):
        return one_egg(path, max(20, N // 3))
    if "libsbx_v_pl_" in os.path.basename(path):
        return one_planet(path, max(20, N // 3))
    if "libsbx_v_rt_" in os.path.basename(path):
        return one_raytracer(path, max(20, N // 3))
    if path != "base":
        shaderbox_amd.LIB_PATH = path
    R = shaderbox_amd.Renderer(0)
    rng = np.random.default_rng(123)
    bad = 0
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
        if i % 11 == 0:
            aux.sun_dir[0] = 0.0                      # a sun in the y-z plane: the y-z light march
        W, H = (640, 360) if i % 4 else (333, 187)
        R.set_variant(0); a = R.render("clouds", W, H, t, mouse=mouse, aux=aux).clone()
        R.set_variant(1); b = R.render("clouds", W, H, t, mouse=mouse, aux=aux)
        same = (a.view(torch.int32) == b.view(torch.int32)) | (torch.isnan(a) & torch.isnan(b))
        if not bool(same.all()):
            bad += 1
    R.set_variant(0)
    print("%-44s frames %d, with a differing pixel: %d%s" % (os.path.basename(path), N, bad, "" if bad == 0 else "   <-- DIFFERENT"))


if __name__ == "__main__":
    args = sys.argv[1:]
    N = int(args[args.index("--frames") + 1]) if "--frames" in args else 120
    if "--one" in args:
        one(args[args.index("--one") + 1], N)
        sys.exit(0)
    if "--one-planet" in args:
        one_planet(args[args.index("--one-planet") + 1], N)
        sys.exit(0)
    if "--one-raytracer" in args:
        one_raytracer(args[args.index("--one-raytracer") + 1], N)
        sys.exit(0)
    if "--one-egg" in args:
        one_egg(args[args.index("--one-egg") + 1], N)
        sys.exit(0)
    names = [a for a in args if not a.startswith("--") and not a.isdigit()]
    paths = ["base"] + [os.path.join(ROOT, "build", "ab", "libsbx_%s.so" % n) for n in names]
    if "--all" in args:
        paths += sorted(glob.glob(os.path.join(ROOT, "build", "ab", "libsbx_v_*.so")))
    if "--all" in args:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one-planet", "base", "--frames", str(max(20, N // 3))], capture_output=True, text=True)
        print(([l for l in r.stdout.splitlines() if "frames" in l] or ["base (planet) FAILED: " + r.stderr[-300:]])[-1])
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one-raytracer", "base", "--frames", str(max(20, N // 3))], capture_output=True, text=True)
        print(([l for l in r.stdout.splitlines() if "frames" in l] or ["base (raytracer) FAILED: " + r.stderr[-300:]])[-1])
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one-egg", "base", "--frames", str(max(20, N // 3))], capture_output=True, text=True)
        print(([l for l in r.stdout.splitlines() if "frames" in l] or ["base (egg) FAILED: " + r.stderr[-300:]])[-1])
    for p in paths:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", p, "--frames", str(N)], capture_output=True, text=True)
        out = [l for l in r.stdout.splitlines() if "frames" in l]
        print(out[-1] if out else "%-44s FAILED: %s" % (os.path.basename(p), r.stderr.strip().splitlines()[-1] if r.stderr.strip() else "?"))
