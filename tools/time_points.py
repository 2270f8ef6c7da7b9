#!/usr/bin/env python3
"""Latency of the host-array point entries (run on the GPU box): an off-centre sbx_main_image call (a one-point launch) and
sbx_main_image_batch with n points, per library.     python tools/time_points.py [base|name ...]   (names: build/ab/libsbx_<name>.so)"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 2 and sys.argv[1] == "--one":
    import numpy as np
    import shaderbox_amd
    name = sys.argv[2]
    if name != "base":
        shaderbox_amd.LIB_PATH = os.path.join(ROOT, "build", "ab", "libsbx_%s.so" % name)
    R = shaderbox_amd.Renderer(0)
    out = []
    for app in ("egg", "clouds"):
        for _ in range(200):
            R.main_image(app, 640, 360, 0.37, (100.25, 50.75))
        t0 = time.perf_counter()
        n = 3000
        for i in range(n):
            R.main_image(app, 640, 360, 0.37, (100.25 + (i & 63), 50.75))
        one = (time.perf_counter() - t0) * 1e6 / n
        rng = np.random.default_rng(1)
        res = []
        for m in (64, 4096, 262144, 4194304):
            fc = rng.uniform(0, 360, (m, 2)).astype(np.float32)
            R.main_image_batch(app, 640, 360, 0.37, fc)
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                R.main_image_batch(app, 640, 360, 0.37, fc)
                ts.append((time.perf_counter() - t0) * 1e3)
            res.append("%d points %.3f ms" % (m, sorted(ts)[2]))
        out.append("%s: off-centre sbx_main_image %.1f us per call; sbx_main_image_batch %s" % (app, one, ", ".join(res)))
    print("%-8s %s" % (name, " | ".join(out)))
    sys.exit(0)
for name in (sys.argv[1:] or ["base"]):
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", name], capture_output=True, text=True)
    print(([l for l in r.stdout.splitlines() if "us per call" in l] or [name + " FAILED " + r.stderr[-300:]])[-1])
