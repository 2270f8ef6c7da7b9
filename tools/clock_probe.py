#!/usr/bin/env python3
"""Steady-state clock/power of the GPU while a library renders frames back to back (run on the GPU box).

    python tools/clock_probe.py [--seconds 3] base name1 ...

Prints ms/frame (serial, one stream) and what rocm-smi reports for sclk / average power during the loop."""
import argparse
import os
import re
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import shaderbox_amd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=3.0)
ap.add_argument("--app", default="clouds")
ap.add_argument("--width", type=int, default=3840)
ap.add_argument("--height", type=int, default=2160)
ap.add_argument("names", nargs="+")
a = ap.parse_args()


def sample(stop, acc):
    while not stop.is_set():
        try:
            o = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            m = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", o)
            p = re.search(r"Power \(W\): ([\d.]+)", o)
            acc.append((int(m.group(1)) if m else None, float(p.group(1)) if p else None))
        except Exception as e:   # noqa: BLE001
            acc.append((None, None))
        time.sleep(0.05)


for name in a.names:
    path = shaderbox_amd.LIB_PATH if name == "base" else os.path.join(ROOT, "build", "ab", "libsbx_%s.so" % name)
    shaderbox_amd.LIB_PATH = path
    R = shaderbox_amd.Renderer(0)
    out = torch.empty((a.height, a.width, 4), dtype=torch.float32, device="cuda")
    for _ in range(5):
        R.render(a.app, a.width, a.height, .37, out=out)
    torch.cuda.synchronize()
    stop, acc = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, acc))
    th.start()
    n = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < a.seconds:
        for _ in range(20):
            R.render(a.app, a.width, a.height, .37, out=out)
        torch.cuda.synchronize()
        n += 20
    dt = time.perf_counter() - t0
    stop.set()
    th.join()
    clk = [c for c, _ in acc if c]
    pw = [p for _, p in acc if p]
    print("%-20s %.3f ms/frame over %d frames | sclk samples %d mean %s MHz min %s max %s | power mean %s W max %s"
          % (name, dt * 1e3 / n, n, len(clk), round(sum(clk) / len(clk)) if clk else None, min(clk) if clk else None,
             max(clk) if clk else None, round(sum(pw) / len(pw)) if pw else None, max(pw) if pw else None))
    R.close()
