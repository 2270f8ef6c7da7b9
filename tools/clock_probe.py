#!/usr/bin/env python3
"""What the chip SUSTAINS (run on the GPU box): frames back to back for seconds, with the shader clock and the board power sampled
from sysfs beside them (bench.py GpuSampler), for the BASELINE frames.

    python tools/clock_probe.py [--seconds 4] [--streams 3] [--configs clouds:3840x2160,atmosphere:7680x4320,planet:7680x4320]
                                [--libs base,name1,...]        (A/B libraries of tools/ab_build.py)

Per config: ms/frame and Mpixels/s over the first 50 ms after an idle second (the clock ramp), over the whole run and over its
last half; sclk mean / min / max, power mean / max; one un-overlapped launch (HIP events) at the end, i.e. at sustained clocks."""
import argparse
import importlib.util
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import shaderbox_amd  # noqa: E402

spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=4.0)
ap.add_argument("--streams", type=int, default=3)
ap.add_argument("--configs", default="clouds:3840x2160,atmosphere:7680x4320,planet:7680x4320")
ap.add_argument("--libs", default="base")
a = ap.parse_args()

dev = torch.device("cuda", 0)
for name in a.libs.split(","):
    shaderbox_amd.LIB_PATH = shaderbox_amd.LIB_PATH if name == "base" else os.path.join(ROOT, "build", "ab", "libsbx_%s.so" % name)
    R = shaderbox_amd.Renderer(0)
    R.set_timing(True)
    streams = [torch.cuda.Stream(device=dev) for _ in range(a.streams)]
    for cfg in a.configs.split(","):
        app, res = cfg.split(":")
        W, H = (int(v) for v in res.split("x"))
        frames = [torch.zeros((H, W, 4), dtype=torch.float32, device=dev) for _ in range(a.streams)]

        def step(i):
            with torch.cuda.stream(streams[i % a.streams]):
                R.render(app, W, H, .37, out=frames[i % a.streams])
        step(0)
        torch.cuda.synchronize()
        time.sleep(1.0)                                           # let the clocks fall back to idle
        marks = []                                                # (seconds since start, frames done)
        sampler = bench.GpuSampler(0)                             # (its sysfs search is host work: before the clock starts)
        n, t0 = 0, time.perf_counter()
        with sampler as smp:
            while time.perf_counter() - t0 < a.seconds:
                for i in range(2 * a.streams):
                    step(i)
                torch.cuda.synchronize()
                n += 2 * a.streams
                marks.append((time.perf_counter() - t0, n))
        total_s = marks[-1][0]

        def rate(lo, hi):
            inside = [(t, k) for t, k in marks if lo <= t <= hi]
            if len(inside) < 2:
                return float("nan")
            return (inside[-1][0] - inside[0][0]) * 1e3 / (inside[-1][1] - inside[0][1])
        k = []
        for _ in range(7):
            R.render(app, W, H, .37, out=frames[0])
            k.append(R.last_kernel_ms())
        torch.cuda.synchronize()
        k = sorted(k[2:])
        s = smp.summary()
        first, whole, tail = rate(0, .05), total_s * 1e3 / n, rate(total_s / 2, total_s)
        print("%-6s %-10s %dx%d  %d frames in flight, %.1f s, %d frames: first 50 ms %.4f ms/frame | whole run %.4f (%.0f Mpixels/s) | "
              "last half %.4f (%.0f Mpixels/s) | one launch afterwards %.4f ms | sclk MHz %s | power W %s | %s"
              % (name, app, W, H, a.streams, total_s, n, first, whole, W * H / whole / 1e3, tail, W * H / tail / 1e3, k[len(k) // 2],
                 s["sclk_mhz"], s["power_w"], s["source"]))
        del frames
        torch.cuda.empty_cache()
    R.close()
