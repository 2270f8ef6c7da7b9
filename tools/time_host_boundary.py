#!/usr/bin/env python3
"""The HOST-visible rate of the per-pixel drop-in (run on the GPU box): what sbx_main_image costs when the frame is not cached — one
launch whose stores go straight into the cache's pinned host frame (16 B/pixel over PCIe; until round 5: launch + copy) — for the BASELINE
frames, beside the kernel alone and beside a plain device-to-pinned copy of the same bytes.  bench.py's `value` is measured with the
frame left in HBM (the C ABI's sbx_render* write device memory); this is the PCIe-inclusive figure DESIGN.md §1 quotes.

    python tools/time_host_boundary.py [--configs clouds:3840x2160,...] [--reps 6]"""
import argparse
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import shaderbox_amd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--configs", default="egg:1920x1080,raytracer:3840x2160,clouds:3840x2160,atmosphere:7680x4320,planet:7680x4320")
ap.add_argument("--reps", type=int, default=6)
a = ap.parse_args()

R = shaderbox_amd.Renderer(0)
R.set_timing(True)
dev = torch.device("cuda", 0)
for cfg in a.configs.split(","):
    app, res = cfg.split(":")
    W, H = (int(v) for v in res.split("x"))
    frame = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
    host = torch.empty((H, W, 4), dtype=torch.float32).pin_memory()
    for _ in range(3):                                            # warm: clocks, the context's pinned buffers (two entries)
        R.main_image(app, W, H, 100.0 + _, (0.5, 0.5))
    miss = []
    for i in range(a.reps):                                       # a new u_time each call: every call is a miss
        t0 = time.perf_counter()
        R.main_image(app, W, H, 0.37 + 0.01 * i, (0.5, 0.5))
        miss.append((time.perf_counter() - t0) * 1e3)
    t0 = time.perf_counter()
    n = 20000
    for i in range(n):                                            # hits: the host copy, no device work (Python + ctypes per call)
        R.main_image(app, W, H, 0.37 + 0.01 * (a.reps - 1), (0.5 + (i % W), 0.5))
    hit_us = (time.perf_counter() - t0) * 1e6 / n
    k = []
    for _ in range(5):
        R.render(app, W, H, .37, out=frame)
        k.append(R.last_kernel_ms())
    torch.cuda.synchronize()
    c = []
    for _ in range(5):
        t0 = time.perf_counter()
        host.copy_(frame, non_blocking=True)
        torch.cuda.synchronize()
        c.append((time.perf_counter() - t0) * 1e3)
    h, hp = [], []
    pageable = torch.empty((H, W, 4), dtype=torch.float32)
    for _ in range(6):                                            # sbx_render_rows_host: strips copied out while the next ones render
        t0 = time.perf_counter()
        R.render_to_host(app, W, H, .37, host)
        h.append((time.perf_counter() - t0) * 1e3)
    for _ in range(3):
        t0 = time.perf_counter()
        R.render_to_host(app, W, H, .37, pageable)
        hp.append((time.perf_counter() - t0) * 1e3)
    z = []
    u = R.uniforms(W, H, .37)
    for _ in range(6):                                            # sbx_render_rows handed the pinned frame: the kernel's stores cross PCIe
        t0 = time.perf_counter()
        R._check(R.lib.sbx_render_rows(R.ctx, shaderbox_amd.app_id(app), ctypes.byref(u), None, 0, H, ctypes.c_void_p(host.data_ptr()), None))
        torch.cuda.synchronize()
        z.append((time.perf_counter() - t0) * 1e3)
    z.sort()
    zm = z[len(z) // 2]
    h.sort(); hp.sort()
    hm, hpm = h[len(h) // 2], hp[len(hp) // 2]
    miss.sort(); k.sort(); c.sort()
    m, km, cm = miss[len(miss) // 2], k[len(k) // 2], c[len(c) // 2]
    mb = W * H * 16 / 1e6
    print("%-10s %dx%d  sbx_main_image miss (one launch storing %.0f MB into the pinned frame) %.3f ms = %.0f Mpixels/s host-visible | kernel alone %.3f ms "
          "(%.0f Mpixels/s) | copy alone %.3f ms (%.1f GB/s) | hit %.2f us per call through ctypes | "
          "sbx_render_rows_host into pinned memory %.3f ms = %.0f Mpixels/s, into pageable memory %.3f ms = %.0f Mpixels/s | "
          "sbx_render_rows storing straight into the pinned frame %.3f ms"
          % (app, W, H, mb, m, W * H / m / 1e3, km, W * H / km / 1e3, cm, mb / cm, hit_us, hm, W * H / hm / 1e3, hpm, W * H / hpm / 1e3, zm))
    del frame, host, pageable
    torch.cuda.empty_cache()
R.close()
