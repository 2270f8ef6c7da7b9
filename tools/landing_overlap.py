#!/usr/bin/env python3
"""Do resident low-occupancy kernels of different streams overlap?  (diagnostic for bench.py's landing model)"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shaderbox_amd
R = shaderbox_amd.Renderer(0)
dev = torch.device("cuda", 0)
src = torch.zeros(8 << 20, dtype=torch.uint8, device=dev)
for ns in (1, 3):
    streams = [torch.cuda.Stream(device=dev) for _ in range(ns)]
    dsts = [torch.zeros_like(src) for _ in range(ns)]
    frames = [torch.empty((1080, 1920, 4), device=dev) for _ in range(ns)]
    for with_render in (False, True):
        def one(i):
            with torch.cuda.stream(streams[i % ns]):
                if with_render:
                    R.render("clouds", 1920, 1080, .37, out=frames[i % ns])
                R.model_landing(src, dsts[i % ns], src.numel(), 14, 600.0)
        for i in range(6): one(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); n = 60
        for i in range(n): one(i)
        torch.cuda.synchronize()
        print("streams %d, render %s: %.3f ms per call (landing kernel alone is 0.600)" % (ns, with_render, (time.perf_counter() - t0) * 1e3 / n))
