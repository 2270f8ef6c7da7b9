#!/usr/bin/env python3
"""What ONE more launch costs the pipelined throughput (run on the GPU box): the same 4K CLOUDS frame rendered as K in-place launches
(the K ranks' row-blocks of a K-way cyclic split, all into one frame) for K = 1, 2, 4, 8, 16, with S frames in flight.

    python tools/launch_granularity.py [--app clouds --width 3840 --height 2160 --streams 3 --ks 1,2,4,8,16] [--flags]

The pixels per frame are the same for every K, so (ms per frame at K) - (ms per frame at 1) over K - 1 is the throughput cost of a
launch that the frames in flight do not hide: the term that separates an N-GPU frame of the store exchange (every rank = one launch
of 1/N of the frame) from N x the one-GPU rate.  --flags adds the store exchange's two flag kernels around every launch."""
import argparse
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import shaderbox_amd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--app", default="clouds")
ap.add_argument("--width", type=int, default=3840)
ap.add_argument("--height", type=int, default=2160)
ap.add_argument("--streams", type=int, default=3)
ap.add_argument("--ks", default="1,2,4,8,16")
ap.add_argument("--flags", action="store_true")
ap.add_argument("--seconds", type=float, default=.6)
a = ap.parse_args()
dev = torch.device("cuda", 0)
R = shaderbox_amd.Renderer(0)
W, H, app = a.width, a.height, a.app
streams = [torch.cuda.Stream(device=dev) for _ in range(a.streams)]
owners = [R.shared_create(H * W * 16, 2) for _ in range(a.streams)]
peers = [R.shared_open(o.export()) for o in owners]
views = [o.tensor((H, W, 4)) for o in owners]
base = None
for K in [int(v) for v in a.ks.split(",")]:
    def launch(j):
        s = j % a.streams
        with torch.cuda.stream(streams[s]):
            if a.flags:
                owners[s].begin(0)
                peers[s].begin(1)
            R.render_rank_in_place(app, W, H, .37, 8, j % K, K, views[s], channels=3)
            if a.flags:
                peers[s].end(1)
    for j in range(4 * K):
        launch(j)
    torch.cuda.synchronize()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < a.seconds:
        for j in range(6 * K):
            launch(n + j)
        torch.cuda.synchronize()
        n += 6 * K
    ms = (time.perf_counter() - t0) * 1e3 / (n / K)
    base = base or ms
    print("K=%2d launches per frame%s: %.4f ms per frame (%.4f per launch)%s" % (K, " + flag kernels" if a.flags else "", ms, ms / K,
          "" if K == 1 else "  -> %.1f us per extra launch, %.2fx of K=1 per 1/K" % ((ms - base) / (K - 1) * 1e3, base / ms * K)))
