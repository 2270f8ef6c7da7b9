#!/usr/bin/env python3
"""On ONE GPU: kernel time of each rank's share of the frame for N = 1, 2, 4, 8 (cyclic row-blocks),
i.e. the compute-only strong-scaling ceiling = T(1) / max_r T_r(N).  Usage: strip_scaling.py [app] [W H]"""
import sys
import torch
sys.path.insert(0, ".")
import shaderbox_amd
from shaderbox_amd import shard

app = sys.argv[1] if len(sys.argv) > 1 else "clouds"
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3840, 2160)
R = shaderbox_amd.Renderer(0)
R.set_timing(True)


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        fn(); ms.append(R.last_kernel_ms())
    ms.sort()
    return ms[len(ms) // 2]


frame = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
t1 = timed(lambda: R.render(app, W, H, 0.37, out=frame))
print("%s %dx%d  N=1: %.3f ms" % (app, W, H, t1))
for br in (8, 16):
    for n in (2, 4, 8):
        slab = torch.empty((shard.rank_rows_max(H, br, n), W, 4), dtype=torch.float32, device="cuda")
        ts = [timed(lambda r=r: R.render_rank(app, W, H, 0.37, br, r, n, out=slab)) for r in range(n)]
        print("  block_rows=%2d N=%d: per-rank ms min %.3f max %.3f  -> compute-only speed-up %.2fx (ideal %d)"
              % (br, n, min(ts), max(ts), t1 / max(ts), n))

# throughput with frames pipelined over 2 streams (what bench.py does): ms per frame of one rank's share
import time
for n in (1, 2, 4, 8):
    slabs = [torch.empty((shard.rank_rows_max(H, 8, n), W, 4), dtype=torch.float32, device="cuda") for _ in range(2)]
    streams = [torch.cuda.Stream() for _ in range(2)]
    worst = 0.0
    for r in range(n):
        for i in range(6):
            with torch.cuda.stream(streams[i % 2]):
                R.render_rank(app, W, H, 0.37, 8, r, n, out=slabs[i % 2])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 40
        for i in range(K):
            with torch.cuda.stream(streams[i % 2]):
                R.render_rank(app, W, H, 0.37, 8, r, n, out=slabs[i % 2])
        torch.cuda.synchronize()
        worst = max(worst, (time.perf_counter() - t0) * 1e3 / K)
    if n == 1:
        base = worst
    print("  2 streams, N=%d: slowest rank %.3f ms/frame -> compute-only speed-up %.2fx" % (n, worst, base / worst))
