#!/usr/bin/env python3
"""On ONE GPU: kernel time of each rank's share of the frame for N = 1, 2, 4, 8 (cyclic row-blocks),
i.e. the compute-only strong-scaling ceiling = T(1) / max_r T_r(N).  Usage: strip_scaling.py [app] [W H]"""
import sys
import torch
sys.path.insert(0, ".")
import shaderbox_amd
from shaderbox_amd import shard

app = sys.argv[1] if len(sys.argv) > 1 else "clouds"
import os
if os.environ.get("SBX_LIB"):              # an A/B library of tools/ab_build.py instead of the shipped one
    shaderbox_amd.LIB_PATH = os.environ["SBX_LIB"]
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
R.set_timing(False)          # per-launch event pairs are not part of the pipelined loop
streams = [torch.cuda.Stream() for _ in range(2)]   # created once: HIP maps streams onto a few hardware queues, and two
                                                     # streams that land on the same queue do not overlap at all
for n in (1, 2, 4, 8):
    torch.cuda.empty_cache()
    slabs = [torch.empty((shard.rank_rows_max(H, 8, n), W, 4), dtype=torch.float32, device="cuda") for _ in range(2)]
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

import os
if os.environ.get("SBX_STRIP_QUICK"):
    sys.exit(0)
# the ROOT's frame at N = 8, emulated on one GPU: its own strip + a 7-slab device copy standing in for the data RCCL's
# receive kernels write into its HBM + the assembly kernel, two frames in flight as in bench.py — for the plain cyclic
# split and for the split with root relief that bench.py's calibration picks (shard.relief_rounds)
n = 8


def frames_per_ms(fn):
    for i in range(6):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 40
    for i in range(K):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / K


def emulate_direct(m0, m, ch=3):
    """the direct exchange (distributed.FramePlan default): root in place, landing of 7 slabs without alpha, peer-only assembly"""
    rows_max = shard.rank_rows_max(H, 8, n, m0, m)
    bufs = [dict(slab=torch.empty((rows_max, W, ch), dtype=torch.float32, device="cuda"),
                 src=torch.rand((n - 1, rows_max, W, ch), dtype=torch.float32, device="cuda"),
                 peers=torch.empty((n - 1, rows_max, W, ch), dtype=torch.float32, device="cuda"),
                 frame=torch.empty((H, W, 4), dtype=torch.float32, device="cuda")) for _ in range(2)]

    def root_frame(i):
        b = bufs[i % 2]
        with torch.cuda.stream(streams[i % 2]):
            R.render_rank_in_place(app, W, H, 0.37, 8, 0, n, b["frame"], root_rounds=m0, rounds=m)
            b["peers"].copy_(b["src"])
            R.assemble_peers(b["peers"], W, H, 8, n, b["frame"], root_rounds=m0, rounds=m)

    def peer_frame(r):
        def f(i):
            with torch.cuda.stream(streams[i % 2]):
                R.render_rank_rows(app, W, H, 0.37, 8, r, n, 0, rows_max, bufs[i % 2]["slab"], root_rounds=m0, rounds=m)
        return f
    root_ms = frames_per_ms(root_frame)
    peer_ms = max(frames_per_ms(peer_frame(r)) for r in range(1, n))
    worst = max(root_ms, peer_ms)
    print("  DIRECT (%d channels), N=8, root sits out rounds >= %d of %d: root (in-place strip + landing + peer assembly) %.3f "
          "ms/frame, slowest peer %.3f ms/frame -> %.2fx of the N=1 frame rate" % (ch, m0, m, root_ms, peer_ms, base / worst))


def emulate(m0, m):
    rows_max = shard.rank_rows_max(H, 8, n, m0, m)
    bufs = [dict(slab=torch.empty((rows_max, W, 4), dtype=torch.float32, device="cuda"),
                 peers=torch.rand((n - 1, rows_max, W, 4), dtype=torch.float32, device="cuda"),
                 gathered=torch.empty((n, rows_max, W, 4), dtype=torch.float32, device="cuda"),
                 frame=torch.empty((H, W, 4), dtype=torch.float32, device="cuda")) for _ in range(2)]

    def root_frame(i):
        b = bufs[i % 2]
        with torch.cuda.stream(streams[i % 2]):
            R.render_rank(app, W, H, 0.37, 8, 0, n, out=b["slab"], root_rounds=m0, rounds=m)
            b["gathered"][0].copy_(b["slab"])
            b["gathered"][1:].copy_(b["peers"])
            R.assemble(b["gathered"], W, H, 8, n, out=b["frame"], root_rounds=m0, rounds=m)

    def peer_frame(r):
        def f(i):
            with torch.cuda.stream(streams[i % 2]):
                R.render_rank(app, W, H, 0.37, 8, r, n, out=bufs[i % 2]["slab"], root_rounds=m0, rounds=m)
        return f
    root_ms = frames_per_ms(root_frame)
    peer_ms = max(frames_per_ms(peer_frame(r)) for r in range(1, n))
    worst = max(root_ms, peer_ms)
    print("  2 streams, N=8, root sits out rounds >= %d of %d: root (strip + landing + assembly) %.3f ms/frame, slowest peer "
          "%.3f ms/frame -> %.2fx of the N=1 frame rate" % (m0, m, root_ms, peer_ms, base / worst))


emulate(1, 1)
# calibration as in bench.py choose_relief(): pipelined cost of the plain 1/8 strip (t_s) and the root-only work (e)
rmax = shard.rank_rows_max(H, 8, n)
src = torch.zeros((n - 1, rmax, W, 4), dtype=torch.float32, device="cuda")
g = torch.zeros((n, rmax, W, 4), dtype=torch.float32, device="cuda")
sl = [torch.empty((rmax, W, 4), dtype=torch.float32, device="cuda") for _ in range(2)]


def strip(i):
    with torch.cuda.stream(streams[i % 2]):
        R.render_rank(app, W, H, 0.37, 8, 0, n, out=sl[i % 2])


t_s = frames_per_ms(strip)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(3):
    if i == 1:
        a.record()
    g[1:].copy_(src)
    R.assemble(g, W, H, 8, n, out=frame)
b.record(); torch.cuda.synchronize()
e = a.elapsed_time(b) / 2.0
m0, m = shard.best_relief(H, 8, n, e / (n * t_s))
print("  calibration: plain strip %.3f ms/frame, root-only work %.3f ms per frame -> relief %d/%d" % (t_s, e, m0, m))
del src, g, sl
emulate(m0, m)

# the direct exchange (bench.py's default at N > 1), every candidate split of bench.py's calibration: it adopts the fastest
for ch in (4, 3):
    for m0, m in [(1, 1), (7, 8), (6, 7), (5, 6), (4, 5), (3, 4), (5, 7), (2, 3), (5, 8), (3, 5), (4, 7), (1, 2)]:
        emulate_direct(m0, m, ch)
