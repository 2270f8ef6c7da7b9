#!/usr/bin/env python3
"""tools/soak_tile_order.py [cases] [seed] — the dispatch order (csrc/sbx_tile_order.h) under a host that does everything at
once: APP_CLOUDS / CLOUDS_SKY / VINYL / EGG launches of more shapes than a context keeps tables for (least-recently-used replacement,
buffers that grow), whole frames and ranks' strips, runs of launches on one stream (tables built, adopted, refreshed, applied) mixed
with launches alternating over three streams (plain order), changing u_time; every frame is compared with the per-lane kernel's
(`set_variant(1)`: never uses a table) bit for bit.  Run on the GPU box."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import shaderbox_amd as sa

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
R = sa.Renderer(0)
R0 = sa.Renderer(0)                        # APP_EGG's reference frames (check)
ref_streams = [torch.cuda.Stream(), torch.cuda.Stream()]
nref = 0
streams = [torch.cuda.Stream() for _ in range(3)]
# (app, W, H): every one has >= 4096 tiles (smaller launches take no order); more shapes than TILE_ORDER_KEYS = 8 per app for clouds
shapes = ([("clouds", w, h) for w, h in ((1920, 1080), (2560, 1440), (3840, 2160), (1280, 720), (2048, 1152), (1600, 900), (3200, 1800),
                                          (2880, 1620), (1366, 768), (3440, 1440), (2560, 1080))]
          + [("clouds_sky", 1920, 1080), ("vinyl", 2048, 1152), ("vinyl", 2560, 1440), ("vinyl_gpu", 2048, 1152),
             ("egg", 1920, 1080), ("egg", 1280, 720), ("egg", 3840, 2160)])
bad = launches = ordered_seen = 0
kinds = {}


def check(app, W, H, t, got, rank=None, world=None):
    global bad
    global nref
    if app == "egg":                                  # (every k_egg variant takes the table: the reference is a context whose launches
        nref += 1                                     # alternate over two streams — plain hot-first order)
        with torch.cuda.stream(ref_streams[nref % 2]):
            ref = R0.render(app, W, H, t)
    else:
        R.set_variant(1)
        ref = R.render(app, W, H, t)
        R.set_variant(0)
    torch.cuda.synchronize()
    if rank is None:
        d = (got.view(torch.int32) != ref.view(torch.int32)).any(dim=-1)
    else:
        from shaderbox_amd import shard
        rows = torch.tensor(shard.rank_row_indices(H, 8, rank, world), device=got.device)
        d = (got.view(torch.int32)[rows] != ref.view(torch.int32)[rows]).any(dim=-1)
    n = int(d.sum().item())
    if n:
        bad += 1
        print("DIFF", app, W, H, t, rank, world, n, flush=True)


for c in range(cases):
    app, W, H = rng.choice(shapes)
    t0 = rng.choice((0.0, 0.37, 2.5, rng.uniform(0, 20)))
    mode = rng.choice(("run", "run", "flight", "strip"))
    n = rng.choice((1, 2, 3, 5, 9, 20, 40))
    kinds[mode] = kinds.get(mode, 0) + 1
    if mode == "strip" and app in ("clouds", "clouds_sky") and H * W >= 1920 * 1080 * 2:
        world = rng.choice((2, 4, 8))
        rank = rng.randrange(world)
        frame = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
        with torch.cuda.stream(streams[0]):
            for k in range(n):
                R.render_rank_in_place(app, W, H, t0 + 0.01 * k, 8, rank, world, frame)
                if rng.random() < 0.3:
                    torch.cuda.synchronize()
        torch.cuda.synchronize()
        launches += n
        check(app, W, H, t0 + 0.01 * (n - 1), frame, rank, world)
        continue
    outs = [torch.empty((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(3)]
    # APP_EGG's table belongs to a scene that stands still: most of its cases repeat one frame (into a poisoned buffer: a tile the table
    # lost would stay NaN)
    dtk = 0.0 if (app == "egg" and rng.random() < 0.75) else 0.01
    for k in range(n):
        s = streams[k % 3] if mode == "flight" else streams[1]
        with torch.cuda.stream(s):
            if dtk == 0.0:
                outs[k % 3].fill_(float("nan"))
            R.render(app, W, H, t0 + dtk * k, out=outs[k % 3])
        if rng.random() < 0.25:
            torch.cuda.synchronize()                  # (a table is adopted when the host sees its event)
    torch.cuda.synchronize()
    launches += n
    ordered_seen += int(R.tile_order(app)[0] > 0)
    for k in range(max(0, n - 3), n):
        check(app, W, H, t0 + dtk * k, outs[k % 3])
print("soak of the dispatch order: %d cases (%s), %d launches over %d shapes, %d cases ended with a table for their shape; %d cases "
      "with a frame differing from the per-lane kernel's" % (cases, ", ".join("%s %d" % kv for kv in sorted(kinds.items())), launches,
                                                             len(shapes), ordered_seen, bad))
sys.exit(1 if bad else 0)
