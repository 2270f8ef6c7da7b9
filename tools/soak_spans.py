#!/usr/bin/env python3
"""One-off soak (run on the GPU box): the multi-GPU schedule of every rank through a loopback world (distributed.LoopbackWorld:
real FramePlans, kernels, span tables, assembly on one device) on random (app, frame size, rank count, block rows, root relief,
pieces, exchange, time, mouse) against one launch of the same frame, bit for bit; and the point-list entry on random pixel
subsets of the same frames.    python tools/soak_spans.py [cases = 300] [seed = 1]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import shaderbox_amd
from shaderbox_amd.distributed import LoopbackWorld

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
R = shaderbox_amd.Renderer(0)
APPS = ["clouds", "atmosphere", "planet", "egg", "raytracer", "sdf_ao", "clouds_sky", "planet_atmosphere", "clouds_best"]
bad = 0
tally = {}
for c in range(n_cases):
    app = APPS[int(rng.integers(len(APPS)))]
    w = int(rng.integers(65, 2600))
    h = int(rng.integers(9, 900))
    n = int(rng.choice([2, 3, 4, 5, 8]))
    br = int(rng.choice([2, 4, 8, 8, 8, 16]))
    m = int(rng.integers(1, 6))
    m0 = int(rng.integers(0, m + 1))
    groups = int(rng.integers(1, 5))
    exchange = str(rng.choice(["spans", "spans", "direct", "stores", "stores", "span_stores", "span_stores", "packed_stores", "packed_stores"]))
    channels = int(rng.choice([3, 4]))
    t = float(rng.uniform(0, 30))
    mouse = (float(rng.uniform(0, 6.3)), 0.0) if app.startswith("clouds") and c % 2 else (0.0, 0.0)
    world = LoopbackWorld(n)
    plans = world.plans(R, w, h, block_rows=br, groups=groups, root_rounds=m0, rounds=m, exchange=exchange, channels=channels)
    plans[0].frame.fill_(-7.0)
    if exchange in ("stores", "span_stores") and channels == 3:
        plans[0].frame[..., 3] = 1.0              # the alpha a three-dword store leaves alone (it comes with the shared frame)
    got = LoopbackWorld.render(plans, app, t, mouse=mouse)
    ref = R.render(app, w, h, t, mouse=mouse)
    same = (got.view(torch.int32) == ref.view(torch.int32)) | (torch.isnan(got) & torch.isnan(ref))
    ok = bool(same.all())
    # a random subset of the frame's pixel centres through the point list
    k = int(rng.integers(1, 5000))
    idx = torch.from_numpy(rng.integers(0, w * h, k)).cuda()
    pts = torch.stack([(idx % w).float() + .5, (idx // w).float() + .5], dim=1).contiguous()
    pg = R.render_points(app, w, h, t, pts, mouse=mouse)
    pr = ref.reshape(-1, 4)[idx]
    ok2 = bool(((pg.view(torch.int32) == pr.view(torch.int32)) | (torch.isnan(pg) & torch.isnan(pr))).all())
    tally[(app, exchange)] = tally.get((app, exchange), 0) + 1
    if not (ok and ok2):
        bad += 1
        print("MISMATCH", app, w, h, n, br, (m0, m), groups, exchange, t, mouse, "frame" if not ok else "points")
    if exchange in ("stores", "span_stores", "packed_stores"):
        torch.cuda.synchronize()
        for p in plans[1:]:
            if p.shared is not None:
                p.shared.close()
        if plans[0].shared is not None:
            plans[0].shared.close()
    del plans, world, got, ref
# the library's own multi-GPU path (sbx_multi_*, all ranks on device 0: copies instead of RCCL), every exchange form
mbad, mcases = 0, 0
for n in (2, 3, 5, 8):
    M = shaderbox_amd.MultiRenderer([0] * n)
    streams = [torch.cuda.Stream() for _ in range(3)]
    for c in range(max(4, n_cases // 16)):
        app = APPS[int(rng.integers(len(APPS)))]
        w, h = int(rng.integers(65, 2000)), int(rng.integers(9, 700))
        m = int(rng.integers(1, 5)); m0 = int(rng.integers(0, m + 1))
        mode = str(rng.choice(["spans", "slabs", "blocks", "peer_stores"]))
        t = float(rng.uniform(0, 30))
        M.set_split(int(rng.choice([2, 4, 8, 16])), m0, m)
        M.set_exchange(mode)
        with torch.cuda.stream(streams[c % 3]):
            got = M.render(app, w, h, t)
        torch.cuda.synchronize()
        ref = R.render(app, w, h, t)
        same = (got.view(torch.int32) == ref.view(torch.int32)) | (torch.isnan(got) & torch.isnan(ref))
        mcases += 1
        if not bool(same.all()):
            mbad += 1
            print("MISMATCH sbx_multi", app, w, h, n, (m0, m), mode, t)
    M.close()
print("sbx_multi: %d cases over 2, 3, 5, 8 ranks and the four exchange forms, %d with a differing pixel" % (mcases, mbad))
print("cases %d: %s" % (n_cases, ", ".join("%s/%s %d" % (a, e, v) for (a, e), v in sorted(tally.items()))))
print("soak of the multi-GPU schedule and the point list: %d cases, %d with a differing pixel" % (n_cases, bad))
