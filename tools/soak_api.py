#!/usr/bin/env python3
"""One-off soak of the frame-granular entry points (run on the GPU box): random row strips (sbx_render_rows y0..y1), rank slabs
with relief and sub-ranges (sbx_render_split / _rgb / _in_place), assemblies, on random apps, sizes, times — every result against
the corresponding pixels of ONE full-frame launch, bit for bit; stream captures replayed.   python tools/soak_api.py [cases=300] [seed=1]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import shaderbox_amd
from shaderbox_amd import shard

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
R = shaderbox_amd.Renderer(0)
APPS = ["clouds", "atmosphere", "planet", "egg", "raytracer", "sdf_ao", "vinyl", "clouds_best", "clouds_ue4", "clouds_sky", "vinyl_gpu",
        "planet_atmosphere"]


def same(a, b):
    return bool(((a.view(torch.int32) == b.view(torch.int32)) | (torch.isnan(a) & torch.isnan(b))).all())


bad = 0
for c in range(n_cases):
    app = APPS[int(rng.integers(len(APPS)))]
    w, h = int(rng.integers(1, 1500)), int(rng.integers(1, 600))
    t = float(rng.uniform(0, 40))
    full = R.render(app, w, h, t)
    errs = []
    # a strip
    y0 = int(rng.integers(0, h)); y1 = int(rng.integers(y0, h + 1))
    s = R.render(app, w, h, t, rows=(y0, y1))
    if not same(s[:y1 - y0], full[y0:y1]):
        errs.append("strip %d..%d" % (y0, y1))
    # a rank's slab, with relief, in two sub-ranges, RGBA and RGB
    n = int(rng.integers(1, 9)); br = int(rng.choice([1, 2, 3, 4, 8, 16]))
    m = int(rng.integers(1, 5)); m0 = int(rng.integers(0, m + 1)) if n > 1 else m
    rank = int(rng.integers(0, n))
    rows = shard.rank_row_indices(h, br, rank, n, m0, m)
    rmax = shard.rank_rows_max(h, br, n, m0, m)
    if rows:
        cut = (int(rng.integers(0, len(rows) + 1)) // br) * br
        for ch in (4, 3):
            slab = torch.full((max(rmax, 1), w, ch), -3.0, dtype=torch.float32, device="cuda")
            R.render_rank_rows(app, w, h, t, br, rank, n, 0, cut, slab, root_rounds=m0, rounds=m)
            R.render_rank_rows(app, w, h, t, br, rank, n, cut, len(rows), slab, root_rounds=m0, rounds=m)
            if not same(slab[:len(rows)], full[rows][..., :ch]):
                errs.append("slab rank %d/%d br %d relief %d/%d ch %d" % (rank, n, br, m0, m, ch))
        frame = torch.full((h, w, 4), -3.0, dtype=torch.float32, device="cuda")
        R.render_rank_in_place(app, w, h, t, br, rank, n, frame, root_rounds=m0, rounds=m)
        mask = torch.zeros(h, dtype=torch.bool, device="cuda"); mask[rows] = True
        if not (same(frame[mask], full[mask]) and bool((frame[~mask] == -3.0).all())):
            errs.append("in place rank %d/%d" % (rank, n))
    # the whole split assembled (gather form)
    if n > 1 and c % 3 == 0:
        g = torch.zeros((n, max(rmax, 1), w, 4), dtype=torch.float32, device="cuda")
        for r in range(n):
            R.render_rank(app, w, h, t, br, r, n, out=g[r], root_rounds=m0, rounds=m)
        out = R.assemble(g, w, h, br, n, root_rounds=m0, rounds=m)
        if not same(out, full):
            errs.append("assemble %d ranks" % n)
    # a captured launch replayed
    if c % 5 == 0:
        st = torch.cuda.Stream()
        buf = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
        R.render(app, w, h, t, out=buf)                         # tables of this frame exist before the capture
        buf.zero_()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.stream(st):
            with torch.cuda.graph(gr, stream=st):
                R.render(app, w, h, t, out=buf)
        gr.replay(); torch.cuda.synchronize()
        if not same(buf, full):
            errs.append("graph replay")
    if errs:
        bad += 1
        print("MISMATCH", app, w, h, t, errs)
print("soak of the frame-granular ABI: %d cases (strips, slabs with relief in pieces, RGB slabs, in-place, assemblies, graph replays), "
      "%d with a differing pixel" % (n_cases, bad))
