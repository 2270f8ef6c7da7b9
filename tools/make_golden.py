#!/usr/bin/env python3
"""Generate tests/golden/*.npz: small float32 RGBA frames of every app at the canonical times.

The reference ships no golden images; these fixtures are outputs of the CPU oracle
(oracle/, itself pinned against SURVEY.md Appendix C by tests/test_oracle_kat.py), committed so that
(a) the oracle cannot drift silently and (b) the GPU parity tests have a fixed target that does not
depend on rebuilding the oracle on the GPU box.  Re-run after any deliberate change of the math spec:
    python tools/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.oracle import APP_IDS, Oracle  # noqa: E402

CASES = {"egg": (64, 64), "clouds": (96, 54), "raytracer": (64, 64), "atmosphere": (64, 36),
         "sdf_ao": (64, 36), "planet": (64, 36), "vinyl": (64, 36), "clouds_best": (96, 54),
         "clouds_ue4": (96, 54), "clouds_tex": (96, 54), "clouds_sky": (96, 54), "vinyl_gpu": (64, 36),
         "planet_atmosphere": (64, 36)}
TIMES = (0.0, 0.37, 2.5)


def fixture_volumes():
    """The two noise volumes of the clouds_tex fixture: closed-form pseudo-random texels (binary64 arithmetic rounded once),
    so that tests rebuild exactly the same bytes: shape 16^3, detail 8^3, RGBA32F with the value in .r."""
    vols = []
    for size, k in ((16, 12.9898), (8, 78.233)):
        i = np.arange(size ** 3, dtype=np.float64)
        r = np.modf(np.abs(np.sin(i * k + 1.0) * 43758.5453))[0]
        v = np.zeros((size, size, size, 4), dtype=np.float32)
        v[..., 0] = r.reshape(size, size, size).astype(np.float32)
        vols.append(v)
    return vols

if __name__ == "__main__":
    o = Oracle(rebuild=True)
    out = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out, exist_ok=True)
    o.set_noise_volumes(*fixture_volumes())
    only = set(sys.argv[1:])                  # python tools/make_golden.py [app ...]: only these fixtures
    for app, (w, h) in CASES.items():
        if only and app not in only:
            continue
        frames = {"t%g" % t: o.render(APP_IDS[app], w, h, t) for t in TIMES}
        np.savez_compressed(os.path.join(out, "%s_%dx%d.npz" % (app, w, h)), **frames)
        print(app, w, h, {k: float(np.nanmean(v[..., :3])) for k, v in frames.items()})
