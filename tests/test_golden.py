"""Committed golden frames (tests/golden, made by tools/make_golden.py from the oracle):
CPU suite: the oracle still reproduces them bit-for-bit; GPU suite: so do the HIP kernels."""
import glob
import os

import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
from make_golden import fixture_volumes  # noqa: E402

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "*.npz")))


def _cases():
    for path in GOLDEN:
        name = os.path.basename(path)[:-4]
        app, res = name.rsplit("_", 1)
        w, h = (int(v) for v in res.split("x"))
        yield path, app, w, h


def _same(a, b):
    return bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))))


def test_golden_set_is_complete():
    assert {c[1] for c in _cases()} == {"egg", "clouds", "raytracer", "atmosphere", "sdf_ao", "planet", "vinyl", "clouds_best",
                                        "clouds_ue4", "clouds_tex", "clouds_sky", "vinyl_gpu", "planet_atmosphere"}


@pytest.mark.parametrize("path,app,w,h", list(_cases()))
def test_oracle_reproduces_golden(oracle, path, app, w, h):
    from oracle.oracle import APP_IDS
    z = np.load(path)
    oracle.set_noise_volumes(*fixture_volumes())
    for key in z.files:
        assert _same(oracle.render(APP_IDS[app], w, h, float(key[1:])), z[key]), (app, key)


@pytest.mark.gpu
@pytest.mark.parametrize("path,app,w,h", list(_cases()))
def test_kernels_reproduce_golden(path, app, w, h):
    import shaderbox_amd
    import torch
    r = shaderbox_amd.Renderer(0)
    if app == "clouds_tex":
        r.set_noise_volumes(*[torch.from_numpy(v).cuda() for v in fixture_volumes()])
    z = np.load(path)
    for key in z.files:
        got = r.render(app, w, h, float(key[1:])).cpu().numpy()
        assert _same(got, z[key]), (app, key)
