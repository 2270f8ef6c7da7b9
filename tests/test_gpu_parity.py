"""GPU parity: HIP kernels (through the C ABI) vs the CPU oracle, same uniforms, small frames.

Bar (BASELINE.json north_star): every float channel within 1e-4 of the reference path.  Because
oracle and kernels implement one math spec with IEEE-exact operations, the kernels are expected to
be BIT-IDENTICAL to the oracle (NaN == NaN); the tests assert the 1e-4 bar and additionally
require zero differing pixels, so any drift is caught long before it reaches 1e-4.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = [
    ("egg", 256, 256), ("egg", 320, 180),
    ("clouds", 256, 144), ("clouds", 192, 108),
    ("raytracer", 256, 256), ("raytracer", 320, 180),
    ("atmosphere", 256, 144),
    ("sdf_ao", 256, 144),
    ("planet", 256, 144),
    ("vinyl", 256, 144), ("vinyl", 160, 160),
    ("clouds_best", 256, 144), ("clouds_best", 160, 160),
]
TIMES = [0.0, 0.37, 2.5]


@pytest.fixture(scope="module")
def renderer():
    import shaderbox_amd
    return shaderbox_amd.Renderer(0)


def compare(gpu, ref):
    """returns (max_abs_diff over non-NaN-matching channels, #pixels with any bit difference)"""
    both_nan = np.isnan(gpu) & np.isnan(ref)
    d = np.where(both_nan, 0.0, np.abs(gpu.astype(np.float64) - ref.astype(np.float64)))
    d = np.nan_to_num(d, nan=np.inf)
    bits = (gpu.view(np.uint32) != ref.view(np.uint32)) & ~both_nan
    return float(d.max()), int(bits.any(axis=-1).sum())


@pytest.mark.parametrize("app,w,h", CASES)
@pytest.mark.parametrize("t", TIMES)
def test_frame_matches_oracle(renderer, oracle, app, w, h, t):
    from oracle.oracle import APP_IDS
    ref = oracle.render(APP_IDS[app], w, h, t)
    gpu = renderer.render(app, w, h, t).cpu().numpy()
    assert gpu.shape == ref.shape == (h, w, 4)
    maxd, nbits = compare(gpu, ref)
    print("%s %dx%d t=%.2f: max|diff|=%.3g, pixels with differing bits: %d" % (app, w, h, t, maxd, nbits))
    assert maxd <= 1e-4, "north_star tolerance: 1e-4 per float channel"
    assert nbits == 0, "kernels are specified to be bit-identical to the oracle"


def test_mouse_and_aux(renderer, oracle):
    """run-time uniforms: u_mouse (camera orbit) and non-default aux blocks"""
    import shaderbox_amd
    from oracle.oracle import APP_CLOUDS, APP_RAYTRACER, APP_SDF_AO
    aux = shaderbox_amd.clouds_defaults()
    aux.cld_march_steps, aux.illum_march_steps, aux.cld_coverage = 40, 3, .6
    aux.sun_dir[0], aux.sun_dir[1], aux.sun_dir[2] = .0, .6, -.8
    aux.wind_dir[0] = .1
    ref = oracle.render(APP_CLOUDS, 160, 90, 1.25, mouse=(1.5, 0.0), aux=aux)
    gpu = renderer.render("clouds", 160, 90, 1.25, mouse=(1.5, 0.0), aux=aux).cpu().numpy()
    assert compare(gpu, ref) == (0.0, 0)
    ref = oracle.render(APP_RAYTRACER, 160, 120, 0.8, mouse=(300.0, 200.0))
    gpu = renderer.render("raytracer", 160, 120, 0.8, mouse=(300.0, 200.0)).cpu().numpy()
    assert compare(gpu, ref) == (0.0, 0)
    aux2 = shaderbox_amd.sdf_ao_defaults()
    aux2.fog_density, aux2.fog_falloff = .25, .3
    ref = oracle.render(APP_SDF_AO, 160, 90, 0.5, aux=aux2)
    gpu = renderer.render("sdf_ao", 160, 90, 0.5, aux=aux2).cpu().numpy()
    assert compare(gpu, ref) == (0.0, 0)


def test_strip_and_rank_invariance(renderer):
    """A frame rendered as strips, or as the cyclic row-blocks of N ranks + assemble, is bit-identical
    to the frame rendered in one launch (pixels are independent; SURVEY.md §4 'strip-invariance')."""
    import torch
    from shaderbox_amd import shard
    for app, w, h, t in [("clouds", 200, 117, .37), ("egg", 203, 95, .37), ("raytracer", 96, 64, .1)]:
        full = renderer.render(app, w, h, t)
        parts = [renderer.render(app, w, h, t, rows=(a, b)) for a, b in [(0, 13), (13, 14), (14, 100 if h > 100 else h), (100 if h > 100 else h, h)]]
        assert torch.equal(torch.cat(parts, 0).view(torch.int32), full.view(torch.int32))
        for nranks, br in [(2, 8), (8, 8), (3, 5), (8, 16)]:
            rmax = shard.rank_rows_max(h, br, nranks)
            slabs = torch.zeros((nranks, rmax, w, 4), dtype=torch.float32, device=full.device)
            for r in range(nranks):
                renderer.render_rank(app, w, h, t, br, r, nranks, out=slabs[r])
            frame = renderer.assemble(slabs, w, h, br, nranks)
            assert torch.equal(frame.view(torch.int32), full.view(torch.int32)), (app, nranks, br)
            # the same slabs produced in pieces (pipelined gather path)
            slabs2 = torch.zeros_like(slabs)
            for r in range(nranks):
                for a, b in [(0, br), (br, 3 * br), (3 * br, rmax)]:
                    renderer.render_rank_rows(app, w, h, t, br, r, nranks, a, min(b, rmax), slabs2[r])
            assert torch.equal(slabs2.view(torch.int32), slabs.view(torch.int32)), (app, nranks, br)


def test_device_math_matches_oracle(renderer, oracle):
    """the device statement of the math spec is bit-identical to the oracle's"""
    import torch
    rng = np.random.default_rng(7)
    ints = np.arange(-(1 << 21), (1 << 21) + 1, dtype=np.float32)
    for fn, a, b in [
        ("sin", ints, None), ("cos", ints[::7], None), ("hash", ints, None),
        ("sin", (rng.standard_normal(1 << 20) * 1000).astype(np.float32), None),
        ("cos", (rng.standard_normal(1 << 20) * 1000).astype(np.float32), None),
        ("tan", (rng.standard_normal(1 << 18) * 10).astype(np.float32), None),
        ("exp", np.concatenate([rng.uniform(-110, 95, 1 << 20), [np.inf, -np.inf, np.nan, 0.0]]).astype(np.float32), None),
        ("pow", np.abs(rng.standard_normal(1 << 20)).astype(np.float32) * 2, rng.choice([1 / 2.2, 1.5, 10, 30, 1500, 2.0, 0.0, -1.0], 1 << 20).astype(np.float32)),
        ("acos", rng.uniform(-1.2, 1.2, 1 << 18).astype(np.float32), None),
        ("atan2", rng.standard_normal(1 << 18).astype(np.float32), rng.standard_normal(1 << 18).astype(np.float32)),
    ]:
        got = renderer.math(fn, torch.from_numpy(a), None if b is None else torch.from_numpy(b)).cpu().numpy()
        if fn == "hash":
            s = oracle.math("sin", a)
            x = (s * np.float32(753.5453123)).astype(np.float32)
            want = (x - np.floor(x)).astype(np.float32)
        else:
            want = oracle.math(fn, a, b)
        same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
        assert same.all(), (fn, int((~same).sum()), a[~same][:5], got[~same][:5], want[~same][:5])


def test_error_codes(renderer):
    import shaderbox_amd
    with pytest.raises(shaderbox_amd.SbxError) as e:
        renderer.render(99, 64, 64, 0.0)
    assert e.value.code == shaderbox_amd.SBX_ERR_UNSUPPORTED
    with pytest.raises(shaderbox_amd.SbxError) as e:
        renderer.render("egg", 64, 64, 0.0, rows=(10, 80))
    assert e.value.code == shaderbox_amd.SBX_ERR_ARG
    assert renderer.render("egg", 64, 64, 0.0, rows=(5, 5)).shape[0] == 0


def test_clouds_cooperative_equals_perlane_full_size(renderer):
    """BASELINE size (3840x2160): the wave-cooperative CLOUDS kernel is bit-identical to the per-lane
    cross-check kernel, which the small-frame tests above tie to the oracle."""
    import torch
    for (w, h, t, mouse) in [(3840, 2160, 0.37, (0.0, 0.0)), (1920, 1080, 2.5, (0.0, 0.0)), (1283, 721, 100.0, (2.0, 0.0))]:
        renderer.set_variant(0)
        a = renderer.render("clouds", w, h, t, mouse=mouse)
        renderer.set_variant(1)
        b = renderer.render("clouds", w, h, t, mouse=mouse)
        renderer.set_variant(0)
        assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (w, h, t)


def test_clouds_4k_against_survey_pixels_and_oracle_rows(renderer, oracle):
    """BASELINE config C4 itself (APP_CLOUDS 3840x2160, t = .37): the GPU frame is checked against
    (a) the six pixel values and the frame mean SURVEY.md Appendix C lists for this very frame
        (verbatim reference headers over glibc libm) within the 1e-4 bar, and
    (b) the oracle on a spread of full-width rows, bit-for-bit."""
    from oracle.oracle import APP_CLOUDS
    W, H, T = 3840, 2160, 0.37
    img = renderer.render("clouds", W, H, T).cpu().numpy()
    survey = {(0, 0): (0.625376225, 0.843884051, 0.938976765), (1920, 1080): (0.630046427, 0.738662779, 0.866104126),
              (2400, 1500): (0.362470716, 0.552011669, 0.753409386), (100, 2100): (0.3802827, 0.577344954, 0.76764071),
              (3839, 2159): (0.835957706, 0.836371362, 0.836967945), (1920, 600): (0.921403289, 0.959852397, 1.00188839)}
    for (x, y), rgb in survey.items():
        assert np.max(np.abs(img[y, x, :3].astype(np.float64) - np.array(rgb))) <= 1e-4, (x, y, img[y, x])
    mean = img[..., :3].reshape(-1, 3).mean(0, dtype=np.float64)
    assert np.max(np.abs(mean - np.array([0.608542, 0.746136, 0.861747]))) <= 2e-6, mean
    rows = [0, 549, 550, 551, 700, 1080, 1500, 1999, 2159]
    ref = oracle.render_rows(APP_CLOUDS, W, H, T, rows)
    maxd, nbits = compare(img[rows], ref)
    assert maxd <= 1e-4 and nbits == 0, (maxd, nbits)


@pytest.mark.parametrize("app,w,h,rows", [("egg", 1920, 1080, [0, 300, 540, 700, 1079]),
                                          ("raytracer", 3840, 2160, [0, 500, 1080, 1600, 2159]),
                                          ("atmosphere", 7680, 4320, [0, 1000, 2160, 3000, 4319]),
                                          ("planet", 7680, 4320, [2160, 2600]),
                                          ("sdf_ao", 1920, 1080, [0, 400, 800, 1079]),
                                          ("vinyl", 1920, 1080, [100, 540, 900]),
                                          ("clouds_best", 3840, 2160, [1100, 1500, 2159])])
def test_full_size_rows_match_oracle(renderer, oracle, app, w, h, rows):
    """BASELINE.json config sizes (C2 EGG 1920x1080, C3 RAYTRACER 3840x2160, C5 7680x4320): full-width rows
    of the full-size frame, rendered as one-row strips on the GPU, equal the oracle bit-for-bit."""
    from oracle.oracle import APP_IDS
    ref = oracle.render_rows(APP_IDS[app], w, h, 0.37, rows)
    got = np.stack([renderer.render(app, w, h, 0.37, rows=(y, y + 1)).cpu().numpy()[0] for y in rows])
    maxd, nbits = compare(got, ref)
    print("%s %dx%d rows %s: max|diff|=%.3g differing pixels=%d" % (app, w, h, rows, maxd, nbits))
    assert maxd <= 1e-4 and nbits == 0


def test_noise_library_matches_oracle(renderer, oracle):
    """noise_iq / hash_w / noise_w / fbm_worley_tile as library functions, and the
    ddsvolgen volume; bit-for-bit against the oracle (which test_oracle_kat pins to Appendix C)."""
    import torch
    rng = np.random.default_rng(3)
    pts = np.concatenate([rng.uniform(-20, 20, (20000, 3)), rng.uniform(0, 1, (20000, 3)),
                          [[.1, .2, .3], [.77, .01, .5], [1, 2, 3], [0, 0, 0], [-7.25, 40.5, 113.125]]]).astype(np.float32)
    for fn, par in [("noise_iq", (0, 0, 0)), ("hash_w", (0, 0, 0)), ("noise_w", (4, 0, 0)), ("noise_w", (8, 0, 0)),
                    ("fbm_worley_tile", (2, 1, .5)), ("fbm_worley_tile", (4, 1, .5))]:
        got = renderer.noise(fn, torch.from_numpy(pts), par).cpu().numpy()
        want = oracle.noise(fn, pts, par)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (fn, par)
    vol = renderer.worley_volume(128)
    ref = oracle.worley_volume(128, 0, 2)
    assert np.array_equal(vol[:2].cpu().numpy().view(np.uint32), ref.view(np.uint32))
    ref = oracle.worley_volume(128, 77, 78)
    assert np.array_equal(vol[77:78].cpu().numpy().view(np.uint32), ref.view(np.uint32))
    assert abs(float(vol[0, 0, 0, 0]) - 1.35212326) < 2e-2          # Appendix C: ddsvolgen voxel (0,0,0)


@pytest.mark.parametrize("w,h", [(1, 1), (7, 3), (33, 9), (8, 64), (257, 2)])
def test_tiny_and_ragged_frames(renderer, oracle, w, h):
    """sizes that do not fill a wave tile / workgroup: every app, bit-for-bit"""
    from oracle.oracle import APP_IDS
    for app in ("egg", "clouds", "raytracer", "atmosphere", "sdf_ao", "planet", "vinyl", "clouds_best"):
        ref = oracle.render(APP_IDS[app], w, h, 0.37, threads=4)
        gpu = renderer.render(app, w, h, 0.37).cpu().numpy()
        assert compare(gpu, ref) == (0.0, 0), (app, w, h)


def test_late_time_and_mouse_against_oracle(renderer, oracle):
    """large lattice indices (wind offset at t = 100: |n| up to ~9e4) and an orbiting camera, vs the oracle"""
    from oracle.oracle import APP_CLOUDS, APP_PLANET
    for t, mouse in [(100.0, (0.0, 0.0)), (37.5, (2.5, 0.0))]:
        ref = oracle.render(APP_CLOUDS, 160, 90, t, mouse=mouse)
        gpu = renderer.render("clouds", 160, 90, t, mouse=mouse).cpu().numpy()
        assert compare(gpu, ref) == (0.0, 0), (t, mouse)
    ref = oracle.render(APP_PLANET, 128, 72, 41.0)
    gpu = renderer.render("planet", 128, 72, 41.0).cpu().numpy()
    assert compare(gpu, ref) == (0.0, 0)


def test_division_by_reciprocal_is_exact(renderer):
    """sbx_math.h div_by(): (float)((double)n * RN64(1/d)) is the IEEE binary32 quotient n/d for every n, d —
    checked on the device: all 2^24 significand patterns x several exponents against the denominators the
    kernels use, random pairs over the whole range, and the special values."""
    import torch
    rng = np.random.default_rng(5)
    dens = np.array([.0135, 7994.0, 1200.0, .4, .055, .0335, .65, .15, .3, 1.0 - .465], dtype=np.float32)
    dens = np.concatenate([dens, np.float32(1.0 - .535) + np.float32(.0135) - np.float32(1.0 - .535) * np.ones(1, np.float32)])
    mant = np.arange(1 << 24, dtype=np.uint32)
    for d in dens:
        for e in (100, 126, 127, 140):                     # n = 1.m * 2^(e-127)
            n = ((np.uint32(e) << 23) | (mant & 0x7fffff)).view(np.float32)
            n = np.where(mant & 0x800000, -n, n).astype(np.float32)
            a = torch.from_numpy(n)
            b = torch.full_like(a, float(d))
            q0 = renderer.math("div", a, b)
            q1 = renderer.math("div_rd", a, b)
            assert torch.equal(q0.view(torch.int32), q1.view(torch.int32)), (d, e)
    bits = rng.integers(0, 1 << 32, size=(2, 1 << 23), dtype=np.uint64).astype(np.uint32)
    a, b = torch.from_numpy(bits[0].view(np.float32).copy()), torch.from_numpy(bits[1].view(np.float32).copy())
    sp = torch.tensor([0.0, -0.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 3.4e38, 1.17549435e-38, 1.0], dtype=torch.float32)
    a = torch.cat([a, sp.repeat_interleave(len(sp))]); b = torch.cat([b, sp.repeat(len(sp))])
    q0 = renderer.math("div", a, b).cpu().numpy(); q1 = renderer.math("div_rd", a, b).cpu().numpy()
    same = (q0.view(np.uint32) == q1.view(np.uint32)) | (np.isnan(q0) & np.isnan(q1))
    assert same.all(), (a.numpy()[~same][:5], b.numpy()[~same][:5], q0[~same][:5], q1[~same][:5])


def test_exp_table_vs_horner(renderer):
    """exp (table form: the spec, oracle m_exp) against the former 13-term form on ALL 2^32 binary32 inputs:
    equal everywhere except x = -89.45233 (0xc2b2e798), whose exact result 1011149.5000000046 * 2^-149 sits
    4.6e-9 of a denormal ulp above a tie: the table form returns the correctly rounded 0x000f6dce, the 13-term
    form 0x000f6dcd.  NaN inputs give NaN in both (the binary32 clamp lets them through)."""
    import torch
    step = 1 << 26
    odd = []
    for lo in range(0, 1 << 32, step):
        bits = torch.arange(lo, lo + step, dtype=torch.int64, device="cuda").to(torch.int32)   # wraps to all patterns
        x = bits.view(torch.float32)
        a = renderer.math("exp", x)
        b = renderer.math("exp_h13", x)
        same = (a.view(torch.int32) == b.view(torch.int32)) | (torch.isnan(a) & torch.isnan(b))
        assert bool((torch.isnan(a) == torch.isnan(x)).all()), lo
        if not bool(same.all()):
            d = ~same
            odd += [(int(p) & 0xffffffff, int(q) & 0xffffffff, int(r) & 0xffffffff) for p, q, r in
                    zip(bits[d].cpu().tolist(), a[d].view(torch.int32).cpu().tolist(), b[d].view(torch.int32).cpu().tolist())]
    assert odd == [(0xc2b2e798, 0x000f6dce, 0x000f6dcd)], odd


def test_per_pixel_main_image_entry(renderer):
    """sbx_main_image — the reference's own per-pixel entry (src/main.h:6-9) — returns the pixels of the frame,
    re-renders when app / uniforms / aux change, clamps coordinates outside the frame."""
    import shaderbox_amd
    W, H = 96, 54
    frame = renderer.render("clouds", W, H, 0.37).cpu().numpy()
    for (x, y) in [(0, 0), (95, 53), (17, 40), (50, 27)]:
        px = np.array(renderer.main_image("clouds", W, H, 0.37, (x + .5, y + .5)), np.float32)
        assert (px.view(np.uint32) == frame[y, x].view(np.uint32)).all()
    frame2 = renderer.render("clouds", W, H, 2.5).cpu().numpy()
    px = np.array(renderer.main_image("clouds", W, H, 2.5, (50.5, 40.5)), np.float32)
    assert (px.view(np.uint32) == frame2[40, 50].view(np.uint32)).all()
    aux = shaderbox_amd.clouds_defaults(renderer.lib)           # an aux block is part of the cache key
    aux.cld_coverage = 0.6
    frame3 = renderer.render("clouds", W, H, 2.5, aux=aux).cpu().numpy()
    px = np.array(renderer.main_image("clouds", W, H, 2.5, (50.5, 40.5), aux=aux), np.float32)
    assert (px.view(np.uint32) == frame3[40, 50].view(np.uint32)).all() and not (frame3[40, 50] == frame2[40, 50]).all()
    egg = renderer.render("egg", 64, 64, 0.0).cpu().numpy()
    px = np.array(renderer.main_image("egg", 64, 64, 0.0, (-3.0, 1000.0)), np.float32)      # clamped to (0, 63)
    assert (px.view(np.uint32) == egg[63, 0].view(np.uint32)).all()
    with pytest.raises(shaderbox_amd.SbxError):
        renderer.main_image(99, 64, 64, 0.0, (.5, .5))


@pytest.mark.parametrize("app,w,h", [("egg", 200, 112), ("sdf_ao", 200, 112), ("vinyl", 200, 112), ("planet", 160, 90), ("clouds", 160, 90)])
def test_exact_skips_over_many_times(renderer, oracle, app, w, h):
    """The kernels leave work out where they can prove it cannot change the result (EGG/SDF_AO: members of the
    union behind a bounding volume; PLANET: noise outside the cloud band / below a smoothstep edge; CLOUDS: octaves of
    samples that cannot be lit).  The bounds move with the frame (feet, knees, sun, wind), so sweep the time."""
    from oracle.oracle import APP_IDS
    for t in (0.1, 0.77, 1.3, 3.9, 7.7, 12.34, 31.0):
        ref = oracle.render(APP_IDS[app], w, h, t)
        gpu = renderer.render(app, w, h, t).cpu().numpy()
        maxd, nbits = compare(gpu, ref)
        assert nbits == 0 and maxd == 0.0, (app, t, maxd, nbits)


def test_clouds_aux_corners(renderer, oracle):
    """APP_CLOUDS run-time parameters that switch code paths: more march steps than the per-frame y table holds
    (general main sample), a wind with a y component (table key), coverage extremes (the stage bounds of the main
    sample compare against 1 - coverage), a sun that is not along -z (general light march)."""
    import shaderbox_amd
    from oracle.oracle import APP_CLOUDS
    W, H = 96, 54
    cases = []
    a = shaderbox_amd.clouds_defaults(); a.cld_march_steps = 1100; a.cld_thick = 137.5; cases.append((a, 0.37))
    a = shaderbox_amd.clouds_defaults(); a.cld_march_steps = 4200; a.cld_thick = 137.5; cases.append((a, 0.37))   # beyond the y table
    a = shaderbox_amd.clouds_defaults(); a.cld_march_steps = 4100; a.sun_dir[0], a.sun_dir[2] = .6, -.8; cases.append((a, 1.1))
    a = shaderbox_amd.clouds_defaults(); a.wind_dir[1] = .05; a.wind_dir[0] = -.1; cases.append((a, 1.7))
    a = shaderbox_amd.clouds_defaults(); a.cld_coverage = .95; cases.append((a, 0.37))
    a = shaderbox_amd.clouds_defaults(); a.cld_coverage = .05; cases.append((a, 0.37))
    a = shaderbox_amd.clouds_defaults(); a.cld_coverage = 1.5; a.sigma_scattering = .4; cases.append((a, 2.5))
    a = shaderbox_amd.clouds_defaults(); a.sun_dir[0], a.sun_dir[1], a.sun_dir[2] = .3, .2, -.9; a.illum_march_steps = 9; cases.append((a, 0.9))
    a = shaderbox_amd.clouds_defaults(); a.illum_march_steps = 0; cases.append((a, 0.37))
    for aux, t in cases:
        ref = oracle.render(APP_CLOUDS, W, H, t, aux=aux)
        gpu = renderer.render("clouds", W, H, t, aux=aux).cpu().numpy()
        assert compare(gpu, ref) == (0.0, 0), (t, aux.cld_march_steps, aux.cld_coverage)


@pytest.mark.timeout(300)
def test_non_finite_uniforms_do_not_hang(renderer):
    """NaN / inf / huge u_time and u_mouse flow through as data (the reference has no error path); in particular the
    cooperative miss loops of the hash cache must terminate when lattice indices are NaN or inf."""
    import torch
    for app in ("clouds", "planet", "vinyl", "egg", "raytracer", "atmosphere", "sdf_ao", "clouds_best"):
        for t in (float("nan"), float("inf"), 1e30, -1e9, 1e-30):
            for mouse in ((0.0, 0.0), (float("nan"), 5.0), (1e20, -3.0)):
                f = renderer.render(app, 96, 54, t, mouse=mouse)
                torch.cuda.synchronize()
                assert tuple(f.shape) == (54, 96, 4)


def test_clouds_random_sweep_default_equals_perlane(renderer):
    """The shipped k_clouds (hash cache, y table, z-only light march, staged main sample, no-lookup light samples)
    against the plain per-lane kernel (sbx_set_variant 1: none of those) on random uniforms and aux blocks: identical
    bits.  The per-lane kernel itself is checked against the oracle by the other tests."""
    import shaderbox_amd
    import torch
    rng = np.random.default_rng(123)
    try:
        for i in range(120):
            aux = shaderbox_amd.clouds_defaults(renderer.lib)
            t = float(rng.uniform(0, 50)) if i % 3 else float(rng.uniform(0, 3))
            mouse = (float(rng.uniform(0, 6.3)), 0.0) if i % 2 else (0.0, 0.0)
            aux.cld_coverage = float(rng.uniform(0.2, 0.9))
            aux.cld_march_steps = int(rng.integers(10, 160))
            aux.illum_march_steps = int(rng.integers(0, 9))
            aux.cld_thick = float(rng.uniform(40, 300))
            aux.sigma_scattering = float(rng.uniform(.02, .6))
            if i % 5 == 0:
                aux.wind_dir[0], aux.wind_dir[1], aux.wind_dir[2] = [float(x) for x in rng.uniform(-.3, .3, 3)]
            if i % 7 == 0:
                d = rng.standard_normal(3); d /= np.linalg.norm(d)
                aux.sun_dir[0], aux.sun_dir[1], aux.sun_dir[2] = [float(x) for x in d]
            W, H = (640, 360) if i % 4 else (333, 187)
            renderer.set_variant(0); a = renderer.render("clouds", W, H, t, mouse=mouse, aux=aux).clone()
            renderer.set_variant(1); b = renderer.render("clouds", W, H, t, mouse=mouse, aux=aux)
            same = (a.view(torch.int32) == b.view(torch.int32)) | (torch.isnan(a) & torch.isnan(b))
            assert bool(same.all()), (i, t, mouse, aux.cld_coverage, aux.cld_march_steps, int((~same).sum()))
    finally:
        renderer.set_variant(0)


@pytest.mark.parametrize("app", ["egg", "sdf_ao", "vinyl", "planet"])
def test_random_sweep_culled_equals_plain(renderer, app):
    """Kernels with exact culling / skips (default) against the same kernels with every member and every octave evaluated
    (sbx_set_variant 1) on random times and mouse positions: identical bits."""
    import torch
    rng = np.random.default_rng(7)
    try:
        for i in range(100):
            t = float(rng.uniform(0, 60)) if i % 3 else float(rng.uniform(0, 3))
            mouse = (float(rng.uniform(0, 640)), float(rng.uniform(0, 360))) if i % 2 else (0.0, 0.0)
            W, H = (640, 360) if i % 4 else (333, 187)
            if app == "planet":
                W, H = (320, 180) if i % 4 else (201, 113)
            renderer.set_variant(0); a = renderer.render(app, W, H, t, mouse=mouse).clone()
            renderer.set_variant(1); b = renderer.render(app, W, H, t, mouse=mouse)
            same = (a.view(torch.int32) == b.view(torch.int32)) | (torch.isnan(a) & torch.isnan(b))
            assert bool(same.all()), (app, i, t, mouse, int((~same).sum()))
    finally:
        renderer.set_variant(0)


@pytest.mark.parametrize("app,w,h", [("egg", 1920, 1080), ("egg", 3840, 2160), ("sdf_ao", 3840, 2160), ("vinyl", 3840, 2160),
                                      ("planet", 7680, 4320)])
def test_full_size_default_equals_plain(renderer, app, w, h):
    """BASELINE.json frame sizes: the default kernels (exact culling / skips) and the plain ones agree on every pixel."""
    import torch
    try:
        for t in (0.37, 2.5):
            renderer.set_variant(0); a = renderer.render(app, w, h, t).clone()
            renderer.set_variant(1); b = renderer.render(app, w, h, t)
            same = (a.view(torch.int32) == b.view(torch.int32)) | (torch.isnan(a) & torch.isnan(b))
            assert bool(same.all()), (app, t, int((~same).sum()))
            del a, b
    finally:
        renderer.set_variant(0)


def test_library_before_torch_in_a_fresh_process():
    """build() followed by smoke() in one interpreter: the library is asked for before anything imported torch.
    (Two HIP runtimes in one process made sbx_create report 'no device'; load_library() now imports torch first.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); import shaderbox_amd; lib = shaderbox_amd.load_library(); "
            "r = shaderbox_amd.Renderer(0); f = r.render('egg', 32, 32, 0.0); print(tuple(f.shape))" % root)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "(32, 32, 4)" in out.stdout, out.stderr[-2000:]


def test_pack_unorm8(renderer):
    """sbx_pack_unorm8 = Direct3D float -> R8G8B8A8_UNORM (NaN -> 0, clamp, * 255 + .5, truncate), optional row flip"""
    import torch
    rng = np.random.default_rng(9)
    a = rng.uniform(-.2, 1.2, size=(37, 53, 4)).astype(np.float32)
    a[0, 0] = [np.nan, -0.0, 1.0, np.inf]; a[1, 1] = [0.5 / 255, 1.5 / 255, 254.5 / 255, -np.inf]
    a[2, 2] = [0.49999 / 255, 0.50001 / 255, 0.999999, 1e-30]
    want = np.where(np.isnan(a) | ~(a > 0), 0, np.minimum(a, 1)).astype(np.float32)
    want = (want * np.float32(255) + np.float32(.5)).astype(np.uint8)
    t = torch.from_numpy(a).cuda()
    assert (renderer.pack_unorm8(t, flip_y=False).cpu().numpy() == want).all()
    assert (renderer.pack_unorm8(t, flip_y=True).cpu().numpy() == want[::-1]).all()
    f = renderer.render("egg", 64, 48, 0.37)
    p = renderer.pack_unorm8(f).cpu().numpy()
    assert p.shape == (48, 64, 4) and (p[..., 3] == 255).all()


def test_pow_of_two_is_exact(renderer, oracle):
    """kern_sdf_ao.hip replaces 1 / pow(2, k), k = 1..5, by the exact power of two: the math spec's pow gives exactly 2^k"""
    import torch
    k = np.arange(1, 6, dtype=np.float32)
    two = np.full(5, 2.0, np.float32)
    assert oracle.math("pow", two, k).tolist() == [2.0, 4.0, 8.0, 16.0, 32.0]
    got = renderer.math("pow", torch.from_numpy(two).cuda(), torch.from_numpy(k).cuda()).cpu().numpy()
    assert got.tolist() == [2.0, 4.0, 8.0, 16.0, 32.0]


def test_pow_table_vs_series(renderer, oracle):
    """pow (table forms of log2 and 2^t: the spec, oracle m_pow) against the former atanh-series / 13-term form: both
    are binary64-internal and accurate to ~1e-16, so they can only differ where the exact value sits that close to a
    binary32 rounding boundary: a tiny fraction of inputs, by one ulp.  Device pow == oracle pow bit for bit."""
    import torch
    rng = np.random.default_rng(17)
    n = 1 << 24
    tot = diff = 0
    for y in (1 / 2.2, 1.5, 10.0, 30.0, 50.0, 1500.0):
        x = torch.from_numpy(np.concatenate([rng.uniform(0, 1.5, n // 2), rng.uniform(0.98, 1.02, n // 4),
                                             np.abs(rng.standard_normal(n // 4)) * 4]).astype(np.float32)).cuda()
        yy = torch.full_like(x, float(np.float32(y)))
        a = renderer.math("pow", x, yy)
        b = renderer.math("pow_h", x, yy)
        d = (a.view(torch.int32) - b.view(torch.int32)).abs()
        d = torch.where(torch.isnan(a) & torch.isnan(b), torch.zeros_like(d), d)
        assert int(d.max()) <= 1, y
        tot += x.numel(); diff += int((d != 0).sum())
        xs = x[:200000].cpu().numpy()
        want = oracle.math("pow", xs, np.full_like(xs, np.float32(y)))
        got = a[:200000].cpu().numpy()
        assert ((got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))).all(), y
    print("pow table vs series: %d of %d inputs differ (by one ulp)" % (diff, tot))
    assert diff <= tot * 1e-6


def test_two_contexts_in_two_threads(renderer, oracle):
    """One caller per context is the rule; two contexts used from two threads on two streams at once must not disturb
    each other (the reference's C++ globals are thread_local for the same reason, src/def.h:7-8)."""
    import threading
    import shaderbox_amd
    import torch
    from oracle.oracle import APP_IDS
    want = {app: oracle.render(APP_IDS[app], 160, 90, 0.37) for app in ("clouds", "planet")}
    errs = []

    def work(app):
        try:
            r = shaderbox_amd.Renderer(0)
            s = torch.cuda.Stream()
            for _ in range(20):
                with torch.cuda.stream(s):
                    f = r.render(app, 160, 90, 0.37)
                s.synchronize()
                g = f.cpu().numpy()
                if compare(g, want[app]) != (0.0, 0):
                    errs.append(app)
        except Exception as e:      # noqa: BLE001
            errs.append(repr(e))
    ts = [threading.Thread(target=work, args=(a,)) for a in ("clouds", "planet")]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs


def test_render_is_graph_capturable(renderer):
    """sbx_render_rows only enqueues work on the given stream, so a host may capture frames into a HIP graph
    (here through torch.cuda.graph) and replay them: same pixels."""
    import torch
    for app, w, h in (("egg", 128, 128), ("clouds", 160, 90)):
        out = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
        ref = renderer.render(app, w, h, 0.37).clone()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            renderer.render(app, w, h, 0.37, out=out)
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(4):
                renderer.render(app, w, h, 0.37, out=out)
        out.zero_(); torch.cuda.synchronize()
        g.replay(); torch.cuda.synchronize()
        assert torch.equal(out.view(torch.int32), ref.view(torch.int32)), app


def test_root_relief_split_invariance(renderer):
    """The split that deals the gather's root fewer row-blocks (sbx_render_split / sbx_assemble_split): N ranks
    emulated on one GPU, assembled, equal the frame rendered in one launch — also at the BASELINE frame size."""
    import torch
    from shaderbox_amd import shard
    cases = [("clouds", 200, 117, 8, 8, 7, 8), ("clouds", 200, 117, 3, 2, 1, 3), ("egg", 203, 95, 4, 4, 0, 2),
             ("planet", 160, 90, 8, 8, 2, 5), ("clouds", 3840, 2160, 8, 8, 7, 8)]
    for app, w, h, nranks, br, m0, m in cases:
        full = renderer.render(app, w, h, .37)
        rmax = shard.rank_rows_max(h, br, nranks, m0, m)
        slabs = torch.zeros((nranks, rmax, w, 4), dtype=torch.float32, device=full.device)
        for r in range(nranks):
            renderer.render_rank(app, w, h, .37, br, r, nranks, out=slabs[r], root_rounds=m0, rounds=m)
        frame = renderer.assemble(slabs, w, h, br, nranks, root_rounds=m0, rounds=m)
        assert torch.equal(frame.view(torch.int32), full.view(torch.int32)), (app, w, nranks, br, m0, m)
        if w <= 256:                 # slab produced in pieces
            slabs2 = torch.zeros_like(slabs)
            for r in range(nranks):
                for a, b in [(0, br), (br, 3 * br), (3 * br, rmax)]:
                    renderer.render_rank_rows(app, w, h, .37, br, r, nranks, a, min(b, rmax), slabs2[r], root_rounds=m0, rounds=m)
            assert torch.equal(slabs2.view(torch.int32), slabs.view(torch.int32))
        del full, slabs, frame
