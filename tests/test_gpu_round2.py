"""GPU tests added in round 2: context-state hazards of APP_CLOUDS' y table, the USE_NOISE_TEX build of
APP_CLOUDS, per-stream timing, and wider full-size oracle coverage."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def compare(gpu, ref):
    both_nan = np.isnan(gpu) & np.isnan(ref)
    d = np.where(both_nan, 0.0, np.abs(gpu.astype(np.float64) - ref.astype(np.float64)))
    d = np.nan_to_num(d, nan=np.inf)
    bits = (gpu.view(np.uint32) != ref.view(np.uint32)) & ~both_nan
    return float(d.max()), int(bits.any(axis=-1).sum())


@pytest.fixture(scope="module")
def renderer():
    import shaderbox_amd
    return shaderbox_amd.Renderer(0)


# ---------------------------------------------------------------------------------------------------------
# y-table state of a context
# ---------------------------------------------------------------------------------------------------------
def test_perlane_variant_first_then_default(oracle):
    """A fresh context that renders APP_CLOUDS with the per-lane kernel first (which builds no y table) and the default
    kernel afterwards with the same key must build the table then (round 1 marked it valid without building it)."""
    import shaderbox_amd
    from oracle.oracle import APP_CLOUDS
    r = shaderbox_amd.Renderer(0)
    ref = oracle.render(APP_CLOUDS, 160, 90, .37)
    r.set_variant(1)
    a = r.render("clouds", 160, 90, .37).cpu().numpy()
    r.set_variant(0)
    b = r.render("clouds", 160, 90, .37).cpu().numpy()
    assert compare(a, ref) == (0.0, 0)
    assert compare(b, ref) == (0.0, 0)
    r.close()


def test_first_clouds_frame_inside_a_capture(oracle):
    """The FIRST APP_CLOUDS render of a fresh context happens inside a stream capture: nothing has executed, so the
    context must not believe its table exists.  Eager renders afterwards (same key, then nine other keys to cycle the eager
    ring) and replays of the graph all give the oracle's pixels."""
    import torch
    import shaderbox_amd
    from oracle.oracle import APP_CLOUDS
    r = shaderbox_amd.Renderer(0)
    r.set_timing(True)                       # timing events must stay out of the capture
    w, h = 160, 90
    ref = oracle.render(APP_CLOUDS, w, h, .37)
    out = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        torch.zeros(1, device="cuda")        # the stream exists before the capture
    s.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        r.render("clouds", w, h, .37, out=out)
    eager = r.render("clouds", w, h, .37).cpu().numpy()           # same key, eager, BEFORE any replay
    assert compare(eager, ref) == (0.0, 0)
    g.replay(); torch.cuda.synchronize()
    assert compare(out.cpu().numpy(), ref) == (0.0, 0)
    aux = shaderbox_amd.clouds_defaults()
    for k in range(9):                                            # nine other keys: the eager ring wraps
        aux.cld_thick = 100.0 + 3.0 * k
        r.render("clouds", 64, 36, .37, aux=aux)
    out.zero_(); torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    assert compare(out.cpu().numpy(), ref) == (0.0, 0)
    assert r.last_kernel_ms() > 0.0                               # timing pairs come from eager launches only
    r.close()


def test_table_ring_reuse_across_streams(oracle):
    """Key changes on every frame (animated cld_thick) over two streams, more rebuilds than ring slots, large frames in
    flight: a rebuild into a reused slot must wait for the launches still reading it.  Every frame equals the per-lane
    kernel's (which uses no table)."""
    import torch
    import shaderbox_amd
    r = shaderbox_amd.Renderer(0)
    w, h = 1280, 720
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    aux = shaderbox_amd.clouds_defaults()
    n = 20
    frames = [torch.empty((h, w, 4), dtype=torch.float32, device="cuda") for _ in range(n)]
    for i in range(n):
        aux.cld_thick = 90.0 + 2.5 * i
        with torch.cuda.stream(streams[i % 2]):
            r.render("clouds", w, h, .37, aux=aux, out=frames[i])
    torch.cuda.synchronize()
    r.set_variant(1)
    for i in range(0, n, 3):
        aux.cld_thick = 90.0 + 2.5 * i
        ref = r.render("clouds", w, h, .37, aux=aux)
        assert torch.equal(ref.view(torch.int32), frames[i].view(torch.int32)), i
    r.close()


def test_timing_is_per_stream(renderer):
    """sbx_last_kernel_ms pairs the events of ONE launch even with launches in flight on two streams; an argument error
    records nothing."""
    import torch
    import shaderbox_amd
    r = shaderbox_amd.Renderer(0)
    r.set_timing(True)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    a = torch.empty((1080, 1920, 4), dtype=torch.float32, device="cuda")
    b = torch.empty((36, 64, 4), dtype=torch.float32, device="cuda")
    # (durations of tiny launches on an idle GPU are dominated by the clock ramp, so only validity is checked: a pair that
    # straddled two launches on two streams would be negative, or fail in hipEventElapsedTime)
    for _ in range(3):
        with torch.cuda.stream(s1):
            r.render("clouds", 1920, 1080, .37, out=a)
        with torch.cuda.stream(s2):
            r.render("clouds", 64, 36, .37, out=b)
    small = r.last_kernel_ms()               # the pair of the LAST timed launch: the small frame on s2
    assert 0.0 < small < 100.0
    with torch.cuda.stream(s1):
        r.render("clouds", 1920, 1080, .37, out=a)
    big = r.last_kernel_ms()
    assert 0.0 < big < 100.0
    aux = shaderbox_amd.clouds_defaults()
    aux.cld_march_steps = -1
    with pytest.raises(shaderbox_amd.SbxError):
        r.render("clouds", 64, 36, .37, aux=aux)
    assert abs(r.last_kernel_ms() - big) < 1e-6                   # the failed call left the last pair alone
    r.close()


# ---------------------------------------------------------------------------------------------------------
# USE_NOISE_TEX build of APP_CLOUDS
# ---------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def volumes(renderer):
    """two different baked volumes (ddsvolgen's tiled-Worley fBm at two sizes), device + host copies"""
    v1 = renderer.worley_volume(32)
    v2 = renderer.worley_volume(16)
    return v1, v2, v1.cpu().numpy(), v2.cpu().numpy()


def test_texture_filter_matches_oracle(renderer, oracle, volumes):
    import torch
    v1, v2, h1, h2 = volumes
    rng = np.random.default_rng(7)
    pts = np.concatenate([rng.uniform(-3, 3, (20000, 3)), rng.uniform(-1e4, 1e4, (2000, 3)),
                          np.array([[0, 0, 0], [1, 1, 1], [.5 / 32, .5 / 32, .5 / 32], [1 - .5 / 32, .25, .75], [-1e-7, 0, 0],
                                    [np.nan, 0, 0], [np.inf, .1, .2], [1e30, -1e30, 3e38]])]).astype(np.float32)
    for dev, host in ((v1, h1), (v2, h2)):
        gpu = renderer.tex3d(dev, torch.from_numpy(pts)).cpu().numpy()
        ref = oracle.tex3d(host, pts)
        both_nan = np.isnan(gpu) & np.isnan(ref)                  # non-finite coordinates: NaN on both sides (sign is not data)
        assert np.array_equal(gpu.view(np.uint32)[~both_nan], ref.view(np.uint32)[~both_nan])
        assert both_nan.sum() <= 3
    # texel centres reproduce the texels
    c = (np.stack(np.meshgrid(np.arange(32), np.arange(32), np.arange(32), indexing="ij"), -1).reshape(-1, 3)[:, ::-1] + .5) / 32
    got = renderer.tex3d(v1, torch.from_numpy(c.astype(np.float32))).cpu().numpy().reshape(32, 32, 32)
    assert np.array_equal(got, h1[..., 0])


@pytest.mark.parametrize("w,h,t", [(192, 108, 0.0), (256, 144, .37), (160, 160, 2.5)])
def test_clouds_tex_matches_oracle(renderer, oracle, volumes, w, h, t):
    from oracle.oracle import APP_CLOUDS_TEX
    v1, v2, h1, h2 = volumes
    renderer.set_noise_volumes(v1, v2)
    oracle.set_noise_volumes(h1, h2)
    ref = oracle.render(APP_CLOUDS_TEX, w, h, t)
    gpu = renderer.render("clouds_tex", w, h, t).cpu().numpy()
    maxd, nbits = compare(gpu, ref)
    assert maxd <= 1e-4 and nbits == 0
    assert np.isfinite(gpu).all() and gpu[h - 1].std() > 0        # clouds, not a flat sky


def test_clouds_tex_aux_and_errors(renderer, oracle, volumes):
    import shaderbox_amd
    from oracle.oracle import APP_CLOUDS_TEX
    v1, v2, h1, h2 = volumes
    fresh = shaderbox_amd.Renderer(0)
    with pytest.raises(shaderbox_amd.SbxError):                   # no volumes bound
        fresh.render("clouds_tex", 64, 36, .37)
    fresh.close()
    renderer.set_noise_volumes(v2, v1)                            # swapped roles, other sizes
    oracle.set_noise_volumes(h2, h1)
    aux = shaderbox_amd.clouds_defaults()
    aux.cld_march_steps, aux.illum_march_steps, aux.cld_coverage, aux.cld_thick = 37, 4, .7, 90.0
    aux.sun_dir[0], aux.sun_dir[1], aux.sun_dir[2] = .3, .5, -.8
    aux.wind_dir[0], aux.wind_dir[1] = .1, .05
    ref = oracle.render(APP_CLOUDS_TEX, 160, 90, 1.25, mouse=(1.5, 0.0), aux=aux)
    gpu = renderer.render("clouds_tex", 160, 90, 1.25, mouse=(1.5, 0.0), aux=aux).cpu().numpy()
    assert compare(gpu, ref) == (0.0, 0)
    # coordinates beyond the power-of-two fast path's range (|c| >= 2^30 / size): the wave falls back to the general wrap
    aux.wind_dir[0], aux.wind_dir[1] = 1e8, 0.0
    ref = oracle.render(APP_CLOUDS_TEX, 96, 54, 1.0, aux=aux)
    gpu = renderer.render("clouds_tex", 96, 54, 1.0, aux=aux).cpu().numpy()
    assert compare(gpu, ref) == (0.0, 0)
    # volumes whose size is not a power of two (any data is a volume): the general wrap for every sample
    c1, c2 = v1[:24, :24, :24].contiguous(), v2[:12, :12, :12].contiguous()
    renderer.set_noise_volumes(c1, c2)
    oracle.set_noise_volumes(c1.cpu().numpy(), c2.cpu().numpy())
    shaderbox_amd.load_library().sbx_aux_clouds_defaults(aux)
    for t in (0.0, .37):
        ref = oracle.render(APP_CLOUDS_TEX, 160, 90, t, aux=aux)
        gpu = renderer.render("clouds_tex", 160, 90, t, aux=aux).cpu().numpy()
        assert compare(gpu, ref) == (0.0, 0)


def test_clouds_tex_full_size_rows(renderer, oracle):
    """3840x2160 with two 128^3 volumes (the size ddsvolgen bakes): evenly spread full rows against the oracle."""
    from oracle.oracle import APP_CLOUDS_TEX
    v = renderer.worley_volume(128)
    v2 = renderer.worley_volume(64)
    renderer.set_noise_volumes(v, v2)
    oracle.set_noise_volumes(v.cpu().numpy(), v2.cpu().numpy())
    W, H = 3840, 2160
    gpu = renderer.render("clouds_tex", W, H, .37)
    rows = list(range(540, H, 100)) + [H - 1]
    ref = oracle.render_rows(APP_CLOUDS_TEX, W, H, .37, rows)
    got = gpu[rows].cpu().numpy()
    assert compare(got, ref) == (0.0, 0)


# ---------------------------------------------------------------------------------------------------------
# wider full-size oracle coverage for the apps that have no second kernel variant
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("app,w,h,nrows", [("raytracer", 3840, 2160, 64), ("atmosphere", 7680, 4320, 64),
                                           ("clouds_best", 3840, 2160, 64)])
def test_full_size_many_rows_match_oracle(renderer, oracle, app, w, h, nrows):
    from oracle.oracle import APP_IDS
    rows = [int(round(i * (h - 1) / (nrows - 1))) for i in range(nrows)]
    gpu = renderer.render(app, w, h, .37)
    got = gpu[rows].cpu().numpy()
    ref = oracle.render_rows(APP_IDS[app], w, h, .37, rows)
    maxd, nbits = compare(got, ref)
    print("%s %dx%d: %d rows, max|diff| %.3g, differing pixels %d" % (app, w, h, nrows, maxd, nbits))
    assert maxd <= 1e-4 and nbits == 0


# ---------------------------------------------------------------------------------------------------------
# multi-GPU frames inside the library (sbx_multi_*): N ranks emulated on one GPU
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("nranks", [1, 2, 3, 8])
def test_library_multi_gpu_frame_equals_single_gpu(renderer, nranks):
    """sbx_multi with every rank on device 0 (transfers are device copies instead of RCCL send/recv): rank 0 renders in
    place, the peers' row-blocks land in their final rows — same bits as one launch, also with root relief, ragged sizes,
    two frames in flight and at the BASELINE frame size."""
    import torch
    import shaderbox_amd
    m = shaderbox_amd.MultiRenderer([0] * nranks)
    assert not m.uses_rccl
    cases = [("clouds", 200, 117, .37), ("egg", 203, 95, .37), ("planet", 160, 90, .37), ("raytracer", 96, 7, .1)]
    if nranks in (2, 8):
        cases.append(("clouds", 3840, 2160, .37))
    for app, w, h, t in cases:
        full = renderer.render(app, w, h, t)
        for split in [(8, 1, 1)] + ([(8, 3, 4), (4, 0, 2)] if nranks > 1 else []):
            m.set_split(*split)
            for mode in ("slabs", "blocks"):           # one transfer per peer + scatter kernel / one per row-block in place
                m.set_exchange(mode)
                got = m.render(app, w, h, t)
                torch.cuda.synchronize()
                assert torch.equal(got.view(torch.int32), full.view(torch.int32)), (app, w, h, nranks, split, mode)
    m.set_exchange("slabs")
    # two frames in flight on two streams, different times
    m.set_split(8, 1, 1)
    s = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [torch.zeros((180, 320, 4), dtype=torch.float32, device="cuda") for _ in range(6)]
    for i in range(6):
        with torch.cuda.stream(s[i % 2]):
            m.render("clouds", 320, 180, .1 * i, out=outs[i])
    torch.cuda.synchronize()
    for i in range(6):
        ref = renderer.render("clouds", 320, 180, .1 * i)
        assert torch.equal(outs[i].view(torch.int32), ref.view(torch.int32)), i
    m.close()


def test_rccl_slice_of_the_library_on_one_gpu(renderer):
    """The RCCL calls sbx_multi_render is made of — dlopen, ncclCommInitAll, a grouped ncclSend / ncclRecv pair on two streams —
    run from a rank to itself on the one GPU of the box (the N-device exchange itself needs N devices)."""
    import ctypes
    step = ctypes.c_int(-1)
    rc = renderer.lib.sbx_multi_rccl_selftest(0, ctypes.byref(step))
    assert (rc, step.value) == (0, 0)


def test_library_multi_gpu_errors_and_noise_volumes(renderer, oracle, volumes):
    import torch
    import shaderbox_amd
    from oracle.oracle import APP_CLOUDS_TEX
    with pytest.raises(shaderbox_amd.SbxError):
        shaderbox_amd.MultiRenderer([0, 99])
    m = shaderbox_amd.MultiRenderer([0, 0, 0])
    with pytest.raises(shaderbox_amd.SbxError):
        m.set_split(0, 1, 1)
    with pytest.raises(shaderbox_amd.SbxError):
        m.render("clouds_tex", 64, 36, .37)                      # no volumes bound on the ranks
    v1, v2, h1, h2 = volumes
    m.set_noise_volumes(v1, v2)
    oracle.set_noise_volumes(h1, h2)
    got = m.render("clouds_tex", 160, 90, .37)
    torch.cuda.synchronize()
    ref = oracle.render(APP_CLOUDS_TEX, 160, 90, .37)
    assert compare(got.cpu().numpy(), ref) == (0.0, 0)
    m.close()


def test_sqrt_n_is_ieee_sqrt(renderer):
    """sqrt_n_ (v_sqrt_f32 + two-sided fix-up, sbx_math.h; used by APP_ATMOSPHERE) against the compiler's IEEE expansion on ALL
    2^32 binary32 inputs: identical everywhere except for non-zero arguments of magnitude below 2^-96, which its callers exclude
    (NaN results compare equal)."""
    import torch
    chunk = 1 << 26
    for start in range(0, 1 << 32, chunk):
        bits = torch.arange(start, start + chunk, dtype=torch.int64, device="cuda").to(torch.int32)   # wraps to the bit pattern
        x = bits.view(torch.float32)
        a = renderer.math("sqrt_n", x)
        b = renderer.math("sqrt_ieee", x)
        same = (a.view(torch.int32) == b.view(torch.int32)) | (torch.isnan(a) & torch.isnan(b))
        excluded = (x.abs() > 0) & (x.abs() < 2.0 ** -96)       # v_sqrt_f32 flushes denormals: -tiny gives -0, not NaN
        bad = ~same & ~excluded
        assert not bool(bad.any()), "first mismatch at bits 0x%08x" % int(bits[bad][0].item() & 0xffffffff)


# ---------------------------------------------------------------------------------------------------------
# the UE4 cloud variant (ue4/volumetric_clouds/Shaders/app_clouds.usf) under the build's host mapping
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("w,h,t", [(192, 108, 0.0), (256, 144, .37), (160, 160, 2.5), (33, 9, 7.0)])
def test_clouds_ue4_matches_oracle(renderer, oracle, w, h, t):
    from oracle.oracle import APP_CLOUDS_UE4
    ref = oracle.render(APP_CLOUDS_UE4, w, h, t)
    gpu = renderer.render("clouds_ue4", w, h, t).cpu().numpy()
    maxd, nbits = compare(gpu, ref)
    assert maxd <= 1e-4 and nbits == 0


def test_clouds_ue4_material_parameters(renderer, oracle):
    import shaderbox_amd
    from oracle.oracle import APP_CLOUDS_UE4
    aux = shaderbox_amd.AuxCloudsUe4()
    renderer.lib.sbx_aux_clouds_ue4_defaults(aux)
    assert (aux.coverage, aux.thickness, aux.fuzziness, aux.use_dirs) == (.5, 15.0, np.float32(.035), 0)
    aux.coverage, aux.thickness, aux.absorbtion, aux.fuzziness = .62, 22.0, .8, .05
    aux.sun_dir[0], aux.sun_dir[1], aux.sun_dir[2] = .0, .6, -.8
    aux.wind_dir[0], aux.wind_dir[1], aux.wind_dir[2] = .3, 0.0, -1.1
    aux.use_dirs = 1
    ref = oracle.render(APP_CLOUDS_UE4, 160, 90, 1.25, mouse=(1.5, 0.0), aux=aux)
    gpu = renderer.render("clouds_ue4", 160, 90, 1.25, mouse=(1.5, 0.0), aux=aux).cpu().numpy()
    assert compare(gpu, ref) == (0.0, 0)
    base = renderer.render("clouds_ue4", 160, 90, 1.25, mouse=(1.5, 0.0)).cpu().numpy()
    assert not np.array_equal(base, gpu)
    # full size: rows of a 3840x2160 frame
    W, H = 3840, 2160
    rows = list(range(20, H, 180))
    got = renderer.render("clouds_ue4", W, H, .37)[rows].cpu().numpy()
    assert compare(got, oracle.render_rows(APP_CLOUDS_UE4, W, H, .37, rows)) == (0.0, 0)


# ---------------------------------------------------------------------------------------------------------
# the direct exchange of the one-process-per-GPU path: root in place, peers' slabs without alpha, peer-only assembly
# ---------------------------------------------------------------------------------------------------------
ALL_APPS = ["clouds", "egg", "raytracer", "atmosphere", "planet", "sdf_ao", "vinyl", "clouds_best", "clouds_ue4"]


@pytest.mark.parametrize("app", ALL_APPS)
def test_rgb_slab_is_the_rgba_slab_without_alpha(renderer, app):
    """sbx_render_split_rgb writes the same three floats per pixel as sbx_render_split, densely (12 bytes per pixel), also for
    sub-ranges of the slab rows; and every kernel's alpha is the constant 1 the RGB form drops."""
    import torch
    from shaderbox_amd import shard
    w, h, br, n, rank = 200, 117, 8, 3, 1                    # ragged width and height
    rmax = shard.rank_rows_max(h, br, n)
    rows = shard.rank_rows(h, br, rank, n)
    rgba = torch.full((rmax, w, 4), -3.0, device="cuda")
    rgb = torch.full((rmax, w, 3), -3.0, device="cuda")
    renderer.render_rank_rows(app, w, h, .37, br, rank, n, 0, rmax, rgba)
    renderer.render_rank_rows(app, w, h, .37, br, rank, n, 0, 16, rgb)          # two launches: the second starts mid-slab
    renderer.render_rank_rows(app, w, h, .37, br, rank, n, 16, rmax, rgb)
    torch.cuda.synchronize()
    a, b = rgba.cpu().numpy(), rgb.cpu().numpy()
    assert np.array_equal(a[:rows, :, :3].view(np.uint32), b[:rows].view(np.uint32))
    assert (a[:rows, :, 3] == 1.0).all()
    assert (b[rows:] == -3.0).all() and (a[rows:] == -3.0).all()              # nothing written past the rank's rows


@pytest.mark.parametrize("channels", [3, 4])
@pytest.mark.parametrize("app,w,h,n,br,relief", [("clouds", 320, 180, 2, 8, (1, 1)), ("egg", 200, 117, 3, 8, (1, 2)),
                                                 ("raytracer", 333, 90, 8, 4, (3, 4)), ("clouds", 3840, 2160, 8, 8, (3, 4)),
                                                 ("planet", 256, 144, 4, 8, (0, 1)), ("sdf_ao", 64, 36, 1, 8, (1, 1))])
def test_direct_exchange_emulated_on_one_gpu(renderer, app, w, h, n, br, relief, channels):
    """N ranks on one device, the calls of distributed.FramePlan(exchange='direct') with the transfer replaced by the slab
    already being where the receive would put it: root in place + peers' slabs + sbx_assemble_peers == one launch."""
    import torch
    from shaderbox_amd import shard
    m0, m = relief
    whole = renderer.render(app, w, h, .37)
    rmax = shard.rank_rows_max(h, br, n, m0, m)
    frame = torch.full((h, w, 4), float("nan"), device="cuda")
    peers = torch.full((max(n - 1, 1), rmax, w, channels), float("nan"), device="cuda")
    renderer.render_rank_in_place(app, w, h, .37, br, 0, n, frame, root_rounds=m0, rounds=m)
    for r in range(1, n):
        renderer.render_rank_rows(app, w, h, .37, br, r, n, 0, rmax, peers[r - 1], root_rounds=m0, rounds=m)
    renderer.assemble_peers(peers, w, h, br, n, frame, root_rounds=m0, rounds=m)
    torch.cuda.synchronize()
    a, b = frame.view(torch.int32), whole.view(torch.int32)
    assert int((a != b).any(dim=-1).sum().item()) == 0


def test_assemble_peers_argument_errors(renderer):
    import ctypes
    import torch
    import shaderbox_amd
    lib = renderer.lib
    frame = torch.zeros((16, 16, 4), device="cuda")
    peers = torch.zeros((1, 8, 16, 3), device="cuda")
    fp = lambda t: ctypes.cast(ctypes.c_void_p(t.data_ptr()), ctypes.POINTER(ctypes.c_float))
    assert lib.sbx_assemble_peers(renderer.ctx, 16, 16, 8, 2, 1, 1, 5, fp(peers), fp(frame), None) == shaderbox_amd.SBX_ERR_ARG
    assert lib.sbx_assemble_peers(renderer.ctx, 16, 16, 8, 2, 1, 1, 3, None, fp(frame), None) == shaderbox_amd.SBX_ERR_ARG
    assert lib.sbx_assemble_peers(renderer.ctx, 16, 16, 8, 2, 2, 1, 3, fp(peers), fp(frame), None) == shaderbox_amd.SBX_ERR_ARG
    assert lib.sbx_assemble_peers(renderer.ctx, 16, 16, 8, 1, 1, 1, 3, None, fp(frame), None) == 0     # a lone rank has no peers
    with pytest.raises(ValueError):
        renderer.assemble_peers(torch.zeros((1, 4, 16, 3), device="cuda"), 16, 16, 8, 2, frame)


def test_regular_frame_exp_equals_exp_on_its_whole_domain(renderer):
    """cl_exp (kern_clouds.hip: no range guard, three-address v_fma_f64, power-of-two scaling after the rounding to binary32)
    against exp_ of the math spec on EVERY binary32 argument the regular-frame kernels can produce: |x| <= 80 (launch_clouds
    admits a frame only if |sigma * dt| <= 80 and the density is in [0, 1)), both signs, zeros, denormals; plus NaNs."""
    import torch
    lim = np.array([80.0], dtype=np.float32).view(np.uint32)[0]          # bit pattern of 80.0f: all patterns below are |x| < 80
    chunk = 1 << 26
    for sign in (0, 0x80000000):
        for start in range(0, int(lim) + 1, chunk):
            stop = min(start + chunk, int(lim) + 1)
            bits = (torch.arange(start, stop, dtype=torch.int64, device="cuda") | sign).to(torch.int32)
            x = bits.view(torch.float32)
            b = renderer.math("exp", x)
            for form in ("exp_reg", "exp_reg_plain"):                   # k_clouds' and k_atmosphere's instruction sequences
                a = renderer.math(form, x)
                bad = a.view(torch.int32) != b.view(torch.int32)
                assert not bool(bad.any()), "%s: first mismatch at bits 0x%08x" % (form, int(bits[bad][0].item() & 0xffffffff))
    nan = torch.tensor([float("nan"), -float("nan")], device="cuda")
    assert bool(torch.isnan(renderer.math("exp_reg", nan)).all()) and bool(torch.isnan(renderer.math("exp_reg_plain", nan)).all())


# ---------------------------------------------------------------------------------------------------------
# every pixel of the BASELINE.json frames
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.timeout(1800)
@pytest.mark.parametrize("app,w,h", [("egg", 1920, 1080), ("raytracer", 3840, 2160), ("clouds", 3840, 2160),
                                     ("atmosphere", 7680, 4320), ("planet", 7680, 4320)])
def test_every_pixel_of_the_baseline_frames(renderer, oracle, app, w, h):
    """C2 .. C5 of BASELINE.json at t = 0.37, mouse 0: the WHOLE frame against the CPU oracle, bit for bit (the other full-size
    tests compare evenly spread rows).  About a minute of host time in total on the GPU box's 256 threads
    (tests/full_frame_parity.py is the stand-alone form and also covers the §8f apps)."""
    from oracle.oracle import APP_IDS
    gpu = renderer.render(app, w, h, 0.37).cpu().numpy()
    bad = 0
    for y0 in range(0, h, 270):
        rows = list(range(y0, min(y0 + 270, h)))
        ref = oracle.render_rows(APP_IDS[app], w, h, 0.37, rows)
        g = gpu[y0:y0 + len(rows)]
        both_nan = np.isnan(g) & np.isnan(ref)
        bad += int(((g.view(np.uint32) != ref.view(np.uint32)) & ~both_nan).any(-1).sum())
    assert bad == 0
