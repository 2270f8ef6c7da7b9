"""GPU tests added in round 4: the per-pixel boundary at arbitrary fragCoords (VERDICT r3 "Next" #3), the 8-GPU form of config 5
(#1), the hardened error paths (#7)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ALL_APPS = ["planet", "clouds", "vinyl", "egg", "raytracer", "atmosphere", "sdf_ao", "clouds_best", "clouds_ue4", "clouds_sky",
            "vinyl_gpu", "planet_atmosphere"]


def compare(gpu, ref):
    both_nan = np.isnan(gpu) & np.isnan(ref)
    d = np.where(both_nan, 0.0, np.abs(gpu.astype(np.float64) - ref.astype(np.float64)))
    d = np.nan_to_num(d, nan=np.inf)
    bits = (gpu.view(np.uint32) != ref.view(np.uint32)) & ~both_nan
    return float(d.max()), int(bits.any(axis=-1).sum())


@pytest.fixture(scope="module")
def renderer():
    import shaderbox_amd
    r = shaderbox_amd.Renderer(0)
    yield r
    r.close()


def oracle_points(oracle, app, w, h, t, pts, mouse=(0.0, 0.0)):
    from oracle.oracle import APP_IDS
    return np.stack([oracle.main_image(APP_IDS[app], w, h, t, float(x), float(y), mouse=mouse) for x, y in pts])


# ---------------------------------------------------------------------------------------------------------
# mainImage(fragColor, fragCoord) at arbitrary coordinates (src/main.h:6-53: fragCoord is divided by u_res, never snapped)
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("app", ALL_APPS)
def test_main_image_off_centre_and_outside_the_frame(renderer, oracle, app):
    """sbx_main_image must evaluate the coordinate it is given: off-centre samples (a supersampling host), coordinates outside
    the frame and on its edges.  Round 3 returned floor(fragCoord)'s pixel, clamped to the frame, with rc 0."""
    w, h, t = 96, 54, 0.37
    pts = [(10.25, 20.75), (47.5, 30.5), (47.9990234375, 30.0009765625), (0.0, 0.0), (96.0, 54.0), (-3.5, 10.5), (50.5, -7.25),
           (120.5, 70.5), (95.5, 53.5), (0.5, 0.5), (31.125, 0.875)]
    got = np.array([renderer.main_image(app, w, h, t, p) for p in pts], dtype=np.float32)
    ref = oracle_points(oracle, app, w, h, t, pts)
    assert compare(got, ref) == (0.0, 0)
    # the two pixel centres above came from the cached frame, the others from one-point launches: both must agree with a
    # frame render
    frame = renderer.render(app, w, h, t).cpu().numpy()
    assert compare(got[8], frame[53, 95]) == (0.0, 0) and compare(got[9], frame[0, 0]) == (0.0, 0)


@pytest.mark.parametrize("app", ALL_APPS)
def test_render_points_random_against_oracle(renderer, oracle, app):
    """sbx_render_points / sbx_main_image_batch: n arbitrary fragCoords in one launch, bit-identical to the oracle's mainImage at
    those coordinates; the lanes of a wave are then NOT neighbouring pixels (hash cache, wave-wide exits and culls must not care)"""
    import torch
    rng = np.random.default_rng(7 + len(app))
    w, h, t = 640, 360, 0.37
    n = 700                                           # not a multiple of the pseudo-frame's row length
    pts = np.stack([rng.uniform(-40, w + 40, n), rng.uniform(-30, h + 30, n)], axis=1).astype(np.float32)
    pts[:64] = np.stack([rng.uniform(300, 302, 64), rng.uniform(200, 201, 64)], axis=1)      # a wave inside one pixel pair
    got_dev = renderer.render_points(app, w, h, t, torch.from_numpy(pts)).cpu().numpy()
    got_host = renderer.main_image_batch(app, w, h, t, pts)
    ref = oracle_points(oracle, app, w, h, t, pts)
    assert compare(got_dev, ref) == (0.0, 0)
    assert compare(got_host, ref) == (0.0, 0)


def test_render_points_pixel_centres_equal_the_frame(renderer):
    """the point list at every pixel centre of a frame, in a scrambled order, is that frame (both paths share the kernels)"""
    import torch
    w, h, t = 200, 120, 1.25
    for app in ("clouds", "egg", "atmosphere", "planet", "raytracer", "sdf_ao"):
        frame = renderer.render(app, w, h, t).cpu().numpy().reshape(-1, 4)
        ys, xs = np.divmod(np.arange(w * h), w)
        perm = np.random.default_rng(3).permutation(w * h)
        pts = np.stack([xs[perm] + .5, ys[perm] + .5], axis=1).astype(np.float32)
        got = renderer.render_points(app, w, h, t, torch.from_numpy(pts)).cpu().numpy()
        assert compare(got, frame[perm]) == (0.0, 0), app


def test_points_with_fractional_resolution_and_bad_arguments(renderer, oracle):
    import shaderbox_amd
    import torch
    pts = np.array([[10.5, 3.25], [100.0, 50.0]], dtype=np.float32)
    got = renderer.render_points("egg", 191.5, 107.25, 0.37, torch.from_numpy(pts)).cpu().numpy()
    ref = oracle_points(oracle, "egg", 191.5, 107.25, 0.37, pts)
    assert compare(got, ref) == (0.0, 0)
    with pytest.raises(shaderbox_amd.SbxError):
        renderer.render_points("egg", 0.0, 100.0, 0.37, torch.from_numpy(pts))
    with pytest.raises(shaderbox_amd.SbxError):
        renderer.render_points(99, 100.0, 100.0, 0.37, torch.from_numpy(pts))
    assert renderer.render_points("egg", 100.0, 100.0, 0.37, torch.zeros((0, 2))).shape == (0, 4)
    # NaN coordinates are data, as in the reference: whatever mainImage makes of them
    nanpt = np.array([[np.nan, 5.5]], dtype=np.float32)
    got = renderer.main_image_batch("clouds", 64, 36, 0.37, nanpt)
    ref = oracle_points(oracle, "clouds", 64, 36, 0.37, nanpt)
    assert compare(got, ref) == (0.0, 0)


def test_main_image_from_several_threads(renderer, oracle):
    """§8b: the per-pixel entry may be called from many host threads (the reference's globals are thread_local, src/def.h:7-8)"""
    import threading
    w, h, t = 64, 36, 0.37
    res = {}

    def work(k):
        out = []
        for i in range(40):
            x, y = (k * 7 + i * 3) % w + (.5 if i % 2 else .3), (k * 5 + i) % h + .5
            out.append(((x, y), renderer.main_image("egg", w, h, t, (x, y))))
        res[k] = out
    th = [threading.Thread(target=work, args=(k,)) for k in range(6)]
    [x.start() for x in th]
    [x.join() for x in th]
    pts = [p for k in sorted(res) for p, _ in res[k]]
    got = np.array([c for k in sorted(res) for _, c in res[k]], dtype=np.float32)
    assert compare(got, oracle_points(oracle, "egg", w, h, t, pts)) == (0.0, 0)


# ---------------------------------------------------------------------------------------------------------
# the 8-GPU form of the BASELINE configs on one GPU: every rank's real schedule (FramePlan) through a loopback world
# ---------------------------------------------------------------------------------------------------------
def loop_frame(renderer, app, w, h, t, n, exchange, groups=1, relief=(1, 1), br=8, frames=1):
    import torch
    from shaderbox_amd.distributed import LoopbackWorld
    world = LoopbackWorld(n)
    plans = world.plans(renderer, w, h, block_rows=br, groups=groups, root_rounds=relief[0], rounds=relief[1], exchange=exchange)
    out = None
    for _ in range(frames):
        plans[0].frame.fill_(-7.0)                    # every pixel must be written again
        out = LoopbackWorld.render(plans, app, t)
    torch.cuda.synchronize()
    return out, world


@pytest.mark.parametrize("app", ["atmosphere", "planet"])
def test_config5_eight_ranks_at_7680x4320_equal_one_launch(renderer, app):
    """BASELINE config 5 as it runs on 8 GPUs — cyclic 8-row blocks, root in place, ONE exchange — with all 8 ranks' FramePlans on
    this one GPU: the direct exchange (whole 3-channel slabs) and the span exchange (only the expensive interval of every block
    crosses; the root renders the rest), plain and with root relief and pipelined groups.  Same bits as one launch."""
    import torch
    w, h, t = 7680, 4320, 0.37
    full = renderer.render(app, w, h, t)
    for exchange, groups, relief in [("direct", 1, (1, 1)), ("spans", 1, (1, 1)), ("spans", 3, (3, 4))]:
        got, world = loop_frame(renderer, app, w, h, t, 8, exchange, groups, relief)
        assert torch.equal(got.view(torch.int32), full.view(torch.int32)), (app, exchange, groups, relief)
        mb = world.bytes_moved / 1e6
        print("%s 7680x4320, 8 ranks, %s, groups %d, relief %s: %.1f MB cross the links per frame" % (app, exchange, groups, relief, mb))
        if exchange == "spans":
            assert mb < 0.66 * 7 * 49.8                # 7 x 49.8 MB with whole slabs
        del got, world
        torch.cuda.empty_cache()


@pytest.mark.parametrize("n", [2, 3, 8])
def test_span_exchange_small_frames(renderer, n):
    """ragged sizes, every app with a span model plus one without, frames repeated into the same buffers"""
    import torch
    for app, w, h, t, br, groups, relief in [("clouds", 1000, 333, .37, 8, 2, (1, 1)), ("atmosphere", 1111, 500, 1.5, 8, 1, (1, 2)),
                                             ("planet", 900, 400, .37, 4, 3, (1, 1)), ("egg", 203, 95, .37, 8, 1, (1, 1)),
                                             ("clouds_sky", 640, 360, .37, 8, 1, (1, 1))]:
        full = renderer.render(app, w, h, t)
        got, _ = loop_frame(renderer, app, w, h, t, n, "spans", groups, relief, br, frames=2)
        assert torch.equal(got.view(torch.int32), full.view(torch.int32)), (app, n)


def test_span_exchange_follows_the_camera(renderer):
    """the span table depends on the camera: APP_CLOUDS with the mouse turned (another layout key), same plans reused"""
    import torch
    from shaderbox_amd.distributed import LoopbackWorld
    world = LoopbackWorld(4)
    plans = world.plans(renderer, 1280, 720, exchange="spans")
    for mouse in [(0.0, 0.0), (2.0, 0.0), (0.0, 0.0)]:
        got = LoopbackWorld.render(plans, "clouds", .37, mouse=mouse)
        full = renderer.render("clouds", 1280, 720, .37, mouse=mouse)
        assert torch.equal(got.view(torch.int32), full.view(torch.int32)), mouse


# ---------------------------------------------------------------------------------------------------------
# hardening (VERDICT r3 "Next" #7)
# ---------------------------------------------------------------------------------------------------------
def test_hash_cache_fault_is_sticky_and_loud(renderer):
    """hc_slow's round bound (sbx_hashcache.h) used to return zeros silently.  Now the wave sets the device's sticky fault word;
    every later render on that device fails with SBX_ERR_FAULT, sbx_last_error says why, sbx_clear_fault re-arms."""
    import torch
    import shaderbox_amd
    assert renderer.fault_status() == 0
    ref = renderer.render("clouds", 96, 54, .37)
    assert renderer.lib.sbx_debug_raise_fault(renderer.ctx, None) == 0      # one wave through the path a failed miss loop takes
    torch.cuda.synchronize()
    assert renderer.fault_status() == shaderbox_amd.SBX_ERR_FAULT
    with pytest.raises(shaderbox_amd.SbxError) as e:
        renderer.render("clouds", 96, 54, .37)
    assert e.value.code == shaderbox_amd.SBX_ERR_FAULT and "hash cache" in str(e.value)
    other = shaderbox_amd.Renderer(0)                                     # the word belongs to the device, not to a context
    with pytest.raises(shaderbox_amd.SbxError):
        other.render("egg", 32, 32, .37)
    assert b"hash cache" in other.lib.sbx_last_error(other.ctx)
    other.clear_fault()
    other.close()
    assert renderer.fault_status() == 0
    again = renderer.render("clouds", 96, 54, .37)
    assert torch.equal(again.view(torch.int32), ref.view(torch.int32))


def test_multi_create_device_lists(renderer):
    """sbx_multi_create with repeated-then-distinct device lists: ranks that share a device and ranks that do not in one world
    (copies instead of RCCL); a list naming a device the box does not have fails cleanly with the reason."""
    import torch
    import shaderbox_amd
    ndev = torch.cuda.device_count()
    if ndev >= 2:
        for devs in ([0, 0, 1], [0, 1, 1, 0]):
            m = shaderbox_amd.MultiRenderer(devs)
            assert not m.uses_rccl
            got = m.render("clouds", 200, 117, .37)
            torch.cuda.synchronize()
            assert torch.equal(got.view(torch.int32), renderer.render("clouds", 200, 117, .37).view(torch.int32))
            m.close()
    else:
        with pytest.raises(shaderbox_amd.SbxError) as e:
            shaderbox_amd.MultiRenderer([0, 0, 1])
        assert "device id out of range" in str(e.value)
        m = shaderbox_amd.MultiRenderer([0, 0, 0, 0])
        assert not m.uses_rccl and m.lib.sbx_multi_ranks(m.m) == 4
        m.close()


# ---------------------------------------------------------------------------------------------------------
# ADVICE r3
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("app", ["vinyl", "vinyl_gpu", "sdf_ao", "egg"])
def test_hardware_min_max_kernels_equal_plain_over_a_time_sweep(renderer, app):
    """the SDF kernels' v_min_f32 / v_max_f32 (pinned with inline asm, sbx_sdf.h hmin_ / hmax_) against the compare-and-select
    spec form (variant 1: plain kernels) over a random u_time sweep, bit for bit — the zero-sign and NaN argument of the proof is
    about operand ORDER, so it is checked where it could break: many poses"""
    rng = np.random.default_rng(11)
    ts = np.concatenate([rng.uniform(0, 20, 24), rng.uniform(-500, 5000, 8), [0.0, 1e5, 3.3e7]])
    for t in ts:
        renderer.set_variant(0)
        a = renderer.render(app, 240, 135, float(t)).cpu().numpy()
        renderer.set_variant(1)
        b = renderer.render(app, 240, 135, float(t)).cpu().numpy()
        renderer.set_variant(0)
        assert compare(a, b) == (0.0, 0), (app, float(t))


def test_atmosphere_below_the_horizon_against_oracle(renderer, oracle):
    """k_atmosphere's length() is sqrt_rs_ (exact for finite x >= 2^-102, NaN at 0): the rays that could bring |s|^2 near 0 are the
    ones that dive through the planet (1 < z2 <= 2, src/app_atmosphere.h:195-207), up to the straight-down direction on the circle
    z2 = 2.  Points all over that band, its two edges and odd resolutions (centre pixels exactly on the axes) against the oracle."""
    import torch
    rng = np.random.default_rng(5)
    w, h = 1920.0, 1080.0
    n = 6000
    z2 = np.concatenate([rng.uniform(0.98, 2.002, n - 2000), rng.uniform(1.9995, 2.0005, 1000), rng.uniform(0.9995, 1.0005, 1000)])
    phi = rng.uniform(-np.pi, np.pi, n)
    phi[:64] = np.repeat([0.0, np.pi / 2, -np.pi / 2, np.pi], 16)
    pcx, pcy = np.sqrt(z2) * np.cos(phi), np.sqrt(z2) * np.sin(phi)
    keep = np.abs(pcy) <= 1.02
    fx = (pcx[keep] / (w / h) + 1.0) * .5 * w
    fy = (pcy[keep] + 1.0) * .5 * h
    pts = np.stack([fx, fy], axis=1).astype(np.float32)
    for t in (0.37, 3.0):
        got = renderer.render_points("atmosphere", w, h, t, torch.from_numpy(pts)).cpu().numpy()
        ref = oracle_points(oracle, "atmosphere", w, h, t, pts)
        assert compare(got, ref) == (0.0, 0), t
    for ww, hh in [(333, 187), (101, 101), (255, 143)]:
        from oracle.oracle import APP_ATMOSPHERE
        got = renderer.render("atmosphere", ww, hh, 0.37).cpu().numpy()
        assert compare(got, oracle.render(APP_ATMOSPHERE, ww, hh, 0.37)) == (0.0, 0), (ww, hh)


def test_bench_relief_calibration_for_the_span_exchange(renderer):
    """bench.py choose_relief('auto') on rank 0 with a stand-in world of 8 and the span exchange: the root's emulated frame (its
    launch over the whole frame + landing of the packed spans + scatter) against the peers', for APP_ATMOSPHERE and APP_CLOUDS"""
    import importlib.util
    import os
    import torch
    from shaderbox_amd import shard
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)

    class FakeDist:
        @staticmethod
        def broadcast(t, src=0):
            return None
    dev = torch.device("cuda", 0)
    streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
    for app, w, h in (("atmosphere", 1920, 1080), ("clouds", 960, 540)):
        m0, m = bench.choose_relief("auto", renderer, FakeDist, torch, dev, app, w, h, .37, 8, 8, 0, streams, "spans", 3)
        assert (m0, m) in bench.relief_candidates() and 0 <= m0 <= m
        assert sum(shard.rank_rows(h, 8, r, 8, m0, m) for r in range(8)) == h
    assert bench.auto_groups("auto", 9.2e6) == 1 and bench.auto_groups("auto", 29.7e6) == 3 and bench.auto_groups("2", 1e9) == 2


def test_clouds_marches_longer_than_the_ring_tables(renderer, oracle):
    """cld_march_steps beyond the 4096 rows of the y-table ring: the context's on-demand table (sbx_capi.hip render_clouds case 4)
    instead of round 3's table-less fallback; same bits as the per-lane kernel and the oracle, across key changes and streams"""
    import torch
    import shaderbox_amd
    from oracle.oracle import APP_CLOUDS
    aux = shaderbox_amd.clouds_defaults()
    w, h = 96, 54
    s2 = torch.cuda.Stream()
    for steps, t in [(4097, .37), (6000, .37), (6000, 1.5), (4100, .37), (100, .37), (9000, .37)]:
        aux.cld_march_steps = steps
        aux.cld_thick = 125.0 * steps / 100.0 if steps < 5000 else 300.0
        renderer.set_variant(0)
        a = renderer.render("clouds", w, h, t, aux=aux).cpu().numpy()
        with torch.cuda.stream(s2):
            a2 = renderer.render("clouds", w, h, t, aux=aux)
        s2.synchronize()
        renderer.set_variant(1)
        b = renderer.render("clouds", w, h, t, aux=aux).cpu().numpy()
        renderer.set_variant(0)
        assert compare(a, b) == (0.0, 0), steps
        assert compare(a2.cpu().numpy(), b) == (0.0, 0), steps
    aux.cld_march_steps, aux.cld_thick = 4500, 200.0
    got = renderer.render("clouds", 64, 36, .37, aux=aux).cpu().numpy()
    assert compare(got, oracle.render(APP_CLOUDS, 64, 36, .37, aux=aux)) == (0.0, 0)


# ---------------------------------------------------------------------------------------------------------
# BASELINE config 5 read literally: APP_PLANET with APP_ATMOSPHERE's sky as its background (labelled extension, parity unpinned)
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("w,h,t", [(256, 144, .37), (333, 187, 0.0), (640, 360, 2.5), (96, 54, 40.0)])
def test_planet_atmosphere_composite_matches_oracle(renderer, oracle, w, h, t):
    """SBX_APP_PLANET_ATMOSPHERE (include/sbx.h): default and plain kernels == the oracle's restatement of the same definition,
    NaN == NaN; where APP_PLANET shows terrain the two apps agree, elsewhere the sky is APP_ATMOSPHERE's"""
    from oracle.oracle import APP_IDS
    renderer.set_variant(0)
    a = renderer.render("planet_atmosphere", w, h, t).cpu().numpy()
    renderer.set_variant(1)
    b = renderer.render("planet_atmosphere", w, h, t).cpu().numpy()
    renderer.set_variant(0)
    ref = oracle.render(APP_IDS["planet_atmosphere"], w, h, t)
    assert compare(a, b) == (0.0, 0)
    assert compare(a, ref) == (0.0, 0)
    plain = renderer.render("planet", w, h, t).cpu().numpy()
    same = ((a.view(np.uint32) == plain.view(np.uint32)) | (np.isnan(a) & np.isnan(plain))).all(-1)
    assert 0.05 < same.mean() < 0.6                     # the terrain pixels; backgrounds differ


def test_planet_atmosphere_through_the_multi_gpu_schedule(renderer):
    import torch
    w, h, t = 1280, 720, .37
    full = renderer.render("planet_atmosphere", w, h, t)
    for exchange in ("spans", "direct"):
        got, _ = loop_frame(renderer, "planet_atmosphere", w, h, t, 8, exchange)
        assert torch.equal(got.view(torch.int32), full.view(torch.int32)), exchange


@pytest.mark.parametrize("nranks", [2, 8])
def test_library_multi_gpu_span_exchange(renderer, nranks):
    """sbx_multi with SBX_MULTI_EXCHANGE_SPANS (all ranks on device 0: copies instead of RCCL): same bits as one launch, with
    root relief, ragged sizes, the BASELINE config-5 frame, frames in flight"""
    import torch
    import shaderbox_amd
    m = shaderbox_amd.MultiRenderer([0] * nranks)
    m.set_exchange("spans")
    cases = [("clouds", 1000, 333, .37), ("atmosphere", 1111, 500, .37), ("planet", 900, 400, .37), ("egg", 203, 95, .37)]
    if nranks == 8:
        cases.append(("atmosphere", 7680, 4320, .37))
    for app, w, h, t in cases:
        full = renderer.render(app, w, h, t)
        for split in [(8, 1, 1), (8, 1, 2), (4, 0, 1)]:
            m.set_split(*split)
            got = m.render(app, w, h, t)
            torch.cuda.synchronize()
            assert torch.equal(got.view(torch.int32), full.view(torch.int32)), (app, w, h, nranks, split)
        del full
    m.set_split(8, 1, 1)
    s = [torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [torch.zeros((360, 640, 4), dtype=torch.float32, device="cuda") for _ in range(6)]
    for i in range(6):
        with torch.cuda.stream(s[i % 3]):                 # three caller streams over the library's two slots
            m.render("clouds", 640, 360, .1 * i, out=outs[i])
    torch.cuda.synchronize()
    for i in range(6):
        assert torch.equal(outs[i].view(torch.int32), renderer.render("clouds", 640, 360, .1 * i).view(torch.int32)), i
    m.close()


def test_points_and_spans_with_noise_textures(renderer, oracle):
    """APP_CLOUDS' USE_NOISE_TEX build through the point list and the span exchange (the volumes are context state)"""
    import torch
    from oracle.oracle import APP_CLOUDS_TEX
    v1, v2 = renderer.worley_volume(32), renderer.worley_volume(16)
    renderer.set_noise_volumes(v1, v2)
    oracle.set_noise_volumes(v1.cpu().numpy(), v2.cpu().numpy())
    rng = np.random.default_rng(17)
    w, h, t = 320, 180, 0.37
    pts = np.stack([rng.uniform(-10, w + 10, 300), rng.uniform(-10, h + 10, 300)], axis=1).astype(np.float32)
    got = renderer.render_points("clouds_tex", w, h, t, torch.from_numpy(pts)).cpu().numpy()
    ref = np.stack([oracle.main_image(APP_CLOUDS_TEX, w, h, t, float(x), float(y)) for x, y in pts])
    assert compare(got, ref) == (0.0, 0)
    full = renderer.render("clouds_tex", 1000, 400, t)
    got, _ = loop_frame(renderer, "clouds_tex", 1000, 400, t, 4, "spans", groups=2)
    assert torch.equal(got.view(torch.int32), full.view(torch.int32))


def test_very_long_point_list(renderer):
    """27.4 million points in one call (the pseudo-frame widens so that the grid's y extent stays legal): three and a bit copies of
    every pixel centre of the 3840x2160 APP_CLOUDS frame equal that frame"""
    import torch
    w, h, t = 3840, 2160, 0.37
    frame = renderer.render("clouds", w, h, t).reshape(-1, 4)
    idx = torch.arange(27_400_001, device="cuda") % (w * h)
    pts = torch.stack([(idx % w).float() + .5, (idx // w).float() + .5], dim=1).contiguous()
    got = renderer.render_points("clouds", w, h, t, pts)
    torch.cuda.synchronize()
    assert torch.equal(got.view(torch.int32), frame[idx].view(torch.int32))


def test_pieces_of_a_three_channel_slab_of_an_odd_width_frame(renderer):
    """found by tools/soak_spans.py: piece g of a 3-channel slab starts at r0 * W * 12 bytes, which is only 4-byte aligned for odd W;
    the ABI used to demand 16 (the alignment of float4 pixels) of every output pointer"""
    import torch
    for app, w, h, n, groups, exchange in [("egg", 203, 95, 3, 3, "direct"), ("clouds", 333, 187, 4, 4, "direct"),
                                            ("atmosphere", 1111, 301, 5, 3, "spans")]:
        full = renderer.render(app, w, h, .37)
        got, _ = loop_frame(renderer, app, w, h, .37, n, exchange, groups=groups)
        assert torch.equal(got.view(torch.int32), full.view(torch.int32)), (app, exchange)


def test_main_image_cache_follows_every_input(renderer, oracle):
    """a random walk over (app, u_res, u_time, u_mouse, aux, fragCoord): whatever the previous call cached, every answer is the
    oracle's mainImage of the CURRENT arguments"""
    import shaderbox_amd
    from oracle.oracle import APP_IDS
    rng = np.random.default_rng(23)
    apps = ["clouds", "egg", "sdf_ao", "atmosphere", "raytracer"]
    app, w, h, t, mouse = "clouds", 48, 27, .37, (0.0, 0.0)
    aux = None
    for i in range(160):
        k = int(rng.integers(0, 7))
        if k == 0:
            app = apps[int(rng.integers(len(apps)))]; aux = None
        elif k == 1:
            w, h = int(rng.integers(8, 64)), int(rng.integers(8, 40))
        elif k == 2:
            t = float(rng.uniform(0, 9))
        elif k == 3:
            mouse = (float(rng.uniform(0, 6)), 0.0)
        elif k == 4 and app == "clouds":
            aux = shaderbox_amd.clouds_defaults(); aux.cld_coverage = float(rng.uniform(.3, .8))
        elif k == 4 and app == "sdf_ao":
            aux = shaderbox_amd.sdf_ao_defaults(); aux.fog_density = float(rng.uniform(.05, .3))
        fx = float(rng.integers(0, w)) + (.5 if i % 3 else float(rng.uniform(0, 1)))
        fy = float(rng.integers(0, h)) + .5
        got = np.array(renderer.main_image(app, w, h, t, (fx, fy), mouse=mouse, aux=aux), dtype=np.float32)
        ref = oracle.main_image(APP_IDS[app], w, h, t, fx, fy, mouse=mouse, aux=aux)
        assert compare(got[None], ref[None]) == (0.0, 0), (i, app, w, h, t, mouse, fx, fy)


def test_atmosphere_dead_rays_over_sun_positions(renderer, oracle):
    """k_atmosphere ends a ray's march once its optical depth has overflowed to +inf (sbx_atmosphere.h "dead rays": every later
    contribution is exactly +0) and skips the angle mapping of waves outside the dome.  Both are claims about IEEE arithmetic for
    EVERY sun position and ray: frames over the sun's whole swing (|sin(t / 2)| from 0 to 1, src/app_atmosphere.h:177-181), wide
    and tall aspect ratios, against the oracle, NaN == NaN."""
    from oracle.oracle import APP_ATMOSPHERE
    rng = np.random.default_rng(31)
    times = list(rng.uniform(0, 6.3, 10)) + [0.0, 3.14159, 3.1415927, 1.5707964, 100.0, 1e4]
    sizes = [(320, 180), (180, 320), (257, 129), (96, 400)]
    for i, t in enumerate(times):
        w, h = sizes[i % len(sizes)]
        got = renderer.render("atmosphere", w, h, float(t)).cpu().numpy()
        assert compare(got, oracle.render(APP_ATMOSPHERE, w, h, float(t))) == (0.0, 0), (t, w, h)


def test_span_entry_points_reject_bad_arguments(renderer):
    """the span entry points fail with SBX_ERR_ARG (never render something else): rank 0 as a peer, a slab range that does not start
    on a block, a bad split, a frame whose table is not on the device yet inside a stream capture"""
    import ctypes
    import torch
    import shaderbox_amd
    lib, ctx = renderer.lib, renderer.ctx
    u = renderer.uniforms(640, 360, .37)
    slab = torch.zeros(640 * 360 * 3, dtype=torch.float32, device="cuda")
    frame = torch.zeros((360, 640, 4), dtype=torch.float32, device="cuda")
    sp = ctypes.c_void_p(slab.data_ptr()); fp = ctypes.c_void_p(frame.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    app = shaderbox_amd.app_id("clouds")
    assert lib.sbx_render_span_peer(ctx, app, ctypes.byref(u), None, 8, 0, 4, 1, 1, 0, 1 << 30, sp, st) == shaderbox_amd.SBX_ERR_ARG
    assert lib.sbx_render_span_peer(ctx, app, ctypes.byref(u), None, 8, 4, 4, 1, 1, 0, 1 << 30, sp, st) == shaderbox_amd.SBX_ERR_ARG
    assert lib.sbx_render_span_peer(ctx, app, ctypes.byref(u), None, 8, 1, 4, 1, 1, 3, 1 << 30, sp, st) == shaderbox_amd.SBX_ERR_ARG
    assert lib.sbx_render_span_root(ctx, app, ctypes.byref(u), None, 8, 4, 3, 2, fp, st) == shaderbox_amd.SBX_ERR_ARG
    assert lib.sbx_render_span_root(ctx, 99, ctypes.byref(u), None, 8, 4, 1, 1, fp, st) < 0
    assert lib.sbx_assemble_spans(ctx, app, ctypes.byref(u), None, 8, 4, 1, 1, None, 1000, fp, st) == shaderbox_amd.SBX_ERR_ARG
    assert lib.sbx_span_table(99, ctypes.byref(u), None, 8, 4, 1, 1, None, None, None) < 0
    # a table that is not on the device yet cannot be uploaded inside a capture: a clean error, and the capture stays usable
    u2 = renderer.uniforms(648, 360, .37)
    frame2 = torch.zeros((360, 648, 4), dtype=torch.float32, device="cuda")
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            rc = lib.sbx_render_span_root(ctx, app, ctypes.byref(u2), None, 8, 5, 1, 1, ctypes.c_void_p(frame2.data_ptr()),
                                          ctypes.c_void_p(s.cuda_stream))
            frame2.add_(0.0)                          # something to capture
    assert rc == shaderbox_amd.SBX_ERR_ARG and b"span table" in lib.sbx_last_error(ctx)
    # outside the capture the same call works and, once the table is there, it can be captured and replayed
    assert lib.sbx_render_span_root(ctx, app, ctypes.byref(u2), None, 8, 5, 1, 1, ctypes.c_void_p(frame2.data_ptr()), st) == 0
    torch.cuda.synchronize()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g2, stream=s):
            assert lib.sbx_render_span_root(ctx, app, ctypes.byref(u2), None, 8, 1, 1, 1, ctypes.c_void_p(frame2.data_ptr()),
                                            ctypes.c_void_p(s.cuda_stream)) in (0, shaderbox_amd.SBX_ERR_ARG)


@pytest.mark.parametrize("app", ["egg", "sdf_ao", "vinyl", "vinyl_gpu", "raytracer"])
def test_witnessed_square_roots_of_the_sdf_kernels(renderer, oracle, app):
    """k_egg / k_sdf_ao / k_vinyl take sqrt_rs_ (five instructions, equal to the IEEE root on [2^-102, inf) by exhaustion) in sdf()
    and record any argument outside that interval; a wave with a record re-runs its pixels with the IEEE roots (csrc/sbx_witness.h);
    k_raytracer does the same with its roots and its normalisations (zero components of a normalised vector are recorded too).
    Variant 2 raises the recording edge to 1.0, so every wave near a primitive re-runs; variant 3 is the culled kernel with the IEEE
    roots only; variant 1 the plain kernel.  All four and the oracle: the same bits, over poses, odd sizes and mouse positions."""
    from oracle.oracle import APP_IDS
    rng = np.random.default_rng(77)
    cases = [(240, 135, .37, (0.0, 0.0)), (333, 187, 2.9, (100.0, 20.0))]
    cases += [(int(rng.integers(64, 400)), int(rng.integers(48, 260)), float(rng.uniform(0, 40)),
               (float(rng.uniform(0, 300)), float(rng.uniform(0, 200)))) for _ in range(8)]
    try:
        for i, (w, h, t, mouse) in enumerate(cases):
            frames = []
            for v in (0, 2, 3, 1):
                renderer.set_variant(v)
                frames.append(renderer.render(app, w, h, t, mouse=mouse).cpu().numpy())
            for v, f in zip((2, 3, 1), frames[1:]):
                assert compare(frames[0], f) == (0.0, 0), (app, v, w, h, t, mouse)
            if i < 3:
                assert compare(frames[0], oracle.render(APP_IDS[app], w, h, t, mouse=mouse)) == (0.0, 0), (app, w, h, t, mouse)
    finally:
        renderer.set_variant(0)
    with pytest.raises(Exception):
        renderer.set_variant(4)
    renderer.set_variant(0)


# ---------------------------------------------------------------------------------------------------------
# SBX_FORMAT_RGBA8: the kernels write the display format themselves (include/sbx.h sbx_set_output_format)
# ---------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def renderer8():
    import shaderbox_amd
    r = shaderbox_amd.Renderer(0)
    r.set_output_format("rgba8")
    yield r
    r.close()


@pytest.mark.parametrize("app", ALL_APPS)
def test_rgba8_frames_equal_the_packed_float_frames(renderer, renderer8, app):
    """a frame, a strip and a rank's slab written as R8G8B8A8_UNORM words by the render kernel == sbx_pack_unorm8 of the float
    pixels, for every app (NaN pixels pack to 0 on both sides), odd sizes included"""
    import torch
    if app == "clouds_tex":
        pytest.skip("needs noise volumes (covered by test_rgba8_with_noise_textures)")
    for w, h, t in [(320, 180, .37), (333, 187, 2.9)]:
        f = renderer.render(app, w, h, t)
        want = renderer.pack_unorm8(f, flip_y=False)
        got = renderer8.render(app, w, h, t)
        assert got.dtype == torch.uint8 and tuple(got.shape) == (h, w, 4)
        assert torch.equal(got, want), (app, w, h, t)
        strip = renderer8.render(app, w, h, t, rows=(41, 97))
        assert torch.equal(strip, want[41:97]), (app, w, h, "strip")
        slab = renderer8.render_rank(app, w, h, t, 8, 2, 3)
        ref = renderer.pack_unorm8(renderer.render_rank(app, w, h, t, 8, 2, 3), flip_y=False)
        from shaderbox_amd import shard
        n = shard.rank_rows(h, 8, 2, 3)
        assert torch.equal(slab[:n], ref[:n]), (app, w, h, "rank slab")


@pytest.mark.parametrize("n", [2, 3, 8])
def test_rgba8_through_the_multi_rank_schedules(renderer, renderer8, n):
    """every exchange form of FramePlan with 4-byte pixels: the assembled uint8 frame == the packed float frame of one launch;
    a third of the float exchange's bytes cross the links"""
    import torch
    for app, w, h, t, exchange, groups, relief in [("clouds", 1000, 333, .37, "spans", 2, (1, 1)), ("atmosphere", 1111, 500, 1.5, "spans", 1, (1, 2)),
                                                   ("planet", 640, 360, .37, "direct", 2, (3, 4)), ("egg", 641, 357, 1.0, "direct", 1, (1, 1)),
                                                   ("raytracer", 500, 300, .37, "direct", 2, (1, 1)), ("atmosphere", 800, 450, .37, "direct", 3, (0, 1))]:
        want = renderer.pack_unorm8(renderer.render(app, w, h, t), flip_y=False)
        from shaderbox_amd.distributed import LoopbackWorld
        world = LoopbackWorld(n)
        plans = world.plans(renderer8, w, h, block_rows=8, groups=groups, root_rounds=relief[0], rounds=relief[1], exchange=exchange)
        for _ in range(2):
            plans[0].frame.fill_(7)
            got = LoopbackWorld.render(plans, app, t)
        torch.cuda.synchronize()
        assert got.dtype == torch.uint8 and torch.equal(got, want), (app, w, h, n, exchange)
        fworld = LoopbackWorld(n)
        fplans = fworld.plans(renderer, w, h, block_rows=8, groups=groups, root_rounds=relief[0], rounds=relief[1], exchange=exchange)
        LoopbackWorld.render(fplans, app, t)
        torch.cuda.synchronize()
        assert abs(fworld.bytes_moved - 3.0 * world.bytes_moved / 2) <= 64, (fworld.bytes_moved, world.bytes_moved)   # (two rgba8 frames)
    # the gather form's assembly (whole RGBA slabs of every rank, the root's included)
    from shaderbox_amd import shard
    w, h, t, app = 333, 187, .37, "sdf_ao"
    rows_max = shard.rank_rows_max(h, 8, n)
    gathered = torch.zeros((n, rows_max, w, 4), dtype=torch.uint8, device="cuda")
    for r in range(n):
        renderer8.render_rank(app, w, h, t, 8, r, n, out=gathered[r])
    got = renderer8.assemble(gathered, w, h, 8, n)
    assert torch.equal(got, renderer.pack_unorm8(renderer.render(app, w, h, t), flip_y=False))


def test_rgba8_leaves_points_and_main_image_float(renderer, renderer8, oracle):
    """sbx_render_points / sbx_main_image(_batch) return float colours whatever the context's output format, and the format can be
    switched back; unknown formats are refused"""
    import torch
    import shaderbox_amd
    w, h, t = 200, 120, .37
    pts = torch.tensor([[10.5, 20.5], [100.25, 60.75], [-3.0, 500.0]], dtype=torch.float32, device="cuda")
    a = renderer8.render_points("egg", w, h, t, pts)
    b = renderer.render_points("egg", w, h, t, pts)
    assert a.dtype == torch.float32 and torch.equal(a.view(torch.int32), b.view(torch.int32))
    assert renderer8.main_image("raytracer", w, h, t, (10.5, 20.5)) == renderer.main_image("raytracer", w, h, t, (10.5, 20.5))
    assert renderer8.main_image("raytracer", w, h, t, (11.5, 20.5)) == renderer.main_image("raytracer", w, h, t, (11.5, 20.5))
    r = shaderbox_amd.Renderer(0)
    try:
        r.set_output_format("rgba8")
        r.set_output_format("rgba32f")
        f = r.render("egg", w, h, t)
        assert f.dtype == torch.float32 and torch.equal(f.view(torch.int32), renderer.render("egg", w, h, t).view(torch.int32))
        assert r.lib.sbx_set_output_format(r.ctx, 7) == shaderbox_amd.SBX_ERR_ARG
    finally:
        r.close()


def test_rgba8_with_noise_textures(renderer):
    import torch
    import shaderbox_amd
    r = shaderbox_amd.Renderer(0)
    try:
        g = torch.Generator(device="cpu").manual_seed(5)
        shape = torch.rand((32, 32, 32, 4), generator=g).cuda()
        detail = torch.rand((16, 16, 16, 4), generator=g).cuda()
        r.set_noise_volumes(shape, detail)
        f = r.render("clouds_tex", 320, 180, .37)
        want = r.pack_unorm8(f, flip_y=False)
        r.set_output_format("rgba8")
        assert torch.equal(r.render("clouds_tex", 320, 180, .37), want)
    finally:
        r.close()


@pytest.mark.parametrize("nranks", [3, 8])
def test_rgba8_through_the_library_multi_gpu_engine(renderer, nranks):
    """sbx_multi_set_output_format(SBX_FORMAT_RGBA8) with every exchange form of the library engine (all ranks on device 0: copies
    instead of RCCL): the uint8 frame == the packed float frame; switching back gives the float frame again"""
    import torch
    import shaderbox_amd
    m = shaderbox_amd.MultiRenderer([0] * nranks)
    try:
        for exchange in ("slabs", "blocks", "spans", "peer_stores"):
            m.set_exchange(exchange)
            for app, w, h, t, split in [("clouds", 1000, 333, .37, (8, 1, 1)), ("atmosphere", 1111, 500, .37, (8, 1, 2)), ("egg", 203, 95, .37, (4, 0, 1))]:
                m.set_split(*split)
                f = renderer.render(app, w, h, t)
                want = renderer.pack_unorm8(f, flip_y=False)
                m.set_output_format("rgba8")
                got = m.render(app, w, h, t)
                torch.cuda.synchronize()
                assert got.dtype == torch.uint8 and torch.equal(got, want), (exchange, app, w, h, split)
                m.set_output_format("rgba32f")
                got = m.render(app, w, h, t)
                torch.cuda.synchronize()
                assert torch.equal(got.view(torch.int32), f.view(torch.int32)), (exchange, app, "float again")
    finally:
        m.close()


def test_witnessed_normalize_equals_ieee_where_it_does_not_record(renderer):
    """sbx_witness.h Wit::normalize — sqrt_rs_, v_rcp_f32 + one Newton step, three div3_ — against v / length(v) in IEEE arithmetic
    (numpy binary32, the dot product in the spec's association) on 24 M vectors whose components span 2^-75 ... 2^30 with zeros,
    denormals, infinities and NaNs mixed in: wherever the form does NOT record, it is the IEEE result bit for bit (and the device's
    own IEEE form agrees with numpy everywhere); the record fires for every vector with a zero, tiny or non-finite component or a
    squared length outside [2^-102, 2^40), and on ordinary vectors it does not fire"""
    import torch
    g = torch.Generator(device="cpu").manual_seed(99)
    for block in range(6):
        n = 4_000_000
        e = torch.rand((n, 3), generator=g) * 105.0 - 75.0                         # exponents
        if block % 2 == 0:
            e = torch.rand((n, 3), generator=g) * 12.0 - 6.0                       # ordinary magnitudes
        m = 1.0 + torch.rand((n, 3), generator=g)
        sgn = torch.where(torch.rand((n, 3), generator=g) < .5, -1.0, 1.0)
        v = (sgn * m * torch.pow(torch.tensor(2.0, dtype=torch.float64), e.double())).float()
        if block == 1:                                                             # special values in a tenth of the rows
            k = n // 10
            sp = torch.tensor([0.0, -0.0, float("inf"), float("-inf"), float("nan"), 1e-42, -1e-45, 3e38], dtype=torch.float32)
            idx = torch.randint(0, 8, (k,), generator=g)
            col = torch.randint(0, 3, (k,), generator=g)
            v[torch.arange(k), col] = sp[idx]
        fast = renderer.noise("wit_normalize", v.cuda()).cpu().numpy()
        ieee = renderer.noise("normalize", v.cuda()).cpu().numpy()
        rec = renderer.noise("wit_record", v.cuda()).cpu().numpy()[:, 0] != 0
        a = v.numpy()
        with np.errstate(all="ignore"):
            xx, yy, zz = a[:, 0] * a[:, 0], a[:, 1] * a[:, 1], a[:, 2] * a[:, 2]
            x = (xx + yy) + zz
            l = np.sqrt(x)
            ref = a / l[:, None]
        same = lambda p, q: (p.view(np.uint32) == q.view(np.uint32)) | (np.isnan(p) & np.isnan(q))
        assert same(ieee, ref).all(), "the device's IEEE normalize against numpy"
        ok = same(fast, ref).all(axis=1)
        assert ok[~rec].all(), (block, int((~ok & ~rec).sum()))
        mn = np.float32(2.0 ** -126)
        must = ~((xx >= mn) & (yy >= mn) & (zz >= mn) & (x >= np.float32(2.0 ** -102)) & (x < np.float32(2.0 ** 40)))
        assert (rec == must).all(), (block, int((rec != must).sum()))
        if block % 2 == 0:
            assert not rec.any()
