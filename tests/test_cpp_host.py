"""The C++ host layer over the C ABI (host/): builds with plain g++; on a GPU the CLI's raw output and the
mainImage() drop-in loop reproduce the oracle."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "host")


@pytest.fixture(scope="module")
def built():
    from shaderbox_amd import build
    build.build(verbose=False)
    subprocess.run(["make", "-s", "-C", HOST, "clean"], check=True)
    subprocess.run(["make", "-s", "-C", HOST, "all", "APP=-DAPP_EGG"], check=True)
    return HOST


def test_hosts_build_and_fail_loudly_without_gpu(built):
    import torch
    assert os.path.exists(os.path.join(built, "sbx_render")) and os.path.exists(os.path.join(built, "mainimage_demo"))
    if not torch.cuda.is_available():
        r = subprocess.run([os.path.join(built, "sbx_render"), "--app", "egg", "--res", "32x32"], capture_output=True, text=True)
        assert r.returncode == 1 and "no CPU path" in r.stderr


def test_cli_rejects_bad_file_patterns(built):
    """--ppm / --f32 patterns are snprintf formats: anything but one %d / %0Nd conversion (and %%) is refused before any GPU work
    (`out_%s.ppm`, `%n`, two conversions were undefined behaviour)"""
    exe = os.path.join(built, "sbx_render")
    for bad in ("out_%s.ppm", "a%n.ppm", "f_%d_%d.ppm", "x%", "p%5.2f.ppm", "q%123456d.ppm"):
        r = subprocess.run([exe, "--app", "egg", "--res", "32x32", "--ppm", bad], capture_output=True, text=True)
        assert r.returncode == 2 and "bad file pattern" in r.stderr, (bad, r.returncode, r.stderr)
    for good in ("out_%04d.ppm", "plain.ppm", "100%%_%d.ppm"):
        r = subprocess.run([exe, "--app", "egg", "--res", "32x32", "--ppm", good], capture_output=True, text=True)
        assert "bad file pattern" not in r.stderr, good


@pytest.mark.gpu
def test_cli_and_mainimage_dropin_match_oracle(built, oracle, tmp_path):
    from oracle.oracle import APP_IDS
    out = str(tmp_path / "f.f32")
    for app, w, h, t in [("clouds", 160, 90, 0.37), ("APP_RAYTRACER", 96, 64, 2.5)]:
        subprocess.run([os.path.join(built, "sbx_render"), "--app", app, "--res", "%dx%d" % (w, h), "--time", str(t),
                        "--f32", out, "--ppm", str(tmp_path / "f.ppm")], check=True)
        got = np.fromfile(out, dtype=np.float32).reshape(h, w, 4)
        ref = oracle.render(APP_IDS[app.lower().replace("app_", "")], w, h, t)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
        # the PPM is the same frame through sbx_pack_unorm8 (R8G8B8A8_UNORM rule), top row first
        raw = open(str(tmp_path / "f.ppm"), "rb").read()
        hdr = ("P6\n%d %d\n255\n" % (w, h)).encode()
        assert raw.startswith(hdr) and len(raw) == len(hdr) + w * h * 3
        img = np.frombuffer(raw[len(hdr):], dtype=np.uint8).reshape(h, w, 3)
        c = ref[::-1, :, :3]
        want = (np.where(np.isnan(c) | ~(c > 0), 0, np.minimum(c, 1)).astype(np.float32) * np.float32(255) + np.float32(.5)).astype(np.uint8)
        assert np.array_equal(img, want)
    r = subprocess.run([os.path.join(built, "mainimage_demo"), "256", "256", "0.37"], capture_output=True, text=True, check=True)
    mean = [float(v) for v in r.stdout.strip().split("=")[-1].split()]
    # EGG 256x256 t=.37 frame mean
    assert np.allclose(mean, [0.401660, 0.545536, 0.539848], atol=2e-6), r.stdout


def test_inclxpnd_flattens_includes(built, tmp_path):
    (tmp_path / "sub").mkdir()
    (tmp_path / "a.h").write_text('#include "sub/b.h"\nint a;\n#include <math.h>\n  #  include "missing.h"\n')
    (tmp_path / "sub" / "b.h").write_text('#include "c.h"\nint b;\n')
    (tmp_path / "sub" / "c.h").write_text('int c;\n#include "b.h"\n')          # cycle back to b.h
    r = subprocess.run([os.path.join(built, "inclxpnd"), str(tmp_path / "a.h")], capture_output=True, text=True, check=True)
    assert r.stdout.splitlines() == ["int c;", "int b;", "int a;", "#include <math.h>", '  #  include "missing.h"']


def test_bench_cpu_baseline_leg(oracle):
    """the cpu_baseline object of bench.py (oracle timed on the host cores) on a tiny frame"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    cb, rows, ref = bench.cpu_baseline("egg", 64, 36, 0.37, 4)
    assert cb["kind"] == "port" and cb["unit"] == "Mpixels/s" and cb["cores"] >= 1 and cb["value"] > 0
    assert rows == list(range(2, 36, 4)) and ref.shape == (len(rows), 64, 4)
    # the parity record: identical rows -> 0 / 0; one flipped bit -> one mismatching pixel
    assert bench.parity(ref.copy(), ref, len(rows))["mismatching_pixels"] == 0
    bad = ref.copy()
    bad[1, 2, 0] = np.nextafter(bad[1, 2, 0], np.float32(2))
    rec = bench.parity(bad, ref, len(rows))
    assert rec["mismatching_pixels"] == 1 and 0 < rec["max_abs_diff"] < 1e-6
    sp = bench.cpu_baseline_speed("egg", 64, 36, 0.37, rows)
    assert sp is None or (sp["value"] > 0 and "march=native" in sp["sample"])
    assert set(bench.OPS_PER_PIXEL) == {"clouds", "egg", "raytracer", "atmosphere", "planet", "sdf_ao"}
    assert cb["one_thread"]["value"] > 0 and cb["thread_equivalents"] > 0 and cb["os_cpu_count"] >= 1 and "note" in cb
    assert 1 <= cb["cores"] <= (cb["affinity"] or cb["os_cpu_count"])


def test_bench_roofline_is_executed_work():
    """roofline.frac is the share of VALU issue slots used (<= 1 by construction, reproducible from the counters); the
    reference-algorithm ratio lives in useful_work_ratio"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    px = 3840 * 2160
    # the round-2 profile: 2.4935e9 instructions, GRBM_GUI_ACTIVE 5.1825e7 (8 XCDs), 3.0284 ms profiled, 2.745 ms un-profiled
    pmc = {"SQ_INSTS_VALU": 2.4935e9, "GRBM_GUI_ACTIVE": 5.1825e7, "kernel_ms_profiled": 3.0284, "WRITE_SIZE": 129600.0,
           "FETCH_SIZE": 187.0, "VALUBusy": 148.9, "source": "test"}
    r, h = bench.rooflines("clouds", px, px, 2.745, 2.72, pmc)
    # primary: against the guide's FIXED 78.64 T lane-ops/s with the profiled launch's own duration (VERDICT r4 Weak #5);
    # secondary: against the peak at the measured shader clock (the share of the issue slots of the cycles that happened)
    assert r["peak"] == 78.64 and abs(r["frac"] - 0.670) < 0.002 and abs(r["frac"] - r["achieved"] / r["peak"]) < 2e-3
    assert abs(r["frac_at_measured_clock"] - 0.752) < 0.002 and r["frac"] <= r["frac_at_measured_clock"] <= 1
    assert abs(r["useful_work_ratio"]["value"] - 1.157) < 0.005 and r["unit"] == "T lane-ops/s"
    assert abs(r["frac_unprofiled_duration"] - 0.739) < 0.003 and r["valu_busy_pct"] == 74.45
    assert r["traffic"] == int(129600 * 1024 + 2 * 187 * 1024) and h["traffic"] == r["traffic"]
    # a rank's eighth of the frame from the committed file: per-pixel count x pixels, nominal clock
    pmc["committed"] = True
    r8, _ = bench.rooflines("clouds", px // 8, px, 0.5, 0.5, pmc)
    assert r8["peak"] == 78.64 and 0 < r8["frac"] <= 1 and r8["frac"] == r8["frac_unprofiled_duration"] and r8["traffic"] is None
    # no counters at all: only the useful-work ratio
    r0, _ = bench.rooflines("egg", 1920 * 1080, 1920 * 1080, 0.26, 0.26, None)
    assert r0["frac"] is None and r0["useful_work_ratio"]["value"] > 0
    # round 6: the issue-cycle-weighted occupancy from the class counters (profiles/r05_apps_pmc.txt, k_clouds): classes x issue
    # cycles / (1024 SIMDs x active cycles); int32 and the un-named instructions priced at both ends
    cls = {"SQ_INSTS_VALU": 2161.02e6, "GRBM_GUI_ACTIVE": 45400700.0, "kernel_ms_profiled": 2.4068, "source": "test",
           "SQ_INSTS_VALU_ADD_F32": 528.788e6, "SQ_INSTS_VALU_MUL_F32": 811.079e6, "SQ_INSTS_VALU_FMA_F32": 133.884e6,
           "SQ_INSTS_VALU_TRANS_F32": 1.25469e6, "SQ_INSTS_VALU_CVT": 60.8648e6, "SQ_INSTS_VALU_INT32": 165.869e6,
           "SQ_INSTS_VALU_ADD_F64": 4.47525e6, "SQ_INSTS_VALU_MUL_F64": 10.0215e6, "SQ_INSTS_VALU_FMA_F64": 159.364e6,
           "SQ_INSTS_VALU_TRANS_F64": 259200.0}
    rw, _ = bench.rooflines("clouds", px, px, 2.39, 2.37, cls)
    iw = rw["issue_weighted"]
    assert abs(iw["frac_lo"] - 0.826) < 0.002 and abs(iw["frac_hi"] - 0.981) < 0.002 and iw["frac_lo"] < iw["frac_lo_at_measured_costs"] < iw["frac_hi_at_measured_costs"]
    assert sum(iw["classes"].values()) == round(cls["SQ_INSTS_VALU"]) and abs(rw["frac"] - 0.7307) < 0.002 and "frac_basis" not in rw
    # a sub-millisecond kernel profiled at idle clocks (EGG 1080p: 1.01 ms at 0.5 GHz): frac takes the un-profiled duration and says so
    egg = {"SQ_INSTS_VALU": 1.2049e8, "GRBM_GUI_ACTIVE": 4.0255e6, "kernel_ms_profiled": 1.0142, "source": "test"}
    re_, _ = bench.rooflines("egg", 1920 * 1080, 1920 * 1080, 0.199, 0.191, egg)
    assert abs(re_["frac"] - 0.4927) < 0.002 and re_["frac"] == re_["frac_unprofiled_duration"] and "0.50 GHz" in re_["frac_basis"]
    assert abs(re_["frac_profiled_pass"] - 0.0967) < 0.001 and re_["issue_weighted"] is None


@pytest.mark.gpu
def test_ddsvolgen_writes_the_volume(built, oracle, tmp_path):
    """host/sbx_ddsvolgen: DX10 volume DDS whose texels are the oracle's fbm_worley_tile volume"""
    import struct
    out = str(tmp_path / "n.dds")
    subprocess.run([os.path.join(built, "sbx_ddsvolgen"), "--size", "32", "--out", out], check=True)
    raw = open(out, "rb").read()
    assert len(raw) == 4 + 124 + 20 + 32 ** 3 * 16
    magic, size, flags, height, width, pitch, depth = struct.unpack_from("<7I", raw, 0)
    assert magic == 0x20534444 and size == 124 and (height, width, depth) == (32, 32, 32) and flags & 0x800000
    fourcc = struct.unpack_from("<I", raw, 4 + 72 + 8)[0]
    dxgi, dim = struct.unpack_from("<2I", raw, 4 + 124)
    assert fourcc == 0x30315844 and dxgi == 2 and dim == 4
    vox = np.frombuffer(raw, dtype=np.float32, offset=148).reshape(32, 32, 32, 4)
    ref = oracle.worley_volume(32, 0, 3)
    assert np.array_equal(vox[:3].view(np.uint32), ref.view(np.uint32))


@pytest.mark.gpu
def test_cli_animation_with_aux_flags_matches_oracle(built, oracle, tmp_path):
    """§8f-1: the run-time aux surface and the per-frame loop of the interactive host (util/hlsltoy/src/hlsltoy.cpp:
    466-491, 502-516) on the command line: a 3-frame sequence with non-default APP_CLOUDS aux values, every frame
    written, every frame equal to the oracle at u_time = time + f * dt."""
    import shaderbox_amd
    from oracle.oracle import APP_CLOUDS, APP_SDF_AO
    w, h, t0, dt = 160, 90, 0.25, 0.5
    subprocess.run([os.path.join(built, "sbx_render"), "--app", "clouds", "--res", "%dx%d" % (w, h), "--time", str(t0), "--dt", str(dt),
                    "--frames", "3", "--coverage", "0.6", "--thick", "100", "--steps", "60", "--light-steps", "4", "--sun", "0.2,0.5,-0.8",
                    "--wind", "0.1,0.02,0.2", "--sigma", "0.2", "--sun-power", "6", "--sun-color", "1,0.8,0.6", "--mouse", "1.5,0",
                    "--f32", str(tmp_path / "seq.f32"), "--ppm", str(tmp_path / "seq_%02d.ppm")], check=True)
    aux = shaderbox_amd.AuxClouds()
    shaderbox_amd.load_library().sbx_aux_clouds_defaults(aux)
    aux.cld_coverage, aux.cld_thick, aux.cld_march_steps, aux.illum_march_steps, aux.sigma_scattering, aux.sun_power = .6, 100., 60, 4, .2, 6.
    aux.sun_dir[0], aux.sun_dir[1], aux.sun_dir[2] = .2, .5, -.8
    aux.wind_dir[0], aux.wind_dir[1], aux.wind_dir[2] = .1, .02, .2
    aux.sun_color[0], aux.sun_color[1], aux.sun_color[2] = 1., .8, .6
    frames = []
    for f in range(3):
        got = np.fromfile(str(tmp_path / ("seq_%04d.f32" % f)), dtype=np.float32).reshape(h, w, 4)
        ref = oracle.render(APP_CLOUDS, w, h, np.float32(t0) + np.float32(f) * np.float32(dt), mouse=(1.5, 0.0), aux=aux)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), f
        assert os.path.getsize(str(tmp_path / ("seq_%02d.ppm" % f))) == len("P6\n%d %d\n255\n" % (w, h)) + w * h * 3
        frames.append(got)
    assert not np.array_equal(frames[0], frames[2])              # the wind moves the clouds
    # APP_SDF_AO's block
    subprocess.run([os.path.join(built, "sbx_render"), "--app", "sdf_ao", "--res", "96x54", "--time", "0.5", "--fog-density", "0.25",
                    "--fog-falloff", "0.3", "--f32", str(tmp_path / "ao.f32")], check=True)
    a2 = shaderbox_amd.AuxSdfAo()
    a2.fog_density, a2.fog_falloff = .25, .3
    got = np.fromfile(str(tmp_path / "ao.f32"), dtype=np.float32).reshape(54, 96, 4)
    assert np.array_equal(got.view(np.uint32), oracle.render(APP_SDF_AO, 96, 54, .5, aux=a2).view(np.uint32))


@pytest.mark.gpu
def test_cli_multi_gpu_and_noise_textures(built, oracle, tmp_path):
    """--gpus N shards the frame inside the library (on a 1-GPU box the ranks share the device); clouds_tex reads the
    .dds volumes sbx_ddsvolgen writes (the files hlsltoy takes as argv[2], argv[3])."""
    from oracle.oracle import APP_CLOUDS_TEX, APP_IDS
    w, h = 200, 117
    for app in ("clouds", "egg"):
        subprocess.run([os.path.join(built, "sbx_render"), "--app", app, "--res", "%dx%d" % (w, h), "--gpus", "3",
                        "--f32", str(tmp_path / "m.f32")], check=True)
        got = np.fromfile(str(tmp_path / "m.f32"), dtype=np.float32).reshape(h, w, 4)
        assert np.array_equal(got.view(np.uint32), oracle.render(APP_IDS[app], w, h, .37).view(np.uint32)), app
    vols = []
    for size, name in ((16, "shape.dds"), (8, "detail.dds")):
        subprocess.run([os.path.join(built, "sbx_ddsvolgen"), "--size", str(size), "--out", str(tmp_path / name)], check=True)
        raw = np.fromfile(str(tmp_path / name), dtype=np.uint8)
        vols.append(raw[4 + 124 + 20:].view(np.float32).reshape(size, size, size, 4).copy())
    oracle.set_noise_volumes(vols[0], vols[1])
    for extra in ([], ["--gpus", "2"]):
        subprocess.run([os.path.join(built, "sbx_render"), "--app", "clouds_tex", "--res", "160x90", "--noise-tex",
                        "%s,%s" % (tmp_path / "shape.dds", tmp_path / "detail.dds"), "--f32", str(tmp_path / "t.f32")] + extra, check=True)
        got = np.fromfile(str(tmp_path / "t.f32"), dtype=np.float32).reshape(90, 160, 4)
        assert np.array_equal(got.view(np.uint32), oracle.render(APP_CLOUDS_TEX, 160, 90, .37).view(np.uint32))
    r = subprocess.run([os.path.join(built, "sbx_render"), "--app", "clouds_tex", "--res", "32x18"], capture_output=True, text=True)
    assert r.returncode == 2 and "--noise-tex" in r.stderr


@pytest.mark.gpu
def test_cli_rgba8_writes_the_same_ppm(built, tmp_path):
    """--rgba8: the kernels write the 8-bit display format (SBX_FORMAT_RGBA8) — one GPU and the library's multi-GPU engine — and
    the .ppm is byte for byte the one of the float frame packed afterwards; --f32 is refused with it"""
    exe = os.path.join(built, "sbx_render")
    for app, extra in (("egg", []), ("atmosphere", ["--gpus", "3"]), ("clouds", ["--gpus", "2", "--exchange", "slabs"])):
        a, b = str(tmp_path / "a.ppm"), str(tmp_path / "b.ppm")
        subprocess.run([exe, "--app", app, "--res", "203x117", "--ppm", a] + extra, check=True)
        subprocess.run([exe, "--app", app, "--res", "203x117", "--ppm", b, "--rgba8"] + extra, check=True)
        assert open(a, "rb").read() == open(b, "rb").read(), (app, extra)
    r = subprocess.run([exe, "--app", "egg", "--res", "32x32", "--rgba8", "--f32", str(tmp_path / "x.f32")], capture_output=True, text=True)
    assert r.returncode == 2 and "--rgba8" in r.stderr


def test_bench_gpu_sampler_degrades_to_nothing_without_its_device():
    """bench.py's sustained leg samples sclk / power of the device it runs on from sysfs, found by PCI address; where there is no such
    device (this container) or no such file it reports None instead of failing — and never another GPU's numbers"""
    import importlib.util
    import time
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    smp = bench.GpuSampler(0, period_s=.01)
    with smp:
        time.sleep(.05)
    s = smp.summary()
    import torch
    if not torch.cuda.is_available():
        assert s["sclk_mhz"] is None and s["power_w"] is None and "None" in s["source"]
    else:
        assert s["sclk_mhz"] is None or s["sclk_mhz"]["samples"] >= 1
