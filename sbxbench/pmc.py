"""sbxbench.pmc — the rocprofv3 counter passes of a short serial run of bench.py, the committed per-launch counters they fall back to,
and the roofline objects of the JSON line (bench.py's docstring says what every field means)."""
import json
import os
import shutil
import subprocess
import sys
import tempfile

from .common import (BENCH, KERNEL_OF, LANES_PER_SIMD_CYCLE, N_SIMD, NOMINAL_CLOCK_HZ, OPS_PER_PIXEL, PEAK_FP32_VECTOR_TFLOPS, PEAK_HBM_GBPS,
                     PEAK_LANEOPS_NOMINAL_T, PMC_ROUND, ROOT, VALU_ISSUE_CYCLES)

PMC_PASSES = [("valu", ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVES", "GRBM_GUI_ACTIVE"]),
              ("busy", ["VALUBusy", "VALUUtilization"]),
              ("wr", ["WRITE_SIZE"]),
              ("rd", ["FETCH_SIZE"]),
              # the executed instruction MIX (roofline.issue_weighted): what the VALU pipes were busy with
              ("cls32", ["SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32",
                         "SQ_INSTS_VALU_CVT", "SQ_INSTS_VALU_INT32"]),
              ("cls64", ["SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"])]
CLASS_COUNTERS = PMC_PASSES[4][1] + PMC_PASSES[5][1]
# issue cycles of one wave64 VALU instruction on its SIMD-32 by class: the architecture's (MI355X_MICROARCH.md: 2-cycle issue; binary64,
# conversions and most non-arithmetic VALU ops at half rate; transcendentals at quarter rate) and the ones tools/ubench_issue.hip
# measured with 8 waves per SIMD (profiles/r02_ubench_issue.txt, "wall" column: they include the loop's own overhead)
ISSUE_CYCLES = {"arch": {"full": 2.0, "half": 4.0, "quarter": 8.0}, "measured": {"full": 2.25, "half": 4.2, "quarter": 8.2}}


def issue_weighted(pmc, active_cycles_per_simd):
    """roofline.issue_weighted (bench.py's docstring) from per-launch class counters; None without them"""
    if not pmc or any(c not in pmc for c in CLASS_COUNTERS) or "SQ_INSTS_VALU" not in pmc or not active_cycles_per_simd:
        return None
    g = lambda k: float(pmc["SQ_INSTS_VALU_" + k])          # noqa: E731
    full = g("ADD_F32") + g("MUL_F32") + g("FMA_F32")
    half = g("CVT") + g("ADD_F64") + g("MUL_F64") + g("FMA_F64")
    quarter = g("TRANS_F32") + g("TRANS_F64")
    int32 = g("INT32")
    other = max(0.0, float(pmc["SQ_INSTS_VALU"]) - full - half - quarter - int32)
    avail = N_SIMD * float(active_cycles_per_simd)
    out = {"classes": {"f32_add_mul_fma": round(full), "f64_and_cvt": round(half), "transcendental": round(quarter), "int32": round(int32),
                       "other_cmp_select_minmax_floor_mov": round(other)},
           "issue_cycles_per_class": ISSUE_CYCLES, "available_simd_cycles": round(avail),
           "what": "sum over classes of instructions x issue cycles / (1024 SIMDs x active cycles of the launch); int32 and the "
                   "instructions no class counter names are priced at full rate (frac_lo) and at half rate (frac_hi); fp32 "
                   "instructions with an SGPR source (half rate) are not told apart by the counters and count as full rate"}
    for tag, c in ISSUE_CYCLES.items():
        base = c["full"] * full + c["half"] * half + c["quarter"] * quarter
        lo, hi = base + c["full"] * (int32 + other), base + c["half"] * (int32 + other)
        sfx = "" if tag == "arch" else "_at_measured_costs"
        out["frac_lo" + sfx], out["frac_hi" + sfx] = round(lo / avail, 4), round(hi / avail, 4)
    return out


def run_pmc_pass(counters, app, W, H, t, outdir, timeout=100):
    """one rocprofv3 counter pass (kernel-trace + pmc only) over a short serial run of this script; returns
    {counter: mean over the dispatches of the app's render kernel}"""
    import csv
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    cmd = [exe, "--kernel-trace", "-f", "csv", "--pmc"] + counters + ["-d", outdir, "-o", "pmc", "--", sys.executable,
           BENCH, "--app", app, "--width", str(W), "--height", str(H), "--time", repr(t), "--steps", "4",
           "--warmup", "1", "--streams", "1", "--no-cpu-baseline", "--pmc", "off", "--no-other-configs"]
    env = dict(os.environ)
    env["TMPDIR"] = "/tmp"
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    try:
        r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
    except (subprocess.TimeoutExpired, OSError):
        return None
    if r.returncode != 0:
        return None
    kname = KERNEL_OF.get(app, "k_" + app)

    def mine(kn):
        return ("sbx::" + kname + "<") in kn or ("sbx::" + kname + "(") in kn
    rows = []
    for base, _, files in os.walk(outdir):
        for f in files:
            if f.endswith("counter_collection.csv"):
                rows += [row for row in csv.DictReader(open(os.path.join(base, f))) if mine(row.get("Kernel_Name", ""))]
    # only the full-frame launches count (the run also renders one 64x36 frame per stream while initialising)
    grid = max([float(row.get("Grid_Size", 0) or 0) for row in rows], default=0.0)
    acc = {}
    for row in rows:
        if float(row.get("Grid_Size", 0) or 0) == grid:
            acc.setdefault(row.get("Counter_Name", "?"), []).append(float(row.get("Counter_Value", "nan")))
    res = {c: sum(v) / len(v) for c, v in acc.items()}
    dur = []
    for base, _, files in os.walk(outdir):
        for f in files:
            if f.endswith("kernel_trace.csv"):
                for row in csv.DictReader(open(os.path.join(base, f))):
                    if mine(row.get("Kernel_Name", "")):
                        try:
                            dur.append((float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) * 1e-6)
                        except (KeyError, ValueError):
                            pass
    dur = [d for d in dur if d >= .5 * max(dur)] if dur else []
    if dur and "GRBM_GUI_ACTIVE" in res:
        res["kernel_ms_profiled"] = sum(dur) / len(dur)
    return res or None


def pmc_committed(app, W, H):
    """the committed per-launch counters of this app from THIS round's profile of the shipped kernels
    (profiles/<PMC_ROUND>_pmc_<app>_<W>x<H>.json): the file of this very frame size if there is one, else another size's (the
    instruction count PER PIXEL is resolution independent to < 1 %; `frame_pixels` says which frame the counters
    belong to and `other_size` flags it).  No fallback to an earlier round's files: counters of kernels that have since changed
    would overstate or understate the executed work.  None if there is none."""
    import glob
    import re
    exact = os.path.join(ROOT, "profiles", "%s_pmc_%s_%dx%d.json" % (PMC_ROUND, app, W, H))
    paths = [exact] if os.path.exists(exact) else sorted(glob.glob(os.path.join(ROOT, "profiles", "%s_pmc_%s_*x*.json" % (PMC_ROUND, app))))
    for path in paths:
        m = re.search(r"_(\d+)x(\d+)\.json$", path)
        if not m:
            continue
        got = {k: v for k, v in json.load(open(path)).items() if isinstance(v, (int, float))}
        got["source"] = "committed: profiles/" + os.path.basename(path) + ("" if path == exact else " (another frame size: per-pixel "
            "counts)")
        got["committed"] = True
        got["other_size"] = path != exact
        got["frame_pixels"] = int(m.group(1)) * int(m.group(2))
        return got
    return None


def pmc_counters(args, app, W, H, t):
    """{counter: per-launch mean} + 'source'.  live: rocprofv3 passes now; else the committed summary"""
    if args.pmc in ("auto", "live"):
        tmp = tempfile.mkdtemp(prefix="sbx_pmc_")
        got = {}
        try:
            for name, counters in PMC_PASSES:
                res = run_pmc_pass(counters, app, W, H, t, os.path.join(tmp, name))
                if res is None:
                    got = None
                    break
                got.update(res)
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
        if got:
            got["source"] = "live: rocprofv3 --kernel-trace --pmc passes of `bench.py --steps 4 --warmup 1 --streams 1` in this run"
            return got
        if args.pmc == "live":
            return None
    return pmc_committed(app, W, H)


def rooflines(app, launch_pixels, frame_pixels, kmean_ms, kmin_ms, pmc):
    """(roofline, roofline_hbm) of one launch of `launch_pixels` pixels.  `pmc`: per-launch counters of a FULL frame of
    `frame_pixels` pixels (live pass or committed file) or None."""
    kernel = KERNEL_OF.get(app, "k_" + app)
    ops = OPS_PER_PIXEL.get(app)
    hbm = 16.0 * launch_pixels / (kmean_ms * 1e-3) / 1e9
    roofline_hbm = {"bound": "hbm", "kernel": kernel, "achieved": round(hbm, 2), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                    "frac": round(hbm / PEAK_HBM_GBPS, 5), "bytes_per_pixel": 16, "traffic": None}
    r = {"bound": "valu", "kernel": kernel, "achieved": None, "peak": None, "unit": "T lane-ops/s", "frac": None,
         "frac_is": "executed work: VALU lane-operations issued (SQ_INSTS_VALU x 64) / the profiled launch's duration, against the FIXED "
                    "peak 1024 SIMD-32 x 32 lanes x 2.4 GHz = 78.64 T lane-ops/s (MI355X_MICROARCH.md); frac_at_measured_clock = the "
                    "same against the peak at the shader clock the launch actually ran at (the share of the issue slots that carried "
                    "an instruction, <= 1 by construction); frac_unprofiled_duration = the profiled instruction count over the "
                    "UN-profiled launch duration (HIP events) against the fixed peak",
         "frac_unprofiled_duration": None,
         "pixels_per_launch": launch_pixels, "kernel_ms": round(kmean_ms, 4), "kernel_ms_min": round(kmin_ms, 4),
         "traffic": None, "valu_busy_pct": None, "pmc_source": pmc.get("source") if pmc else None}
    if ops is not None:
        alg = ops * launch_pixels / (kmean_ms * 1e-3) / 1e12
        r["useful_work_ratio"] = {"value": round(alg / PEAK_FP32_VECTOR_TFLOPS, 5), "achieved": round(alg, 4),
                                  "peak": PEAK_FP32_VECTOR_TFLOPS, "unit": "TFLOP/s", "ops_per_pixel": ops,
                                  "vs_scalar_issue_ceiling": round(alg / 39.3, 5),       # 256 CU x 64 lanes x 2.4 GHz
                                  "what": "reference-algorithm scalar fp ops (SURVEY.md 8d) / un-overlapped launch time / fp32 vector "
                                          "peak: a speed-up measure, NOT utilisation (the kernel executes far fewer operations than "
                                          "the reference algorithm for the same bits, so it may exceed 1)"}
    if not pmc or "SQ_INSTS_VALU" not in pmc:
        return r, roofline_hbm
    frame_pixels = pmc.get("frame_pixels", frame_pixels)       # (a committed file may be of another frame size)
    scale = launch_pixels / float(frame_pixels)                  # counters are per FULL-frame launch
    insts = pmc["SQ_INSTS_VALU"] * scale
    r["valu_insts_per_launch"] = round(insts)
    r["valu_insts_per_pixel"] = round(pmc["SQ_INSTS_VALU"] / frame_pixels, 2)
    nominal = insts * 64.0 / (kmean_ms * 1e-3) / 1e12
    r["frac_unprofiled_duration"] = round(nominal / PEAK_LANEOPS_NOMINAL_T, 4)
    live = not pmc.get("committed") and scale == 1.0 and "GRBM_GUI_ACTIVE" in pmc and pmc.get("kernel_ms_profiled")
    if live:
        # PRIMARY: the instructions of the profiled launch / ITS duration (same rocprofv3 pass) against the guide's FIXED peak,
        # 1024 SIMD-32 x 32 lanes x 2.4 GHz = 78.64 T lane-ops/s — whatever clock DVFS actually held.
        # SECONDARY: the same against the peak at the MEASURED shader clock (GRBM_GUI_ACTIVE is summed over the 8 XCDs: / 8 =
        # shader cycles the launch was active) = the share of the issue slots of the cycles that happened.
        active = pmc["GRBM_GUI_ACTIVE"] / 8.0
        dur = pmc["kernel_ms_profiled"] * 1e-3
        clock = active / dur
        r["achieved"] = round(insts * 64.0 / dur / 1e12, 3)
        r["peak"] = round(PEAK_LANEOPS_NOMINAL_T, 2)
        r["frac"] = round(insts * 64.0 / dur / 1e12 / PEAK_LANEOPS_NOMINAL_T, 4)
        r["peak_at_measured_clock"] = round(N_SIMD * LANES_PER_SIMD_CYCLE * clock / 1e12, 3)
        r["frac_at_measured_clock"] = round(insts * VALU_ISSUE_CYCLES / (N_SIMD * active), 4)
        r["shader_clock_ghz_profiled"] = round(clock / 1e9, 3)
        r["kernel_ms_profiled"] = round(pmc["kernel_ms_profiled"], 4)
        r["issue_weighted"] = issue_weighted(pmc, active)
        if clock < 0.85 * NOMINAL_CLOCK_HZ:
            # a sub-millisecond kernel under the profiler: every launch waits for its counters to be read, the chip idles between
            # launches and the profiled launch ran at a fraction of the clock (EGG 1080p: 1.01 ms at 0.50 GHz against 0.20 ms
            # unprofiled).  Instruction counts do not depend on the clock: the primary figure is then the profiled count over the
            # UN-profiled launch duration (HIP events), and says so.
            r["frac_profiled_pass"], r["achieved_profiled_pass"] = r["frac"], r["achieved"]
            r["achieved"], r["frac"] = round(nominal, 3), r["frac_unprofiled_duration"]
            r["frac_basis"] = ("profiled instruction count / UN-profiled launch duration: the profiled launch ran at %.2f GHz (idle clocks "
                               "between counter reads)" % (clock / 1e9))
    else:
        # no counters of THIS launch (rocprofv3 unusable, or a rank's strip at N > 1): the committed profile's instruction count
        # per pixel x this launch's pixels, against the nominal-clock peak
        r["achieved"] = round(nominal, 3)
        r["peak"] = round(PEAK_LANEOPS_NOMINAL_T, 2)
        r["frac"] = r["frac_unprofiled_duration"]
        r["frac_is"] += "; here from the committed per-pixel instruction count x this launch's pixels"
        if pmc.get("GRBM_GUI_ACTIVE") and scale == 1.0:
            r["issue_weighted"] = issue_weighted(pmc, pmc["GRBM_GUI_ACTIVE"] / 8.0)
            if r["issue_weighted"]:
                r["issue_weighted"]["what"] += "; counters AND active cycles from the committed profile of this kernel"
    if "WRITE_SIZE" in pmc and "FETCH_SIZE" in pmc and scale == 1.0:      # KB; gfx950: FETCH_SIZE counts half of a wide streaming read
        traffic = int(pmc["WRITE_SIZE"] * 1024 + 2 * pmc["FETCH_SIZE"] * 1024)
        r["traffic"] = roofline_hbm["traffic"] = traffic
        r["traffic_over_algorithmic"] = round(traffic / (16.0 * launch_pixels), 4)
    if "VALUBusy" in pmc:
        r["valu_busy_pct"] = round(pmc["VALUBusy"] / 2.0, 2)      # gfx94x formula assumes 4-cycle issue; gfx950 issues in 2
        r["valu_busy_pct_raw_rocprof"] = round(pmc["VALUBusy"], 2)
    if "VALUUtilization" in pmc:
        r["valu_lane_utilization_pct"] = round(pmc["VALUUtilization"], 2)
    return r, roofline_hbm
