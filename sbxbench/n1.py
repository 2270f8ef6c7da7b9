"""sbxbench.n1 — N = 1: the frame is one kernel launch.  The headline's timed region, the other BASELINE configs, the sustained leg."""
import os
import shutil
import tempfile
import time

from .common import KERNEL_OF, OTHER_CONFIGS, parity, steady_state
from .pmc import PMC_PASSES, pmc_committed, pmc_counters, rooflines, run_pmc_pass

class GpuSampler:
    """shader clock and board power of one GPU, sampled from sysfs by a thread (no subprocess per sample): pp_dpm_sclk's starred
    level or hwmon freq1_input, hwmon power1_average / power1_input.  What the box does not expose stays None."""

    def __init__(self, index=0, period_s=.02):
        import glob
        import threading
        self.period = period_s
        self.clk, self.pw = [], []
        # the card of HIP device `index` by its PCI address: a host shows every GPU (and their partitions) under /sys/class/drm,
        # this process is given one of them, and card0 is somebody else's as often as not
        self.pci = None
        try:
            import torch
            pr = torch.cuda.get_device_properties(index)
            self.pci = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        except Exception:                                  # noqa: BLE001  (no such attributes: nothing is sampled)
            pass
        cards = [c for c in glob.glob("/sys/class/drm/card[0-9]*/device")
                 if self.pci and os.path.realpath(c).lower().endswith(self.pci) and os.path.exists(os.path.join(c, "pp_dpm_sclk"))]
        self.dpm = os.path.join(cards[0], "pp_dpm_sclk") if cards else None
        base = os.path.dirname(self.dpm) if self.dpm else None
        hw = sorted(glob.glob(os.path.join(base, "hwmon", "hwmon*"))) if base else []
        self.freq = next((os.path.join(h, "freq1_input") for h in hw if os.path.exists(os.path.join(h, "freq1_input"))), None)
        self.power = next((os.path.join(h, n) for h in hw for n in ("power1_average", "power1_input") if os.path.exists(os.path.join(h, n))), None)
        self._stop = threading.Event()
        self._th = threading.Thread(target=self._run, daemon=True)

    def _read(self):
        mhz = None
        try:
            if self.freq:
                mhz = float(open(self.freq).read()) / 1e6
            elif self.dpm:
                for line in open(self.dpm):
                    if "*" in line:
                        mhz = float(line.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
        except (OSError, ValueError, IndexError):
            pass
        w = None
        try:
            if self.power:
                w = float(open(self.power).read()) / 1e6
        except (OSError, ValueError):
            pass
        return mhz, w

    def _run(self):
        while not self._stop.is_set():
            mhz, w = self._read()
            if mhz:
                self.clk.append(mhz)
            if w:
                self.pw.append(w)
            time.sleep(self.period)

    def __enter__(self):
        self._th.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self._th.join()

    @staticmethod
    def _stat(v, nd):
        return None if not v else {"mean": round(sum(v) / len(v), nd), "min": round(min(v), nd), "max": round(max(v), nd), "samples": len(v)}

    def summary(self):
        return {"sclk_mhz": self._stat(self.clk, 0), "power_w": self._stat(self.pw, 1),
                "source": "sysfs of PCI device %s: %s, %s" % (self.pci, self.freq or self.dpm, self.power)}


def sustained(torch, dev, step, ns, pixels, seconds, value, serial):
    """what the chip SUSTAINS: the timed region's loop (frames_in_flight launches overlapping) kept up for `seconds`, outside the
    timed region, with the shader clock and the board power sampled beside it.  `value` is K frames after a short warm-up; this is
    thousands of frames at whatever clock the power limit allows."""
    sampler = GpuSampler(dev.index or 0)                 # (finds the device's sysfs entries: tens of ms of host work, before the clock starts)
    for i in range(2 * ns):
        step(i)
    torch.cuda.synchronize(dev)
    n, t0, marks = 0, time.perf_counter(), []
    with sampler as smp:
        while time.perf_counter() - t0 < seconds:
            for i in range(8 * ns):
                step(i)
            torch.cuda.synchronize(dev)
            n += 8 * ns
            marks.append((time.perf_counter() - t0, n))
        dt = time.perf_counter() - t0
    ms = dt * 1e3 / n
    v = pixels / (ms * 1e-3) / 1e6

    def part(lo, hi):                                    # Mpixels/s of the batches that ended in [lo, hi] seconds
        inside = [(t, k) for t, k in marks if lo <= t <= hi]
        if len(inside) < 2:
            return None
        return round(pixels * (inside[-1][1] - inside[0][1]) / (inside[-1][0] - inside[0][0]) / 1e6, 3)
    out = {"value": round(v, 3), "unit": "Mpixels/s", "ms_per_step": round(ms, 4), "frames": n, "seconds": round(dt, 3),
           "frames_in_flight": ns, "first_half": part(0, dt / 2), "second_half": part(dt / 2, dt),
           "value_over_sustained": round(value / v, 4), "value_serial_over_sustained": round(serial / v, 4),
           "what": "the timed loop (same launches, same streams) held for %.1f s after the timed region, one synchronisation per %d "
               "frames; "
                   "sclk / power of THIS device (by PCI address) sampled every 20 ms from sysfs; first_half / second_half show whether "
                   "the rate drifts over seconds.  `value` (K frames after the pre-roll) within a per cent of this = the short window "
                   "measured the steady state" % (seconds, 8 * ns)}
    out.update(smp.summary())
    return out


def time_config(R, torch, dev, streams, app, W, H, t, steps=10, warmup=2, check_rows=0, pmc_mode="off", precision="exact"):
    """one config as the headline is measured: launches one at a time (`value`), `frames_in_flight` launches overlapping
    (`value_pipelined`), and the un-overlapped launch by HIP events (`kernel_ms`)"""
    if precision != "exact":
        R.set_precision(precision)
        try:
            out = time_config(R, torch, dev, streams, app, W, H, t, steps, warmup, check_rows, "off")
        finally:
            R.set_precision("exact")
        out["workload"] += " — OPT-IN TOLERANCE TIER SBX_PRECISION_1E4 (include/sbx.h: binary32 exp2 instead of the math spec's exp; within 1e-4 per channel, NOT bit-exact; never part of `value`)"
        out["precision"] = "1e-4"
        out["roofline"] = None
        return out
    ns = len(streams)
    frames = [torch.zeros((H, W, 4), dtype=torch.float32, device=dev) for _ in range(ns)]

    def step(i):                              # frames in flight: consecutive frames alternate over the streams
        with torch.cuda.stream(streams[i % ns]):
            R.render(app, W, H, t, out=frames[i % ns])

    def step1(i):                             # one launch at a time: one stream, one framebuffer
        with torch.cuda.stream(streams[0]):
            R.render(app, W, H, t, out=frames[0])
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(4):
        step(i)
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) * 1e3 / 4
    est = dt
    # a sub-millisecond kernel after seconds of host work (the previous config's oracle rows) starts at idle clocks: ~30 ms of
    # back-to-back launches first, so that neither figure below is the DVFS ramp's ...
    for _ in range(max(3, min(300, int(30.0 / max(est, .01))))):
        R.render(app, W, H, t, out=frames[0])
    torch.cuda.synchronize(dev)
    # ... and timed regions of at least ~20 ms: ten 0.15 ms frames are 1.5 ms, of which the ramp-in of the first launches and the
    # final synchronisation are a fifth (EGG 1080p read 0.150 ms per frame that way against 0.121 over 60 frames)
    steps = max(steps, min(400, int(20.0 / max(est, .01))))
    R.set_timing(False)                       # (the per-launch timing events are for the kernel_ms loop below)
    t0 = time.perf_counter()
    for i in range(steps):                    # the config's `value`: SURVEY.md 8d's form, launches one after the other
        step1(i)
    torch.cuda.synchronize(dev)
    ms = (time.perf_counter() - t0) * 1e3 / steps
    t0 = time.perf_counter()
    for i in range(steps):                    # ... and with `frames_in_flight` launches overlapping: `value_pipelined`
        step(i)
    torch.cuda.synchronize(dev)
    ms_pipe = (time.perf_counter() - t0) * 1e3 / steps
    R.set_timing(True)
    k = []
    for i in range(13):                       # median of >= 10 launches after 2 warm-ups (the first two are dropped)
        R.render(app, W, H, t, out=frames[0])
        k.append(R.last_kernel_ms())
    k = k[2:]
    torch.cuda.synchronize(dev)
    par = None
    if check_rows:
        # parity of this config in the same record: evenly spread full rows of the last rendered frame against the CPU oracle
        from oracle.oracle import APP_IDS, Oracle
        rows = sorted(set(int(round(i * (H - 1) / (check_rows - 1))) for i in range(check_rows)))
        ref = Oracle().render_rows(APP_IDS[app], W, H, t, rows)
        par = parity(frames[0][rows].cpu().numpy(), ref, len(rows))
    del frames
    kmean = sorted(k)[len(k) // 2]
    pmc = None
    if pmc_mode in ("auto", "live"):
        tmp = tempfile.mkdtemp(prefix="sbx_pmc_")
        try:
            pmc = run_pmc_pass(PMC_PASSES[0][1], app, W, H, t, os.path.join(tmp, "valu"))
            for name in ("cls32", "cls64"):              # the instruction mix (roofline.issue_weighted): two more passes
                more = run_pmc_pass(dict(PMC_PASSES)[name], app, W, H, t, os.path.join(tmp, name)) if pmc else None
                if more:
                    pmc.update({k: v for k, v in more.items() if k.startswith("SQ_INSTS_VALU_")})
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
        if pmc:
            pmc["source"] = "live: rocprofv3 --kernel-trace --pmc passes (instruction count, then the class counters) of `bench.py --app %s --steps 4 --warmup 1 --streams 1` in this run" % app
    if not pmc and pmc_mode != "off":
        pmc = pmc_committed(app, W, H)
    roofline, _ = rooflines(app, W * H, W * H, kmean, min(k), pmc)
    return {"workload": "APP_%s %dx%d u_time=%g" % (app.upper(), W, H, t), "value": round(W * H / (ms * 1e-3) / 1e6, 2),
            "unit": "Mpixels/s", "ms_per_step": round(ms, 4), "steps": steps,
            "value_is": "one launch at a time (back to back on one stream, wall clock / steps): SURVEY.md 8d's metric",
            "scene": "the canonical frame rendered K times — a scene that stands still, as BASELINE's configs are; from "
                     "the fourth launch on the library dispatches APP_CLOUDS / APP_EGG / APP_VINYL tiles by the cost "
                     "earlier launches measured (same pixels).  A scene that MOVES: APP_CLOUDS rebuilds that table behind "
                     "every launch (about -4 % instead of -7 %), APP_EGG keeps its own hot-first order (no gain) — "
                     "DESIGN.md 5.1",
            "value_pipelined": round(W * H / (ms_pipe * 1e-3) / 1e6, 2), "ms_per_step_pipelined": round(ms_pipe, 4), "frames_in_flight": ns,
            "kernel": KERNEL_OF.get(app), "kernel_ms": round(kmean, 4),
            "serial_value": round(W * H / (kmean * 1e-3) / 1e6, 2), "value_serial": round(W * H / (kmean * 1e-3) / 1e6, 2),
            "roofline": roofline,
            "hbm_store_gbps": round(16.0 * W * H / (kmean * 1e-3) / 1e9, 1), "parity": par}


def other_configs(R, torch, dev, streams, t, check_rows=16, pmc_mode="auto"):
    out = [time_config(R, torch, dev, streams, a, w, h, t, check_rows=check_rows, pmc_mode=pmc_mode) for a, w, h in OTHER_CONFIGS]
    # the labelled tolerance tier of APP_ATMOSPHERE, after the exact configs and never instead of one
    out.append(time_config(R, torch, dev, streams, "atmosphere", 7680, 4320, t, check_rows=check_rows, precision="1e-4"))
    return out


def bench_n1(args, R, torch, dev, streams, app, W, H, t):
    """N = 1: the JSON object of the headline and the exit status.  TWO timed regions of --steps frames each, both bracketed by a
    synchronize on both sides:
      1. one launch at a time, back to back on ONE stream                -> `value`, `ms_per_step`  (Mpixels/s per launch;
                                                                             the dominant kernel's time per step cannot exceed the step)
      2. --streams frames in flight (a frame's drain overlaps the next)   -> `value_pipelined`, `ms_per_step_pipelined`
    (Until round 5 region 2 WAS `value`; its 6-9 % over region 1 is overlap credit, not kernel time: VERDICT r5 #3.)"""
    from .cpu import cpu_baseline, cpu_baseline_port, cpu_baseline_speed
    ns = len(streams)
    status = 0
    frames = [torch.empty((H, W, 4), dtype=torch.float32, device=dev) for _ in range(ns)]

    def step(i=0):
        with torch.cuda.stream(streams[i % ns]):
            R.render(app, W, H, t, out=frames[i % ns])

    def step1(i=0):
        with torch.cuda.stream(streams[0]):
            R.render(app, W, H, t, out=frames[0])
    for f in frames:
        f.zero_()                                       # first touch of the framebuffers (page mapping) is not rendering
    torch.cuda.synchronize(dev)
    # pre-roll: the chip comes out of seconds of host work (imports, context, first touches) at idle clocks and needs ~25 ms of
    # launches to reach the clock it then holds; the driver's 5 warm-up frames are 11 ms.  Frames until --preroll-ms have passed.
    preroll_frames, t0 = 0, time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < args.preroll_ms:
        for i in range(ns):
            step(i)
        torch.cuda.synchronize(dev)
        preroll_frames += ns
    for i in range(3):                                  # the library applies its dispatch order from the fourth consecutive launch on one
        step1(i)                                        # stream: the pre-roll ends the way region 1 launches, whatever --warmup is
    torch.cuda.synchronize(dev)
    preroll_frames += 3
    # ---- timed region 1: the contract's K steps, one launch at a time.  (The per-launch timing events of sbx_set_timing are for the
    # kernel_ms loop below; inside the timed regions they would be two more packets per launch on the stream: off.)
    R.set_timing(False)
    for i in range(args.warmup):
        step1(i)
    torch.cuda.synchronize(dev)
    done1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        step1(i)
        done1[i].record(streams[0])
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    # the dominant kernel's time per step INSIDE this region: completion of frame 0 to completion of frame K - 1, over K - 1 launches
    span1 = done1[0].elapsed_time(done1[-1]) / max(args.steps - 1, 1) if args.steps > 1 else None
    # ---- timed region 2: the same K frames with ns in flight
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize(dev)
    step_done = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]     # completion of every timed frame
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
        step_done[i].record(streams[i % ns])
    torch.cuda.synchronize(dev)
    elapsed_pipe = time.perf_counter() - t0
    # per-launch kernel duration, HIP events on the launch stream (outside the timed regions, one launch at a time, so that the
    # event queries do not perturb them)
    R.set_timing(True)
    kernel_ms = []
    for _ in range(12):
        R.render(app, W, H, t, out=frames[0])
        kernel_ms.append(R.last_kernel_ms())
    kernel_ms = kernel_ms[2:]                            # median of >= 10 launches after 2 warm-ups
    R.set_timing(False)
    torch.cuda.synchronize(dev)
    kmean = sorted(kernel_ms)[len(kernel_ms) // 2]
    pixels = W * H
    ms_per_step = elapsed * 1e3 / args.steps
    ms_pipe = elapsed_pipe * 1e3 / args.steps
    value = pixels / (ms_per_step * 1e-3) / 1e6
    value_pipe = pixels / (ms_pipe * 1e-3) / 1e6
    pmc = pmc_counters(args, app, W, H, t) if args.pmc != "off" else None
    roofline, roofline_hbm = rooflines(app, pixels, pixels, kmean, min(kernel_ms), pmc)
    serial = round(pixels / (kmean * 1e-3) / 1e6, 3)
    out = {"metric": "Mpixels/s, APP_%s %dx%d" % (app.upper(), W, H), "value": round(value, 3),
           "unit": "Mpixels/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "APP_%s %dx%d u_time=%g u_mouse=0 default aux, fragCoord=(x+.5,y+.5)" % (app.upper(), W, H, t),
                      "frames_in_flight": 1, "parallelism": "1 GPU, one launch per frame, one launch at a time",
                      "preroll": "%d untimed frames (>= %g ms) before the warm-up steps" % (preroll_frames, args.preroll_ms)},
           "value_is": "SURVEY.md 8d: the K timed frames launched one at a time (back to back on one stream), wall clock between two "
                       "synchronisations / K",
           "scene": "the canonical frame rendered K times — a scene that stands still, as BASELINE's configs are; from "
                    "the fourth launch on the library dispatches APP_CLOUDS / APP_EGG / APP_VINYL tiles by the cost "
                    "earlier launches measured (same pixels).  A scene that MOVES: APP_CLOUDS rebuilds that table behind "
                    "every launch (about -4 % instead of -7 %), APP_EGG keeps its own hot-first order (no gain) — "
                    "DESIGN.md 5.1",
           # the same K frames with `frames_in_flight` launches overlapping (one framebuffer per stream: the drain of a frame's
           # last, longest waves overlaps the start of the next frame) — a throughput figure of independent frames, NOT the metric
           "value_pipelined": round(value_pipe, 3), "ms_per_step_pipelined": round(ms_pipe, 4), "frames_in_flight_pipelined": ns,
           # ... and one un-overlapped launch bracketed by HIP events (median of 10 after 2 warm-ups)
           "value_serial": serial,
           "serial": {"value": serial, "unit": "Mpixels/s", "what": "one un-overlapped launch (HIP events), %d pixels" % pixels},
           "kernel_ms_in_timed_region": round(span1, 4) if span1 else None,
           "steady_state": steady_state(step_done, ns, pixels),
           "roofline": roofline, "roofline_hbm": roofline_hbm}
    last_timed = frames[(args.steps - 1) % ns].clone() if not args.no_cpu_baseline else None
    if args.sustained_seconds > 0:
        out["sustained"] = sustained(torch, dev, step, ns, W * H, args.sustained_seconds, value_pipe, serial)
    if not args.no_cpu_baseline:
        base, rows, ref = cpu_baseline(app, W, H, t, args.cpu_row_stride)
        out["cpu_baseline"] = base
        # parity of the TIMED frame: the strict port's rows (what the kernels are bit-compared with) against the same rows of the GPU frame
        port, ref = cpu_baseline_port(app, W, H, t, rows)
        out["cpu_baseline_port"] = port
        gpu = last_timed[rows].cpu().numpy()
        del last_timed
        out["parity"] = parity(gpu, ref, len(rows))
        if not (out["parity"]["max_abs_diff"] <= 1e-4):
            status = 3
        speed = cpu_baseline_speed(app, W, H, t, rows)
        if speed is not None:
            out["cpu_baseline_speed"] = speed
    if not args.no_other_configs and app == "clouds":
        out["other_configs"] = other_configs(R, torch, dev, streams, t, check_rows=0 if args.no_cpu_baseline else 16,
                                             pmc_mode=args.pmc)
        if any(c["parity"] and not (c["parity"]["max_abs_diff"] <= 1e-4) for c in out["other_configs"]):
            status = 3
    return out, status
