"""sbxbench.cpu — the cpu_baseline leg: the CPU oracle (test infrastructure, oracle/) timed on this host's cores AFTER the timed
region, on a bounded sample of the same frame.  Nothing here runs between the start and the end of a timed region."""
import os
import time

def cpu_rows(H, stride, cores, rows_per_s=None, target_s=12.0):
    """every stride-th row of the frame.  stride 0 = choose: from a measured rate (rows per second of this host, this app) so
    that the sample is ~target_s of wall time, else from the core count"""
    if stride <= 0:
        if rows_per_s:
            stride = max(1, min(16, int(H / max(rows_per_s * target_s, 1.0))))
        else:
            stride = 8 if cores <= 16 else (4 if cores <= 64 else 2)
    return stride, list(range(stride // 2, H, stride))


def host_cpu_facts():
    """what the threads of the CPU leg can actually get: scheduler affinity and the cgroup CPU quota of this process"""
    facts = {"os_cpu_count": os.cpu_count() or 1}
    try:
        facts["affinity"] = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        facts["affinity"] = None
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().strip()
        except OSError:
            continue
        if path.endswith("cpu.max"):
            quota = txt                                   # "max 100000" or "<quota_us> <period_us>"
        else:
            try:
                period = open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()
            except OSError:
                period = "?"
            quota = "%s %s" % (txt, period)
        break
    facts["cgroup_cpu_max"] = quota
    # CPUs this process can actually keep busy: the affinity mask capped by the cgroup quota (quota_us / period_us)
    eff = facts["affinity"] or facts["os_cpu_count"]
    try:
        q, per = (quota or "max 0").split()[:2]
        if q != "max" and float(q) > 0 and float(per) > 0:
            eff = max(1, min(eff, int(-(-float(q) // float(per)))))
    except ValueError:
        pass
    facts["effective_cpus"] = eff
    return facts


def _timed_rows(o, app_id, W, H, t, rows, cores):
    t0 = time.perf_counter()
    ref = o.render_rows(app_id, W, H, t, rows, threads=cores)
    dt = time.perf_counter() - t0
    # one thread on a few of the same rows: the per-thread rate the all-thread figure can be read against
    one_rows = rows[len(rows) // 2:len(rows) // 2 + 2]
    t0 = time.perf_counter()
    o.render_rows(app_id, W, H, t, one_rows, threads=1)
    dt1 = time.perf_counter() - t0
    return ref, dt, len(rows) * W / dt / 1e6, len(one_rows) * W / dt1 / 1e6, len(one_rows), dt1


def cpu_baseline(app, W, H, t, stride):
    """THE cpu_baseline of the line: the restatement of the reference's shader headers (oracle/ref_apps.h) over glibc's libm — the
    closest thing in this image to the author's C++ / VML build (/root/reference/src/Makefile:12-16: the headers compiled as C++
    against <cmath>) — strict flags, all the threads this process may keep busy, bounded sample.  Returns the object, the row
    indices and the rendered rows.  (Parity is NOT checked against these rows: the kernels are bit-compared with the sbx math
    spec's port, cpu_baseline_port.)"""
    from oracle.oracle import APP_IDS, Oracle
    o = Oracle(variant="_libm")
    facts = host_cpu_facts()
    cores = facts["effective_cpus"]                      # threads used = CPUs this process may run on AND is allowed to keep busy
    # calibration (also warms threads and caches): 8 rows spread over the frame -> rows per second -> a ~8 s sample of this
    # (the faster) variant, ~15 s of the port on the same rows
    cal = [int((k + .5) * H / 8) for k in range(8)]
    t0 = time.perf_counter()
    o.render_rows(APP_IDS[app], W, H, t, cal, threads=cores)
    stride, rows = cpu_rows(H, stride, cores, rows_per_s=len(cal) / max(time.perf_counter() - t0, 1e-6), target_s=8.0)
    ref, dt, value, one, n1, dt1 = _timed_rows(o, APP_IDS[app], W, H, t, rows, cores)
    return ({"value": round(value, 4), "unit": "Mpixels/s", "cores": cores, "kind": "port",
             "variant": "the restatement over glibc libm (oracle/libsbx_oracle_libm.so: sin / cos / exp / pow / acos / atan2 from "
                 "<cmath>), "
                        "g++ -O2 -ffp-contract=off",
             "sample": "%d of %d rows (every %dth row) of the same %dx%d frame in 64-pixel tiles, %.1f "
                 "s" % (len(rows), H, stride, W, H, dt),
             "affinity": facts["affinity"], "os_cpu_count": facts["os_cpu_count"], "cgroup_cpu_max": facts["cgroup_cpu_max"],
             "cores_is": "threads used = min(scheduler affinity, cgroup CPU quota rounded up)",
             "one_thread": {"value": round(one, 5), "unit": "Mpixels/s", "sample": "%d rows, %.1f s" % (n1, dt1)},
             "thread_equivalents": round(value / one, 1) if one > 0 else None,
             "note": "reproduces SURVEY.md Appendix C's survey-probe values to the printed digit (tests/test_oracle_kat_libm.py); within "
                     "1e-4 of the GPU frame, not bit-equal to it (libm's last bits are not the math spec's)"}, rows, ref)


def cpu_baseline_port(app, W, H, t, rows):
    """The sbx math spec's port (oracle/libsbx_oracle.so: binary64, correctly rounded transcendentals — what the HIP kernels are
    bit-compared with) on the SAME rows and threads.  Returns (object, rendered rows): the rows are the parity check's reference."""
    from oracle.oracle import APP_IDS, Oracle
    o = Oracle()
    facts = host_cpu_facts()
    cores = facts["effective_cpus"]
    o.render_rows(APP_IDS[app], W, H, t, rows[:max(1, cores // 60)], threads=cores)
    ref, dt, value, one, n1, dt1 = _timed_rows(o, APP_IDS[app], W, H, t, list(rows), cores)
    return ({"value": round(value, 4), "unit": "Mpixels/s", "cores": cores, "kind": "port",
             "variant": "the sbx math spec's port (binary64 transcendentals, correctly rounded): the parity oracle; about 2x slower per "
                        "thread than libm",
             "sample": "the same %d rows, %.1f s, g++ -O2 -ffp-contract=off" % (len(rows), dt),
             "one_thread": {"value": round(one, 5), "unit": "Mpixels/s", "sample": "%d rows, %.1f s" % (n1, dt1)}}, ref)


def cpu_baseline_speed(app, W, H, t, rows):
    """The same sample with the reference build's optimisation level (-O3 -march=native -funroll-loops), compiled HERE."""
    from oracle.oracle import APP_IDS, Oracle
    try:
        o = Oracle(variant="_speed", subdir="_speed", rebuild=True)
    except Exception:
        return None
    facts = host_cpu_facts()
    cores = facts["effective_cpus"]
    o.render_rows(APP_IDS[app], W, H, t, rows[:max(1, cores // 60)], threads=cores)
    t0 = time.perf_counter()
    o.render_rows(APP_IDS[app], W, H, t, rows, threads=cores)
    dt = time.perf_counter() - t0
    return {"value": round(len(rows) * W / dt / 1e6, 4), "unit": "Mpixels/s", "cores": cores, "kind": "port",
            "sample": "the same %d rows, %.1f s, g++ -O3 -march=native -funroll-loops (timing only: contraction allowed, "
                      "pixels not compared)" % (len(rows), dt)}
