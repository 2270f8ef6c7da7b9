"""sbxbench.dist — N > 1: one process per GPU (torch.distributed / RCCL), the SAME frame sharded as cyclic row-blocks through
shaderbox_amd.distributed.FramePlan; and --engine lib (one process drives the GPUs through the library's own multi-GPU path)."""
import json
import os
import time

from shaderbox_amd import tuning

from .common import claim_stdout, steady_state
from .pmc import pmc_committed, rooflines

def rank_launch_pixels(R, app, W, H, t, br, world, rank, relief, exchange):
    """pixels the launch(es) of `rank` render per frame"""
    from shaderbox_amd import shard
    if exchange not in ("spans", "span_stores", "packed_stores") or world == 1:
        return shard.rank_rows(H, br, rank, world, relief[0], relief[1]) * W
    table, pix, _ = R.span_table(app, W, H, t, br, world, relief[0], relief[1])
    if rank > 0:
        return int(pix[rank])
    own = shard.rank_rows(H, br, 0, world, relief[0], relief[1]) * W
    outside = sum((min(H, (g + 1) * br) - g * br) * (W - int(x1 - x0)) for g, (x0, x1, _, owner) in enumerate(table) if owner > 0)
    return own + outside


def dist_frame_bench(R, dist, torch, dev, streams, args, app, W, H, t, world, rank, steps, warmup):
    """relief calibration, plans, first touch, warm-up, the timed K frames (barrier + synchronize on both sides, MAX over ranks),
    every rank's un-overlapped launch, the phases of serial frames, the assembled frame against one launch.  Collective: every
    rank calls it; the returned dict is complete on rank 0."""
    ns = len(streams)
    br = args.block_rows
    fdist = dist
    if args.backend == "gloo":
        from shaderbox_amd.distributed import HostStagedDist
        fdist = HostStagedDist(dist, torch)

    def sync():
        dist.barrier()
        torch.cuda.synchronize(dev)

    # Which exchange form, how much relief for the root, how many pieces: the product's own tuning (shaderbox_amd/tuning.py
    # choose_exchange — every candidate is set up on the ranks at hand, soaked against one-launch renders, timed with the frames in
    # flight, and the fastest runs; the phase is bounded by --trial-budget-s).  bench.py adds nothing to it.
    if not hasattr(args, "channels_asked"):
        args.channels_asked = args.channels              # (auto overwrites args.channels with what it chose: later configs start from the flag again)
    choice = tuning.choose_exchange(R, dist, torch, dev, streams, app, W, H, t, world=world, rank=rank, block_rows=br,
                                    exchange=args.exchange, channels=args.channels_asked, root_rounds=args.root_rounds,
                                    groups=args.gather_groups, fdist=fdist if fdist is not dist else None,
                                    budget_s=args.trial_budget_s, preroll_ms=args.preroll_ms)
    exchange, relief, payload, groups, plans, trials = (choice.exchange, choice.relief, choice.payload_bytes_per_peer, choice.groups,
                                                        choice.plans, choice.trials)
    args.channels = choice.channels
    # who the ranks are: one entry per rank (collective), and — under RCCL — the library's version and the communicator's size, so
    # that "N ranks over RCCL on N devices" is answerable from the record
    import socket
    props = torch.cuda.get_device_properties(dev)
    ident = [None] * world
    dist.all_gather_object(ident, {"rank": rank, "device": int(dev.index or 0), "pci_bus_id": getattr(props, "pci_bus_id", None),
                                   "host": socket.gethostname()})
    rccl = {"version": None, "nranks": None, "devices": ident}
    if args.backend == "nccl":
        try:
            rccl["version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:                                # noqa: BLE001
            rccl["version"] = "?"
        rccl["nranks"] = dist.get_world_size()

    def step(i=0):
        with torch.cuda.stream(streams[i % ns]):
            plans[i % ns].render(app, t)              # the rank's launch(es) + the ONE exchange + assembly on rank 0

    for i in range(ns):                                 # builds the span layout, touches every buffer (page mapping)
        step(i)
    sync()
    # pre-roll, as at N = 1: frames until --preroll-ms have passed on rank 0 (every rank runs the same count) — the clocks, and the
    # first use of a form's mappings (a form asked for with --exchange has had no trial: its first frames read 2-4x slow)
    npre = torch.zeros(1, dtype=torch.int64, device=(tuning.CONFIG.coll_dev or dev))
    if rank == 0:
        t_pre, k_pre = time.perf_counter(), 0
        for i in range(ns):
            step(i)
        torch.cuda.synchronize(dev)
        one = max((time.perf_counter() - t_pre) / ns, 1e-5)
        npre[0] = max(0, min(400, int(args.preroll_ms * 1e-3 / one) - ns))
    else:
        for i in range(ns):
            step(i)
    dist.broadcast(npre, src=0)
    for i in range(int(npre.item())):
        step(i)
    sync()
    # TWO timed regions of `steps` frames, as at N = 1 (barrier + synchronize on both sides, MAX over ranks):
    #   1. one frame at a time — every rank's launch(es), the exchange and the assembly of frame k on ONE stream before frame k + 1
    #      starts there -> `value_one_at_a_time`: a frame's LATENCY through the whole pipeline (N = 1's `value` is this form)
    #   2. ns frames in flight (one plan per stream) -> `value`, `ms_per_step` (= `value_pipelined`): the sequence's THROUGHPUT
    def step1(i=0):
        with torch.cuda.stream(streams[0]):
            plans[0].render(app, t)
    R.set_timing(False)                                 # (per-launch timing events: for the un-overlapped launches below only)
    for i in range(3):                                  # (pre-roll of region 1: the library applies its dispatch order from the fourth
        step1(i)                                        # consecutive launch on one stream, DESIGN 5.1 — whatever --warmup is)
    for i in range(warmup):
        step1(i)
    sync()
    t0 = time.perf_counter()
    for i in range(steps):
        step1(i)
    sync()
    elapsed = time.perf_counter() - t0
    for i in range(warmup):
        step(i)
    sync()
    step_done = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
        step_done[i].record(streams[i % ns])
    sync()
    elapsed_pipe = time.perf_counter() - t0
    # every rank's own launch, un-overlapped
    R.set_timing(True)
    km = []
    frame0 = plans[0].frame
    scratch = None
    for _ in range(min(max(steps, 3), 8)):
        if exchange in ("spans", "span_stores", "packed_stores") and world > 1:
            if rank == 0:
                R.render_span_root(app, W, H, t, br, world, frame0, root_rounds=relief[0], rounds=relief[1])
            elif exchange == "span_stores":             # (in place into the owner's frame: the same pixels it holds already)
                R.render_span_peer_in_place(app, W, H, t, br, rank, world, plans[0].shared, root_rounds=relief[0], rounds=relief[1], channels=args.channels)
            else:
                R.render_span_peer(app, W, H, t, br, rank, world, 0, 1 << 30, plans[0].slab, root_rounds=relief[0], rounds=relief[1])
        else:
            if scratch is None:
                scratch = torch.empty((plans[0].rows_max, W, 4), dtype=torch.float32, device=dev)
            R.render_rank(app, W, H, t, br, rank, world, out=scratch, root_rounds=relief[0], rounds=relief[1])
        km.append(R.last_kernel_ms())
    del scratch
    sync()
    mine = torch.tensor([elapsed, sum(km) / len(km), min(km), float(rank_launch_pixels(R, app, W, H, t, br, world, rank, relief, exchange)),
                         elapsed_pipe],
                        dtype=torch.float64, device=tuning.CONFIG.coll_dev or dev)
    allr = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allr, mine)
    phases = dist_phases(plans[0], torch, dist, dev, app, t, world, rank)
    res = {"relief": relief, "groups": groups, "payload_bytes_per_peer": payload, "ns": ns, "steps": steps, "warmup": warmup,
           "exchange": exchange, "exchange_trials_ms": trials, "exchange_notes": choice.notes, "rccl": rccl}
    if rank == 0:
        per = [[float(x) for x in v] for v in allr]
        slow = max(range(world), key=lambda r: per[r][1])
        res.update({"elapsed": max(p[0] for p in per), "elapsed_pipe": max(p[4] for p in per), "kmean": per[slow][1], "kmin": per[slow][2], "launch_pixels": int(per[slow][3]),
                    "slowest_rank": slow, "per_rank_launch_ms": [round(p[1], 4) for p in per],
                    "steady": steady_state(step_done, ns, W * H), "phases": phases})
        # the assembled frame of the multi-GPU path against a one-launch render of the same frame: same bits
        whole = R.render(app, W, H, t)
        frame = plans[(steps - 1) % ns].frame
        res["mismatching_pixels"] = int((frame.view(torch.int32) != whole.view(torch.int32)).any(dim=-1).sum().item())
        res["frame"] = frame
        del whole
    res["plans"] = plans
    return res


def dist_line(res, args, app, W, H, t, world):
    """rank 0: the JSON object of one N > 1 measurement"""
    pixels = W * H
    relief, ns = res["relief"], res["ns"]
    ms_per_step = res["elapsed"] * 1e3 / res["steps"]
    ms_pipe = res["elapsed_pipe"] * 1e3 / res["steps"]
    pmc = pmc_committed(app, W, H) if args.pmc != "off" else None
    roofline, roofline_hbm = rooflines(app, res["launch_pixels"], pixels, res["kmean"], res["kmin"], pmc)
    if roofline is not None:
        roofline["rank"] = "slowest (rank %d of the un-overlapped launches %s ms; %d pixels)" % (res["slowest_rank"], res["per_rank_launch_ms"],
                                                                                              res["launch_pixels"])
        if res["exchange"] in ("spans", "span_stores", "packed_stores") and world > 1 and roofline.get("frac") is not None:
            roofline["frac_is"] += ("; NOTE a span launch renders mostly the frame's EXPENSIVE pixels, so the frame-average instruction "
                                    "count per pixel understates its work: read this frac as a lower bound")
    ph = res["phases"]
    serial_ms = max((p["render_ms"] + p["exchange_wait_ms"] + p["assemble_ms"]) for p in ph["per_rank"]) if ph else None
    exch = {"direct": "1 grouped RCCL send/recv of the peers' %d-channel slabs to the root (root in place)" % args.channels,
            "gather": "1 RCCL gather of RGBA slabs",
            "stores": "the peers' own %d-byte pixel stores into the root's frame, mapped through HIP IPC (no RCCL call, no landing area, "
                      "no scatter; two flag kernels per rank and frame)" % (12 if args.channels == 3 else 16),
            "span_stores": "the peers' own %d-byte pixel stores of the SPANS of their row-blocks into the root's frame, mapped through "
                "HIP IPC "
                           "(the root renders its blocks and everything outside the spans; no RCCL call, no landing area, no scatter)"
                           % (12 if args.channels == 3 else 16),
            "packed_stores": "the peers' own stores of the packed 3-channel SPANS of their row-blocks (12 contiguous bytes per pixel) "
                "straight "
                             "into the root's landing area, mapped through HIP IPC; the root renders its blocks and everything outside the "
                             "spans, then scatters (no RCCL call, no receive kernels on the root)",
            "spans": "1 grouped RCCL send/recv of the peers' packed 3-channel SPANS (the root renders its blocks and everything "
                     "outside the spans in place)"}[res["exchange"]]
    if args.backend != "nccl":
        exch = exch.replace("RCCL", "gloo (host-staged, TEST form)")
    # north_star asks for "a single RCCL gather over xGMI": whatever form `value` ran with, the RCCL forms' PIPELINED figures of the SAME
    # ranks (compare with `value_pipelined`) are first-class keys — from the trial phase when the form was chosen by trial (ms per pipelined frame of 12), from the timed
    # region itself when the run's form is that RCCL form; null under gloo (a test transport) or when the form was not tried
    trials = res.get("exchange_trials_ms") or {}

    def rccl_value(form):
        if args.backend != "nccl":
            return None
        if res["exchange"] == form:
            return round(pixels / (ms_pipe * 1e-3) / 1e6, 3)          # (like the trials: frames in flight)
        ms = trials.get(form)
        return round(pixels / (ms * 1e-3) / 1e6, 3) if isinstance(ms, (int, float)) and ms > 0 else None
    out_rccl = dict(res.get("rccl") or {})
    out_rccl["in_timed_region"] = args.backend == "nccl" and res["exchange"] in ("spans", "direct", "gather")
    # `value` at N > 1 is the THROUGHPUT of the frame sequence (ns frames in flight, one plan per stream), not a frame's latency: at N > 1
    # a frame contains the exchange (9-12 MB over one link at 4K: ~0.2 ms beside a 0.28-0.39 ms strip), which every host of a frame
    # sequence overlaps with the next frame's rendering; the latency form is `value_one_at_a_time`.  At N = 1 there is no exchange and the
    # two forms agree to 0.3 % (its `value` is the one-at-a-time form; `value_pipelined` beside it).  DESIGN.md 8.
    return {"metric": "Mpixels/s, APP_%s %dx%d" % (app.upper(), W, H), "value": round(pixels / (ms_pipe * 1e-3) / 1e6, 3),
            "unit": "Mpixels/s", "n_gpus": world, "steps": res["steps"], "warmup": res["warmup"],
            "ms_per_step": round(ms_pipe, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "value_is": "THROUGHPUT of the K timed frames with %d in flight (one plan per stream: every rank's launches, the exchange and "
                        "the assembly of frame k overlap frame k + 1's rendering), barrier + synchronize on both sides, slowest rank.  A "
                        "frame's LATENCY through the whole pipeline (the K frames one at a time) is `value_one_at_a_time`: at N > 1 it "
                        "contains the exchange, which a frame sequence hides; at N = 1 the two forms agree to 0.3 %%" % ns,
            "value_pipelined": round(pixels / (ms_pipe * 1e-3) / 1e6, 3), "ms_per_step_pipelined": round(ms_pipe, 4),
            "frames_in_flight_pipelined": ns,
            "value_one_at_a_time": round(pixels / (ms_per_step * 1e-3) / 1e6, 3), "ms_per_step_one_at_a_time": round(ms_per_step, 4),
            "config": {"workload": "APP_%s %dx%d u_time=%g u_mouse=0 default aux, fragCoord=(x+.5,y+.5)" % (app.upper(), W, H, t),
                       "frames_in_flight": ns,
                       "parallelism": "cyclic %d-row blocks over %d GPUs (root sits out rounds >= %d of %d) + %s (in %d pipelined "
                                      "pieces)%s" % (args.block_rows, world, relief[0], relief[1], exch, res["groups"],
                                                     "" if res["exchange"] in ("stores", "span_stores") else " + assemble")},
            "backend": "RCCL" if args.backend == "nccl" else "gloo with host-staged transfers (TEST form: ranks may share a GPU, nothing "
                "here "
                                                                "says anything about xGMI)",
            "value_rccl_spans": rccl_value("spans"), "value_rccl_direct": rccl_value("direct"), "rccl": out_rccl,
            "exchange": {"kind": res["exchange"], "notes": res.get("exchange_notes"), "chosen": "measured on these ranks: ms per "
                "pipelined frame %s" % res["exchange_trials_ms"]
                         if res.get("exchange_trials_ms") else "as asked (--exchange)" if args.exchange != "auto" else "one rank: nothing to choose",
                         "bytes_per_peer": res["payload_bytes_per_peer"], "pieces": res["groups"],
                         "link_ms_at_76p8_GBps": round(res["payload_bytes_per_peer"] / 76.8e9 * 1e3, 4),
                         "what": "the largest peer payload of one frame; one xGMI link per peer, 76.8 GB/s per direction at its peak"},
            "value_serial": round(pixels / (serial_ms * 1e-3) / 1e6, 3) if serial_ms else None,
            "serial": {"value": round(res["launch_pixels"] / (res["kmean"] * 1e-3) / 1e6, 3), "unit": "Mpixels/s",
                       "what": "the slowest rank's un-overlapped launch (HIP events), %d pixels; value_serial = the frame's pixels / "
                               "one serial frame of the whole pipeline (render + exchange wait + assemble on the root, `phases`)"
                               % res["launch_pixels"]},
            "steady_state": res["steady"], "roofline": roofline, "roofline_hbm": roofline_hbm, "phases": ph,
            "parity": {"against": "one-launch render of the same frame on rank 0", "rows": H,
                       "mismatching_pixels": res["mismatching_pixels"]}}


def bench_lib(args):
    """--engine lib: one process, N ranks inside the library (sbx_multi_*)."""
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import shaderbox_amd
    ndev = torch.cuda.device_count()
    n = args.gpus
    devices = list(range(n)) if ndev >= n else [i % max(ndev, 1) for i in range(n)]
    M = shaderbox_amd.MultiRenderer(devices)
    m0, m = (1, 1) if args.root_rounds == "auto" else tuple(int(v) for v in args.root_rounds.split("/"))
    M.set_split(args.block_rows, m0, m)
    M.set_exchange(args.lib_exchange)
    W, H, app, t = args.width, args.height, args.app, args.time
    dev = torch.device("cuda", devices[0])
    torch.cuda.set_device(dev)
    ns = max(1, min(2, args.streams))
    streams = [torch.cuda.Stream(device=dev) for _ in range(ns)]
    frames = [torch.zeros((H, W, 4), dtype=torch.float32, device=dev) for _ in range(ns)]

    def step(i):
        with torch.cuda.stream(streams[i % ns]):
            M.render(app, W, H, t, out=frames[i % ns])

    def sync():
        for d in sorted(set(devices)):
            torch.cuda.synchronize(d)
    for i in range(2):
        step(i)                       # one-time initialisation: code objects, y tables, peer links, slabs
    sync()
    for i in range(args.warmup):
        step(i)
    sync()
    step_done = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
        step_done[i].record(streams[i % ns])
    sync()
    elapsed = time.perf_counter() - t0
    R = shaderbox_amd.Renderer(devices[0])
    R.set_timing(True)
    whole = R.render(app, W, H, t)
    a, b = frames[(args.steps - 1) % ns].view(torch.int32), whole.view(torch.int32)
    bad = int((a != b).any(dim=-1).sum().item())
    # the slowest rank's un-overlapped launch (rank 1 has the most rows of a plain split)
    from shaderbox_amd import shard
    rows = [shard.rank_rows(H, args.block_rows, r, n, m0, m) for r in range(n)]
    slow = max(range(n), key=lambda r: rows[r])
    slab = torch.empty((shard.rank_rows_max(H, args.block_rows, n, m0, m), W, 4), dtype=torch.float32, device=dev)
    km = []
    for _ in range(5):
        R.render_rank(app, W, H, t, args.block_rows, slow, n, out=slab, root_rounds=m0, rounds=m)
        km.append(R.last_kernel_ms())
    torch.cuda.synchronize(dev)
    roofline, roofline_hbm = rooflines(app, rows[slow] * W, W * H, sum(km) / len(km), min(km),
                                       pmc_committed(app, W, H) if args.pmc != "off" else None)
    if roofline is not None:
        roofline["rank"] = "slowest (rank %d: %d rows), one un-overlapped launch on device %d" % (slow, rows[slow], devices[0])
    ms_per_step = elapsed * 1e3 / args.steps
    steady = steady_state(step_done, ns, W * H)
    out = {"metric": "Mpixels/s, APP_%s %dx%d" % (app.upper(), W, H), "value": round(W * H / (ms_per_step * 1e-3) / 1e6, 3),
           "unit": "Mpixels/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "APP_%s %dx%d u_time=%g u_mouse=0 default aux, fragCoord=(x+.5,y+.5)" % (app.upper(), W, H, t),
                      "frames_in_flight": ns, "engine": "lib (one process, sbx_multi_*)",
                      "parallelism": "cyclic %d-row blocks over %d ranks on devices %s, %s, root renders in place"
                                     % (args.block_rows, n, devices,
                                        ("%s, %s" % ("RCCL send/recv" if M.uses_rccl else
                                                     "device copies (ranks share devices: emulation, not a scaling number)",
                                                     {"slabs": "one per peer of its whole 3-channel slab + one scatter kernel",
                                                      "spans": "one per peer of its packed 3-channel spans + one scatter kernel, rank 0 "
                                                          "renders the rest",
                                                      "peer_stores": "none: every rank stores its pixels into rank 0's frame through peer "
                                                          "access",
                                                      "blocks": "one per row-block into the final rows"}[args.lib_exchange])))},
           "steady_state": steady, "roofline": roofline, "roofline_hbm": roofline_hbm,
           "parity": {"against": "one-launch render of the same frame", "rows": H, "mismatching_pixels": bad}}
    status = 3 if bad else 0
    if not args.no_cpu_baseline:
        from .common import parity
        from .cpu import cpu_baseline, cpu_baseline_port
        base, crow, _ = cpu_baseline(app, W, H, t, args.cpu_row_stride)
        out["cpu_baseline"] = base
        out["cpu_baseline_port"], ref = cpu_baseline_port(app, W, H, t, crow)
        out["parity"]["oracle"] = parity(frames[(args.steps - 1) % ns][crow].cpu().numpy(), ref, len(crow))
        if not (out["parity"]["oracle"]["max_abs_diff"] <= 1e-4):
            status = 3
    claim_stdout()(json.dumps(out))
    M.close()
    return status


def dist_phases(plan, torch, dist, dev, app, t, world, rank, reps=5):
    """N > 1: per-rank render_ms / exchange_wait_ms / assemble_ms of SERIAL frames (events on the frame's stream, one frame at a
    time, outside the timed region), gathered to rank 0: what a rank's frame consists of when nothing overlaps it."""
    acc = {}
    for _ in range(reps):
        marks = []

        def mark(name):
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream(dev))
            marks.append((name, ev))
        dist.barrier()
        torch.cuda.synchronize(dev)
        mark("start")
        h0 = time.perf_counter()
        plan.render(app, t, mark=mark)
        acc.setdefault("host", []).append((time.perf_counter() - h0) * 1e3)      # what the host thread spends submitting one frame
        mark("end")
        torch.cuda.synchronize(dev)
        for (_, e0), (name, e1) in zip(marks[:-1], marks[1:]):
            acc.setdefault(name, []).append(e0.elapsed_time(e1))
    names = ["render", "exchange", "assemble", "end", "host"]
    mine = torch.tensor([sum(acc.get(n, [0.0])) / max(len(acc.get(n, [0.0])), 1) for n in names], dtype=torch.float64, device=(tuning.CONFIG.coll_dev or dev))
    allr = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allr, mine)
    if rank != 0:
        return None
    return {"what": "serial frames, events on the frame's stream: render = the rank's own launch(es), exchange_wait = until its "
                    "send is out / the root's receives have landed (the root posts them before its render, so this is what the "
                    "render did not hide), assemble = the root's scatter kernel",
            "per_rank": [{"rank": i, "render_ms": round(float(v[0]), 4), "exchange_wait_ms": round(float(v[1]), 4),
                          "assemble_ms": round(float(v[2] + v[3]), 4), "host_submit_ms": round(float(v[4]), 4)} for i, v in enumerate(allr)],
            "host_submit_ms_is": "wall time of the host thread inside one frame's calls (launches, the grouped send / receive, waits are "
                                 "stream-level): if it approaches ms_per_step the pipeline is bound by the host, not by the GPUs"}
