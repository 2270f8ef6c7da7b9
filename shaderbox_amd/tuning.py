"""Tuning of the multi-GPU frame on the node at hand: which exchange form, how much relief for the root, how many
pieces — decided by MEASURING on the ranks that will run the frames, because what wins depends on what the links deliver.

    choice = tuning.choose_exchange(renderer, dist, torch, dev, streams, app, W, H, t, world=world, rank=rank)
    plans = choice.plans                     # one FramePlan per stream, ready to render
    choice.exchange, choice.channels, choice.relief, choice.groups, choice.trials

This is what a host needs beside FramePlan (shaderbox_amd/distributed.py) to run the reference's frame loop
(util/hlsltoy/src/hlsltoy.cpp:494-516) over N GPUs; bench.py calls it and adds nothing of its own (until round 6 the
100 lines of trials lived in bench.py: VERDICT r5 #5).

* `choose_relief`   the root's relief (root_rounds / rounds of the cyclic split), measured on rank 0 with every rank's part of a
                    frame run on its device (`emulated_frame_ms`, `Landing`: a model of RCCL's receive kernels), broadcast;
* `choose_exchange` the exchange form: each candidate is set up on all ranks, warmed, SOAKED (several frames at DIFFERENT times,
                    each compared bit for bit with a one-launch render: a stale line of an earlier frame shows), timed with the
                    frames in flight (barrier + synchronize, slowest rank), and the fastest runs.  A candidate that cannot be set
                    up, faults or differs is dropped on every rank alike.  The whole phase is bounded by `budget_s` of wall time.
  Which forms are candidates: the RCCL forms ("spans", "direct") always.  The store forms (the peers' own pixel stores over xGMI:
  "stores", "span_stores", "packed_stores") only when the ranks share one device (the 1-GPU test form) or when
  SBX_ENABLE_PEER_STORES=1 says that stores across distinct devices have been validated on this kind of node (their
  visibility rides on kernel boundaries between devices, which single-GPU runs cannot show), and only after a PRE-FLIGHT opened one
  HIP-IPC mapping between rank 0 and rank 1 — a container without IPC costs one line, not eight time-outs.
"""
import os
import time

from . import shard


class _Config:
    """process-wide settings of the calibration (bench.py sets them from its flags; the defaults are the measured ones)"""

    def __init__(self):
        self.landing = {"wgs_per_peer": 2, "link_gbps": 50.0}     # how the emulated root lands the peers' payloads (None: a device copy)
        self.coll_dev = None            # device of the small bookkeeping collectives (the GPU under RCCL, the CPU under gloo)
        self.side_streams = []          # Landing's streams (created once per process)


CONFIG = _Config()


def auto_groups(spec, payload_bytes_per_peer):
    """pieces the one exchange is issued in: 'auto' = one per ~12 MB of a peer's payload (a 4K CLOUDS slab goes out whole, an 8K
    slab in 3-4 pieces that leave while the rest renders), at most 8"""
    if spec not in ("auto", "0", 0):
        return max(1, int(spec))
    return max(1, min(8, int(-(-payload_bytes_per_peer // 12e6))))


def relief_candidates(max_rounds=8):
    """(root_rounds, rounds) from the plain split down to a root that renders NO block of its own (0/1: with the span exchange
    the root also renders everything outside the peers' spans, which at 7680x4320 is most of a share), coarsest cycle first"""
    seen, out = set(), []
    for m in range(1, max_rounds + 1):
        for m0 in range(m, 0, -1):
            f = m0 / m
            if f >= .5 and f not in seen:
                seen.add(f)
                out.append((m0, m))
    out = sorted(out, key=lambda c: -c[0] / c[1])
    return out + [(1, 3), (1, 4), (1, 6), (0, 1)]


def choose_relief(spec, R, dist, torch, dev, app, W, H, t, br, world, rank, streams, exchange="direct", channels=3):
    """(root_rounds, rounds) of the split, identical on every rank.  'auto': rank 0 MEASURES the candidates — for each split,
    with the launches in flight on the timed loop's own streams, the root's frame (its strip + landing world-1 slabs in its HBM,
    a device copy standing in for RCCL's receive kernels, + the assembly kernel) and a peer's frame (ranks 1 and world-1) — and
    broadcasts the split whose slower side is fastest.  (Round 1 modelled it from two isolated measurements; HBM-bound copies
    that run beside render waves take longer than alone, and a strip's time is not proportional to its rows, so the model
    under-relieved the root.)"""
    if world <= 1:
        return (1, 1)
    if spec != "auto":
        m0, m = (int(v) for v in spec.split("/"))
        return (m0, m)
    if exchange == "stores":
        return (1, 1)                                   # the root does nothing for the others: the plain split, nothing to calibrate
    pick = torch.zeros(2, dtype=torch.int64, device=(CONFIG.coll_dev or dev))
    if rank == 0:
        ch = channels if exchange in ("direct", "span_stores") else (3 if exchange in ("spans", "packed_stores") else 4)
        st = streams                                    # the loop's own streams (no extra hardware queues)
        nb = max(2, len(st))
        frames = [torch.empty((H, W, 4), dtype=getattr(R, "pixel_dtype", torch.float32), device=dev) for _ in range(nb)]

        def per_frame(fn, k=18):
            return timed_loop(torch, dev, fn, k, min_ms=25.0)

        best = None
        for m0, m in relief_candidates():
            cost = max(emulated_frame_ms(R, torch, dev, st, frames, app, W, H, t, br, world, r, m0, m, exchange, ch, per_frame)
                       for r in sorted({0, 1, world - 1}))
            if best is None or cost < best[0] * .995:        # a later (more relieved) split must win by a margin
                best = (cost, (m0, m))
        pick[0], pick[1] = best[1]
        del frames
        torch.cuda.empty_cache()
    dist.broadcast(pick, src=0)
    return (int(pick[0].item()), int(pick[1].item()))


class Landing:
    """The peers' payloads arriving in the emulated root's HBM, BESIDE the root's own render as on a real node (FramePlan posts the
    grouped receive before the root's launch; RCCL runs it on its own stream).  begin(): fork a side stream off the frame's stream
    and start the landing there; end(): the frame's stream waits for it (what work.wait() does) before the scatter.
    With CONFIG.landing set the landing is sbx_model_landing — `wgs_per_peer` workgroups per peer stay resident for as long as ONE peer's
    payload needs on its link (the peers arrive in parallel over their own links) and write all the bytes at that pace: the CUs and
    the HBM writes of RCCL's receive kernels.  Without: a device copy at HBM speed (round 4's stand-in, which holds the whole chip
    for a few microseconds instead of a few CUs for the link time)."""

    def __init__(self, R, torch, dev, nslots):
        self.R, self.t = R, torch
        # the side streams are made ONCE per process: HIP deals streams onto a few hardware queues in creation order, and a fresh set
        # per figure lands on other queues every time — some of them a render stream's, whose launches then wait behind a landing
        # kernel that is resident for the link time (the root's figures of one sweep came out bimodal, 1.45 / 2.4 ms)
        while len(CONFIG.side_streams) < nslots:
            CONFIG.side_streams.append(torch.cuda.Stream(device=dev))
        self.side = CONFIG.side_streams[:nslots]
        self.ev0 = [torch.cuda.Event() for _ in range(nslots)]
        self.ev1 = [torch.cuda.Event() for _ in range(nslots)]

    def begin(self, slot, dst, src, peers):
        t = self.t
        main = t.cuda.current_stream()
        self.ev0[slot].record(main)
        self.side[slot].wait_event(self.ev0[slot])
        with t.cuda.stream(self.side[slot]):
            n = src.numel() * src.element_size()
            if CONFIG.landing and n % 16 == 0 and n > 0 and peers > 0:
                us = n / peers / (CONFIG.landing["link_gbps"] * 1e9) * 1e6
                self.R.model_landing(src, dst, n, CONFIG.landing["wgs_per_peer"] * peers, us)
            else:
                dst.view(-1)[:src.numel()].copy_(src.view(-1))
            self.ev1[slot].record(self.side[slot])

    def end(self, slot):
        self.t.cuda.current_stream().wait_event(self.ev1[slot])


def timed_loop(torch, dev, fn, k=24, min_ms=60.0):
    """ms per call of fn(i) with the calls in flight: a first batch of k sizes a second one that lasts >= min_ms and is timed with ONE
    synchronisation at its end.  (Round 4 timed k = 24 calls whatever they were: 24 eighth-frames are 7 ms, of which the ramp-in and
    the drain of the pipeline — the last launches finish on an emptying chip — are 3-4 %; the same loop over different ranks' eighths
    for 0.6 s gives 0.279 ms per launch where the 24-call window read 0.293-0.302, tools/launch_granularity.py.)"""
    for i in range(6):
        fn(i)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(k):
        fn(i)
    torch.cuda.synchronize(dev)
    est = (time.perf_counter() - t0) * 1e3 / k
    n = max(k, min(4000, int(min_ms / max(est, 1e-3)) + 1))
    t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) * 1e3 / n


def emulated_frame_ms(R, torch, dev, st, frames, app, W, H, t, br, world, r, m0, m, exchange, ch, per_frame):
    """ms per frame of rank `r`'s part of a `world`-rank frame, ALL of it on this one device with the launches in flight on the
    streams `st`: a peer = its launch; the root = its launch BESIDE the landing of the peers' payloads in its HBM (`Landing`: a model
    of RCCL's receive kernels on their own stream) + the assembly kernel behind both; under the store exchange the root is an ordinary '
        'rank (its launch and the two
    flag kernels), and so is a peer (which renders in place into a frame on this device).  Used by the relief calibration on
    rank 0, by --emulate-ranks and by tools/strip_scaling.py; it knows nothing about the links."""
    nb = len(frames)
    pdt = getattr(R, "pixel_dtype", torch.float32)          # uint8 after R.set_output_format("rgba8"): 4 bytes per pixel anywhere
    epp = 4 if pdt == torch.uint8 else 3                    # buffer elements per pixel of a span slab
    if pdt == torch.uint8:
        ch = 4
    if exchange == "span_stores":
        owners = [R.shared_create(H * W * (4 if pdt == torch.uint8 else 16), 1 if r == 0 else 2) for _ in range(nb)]
        peers = [R.shared_open(o.export()) for o in owners] if r > 0 else []
        views = [o.tensor((H, W, 4)) for o in owners]

        def one(i):
            with torch.cuda.stream(st[i % len(st)]):
                o = owners[i % nb]
                o.begin(0)
                if r == 0:
                    R.render_span_root(app, W, H, t, br, world, views[i % nb], root_rounds=m0, rounds=m)
                    o.end(0)
                else:
                    p = peers[i % nb]
                    p.begin(1)
                    R.render_span_peer_in_place(app, W, H, t, br, r, world, p, root_rounds=m0, rounds=m, channels=ch)
                    p.end(1)
        try:
            return per_frame(one)
        finally:
            torch.cuda.synchronize(dev)
            del views
            for p in peers:
                p.close()
            for o in owners:
                o.close()
    if exchange == "stores":
        # one shared frame per stream, as FramePlan keeps them; a peer is driven together with its owner's "go" (one more flag kernel
        # than a real peer launches: on the pessimistic side)
        owners = [R.shared_create(H * W * (4 if pdt == torch.uint8 else 16), 1 if r == 0 else 2) for _ in range(nb)]
        peers = [R.shared_open(o.export()) for o in owners] if r > 0 else []
        views = [o.tensor((H, W, 4)) for o in owners]

        def one(i):
            with torch.cuda.stream(st[i % len(st)]):
                o = owners[i % nb]
                o.begin(0)
                if r == 0:
                    R.render_rank_in_place(app, W, H, t, br, 0, world, views[i % nb], root_rounds=m0, rounds=m, channels=ch)
                    o.end(0)
                else:
                    p = peers[i % nb]
                    p.begin(1)
                    R.render_rank_in_place(app, W, H, t, br, r, world, p, root_rounds=m0, rounds=m, channels=ch)
                    p.end(1)
        try:
            return per_frame(one)
        finally:
            torch.cuda.synchronize(dev)
            del views
            for p in peers:
                p.close()
            for o in owners:
                o.close()
    if exchange == "packed_stores":
        # the span exchange with the peers' stores as its transport: no landing kernels on the root, the scatter stays
        _, pix, _ = R.span_table(app, W, H, t, br, world, m0, m)
        stride = (int(max(pix[1:])) + 63) // 64 * 64
        land_el = max(world - 1, 1) * max(stride, 1) * epp
        owners = [R.shared_create(land_el * (1 if pdt == torch.uint8 else 4), 1 if r == 0 else 2) for _ in range(nb)]
        peers = [R.shared_open(o.export()) for o in owners] if r > 0 else []
        views = [o.tensor((land_el,)) for o in owners]

        def one(i):
            with torch.cuda.stream(st[i % len(st)]):
                o = owners[i % nb]
                o.begin(0)
                if r == 0:
                    R.render_span_root(app, W, H, t, br, world, frames[i % nb], root_rounds=m0, rounds=m)
                    o.end(0)
                    R.assemble_spans(app, W, H, t, br, world, views[i % nb], stride, frames[i % nb], root_rounds=m0, rounds=m)
                else:
                    p = peers[i % nb]
                    p.begin(1)
                    R.render_span_peer(app, W, H, t, br, r, world, 0, 1 << 30, (p, (r - 1) * stride * epp * (1 if pdt == torch.uint8 else 4)),
                                       root_rounds=m0, rounds=m)
                    p.end(1)
        try:
            return per_frame(one)
        finally:
            torch.cuda.synchronize(dev)
            del views
            for p in peers:
                p.close()
            for o in owners:
                o.close()
    if exchange == "spans":
        _, pix, _ = R.span_table(app, W, H, t, br, world, m0, m)
        stride = (int(max(pix[1:])) + 63) // 64 * 64
        if r > 0:
            slabs = [torch.empty((max(int(pix[r]), 1) * epp,), dtype=pdt, device=dev) for _ in range(nb)]

            def peer(i):
                with torch.cuda.stream(st[i % len(st)]):
                    R.render_span_peer(app, W, H, t, br, r, world, 0, 1 << 30, slabs[i % nb], root_rounds=m0, rounds=m)
            return per_frame(peer)
        total = sum(int(p) for p in pix[1:])
        tot_el = (max(total, 1) * epp + 15) // 16 * 16          # (whole 16-byte units for the landing model)
        src = torch.zeros((tot_el,), dtype=pdt, device=dev)
        land_el = max((world - 1) * max(stride, 1) * epp, tot_el)
        lands = [torch.zeros((land_el,), dtype=pdt, device=dev) for _ in range(nb)]

        ld = Landing(R, torch, dev, nb)

        def root(i):
            with torch.cuda.stream(st[i % len(st)]):
                ld.begin(i % nb, lands[i % nb], src, world - 1)
                R.render_span_root(app, W, H, t, br, world, frames[i % nb], root_rounds=m0, rounds=m)
                ld.end(i % nb)
                R.assemble_spans(app, W, H, t, br, world, lands[i % nb], stride, frames[i % nb], root_rounds=m0, rounds=m)
        return per_frame(root)
    rmax = shard.rank_rows_max(H, br, world, m0, m)
    slabs = [torch.empty((rmax, W, ch), dtype=pdt, device=dev) for _ in range(nb)]
    if r > 0:
        def peer(i):
            with torch.cuda.stream(st[i % len(st)]):
                R.render_rank_rows(app, W, H, t, br, r, world, 0, rmax, slabs[i % nb], root_rounds=m0, rounds=m)
        return per_frame(peer)
    src = torch.zeros((world - 1, rmax, W, ch), dtype=pdt, device=dev)
    lands = [torch.zeros((world, rmax, W, ch), dtype=pdt, device=dev) for _ in range(nb)]

    ld = Landing(R, torch, dev, nb)

    def root(i):
        with torch.cuda.stream(st[i % len(st)]):
            g, f = lands[i % nb], frames[i % nb]
            ld.begin(i % nb, g[1:], src, world - 1)
            if exchange == "direct":
                R.render_rank_in_place(app, W, H, t, br, 0, world, f, root_rounds=m0, rounds=m)
                ld.end(i % nb)
                R.assemble_peers(g[1:], W, H, br, world, f, root_rounds=m0, rounds=m)
            else:
                R.render_rank_rows(app, W, H, t, br, 0, world, 0, rmax, slabs[i % nb], root_rounds=m0, rounds=m)
                g[0].copy_(slabs[i % nb])
                ld.end(i % nb)
                R.assemble(g, W, H, br, world, out=f, root_rounds=m0, rounds=m)
    return per_frame(root)


# ---------------------------------------------------------------------------------------------------------
# the exchange form
# ---------------------------------------------------------------------------------------------------------
RCCL_FORMS = (("spans", None), ("direct", None))
STORE_FORMS = (("stores", 3), ("stores", 4), ("span_stores", 3), ("span_stores", 4), ("packed_stores", None))
SPAN_FORMS = ("spans", "span_stores", "packed_stores")


class ExchangeChoice:
    """what choose_exchange decided, and the plans (one per stream) that run it"""

    def __init__(self, exchange, channels, relief, groups, payload, plans, trials, notes):
        self.exchange, self.channels, self.relief, self.groups = exchange, channels, relief, groups
        self.payload_bytes_per_peer, self.plans, self.trials, self.notes = payload, plans, trials, notes


def form_name(exchange, channels):
    return exchange if channels in (None, 3) else exchange + "_16B"


def payload_bytes(R, app, W, H, t, br, world, relief, exchange, channels):
    """bytes one peer puts on its link per frame"""
    if world <= 1:
        return 0
    if exchange in SPAN_FORMS:
        per_pixel = 16 if (exchange == "span_stores" and channels == 4) else 12
        return per_pixel * int(max(R.span_table(app, W, H, t, br, world, relief[0], relief[1])[1][1:]))
    per_pixel = 12 if (exchange in ("direct", "stores") and channels == 3) else 16
    return per_pixel * W * shard.rank_rows_max(H, br, world, *relief)


def make_plans(R, dist, torch, dev, streams, app, W, H, t, br, world, rank, exchange, channels, root_rounds="auto", groups="auto",
               fdist=None):
    """(relief, payload, groups, plans) of one exchange form: the relief calibrated on rank 0 for THIS form, one FramePlan per stream"""
    from .distributed import FramePlan
    relief = choose_relief(root_rounds, R, dist, torch, dev, app, W, H, t, br, world, rank, streams, exchange, channels)
    payload = payload_bytes(R, app, W, H, t, br, world, relief, exchange, channels)
    g = auto_groups(groups, payload)
    plans = [FramePlan(R, fdist if fdist is not None else dist, W, H, br, groups=g, root_rounds=relief[0], rounds=relief[1],
                       exchange=exchange, channels=channels) for _ in streams]
    return relief, payload, g, plans


def ranks_share_a_device(dist, torch, dev, world):
    """do all ranks of the group drive ONE physical device (the 1-GPU test form)?  By host name and PCI bus id."""
    import socket
    props = torch.cuda.get_device_properties(dev)
    me = (socket.gethostname(), getattr(props, "pci_bus_id", None), getattr(props, "pci_device_id", None), str(getattr(props, "uuid", "")))
    if me[1] is None and not me[3]:
        me = me + (int(dev.index or 0),)
    allv = [None] * world
    dist.all_gather_object(allv, me)
    return all(v == allv[0] for v in allv)


def ipc_preflight(R, dist, torch, dev, world, rank):
    """ONE HIP-IPC mapping between rank 0 and rank 1, before any store form is tried: (ok on every rank, why not)."""
    ok, why, owner, peer = True, None, None, None
    try:
        box = [None]
        if rank == 0:
            owner = R.shared_create(4096, 2)
            box[0] = owner.export()
        dist.broadcast_object_list(box, src=0)
        if rank == 1:
            peer = R.shared_open(box[0])
    except Exception as e:                               # noqa: BLE001
        ok, why = False, "rank %d: %s: %s" % (rank, type(e).__name__, str(e)[:160])
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=CONFIG.coll_dev or dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    for o in (peer, owner):
        try:
            if o is not None:
                o.close()
        except Exception:                                # noqa: BLE001
            pass
    return bool(flag.item()), why


def _dsync(torch, dev):
    if getattr(dev, "type", "cuda") == "cuda":
        torch.cuda.synchronize(dev)


def _on(torch, stream):
    """the stream context of a frame (None: the CPU tests' stand-in renderer has no streams)"""
    import contextlib
    return contextlib.nullcontext() if stream is None else torch.cuda.stream(stream)


def _empty_cache(torch, dev):
    if getattr(dev, "type", "cuda") == "cuda":
        torch.cuda.empty_cache()


def choose_exchange(R, dist, torch, dev, streams, app, W, H, t, *, world, rank, block_rows=shard.DEFAULT_BLOCK_ROWS, exchange="auto",
                    channels=3, root_rounds="auto", groups="auto", fdist=None, budget_s=20.0, preroll_ms=40.0, soak_frames=None,
                    allow_stores=None, trial_frames=12):
    """The exchange form for frames of (app, W x H) on the ranks of `dist`, and its plans.  Collective: every rank calls it with the
    same arguments.  `exchange` other than "auto": that form, no trial.  `budget_s`: wall time (rank 0's clock) after which no further
    candidate is started — the first one always runs; what was cut is in `.notes["cut"]`.  `allow_stores`: None = the rule of the
    module docstring (one shared device, or SBX_ENABLE_PEER_STORES=1), True / False = say so."""
    ns, br = len(streams), block_rows
    notes = {"candidates": [], "cut": [], "stores": None, "budget_s": budget_s}

    def sync():
        dist.barrier()
        _dsync(torch, dev)

    def agreed(ok):                                      # every rank's verdict on a step of a trial: all of them, or none
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=CONFIG.coll_dev or dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(flag.item())

    def prepare(ex, ch):
        return make_plans(R, dist, torch, dev, streams, app, W, H, t, br, world, rank, ex, channels if ch is None else ch, root_rounds,
                          groups, fdist)

    if exchange != "auto" or world == 1:
        ex = exchange if exchange != "auto" else "spans"  # (one rank: nothing to exchange)
        relief, payload, g, plans = prepare(ex, None)
        return ExchangeChoice(ex, plans[0].channels, relief, g, payload, plans, None, notes)

    t_start = time.perf_counter()
    # the clocks first (~25 ms of launches until DVFS holds its clock) — or the form tried first pays for the ramp
    scratch = R.empty((H, W, 4))
    while preroll_ms > 0 and (time.perf_counter() - t_start) * 1e3 < max(preroll_ms, 40.0):
        for _ in range(4):
            R.render(app, W, H, t, out=scratch)
        _dsync(torch, dev)
    del scratch

    cands = list(RCCL_FORMS)
    if allow_stores is None:
        shared = ranks_share_a_device(dist, torch, dev, world)
        allow_stores = shared or os.environ.get("SBX_ENABLE_PEER_STORES", "0") == "1"
        notes["stores"] = ("ranks share one device (test form)" if shared else "SBX_ENABLE_PEER_STORES=1" if allow_stores else
                           "not tried: the ranks drive distinct devices and SBX_ENABLE_PEER_STORES is not 1 (store visibility between "
                           "devices has not been validated on this kind of node)")
    if allow_stores:
        ok, why = ipc_preflight(R, dist, torch, dev, world, rank)
        if ok:
            cands += list(STORE_FORMS)
        else:
            notes["stores"] = "not tried: the HIP-IPC pre-flight failed (%s)" % (why or "on another rank")
    notes["candidates"] = [form_name(ex, ch) for ex, ch in cands]
    soak = 2 * ns if soak_frames is None else int(soak_frames)

    trials, best = {}, None
    # (The form tried FIRST reads slow whatever it is — first use of the mappings and of the processes' queues — so the first
    # candidate is tried twice and its first reading is thrown away.)
    for k_trial, (ex, ch) in enumerate([cands[0]] + cands):
        name = form_name(ex, ch)
        if k_trial == 0:
            name = "(first trial, discarded) " + name
        over = torch.tensor([1 if (k_trial > 1 and time.perf_counter() - t_start > budget_s) else 0], dtype=torch.int32,
                            device=CONFIG.coll_dev or dev)
        dist.broadcast(over, src=0)                      # rank 0's clock decides for all
        if int(over.item()):
            notes["cut"].append(name)
            continue
        # A form that cannot be set up on these devices, that faults, or whose frames differ from one launch is DROPPED, on every
        # rank alike, and the record says so: a trial must never take the run down with it.
        cand, why = None, None
        try:
            cand = prepare(ex, ch)
        except Exception as e:                           # noqa: BLE001
            why = "set-up failed on rank %d: %s: %s" % (rank, type(e).__name__, str(e)[:200])
        if not agreed(cand is not None):
            trials[name] = "unavailable (%s)" % (why or "set-up failed on another rank")
            cand = None
            _empty_cache(torch, dev)
            continue
        cplans = cand[3]
        ms, why = None, None
        try:
            for i in range(3 * ns):                      # first use of a form: mappings, code objects, first touch of a mapped frame
                with _on(torch, streams[i % ns]):
                    cplans[i % ns].render(app, t)
            sync()
            # SOAK: frames at DIFFERENT times, ns in flight, each against a one-launch render of its own time — a stale line of an
            # earlier frame (a store that had not reached the owner when its flag did) is a differing pixel here, where a
            # repeat of one frame would hide it
            for base in range(0, soak, ns):
                times = [t + 1e-3 * (base + j + 1) for j in range(min(ns, soak - base))]
                for j, tk in enumerate(times):
                    with _on(torch, streams[j % ns]):
                        cplans[j % ns].render(app, tk)
                sync()
                if rank == 0 and why is None:
                    for j, tk in enumerate(times):
                        whole = R.render(app, W, H, tk)
                        _dsync(torch, dev)
                        if bool((cplans[j % ns].frame.view(torch.int32) != whole.view(torch.int32)).any().item()):
                            why = "frame %d of its soak differs from a one-launch render" % (base + j)
                        del whole
            if R.fault_status() != 0:
                why = "a wait of the exchange timed out (fault word)"
            if why is None:
                t0 = time.perf_counter()
                for i in range(trial_frames):
                    with _on(torch, streams[i % ns]):
                        cplans[i % ns].render(app, t)
                sync()
                ms = (time.perf_counter() - t0) * 1e3 / trial_frames
        except Exception as e:                           # noqa: BLE001
            why = "%s: %s" % (type(e).__name__, str(e)[:200])
        if not agreed(why is None):
            trials[name] = "dropped (%s)" % (why or "failed on another rank")
            try:
                _dsync(torch, dev)
                if R.fault_status() != 0:
                    R.clear_fault()
            except Exception:                            # noqa: BLE001
                pass
            del cand, cplans
            _empty_cache(torch, dev)
            continue
        dt = torch.tensor([ms], dtype=torch.float64, device=CONFIG.coll_dev or dev)
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        trials[name] = round(float(dt.item()), 4)
        if k_trial > 0 and (best is None or trials[name] < best[0]):
            best = (trials[name], ex, cand, ch)
        del cand, cplans
        _empty_cache(torch, dev)
    if best is None:
        raise RuntimeError("no exchange form could be set up on these ranks: %s" % trials)
    relief, payload, g, plans = best[2]
    notes["trial_seconds"] = round(time.perf_counter() - t_start, 2)
    return ExchangeChoice(best[1], plans[0].channels, relief, g, payload, plans, trials, notes)
