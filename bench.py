#!/usr/bin/env python3
"""bench.py — headline benchmark: Mpixels/s of the mainImage() hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--app clouds] [--width 3840] [--height 2160]

Workload (BASELINE.json `metric`): APP_CLOUDS, 3840x2160, canonical frame u_time = 0.37, u_mouse = 0,
default aux uniforms.  One "step" = one whole frame rendered into an RGBA32F framebuffer resident in
HBM (nothing crosses PCIe inside the timed region).

Frames are independent, so consecutive frames are pipelined over two HIP streams (double-buffered
framebuffers, --streams): the drain of one frame's kernel (its last, longest waves) overlaps the start of
the next frame.  The timed region still runs from the first launch to the completion of all K frames.
One-time initialisation (code-object load, APP_CLOUDS' y table, first submission on each stream, first touch of the
framebuffers, RCCL peer set-up) happens once before the W warm-up steps and is not a step.

N = 1 : the frame is one kernel launch.
N > 1 : one process per GPU (torch.distributed / RCCL).  The SAME frame is sharded as cyclic 8-row
        blocks (shaderbox_amd/shard.py), every rank renders its blocks, ONE gather over xGMI brings the
        slabs to rank 0, and one small kernel scatters them to their rows.  Total work is fixed ->
        "scaling": "strong".  Time = barrier + synchronize bracket, max over ranks.

Extra objects on the JSON line:
                 kernel_ms is the duration of an un-overlapped launch (measured after the timed region, one launch at a
                 time); the committed rocprofv3 summary (profiles/r01_clouds_final_rocprof_summary.txt) is of this
                 command with --streams 1 — with two frames in flight the per-kernel durations rocprof reports are
                 stretched by the overlap.
  roofline_hbm : the same kernel against HBM (16 B/pixel written once): far from the bound by design.
  roofline     : dominant kernel (the app's render kernel).  The path is VALU-bound (no MFMA, 16 B/pixel
                 of HBM traffic), so bound = "valu": achieved = algorithmic scalar fp ops per launch
                 (SURVEY.md §8d per-pixel count x pixels) / mean launch duration measured with HIP events on
                 the launch stream; peak = 157.3 TFLOP/s fp32 vector (MI355X_MICROARCH.md); the same figure against the
                 39.3 T lane-ops/s scalar-issue ceiling of SURVEY.md §8d is `frac_of_scalar_issue_ceiling` (above 1: the
                 count is of the REFERENCE algorithm's operations, most of which the kernel no longer executes).
                 `roofline_hbm` gives the framebuffer store rate for completeness.
  cpu_baseline : the CPU oracle (kind "port") timed on this host's cores on a bounded sample of the same
                 frame (every 8th row), rank 0, N = 1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic scalar fp ops per pixel at the canonical frame (every
# transcendental counted as ONE op), measured at the listed resolution
OPS_PER_PIXEL = {"clouds": 60248.0, "egg": 15276.0, "raytracer": 564.0, "atmosphere": 2493.0,
                 "planet": 21253.0, "sdf_ao": 7255.0}       # (no survey count for vinyl: roofline is omitted there)
# HBM bytes per launch from the rocprofv3 PMC passes committed under profiles/ (WRITE_SIZE + 2 x FETCH_SIZE,
# KB -> bytes, per MI355X_MICROARCH.md), for the default workload only; see DESIGN.md §6
MEASURED_TRAFFIC_BYTES = {("clouds", 3840, 2160): int(129600 * 1024 + 2 * 187.415 * 1024)}   # profiles/r01_clouds_final_*
PEAK_FP32_VECTOR_TFLOPS = 157.3
SCALAR_ISSUE_TLANEOPS = 39.3        # 256 CU x 64 lanes x 2.4 GHz: one non-packed, non-FMA lane-op per lane per cycle (SURVEY.md §8d ii)
PEAK_HBM_GBPS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--app", default="clouds")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--time", type=float, default=0.37)
    ap.add_argument("--block-rows", type=int, default=8)
    ap.add_argument("--gather-groups", type=int, default=1,
                    help="N>1: issue the gather in this many pipelined pieces (1 = one plain gather)")
    ap.add_argument("--root-rounds", default="auto",
                    help="N>1: root relief 'm0/m' — row-blocks go out in cycles of m rounds and rank 0 (the gather's root, which "
                         "also lands N-1 slabs and assembles the frame) sits out the rounds >= m0; '1/1' = plain cyclic split; "
                         "'auto' (default) measures the root's per-frame landing+assembly cost against its render time on rank 0 "
                         "and picks m0/8 so that all ranks finish together (shaderbox_amd/shard.py relief_rounds)")
    ap.add_argument("--streams", type=int, default=2,
                    help="frames in flight: consecutive frames alternate over this many HIP streams, each with its own "
                         "framebuffers, so the drain of one frame's kernel overlaps the next frame (1 = strictly serial)")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the torch.distributed/RCCL path even with one rank (smoke test of the N>1 code on 1 GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-row-stride", type=int, default=0,
                    help="cpu_baseline renders every k-th row of the frame (0 = pick from the core count: ~10 s of wall time)")
    args = ap.parse_args()

    # HIP maps streams onto a few hardware queues (4 by default); two streams that share a queue do not overlap at all,
    # and this process uses up to four (default, two frame streams, RCCL's): ask for more queues before HIP starts.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    import shaderbox_amd
    from shaderbox_amd import shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                             % (args.gpus, args.gpus))
    dist = None
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    R = shaderbox_amd.Renderer(local_rank)
    R.set_timing(True)
    W, H, app, t = args.width, args.height, args.app, args.time
    br = args.block_rows

    ns = max(1, args.streams)
    streams = [torch.cuda.Stream(device=dev) for _ in range(ns)] if ns > 1 else [torch.cuda.current_stream(dev)]
    if not use_dist:
        frames = [torch.empty((H, W, 4), dtype=torch.float32, device=dev) for _ in range(ns)]
        frame = frames[0]

        def step(i=0):
            with torch.cuda.stream(streams[i % ns]):
                R.render(app, W, H, t, out=frames[i % ns])
    else:
        from shaderbox_amd.distributed import FramePlan
        relief = choose_relief(args.root_rounds, R, dist, torch, dev, app, W, H, t, br, world, rank, streams)
        plans = [FramePlan(R, dist, W, H, br, groups=args.gather_groups, root_rounds=relief[0], rounds=relief[1])
                 for _ in range(ns)]
        slab = plans[0].slab

        def step(i=0):
            with torch.cuda.stream(streams[i % ns]):
                plans[i % ns].render(app, t)      # render_rank + the single RCCL gather + assemble on rank 0

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # One-time initialisation, outside warm-up and timing (context creation is excluded): the first launch
    # of a kernel loads the code object and builds APP_CLOUDS' y table; the first RCCL transfer sets up the peer links.
    for st in streams:                                  # a HIP stream's hardware queue is created on its first submission
        with torch.cuda.stream(st):
            R.render(app, 64, 36, t)
    if not use_dist:
        for f in frames:
            f.zero_()                                   # first touch of the framebuffers (page mapping) is not rendering
    else:
        for p in plans:
            for buf in (p.slab, p.gathered, p.frame):
                if buf is not None:
                    buf.zero_()
    if dist is not None:
        tiny = torch.zeros(4, device=dev)
        dist.gather(tiny, [torch.zeros(4, device=dev) for _ in range(world)] if rank == 0 else None, dst=0)
    sync()
    for i in range(args.warmup):
        step(i)
    sync()
    kernel_ms = []
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    sync()
    elapsed = time.perf_counter() - t0
    # per-launch kernel duration, HIP events on the launch stream (re-run outside the timed region so that
    # the event queries do not perturb it)
    for _ in range(min(args.steps, 5)):
        if not use_dist:
            R.render(app, W, H, t, out=frame)
        else:
            R.render_rank(app, W, H, t, br, rank, world, out=slab, root_rounds=relief[0], rounds=relief[1])
        kernel_ms.append(R.last_kernel_ms())
    sync()
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        km = torch.tensor([sum(kernel_ms) / len(kernel_ms)], dtype=torch.float64, device=dev)
        dist.all_reduce(km, op=dist.ReduceOp.MAX)
        kmean = float(km.item())
    else:
        kmean = sum(kernel_ms) / len(kernel_ms)

    if rank == 0:
        pixels = W * H
        ms_per_step = elapsed * 1e3 / args.steps
        value = pixels / (ms_per_step * 1e-3) / 1e6
        ops = OPS_PER_PIXEL.get(app)
        launch_pixels = pixels if not use_dist else shard.rank_rows(H, br, 0, world, relief[0], relief[1]) * W
        roofline = None
        if ops is not None:
            achieved = ops * launch_pixels / (kmean * 1e-3) / 1e12
            roofline = {"bound": "valu", "kernel": "k_" + app, "achieved": round(achieved, 4),
                        "peak": PEAK_FP32_VECTOR_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(achieved / PEAK_FP32_VECTOR_TFLOPS, 5),
                        "frac_of_scalar_issue_ceiling": round(achieved / SCALAR_ISSUE_TLANEOPS, 4),
                        "ops_per_pixel": ops, "pixels_per_launch": launch_pixels,
                        "kernel_ms": round(kmean, 4),
                        "traffic": MEASURED_TRAFFIC_BYTES.get((app, W, H)) if world == 1 else None}
            hbm = 16.0 * launch_pixels / (kmean * 1e-3) / 1e9
            roofline_hbm = {"bound": "hbm", "kernel": "k_" + app, "achieved": round(hbm, 2), "peak": PEAK_HBM_GBPS,
                            "unit": "GB/s", "frac": round(hbm / PEAK_HBM_GBPS, 5), "bytes_per_pixel": 16,
                            "traffic": MEASURED_TRAFFIC_BYTES.get((app, W, H)) if world == 1 else None}
        out = {"metric": "Mpixels/s, APP_%s %dx%d" % (app.upper(), W, H), "value": round(value, 3),
               "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "APP_%s %dx%d u_time=%g u_mouse=0 default aux, fragCoord=(x+.5,y+.5)"
                                      % (app.upper(), W, H, t),
                          "frames_in_flight": ns,
                          "parallelism": "1 GPU, one launch per frame" if world == 1 else
                                         "cyclic %d-row blocks over %d GPUs (root sits out rounds >= %d of %d) + 1 RCCL gather "
                                         "(in %d pipelined pieces) + assemble" % (br, world, relief[0], relief[1], args.gather_groups)},
               "roofline": roofline, "roofline_hbm": roofline_hbm if ops is not None else None}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(app, W, H, t, args.cpu_row_stride)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def choose_relief(spec, R, dist, torch, dev, app, W, H, t, br, world, rank, streams):
    """(root_rounds, rounds) of the split, identical on every rank.  'auto': rank 0 measures, with two launches in flight
    as in the timed loop, what its plain 1/N strip costs per frame (t_s) and what only the root has to do per frame — landing
    world-1 slabs in its HBM (a device copy stands in for RCCL's receive kernels) and the assembly kernel (e); the ranks
    adopt shard.best_relief(H, br, world, e / (world * t_s)), the split whose modelled slowest rank is fastest."""
    from shaderbox_amd import shard
    if world <= 1:
        return (1, 1)
    if spec != "auto":
        m0, m = (int(v) for v in spec.split("/"))
        return (m0, m)
    pick = torch.zeros(2, dtype=torch.int64, device=dev)
    if rank == 0:
        rmax = shard.rank_rows_max(H, br, world)
        frame = torch.empty((H, W, 4), dtype=torch.float32, device=dev)
        slabs = [torch.empty((rmax, W, 4), dtype=torch.float32, device=dev) for _ in range(2)]
        src = torch.zeros((world - 1, rmax, W, 4), dtype=torch.float32, device=dev)
        gathered = torch.zeros((world, rmax, W, 4), dtype=torch.float32, device=dev)
        st = streams                                    # the loop's own streams (no extra hardware queues)

        def strips(k):
            for i in range(k):
                with torch.cuda.stream(st[i % len(st)]):
                    R.render_rank(app, W, H, t, br, 0, world, out=slabs[i % 2])
            torch.cuda.synchronize(dev)
        strips(4)
        t0 = time.perf_counter()
        strips(16)
        t_s = (time.perf_counter() - t0) * 1e3 / 16
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for i in range(3):
            if i == 1:
                a.record()
            gathered[1:].copy_(src)
            R.assemble(gathered, W, H, br, world, out=frame)
        b.record()
        torch.cuda.synchronize(dev)
        e = a.elapsed_time(b) / 2.0
        m0, m = shard.best_relief(H, br, world, e / (world * t_s))
        pick[0], pick[1] = m0, m
        del frame, slabs, src, gathered
    dist.broadcast(pick, src=0)
    return (int(pick[0].item()), int(pick[1].item()))


def cpu_baseline(app, W, H, t, stride):
    """The CPU oracle ('port' of the reference path, oracle/) on this host's cores, bounded sample."""
    from oracle.oracle import APP_IDS, Oracle
    o = Oracle()
    cores = os.cpu_count() or 1
    if stride <= 0:
        stride = 8 if cores <= 16 else (4 if cores <= 64 else 2)
    rows = list(range(stride // 2, H, stride))
    o.render_rows(APP_IDS[app], W, H, t, rows[:cores], threads=cores)   # warm the threads/caches
    t0 = time.perf_counter()
    o.render_rows(APP_IDS[app], W, H, t, rows, threads=cores)
    dt = time.perf_counter() - t0
    return {"value": round(len(rows) * W / dt / 1e6, 4), "unit": "Mpixels/s", "cores": cores, "kind": "port",
            "sample": "%d of %d rows (every %dth row) of the same %dx%d frame, %.1f s, g++ -O2 -ffp-contract=off"
                      % (len(rows), H, stride, W, H, dt)}


if __name__ == "__main__":
    main()
