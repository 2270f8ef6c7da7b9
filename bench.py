#!/usr/bin/env python3
"""bench.py — headline benchmark: Mpixels/s of the mainImage() hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--app clouds] [--width 3840] [--height 2160]

Workload (BASELINE.json `metric`): APP_CLOUDS, 3840x2160, canonical frame u_time = 0.37, u_mouse = 0,
default aux uniforms.  One "step" = one whole frame rendered into an RGBA32F framebuffer resident in
HBM (nothing crosses PCIe inside the timed region).

Frames are independent, so consecutive frames are pipelined over three HIP streams (one framebuffer per
stream, --streams; 1 / 2 / 3 / 4 in flight: 2 849 / 3 088 / 3 145 / 3 122 Mpixels/s): the drain of one frame's kernel (its last, longest waves) overlaps the start of
the next frame.  The timed region still runs from the first launch to the completion of all K frames.
One-time initialisation (code-object load, APP_CLOUDS' y table, first submission on each stream, first touch of the
framebuffers, RCCL peer set-up) happens once before the W warm-up steps and is not a step.

N = 1 : the frame is one kernel launch.
N > 1 : one process per GPU (torch.distributed / RCCL; `--exchange auto` tries the span exchange and the whole-slab exchange on the
        ranks at hand and runs the faster, `exchange.chosen` in the line).  `python bench.py --gpus N` launched as a plain command starts
        its own N ranks (re-executes itself under torch.distributed.run on 127.0.0.1); launched by torch.distributed.run
        it uses the ranks it is given.  The SAME frame is sharded as cyclic 8-row blocks (shaderbox_amd/shard.py), every
        rank renders its blocks, ONE exchange over xGMI brings the slabs to rank 0, and one small kernel scatters them to
        their rows.  Total work is fixed -> "scaling": "strong".  Time = barrier + synchronize bracket, max over ranks.
        The line also carries `phases` (per rank: render_ms / exchange_wait_ms / assemble_ms of serial frames timed with
        events after the timed region) and `steady_state` (the frames completed between the end of the first burst of
        `frames_in_flight` frames and the start of the last one, rank 0: the pipeline's rate without ramp-in and drain),
        `exchange` (kind, how it was chosen, bytes per peer, pieces, link time at the xGMI peak), `value_serial`, and
        `other_configs` = BASELINE config 5 (APP_ATMOSPHERE and APP_PLANET 7680x4320) through the same schedule.
One GPU only: `--emulate-ranks N` runs every rank's schedule through a loopback world on this device, checks the frames against
        one launch and prints MODELLED N-GPU figures with the exchange budget; `--gpus N --backend gloo` runs the whole N-process
        program with the ranks sharing the GPU and the transfers staged through the host (a test form: its rates mean nothing).

Extra objects on the JSON line (N = 1 unless noted):
  roofline     : dominant kernel (the app's render kernel).  The path is VALU-bound (no MFMA, 16 B/pixel of HBM traffic),
                 so bound = "valu", and the object reports EXECUTED work, which cannot exceed the peak:
                   achieved            VALU lane-operations the kernel actually issued per second: SQ_INSTS_VALU x 64 lanes /
                                       the launch's duration in the same rocprofv3 pass
                   peak                FIXED: 1024 SIMD-32 x 32 lanes x 2.4 GHz = 78.64 T lane-ops/s (MI355X_MICROARCH.md: 4 SIMD-32
                                       per CU, 2-cycle wave64 issue), whatever clock DVFS held
                   frac                achieved / peak (the primary figure).  frac_at_measured_clock = SQ_INSTS_VALU x 2 issue
                                       cycles / (1024 SIMDs x active cycles): the share of the VALU issue slots of the cycles that
                                       actually happened (shader clock = GRBM_GUI_ACTIVE / 8 XCDs / duration); frac_unprofiled_duration
                                       = the profiled instruction count over the UN-profiled launch duration (HIP events), fixed peak
                   useful_work_ratio   NOT a utilisation figure: the REFERENCE algorithm's scalar fp ops per launch (SURVEY.md
                                       §8d per-pixel count x pixels) / un-overlapped launch duration / 157.3 TFLOP/s.  It can
                                       exceed 1 because the kernel executes far fewer operations than the reference algorithm
                                       (hashes shared per wave, clear samples proved away) for the same bits.
                   valu_busy_pct       rocprofv3's VALUBusy halved (its gfx94x formula assumes 4-cycle issue).
                   traffic             HBM bytes per launch: WRITE_SIZE + 2 x FETCH_SIZE (gfx950 correction of
                                       MI355X_MICROARCH.md), separate PMC passes.
                 The PMC-derived fields are measured in this run (`--pmc auto`: rocprofv3 is started on a short serial run
                 of this script, one counter group per pass; `pmc_source` says so) or, when rocprofv3 is not usable or N > 1,
                 derived from the committed per-launch instruction count under profiles/ (named in `pmc_source`) at the
                 nominal clock.
  roofline_hbm : the same kernel against HBM (16 B/pixel written once): far from the bound by design.
  serial       : Mpixels/s of one un-overlapped launch, median of 10 after 2 warm-ups (SURVEY.md §8d defines the metric per launch;
                 `value` has `frames_in_flight` launches overlapping); also as the top-level key `value_serial`.
  parity       : rows of the timed GPU frame against the CPU oracle's rows of the same frame (the ones cpu_baseline
                 renders): max |diff| and pixels with any differing bit.  > 1e-4 -> non-zero exit status.
                 N > 1: the assembled frame against a one-launch render of the same frame on rank 0 (bit-identical).
  cpu_baseline : the CPU oracle (kind "port") timed on this host's cores on a bounded sample of the same frame (every
                 k-th row, dealt to the threads in 64-pixel tiles), strict build (g++ -O2 -ffp-contract=off).  `cores` = the
                 threads used; `affinity`, `cgroup_cpu_max` and the one-thread rate say what those threads could get.  Printed
                 for every N (rank 0, after the timed region).
  cpu_baseline_speed : the same sample with the optimisation level of the reference's own C++ build
                 (-O3 -march=native -funroll-loops, /root/reference/src/Makefile:12-13), compiled on this host at run time.
  other_configs: the other BASELINE.json GPU configs, timed the same way (pipelined frames + un-overlapped kernel time), each
                 with its own `parity` object: 16 evenly spread full rows of the rendered frame against the CPU oracle.
"""
import argparse
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic scalar fp ops per pixel at the canonical frame (every
# transcendental counted as ONE op), measured at the listed resolution
OPS_PER_PIXEL = {"clouds": 60248.0, "egg": 15276.0, "raytracer": 564.0, "atmosphere": 2493.0,
                 "planet": 21253.0, "sdf_ao": 7255.0}       # (no survey count for vinyl / clouds_best / clouds_tex)
PEAK_FP32_VECTOR_TFLOPS = 157.3
PEAK_HBM_GBPS = 8000.0
N_SIMD = 1024                       # 256 CU x 4
VALU_ISSUE_CYCLES = 2.0             # wave64 VALU instruction on a SIMD-32 (MI355X_MICROARCH.md)
NOMINAL_CLOCK_HZ = 2.4e9
LANES_PER_SIMD_CYCLE = 32           # a SIMD-32 retires half a wave64 instruction per cycle
PEAK_LANEOPS_NOMINAL_T = N_SIMD * LANES_PER_SIMD_CYCLE * NOMINAL_CLOCK_HZ / 1e12     # 78.6 T lane-ops/s
PMC_ROUND = "r05"
SIDE_STREAMS = []                   # Landing's streams (created once, right after the render streams)
LANDING = {"wgs_per_peer": 2, "link_gbps": 50.0}     # how the emulated root lands the peers' payloads (main() sets it from the flags)
COLL_DEV = None                     # device of the small bookkeeping collectives (set in main: the GPU under RCCL, the CPU under gloo)                   # committed per-launch counters: profiles/<PMC_ROUND>_pmc_<app>_<W>x<H>.json
# the other BASELINE.json configs that fit one GPU: (app, W, H) — C2, C3, C5 (both apps)
OTHER_CONFIGS = [("egg", 1920, 1080), ("raytracer", 3840, 2160), ("atmosphere", 7680, 4320), ("planet", 7680, 4320)]
KERNEL_OF = {"clouds": "k_clouds", "egg": "k_egg", "raytracer": "k_raytracer", "atmosphere": "k_atmosphere",
             "planet": "k_planet", "sdf_ao": "k_sdf_ao", "vinyl": "k_vinyl", "clouds_best": "k_clouds_best",
             "clouds_tex": "k_clouds_tex"}


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args, argv):
    """`python bench.py --gpus N` as a plain command: start N ranks of this script under torch.distributed.run."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("GPU_MAX_HW_QUEUES", "8")
    env["SBX_BENCH_SELF_LAUNCHED"] = "1"
    return subprocess.call(cmd, env=env)


def launch_check(args):
    """--launch-check: ranks only rendezvous (gloo, no GPU) and rank 0 prints one JSON line; tests the launcher."""
    import torch
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([float(rank)])
    dist.all_reduce(t)
    dist.barrier()
    if rank == 0:
        claim_stdout()(json.dumps({"launch_check": True, "n_gpus": world, "rank_sum": float(t.item()),
                                   "self_launched": os.environ.get("SBX_BENCH_SELF_LAUNCHED") == "1"}))
    dist.destroy_process_group()
    return 0


_emit = None


def claim_stdout():
    """The contract: stdout carries ONE JSON line.  Libraries inside this process must not add to it — RCCL prints a version
    banner to the C stdout, flushed at exit, i.e. AFTER the line — so descriptor 1 is pointed at stderr for the life of the
    process and the line goes to the saved descriptor."""
    global _emit
    if _emit is None:
        sys.stdout.flush()
        real = os.dup(1)
        os.dup2(2, 1)

        def _emit(line):
            os.write(real, (line + "\n").encode())
    return _emit


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
    ap.add_argument("--gather-groups", default="auto",
                    help="N>1: issue the one exchange in this many pipelined pieces ('auto': one per ~12 MB of a peer's payload, "
                         "so a 4K slab goes out whole and an 8K one in 3 pieces; 1 = one plain exchange)")
    ap.add_argument("--exchange", choices=["auto", "spans", "direct", "gather", "stores", "span_stores", "packed_stores"], default="auto",
                    help="N>1 (engine dist): 'auto' (default) = try 'stores' and 'span_stores' (12- and 16-byte pixels), 'packed_stores', 'spans' and 'direct' on the ranks at hand (a few "
                         "pipelined frames each) and run the fastest; 'span_stores' = 'stores' with only the SPANS of the peers' row-blocks "
                         "stored in place, the root renders the rest (fewer bytes on the links: 30 instead of 50 MB per peer at 7680x4320); 'stores' = the peers map the root's frame (HIP IPC) and render their row-blocks IN PLACE "
                         "into it: the exchange is their own pixel stores over xGMI (12 bytes per pixel with --channels 3), the root lands, "
                         "receives and scatters nothing (distributed.py, include/sbx.h sbx_shared_*); 'packed_stores' = the span exchange with the peers' own stores as transport: packed spans (12 contiguous bytes per pixel) straight into the root's mapped landing area, the root scatters; 'spans' = only the expensive interval of every row-block is dealt to the peers and "
                         "sent, the root renders the rest in place (distributed.py; config 5's 49.8 MB per peer become 29.7 MB); "
                         "'direct' = the root renders its blocks in place and receives the peers' whole slabs by ONE grouped "
                         "send/recv; 'gather' = dist.gather of equal RGBA slabs + assembly of all of them (round 1)")
    ap.add_argument("--channels", type=int, choices=[3, 4], default=3,
                    help="N>1, exchange direct: floats per pixel that cross xGMI (3: alpha, the constant 1 of main.h:52, is written "
                         "by the root's assembly)")
    ap.add_argument("--root-rounds", default="auto",
                    help="N>1: root relief 'm0/m' — row-blocks go out in cycles of m rounds and rank 0 (the gather's root, which "
                         "also lands N-1 slabs and assembles the frame) sits out the rounds >= m0; '1/1' = plain cyclic split; "
                         "'auto' (default) measures the root's per-frame landing+assembly cost against its render time on rank 0 "
                         "and picks the split whose modelled slowest rank is fastest (shaderbox_amd/shard.py best_relief)")
    ap.add_argument("--streams", type=int, default=3,
                    help="frames in flight: consecutive frames alternate over this many HIP streams, each with its own "
                         "framebuffers, so the drain of one frame's kernel overlaps the next frame (1 = strictly serial)")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the torch.distributed/RCCL path even with one rank (smoke test of the N>1 code on 1 GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-row-stride", type=int, default=0,
                    help="cpu_baseline renders every k-th row of the frame (0 = pick from the core count: ~10 s of wall time)")
    ap.add_argument("--pmc", default="auto", choices=["auto", "live", "profiles", "off"],
                    help="where roofline.traffic / valu_issue_frac / valu_busy_pct come from: 'live' = rocprofv3 passes of a short "
                         "serial run of this script, 'profiles' = the committed summary, 'auto' = live, else profiles, else null")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--engine", default="dist", choices=["dist", "lib"],
                    help="N>1: 'dist' = one process per GPU (torch.distributed / RCCL gather, the contract's launch shape); 'lib' = ONE "
                         "process drives the N GPUs through the library's own multi-GPU path (sbx_multi_*: RCCL send/recv per "
                         "row-block straight into the final rows, no assembly pass); with fewer GPUs than N the ranks share devices")
    ap.add_argument("--lib-exchange", choices=["slabs", "blocks", "spans", "peer_stores"], default="spans",
                    help="--engine lib: 'spans' (default) = the span exchange inside the library; 'slabs' = one send/receive per peer of its "
                         "whole 3-channel slab + one scatter kernel on the root; 'blocks' = one send/receive pair per row-block straight "
                         "into the final rows (round 2); 'peer_stores' = every rank stores into rank 0's frame through peer access")
    ap.add_argument("--emulate-ranks", type=int, default=0,
                    help="ONE GPU, no process group: run the N-rank schedule of the headline and of config 5 through a loopback world "
                         "(every rank's real FramePlan, kernels, span tables, assembly on this device), check the frames against one "
                         "launch, time every rank's part with frames in flight and print the MODELLED N-GPU figures with the exchange "
                         "budget (n_gpus stays 1, 'emulated_ranks' says so; link rates are assumptions: --link-gbps)")
    ap.add_argument("--link-gbps", type=float, default=50.0, help="--emulate-ranks: the per-direction xGMI rate of the budget")
    ap.add_argument("--rccl-wgs-per-peer", type=int, default=2,
                    help="the emulated root of the send/recv exchanges (relief calibration, --emulate-ranks, tools/strip_scaling.py): "
                         "RCCL's grouped receive is modelled as this many 256-thread workgroups PER PEER that stay resident for the time "
                         "the peer's payload needs on its link (--link-gbps) and write it into the landing area at that pace "
                         "(include/sbx_test.h sbx_model_landing); 0 = round 4's stand-in, a plain device copy at HBM speed")
    ap.add_argument("--format", choices=["rgba32f", "rgba8"], default="rgba32f",
                    help="--emulate-ranks only: the pixels the kernels write and the exchange carries — float (the metric's frame), or "
                         "SBX_FORMAT_RGBA8, the 4-byte display format of the reference's hosts (include/sbx.h); the N = 1 figure the "
                         "speed-up refers to is measured in the same format")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="N>1 process group: 'nccl' = RCCL (the product path); 'gloo' = TEST ONLY: the ranks may share a GPU (rank r on "
                         "device r mod device count), point-to-point transfers are staged through host memory "
                         "(distributed.HostStagedDist) — runs the whole N > 1 program on a 1-GPU box, measures nothing about xGMI")
    ap.add_argument("--preroll-ms", type=float, default=40.0,
                    help="N = 1: back-to-back frames for at least this long BEFORE the warm-up steps (not steps, not timed): the first ~25 ms "
                         "of launches after host work run at ramping clocks (profiles/r04_streams3_trace.txt), and 5 warm-up frames are 11 ms")
    ap.add_argument("--sustained-seconds", type=float, default=2.5,
                    help="N = 1: after the timed region, frames back to back for this long with the shader clock and the board power "
                         "sampled beside them -> the `sustained` object (0 = skip)")
    ap.add_argument("--launch-check", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    global LANDING
    LANDING = {"wgs_per_peer": args.rccl_wgs_per_peer, "link_gbps": args.link_gbps} if args.rccl_wgs_per_peer > 0 else None
    if args.backend == "gloo" and args.exchange == "gather":
        raise SystemExit("--backend gloo stages point-to-point transfers only: use --exchange auto, spans or direct")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.engine == "lib" and args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        claim_stdout()
        sys.exit(bench_lib(args))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: become the launcher of N ranks (the driver's command shape at N = 1, 2, 4, 8)
        sys.exit(self_launch(args, sys.argv[1:]))
    if args.launch_check:
        claim_stdout()
        sys.exit(launch_check(args))
    if world != args.gpus and not (world == 1 and args.gpus == 1):
        raise SystemExit("bench.py --gpus %d runs under WORLD_SIZE=%d: launch with --nproc-per-node %d" % (args.gpus, world, args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    claim_stdout()

    # HIP maps streams onto a few hardware queues (4 by default); two streams that share a queue do not overlap at all,
    # and this process uses up to four (default, two frame streams, RCCL's): ask for more queues before HIP starts.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    import shaderbox_amd
    from shaderbox_amd import shard

    dist = None
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "gloo":
            local_rank = local_rank % max(torch.cuda.device_count(), 1)      # test form: ranks may share a device
            torch.cuda.set_device(local_rank)
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    global COLL_DEV
    COLL_DEV = dev if args.backend == "nccl" else torch.device("cpu")      # where the small bookkeeping collectives live
    torch.cuda.set_device(dev)
    R = shaderbox_amd.Renderer(local_rank)
    R.set_timing(True)
    W, H, app, t = args.width, args.height, args.app, args.time
    br = args.block_rows

    ns = max(1, args.streams)
    streams = [torch.cuda.Stream(device=dev) for _ in range(ns)] if ns > 1 else [torch.cuda.current_stream(dev)]
    for st in streams:                                  # a HIP stream's hardware queue is created on its first submission
        with torch.cuda.stream(st):
            R.render(app, 64, 36, t)
    if args.emulate_ranks > 1 or use_dist:              # the emulated root's landing streams, on the queues after the render streams'
        for _ in range(max(2, ns)):
            SIDE_STREAMS.append(torch.cuda.Stream(device=dev))
            with torch.cuda.stream(SIDE_STREAMS[-1]):
                R.render(app, 64, 36, t)
    if dist is not None:                                # the first RCCL transfer sets up the peer links
        tiny = torch.zeros(4, device=COLL_DEV)
        dist.gather(tiny, [torch.zeros(4, device=COLL_DEV) for _ in range(world)] if rank == 0 else None, dst=0)
        dist.barrier()
    torch.cuda.synchronize(dev)

    status = 0
    if args.format != "rgba32f" and not (args.emulate_ranks > 1 and not use_dist):
        sys.exit("--format rgba8 is an option of --emulate-ranks (the measured metric is the float frame)")
    if args.emulate_ranks > 1 and not use_dist:
        R.set_output_format(args.format)
        sys.exit(bench_emulated(args, R, torch, dev, streams, app, W, H, t))
    if use_dist:
        res = dist_frame_bench(R, dist, torch, dev, streams, args, app, W, H, t, world, rank, args.steps, args.warmup)
        out = None
        if rank == 0:
            out = dist_line(res, args, app, W, H, t, world)
            if res["mismatching_pixels"]:
                status = 3
            if not args.no_cpu_baseline:
                base, rows, ref = cpu_baseline(app, W, H, t, args.cpu_row_stride)
                out["cpu_baseline"] = base
                par = parity(res["frame"][rows].cpu().numpy(), ref, len(rows))
                out["parity"]["oracle"] = par
                if not (par["max_abs_diff"] <= 1e-4):
                    status = 3
        res.pop("frame", None)
        res.pop("plans", None)
        torch.cuda.empty_cache()
        # BASELINE config 5 on the same ranks: APP_ATMOSPHERE and APP_PLANET 7680x4320 through the same FramePlan schedule
        if not args.no_other_configs and app == "clouds":
            others = []
            for oa, ow, oh in DIST_OTHER_CONFIGS:
                r2 = dist_frame_bench(R, dist, torch, dev, streams, args, oa, ow, oh, t, world, rank, min(args.steps, 10),
                                      min(args.warmup, 2))
                if rank == 0:
                    o2 = dist_line(r2, args, oa, ow, oh, t, world)
                    others.append({k: o2[k] for k in ("value", "unit", "ms_per_step", "steps", "value_serial", "serial", "steady_state",
                                                      "roofline", "phases", "parity", "exchange")} |
                                  {"workload": o2["config"]["workload"], "parallelism": o2["config"]["parallelism"],
                                   "kernel": KERNEL_OF.get(oa)})
                    if r2["mismatching_pixels"]:
                        status = 3
                r2.clear()
                torch.cuda.empty_cache()
            if rank == 0:
                out["other_configs"] = others
        if rank == 0:
            claim_stdout()(json.dumps(out))
        dist.barrier()
        dist.destroy_process_group()
        sys.exit(status)

    # ---- N = 1: the frame is one kernel launch ---------------------------------------------------------------
    frames = [torch.empty((H, W, 4), dtype=torch.float32, device=dev) for _ in range(ns)]

    def step(i=0):
        with torch.cuda.stream(streams[i % ns]):
            R.render(app, W, H, t, out=frames[i % ns])
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
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize(dev)
    step_done = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]     # completion of every timed frame
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
        step_done[i].record(streams[i % ns])
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    # per-launch kernel duration, HIP events on the launch stream (re-run outside the timed region, one launch at a time,
    # so that the event queries do not perturb it and the launches do not overlap)
    kernel_ms = []
    for _ in range(12):
        R.render(app, W, H, t, out=frames[0])
        kernel_ms.append(R.last_kernel_ms())
    kernel_ms = kernel_ms[2:]                            # median of >= 10 launches after 2 warm-ups
    torch.cuda.synchronize(dev)
    kmean = sorted(kernel_ms)[len(kernel_ms) // 2]
    pixels = W * H
    ms_per_step = elapsed * 1e3 / args.steps
    value = pixels / (ms_per_step * 1e-3) / 1e6
    pmc = pmc_counters(args, app, W, H, t) if args.pmc != "off" else None
    roofline, roofline_hbm = rooflines(app, pixels, pixels, kmean, min(kernel_ms), pmc)
    serial = round(pixels / (kmean * 1e-3) / 1e6, 3)
    out = {"metric": "Mpixels/s, APP_%s %dx%d" % (app.upper(), W, H), "value": round(value, 3),
           "unit": "Mpixels/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "APP_%s %dx%d u_time=%g u_mouse=0 default aux, fragCoord=(x+.5,y+.5)" % (app.upper(), W, H, t),
                      "frames_in_flight": ns, "parallelism": "1 GPU, one launch per frame",
                      "preroll": "%d untimed frames (>= %g ms) before the warm-up steps" % (preroll_frames, args.preroll_ms)},
           # `value` has frames_in_flight launches overlapping (the timed region's wall clock); `value_serial` is SURVEY.md 8d's
           # form: one un-overlapped launch, HIP events.  Compare like with like across N: value with value, serial with serial.
           "value_serial": serial,
           "serial": {"value": serial, "unit": "Mpixels/s", "what": "one un-overlapped launch (HIP events), %d pixels" % pixels},
           "steady_state": steady_state(step_done, ns, pixels),
           "roofline": roofline, "roofline_hbm": roofline_hbm}
    last_timed = frames[(args.steps - 1) % ns].clone() if not args.no_cpu_baseline else None
    if args.sustained_seconds > 0:
        out["sustained"] = sustained(torch, dev, step, ns, W * H, args.sustained_seconds, value, serial)
    if not args.no_cpu_baseline:
        base, rows, ref = cpu_baseline(app, W, H, t, args.cpu_row_stride)
        out["cpu_baseline"] = base
        # parity of the TIMED frame: the oracle rows just rendered against the same rows of the GPU frame
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
    claim_stdout()(json.dumps(out))
    sys.exit(status)


# ---------------------------------------------------------------------------------------------------------
# N > 1: one (app, size) through FramePlan on the ranks this process group has
# ---------------------------------------------------------------------------------------------------------
DIST_OTHER_CONFIGS = [("atmosphere", 7680, 4320), ("planet", 7680, 4320)]      # BASELINE config 5, both apps as written


def auto_groups(spec, payload_bytes_per_peer):
    """pieces the one exchange is issued in: 'auto' = one per ~12 MB of a peer's payload (a 4K CLOUDS slab goes out whole, an 8K
    slab in 3-4 pieces that leave while the rest renders), at most 8"""
    if spec not in ("auto", "0", 0):
        return max(1, int(spec))
    return max(1, min(8, int(-(-payload_bytes_per_peer // 12e6))))


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
    from shaderbox_amd import shard
    from shaderbox_amd.distributed import FramePlan
    ns = len(streams)
    br = args.block_rows
    fdist = dist
    if args.backend == "gloo":
        from shaderbox_amd.distributed import HostStagedDist
        fdist = HostStagedDist(dist, torch)

    def sync():
        dist.barrier()
        torch.cuda.synchronize(dev)

    def prepare(exchange, channels=None):
        """relief (calibrated on rank 0 for THIS exchange), payload, pieces and the ranks' plans"""
        if not hasattr(args, "channels_asked"):
            args.channels_asked = args.channels          # (auto overwrites args.channels with what it chose: later configs start from the flag again)
        channels = args.channels_asked if channels is None else channels
        relief = choose_relief(args.root_rounds, R, dist, torch, dev, app, W, H, t, br, world, rank, streams, exchange, channels)
        payload = 0
        if world > 1:
            if exchange in ("spans", "span_stores", "packed_stores"):
                payload = (16 if (exchange == "span_stores" and channels == 4) else 12) * int(max(R.span_table(app, W, H, t, br, world, relief[0], relief[1])[1][1:]))
            else:
                payload = (12 if (exchange in ("direct", "stores") and channels == 3) else 16) * W * shard.rank_rows_max(H, br, world, *relief)
        groups = auto_groups(args.gather_groups, payload)
        plans = [FramePlan(R, fdist, W, H, br, groups=groups, root_rounds=relief[0], rounds=relief[1], exchange=exchange,
                           channels=channels) for _ in range(ns)]
        return relief, payload, groups, plans

    # `--exchange auto` (default): the store exchange costs the root nothing but puts every pixel store on a link; the span exchange
    # sends fewer bytes but gives the root more to render; whole slabs cost the root a landing and a scatter; which one wins depends
    # on what the links deliver, and that is only known on the node — so all three are TRIED on the ranks at hand (a few pipelined frames
    # each, barrier + synchronize around them, the slowest rank's time) and the faster one runs the timed region.  With one rank
    # there is nothing to exchange: spans.
    trials = None
    if args.exchange != "auto":
        exchange = args.exchange
        relief, payload, groups, plans = prepare(exchange)
    elif world == 1:
        exchange = "spans"
        relief, payload, groups, plans = prepare(exchange)
    else:
        trials, best = {}, None
        # the clocks first (VERDICT r4 Weak #6: ~25 ms of launches until DVFS holds its clock) — or the form tried first pays for the
        # ramp: 4.8 against 2.5 ms per frame for the same work in a 2-process run on one GPU
        t_pre = time.perf_counter()
        scratch_pre = R.empty((H, W, 4))
        while (time.perf_counter() - t_pre) * 1e3 < max(args.preroll_ms, 40.0):
            for _ in range(4):
                R.render(app, W, H, t, out=scratch_pre)
            torch.cuda.synchronize(dev)
        del scratch_pre

        def agreed(ok):                                  # every rank's verdict on a step of a trial: all of them, or none
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=COLL_DEV or dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return bool(flag.item())
        # The store forms are tried with 12-byte stores (R, G, B of every float4 pixel: fewest bytes) AND with whole 16-byte pixels:
        # a link may take partial-line stores far below its rate — PCIe does, 11.9 against 51.8 GB/s (tools/time_link_stores.py,
        # profiles/r05_link_stores.txt) — and then the 16-byte form wins although it carries a third more.
        # (The form tried FIRST reads slow whatever it is — 4.7-11.5 ms per frame against 2.5 for the same work one trial later, in
        # 2-process runs on one GPU, pre-roll or not: first use of the mappings and of two processes' queues — so the first form is
        # tried twice and its first reading is thrown away.)
        for k_trial, (ex, ch) in enumerate((("stores", 3), ("stores", 3), ("stores", 4), ("span_stores", 3), ("span_stores", 4), ("packed_stores", None),
                                            ("spans", None), ("direct", None))):
            name = ex if ch in (None, 3) else ex + "_16B"
            if k_trial == 0:
                name = "(first trial, discarded) " + name
            # A form that cannot be set up on these devices (the store exchange needs HIP IPC and peer mapping), that faults, or
            # whose frame differs from one launch is DROPPED, on every rank alike, and the line says so: the trial must never
            # take the run down with it.
            cand, why = None, None
            try:
                cand = prepare(ex, ch)
            except Exception as e:                       # noqa: BLE001
                why = "set-up failed on rank %d: %s: %s" % (rank, type(e).__name__, str(e)[:200])
            if not agreed(cand is not None):
                trials[name] = "unavailable (%s)" % (why or "set-up failed on another rank")
                cand = None
                torch.cuda.empty_cache()
                continue
            cplans = cand[3]
            ms, why = None, None
            try:
                nwarm = 3 * ns                           # (first use of a form pays for mappings, code objects, the peers' first
                for i in range(nwarm):                   #  touch of a mapped frame: the form tried FIRST must not lose to that)
                    with torch.cuda.stream(streams[i % ns]):
                        cplans[i % ns].render(app, t)
                sync()
                if rank == 0:                            # the trial's own frame against one launch, bit for bit
                    whole = R.render(app, W, H, t)
                    got = cplans[(nwarm - 1) % ns].frame
                    torch.cuda.synchronize(dev)
                    if bool((got.view(torch.int32) != whole.view(torch.int32)).any().item()):
                        why = "its frame differs from a one-launch render"
                    del whole
                if R.fault_status() != 0:
                    why = "a wait of the exchange timed out (fault word)"
                if why is None:
                    t0 = time.perf_counter()
                    ktrial = 12
                    for i in range(ktrial):
                        with torch.cuda.stream(streams[i % ns]):
                            cplans[i % ns].render(app, t)
                    sync()
                    ms = (time.perf_counter() - t0) * 1e3 / ktrial
            except Exception as e:                       # noqa: BLE001
                why = "%s: %s" % (type(e).__name__, str(e)[:200])
            if not agreed(why is None):
                trials[name] = "dropped (%s)" % (why or "failed on another rank")
                try:
                    torch.cuda.synchronize(dev)
                    if R.fault_status() != 0:
                        R.clear_fault()
                except Exception:                        # noqa: BLE001
                    pass
                del cand, cplans
                torch.cuda.empty_cache()
                continue
            dt = torch.tensor([ms], dtype=torch.float64, device=COLL_DEV or dev)
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
            trials[name] = round(float(dt.item()), 4)
            if k_trial > 0 and (best is None or trials[name] < best[0]):
                best = (trials[name], ex, cand, ch)
            del cand, cplans
            torch.cuda.empty_cache()
        if best is None:
            raise SystemExit("no exchange form could be set up on these ranks: %s" % trials)
        exchange = best[1]
        relief, payload, groups, plans = best[2]
        args.channels = best[3] if best[3] is not None else args.channels_asked     # (what the rest of the run and the line's text say)
        best = None

    def step(i=0):
        with torch.cuda.stream(streams[i % ns]):
            plans[i % ns].render(app, t)              # the rank's launch(es) + the ONE exchange + assembly on rank 0

    for i in range(ns):                                 # builds the span layout, touches every buffer (page mapping)
        step(i)
    sync()
    # pre-roll, as at N = 1: frames until --preroll-ms have passed on rank 0 (every rank runs the same count) — the clocks, and the
    # first use of a form's mappings (a form asked for with --exchange has had no trial: its first frames read 2-4x slow)
    npre = torch.zeros(1, dtype=torch.int64, device=COLL_DEV or dev)
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
    for i in range(warmup):
        step(i)
    sync()
    step_done = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
        step_done[i].record(streams[i % ns])
    sync()
    elapsed = time.perf_counter() - t0
    # every rank's own launch, un-overlapped
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
    mine = torch.tensor([elapsed, sum(km) / len(km), min(km), float(rank_launch_pixels(R, app, W, H, t, br, world, rank, relief, exchange))],
                        dtype=torch.float64, device=COLL_DEV)
    allr = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allr, mine)
    phases = dist_phases(plans[0], torch, dist, dev, app, t, world, rank)
    res = {"relief": relief, "groups": groups, "payload_bytes_per_peer": payload, "ns": ns, "steps": steps, "warmup": warmup,
           "exchange": exchange, "exchange_trials_ms": trials}
    if rank == 0:
        per = [[float(x) for x in v] for v in allr]
        slow = max(range(world), key=lambda r: per[r][1])
        res.update({"elapsed": max(p[0] for p in per), "kmean": per[slow][1], "kmin": per[slow][2], "launch_pixels": int(per[slow][3]),
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


def bench_emulated(args, R, torch, dev, streams, app, W, H, t):
    """--emulate-ranks N on one GPU: see the option's help.  Everything printed as 'modelled' is max(root, slowest peer, link)
    of parts timed on THIS device one after the other; no second GPU, no link, no RCCL kernel was involved."""
    from shaderbox_amd import shard
    from shaderbox_amd.distributed import LoopbackWorld
    n, br = args.emulate_ranks, args.block_rows

    class OneRank:                                       # choose_relief's broadcast of rank 0's pick to itself
        @staticmethod
        def broadcast(tensor, src=0):
            return None

    def per_frame(fn, k=24):
        return timed_loop(torch, dev, fn, k)
    out_cfgs, status = [], 0
    cfgs = [(app, W, H)] + ([] if args.no_other_configs or app != "clouds" else DIST_OTHER_CONFIGS)
    for a, w, h in cfgs:
        frames = [torch.empty((h, w, 4), dtype=R.pixel_dtype, device=dev) for _ in range(max(2, len(streams)))]

        def whole(i):
            with torch.cuda.stream(streams[i % len(streams)]):
                R.render(a, w, h, t, out=frames[i % len(frames)])
        R.set_timing(False)
        p1 = per_frame(whole)
        # 'auto': both exchange forms are modelled, the faster one is reported (what the ranks of a real node decide by trying both)
        pick = None
        tried = {}
        # (the store forms with 12- and with 16-byte pixels, as the ranks of a node try them: a link that takes partial-pixel stores
        # below its rate — PCIe does, profiles/r05_link_stores.txt — makes the 16-byte reading the one that counts)
        forms = ((("stores", 3), ("stores", 4), ("span_stores", 3), ("span_stores", 4), ("packed_stores", None), ("spans", None), ("direct", None))
                 if args.exchange == "auto" else ((args.exchange, None),))
        pick16 = None
        for ex, chx in forms:
            name = ex if chx in (None, 3) else ex + "_16B"
            ch = ((chx or args.channels) if ex in ("stores", "span_stores") else 3) if ex != "gather" else 4
            relief = choose_relief(args.root_rounds, R, OneRank, torch, dev, a, w, h, t, br, n, 0, streams, ex, ch)
            R.set_timing(False)
            ranks_ms = [emulated_frame_ms(R, torch, dev, streams, frames, a, w, h, t, br, n, r, relief[0], relief[1], ex, ch, per_frame)
                        for r in range(n)]
            if ex in ("spans", "span_stores", "packed_stores"):
                pix = R.span_table(a, w, h, t, br, n, relief[0], relief[1])[1]
                payload = (4 if R.rgba8 else (16 if (ex == "span_stores" and ch == 4) else 12)) * int(max(pix[1:]))
            else:
                payload = (4 if R.rgba8 else (12 if ch == 3 else 16)) * w * shard.rank_rows_max(h, br, n, *relief)
            link_peak, link_real = payload / 76.8e9 * 1e3, payload / (args.link_gbps * 1e9) * 1e3
            modelled = max(max(ranks_ms), link_real)
            tried[name] = {"relief": "%d/%d" % relief, "root_ms": round(ranks_ms[0], 4), "slowest_peer_ms": round(max(ranks_ms[1:]), 4),
                           "bytes_per_peer": payload, "link_ms": round(link_real, 4), "modelled_ms_per_frame": round(modelled, 4),
                           "modelled_speedup": round(p1 / modelled, 3)}
            if pick is None or modelled < pick[0]:
                pick = (modelled, ex, relief, ch, ranks_ms, payload, link_peak, link_real)
            partial = ex in ("stores", "span_stores") and ch == 3 and not R.rgba8      # 12-byte stores at a 16-byte stride
            if not partial and (pick16 is None or modelled < pick16[0]):
                pick16 = (modelled, name, "%d/%d" % relief)
        modelled, exchange, relief, ch, ranks_ms, payload, link_peak, link_real = pick
        R.set_timing(True)
        # the frame of the N-rank schedule itself (FramePlans of all ranks, loopback transfers) against one launch
        world = LoopbackWorld(n)
        plans = world.plans(R, w, h, block_rows=br, groups=auto_groups(args.gather_groups, payload), root_rounds=relief[0],
                            rounds=relief[1], exchange=exchange if exchange != "gather" else "direct", channels=ch if ch in (3, 4) else args.channels)
        got = LoopbackWorld.render(plans, a, t)
        ref = R.render(a, w, h, t)
        torch.cuda.synchronize(dev)
        bad = int((got.view(torch.int32) != ref.view(torch.int32)).any(dim=-1).sum().item())
        status = 3 if bad else status
        out_cfgs.append({"workload": "APP_%s %dx%d u_time=%g" % (a.upper(), w, h, t), "n1_ms_per_frame_pipelined": round(p1, 4),
                         "relief": "%d/%d" % relief, "exchange": exchange, "exchanges_tried": tried, "pixel_format": args.format,
                         "bytes_per_peer": payload,
                         "bytes_moved_per_frame": world.bytes_moved,
                         "link_ms_at_76p8_GBps": round(link_peak, 4), "link_ms_at_%g_GBps" % args.link_gbps: round(link_real, 4),
                         "root_ms": round(ranks_ms[0], 4), "slowest_peer_ms": round(max(ranks_ms[1:]), 4),
                         "per_rank_ms": [round(v, 4) for v in ranks_ms],
                         "modelled_ms_per_frame": round(modelled, 4), "modelled_speedup": round(p1 / modelled, 3),
                         "without_partial_pixel_stores": None if pick16 is None else {
                             "exchange": pick16[1], "relief": pick16[2], "modelled_ms_per_frame": round(pick16[0], 4),
                             "modelled_speedup": round(p1 / pick16[0], 3),
                             "what": "the best form that stores or sends WHOLE pixels / packed slabs: what counts if a link takes 12-byte "
                                     "stores at a 16-byte stride below its rate (over PCIe: 4.3x below, profiles/r05_link_stores.txt)"},
                         "modelled_value_mpixels_s": round(w * h / (modelled * 1e-3) / 1e6, 1),
                         "bound": "link" if link_real >= max(ranks_ms) else ("root" if ranks_ms[0] >= max(ranks_ms[1:]) else "peer compute"),
                         "parity": {"against": "one-launch render of the same frame", "rows": h, "mismatching_pixels": bad}})
        del frames, plans, got, ref, world
        torch.cuda.empty_cache()
    head = out_cfgs[0]
    out = {"metric": "Mpixels/s, APP_%s %dx%d" % (app.upper(), W, H), "value": head["modelled_value_mpixels_s"], "unit": "Mpixels/s",
           "n_gpus": 1, "emulated_ranks": n, "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["modelled_ms_per_frame"],
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "value_is": "MODELLED for %d GPUs from parts timed on ONE: max(root's frame incl. landing and scatter, slowest peer's frame, link "
                       "time at %g GB/s), compute and transfer overlapped; not a measurement of %d GPUs" % (n, args.link_gbps, n),
           "landing_model": ("RCCL's grouped receive on the root = %d workgroups per peer resident for the link time at %g GB/s, writing the "
                             "payload at that pace (sbx_model_landing)" % (LANDING["wgs_per_peer"], LANDING["link_gbps"])) if LANDING
                            else "a device copy of the payload at HBM speed (round 4's stand-in)",
           "config": {"workload": head["workload"], "frames_in_flight": len(streams),
                      "parallelism": "cyclic %d-row blocks over %d EMULATED ranks on one device, exchange %s" % (br, n, args.exchange)},
           "emulated": out_cfgs}
    claim_stdout()(json.dumps(out))
    return status


def dist_line(res, args, app, W, H, t, world):
    """rank 0: the JSON object of one N > 1 measurement"""
    pixels = W * H
    relief, ns = res["relief"], res["ns"]
    ms_per_step = res["elapsed"] * 1e3 / res["steps"]
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
            "span_stores": "the peers' own %d-byte pixel stores of the SPANS of their row-blocks into the root's frame, mapped through HIP IPC "
                           "(the root renders its blocks and everything outside the spans; no RCCL call, no landing area, no scatter)"
                           % (12 if args.channels == 3 else 16),
            "packed_stores": "the peers' own stores of the packed 3-channel SPANS of their row-blocks (12 contiguous bytes per pixel) straight "
                             "into the root's landing area, mapped through HIP IPC; the root renders its blocks and everything outside the "
                             "spans, then scatters (no RCCL call, no receive kernels on the root)",
            "spans": "1 grouped RCCL send/recv of the peers' packed 3-channel SPANS (the root renders its blocks and everything "
                     "outside the spans in place)"}[res["exchange"]]
    if args.backend != "nccl":
        exch = exch.replace("RCCL", "gloo (host-staged, TEST form)")
    return {"metric": "Mpixels/s, APP_%s %dx%d" % (app.upper(), W, H), "value": round(pixels / (ms_per_step * 1e-3) / 1e6, 3),
            "unit": "Mpixels/s", "n_gpus": world, "steps": res["steps"], "warmup": res["warmup"],
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "APP_%s %dx%d u_time=%g u_mouse=0 default aux, fragCoord=(x+.5,y+.5)" % (app.upper(), W, H, t),
                       "frames_in_flight": ns,
                       "parallelism": "cyclic %d-row blocks over %d GPUs (root sits out rounds >= %d of %d) + %s (in %d pipelined "
                                      "pieces)%s" % (args.block_rows, world, relief[0], relief[1], exch, res["groups"],
                                                     "" if res["exchange"] in ("stores", "span_stores") else " + assemble")},
            "backend": "RCCL" if args.backend == "nccl" else "gloo with host-staged transfers (TEST form: ranks may share a GPU, nothing here "
                                                                "says anything about xGMI)",
            "exchange": {"kind": res["exchange"], "chosen": "measured on these ranks: ms per pipelined frame %s" % res["exchange_trials_ms"]
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
                                                      "spans": "one per peer of its packed 3-channel spans + one scatter kernel, rank 0 renders the rest",
                                                      "peer_stores": "none: every rank stores its pixels into rank 0's frame through peer access",
                                                      "blocks": "one per row-block into the final rows"}[args.lib_exchange])))},
           "steady_state": steady, "roofline": roofline, "roofline_hbm": roofline_hbm,
           "parity": {"against": "one-launch render of the same frame", "rows": H, "mismatching_pixels": bad}}
    status = 3 if bad else 0
    if not args.no_cpu_baseline:
        base, crow, ref = cpu_baseline(app, W, H, t, args.cpu_row_stride)
        out["cpu_baseline"] = base
        out["parity"]["oracle"] = parity(frames[(args.steps - 1) % ns][crow].cpu().numpy(), ref, len(crow))
        if not (out["parity"]["oracle"]["max_abs_diff"] <= 1e-4):
            status = 3
    claim_stdout()(json.dumps(out))
    M.close()
    return status


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
           "what": "the timed loop (same launches, same streams) held for %.1f s after the timed region, one synchronisation per %d frames; "
                   "sclk / power of THIS device (by PCI address) sampled every 20 ms from sysfs; first_half / second_half show whether "
                   "the rate drifts over seconds.  `value` (K frames after the pre-roll) within a per cent of this = the short window "
                   "measured the steady state" % (seconds, 8 * ns)}
    out.update(smp.summary())
    return out


def steady_state(step_done, ns, pixels):
    """The pipeline's rate without its ramp-in and its drain: with ns frames in flight the frames complete in bursts of about
    ns (they share the GPU), the first burst ends at ~ns frame times and the last burst drains on an emptying chip, so the
    rate is taken between the end of the first burst (frame ns - 1) and the end of the last burst that finishes at least ns
    frames before the end — a whole number of bursts (round 5: a window of 14 frames with 3 in flight read 6 % low)."""
    K = len(step_done)
    i1 = ns - 1
    i2 = i1 + ns * ((K - 1 - ns - i1) // ns)       # whole bursts only: a window that cuts a burst counts its wait, not its frames
    if i2 - i1 < 2:
        i2 = K - 1 - ns                             # too few timed frames for whole bursts: the plain window
    if i2 - i1 < 2:
        return None
    span_ms = step_done[i1].elapsed_time(step_done[i2])
    if not span_ms > 0:
        return None
    return {"value": round(pixels * (i2 - i1) / (span_ms * 1e-3) / 1e6, 3), "unit": "Mpixels/s",
            "ms_per_step": round(span_ms / (i2 - i1), 4),
            "what": "rank 0: the %d frames completed between timed frame %d and timed frame %d (events on the frames' streams): "
                    "neither the ramp-in of the first %d frames nor the drain of the last %d is in it" % (i2 - i1, i1, i2, ns, ns)}


def parity(gpu, ref, nrows):
    import numpy as np
    both_nan = np.isnan(gpu) & np.isnan(ref)
    d = np.where(both_nan, 0.0, np.abs(gpu.astype(np.float64) - ref.astype(np.float64)))
    d = np.nan_to_num(d, nan=np.inf)
    bits = (gpu.view(np.uint32) != ref.view(np.uint32)) & ~both_nan
    return {"against": "CPU oracle (oracle/), same frame", "rows": nrows, "pixels": int(gpu.shape[0] * gpu.shape[1]),
            "max_abs_diff": float(d.max()), "mismatching_pixels": int(bits.any(axis=-1).sum()), "tolerance": 1e-4}


def time_config(R, torch, dev, streams, app, W, H, t, steps=10, warmup=2, check_rows=0, pmc_mode="off", precision="exact"):
    """pipelined frames (as the headline) + un-overlapped kernel time of one config"""
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

    def step(i):
        with torch.cuda.stream(streams[i % ns]):
            R.render(app, W, H, t, out=frames[i % ns])
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
    # ... and a timed region of at least ~20 ms: ten 0.15 ms frames are 1.5 ms, of which the ramp-in of the first launches and the
    # final synchronisation are a fifth (EGG 1080p read 0.150 ms per frame that way against 0.121 over 60 frames)
    steps = max(steps, min(400, int(20.0 / max(est, .01))))
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize(dev)
    ms = (time.perf_counter() - t0) * 1e3 / steps
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
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
        if pmc:
            pmc["source"] = "live: one rocprofv3 --kernel-trace --pmc pass of `bench.py --app %s --steps 4 --warmup 1 --streams 1` in this run" % app
    if not pmc and pmc_mode != "off":
        pmc = pmc_committed(app, W, H)
    roofline, _ = rooflines(app, W * H, W * H, kmean, min(k), pmc)
    return {"workload": "APP_%s %dx%d u_time=%g" % (app.upper(), W, H, t), "value": round(W * H / (ms * 1e-3) / 1e6, 2),
            "unit": "Mpixels/s", "ms_per_step": round(ms, 4), "steps": steps, "frames_in_flight": ns,
            "kernel": KERNEL_OF.get(app), "kernel_ms": round(kmean, 4),
            "serial_value": round(W * H / (kmean * 1e-3) / 1e6, 2), "value_serial": round(W * H / (kmean * 1e-3) / 1e6, 2),
            "roofline": roofline,
            "hbm_store_gbps": round(16.0 * W * H / (kmean * 1e-3) / 1e9, 1), "parity": par}


def other_configs(R, torch, dev, streams, t, check_rows=16, pmc_mode="auto"):
    out = [time_config(R, torch, dev, streams, a, w, h, t, check_rows=check_rows, pmc_mode=pmc_mode) for a, w, h in OTHER_CONFIGS]
    # the labelled tolerance tier of APP_ATMOSPHERE, after the exact configs and never instead of one
    out.append(time_config(R, torch, dev, streams, "atmosphere", 7680, 4320, t, check_rows=check_rows, precision="1e-4"))
    return out


# ---------------------------------------------------------------------------------------------------------
# PMC-derived roofline fields
# ---------------------------------------------------------------------------------------------------------
PMC_PASSES = [("valu", ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVES", "GRBM_GUI_ACTIVE"]),
              ("busy", ["VALUBusy", "VALUUtilization"]),
              ("wr", ["WRITE_SIZE"]),
              ("rd", ["FETCH_SIZE"])]


def run_pmc_pass(counters, app, W, H, t, outdir, timeout=100):
    """one rocprofv3 counter pass (kernel-trace + pmc only) over a short serial run of this script; returns
    {counter: mean over the dispatches of the app's render kernel}"""
    import csv
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    cmd = [exe, "--kernel-trace", "-f", "csv", "--pmc"] + counters + ["-d", outdir, "-o", "pmc", "--", sys.executable,
           os.path.abspath(__file__), "--app", app, "--width", str(W), "--height", str(H), "--time", repr(t), "--steps", "4",
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
        got["source"] = "committed: profiles/" + os.path.basename(path) + ("" if path == exact else " (another frame size: per-pixel counts)")
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
    else:
        # no counters of THIS launch (rocprofv3 unusable, or a rank's strip at N > 1): the committed profile's instruction count
        # per pixel x this launch's pixels, against the nominal-clock peak
        r["achieved"] = round(nominal, 3)
        r["peak"] = round(PEAK_LANEOPS_NOMINAL_T, 2)
        r["frac"] = r["frac_unprofiled_duration"]
        r["frac_is"] += "; here from the committed per-pixel instruction count x this launch's pixels"
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
    mine = torch.tensor([sum(acc.get(n, [0.0])) / max(len(acc.get(n, [0.0])), 1) for n in names], dtype=torch.float64, device=COLL_DEV or dev)
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
    from shaderbox_amd import shard
    if world <= 1:
        return (1, 1)
    if spec != "auto":
        m0, m = (int(v) for v in spec.split("/"))
        return (m0, m)
    if exchange == "stores":
        return (1, 1)                                   # the root does nothing for the others: the plain split, nothing to calibrate
    pick = torch.zeros(2, dtype=torch.int64, device=COLL_DEV or dev)
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
    With LANDING set the landing is sbx_model_landing — `wgs_per_peer` workgroups per peer stay resident for as long as ONE peer's
    payload needs on its link (the peers arrive in parallel over their own links) and write all the bytes at that pace: the CUs and
    the HBM writes of RCCL's receive kernels.  Without: a device copy at HBM speed (round 4's stand-in, which holds the whole chip
    for a few microseconds instead of a few CUs for the link time)."""

    def __init__(self, R, torch, dev, nslots):
        self.R, self.t = R, torch
        # the side streams are made ONCE per process: HIP deals streams onto a few hardware queues in creation order, and a fresh set
        # per figure lands on other queues every time — some of them a render stream's, whose launches then wait behind a landing
        # kernel that is resident for the link time (the root's figures of one sweep came out bimodal, 1.45 / 2.4 ms)
        while len(SIDE_STREAMS) < nslots:
            SIDE_STREAMS.append(torch.cuda.Stream(device=dev))
        self.side = SIDE_STREAMS[:nslots]
        self.ev0 = [torch.cuda.Event() for _ in range(nslots)]
        self.ev1 = [torch.cuda.Event() for _ in range(nslots)]

    def begin(self, slot, dst, src, peers):
        t = self.t
        main = t.cuda.current_stream()
        self.ev0[slot].record(main)
        self.side[slot].wait_event(self.ev0[slot])
        with t.cuda.stream(self.side[slot]):
            n = src.numel() * src.element_size()
            if LANDING and n % 16 == 0 and n > 0 and peers > 0:
                us = n / peers / (LANDING["link_gbps"] * 1e9) * 1e6
                self.R.model_landing(src, dst, n, LANDING["wgs_per_peer"] * peers, us)
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
    of RCCL's receive kernels on their own stream) + the assembly kernel behind both; under the store exchange the root is an ordinary rank (its launch and the two
    flag kernels), and so is a peer (which renders in place into a frame on this device).  Used by the relief calibration on
    rank 0, by --emulate-ranks and by tools/strip_scaling.py; it knows nothing about the links."""
    from shaderbox_amd import shard
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


def cpu_baseline(app, W, H, t, stride):
    """The CPU oracle ('port' of the reference path, oracle/) on this host's cores, bounded sample.  Returns the
    baseline object, the row indices and the rendered rows (the parity check reuses them)."""
    from oracle.oracle import APP_IDS, Oracle
    o = Oracle()
    facts = host_cpu_facts()
    cores = facts["effective_cpus"]                      # threads used = CPUs this process may run on AND is allowed to keep busy
    # calibration (also warms threads and caches): 8 rows spread over the frame -> rows per second -> a ~12 s sample
    cal = [int((k + .5) * H / 8) for k in range(8)]
    t0 = time.perf_counter()
    o.render_rows(APP_IDS[app], W, H, t, cal, threads=cores)
    stride, rows = cpu_rows(H, stride, cores, rows_per_s=len(cal) / max(time.perf_counter() - t0, 1e-6))
    t0 = time.perf_counter()
    ref = o.render_rows(APP_IDS[app], W, H, t, rows, threads=cores)
    dt = time.perf_counter() - t0
    # one thread on a few of the same rows: the per-thread rate the all-thread figure can be read against
    one_rows = rows[len(rows) // 2:len(rows) // 2 + 2]
    t0 = time.perf_counter()
    o.render_rows(APP_IDS[app], W, H, t, one_rows, threads=1)
    dt1 = time.perf_counter() - t0
    value, one = len(rows) * W / dt / 1e6, len(one_rows) * W / dt1 / 1e6
    return ({"value": round(value, 4), "unit": "Mpixels/s", "cores": cores, "kind": "port",
             "sample": "%d of %d rows (every %dth row) of the same %dx%d frame in 64-pixel tiles, %.1f s, g++ -O2 -ffp-contract=off"
                       % (len(rows), H, stride, W, H, dt),
             "affinity": facts["affinity"], "os_cpu_count": facts["os_cpu_count"], "cgroup_cpu_max": facts["cgroup_cpu_max"],
             "cores_is": "threads used = min(scheduler affinity, cgroup CPU quota rounded up)",
             "one_thread": {"value": round(one, 5), "unit": "Mpixels/s", "sample": "%d rows, %.1f s" % (len(one_rows), dt1)},
             "thread_equivalents": round(value / one, 1) if one > 0 else None,
             "note": "the port evaluates sin/cos/exp/pow in binary64 by the sbx math spec (correctly rounded); the reference's own "
                     "headers over glibc libm ran about 2x faster per thread in the survey's probe (BASELINE.md: 0.112 vs 0.056 "
                     "Mpixels/s on the same 8 vCPU), so this understates the reference's C++ path by about that factor; "
                     "cpu_baseline_speed is the same port at the reference Makefile's optimisation level"}, rows, ref)


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


if __name__ == "__main__":
    main()
