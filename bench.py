#!/usr/bin/env python3
"""bench.py — headline benchmark: Mpixels/s of the mainImage() hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--app clouds] [--width 3840] [--height 2160]

Workload (BASELINE.json `metric`): APP_CLOUDS, 3840x2160, canonical frame u_time = 0.37, u_mouse = 0,
default aux uniforms.  One "step" = one whole frame rendered into an RGBA32F framebuffer resident in
HBM (nothing crosses PCIe inside the timed region).

`value` (N = 1) is SURVEY.md 8d's metric: the K timed frames launched ONE AT A TIME, back to back on one stream, wall clock between
two synchronisations / K — so the dominant kernel's time per step cannot exceed the step.  Frames are independent, and a second timed
region runs the same K frames pipelined over --streams HIP streams (one framebuffer per stream: the drain of one frame's kernel, its
last, longest waves, overlaps the start of the next; 1 / 2 / 3 / 4 in flight: 2 849 / 3 088 / 3 145 / 3 122 Mpixels/s): that is
`value_pipelined` / `ms_per_step_pipelined`, a throughput figure, not the metric (until round 5 it was `value`).
One-time initialisation (code-object load, APP_CLOUDS' y table, first submission on each stream, first touch of the
framebuffers, RCCL peer set-up) happens once before the W warm-up steps and is not a step.

N = 1 : the frame is one kernel launch.
N > 1 : one process per GPU (torch.distributed / RCCL; `--exchange auto` = shaderbox_amd.tuning.choose_exchange: the RCCL forms — span
        exchange, whole slabs — and, where allowed, the store forms are tried on the ranks at hand within --trial-budget-s and the fastest
        runs; `exchange.chosen` / `exchange.notes` in the line; `value_rccl_spans` / `value_rccl_direct` and the `rccl` object (version,
        nranks, one device per rank) are first-class keys whatever ran).  `python bench.py --gpus N` launched as a plain command starts
        its own N ranks (re-executes itself under torch.distributed.run on 127.0.0.1); launched by torch.distributed.run
        it uses the ranks it is given.  The SAME frame is sharded as cyclic 8-row blocks (shaderbox_amd/shard.py), every
        rank renders its blocks, ONE exchange over xGMI brings the slabs to rank 0, and one small kernel scatters them to
        their rows.  Total work is fixed -> "scaling": "strong".  Time = barrier + synchronize bracket, max over ranks.
        `value` at N > 1 = the THROUGHPUT of the K timed frames with --streams of them in flight (a frame at N > 1 contains the
        exchange, which a frame sequence overlaps with the next frame's rendering); the K frames one at a time = `value_one_at_a_time`.
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
                   issue_weighted      the VALU issue cycles the executed instruction MIX needs / the issue cycles that happened: class
                                       counters (SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F32 / _F64, _CVT, _INT32) x issue cycles per class
                                       (2 full rate, 4 half rate — binary64, conversions —, 8 transcendental; also at the costs measured
                                       by tools/ubench_issue.hip: 2.25 / 4.2 / 8.2) / (1024 SIMDs x active cycles).  The instructions
                                       no class counter names (compares, selects, min / max, floor, moves) and the integer class are
                                       priced at both ends: `frac_lo` (all full rate) ... `frac_hi` (all half rate).  Near 1 = the
                                       VALU pipes are busy and `frac` < 1 is the price of half-rate instruction classes, not idle slots.
  serial       : Mpixels/s of one un-overlapped launch bracketed by HIP events, median of 10 after 2 warm-ups; also as the top-level key
                 `value_serial`.  (`value` is the same launches by the wall clock of K of them back to back.)
  parity       : rows of the timed GPU frame against the CPU oracle's rows of the same frame (the ones cpu_baseline
                 renders): max |diff| and pixels with any differing bit.  > 1e-4 -> non-zero exit status.
                 N > 1: the assembled frame against a one-launch render of the same frame on rank 0 (bit-identical).
  cpu_baseline : the CPU restatement of the reference's headers OVER GLIBC LIBM (kind "port", oracle/libsbx_oracle_libm.so: the closest
                 thing here to the author's C++ / VML build, /root/reference/src/Makefile:12-16) timed on this host's cores on a bounded
                 sample of the same frame (every k-th row, dealt to the threads in 64-pixel tiles), strict build (g++ -O2
                 -ffp-contract=off).  `cores` = the threads used; `affinity`, `cgroup_cpu_max` and the one-thread rate say what those
                 threads could get.  Printed for every N (rank 0, after the timed region).
  cpu_baseline_port  : the same rows by the sbx math spec's port (binary64 transcendentals: what the kernels are bit-compared with; the
                 parity object's reference), ~2x slower.
  cpu_baseline_speed : the port with the optimisation level of the reference's own C++ build
                 (-O3 -march=native -funroll-loops, /root/reference/src/Makefile:12-13), compiled on this host at run time.
  other_configs: the other BASELINE.json GPU configs, timed the same way (pipelined frames + un-overlapped kernel time), each
                 with its own `parity` object: 16 evenly spread full rows of the rendered frame against the CPU oracle.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from sbxbench.common import (DIST_OTHER_CONFIGS, KERNEL_OF, OPS_PER_PIXEL, OTHER_CONFIGS, claim_stdout, launch_check, parity,  # noqa: E402,F401
                             self_launch, steady_state)
from sbxbench.cpu import cpu_baseline, cpu_baseline_port, cpu_baseline_speed, cpu_rows, host_cpu_facts  # noqa: E402,F401
from sbxbench.pmc import pmc_committed, pmc_counters, rooflines, run_pmc_pass  # noqa: E402,F401


def __getattr__(name):
    """the heavier legs on demand (they import torch-facing modules): bench.GpuSampler, bench.choose_relief, ... as before the split"""
    import importlib
    for mod in ("sbxbench.n1", "sbxbench.dist", "sbxbench.emulate", "shaderbox_amd.tuning"):
        m = importlib.import_module(mod)
        if hasattr(m, name):
            return getattr(m, name)
    raise AttributeError(name)


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
                    help="N>1 (engine dist): 'auto' (default) = shaderbox_amd.tuning.choose_exchange: the RCCL forms 'spans' and 'direct' — and, "
                         "where peer stores are allowed (SBX_ENABLE_PEER_STORES=1, or the ranks share one device) and a HIP-IPC pre-flight "
                         "passes, the store forms with 12- and 16-byte pixels — are soaked and timed on the ranks at hand within "
                         "--trial-budget-s, and the fastest runs; 'span_stores' = 'stores' with only the SPANS of the peers' "
                             "row-blocks "
                         "stored in place, the root renders the rest (fewer bytes on the links: 30 instead of 50 MB per peer at "
                             "7680x4320); 'stores' = the peers map the root's frame (HIP IPC) and render their row-blocks IN PLACE "
                         "into it: the exchange is their own pixel stores over xGMI (12 bytes per pixel with --channels 3), the root "
                             "lands, "
                         "receives and scatters nothing (distributed.py, include/sbx.h sbx_shared_*); 'packed_stores' = the span exchange "
                             "with the peers' own stores as transport: packed spans (12 contiguous bytes per pixel) straight into the "
                                 "root's mapped landing area, the root scatters; 'spans' = only the expensive interval of every row-block "
                                     "is dealt to the peers and "
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
                    help="--engine lib: 'spans' (default) = the span exchange inside the library; 'slabs' = one send/receive per peer of "
                        "its "
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
                    help="N = 1: back-to-back frames for at least this long BEFORE the warm-up steps (not steps, not timed): the first "
                        "~25 ms "
                         "of launches after host work run at ramping clocks (profiles/r04_streams3_trace.txt), and 5 warm-up frames are "
                             "11 ms")
    ap.add_argument("--sustained-seconds", type=float, default=2.5,
                    help="N = 1: after the timed region, frames back to back for this long with the shader clock and the board power "
                         "sampled beside them -> the `sustained` object (0 = skip)")
    ap.add_argument("--trial-budget-s", type=float, default=20.0,
                    help="N>1, --exchange auto: wall time (rank 0's clock) after which no further exchange form is tried, per config "
                         "(three configs: at most ~60 s; the first form always runs; what was cut is in exchange.notes)")
    ap.add_argument("--launch-check", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    from shaderbox_amd import tuning
    tuning.CONFIG.landing = {"wgs_per_peer": args.rccl_wgs_per_peer, "link_gbps": args.link_gbps} if args.rccl_wgs_per_peer > 0 else None
    if args.backend == "gloo" and args.exchange == "gather":
        raise SystemExit("--backend gloo stages point-to-point transfers only: use --exchange auto, spans or direct")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.engine == "lib" and args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        claim_stdout()
        from sbxbench.dist import bench_lib
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
    tuning.CONFIG.coll_dev = COLL_DEV = dev if args.backend == "nccl" else torch.device("cpu")      # where the small bookkeeping collectives live
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
            tuning.CONFIG.side_streams.append(torch.cuda.Stream(device=dev))
            with torch.cuda.stream(tuning.CONFIG.side_streams[-1]):
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
        from sbxbench.emulate import bench_emulated
        sys.exit(bench_emulated(args, R, torch, dev, streams, app, W, H, t))
    if use_dist:
        from sbxbench.dist import dist_frame_bench, dist_line
        res = dist_frame_bench(R, dist, torch, dev, streams, args, app, W, H, t, world, rank, args.steps, args.warmup)
        out = None
        if rank == 0:
            out = dist_line(res, args, app, W, H, t, world)
            if res["mismatching_pixels"]:
                status = 3
            if not args.no_cpu_baseline:
                base, rows, _ = cpu_baseline(app, W, H, t, args.cpu_row_stride)
                out["cpu_baseline"] = base
                out["cpu_baseline_port"], ref = cpu_baseline_port(app, W, H, t, rows)
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
                    others.append({k: o2[k] for k in ("value", "unit", "ms_per_step", "steps", "value_pipelined", "ms_per_step_pipelined", "value_serial", "serial", "steady_state",
                                                      "roofline", "phases", "parity", "exchange", "value_rccl_spans", "value_rccl_direct")} |
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
    from sbxbench.n1 import bench_n1
    out, status = bench_n1(args, R, torch, dev, streams, app, W, H, t)
    claim_stdout()(json.dumps(out))
    sys.exit(status)


if __name__ == "__main__":
    main()
