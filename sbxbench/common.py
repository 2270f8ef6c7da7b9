"""sbxbench.common — constants of the measurement, the one-JSON-line contract, small helpers shared by the legs of bench.py."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
if ROOT not in sys.path:
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
PMC_ROUND = "r06"                   # committed per-launch counters: profiles/<PMC_ROUND>_pmc_<app>_<W>x<H>.json
# the other BASELINE.json configs that fit one GPU: (app, W, H) — C2, C3, C5 (both apps)
OTHER_CONFIGS = [("egg", 1920, 1080), ("raytracer", 3840, 2160), ("atmosphere", 7680, 4320), ("planet", 7680, 4320)]
DIST_OTHER_CONFIGS = [("atmosphere", 7680, 4320), ("planet", 7680, 4320)]      # BASELINE config 5, both apps as written
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
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), BENCH] + argv
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
