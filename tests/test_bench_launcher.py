"""bench.py's launcher and JSON contract.

CPU: `python bench.py --gpus N` started as a plain command (no torch.distributed.run around it, the shape the driver
uses at N = 1) must start its own N ranks; --launch-check makes the ranks rendezvous over gloo without touching a GPU.
GPU: the JSON line of a short run carries the contract keys, the parity record and the N > 1 code path (--force-dist)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args, timeout=900):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode == 0:
        # the contract: ONE JSON line on stdout and nothing else (RCCL's version banner, flushed at exit, used to follow it)
        assert len(r.stdout.strip().splitlines()) == 1, r.stdout[-600:]
    return r, (json.loads(lines[-1]) if lines else None)


@pytest.mark.parametrize("n", [2, 3])
def test_plain_command_starts_its_own_ranks(n):
    r, line = run_bench("--gpus", str(n), "--launch-check", "--steps", "3", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    assert line == {"launch_check": True, "n_gpus": n, "rank_sum": float(sum(range(n))), "self_launched": True}


def test_under_torchrun_uses_the_given_ranks():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29617", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["self_launched"] is False


@pytest.mark.gpu
def test_bench_line_contract_and_parity():
    r, line = run_bench("--steps", "10", "--warmup", "1", "--width", "640", "--height", "360", "--pmc", "off", "--cpu-row-stride", "8")
    assert r.returncode == 0, r.stderr[-2000:]
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "cpu_baseline_port", "cpu_baseline_speed", "parity", "serial",
              "value_serial", "value_pipelined", "ms_per_step_pipelined", "kernel_ms_in_timed_region", "other_configs"):
        assert k in line, k
    # `value` is SURVEY.md 8d's metric — the K frames one launch at a time — so a step cannot be shorter than the dominant kernel's
    # share of it; the pipelined throughput is a second, labelled figure
    assert line["config"]["frames_in_flight"] == 1 and line["frames_in_flight_pipelined"] == 3
    assert line["kernel_ms_in_timed_region"] <= line["ms_per_step"] * 1.02 and line["value_pipelined"] > 0
    assert all("value_pipelined" in c and c["ms_per_step"] > 0 for c in line["other_configs"])
    # the first CPU baseline is the restatement over glibc libm, the parity reference is the math spec's port
    assert "libm" in line["cpu_baseline"]["variant"] and "math spec" in line["cpu_baseline_port"]["variant"]
    assert line["cpu_baseline_port"]["value"] > 0 and line["cpu_baseline_speed"]["value"] > 0
    assert line["parity"]["mismatching_pixels"] == 0 and line["parity"]["max_abs_diff"] == 0.0 and line["parity"]["rows"] == 45
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] >= 1 and line["cpu_baseline"]["value"] > 0
    rf = line["roofline"]
    assert rf["bound"] == "valu" and rf["unit"] == "T lane-ops/s" and "frac_of_scalar_issue_ceiling" not in rf
    assert rf["useful_work_ratio"]["value"] > 0 and rf["useful_work_ratio"]["ops_per_pixel"] == 60248.0
    assert rf["frac"] is None or 0 < rf["frac"] <= 1.0            # --pmc off: no counters, or the committed file's (<= 1 either way)
    assert line["cpu_baseline"]["one_thread"]["value"] > 0 and "affinity" in line["cpu_baseline"] and "cgroup_cpu_max" in line["cpu_baseline"]
    assert line["steady_state"]["value"] > 0 and line["value_serial"] == line["serial"]["value"] > 0
    exact = [c for c in line["other_configs"] if "precision" not in c]
    assert [c["kernel"] for c in exact] == ["k_egg", "k_raytracer", "k_atmosphere", "k_planet"]
    assert all(c["value"] > 0 and c["kernel_ms"] > 0 for c in line["other_configs"])
    assert all(c["parity"]["rows"] == 16 and c["parity"]["mismatching_pixels"] == 0 for c in exact)
    # the labelled tolerance tier comes after the exact configs: within 1e-4, NOT bit-exact, and says so
    tier = [c for c in line["other_configs"] if "precision" in c]
    assert len(tier) == 1 and tier[0]["precision"] == "1e-4" and "TOLERANCE TIER" in tier[0]["workload"] and tier[0]["kernel"] == "k_atmosphere"
    assert tier[0]["parity"]["max_abs_diff"] <= 1e-4 and tier[0]["parity"]["mismatching_pixels"] > 0
    assert line["sustained"]["value"] > 0 and line["sustained"]["frames"] > 0 and "preroll" in line["config"]


@pytest.mark.gpu
def test_bench_multi_gpu_code_path_on_one_gpu():
    """--force-dist runs render_rank + RCCL gather + assemble with one rank; the record carries the frame comparison"""
    r, line = run_bench("--force-dist", "--steps", "10", "--warmup", "1", "--width", "640", "--height", "360")
    assert r.returncode == 0, r.stderr[-2000:]
    assert line["n_gpus"] == 1 and line["parity"]["mismatching_pixels"] == 0 and line["parity"]["rows"] == 360
    # the N > 1 line: per-rank phases, the steady-state rate beside the strict value, a roofline and the CPU baseline
    ph = line["phases"]["per_rank"]
    assert len(ph) == 1 and ph[0]["render_ms"] > 0 and ph[0]["exchange_wait_ms"] >= 0 and ph[0]["assemble_ms"] >= 0
    assert line["steady_state"]["value"] > 0 and line["steady_state"]["ms_per_step"] > 0
    assert line["roofline"]["bound"] == "valu" and (line["roofline"]["frac"] is None or line["roofline"]["frac"] <= 1.0)
    assert line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["cores"] >= 1
    assert line["parity"]["oracle"]["mismatching_pixels"] == 0
    # the default exchange is the span exchange; BASELINE config 5 runs through the same schedule (VERDICT r3 "Next" #1a)
    assert line["exchange"]["kind"] == "spans" and line["value_serial"] > 0      # one rank: auto means spans
    oc = line["other_configs"]
    assert [c["kernel"] for c in oc] == ["k_atmosphere", "k_planet"] and all("7680x4320" in c["workload"] for c in oc)
    assert all(c["parity"]["mismatching_pixels"] == 0 and c["value"] > 0 and c["phases"]["per_rank"][0]["render_ms"] > 0 for c in oc)
    r3, line3 = run_bench("--force-dist", "--exchange", "direct", "--steps", "4", "--warmup", "1", "--width", "640", "--height", "360",
                          "--no-cpu-baseline", "--no-other-configs")
    assert r3.returncode == 0 and line3["exchange"]["kind"] == "direct" and line3["parity"]["mismatching_pixels"] == 0


@pytest.mark.gpu
def test_bench_library_engine_on_one_gpu():
    """--engine lib: one process, the N-rank schedule of sbx_multi_* (ranks share the device on a 1-GPU box)"""
    r, line = run_bench("--gpus", "4", "--engine", "lib", "--steps", "10", "--warmup", "1", "--width", "640", "--height", "360")
    assert r.returncode == 0, r.stderr[-2000:]
    assert line["n_gpus"] == 4 and line["parity"]["mismatching_pixels"] == 0 and "sbx_multi" in line["config"]["engine"]
    assert line["steady_state"]["value"] > 0 and line["cpu_baseline"]["value"] > 0 and line["roofline"]["bound"] == "valu"
    r2, line2 = run_bench("--gpus", "4", "--engine", "lib", "--lib-exchange", "blocks", "--steps", "4", "--warmup", "1", "--width", "640",
                          "--height", "360", "--no-cpu-baseline")
    assert r2.returncode == 0 and line2["parity"]["mismatching_pixels"] == 0 and "per row-block" in line2["config"]["parallelism"]


@pytest.mark.gpu
def test_bench_emulated_ranks_on_one_gpu():
    """--emulate-ranks N: every rank's real schedule through a loopback world on the one GPU, frames checked against one launch,
    the N-GPU figures printed as MODELLED with the exchange budget beside them"""
    r, line = run_bench("--emulate-ranks", "4", "--width", "960", "--height", "540", "--no-other-configs")
    assert r.returncode == 0, r.stderr[-2000:]
    assert line["n_gpus"] == 1 and line["emulated_ranks"] == 4 and "MODELLED" in line["value_is"]
    e = line["emulated"][0]
    assert e["parity"]["mismatching_pixels"] == 0 and len(e["per_rank_ms"]) == 4 and e["modelled_speedup"] > 1.0
    assert e["bytes_moved_per_frame"] > 0 and e["bound"] in ("link", "root", "peer compute")
    # the store forms are modelled with 12- and with 16-byte pixels; the best form WITHOUT partial-pixel stores is named beside the pick
    assert all(k in e["exchanges_tried"] for k in ("stores", "stores_16B", "span_stores", "span_stores_16B", "packed_stores", "spans", "direct"))
    wp = e["without_partial_pixel_stores"]
    assert wp["exchange"] in ("stores_16B", "span_stores_16B", "packed_stores", "spans", "direct") and wp["modelled_speedup"] <= e["modelled_speedup"] + 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("n,exchange", [(2, "spans"), (3, "direct"), (2, "auto"), (3, "stores"), (2, "span_stores"), (3, "packed_stores")])
def test_bench_whole_multi_rank_program_on_one_gpu(n, exchange):
    """`bench.py --gpus N --backend gloo`: N real processes (self-launched ranks, rendezvous, the relief calibration and its broadcast,
    FramePlan's schedule with its pieces, the all_gathers of the per-rank figures, config 5 at 7680x4320 through the same schedule,
    phases, parity) with the ranks SHARING the box's one GPU — RCCL refuses duplicate devices, so the transfers are staged through the
    host (distributed.HostStagedDist).  Everything but the transport is the code the driver's N = 2, 4, 8 runs execute."""
    args = ["--gpus", str(n), "--backend", "gloo", "--exchange", exchange, "--steps", "4", "--warmup", "1", "--width", "960",
            "--height", "540", "--no-cpu-baseline"]
    if n == 3 or exchange in ("auto", "span_stores"):
        args.append("--no-other-configs")
    r, line = run_bench(*args, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    assert line["n_gpus"] == n and "gloo" in line["backend"]
    if exchange == "auto":
        # shaderbox_amd.tuning.choose_exchange: the RCCL forms first; the ranks share the one GPU here, so the store forms are
        # candidates too (after the HIP-IPC pre-flight); whatever the time budget cut is named
        assert line["exchange"]["kind"] in ("stores", "span_stores", "packed_stores", "spans", "direct") and "measured on these ranks" in line["exchange"]["chosen"]
        notes = line["exchange"]["notes"]
        assert notes["candidates"][:2] == ["spans", "direct"] and "share one device" in notes["stores"]
        tried = [k for k in ("stores", "stores_16B", "span_stores", "span_stores_16B", "packed_stores", "spans", "direct")
                 if k in line["exchange"]["chosen"] or k in notes["cut"]]
        assert tried == ["stores", "stores_16B", "span_stores", "span_stores_16B", "packed_stores", "spans", "direct"]
        assert "spans" in line["exchange"]["chosen"]
    else:
        assert line["exchange"]["kind"] == exchange
    # the first-contact contract: the RCCL forms' figures and who the ranks are, as first-class keys; under gloo
    # (a test transport) the RCCL fields are null, the identity of the ranks is not
    assert "value_rccl_spans" in line and "value_rccl_direct" in line and line["value_rccl_spans"] is None and line["value_rccl_direct"] is None
    assert line["rccl"]["version"] is None and line["rccl"]["nranks"] is None and line["rccl"]["in_timed_region"] is False
    assert [d["rank"] for d in line["rccl"]["devices"]] == list(range(n)) and all("device" in d and "host" in d for d in line["rccl"]["devices"])
    assert line["parity"]["mismatching_pixels"] == 0 and line["parity"]["rows"] == 540
    assert len(line["phases"]["per_rank"]) == n and all(p["render_ms"] > 0 for p in line["phases"]["per_rank"])
    assert line["value"] > 0 and line["value_serial"] > 0 and line["roofline"]["bound"] == "valu"
    # N > 1: `value` is the throughput of the frame sequence (frames in flight), a frame's latency stands beside it
    assert line["value"] == line["value_pipelined"] and line["ms_per_step"] == line["ms_per_step_pipelined"]
    assert line["value_one_at_a_time"] > 0 and line["ms_per_step_one_at_a_time"] > 0 and "THROUGHPUT" in line["value_is"]
    assert line["config"]["frames_in_flight"] == line["frames_in_flight_pipelined"] >= 1
    if n == 2 and exchange == "spans":
        oc = line["other_configs"]
        assert [c["kernel"] for c in oc] == ["k_atmosphere", "k_planet"]
        assert all(c["parity"]["mismatching_pixels"] == 0 and len(c["phases"]["per_rank"]) == 2 for c in oc)
        # spans: whatever relief the calibration picked (down to a root without rows, when the peer holds every span of the frame),
        # well under the frame's RGB bytes
        assert all(c["exchange"]["bytes_per_peer"] < 0.7 * 12 * 7680 * 4320 for c in oc)
