#!/usr/bin/env python3
"""tools/strip_scaling.py — ON ONE GPU: what each rank of an N-GPU frame would do, timed, and the exchange budget beside it.

    python tools/strip_scaling.py [--app clouds|atmosphere|planet|...] [--width 3840 --height 2160] [--ranks 2,4,8]
                                  [--exchanges stores,spans,direct] [--channels 3|4] [--rccl-wgs-per-peer 2] [--quick]

For N = 1 and every N asked for:
  * compute only: the un-overlapped launch of every rank's share (cyclic 8-row blocks) and the same with frames in flight;
  * the ROOT's frame emulated on this device (bench.py emulated_frame_ms: its launch + the landing of the peers' payload bytes —
    sbx_model_landing: --rccl-wgs-per-peer workgroups per peer resident for the link time, writing the bytes at the link's pace,
    the model of RCCL's grouped receive — + the assembly kernel), for every candidate root relief of bench.py's calibration, for
    the span exchange and the direct exchange; and for the STORE exchange (the peers render in place into the root's frame: the
    root is an ordinary rank, relief 1/1, no landing, no scatter; the link carries the pixel stores: 12 / 16 / 4 bytes per pixel);
  * the exchange budget: bytes per peer, time on one xGMI link at its 76.8 GB/s peak and at a stated realistic rate;
  * the modelled N-GPU frame time = max(root's frame, slowest peer's frame, link time) — compute and transfer fully overlapped,
    which needs >= 2 frames in flight or pipelined pieces — and the pessimistic one, peer + link back to back.
Nothing here has touched a second GPU: the link rates are assumptions, RCCL's own kernels on the root are a model (above)."""
import argparse
import importlib.util
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # as bench.py: streams that share a hardware queue do not overlap
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import shaderbox_amd  # noqa: E402
from shaderbox_amd import shard, tuning  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--app", default="clouds")
ap.add_argument("--width", type=int, default=3840)
ap.add_argument("--height", type=int, default=2160)
ap.add_argument("--time", type=float, default=.37)
ap.add_argument("--ranks", default="2,4,8")
ap.add_argument("--exchanges", default="stores,span_stores,packed_stores,spans,direct")
ap.add_argument("--channels", type=int, choices=[3, 4], default=3, help="store exchange, float pixels: dwords a peer stores per pixel")
ap.add_argument("--rccl-wgs-per-peer", type=int, default=2, help="landing model of the send/recv exchanges (bench.py); 0 = a plain device copy")
ap.add_argument("--streams", type=int, default=3)
ap.add_argument("--link-gbps", type=float, default=50.0, help="the 'realistic' per-direction rate of one xGMI link for the budget")
ap.add_argument("--quick", action="store_true", help="compute-only part")
ap.add_argument("--format", choices=["rgba32f", "rgba8"], default="rgba32f", help="pixels written and exchanged (include/sbx.h SBX_FORMAT_RGBA8)")
a = ap.parse_args()
if os.environ.get("SBX_LIB"):              # an A/B library of tools/ab_build.py instead of the shipped one
    shaderbox_amd.LIB_PATH = os.environ["SBX_LIB"]
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
tuning.CONFIG.landing = {"wgs_per_peer": a.rccl_wgs_per_peer, "link_gbps": a.link_gbps} if a.rccl_wgs_per_peer > 0 else None

app, W, H, t = a.app, a.width, a.height, a.time
dev = torch.device("cuda", 0)
R = shaderbox_amd.Renderer(0)
R.set_output_format(a.format)
BPP = 4 if R.rgba8 else 12                  # bytes per pixel on a link
R.set_timing(True)
streams = [torch.cuda.Stream() for _ in range(a.streams)]   # created once: HIP maps streams onto a few hardware queues
for s in streams:
    with torch.cuda.stream(s):
        R.render(app, 64, 36, t)
for _ in streams:                            # the emulated root's landing streams (bench.Landing), on the hardware queues after these
    tuning.CONFIG.side_streams.append(torch.cuda.Stream())
    with torch.cuda.stream(tuning.CONFIG.side_streams[-1]):
        R.render(app, 64, 36, t)
torch.cuda.synchronize()


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        fn(); ms.append(R.last_kernel_ms())
    ms.sort()
    return ms[len(ms) // 2]


def per_frame(fn, k=24):
    return bench.timed_loop(torch, dev, fn, k)          # (>= 60 ms per figure: a 24-call window of eighth-frames is mostly ramp and drain)


frames = [torch.empty((H, W, 4), dtype=R.pixel_dtype, device=dev) for _ in range(len(streams))]
t1 = timed(lambda: R.render(app, W, H, t, out=frames[0]))


def whole(i):
    with torch.cuda.stream(streams[i % len(streams)]):
        R.render(app, W, H, t, out=frames[i % len(frames)])


R.set_timing(False)
p1 = per_frame(whole)
R.set_timing(True)
print("pixel format %s (%d bytes per pixel on a link; store exchange: %d); landing model: %s"
      % (a.format, BPP, 4 if R.rgba8 else 4 * a.channels, tuning.CONFIG.landing or "device copy"))
print("%s %dx%d  N=1: one launch %.3f ms, %d frames in flight %.3f ms/frame" % (app, W, H, t1, len(streams), p1))
ranks = [int(v) for v in a.ranks.split(",") if v]
for n in ranks:
    slab = torch.empty((shard.rank_rows_max(H, 8, n), W, 4), dtype=R.pixel_dtype, device=dev)
    ts = [timed(lambda r=r: R.render_rank(app, W, H, t, 8, r, n, out=slab)) for r in range(n)]
    print("  N=%d one launch per rank (8-row blocks, whole rows): min %.3f max %.3f ms -> compute-only %.2fx (ideal %d)"
          % (n, min(ts), max(ts), t1 / max(ts), n))
    del slab
if a.quick:
    sys.exit(0)

R.set_timing(False)
for n in ranks:
    for exchange in a.exchanges.split(","):
        ch = a.channels if exchange in ("stores", "span_stores") else 3
        print("  --- N=%d, exchange %s" % (n, exchange))
        best = None
        for m0, m in (bench.relief_candidates() if exchange != "stores" else [(1, 1)]):
            if exchange == "stores":
                bpp = 4 if R.rgba8 else 4 * ch
                payload = bpp * W * shard.rank_rows_max(H, 8, n, m0, m)
                total = bpp * W * sum(shard.rank_rows(H, 8, r, n, m0, m) for r in range(1, n))
            elif exchange in ("spans", "span_stores", "packed_stores"):
                bpp = BPP if exchange in ("spans", "packed_stores") or R.rgba8 else 4 * ch
                _, pix, _ = R.span_table(app, W, H, t, 8, n, m0, m)
                payload = bpp * int(max(pix[1:]))
                total = bpp * sum(int(p) for p in pix[1:])
            else:
                payload = BPP * W * shard.rank_rows_max(H, 8, n, m0, m)
                total = BPP * W * sum(shard.rank_rows(H, 8, r, n, m0, m) for r in range(1, n))
            root_ms = bench.emulated_frame_ms(R, torch, dev, streams, frames, app, W, H, t, 8, n, 0, m0, m, exchange, ch, per_frame)
            peer_ms = max(bench.emulated_frame_ms(R, torch, dev, streams, frames, app, W, H, t, 8, n, r, m0, m, exchange, ch, per_frame)
                          for r in sorted({1, n - 1}))
            link_peak, link_real = payload / 76.8e9 * 1e3, payload / (a.link_gbps * 1e9) * 1e3
            over = max(root_ms, peer_ms, link_real)
            serial = max(root_ms, peer_ms + link_real)
            print("    relief %d/%d: root %.3f ms/frame, slowest peer %.3f, payload %.1f MB/peer (%.1f MB into the root): link %.3f ms at "
                  "76.8 GB/s, %.3f at %.0f GB/s -> modelled frame %.3f ms overlapped = %.2fx of N=1 pipelined (%.3f ms = %.2fx if "
                  "a peer's transfer only starts after its render)" % (m0, m, root_ms, peer_ms, payload / 1e6, total / 1e6, link_peak,
                                                                       link_real, a.link_gbps, over, p1 / over, serial, p1 / serial))
            if best is None or over < best[0]:
                best = (over, m0, m, root_ms, peer_ms, link_real)
            torch.cuda.empty_cache()
        print("    best for N=%d %s: relief %d/%d, %.3f ms/frame -> %.2fx (root %.3f, peer %.3f, link %.3f at %.0f GB/s)"
              % (n, exchange, best[1], best[2], best[0], p1 / best[0], best[3], best[4], best[5], a.link_gbps))
