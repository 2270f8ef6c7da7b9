"""sbxbench.emulate — --emulate-ranks: every rank's schedule of an N-rank frame through a loopback world on ONE device, checked
against one launch, timed with frames in flight, and MODELLED N-GPU figures with the exchange budget (n_gpus stays 1)."""
import json

from shaderbox_amd import tuning
from shaderbox_amd.tuning import auto_groups, choose_relief, emulated_frame_ms, timed_loop

from .common import DIST_OTHER_CONFIGS, claim_stdout


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
           "value_is": "MODELLED for %d GPUs from parts timed on ONE: max(root's frame incl. landing and scatter, slowest peer's frame, "
               "link "
                       "time at %g GB/s), compute and transfer overlapped; not a measurement of %d GPUs" % (n, args.link_gbps, n),
           "landing_model": ("RCCL's grouped receive on the root = %d workgroups per peer resident for the link time at %g GB/s, writing "
               "the "
                             "payload at that pace "
                                 "(sbx_model_landing)" % (tuning.CONFIG.landing["wgs_per_peer"], tuning.CONFIG.landing["link_gbps"])) if tuning.CONFIG.landing
                            else "a device copy of the payload at HBM speed (round 4's stand-in)",
           "config": {"workload": head["workload"], "frames_in_flight": len(streams),
                      "parallelism": "cyclic %d-row blocks over %d EMULATED ranks on one device, exchange %s" % (br, n, args.exchange)},
           "emulated": out_cfgs}
    claim_stdout()(json.dumps(out))
    return status
