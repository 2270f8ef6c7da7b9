#!/usr/bin/env python3
"""HBM-bound helpers on one GPU: k_assemble (16 B read + 16 B write per pixel) and k_pack_unorm8 (16 B read + 4 B write),
torch events on the launch stream, median of 21."""
import sys
import torch
sys.path.insert(0, ".")
import shaderbox_amd

R = shaderbox_amd.Renderer(0)
for w, h in [(3840, 2160), (7680, 4320)]:
    f = torch.rand((h, w, 4), dtype=torch.float32, device="cuda")
    rows_max = shaderbox_amd.shard.rank_rows_max(h, 8, 8)
    g = torch.rand((8, rows_max, w, 4), dtype=torch.float32, device="cuda")   # stands in for the gathered slabs of 8 ranks
    out = torch.empty_like(f)
    for name, fn, nbytes in [("assemble", lambda: R.assemble(g, w, h, 8, 8, out=out), 32 * w * h),
                             ("pack_unorm8", lambda: R.pack_unorm8(f), 20 * w * h)]:
        fn(); torch.cuda.synchronize()
        ms = []
        for _ in range(21):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize()
            ms.append(a.elapsed_time(b))
        ms.sort()
        print("%-12s %5dx%-5d %7.3f ms  %7.1f GB/s" % (name, w, h, ms[10], nbytes / ms[10] / 1e6))
