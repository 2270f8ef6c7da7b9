#!/usr/bin/env python3
"""tools/egg_group_model.py — offline model of k_egg's cooperative finish from the per-pixel step map (tools/egg_steps_dump.py):
for a workgroup shape, when does each workgroup first have <= 64 rays marching, and how many wave-steps are left then.
    python tools/egg_group_model.py [gpurun_out/egg_steps_1920x1080.npz]"""
import sys
import numpy as np

f = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/egg_steps_1920x1080.npz"
z = np.load(f)
steps = z["steps"].astype(np.int32)
H, W = steps.shape
for (gw, gh, name) in [(64, 4, "64x4 strip"), (32, 8, "32x8 (2x2 tiles of 16x4)"), (16, 16, "16x16 (4 tiles stacked)")]:
    Hc, Wc = H // gh * gh, W // gw * gw
    g = steps[:Hc, :Wc].reshape(Hc // gh, gh, Wc // gw, gw).transpose(0, 2, 1, 3).reshape(Hc // gh, Wc // gw, gh * gw)
    mx = g.max(axis=2)
    for K0, DK in [(8, 4), (8, 8), (4, 4), (12, 4), (16, 8)]:
        # checkpoints K0, K0+DK, ...: first checkpoint i with 0 < S(i) <= 64, S(i) = #rays with steps >= i
        crit = np.zeros(mx.shape)          # model of the group's critical path in "normal step" units: normal steps + coop steps / 2.2
        coop_groups = 0
        coop_steps = 0
        late = []
        for i in range(K0, 80, DK):
            pass
        cps = list(range(K0, 80, DK))
        S = np.stack([(g >= i).sum(axis=2) for i in cps], axis=0)          # [ncp, gy, gx]
        ok = (S <= 64)
        first = np.where(ok.any(axis=0), ok.argmax(axis=0), len(cps))      # index of the first checkpoint with S <= 64
        ci = np.array(cps + [80])[first]                                    # the step it happens at
        Sat = np.take_along_axis(S, np.minimum(first, len(cps) - 1)[None], axis=0)[0]
        is_coop = (first < len(cps)) & (Sat > 0) & (mx >= ci)
        normal = np.where(is_coop, ci, np.minimum(mx + 1, 80))
        coop = np.where(is_coop, mx - ci + 1, 0)
        path = normal + coop / 2.2
        long_groups = (mx >= 40)
        print("%-26s K0 %2d DK %d: groups %d, with a cooperative finish %d (mean S %.1f); of the %d groups with a ray >= 40 steps: "
              "coop %d, start step p50 %d p90 %d max %d; critical path (normal-step units) max %.0f p99.9 %.0f (now: %d)"
              % (name, K0, DK, mx.size, is_coop.sum(), Sat[is_coop].mean(), long_groups.sum(), (is_coop & long_groups).sum(),
                 np.percentile(ci[is_coop & long_groups], 50), np.percentile(ci[is_coop & long_groups], 90), ci[is_coop & long_groups].max(),
                 path.max(), np.percentile(path, 99.9), min(mx.max() + 1, 80)))
