#!/usr/bin/env python3
"""tools/egg_lone_wave.py — the critical path of k_egg's longest waves, measured: the census library (tools/egg_census.py --build)
finds the waves of a 1920x1080 launch that last longest; the SHIPPED library then renders the 64 pixels of each such tile ALONE
(one point-list launch = one wave on an idle chip) and the launch is timed with HIP events.  A launch cannot end before its
longest wave has run, so this is the floor of one un-overlapped launch.  Run on the GPU box."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
W, H, TW, TH = 1920, 1080, 16, 4
if "--find" in sys.argv:
    import numpy as np
    import torch
    import shaderbox_amd as sa
    sa.LIB_PATH = os.path.join(ROOT, "build", "ab", "libsbx_eggstats.so")
    r = sa.Renderer()
    for _ in range(5):
        a = r.render("egg", W, H, 0.37)
    a = r.render("egg", W, H, 0.37).cpu().numpy().view(np.uint32).reshape(H, W, 4)
    w0 = a[::TH, ::TW]
    dur = w0[..., 1].astype(np.int64)
    steps = (w0[..., 2] & 0xff).astype(np.int64)
    nsh = ((w0[..., 2] >> 8) & 0xff).astype(np.int64)
    order = np.argsort(dur.ravel())[::-1][:12]
    for o in order:
        ty, tx = divmod(int(o), dur.shape[1])
        print("TILE %d %d %d %d %d" % (tx, ty, dur[ty, tx], steps[ty, tx], nsh[ty, tx]))
    sys.exit(0)

out = subprocess.run([sys.executable, os.path.abspath(__file__), "--find"], capture_output=True, text=True).stdout
tiles = [tuple(int(v) for v in l.split()[1:]) for l in out.splitlines() if l.startswith("TILE")]
import numpy as np
import torch
import shaderbox_amd as sa
r = sa.Renderer()
r.set_timing(True)
frame = r.render("egg", W, H, 0.37)
for _ in range(50):
    r.render("egg", W, H, 0.37, out=frame)
torch.cuda.synchronize()
print("k_egg %dx%d: the twelve longest waves of one launch (census), each rendered alone by the shipped library" % (W, H))
print(" tile (x, y)   in the launch: us / trace steps / shadow lanes   alone: launch ms (min / median of 15)   same pixels")
for tx, ty, d, st, ns in tiles:
    ys, xs = np.divmod(np.arange(64), TW)
    pts = np.stack([tx * TW + xs + .5, ty * TH + ys + .5], axis=1).astype(np.float32)
    p = torch.from_numpy(pts).cuda()
    ms = []
    for _ in range(17):
        got = r.render_points("egg", W, H, 0.37, p)
        ms.append(r.last_kernel_ms())
    ms = sorted(ms[2:])
    ref = frame[ty * TH:(ty + 1) * TH, tx * TW:(tx + 1) * TW].reshape(64, 4)
    same = bool(torch.equal(got.view(torch.int32), ref.view(torch.int32)))
    print(" (%3d, %3d)     %7.1f / %2d / %2d                                   %.4f / %.4f                          %s"
          % (tx, ty, d * .01, st, ns, ms[0], ms[len(ms) // 2], same))
# an empty launch for the fixed cost
p = torch.tensor([[5.5, 1075.5]] * 64, dtype=torch.float32).cuda()          # sky pixels: a few steps
ms = sorted(r.last_kernel_ms() for _ in range(15) if r.render_points("egg", W, H, 0.37, p) is not None)
print(" a wave of sky pixels alone: %.4f ms (launch + event overhead)" % ms[len(ms) // 2])
