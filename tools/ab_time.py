#!/usr/bin/env python3
"""Time A/B variants built by tools/ab_build.py on the GPU box and check that they produce the same bits.

    python tools/ab_time.py [--app clouds --width 3840 --height 2160 --reps 30] base name1 name2 ...
('base' = the normal shaderbox_amd/lib/libsbx.so)

EVERY variant runs in its own process: two builds of libsbx in one process register kernels of the same name, and the HIP
runtime then launches one of them for both (measured in round 2: the 'A/B' of one process reports B's time for A)."""
import argparse
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("--app", default="clouds")
ap.add_argument("--width", type=int, default=3840)
ap.add_argument("--height", type=int, default=2160)
ap.add_argument("--time", type=float, default=.37)
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--lds-pad", default="", help="comma list of SBX_DEBUG_LDS_PAD values to sweep for every library (occupancy experiment)")
ap.add_argument("--one", default="", help=argparse.SUPPRESS)
ap.add_argument("--ref", default="", help=argparse.SUPPRESS)
ap.add_argument("names", nargs="*")
a = ap.parse_args()

if not a.one:
    ref = "/tmp/sbx_ab_ref_%d.pt" % os.getpid()
    pads = [p for p in a.lds_pad.split(",") if p] or [""]
    for name in a.names:
        for pad in pads:
            env = dict(os.environ)
            if pad:
                env["SBX_DEBUG_LDS_PAD"] = pad
            subprocess.call([sys.executable, os.path.abspath(__file__), "--app", a.app, "--width", str(a.width), "--height", str(a.height),
                             "--time", repr(a.time), "--reps", str(a.reps), "--one", name, "--ref", ref], env=env)
    if os.path.exists(ref):
        os.remove(ref)
    sys.exit(0)

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402
import shaderbox_amd  # noqa: E402

name = a.one
path = shaderbox_amd.LIB_PATH if name == "base" else os.path.join(ROOT, "build", "ab", "libsbx_%s.so" % name)
label = name + ("+pad" + os.environ["SBX_DEBUG_LDS_PAD"] if os.environ.get("SBX_DEBUG_LDS_PAD") else "")
if not os.path.exists(path):
    print("%-24s missing" % label)
    sys.exit(0)
shaderbox_amd.LIB_PATH = path
R = shaderbox_amd.Renderer(0)
R.set_timing(True)
if a.app == "clouds_tex":          # the two baked volumes of APP_CLOUDS' USE_NOISE_TEX build (128^3 shape, 64^3 detail)
    R.set_noise_volumes(R.worley_volume(128), R.worley_volume(64))
    torch.cuda.synchronize()
out = torch.empty((a.height, a.width, 4), dtype=torch.float32, device="cuda")
for _ in range(5):
    R.render(a.app, a.width, a.height, a.time, out=out)
torch.cuda.synchronize()
ms = []
for _ in range(a.reps):
    R.render(a.app, a.width, a.height, a.time, out=out)
    ms.append(R.last_kernel_ms())
# back to back on one stream (steady state), then pipelined over two streams as bench.py does
R.set_timing(False)
t0 = time.perf_counter()
for i in range(100):
    R.render(a.app, a.width, a.height, a.time, out=out)
torch.cuda.synchronize()
b2b = (time.perf_counter() - t0) * 1e3 / 100
st = [torch.cuda.Stream(), torch.cuda.Stream()]
outs = [out, torch.empty_like(out)]
for i in range(4):
    with torch.cuda.stream(st[i % 2]):
        R.render(a.app, a.width, a.height, a.time, out=outs[i % 2])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(100):
    with torch.cuda.stream(st[i % 2]):
        R.render(a.app, a.width, a.height, a.time, out=outs[i % 2])
torch.cuda.synchronize()
pipe = (time.perf_counter() - t0) * 1e3 / 100
same = ""
if a.ref:
    if not os.path.exists(a.ref):
        torch.save(out.cpu(), a.ref)
    else:
        r = torch.load(a.ref)
        o = out.cpu()
        same = "same bits" if torch.equal(r.view(torch.int32), o.view(torch.int32)) else \
            "DIFFERENT PIXELS: %d" % int((r.view(torch.int32) != o.view(torch.int32)).any(-1).sum())
ms.sort()
print("%-24s serial ms: min %.3f  median %.3f | back-to-back %.3f | 2 in flight %.3f ms/frame  %s"
      % (label, ms[0], ms[len(ms) // 2], b2b, pipe, same))
