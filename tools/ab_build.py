#!/usr/bin/env python3
"""Build A/B variants of libsbx.so (run HERE, hipcc cross-compiles): one source recompiled with extra flags, linked with the
other objects of the normal build.

    python tools/ab_build.py name1:kern_clouds.hip:-DFOO=1,-DBAR name2:kern_planet.hip:-DX ...
-> build/ab/libsbx_<name>.so   (build/ is git-ignored but travels with gpurun); time them with tools/ab_time.py."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from shaderbox_amd import build as b   # noqa: E402

b.build(verbose=False)
out = os.path.join(ROOT, "build", "ab")
os.makedirs(out, exist_ok=True)
for spec in sys.argv[1:]:
    name, src, flags = (spec.split(":") + ["", ""])[:3]
    flags = [f for f in flags.split(",") if f]
    obj = os.path.join(out, "%s_%s.o" % (name, os.path.splitext(src)[0]))
    cmd = [b.HIPCC] + b.FLAGS + flags + ["-c", os.path.join(b.CSRC, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        print("FAILED", name, r.stderr[-3000:])
        continue
    objs = [os.path.join(b.OBJ, os.path.splitext(s)[0] + ".o") if s != src else obj for s in b.SOURCES]
    lib = os.path.join(out, "libsbx_%s.so" % name)
    subprocess.run([b.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, check=True)
    print("built", lib, flags)
