#!/usr/bin/env python3
"""Registers, scratch, LDS, occupancy and code size of every kernel, compiled with the flags of the product build
(shaderbox_amd/build.py FLAGS) — run HERE, hipcc cross-compiles.   python tools/kernel_resources.py [source.hip ...]
-S listings are left in build/asm/<source>.s (tools/isa_spills.py, tools/isa_mix.py read them)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from shaderbox_amd import build as b   # noqa: E402

out = os.path.join(ROOT, "build", "asm")
os.makedirs(out, exist_ok=True)
srcs = sys.argv[1:] or [s for s in b.SOURCES if s.startswith("kern_")]
print("%-62s %5s %5s %7s %6s %4s %8s" % ("kernel", "VGPR", "SGPR", "scratch", "LDS", "occ", "code B"))
for src in srcs:
    lst = os.path.join(out, os.path.splitext(src)[0] + ".s")
    cmd = [b.HIPCC] + [f for f in b.FLAGS if f != "-fPIC"] + b.EXTRA.get(src, []) + ["-S", "--cuda-device-only", os.path.join(b.CSRC, src), "-o", lst]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        print("FAILED", src, r.stderr[-2000:])
        continue
    txt = open(lst).read()
    # a kernel's label, then (after its body) its "; Kernel info:" block
    labels = [(m.start(), m.group(1)) for m in re.finditer(r"^(_Z\w+):", txt, re.M)]
    for m in re.finditer(r"; Kernel info:\n; codeLenInByte = (\d+)\n; TotalNumSgprs: (\d+)\n; NumVgprs: (\d+)\n.*?; ScratchSize: (\d+)\n.*?"
                         r"; LDSByteSize: (\d+).*?; Occupancy: (\d+)", txt, re.S):
        mangled = [n for pos, n in labels if pos < m.start()][-1]
        name = subprocess.run(["c++filt", mangled], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name).replace("void ", "")
        print("%-62s %5s %5s %7s %6s %4s %8s" % (name[:62], m.group(3), m.group(2), m.group(4), m.group(5), m.group(6), m.group(1)))
