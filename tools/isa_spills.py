#!/usr/bin/env python3
"""Where a kernel's scratch (VGPR spill) traffic sits: per basic block of a hipcc -S listing, the scratch loads/stores with the
loop depth of the block.   python tools/isa_spills.py listing.s mangled-prefix"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
start = [i for i, l in enumerate(lines) if l.startswith(sys.argv[2]) and ":" in l][0]
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
label, depth = "entry", 0
out = {}
for i in range(start, end):
    l = lines[i]
    m = re.match(r"^(\.LBB\d+_\d+|; %bb\.\d+):?", l)
    if m:
        label = m.group(1)
        d = re.search(r"Depth=(\d+)", l)
        depth = int(d.group(1)) if d else 0
    d2 = re.search(r"This (Inner )?Loop Header: Depth=(\d+)", l)
    if d2:
        depth = int(d2.group(2))
    if "scratch_" in l:
        k = (i - start, label, depth)
        out.setdefault((label, depth), []).append((i - start, l.strip().split()[0]))
for (label, depth), v in out.items():
    print("depth %d  %-14s line %5d  %s" % (depth, label, v[0][0], " ".join(x for _, x in v)))
