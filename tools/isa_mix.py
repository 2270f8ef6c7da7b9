#!/usr/bin/env python3
"""Static instruction mix of one kernel in a hipcc -S listing: counts per opcode and, for VALU, per operand kind
(S = reads an SGPR/VCC/EXEC source, L = 32-bit literal, v = VGPR/inline constants only).  tools/ubench_issue.hip measured that
fp32 VALU operations with an SGPR source issue at half rate on gfx950 (4.2 vs 2.25 cycles per wave-instruction).

    python tools/isa_mix.py listing.s mangled-name-prefix [first_line last_line]"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
pref = sys.argv[2]
start = [i for i, l in enumerate(lines) if l.startswith(pref) and ":" in l][0]
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
if len(sys.argv) > 4:
    start, end = int(sys.argv[3]), int(sys.argv[4])
body = [l.strip() for l in lines[start:end] if l.startswith("\t") and not l.strip().startswith((".", ";"))]
c = collections.Counter()
kind = collections.Counter()
for l in body:
    op = l.split()[0]
    c[op] += 1
    if op.startswith("v_"):
        srcs = l[len(op):].split(";")[0].split(",")[1:]
        s = any(re.match(r"\s*-?\|?(s\d+|s\[\d+:\d+\]|vcc|exec)", x) for x in srcs)
        lit = any(re.match(r"\s*0x[0-9a-f]+", x) for x in srcs)
        kind[(op, "S" if s else ("L" if lit else "v"))] += 1
print(len(body), "instructions;", "VALU", sum(v for k, v in c.items() if k.startswith("v_")), "SALU",
      sum(v for k, v in c.items() if k.startswith("s_")), "DS", sum(v for k, v in c.items() if k.startswith("ds_")))
for k, v in c.most_common(40):
    print("%-28s %d" % (k, v))
agg = collections.Counter()
for (op, k), v in kind.items():
    agg[k] += v
print("---- VALU by operand kind:", dict(agg))
for (op, k), v in sorted(kind.items(), key=lambda x: -x[1])[:45]:
    print("%-26s %s %d" % (op, k, v))
