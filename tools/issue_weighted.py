#!/usr/bin/env python3
"""tools/issue_weighted.py [round] — the issue-cycle-weighted VALU occupancy of the five BASELINE kernels from the class counters of
profiles/<round>_apps_pmc.txt (tools/profile_baseline.sh): what roofline.issue_weighted of the bench line holds, as a table, plus the
STATIC split of the instructions no class counter names (from the hipcc -S listing of the kernel: how many of its compare / select /
min-max / floor / move / integer instructions issue at half rate).  Run HERE (no GPU): python tools/issue_weighted.py r06
-> profiles/<round>_issue_weighted.txt"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sbxbench.pmc import CLASS_COUNTERS, ISSUE_CYCLES, issue_weighted  # noqa: E402

rnd = sys.argv[1] if len(sys.argv) > 1 else "r06"
KERNELS = [("clouds", "k_clouds<", "kern_clouds.hip", "_ZN3sbx8k_cloudsILb1ELb1ELi1ELb1"), ("egg", "k_egg<", "kern_egg.hip", "_ZN3sbx5k_eggILb1ELi1"),
           ("raytracer", "k_raytracer<", "kern_raytracer.hip", "_ZN3sbx11k_raytracerILi1ELb1"),
           ("atmosphere", "k_atmosphere<", "kern_atmosphere.hip", "_ZN3sbx12k_atmosphereILb1ELi0"),
           ("planet", "k_planet<true, false>", "kern_planet.hip", "_ZN3sbx8k_planetILb1ELb0")]
WANT = ["SQ_INSTS_VALU", "GRBM_GUI_ACTIVE"] + CLASS_COUNTERS
cur, acc = None, {k[0]: {} for k in KERNELS}
for line in open(os.path.join(ROOT, "profiles", "%s_apps_pmc.txt" % rnd)):
    if line.startswith("kernel "):
        cur = next((k[0] for k in KERNELS if ("sbx::" + k[1]) in line), None)
        continue
    m = re.match(r"\s+(\S+)\s+dispatches=\d+ mean=(\S+)", line)
    if m and cur and m.group(1) in WANT:
        acc[cur][m.group(1)] = float(m.group(2))

# half-rate opcodes among those the class counters do not name (tools/ubench_issue.hip, profiles/r02_ubench_issue.txt: 4.1-4.3 cycles)
HALF = re.compile(r"v_(cmp|cmpx|cndmask|min|max|med3|floor|fract|trunc|rndne|ceil|lshl|lshr|ashr|lshlrev|lshrrev|ashrrev|mad_u|mad_i|mul_lo|mul_hi|"
                  r"bfe|bfi|perm|readlane|readfirstlane|writelane|mov_b32_dpp|ldexp|frexp|sad|alignbit|mbcnt|lshl_add|lshl_or|and_or|or3|add3|xad|"
                  r"cvt_pk|pk_)")
NAMED = re.compile(r"v_(add_f32|sub_f32|subrev_f32|mul_f32|fma_f32|fmac_f32|fmaak_f32|fmamk_f32|mad_f32|mac_f32|add_f64|mul_f64|fma_f64|"
                   r"rcp|rsq|sqrt|exp|log|sin|cos|cvt_)")
INT = re.compile(r"v_(add_u32|sub_u32|subrev_u32|add_co|addc|sub_co|subb|add_i32|and_b32|or_b32|xor_b32|not_b32|mul_u32_u24|mul_i32_i24|"
                 r"lshl|lshr|ashr|lshlrev|lshrrev|ashrrev|mad_u|mad_i|mul_lo|mul_hi|bfe|bfi|and_or|or3|add3|lshl_add|lshl_or|xad)")


def static_split(src, prefix):
    """(half-rate share of the listing's INT32-class VALU instructions, of its other un-named VALU instructions)"""
    from shaderbox_amd import build as b
    lst = os.path.join(ROOT, "build", "asm", os.path.splitext(src)[0] + ".s")
    os.makedirs(os.path.dirname(lst), exist_ok=True)
    cmd = [b.HIPCC] + [f for f in b.FLAGS if f != "-fPIC"] + b.EXTRA.get(src, []) + ["-S", "--cuda-device-only", os.path.join(b.CSRC, src), "-o", lst]
    if subprocess.run(cmd, capture_output=True).returncode != 0:
        return None
    lines = open(lst).read().split("\n")
    st = [i for i, l in enumerate(lines) if l.startswith(prefix) and ":" in l]
    if not st:
        return None
    end = next(i for i in range(st[0], len(lines)) if lines[i].startswith(".Lfunc_end"))
    ops = [l.split()[0] for l in lines[st[0]:end] if l.startswith("\t") and l.strip().startswith("v_")]
    ints = [o for o in ops if INT.match(o)]
    other = [o for o in ops if not NAMED.match(o) and not INT.match(o)]
    ih = sum(1 for o in ints if HALF.match(o)) / max(len(ints), 1)
    oh = sum(1 for o in other if HALF.match(o)) / max(len(other), 1)
    return ih, oh, len(ops), len(ints), len(other)


out = ["# Issue-cycle-weighted VALU occupancy of the BASELINE kernels (VERDICT r5 #2): sum over instruction classes of executed instructions x",
       "# issue cycles per wave64 instruction on a SIMD-32 / (1024 SIMDs x active cycles of the launch).  Counters: profiles/%s_apps_pmc.txt." % rnd,
       "# costs 'arch' = 2 (fp32 add / mul / fma), 4 (binary64, conversions, and the half-rate share of int32 / other), 8 (transcendental);",
       "# 'measured' = 2.25 / 4.2 / 8.2 (tools/ubench_issue.hip with 8 waves per SIMD; they include that loop's own overhead).",
       "# lo / hi: int32 and the un-named instructions all at full / all at half rate; 'static' prices them by the half-rate share of",
       "# those instruction kinds in the kernel's LISTING (every instruction of the kernel counted once: not an execution profile).",
       "# A figure near 1 says the VALU pipes were busy: the distance of roofline.frac from 1 is then the price of half-rate classes, not idle slots.",
       ""]
for app, pat, src, prefix in KERNELS:
    c = acc[app]
    if any(k not in c for k in WANT):
        out.append("%-10s (no class counters in the profile)" % app)
        continue
    w = issue_weighted(c, c["GRBM_GUI_ACTIVE"] / 8.0)
    cl = w["classes"]
    tot = c["SQ_INSTS_VALU"]
    out.append("%s   VALU instructions per launch %.4g, active SIMD-cycles %.4g, plain 2-cycle issue fraction %.3f" %
               (pat, tot, w["available_simd_cycles"], tot * 2.0 / w["available_simd_cycles"]))
    out.append("    classes (M): fp32 add/mul/fma %.1f  binary64+cvt %.1f  transcendental %.2f  int32 %.1f  other %.1f" %
               tuple(cl[k] / 1e6 for k in ("f32_add_mul_fma", "f64_and_cvt", "transcendental", "int32", "other_cmp_select_minmax_floor_mov")))
    out.append("    issue-weighted occupancy: arch costs lo %.3f hi %.3f | measured costs lo %.3f hi %.3f" %
               (w["frac_lo"], w["frac_hi"], w["frac_lo_at_measured_costs"], w["frac_hi_at_measured_costs"]))
    sp = static_split(src, prefix)
    if sp:
        ih, oh, nops, nint, noth = sp
        for tag, cst in ISSUE_CYCLES.items():
            cyc = (cst["full"] * cl["f32_add_mul_fma"] + cst["half"] * cl["f64_and_cvt"] + cst["quarter"] * cl["transcendental"] +
                   cl["int32"] * (cst["half"] * ih + cst["full"] * (1 - ih)) +
                   cl["other_cmp_select_minmax_floor_mov"] * (cst["half"] * oh + cst["full"] * (1 - oh)))
            out.append("    static split (%d VALU instructions in the listing; half-rate share of its %d int32-class: %.2f, of its %d other: %.2f) "
                       "-> %s costs %.3f" % (nops, nint, ih, noth, oh, tag, cyc / w["available_simd_cycles"]))
    out.append("")
path = os.path.join(ROOT, "profiles", "%s_issue_weighted.txt" % rnd)
open(path, "w").write("\n".join(out) + "\n")
print("\n".join(out))


def loop_budget(src, prefix, title):
    """per loop of the kernel's listing: VALU instructions by issue class, and the half-rate ones by opcode — where the half-rate
    issue cycles of an iteration come from (static: one iteration's instructions, whatever the lanes' branches)"""
    lst = os.path.join(ROOT, "build", "asm", os.path.splitext(src)[0] + ".s")
    lines = open(lst).read().split("\n")
    st = [i for i, l in enumerate(lines) if l.startswith(prefix) and ":" in l][0]
    end = next(i for i in range(st, len(lines)) if lines[i].startswith(".Lfunc_end"))
    # loop extents: a block label with "Loop Header: Depth=d" opens a loop; it runs to the last line "in Loop: Header=<that label>"
    heads = []
    for i in range(st, end):
        m = re.search(r"Loop Header: Depth=(\d+)", lines[i])
        if m:
            j = i
            while j > st and not re.match(r"^\.LBB\d+_\d+:", lines[j]):
                j -= 1
            lab = lines[j].split(":")[0].lstrip(".").lstrip("L")          # .LBB10_33 -> BB10_33, as the comments name it
            last = max([k for k in range(j, end) if re.search(r"(Header=|Loop )" + lab + r"\b", lines[k])] + [j])
            nxt = next((k for k in range(last + 1, end) if re.match(r"^\.LBB\d+_\d+:", lines[k])), end)
            heads.append((j, nxt, int(m.group(1)), lab))
    rows = ["## %s: VALU instructions of one pass over each loop of the listing, by issue class" % title,
            "#   loop (label, depth, listing lines)            total  fp32-full  f64  cvt  trans  int-full int-half  other-full other-half   half-rate opcodes (count)"]
    F32 = re.compile(r"v_(add_f32|sub_f32|subrev_f32|mul_f32|fma_f32|fmac_f32|fmaak_f32|fmamk_f32|mad_f32|mac_f32)")
    for a, b, depth, lab in heads:
        ops = [l.split()[0] for l in lines[a:b] if l.startswith("\t") and l.strip().startswith("v_")]
        if len(ops) < 40:
            continue
        import collections
        cnt = collections.Counter()
        halfops = collections.Counter()
        for o in ops:
            if F32.match(o):
                cnt["f32"] += 1
            elif re.match(r"v_(add_f64|mul_f64|fma_f64)", o):
                cnt["f64"] += 1
                halfops[o] += 1
            elif o.startswith("v_cvt_"):
                cnt["cvt"] += 1
                halfops[o] += 1
            elif re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)", o):
                cnt["trans"] += 1
            elif INT.match(o):
                k = "int_half" if HALF.match(o) else "int_full"
                cnt[k] += 1
                if k == "int_half":
                    halfops[o] += 1
            else:
                k = "oth_half" if HALF.match(o) else "oth_full"
                cnt[k] += 1
                if k == "oth_half":
                    halfops[o] += 1
        top = ", ".join("%s %d" % (o.replace("_e32", "").replace("_e64", ""), n) for o, n in halfops.most_common(8))
        rows.append("    %-14s d%d %5d-%-5d %14d %9d %5d %4d %5d %9d %8d %10d %10d   %s" %
                    (lab, depth, a - st, b - st, len(ops), cnt["f32"], cnt["f64"], cnt["cvt"], cnt["trans"], cnt["int_full"], cnt["int_half"],
                     cnt["oth_full"], cnt["oth_half"], top))
    return rows


extra = loop_budget("kern_clouds.hip", "_ZN3sbx8k_cloudsILb1ELb1ELi1ELb1", "k_clouds<true, true, 1, true> (the headline's kernel)")
open(path, "a").write("\n".join(extra) + "\n")
print("\n".join(extra))
