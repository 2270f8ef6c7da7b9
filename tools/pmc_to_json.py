#!/usr/bin/env python3
"""profiles/rNN_apps_pmc.txt (tools/profile_baseline.sh) -> profiles/rNN_pmc_<app>_<WxH>.json: the per-launch counters bench.py falls back
to at N > 1 or where rocprofv3 is not usable (bench.py pmc_committed).    python tools/pmc_to_json.py r05"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r06"
KERNELS = {"clouds": ("k_clouds<", 3840, 2160), "egg": ("k_egg<", 1920, 1080), "raytracer": ("k_raytracer<", 3840, 2160),
           "atmosphere": ("k_atmosphere<", 7680, 4320), "planet": ("k_planet<true, false>", 7680, 4320)}
WANT = ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVES", "GRBM_GUI_ACTIVE", "VALUBusy", "VALUUtilization", "WRITE_SIZE", "FETCH_SIZE",
        "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_INSTS_SALU",
        "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_CVT",
        "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"]
src = os.path.join(ROOT, "profiles", "%s_apps_pmc.txt" % rnd)
cur, acc = None, {k: {} for k in KERNELS}
for line in open(src):
    if line.startswith("kernel "):
        cur = next((k for k, (pat, _, _) in KERNELS.items() if ("sbx::" + pat) in line), None)
        continue
    m = re.match(r"\s+(\S+)\s+dispatches=\d+ mean=(\S+)", line)
    if m and cur and m.group(1) in WANT:
        acc[cur][m.group(1)] = float(m.group(2))
for app, (_, w, h) in KERNELS.items():
    if "SQ_INSTS_VALU" not in acc[app]:
        print("no counters for", app)
        continue
    out = {k: acc[app][k] for k in WANT if k in acc[app]}
    out["from"] = "profiles/%s_apps_pmc.txt (tools/profile_baseline.sh, the round's final kernels)" % rnd
    path = os.path.join(ROOT, "profiles", "%s_pmc_%s_%dx%d.json" % (rnd, app, w, h))
    json.dump(out, open(path, "w"), indent=1)
    print(path, out.get("SQ_INSTS_VALU"))
