#!/usr/bin/env python3
"""Build A/B variants of libsbx.so (run HERE, hipcc cross-compiles): one source recompiled with extra flags, linked with the
other objects of the normal build.

    python tools/ab_build.py name1:kern_clouds.hip:-DFOO=1,-DBAR name2:kern_planet.hip:-DX ...
    python tools/ab_build.py --all-variants        # every non-default setting of kern_clouds.hip's CL_*, kern_planet.hip's PL_*, kern_raytracer.hip's RT_* and kern_egg.hip's EGG_* switches, one at a time
-> build/ab/libsbx_<name>.so   (build/ is git-ignored but travels with gpurun); time them with tools/ab_time.py, check them
with tools/sweep_clouds_variants.py (same bits as the per-lane kernel on random frames).

--all-variants exists so that the A/B switches cannot rot: a switch whose other setting no longer compiles, or no longer
renders the same bits, is a bug or gets deleted.  Switches that change pixels ON PURPOSE are not in the list."""
import concurrent.futures
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from shaderbox_amd import build as b   # noqa: E402

# kern_clouds.hip: (switch, non-default value).  Not listed: CL_ABLATE_LIGHT (wrong pixels by design), CL_TW / CL_TX (tile
# shapes, timed in profiles/r01_tile_shapes.txt), CL_PARK_N (a size).
CLOUDS_VARIANTS = [("CL_PARK", 0), ("CL_LIPSKIP", 0), ("CL_LIPSKIP2", 0), ("CL_EPILOGUE_RELOAD", 0), ("CL_EXP_ASM", 0), ("CL_SEED", 0),
                   ("CL_NO_REG", 1), ("CL_EXP_LDS", 0), ("CL_NOTAB_GEN", 0), ("CL_MIN_WAVES_GEN", 4), ("CL_MAX3", 0),
                   ("CL_MIN_WAVES_YZ", 4), ("CL_YZ_MARCH", 0), ("CL_MIN_WAVES", 5), ("CL_EXP64", 0), ("CL_EXP_SMALL", 0),
                   ("CL_EXP_SMALL_ASM", 0), ("CL_YZ_SM", 0), ("CL_DIV3", 0), ("CL_EXP4K", 0), ("CL_TOP_FIRST", "true"),
                   ("CL_PRESCALE", 0)]


# kern_planet.hip / sbx_hashcache.h as k_planet uses it.  Not listed: PL_MIN_WAVES, PL_TW, PL_BATCH* (shapes and sizes).
PLANET_VARIANTS = [("PL_ATM_FIN", 0), ("PL_PAIRS", 0), ("PL_SPEC", 0), ("PL_TB2", 0), ("SBX_HC_MAGIC_SLOT", 0), ("PL_DIV3", 0), ("PL_MED3", 0), ("PL_EXP4K", 0),
                   ("PL_SQRT_RS", 0), ("PL_SQRT_N", 0), ("PL_PARK", 0), ("PL_TLAST", 0), ("PL_ROLL_DETAIL", 0)]


# kern_raytracer.hip
RT_VARIANTS = [("RT_AXIS_PLANES", 0), ("RT_WITNESS", 0), ("RT_LDS_FRAME", 0)]


# kern_egg.hip.  EGG_COOP 1 = round 6's survivor queue + finisher kernel (bit-exact, measured, not faster: shipped off, kept buildable)
EGG_VARIANTS = [("EGG_COOP", 1), ("EGG_HOT_FIRST", 0), ("EGG_WITNESS", 0), ("EGG_VCONST", 0)]


def build_one(spec):
    name, src, flags = (spec.split(":") + ["", ""])[:3]
    flags = [f for f in flags.split(",") if f]
    out = os.path.join(ROOT, "build", "ab")
    obj = os.path.join(out, "%s_%s.o" % (name, os.path.splitext(src)[0]))
    extra = [] if "NOEXTRA" in flags else b.EXTRA.get(src, [])          # NOEXTRA: without the source's own extras of build.py
    flags = [f for f in flags if f != "NOEXTRA"]
    cmd = [b.HIPCC] + b.FLAGS + extra + flags + ["-c", os.path.join(b.CSRC, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        return "FAILED %s %s\n%s" % (name, flags, r.stderr[-3000:])
    objs = [os.path.join(b.OBJ, os.path.splitext(s)[0] + ".o") if s != src else obj for s in b.SOURCES]
    lib = os.path.join(out, "libsbx_%s.so" % name)
    subprocess.run([b.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, check=True)
    return "built %s %s" % (lib, flags)


def main():
    b.build(verbose=False)
    os.makedirs(os.path.join(ROOT, "build", "ab"), exist_ok=True)
    specs = [a for a in sys.argv[1:] if not a.startswith("--")]
    if "--all-variants" in sys.argv:
        src = open(os.path.join(b.CSRC, "kern_clouds.hip")).read()
        for k, v in CLOUDS_VARIANTS:
            if ("#ifndef %s\n" % k) not in src and ("#ifndef %s " % k) not in src:
                print("(switch %s no longer exists)" % k)
                continue
            if v is None:
                continue
            specs.append("v_%s_%s:kern_clouds.hip:-D%s=%s" % (k.lower(), v, k, v))
        psrc = open(os.path.join(b.CSRC, "kern_planet.hip")).read() + open(os.path.join(b.CSRC, "sbx_hashcache.h")).read()
        for k, v in PLANET_VARIANTS:
            if ("#ifndef %s\n" % k) not in psrc and ("#ifndef %s " % k) not in psrc:
                print("(switch %s no longer exists)" % k)
                continue
            specs.append("v_pl_%s_%s:kern_planet.hip:-D%s=%s" % (k.lower(), v, k, v))
        rsrc = open(os.path.join(b.CSRC, "kern_raytracer.hip")).read()
        for k, v in RT_VARIANTS:
            if ("#ifndef %s\n" % k) not in rsrc and ("#ifndef %s " % k) not in rsrc:
                print("(switch %s no longer exists)" % k)
                continue
            specs.append("v_rt_%s_%s:kern_raytracer.hip:-D%s=%s" % (k.lower(), v, k, v))
        esrc = open(os.path.join(b.CSRC, "kern_egg.hip")).read()
        for k, v in EGG_VARIANTS:
            if ("#ifndef %s\n" % k) not in esrc and ("#ifndef %s " % k) not in esrc:
                print("(switch %s no longer exists)" % k)
                continue
            specs.append("v_egg_%s_%s:kern_egg.hip:-D%s=%s" % (k.lower(), v, k, v))
    failed = 0
    with concurrent.futures.ThreadPoolExecutor(max_workers=4) as ex:
        for msg in ex.map(build_one, specs):
            print(msg)
            failed += msg.startswith("FAILED")
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
