"""Build libsbx.so (HIP kernels + C ABI) for gfx950, in-tree.

    python -m shaderbox_amd.build          # incremental
    python -m shaderbox_amd.build --force  # rebuild everything

hipcc cross-compiles without a GPU.  Flags that are part of the math spec:
-ffp-contract=off (hipcc contracts a*b+c into v_fma by default, which would change results) and
no fast-math; fp32 denormals stay on and fp32 divide/sqrt stay correctly rounded (hipcc defaults).
"""
import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "lib", "libsbx.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-fno-slp-vectorize",
         "-fno-gpu-flush-denormals-to-zero", "-fhip-fp32-correctly-rounded-divide-sqrt", "-mfma",
         "-Wall", "-Wno-unused-function"]
SOURCES = ["kern_clouds.hip", "kern_clouds_tex.hip", "kern_egg.hip", "kern_raytracer.hip", "kern_atmosphere.hip", "kern_sdf_ao.hip",
           "kern_planet.hip", "kern_vinyl.hip", "kern_clouds_best.hip", "kern_clouds_ue4.hip", "kern_util.hip", "kern_noise.hip", "sbx_capi.hip", "sbx_multi.hip"]
# -fno-slp-vectorize: measured on MI355X, the SLP vectoriser's v_pk_{mul,add}_f32 are a net loss for every
# kernel here (CLOUDS 8.7 -> 6.5 ms, PLANET 57 -> 41 ms, SDF_AO 1.75 -> 1.37 ms at 4K/8K): a packed op issues
# in ~4.7 cycles against 2 x 2.9 for the scalar pair (profiles/r01_ubench_valu.txt), needs its constants in
# VGPR pairs (no literals), and the extra live registers cost a wave of occupancy.
EXTRA = {}
HEADERS = ["sbx_math.h", "sbx_vec.h", "sbx_frame.h", "sbx_device.h", "sbx_noise.h", "sbx_hashcache.h", "sbx_sdf.h", "../../include/sbx.h"]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src, force):
    obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
    deps = [os.path.join(CSRC, src)] + [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    if force or _newer(obj, deps):
        cmd = [HIPCC] + FLAGS + EXTRA.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
        return obj, True
    return obj, False


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    with concurrent.futures.ThreadPoolExecutor(max_workers=4) as ex:
        results = list(ex.map(lambda s: _compile(s, force), SOURCES))
    objs = [o for o, _ in results]
    if force or any(c for _, c in results) or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("built", LIB)
    elif verbose:
        print("up to date:", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
