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
           "kern_planet.hip", "kern_vinyl.hip", "kern_clouds_best.hip", "kern_clouds_ue4.hip", "kern_util.hip", "kern_noise.hip", "sbx_capi.hip", "sbx_multi.hip", "sbx_shared.hip"]
# -fno-slp-vectorize: measured on MI355X, the SLP vectoriser's v_pk_{mul,add}_f32 are a net loss for every
# kernel here (CLOUDS 8.7 -> 6.5 ms, PLANET 57 -> 41 ms, SDF_AO 1.75 -> 1.37 ms at 4K/8K): a packed op issues
# in ~4.7 cycles against 2 x 2.9 for the scalar pair (profiles/r01_ubench_valu.txt), needs its constants in
# VGPR pairs (no literals), and the extra live registers cost a wave of occupancy.
# per-source extras.  kern_atmosphere.hip: the GCN max-ILP scheduling strategy — its march is long chains of binary64 fma (the exp
# cores) that the default occupancy-first strategy serialises more than it must: 7680x4320 3.51 -> 3.42 ms (2 in flight 3.46 ->
# 3.36), same bits.  Tried on every other kernel source (profiles/r04_log.md): no gain, or a loss (clouds_best +15 %, vinyl +6 %).
EXTRA = {"kern_atmosphere.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]}


def _headers():
    """Every header a translation unit can see: all of csrc/*.h (a glob, not a hand-kept list: round 2's list missed
    sbx_ldsframe.h) and the public include/*.h."""
    import glob
    inc = os.path.join(HERE, "..", "include")
    return sorted(glob.glob(os.path.join(CSRC, "*.h"))) + sorted(glob.glob(os.path.join(inc, "*.h")))


_HIPCC_VERSION = None


def _hipcc_version():
    global _HIPCC_VERSION
    if _HIPCC_VERSION is None:
        try:
            _HIPCC_VERSION = subprocess.run([HIPCC, "--version"], capture_output=True, text=True).stdout
        except OSError:
            _HIPCC_VERSION = "?"
    return _HIPCC_VERSION


def _key(src):
    """Content key of one translation unit: the source, every header under csrc/ and include/, the flags, the compiler path
    and the compiler's version string.  Staleness is
    decided by CONTENT, not by mtime: objects and the library travel to the GPU box outside git (an mtime rule can
    ship a stale binary after a checkout or a copy)."""
    import hashlib
    h = hashlib.sha256()
    for path in [os.path.join(CSRC, src)] + _headers():
        with open(path, "rb") as f:
            h.update(os.path.basename(path).encode())
            h.update(f.read())
    h.update(" ".join([HIPCC] + FLAGS + EXTRA.get(src, [])).encode())
    h.update(_hipcc_version().encode())
    return h.hexdigest()


def _compile(src, force):
    obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
    keyfile = obj + ".key"
    key = _key(src)
    old = open(keyfile).read().strip() if os.path.exists(keyfile) else ""
    if force or not os.path.exists(obj) or old != key:
        cmd = [HIPCC] + FLAGS + EXTRA.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
        with open(keyfile, "w") as f:
            f.write(key)
        return obj, True, key
    return obj, False, key


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    with concurrent.futures.ThreadPoolExecutor(max_workers=4) as ex:
        results = list(ex.map(lambda s: _compile(s, force), SOURCES))
    objs = [o for o, _, _ in results]
    libkey = "\n".join(k for _, _, k in results)
    libkeyfile = LIB + ".key"
    old = open(libkeyfile).read() if os.path.exists(libkeyfile) else ""
    if force or any(c for _, c, _ in results) or not os.path.exists(LIB) or old != libkey:
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        with open(libkeyfile, "w") as f:
            f.write(libkey)
        if verbose:
            print("built", LIB)
    elif verbose:
        print("up to date:", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
