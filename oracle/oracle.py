"""ctypes front end of the CPU oracle (oracle/libsbx_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by the
cpu_baseline leg of bench.py.  The product package (shaderbox_amd/) never imports it.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

APP_PLANET, APP_CLOUDS, APP_VINYL, APP_EGG, APP_RAYTRACER, APP_ATMOSPHERE, APP_SDF_AO, APP_CLOUDS_BEST, APP_CLOUDS_TEX, APP_CLOUDS_UE4, APP_CLOUDS_SKY, APP_VINYL_GPU, APP_PLANET_ATMOSPHERE = range(13)
APP_IDS = {"planet": APP_PLANET, "clouds": APP_CLOUDS, "egg": APP_EGG, "raytracer": APP_RAYTRACER,
           "atmosphere": APP_ATMOSPHERE, "sdf_ao": APP_SDF_AO, "vinyl": APP_VINYL, "clouds_best": APP_CLOUDS_BEST,
           "clouds_tex": APP_CLOUDS_TEX, "clouds_ue4": APP_CLOUDS_UE4, "clouds_sky": APP_CLOUDS_SKY, "vinyl_gpu": APP_VINYL_GPU, "planet_atmosphere": APP_PLANET_ATMOSPHERE}


def build(variant="", subdir=""):
    """(Re)build the oracle library with oracle/Makefile; returns its path."""
    name = os.path.join(subdir, "libsbx_oracle%s.so" % variant)
    subprocess.run(["make", "-s", "-C", _HERE, name], check=True)
    return os.path.join(_HERE, name)


class Oracle:
    def __init__(self, variant="", rebuild=False, subdir=""):
        path = os.path.join(_HERE, subdir, "libsbx_oracle%s.so" % variant)
        if rebuild or not os.path.exists(path):
            path = build(variant, subdir)
        self.lib = ctypes.CDLL(path)
        fp = ctypes.POINTER(ctypes.c_float)
        self.lib.sbxo_main_image.argtypes = [ctypes.c_int, fp, ctypes.c_void_p, ctypes.c_float, ctypes.c_float, fp]
        self.lib.sbxo_render_rows.argtypes = [ctypes.c_int, fp, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int),
                                              ctypes.c_int, fp, ctypes.c_int]
        self.lib.sbxo_math.argtypes = [ctypes.c_char_p, fp, fp, fp, ctypes.c_long]
        self.lib.sbxo_kat.argtypes = [ctypes.c_char_p, fp, fp]
        self.lib.sbxo_noise.argtypes = [ctypes.c_char_p, fp, fp, fp, ctypes.c_long]
        self.lib.sbxo_worley_volume.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, fp]
        self.lib.sbxo_set_noise_volumes.argtypes = [ctypes.c_int, fp, ctypes.c_int, fp]
        self.lib.sbxo_tex3d.argtypes = [ctypes.c_int, fp, fp, fp, ctypes.c_long]
        self._volumes = None

    @staticmethod
    def _uni(width, height, time, mouse):
        return np.array([width, height, mouse[0], mouse[1], time], dtype=np.float32)

    @staticmethod
    def _fp(a):
        return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))

    @staticmethod
    def _aux(aux):
        if aux is None:
            return None, None
        buf = np.frombuffer(bytes(aux), dtype=np.uint8).copy()
        return buf, buf.ctypes.data_as(ctypes.c_void_p)

    def main_image(self, app, width, height, time, fx, fy, mouse=(0.0, 0.0), aux=None):
        u = self._uni(width, height, time, mouse)
        out = np.zeros(4, dtype=np.float32)
        keep, auxp = self._aux(aux)
        rc = self.lib.sbxo_main_image(int(app), self._fp(u), auxp, float(fx), float(fy), self._fp(out))
        if rc != 0:
            raise ValueError("oracle: unsupported app %r" % (app,))
        return out

    def render_rows(self, app, width, height, time, rows, mouse=(0.0, 0.0), aux=None, threads=None):
        """rows: iterable of global row indices (0 = bottom). Returns float32 [len(rows), W, 4]."""
        rows = np.ascontiguousarray(np.asarray(list(rows), dtype=np.int32))
        u = self._uni(width, height, time, mouse)
        out = np.zeros((len(rows), int(width), 4), dtype=np.float32)
        keep, auxp = self._aux(aux)
        if threads is None:
            threads = os.cpu_count() or 1
        rc = self.lib.sbxo_render_rows(int(app), self._fp(u), auxp,
                                       rows.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), len(rows),
                                       self._fp(out), int(threads))
        if rc != 0:
            raise ValueError("oracle: unsupported app %r" % (app,))
        return out

    def render(self, app, width, height, time, mouse=(0.0, 0.0), aux=None, threads=None):
        """Whole frame, float32 [H, W, 4], row 0 = bottom."""
        return self.render_rows(app, width, height, time, range(int(height)), mouse, aux, threads)

    def math(self, fn, a, b=None):
        a = np.ascontiguousarray(a, dtype=np.float32)
        b = a if b is None else np.ascontiguousarray(np.broadcast_to(np.asarray(b, dtype=np.float32), a.shape))
        out = np.empty_like(a)
        rc = self.lib.sbxo_math(fn.encode(), self._fp(a), self._fp(b), self._fp(out), a.size)
        if rc != 0:
            raise ValueError("oracle: unknown math function %r" % fn)
        return out

    def noise(self, fn, xyz, par=(0.0, 0.0, 0.0)):
        """library noise functions over points xyz[n,3] -> float32 [n,3]"""
        xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        par = np.ascontiguousarray(np.asarray(par, dtype=np.float32))
        out = np.zeros_like(xyz)
        if self.lib.sbxo_noise(fn.encode(), self._fp(xyz), self._fp(par), self._fp(out), len(xyz)) != 0:
            raise ValueError("oracle: unknown noise function %r" % fn)
        return out

    def set_noise_volumes(self, shape_rgba, detail_rgba):
        """Bind the two RGBA32F [size, size, size, 4] volumes of APP_CLOUDS' USE_NOISE_TEX build (t1 = shape, t2 = detail)."""
        a = np.ascontiguousarray(shape_rgba, dtype=np.float32)
        b = np.ascontiguousarray(detail_rgba, dtype=np.float32)
        assert a.ndim == 4 and a.shape[3] == 4 and a.shape[0] == a.shape[1] == a.shape[2]
        assert b.ndim == 4 and b.shape[3] == 4 and b.shape[0] == b.shape[1] == b.shape[2]
        self._volumes = (a, b)                      # the oracle keeps the pointers
        self.lib.sbxo_set_noise_volumes(a.shape[0], self._fp(a), b.shape[0], self._fp(b))

    def tex3d(self, rgba, xyz):
        """SampleLevel(linear, wrap, 0).r of the volume at points xyz[n, 3] -> float32 [n]"""
        v = np.ascontiguousarray(rgba, dtype=np.float32)
        xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        out = np.zeros(len(xyz), dtype=np.float32)
        self.lib.sbxo_tex3d(v.shape[0], self._fp(v), self._fp(xyz), self._fp(out), len(xyz))
        return out

    def worley_volume(self, size, z0=0, z1=None):
        z1 = size if z1 is None else z1
        out = np.zeros((z1 - z0, size, size, 4), dtype=np.float32)
        self.lib.sbxo_worley_volume(int(size), int(z0), int(z1), self._fp(out))
        return out

    def kat(self, name, args, nout):
        a = np.ascontiguousarray(np.asarray(args, dtype=np.float32))
        out = np.zeros(max(nout, 1), dtype=np.float32)
        rc = self.lib.sbxo_kat(name.encode(), self._fp(a), self._fp(out))
        if rc != 0:
            raise ValueError("oracle: unknown KAT %r" % name)
        return out[:nout]
