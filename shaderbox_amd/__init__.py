"""shaderbox_amd — MI355X-native renderer for shaderbox's mainImage() hot path.

Python host layer over the C ABI of libsbx.so (include/sbx.h).  PyTorch is used only as plumbing:
device memory (framebuffers are torch CUDA tensors), streams and torch.distributed.  All pixels
are produced by the hand-written HIP kernels in shaderbox_amd/csrc; there is no CPU or PyTorch
fallback — if the library or a gfx950 device is missing, construction raises.

The surface mirrors the reference's: an app is chosen by its APP_* project define
(/root/reference/README.md:11-22), a frame is a pure function of the uniforms u_res / u_time /
u_mouse (+ the per-app aux block with the defaults of /root/reference/src/uniform_buffer.h:39-60).
"""
import ctypes
import os

from . import shard  # noqa: F401  (pure-python row-block arithmetic, no GPU needed)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libsbx.so")

APP_PLANET, APP_CLOUDS, APP_VINYL, APP_EGG, APP_RAYTRACER, APP_ATMOSPHERE, APP_SDF_AO, APP_CLOUDS_BEST, APP_CLOUDS_TEX, APP_CLOUDS_UE4, APP_CLOUDS_SKY, APP_VINYL_GPU, APP_PLANET_ATMOSPHERE = range(13)
APPS = {"APP_PLANET": APP_PLANET, "APP_CLOUDS": APP_CLOUDS, "APP_VINYL": APP_VINYL, "APP_EGG": APP_EGG,
        "APP_RAYTRACER": APP_RAYTRACER, "APP_ATMOSPHERE": APP_ATMOSPHERE, "APP_SDF_AO": APP_SDF_AO,
        "APP_CLOUDS_BEST": APP_CLOUDS_BEST,    # src/app_clouds_best.h (stand-alone shader, not an APP_* define)
        "APP_CLOUDS_TEX": APP_CLOUDS_TEX,      # APP_CLOUDS + USE_NOISE_TEX (src/app_clouds.h:9)
        "APP_CLOUDS_UE4": APP_CLOUDS_UE4,      # ue4/volumetric_clouds/Shaders/app_clouds.usf (host mapping: include/sbx.h)
        "APP_CLOUDS_SKY": APP_CLOUDS_SKY,      # APP_CLOUDS + SKY_SPHERE (src/app_clouds.h:8,14-19,154-162)
        "APP_VINYL_GPU": APP_VINYL_GPU,        # APP_VINYL with the 180 march steps of its GLSL / HLSL builds (src/app_vinyl.h:411-416)
        "APP_PLANET_ATMOSPHERE": APP_PLANET_ATMOSPHERE}   # config 5's composite: APP_PLANET with APP_ATMOSPHERE's sky as background (include/sbx.h)

SBX_OK, SBX_ERR_ARG, SBX_ERR_UNSUPPORTED, SBX_ERR_HIP, SBX_ERR_NO_DEVICE, SBX_ERR_FAULT = 0, -1, -2, -3, -4, -5
SBX_FORMAT_RGBA32F, SBX_FORMAT_RGBA8 = 0, 1
SBX_ABI_VERSION = 2                          # include/sbx.h; load_library() refuses a libsbx.so built from another header


class SbxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libsbx error %d: %s" % (code, msg))
        self.code = code


class Uniforms(ctypes.Structure):           # sbx_uniforms (cbuffer b0)
    _fields_ = [("u_res", ctypes.c_float * 2), ("u_mouse", ctypes.c_float * 2), ("u_time", ctypes.c_float),
                ("_pad", ctypes.c_float * 3)]


class AuxClouds(ctypes.Structure):          # sbx_aux_clouds (cbuffer b1, APP_CLOUDS)
    _fields_ = [("wind_dir", ctypes.c_float * 3), ("_pad0", ctypes.c_float),
                ("sun_dir", ctypes.c_float * 3), ("_pad1", ctypes.c_float),
                ("sun_color", ctypes.c_float * 3), ("_pad2", ctypes.c_float),
                ("sun_power", ctypes.c_float), ("cld_march_steps", ctypes.c_int32),
                ("illum_march_steps", ctypes.c_int32), ("sigma_scattering", ctypes.c_float),
                ("cld_coverage", ctypes.c_float), ("cld_thick", ctypes.c_float),
                ("atm_radius", ctypes.c_float), ("atm_ground_y", ctypes.c_float)]


class AuxCloudsUe4(ctypes.Structure):       # sbx_aux_clouds_ue4
    _fields_ = [("coverage", ctypes.c_float), ("thickness", ctypes.c_float), ("absorbtion", ctypes.c_float),
                ("fuzziness", ctypes.c_float), ("sun_dir", ctypes.c_float * 3), ("_pad1", ctypes.c_float),
                ("wind_dir", ctypes.c_float * 3), ("use_dirs", ctypes.c_int32)]


class AuxSdfAo(ctypes.Structure):           # sbx_aux_sdf_ao (cbuffer b1, APP_SDF_AO)
    _fields_ = [("fog_density", ctypes.c_float), ("fog_falloff", ctypes.c_float), ("_pad", ctypes.c_float * 2)]


class Stats(ctypes.Structure):              # sbx_stats
    _fields_ = [("render_launches", ctypes.c_uint64), ("main_image_hits", ctypes.c_uint64),
                ("main_image_frames", ctypes.c_uint64), ("main_image_points", ctypes.c_uint64),
                ("_reserved", ctypes.c_uint64 * 4)]


class SharedHandle(ctypes.Structure):       # sbx_shared_handle
    _fields_ = [("opaque", ctypes.c_ubyte * 192)]


def app_id(app):
    """Accepts an int, 'APP_CLOUDS', 'clouds', ..."""
    if isinstance(app, int):
        return app
    key = str(app).upper()
    if not key.startswith("APP_"):
        key = "APP_" + key
    if key not in APPS:
        raise ValueError("unknown app %r" % (app,))
    return APPS[key]


def load_library(path=None):
    """dlopen libsbx.so and declare the prototypes of include/sbx.h.  Raises if it is not built."""
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise ImportError("libsbx.so is not built (%s). Run `python -m shaderbox_amd.build`; "
                          "there is no fallback path." % path)
    # Load order matters: PyTorch brings its own copy of the HIP runtime.  If libsbx.so (linked against the system
    # libamdhip64) is loaded first and torch afterwards, the process ends up with two runtimes and device enumeration
    # fails (observed: sbx_create -> SBX_ERR_NO_DEVICE on a GPU box).  torch is this package's device-memory/stream
    # provider anyway, so import it before the dlopen whenever it is installed.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(path)
    vp, ci, fp = ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p
    try:
        lib.sbx_abi_version.argtypes = []
        got = int(lib.sbx_abi_version())
    except AttributeError:
        got = 1                              # (libraries before ABI 2 do not export the function)
    if got != SBX_ABI_VERSION:
        raise ImportError("%s has ABI %d, this package binds ABI %d (include/sbx.h): rebuild with `python -m shaderbox_amd.build`"
                          % (path, got, SBX_ABI_VERSION))
    lib.sbx_aux_clouds_defaults.argtypes = [ctypes.POINTER(AuxClouds)]
    lib.sbx_aux_clouds_defaults.restype = None
    lib.sbx_aux_sdf_ao_defaults.argtypes = [ctypes.POINTER(AuxSdfAo)]
    lib.sbx_aux_sdf_ao_defaults.restype = None
    lib.sbx_aux_clouds_ue4_defaults.argtypes = [ctypes.POINTER(AuxCloudsUe4)]
    lib.sbx_aux_clouds_ue4_defaults.restype = None
    lib.sbx_create.argtypes = [ci, ctypes.POINTER(vp)]
    lib.sbx_destroy.argtypes = [vp]
    lib.sbx_destroy.restype = None
    lib.sbx_render_rows.argtypes = [vp, ci, ctypes.POINTER(Uniforms), vp, ci, ci, fp, vp]
    lib.sbx_render_rows_host.argtypes = [vp, ci, ctypes.POINTER(Uniforms), vp, ci, ci, vp, vp]
    lib.sbx_render_rank.argtypes = [vp, ci, ctypes.POINTER(Uniforms), vp, ci, ci, ci, fp, vp]
    lib.sbx_pack_unorm8.argtypes = [vp, ci, ci, fp, vp, ci, vp]
    lib.sbx_set_output_format.argtypes = [vp, ci]
    lib.sbx_main_image.argtypes = [vp, ci, ctypes.POINTER(Uniforms), vp, ctypes.POINTER(ctypes.c_float * 2),
                                   ctypes.POINTER(ctypes.c_float * 4)]
    lib.sbx_main_image_batch.argtypes = [vp, ci, ctypes.POINTER(Uniforms), vp, ctypes.c_size_t, fp, fp]
    lib.sbx_render_points.argtypes = [vp, ci, ctypes.POINTER(Uniforms), vp, ctypes.c_size_t, fp, fp, vp]
    lib.sbx_render_rank_rows.argtypes = [vp, ci, ctypes.POINTER(Uniforms), vp, ci, ci, ci, ci, ci, fp, vp]
    lib.sbx_rank_rows.argtypes = [ci, ci, ci, ci]
    lib.sbx_rank_rows_max.argtypes = [ci, ci, ci]
    lib.sbx_assemble.argtypes = [vp, ci, ci, ci, ci, fp, fp, vp]
    lib.sbx_split_rank_rows.argtypes = [ci, ci, ci, ci, ci, ci]
    lib.sbx_split_rows_max.argtypes = [ci, ci, ci, ci, ci]
    lib.sbx_render_split.argtypes = [vp, ci, ctypes.POINTER(Uniforms), vp, ci, ci, ci, ci, ci, ci, ci, fp, vp]
    lib.sbx_assemble_split.argtypes = [vp, ci, ci, ci, ci, ci, ci, fp, fp, vp]
    lib.sbx_render_split_rgb.argtypes = [vp, ci, ctypes.POINTER(Uniforms), vp, ci, ci, ci, ci, ci, ci, ci, fp, vp]
    lib.sbx_assemble_peers.argtypes = [vp, ci, ci, ci, ci, ci, ci, ci, fp, fp, vp]
    lib.sbx_span_table.argtypes = [ci, ctypes.POINTER(Uniforms), vp, ci, ci, ci, ci, vp, vp, vp]
    lib.sbx_render_span_peer.argtypes = [vp, ci, ctypes.POINTER(Uniforms), vp, ci, ci, ci, ci, ci, ci, ci, fp, vp]
    lib.sbx_render_span_root.argtypes = [vp, ci, ctypes.POINTER(Uniforms), vp, ci, ci, ci, ci, fp, vp]
    lib.sbx_assemble_spans.argtypes = [vp, ci, ctypes.POINTER(Uniforms), vp, ci, ci, ci, ci, fp, ctypes.c_int64, fp, vp]
    lib.sbx_fault_status.argtypes = [vp]
    lib.sbx_clear_fault.argtypes = [vp]
    lib.sbx_debug_raise_fault.argtypes = [vp, vp]
    lib.sbx_set_timing.argtypes = [vp, ci]
    lib.sbx_set_variant.argtypes = [vp, ci]
    lib.sbx_last_kernel_ms.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]
    lib.sbx_math_eval.argtypes = [vp, ctypes.c_char_p, fp, fp, fp, ctypes.c_size_t, vp]
    lib.sbx_noise_eval.argtypes = [vp, ctypes.c_char_p, fp, fp, fp, ctypes.c_size_t, vp]
    lib.sbx_worley_volume.argtypes = [vp, ci, fp, vp]
    lib.sbx_render_split_in_place.argtypes = [vp, ci, ctypes.POINTER(Uniforms), vp, ci, ci, ci, ci, ci, fp, vp]
    lib.sbx_multi_create.argtypes = [ci, ctypes.POINTER(ci), ctypes.POINTER(vp)]
    lib.sbx_multi_destroy.argtypes = [vp]
    lib.sbx_multi_destroy.restype = None
    lib.sbx_multi_ranks.argtypes = [vp]
    lib.sbx_multi_uses_rccl.argtypes = [vp]
    lib.sbx_multi_set_split.argtypes = [vp, ci, ci, ci]
    lib.sbx_multi_set_variant.argtypes = [vp, ci]
    lib.sbx_multi_set_output_format.argtypes = [vp, ci]
    lib.sbx_multi_set_exchange.argtypes = [vp, ci]
    lib.sbx_multi_create_error.argtypes = []
    lib.sbx_multi_create_error.restype = ctypes.c_char_p
    lib.sbx_multi_set_noise_volumes.argtypes = [vp, ci, fp, ci, fp]
    lib.sbx_multi_render.argtypes = [vp, ci, ctypes.POINTER(Uniforms), vp, fp, vp]
    lib.sbx_multi_rccl_selftest.argtypes = [ci, ctypes.POINTER(ci)]
    lib.sbx_multi_last_error.argtypes = [vp]
    lib.sbx_multi_last_error.restype = ctypes.c_char_p
    lib.sbx_set_noise_volumes.argtypes = [vp, ci, fp, ci, fp, vp]
    lib.sbx_tex3d_eval.argtypes = [vp, ci, fp, fp, fp, ctypes.c_size_t, vp]
    lib.sbx_last_error.argtypes = [vp]
    lib.sbx_last_error.restype = ctypes.c_char_p
    lib.sbx_version.restype = ctypes.c_char_p
    lib.sbx_set_precision.argtypes = [vp, ci]
    lib.sbx_get_stats.argtypes = [vp, ctypes.POINTER(Stats)]
    lib.sbx_reset_stats.argtypes = [vp]
    # the store exchange (include/sbx.h sbx_shared_*)
    lib.sbx_shared_create.argtypes = [vp, ctypes.c_size_t, ci, ctypes.POINTER(vp)]
    lib.sbx_shared_export.argtypes = [vp, ctypes.POINTER(SharedHandle)]
    lib.sbx_shared_open.argtypes = [vp, ctypes.POINTER(SharedHandle), ctypes.POINTER(vp)]
    lib.sbx_shared_close.argtypes = [vp]
    lib.sbx_shared_close.restype = None
    lib.sbx_shared_frame.argtypes = [vp]
    lib.sbx_shared_frame.restype = vp
    lib.sbx_debug_tile_order.argtypes = [vp, ci, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_uint),
                                         ctypes.c_size_t]
    lib.sbx_shared_bytes.argtypes = [vp]
    lib.sbx_shared_bytes.restype = ctypes.c_size_t
    lib.sbx_shared_frame_begin.argtypes = [vp, ci, vp]
    lib.sbx_shared_frame_end.argtypes = [vp, ci, vp]
    lib.sbx_render_split_in_place_rgb.argtypes = [vp, ci, ctypes.POINTER(Uniforms), vp, ci, ci, ci, ci, ci, fp, vp]
    lib.sbx_render_span_peer_in_place.argtypes = [vp, ci, ctypes.POINTER(Uniforms), vp, ci, ci, ci, ci, ci, ci, fp, vp]
    # include/sbx_test.h (test hooks and the landing model of the scaling tools)
    lib.sbx_shared_set_timeout_ms.argtypes = [vp, ci]
    lib.sbx_model_landing.argtypes = [vp, vp, vp, ctypes.c_size_t, ci, ctypes.c_float, vp]
    return lib


def clouds_defaults(lib=None):
    lib = lib or load_library()
    a = AuxClouds()
    lib.sbx_aux_clouds_defaults(ctypes.byref(a))
    return a


def sdf_ao_defaults(lib=None):
    lib = lib or load_library()
    a = AuxSdfAo()
    lib.sbx_aux_sdf_ao_defaults(ctypes.byref(a))
    return a


def uniforms(width, height, time, mouse=(0.0, 0.0)):
    u = Uniforms()
    u.u_res[0], u.u_res[1] = float(width), float(height)
    u.u_mouse[0], u.u_mouse[1] = float(mouse[0]), float(mouse[1])
    u.u_time = float(time)
    return u


def span_table(app, width, height, time, block_rows, nranks, root_rounds=1, rounds=1, mouse=(0.0, 0.0), aux=None, lib=None):
    """sbx_span_table (host only, no GPU needed): (table, rank_pixels, max_width) of the span exchange — table int32
    [nblocks, 4] = {x0, x1, offset in the owner's packed slab (pixels), owner} per global row-block; rank_pixels[r] = pixels of
    rank r's packed slab; max_width = the widest span among the peers' blocks."""
    import numpy as np
    lib = lib or load_library()
    u = uniforms(width, height, time, mouse)
    nb = shard.num_blocks(int(height), int(block_rows))
    table = np.zeros((nb, 4), dtype=np.int32)
    pix = np.zeros(int(nranks), dtype=np.int64)
    mw = ctypes.c_int32(0)
    auxp = ctypes.cast(ctypes.byref(aux), ctypes.c_void_p) if aux is not None else None
    rc = lib.sbx_span_table(app_id(app), ctypes.byref(u), auxp, int(block_rows), int(nranks), int(root_rounds), int(rounds),
                            table.ctypes.data_as(ctypes.c_void_p), pix.ctypes.data_as(ctypes.c_void_p),
                            ctypes.cast(ctypes.byref(mw), ctypes.c_void_p))
    if rc < 0:
        raise SbxError(rc, "sbx_span_table: bad arguments")
    return table, pix, int(mw.value)


class _RawDeviceArray:
    """A device pointer as an object torch.as_tensor understands (__cuda_array_interface__, no copy)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(int(v) for v in shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


class SharedFrame:
    """One sbx_shared (include/sbx.h, the store exchange): a frame in the OWNER's device memory that the other ranks of a split
    map and render into in place.  The owner gets it from Renderer.shared_create and may view it as a tensor; a peer gets it from
    Renderer.shared_open(handle bytes) and only ever passes its address to the render calls (the memory is another device's, or
    another process's)."""

    def __init__(self, renderer, handle, owner, nbytes):
        self.r, self.h, self.owner, self.nbytes = renderer, handle, bool(owner), int(nbytes)

    @property
    def ptr(self):
        return int(self.r.lib.sbx_shared_frame(self.h) or 0)

    def export(self):
        """the opaque handle as bytes, to be sent to the other ranks by any means (only the owner exports)"""
        hd = SharedHandle()
        self.r._check(self.r.lib.sbx_shared_export(self.h, ctypes.byref(hd)))
        return bytes(hd.opaque)

    def tensor(self, shape, dtype=None):
        """the owner's view of the frame as a device tensor (no copy)"""
        torch = self.r.torch
        dtype = dtype or self.r.pixel_dtype
        assert self.owner, "only the owner views the frame as a tensor; peers hold an address in another device / process"
        n = 1
        for v in shape:
            n *= int(v)
        assert n * (1 if dtype == torch.uint8 else 4) <= self.nbytes
        t = torch.as_tensor(_RawDeviceArray(self.ptr, shape, "|u1" if dtype == torch.uint8 else "<f4"), device=self.r.tdev)
        assert t.data_ptr() == self.ptr, "torch copied the shared frame instead of viewing it"
        t._sbx_keepalive = self
        return t

    def begin(self, rank):
        """owner (rank 0): the frame may be overwritten from here on in stream order; peer: wait for that"""
        self.r._check(self.r.lib.sbx_shared_frame_begin(self.h, int(rank), self.r._stream()))

    def end(self, rank):
        """peer: my rows of this frame are in place; owner: the stream continues when every peer's are"""
        self.r._check(self.r.lib.sbx_shared_frame_end(self.h, int(rank), self.r._stream()))

    def set_timeout_ms(self, ms):
        self.r._check(self.r.lib.sbx_shared_set_timeout_ms(self.h, int(ms)))

    def close(self):
        if self.h:
            self.r.lib.sbx_shared_close(self.h)
            self.h = None

    def __del__(self):
        try:
            if self.r.ctx:
                self.close()
        except Exception:
            pass


class Renderer:
    """One sbx_ctx on one GPU.  Framebuffers are float32 torch tensors [rows, W, 4] on that GPU,
    row 0 = bottom row of the strip (reference convention, src/main.h:40-43) — or, after set_output_format("rgba8"), uint8
    tensors [rows, W, 4] (one R8G8B8A8_UNORM word per pixel, include/sbx.h), slabs included."""

    def __init__(self, device=0):
        import torch
        if not torch.cuda.is_available():
            raise SbxError(SBX_ERR_NO_DEVICE, "no GPU visible; shaderbox_amd has no CPU fallback")
        self.torch = torch
        self.lib = load_library()
        self.device = int(device)
        h = ctypes.c_void_p()
        rc = self.lib.sbx_create(self.device, ctypes.byref(h))
        if rc != SBX_OK:
            raise SbxError(rc, "sbx_create failed (need a gfx950 device; there is no fallback)")
        self.ctx = h
        self.tdev = torch.device("cuda", self.device)
        self.pixel_dtype = torch.float32

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.sbx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers ------------------------------------------------------------------------------
    def _check(self, rc):
        if rc != SBX_OK:
            raise SbxError(rc, (self.lib.sbx_last_error(self.ctx) or b"").decode())

    def _stream(self):
        return ctypes.c_void_p(self.torch.cuda.current_stream(self.tdev).cuda_stream)

    @staticmethod
    def uniforms(width, height, time, mouse=(0.0, 0.0)):
        u = Uniforms()
        u.u_res[0], u.u_res[1] = float(width), float(height)
        u.u_mouse[0], u.u_mouse[1] = float(mouse[0]), float(mouse[1])
        u.u_time = float(time)
        return u

    @staticmethod
    def _auxp(aux):
        return ctypes.cast(ctypes.byref(aux), ctypes.c_void_p) if aux is not None else None

    def set_output_format(self, fmt):
        """'rgba32f' (default) or 'rgba8': what the frame-granular calls write per pixel (sbx_set_output_format).  With 'rgba8'
        every frame / strip / slab buffer is a uint8 tensor with 4 bytes per pixel; points and main_image stay float."""
        code = {"rgba32f": SBX_FORMAT_RGBA32F, "rgba8": SBX_FORMAT_RGBA8}[fmt]
        self._check(self.lib.sbx_set_output_format(self.ctx, code))
        self.pixel_dtype = self.torch.uint8 if code == SBX_FORMAT_RGBA8 else self.torch.float32

    def set_precision(self, tier):
        """'exact' (default: bit-identical to the oracle) or '1e-4' (include/sbx.h SBX_PRECISION_1E4: APP_ATMOSPHERE with the hardware's
        binary32 exp2, within 1e-4 per channel of the exact frame; every other app ignores it)"""
        self._check(self.lib.sbx_set_precision(self.ctx, {"exact": 0, "1e-4": 1}[tier]))

    @property
    def rgba8(self):
        return self.pixel_dtype == self.torch.uint8

    def empty(self, shape, zero=False):
        """device buffer of the current pixel type (used by distributed.FramePlan)"""
        f = self.torch.zeros if zero else self.torch.empty
        return f(tuple(shape), dtype=self.pixel_dtype, device=self.tdev)

    def _is_pixels(self, t):
        return t.is_cuda and t.dtype == self.pixel_dtype and t.is_contiguous()

    def _buffer(self, rows, width, out):
        if out is None:
            return self.torch.empty((rows, width, 4), dtype=self.pixel_dtype, device=self.tdev)
        assert self._is_pixels(out)
        assert out.numel() >= rows * width * 4
        return out

    # -- the hot path ---------------------------------------------------------------------------
    def render(self, app, width, height, time, mouse=(0.0, 0.0), aux=None, rows=None, out=None):
        """Render rows [y0, y1) (default: the whole frame) of `app`; asynchronous on the current stream."""
        y0, y1 = (0, int(height)) if rows is None else (int(rows[0]), int(rows[1]))
        u = self.uniforms(width, height, time, mouse)
        buf = self._buffer(max(y1 - y0, 0), int(width), out)
        self._check(self.lib.sbx_render_rows(self.ctx, app_id(app), ctypes.byref(u), self._auxp(aux), y0, y1,
                                             ctypes.c_void_p(buf.data_ptr()), self._stream()))
        return buf

    def render_to_host(self, app, width, height, time, out, mouse=(0.0, 0.0), aux=None, rows=None):
        """sbx_render_rows_host: rows [y0, y1) (default: the whole frame) into HOST memory `out` — a C-contiguous numpy array or CPU
        torch tensor (pinned or not) of rows x width x 4 float32 (uint8 with the RGBA8 format); returns `out` when the pixels are
        there (strips are copied out while the next ones render)."""
        y0, y1 = (0, int(height)) if rows is None else (int(rows[0]), int(rows[1]))
        u = self.uniforms(width, height, time, mouse)
        n = max(y1 - y0, 0) * int(width) * 4
        if hasattr(out, "data_ptr"):
            assert not out.is_cuda and out.is_contiguous() and out.numel() >= n and out.dtype == (self.torch.uint8 if self.rgba8 else self.torch.float32)
            ptr = out.data_ptr()
        else:
            import numpy as np
            assert out.flags["C_CONTIGUOUS"] and out.size >= n and out.dtype == (np.uint8 if self.rgba8 else np.float32)
            ptr = out.ctypes.data
        self._check(self.lib.sbx_render_rows_host(self.ctx, app_id(app), ctypes.byref(u), self._auxp(aux), y0, y1,
                                                  ctypes.c_void_p(ptr), self._stream()))
        return out

    def render_rank(self, app, width, height, time, block_rows, rank, nranks, mouse=(0.0, 0.0), aux=None, out=None,
                    root_rounds=1, rounds=1):
        """Render the cyclic row-blocks of `rank` (optionally with root relief, shard.py); `out` has
        shard.rank_rows_max() rows (tail rows unused)."""
        u = self.uniforms(width, height, time, mouse)
        buf = self._buffer(shard.rank_rows_max(int(height), block_rows, nranks, root_rounds, rounds), int(width), out)
        self._check(self.lib.sbx_render_split(self.ctx, app_id(app), ctypes.byref(u), self._auxp(aux), block_rows,
                                              rank, nranks, root_rounds, rounds, 0, 0x7fffffff,
                                              ctypes.c_void_p(buf.data_ptr()), self._stream()))
        return buf

    def pack_unorm8(self, frame, flip_y=True):
        """float RGBA rows [rows, W, 4] (device) -> uint8 [rows, W, 4], the R8G8B8A8_UNORM back-buffer write of hlsltoy
        (Direct3D float -> UNORM rule); flip_y puts the top row first."""
        assert frame.is_cuda and frame.dtype == self.torch.float32 and frame.is_contiguous() and frame.shape[-1] == 4
        rows, width = int(frame.shape[0]), int(frame.shape[1])
        out = self.torch.empty((rows, width, 4), dtype=self.torch.uint8, device=self.tdev)
        self._check(self.lib.sbx_pack_unorm8(self.ctx, width, rows, ctypes.c_void_p(frame.data_ptr()),
                                             ctypes.c_void_p(out.data_ptr()), 1 if flip_y else 0, self._stream()))
        return out

    def render_points(self, app, width, height, time, frag, mouse=(0.0, 0.0), aux=None, out=None):
        """mainImage at arbitrary fragCoords: `frag` is a float32 device tensor [n, 2]; returns [n, 4] (device).  u_res =
        (width, height) need not be whole numbers here.  Asynchronous on the current stream."""
        u = self.uniforms(width, height, time, mouse)
        frag = frag.to(self.tdev, self.torch.float32).contiguous().view(-1, 2)
        n = int(frag.shape[0])
        if out is None:
            out = self.torch.empty((n, 4), dtype=self.torch.float32, device=self.tdev)
        assert out.is_cuda and out.dtype == self.torch.float32 and out.is_contiguous() and out.numel() >= 4 * n
        self._check(self.lib.sbx_render_points(self.ctx, app_id(app), ctypes.byref(u), self._auxp(aux), n,
                                               ctypes.c_void_p(frag.data_ptr()), ctypes.c_void_p(out.data_ptr()), self._stream()))
        return out

    def main_image_batch(self, app, width, height, time, frag_coords, mouse=(0.0, 0.0), aux=None):
        """sbx_main_image_batch: host arrays in, host arrays out — numpy float32 [n, 2] -> [n, 4]; synchronous."""
        import numpy as np
        u = self.uniforms(width, height, time, mouse)
        fc = np.ascontiguousarray(frag_coords, dtype=np.float32).reshape(-1, 2)
        out = np.zeros((fc.shape[0], 4), dtype=np.float32)
        self._check(self.lib.sbx_main_image_batch(self.ctx, app_id(app), ctypes.byref(u), self._auxp(aux), fc.shape[0],
                                                  fc.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def main_image(self, app, width, height, time, frag_coord, mouse=(0.0, 0.0), aux=None):
        """void mainImage(out vec4 fragColor, in vec2 fragCoord) (src/main.h:6-9) for hosts that loop over the
        pixels themselves: returns the RGBA tuple of mainImage at frag_coord.  For the centre of a pixel of the frame the
        first call renders the whole frame on the GPU and later calls read from the host copy; any other coordinate
        (off-centre, outside the frame) is evaluated exactly, by a one-point launch."""
        u = self.uniforms(width, height, time, mouse)
        fc = (ctypes.c_float * 2)(float(frag_coord[0]), float(frag_coord[1]))
        out = (ctypes.c_float * 4)()
        self._check(self.lib.sbx_main_image(self.ctx, app_id(app), ctypes.byref(u), self._auxp(aux), ctypes.byref(fc),
                                            ctypes.byref(out)))
        return tuple(out)

    def render_rank_rows(self, app, width, height, time, block_rows, rank, nranks, r0, r1, slab, mouse=(0.0, 0.0),
                         aux=None, root_rounds=1, rounds=1):
        """Render slab rows [r0, r1) of `rank` into slab[r0:r1] (pipelined multi-GPU frames).  A slab whose last dimension
        is 3 is written without alpha (sbx_render_split_rgb: what crosses xGMI in the direct exchange)."""
        u = self.uniforms(width, height, time, mouse)
        if r1 <= r0:
            return slab
        assert self._is_pixels(slab) and slab.shape[-1] in ((4,) if self.rgba8 else (3, 4))
        assert slab.shape[1] == int(width) and slab.shape[0] >= r1
        view = slab[r0:r1]
        fn = self.lib.sbx_render_split_rgb if slab.shape[-1] == 3 else self.lib.sbx_render_split
        self._check(fn(self.ctx, app_id(app), ctypes.byref(u), self._auxp(aux), block_rows, rank, nranks, root_rounds, rounds,
                       int(r0), int(r1), ctypes.c_void_p(view.data_ptr()), self._stream()))
        return slab

    def render_rank_in_place(self, app, width, height, time, block_rows, rank, nranks, frame, mouse=(0.0, 0.0), aux=None,
                             root_rounds=1, rounds=1, channels=4):
        """The rows of `rank` at their global positions of the full-size `frame` [H, W, 4]; other rows are not touched.
        `frame` is a tensor of this device or a SharedFrame (the store exchange: the owner's frame mapped by a peer).
        channels = 3: only R, G, B of every float4 pixel are written (sbx_render_split_in_place_rgb; the alpha is in the frame)."""
        u = self.uniforms(width, height, time, mouse)
        if isinstance(frame, SharedFrame):
            assert frame.nbytes >= int(height) * int(width) * (4 if self.rgba8 else 16)
            ptr = frame.ptr
        else:
            assert self._is_pixels(frame)
            assert tuple(frame.shape) == (int(height), int(width), 4)
            ptr = frame.data_ptr()
        fn = self.lib.sbx_render_split_in_place_rgb if int(channels) == 3 else self.lib.sbx_render_split_in_place
        self._check(fn(self.ctx, app_id(app), ctypes.byref(u), self._auxp(aux), block_rows, rank, nranks, root_rounds, rounds,
                       ctypes.c_void_p(ptr), self._stream()))
        return frame

    def render_span_peer_in_place(self, app, width, height, time, block_rows, rank, nranks, frame, mouse=(0.0, 0.0), aux=None,
                                  root_rounds=1, rounds=1, channels=4):
        """The spans of peer `rank`'s row-blocks at their place in the owner's full-size frame (a SharedFrame, or a tensor of this
        device): sbx_render_span_peer_in_place, the store exchange with spans."""
        u = self.uniforms(width, height, time, mouse)
        ptr = frame.ptr if isinstance(frame, SharedFrame) else frame.data_ptr()
        self._check(self.lib.sbx_render_span_peer_in_place(self.ctx, app_id(app), ctypes.byref(u), self._auxp(aux), block_rows, rank, nranks,
                                                           root_rounds, rounds, 4 if self.rgba8 else int(channels), ctypes.c_void_p(ptr),
                                                           self._stream()))
        return frame

    # -- the store exchange (include/sbx.h sbx_shared_*) --------------------------------------------------------
    def shared_create(self, nbytes, nranks):
        """a frame of `nbytes` bytes on this device that `nranks` ranks render into (this renderer is its owner, rank 0)"""
        h = ctypes.c_void_p()
        self._check(self.lib.sbx_shared_create(self.ctx, int(nbytes), int(nranks), ctypes.byref(h)))
        return SharedFrame(self, h, True, nbytes)

    def shared_open(self, handle_bytes):
        """map the frame another rank exported (SharedFrame.export()) for rendering from this device"""
        hd = SharedHandle()
        assert len(handle_bytes) == ctypes.sizeof(hd)
        ctypes.memmove(ctypes.byref(hd), bytes(handle_bytes), ctypes.sizeof(hd))
        h = ctypes.c_void_p()
        self._check(self.lib.sbx_shared_open(self.ctx, ctypes.byref(hd), ctypes.byref(h)))
        return SharedFrame(self, h, False, int(self.lib.sbx_shared_bytes(h)))

    def model_landing(self, src, dst, nbytes, workgroups, duration_us):
        """include/sbx_test.h sbx_model_landing: the scaling tools' stand-in for RCCL's receive kernels on the frame's owner"""
        self._check(self.lib.sbx_model_landing(self.ctx, ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(dst.data_ptr()), int(nbytes),
                                               int(workgroups), float(duration_us), self._stream()))

    def stats(self):
        st = Stats()
        self._check(self.lib.sbx_get_stats(self.ctx, ctypes.byref(st)))
        return {k: int(getattr(st, k)) for k in ("render_launches", "main_image_hits", "main_image_frames", "main_image_points")}

    def reset_stats(self):
        self._check(self.lib.sbx_reset_stats(self.ctx))

    # -- the span exchange: only the expensive part of each row-block is sharded (include/sbx.h) ------------------
    def span_table(self, app, width, height, time, block_rows, nranks, root_rounds=1, rounds=1, mouse=(0.0, 0.0), aux=None):
        return span_table(app, width, height, time, block_rows, nranks, root_rounds, rounds, mouse, aux, lib=self.lib)

    def render_span_peer(self, app, width, height, time, block_rows, rank, nranks, r0, r1, slab, mouse=(0.0, 0.0), aux=None,
                         root_rounds=1, rounds=1):
        """Slab rows [r0, r1) (whole blocks) of peer `rank`, spans only, packed 3 floats per pixel into `slab` (a flat float32
        device buffer of >= 3 * rank_pixels[rank] floats: the table's offsets are absolute within it; 'rgba8': 4 bytes per
        pixel of a flat uint8 buffer)."""
        u = self.uniforms(width, height, time, mouse)
        if isinstance(slab, tuple):                       # (SharedFrame, byte offset): the rank's slab inside the owner's mapped landing area
            ptr = slab[0].ptr + int(slab[1])
        else:
            assert self._is_pixels(slab)
            ptr = slab.data_ptr()
        self._check(self.lib.sbx_render_span_peer(self.ctx, app_id(app), ctypes.byref(u), self._auxp(aux), block_rows, rank, nranks,
                                                  root_rounds, rounds, int(r0), int(r1), ctypes.c_void_p(ptr), self._stream()))
        return slab

    def render_span_root(self, app, width, height, time, block_rows, nranks, frame, mouse=(0.0, 0.0), aux=None, root_rounds=1,
                         rounds=1):
        """The owner's launch of the span exchange, in place over the whole `frame` [H, W, 4]: rank 0's row-blocks in full and
        every other block outside its span."""
        u = self.uniforms(width, height, time, mouse)
        assert self._is_pixels(frame)
        assert tuple(frame.shape) == (int(height), int(width), 4)
        self._check(self.lib.sbx_render_span_root(self.ctx, app_id(app), ctypes.byref(u), self._auxp(aux), block_rows, nranks,
                                                  root_rounds, rounds, ctypes.c_void_p(frame.data_ptr()), self._stream()))
        return frame

    def assemble_spans(self, app, width, height, time, block_rows, nranks, peers, stride_pixels, frame, mouse=(0.0, 0.0), aux=None,
                       root_rounds=1, rounds=1):
        """Scatter the peers' packed span slabs (`peers`: flat float32, slab of rank r at (r - 1) * stride_pixels * 3) into
        `frame`, alpha = 1; everything else in the frame is left as sbx_render_span_root wrote it."""
        u = self.uniforms(width, height, time, mouse)
        assert self._is_pixels(peers) and self._is_pixels(frame)
        assert peers.numel() >= (int(nranks) - 1) * int(stride_pixels) * (4 if self.rgba8 else 3)
        self._check(self.lib.sbx_assemble_spans(self.ctx, app_id(app), ctypes.byref(u), self._auxp(aux), block_rows, nranks,
                                                root_rounds, rounds, ctypes.c_void_p(peers.data_ptr()), int(stride_pixels),
                                                ctypes.c_void_p(frame.data_ptr()), self._stream()))
        return frame

    def assemble_peers(self, peers, width, height, block_rows, nranks, frame, root_rounds=1, rounds=1):
        """Root side of the direct exchange: scatter the slabs of ranks 1 .. nranks-1 (`peers` [nranks-1, rows_max, W, 3|4])
        to their rows of `frame` [H, W, 4] (alpha = 1 for 3-channel slabs); rank 0's rows are left as rendered in place."""
        if int(nranks) == 1:
            return frame
        ch = int(peers.shape[-1])
        need = (int(nranks) - 1) * shard.rank_rows_max(int(height), int(block_rows), int(nranks), root_rounds, rounds) * int(width) * ch
        if not (self._is_pixels(peers) and ch in ((4,) if self.rgba8 else (3, 4)) and peers.numel() >= need):
            raise ValueError("peers must be a contiguous device tensor of >= (nranks-1) * rows_max * W * C = %d elements of the "
                             "renderer's pixel type" % need)
        assert self._is_pixels(frame)
        assert tuple(frame.shape) == (int(height), int(width), 4)
        self._check(self.lib.sbx_assemble_peers(self.ctx, int(width), int(height), block_rows, nranks, root_rounds, rounds, ch,
                                                ctypes.c_void_p(peers.data_ptr()), ctypes.c_void_p(frame.data_ptr()),
                                                self._stream()))
        return frame

    def assemble(self, gathered, width, height, block_rows, nranks, out=None, root_rounds=1, rounds=1):
        """Root side: scatter the rank-major gathered slabs to their global rows -> [H, W, 4]."""
        frame = self._buffer(int(height), int(width), out)
        need = int(nranks) * shard.rank_rows_max(int(height), int(block_rows), int(nranks), root_rounds, rounds) * int(width) * 4
        if not (self._is_pixels(gathered) and gathered.numel() >= need):
            raise ValueError("gathered must be a contiguous device tensor of >= nranks * rows_max * W * 4 = %d elements of the "
                             "renderer's pixel type" % need)
        self._check(self.lib.sbx_assemble_split(self.ctx, int(width), int(height), block_rows, nranks, root_rounds, rounds,
                                                ctypes.c_void_p(gathered.data_ptr()), ctypes.c_void_p(frame.data_ptr()),
                                                self._stream()))
        return frame

    def fault_status(self):
        """0, or SBX_ERR_FAULT once a kernel on this device has reported an invariant violation (sticky until clear_fault)"""
        return int(self.lib.sbx_fault_status(self.ctx))

    def clear_fault(self):
        self._check(self.lib.sbx_clear_fault(self.ctx))

    def set_variant(self, variant):
        """0 = default kernels, 1 = per-lane cross-check kernels (bit-identical by specification)."""
        self._check(self.lib.sbx_set_variant(self.ctx, int(variant)))

    def tile_order(self, app, capacity=1 << 20):
        """include/sbx_test.h sbx_debug_tile_order: (tables built for the app's current launch shape, launches since the last one,
        the current table as a numpy array of bx | by << 16 words or None)"""
        import numpy as np
        built, since = ctypes.c_int(), ctypes.c_int()
        buf = (ctypes.c_uint * int(capacity))()
        n = self.lib.sbx_debug_tile_order(self.ctx, app_id(app), ctypes.byref(built), ctypes.byref(since), buf, int(capacity))
        self._check(min(n, 0))
        return built.value, since.value, (np.frombuffer(buf, dtype=np.uint32, count=n).copy() if n > 0 else None)

    def set_timing(self, enabled=True):
        self._check(self.lib.sbx_set_timing(self.ctx, 1 if enabled else 0))

    def last_kernel_ms(self):
        ms = ctypes.c_float()
        self._check(self.lib.sbx_last_kernel_ms(self.ctx, ctypes.byref(ms)))
        return ms.value

    def noise(self, fn, xyz, params=(0.0, 0.0, 0.0)):
        """Library noise functions (noise_iq / hash_w / noise_w / fbm_worley_tile) over points xyz[n,3] -> [n,3]."""
        xyz = xyz.to(self.tdev, self.torch.float32).contiguous().view(-1, 3)
        par = (ctypes.c_float * 3)(*[float(v) for v in params])
        out = self.torch.empty_like(xyz)
        self._check(self.lib.sbx_noise_eval(self.ctx, fn.encode(), ctypes.c_void_p(xyz.data_ptr()),
                                            ctypes.cast(par, ctypes.c_void_p), ctypes.c_void_p(out.data_ptr()),
                                            xyz.shape[0], self._stream()))
        return out

    def worley_volume(self, size=128):
        """The ddsvolgen noise volume: float32 [size, size, size, 4] (z, y, x, rgba)."""
        out = self.torch.empty((size, size, size, 4), dtype=self.torch.float32, device=self.tdev)
        self._check(self.lib.sbx_worley_volume(self.ctx, int(size), ctypes.c_void_p(out.data_ptr()), self._stream()))
        return out

    def set_noise_volumes(self, shape_rgba, detail_rgba):
        """Bind the two 3-D noise textures of APP_CLOUDS' USE_NOISE_TEX build (t1 = shape, t2 = Worley detail): float32
        device tensors [size, size, size, 4] as worley_volume() / ddsvolgen produce them.  The library keeps its own copy
        of the .r channel; render with app 'clouds_tex' afterwards (same stream)."""
        for v in (shape_rgba, detail_rgba):
            assert v.is_cuda and v.dtype == self.torch.float32 and v.is_contiguous() and v.dim() == 4 and v.shape[3] == 4
            assert v.shape[0] == v.shape[1] == v.shape[2]
        self._check(self.lib.sbx_set_noise_volumes(self.ctx, int(shape_rgba.shape[0]), ctypes.c_void_p(shape_rgba.data_ptr()),
                                                   int(detail_rgba.shape[0]), ctypes.c_void_p(detail_rgba.data_ptr()),
                                                   self._stream()))

    def tex3d(self, rgba, xyz):
        """SampleLevel(linear, wrap, 0).r of an RGBA32F device volume at points xyz[n, 3] (the texture-filter spec)."""
        assert rgba.is_cuda and rgba.dtype == self.torch.float32 and rgba.is_contiguous() and rgba.shape[3] == 4
        xyz = xyz.to(self.tdev, self.torch.float32).contiguous().view(-1, 3)
        out = self.torch.empty((xyz.shape[0],), dtype=self.torch.float32, device=self.tdev)
        self._check(self.lib.sbx_tex3d_eval(self.ctx, int(rgba.shape[0]), ctypes.c_void_p(rgba.data_ptr()),
                                            ctypes.c_void_p(xyz.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                            xyz.shape[0], self._stream()))
        return out

    def math(self, fn, a, b=None):
        """Evaluate the device math spec elementwise (parity tests)."""
        a = a.to(self.tdev, self.torch.float32).contiguous()
        bp = None
        if b is not None:
            b = b.to(self.tdev, self.torch.float32).contiguous()
            bp = ctypes.c_void_p(b.data_ptr())
        out = self.torch.empty_like(a)
        self._check(self.lib.sbx_math_eval(self.ctx, fn.encode(), ctypes.c_void_p(a.data_ptr()), bp,
                                           ctypes.c_void_p(out.data_ptr()), a.numel(), self._stream()))
        return out


class MultiRenderer:
    """One sbx_multi: a single process driving `devices` (rank i on devices[i]; rank 0 owns the frame).  With distinct
    devices the row-blocks travel over RCCL inside the library; repeated devices run the same schedule with device copies
    (N ranks on one GPU).  Frames are float32 torch tensors [H, W, 4] on devices[0]."""

    def __init__(self, devices):
        import torch
        if not torch.cuda.is_available():
            raise SbxError(SBX_ERR_NO_DEVICE, "no GPU visible; shaderbox_amd has no CPU fallback")
        self.torch = torch
        self.lib = load_library()
        self.devices = [int(d) for d in devices]
        arr = (ctypes.c_int * len(self.devices))(*self.devices)
        h = ctypes.c_void_p()
        rc = self.lib.sbx_multi_create(len(self.devices), arr, ctypes.byref(h))
        if rc != SBX_OK:
            raise SbxError(rc, "sbx_multi_create failed: " + (self.lib.sbx_multi_create_error() or b"").decode())
        self.m = h
        self.tdev = torch.device("cuda", self.devices[0])

    def close(self):
        if getattr(self, "m", None):
            self.lib.sbx_multi_destroy(self.m)
            self.m = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != SBX_OK:
            raise SbxError(rc, (self.lib.sbx_multi_last_error(self.m) or b"").decode())

    @property
    def uses_rccl(self):
        return self.lib.sbx_multi_uses_rccl(self.m) == 1

    def set_split(self, block_rows=8, root_rounds=1, rounds=1):
        self._check(self.lib.sbx_multi_set_split(self.m, int(block_rows), int(root_rounds), int(rounds)))

    def set_output_format(self, fmt):
        """'rgba32f' or 'rgba8' on every rank (sbx_multi_set_output_format): render() then returns / fills uint8 [H, W, 4]"""
        code = {"rgba32f": SBX_FORMAT_RGBA32F, "rgba8": SBX_FORMAT_RGBA8}[fmt]
        self._check(self.lib.sbx_multi_set_output_format(self.m, code))
        self.pixel_dtype = self.torch.uint8 if code == SBX_FORMAT_RGBA8 else self.torch.float32

    def set_variant(self, variant):
        self._check(self.lib.sbx_multi_set_variant(self.m, int(variant)))

    def set_exchange(self, mode):
        """'slabs' (default): one send / receive per peer of its whole 3-channel slab + one scatter kernel on rank 0;
        'blocks': one send / receive pair per row-block straight into the final rows (the round-2 form);
        'spans': the span exchange — only the expensive interval of every row-block is dealt to the peers and sent, rank 0
        renders the rest in place (include/sbx.h); 'peer_stores': every rank writes its pixels straight into rank 0's frame
        through peer access (no RCCL, no slabs, nothing for rank 0 to land or scatter; unmeasured on more than one device)."""
        self._check(self.lib.sbx_multi_set_exchange(self.m, {"slabs": 0, "blocks": 1, "spans": 2, "peer_stores": 3}[mode] if isinstance(mode, str) else int(mode)))

    def set_noise_volumes(self, shape_rgba, detail_rgba):
        self._check(self.lib.sbx_multi_set_noise_volumes(self.m, int(shape_rgba.shape[0]), ctypes.c_void_p(shape_rgba.data_ptr()),
                                                         int(detail_rgba.shape[0]), ctypes.c_void_p(detail_rgba.data_ptr())))

    def render(self, app, width, height, time, mouse=(0.0, 0.0), aux=None, out=None):
        """The whole frame over all ranks; asynchronous on the current stream of devices[0]."""
        u = Renderer.uniforms(width, height, time, mouse)
        dt = getattr(self, "pixel_dtype", self.torch.float32)
        if out is None:
            out = self.torch.empty((int(height), int(width), 4), dtype=dt, device=self.tdev)
        assert out.is_cuda and out.dtype == dt and out.is_contiguous() and out.numel() >= int(height) * int(width) * 4
        stream = ctypes.c_void_p(self.torch.cuda.current_stream(self.tdev).cuda_stream)
        self._check(self.lib.sbx_multi_render(self.m, app_id(app), ctypes.byref(u), Renderer._auxp(aux),
                                              ctypes.c_void_p(out.data_ptr()), stream))
        return out
