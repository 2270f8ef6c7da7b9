#!/usr/bin/env python3
"""How does a real link treat the store exchange's pixel stores?  (run on the GPU box)  The one link a 1-GPU box has is PCIe: the
render kernels store a whole frame IN PLACE (sbx_render_split_in_place*, rank 0 of 1) into PINNED HOST memory as 16-byte pixels,
as 12-byte stores at a 16-byte stride (channels = 3, the store exchange's float form) and as 4-byte RGBA8 pixels, beside the same
launch into HBM and a DMA copy of the same frame.  The store-bound frames (RAYTRACER 4K: 0.14 ms of kernel for 133 MB) show the link's
rate for each form; the others show how much of the transfer hides behind the kernel.  PCIe is not xGMI — this is evidence about partial
-line stores leaving the chip, not a measurement of the 8-GPU exchange.     python tools/time_link_stores.py"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import shaderbox_amd  # noqa: E402
from shaderbox_amd import app_id  # noqa: E402

dev = torch.device("cuda", 0)
R = shaderbox_amd.Renderer(0)
R8 = shaderbox_amd.Renderer(0)
R8.set_output_format("rgba8")


def med(f, n=7):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        f()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[n // 2]


for app, W, H in (("raytracer", 3840, 2160), ("egg", 1920, 1080), ("clouds", 3840, 2160), ("atmosphere", 7680, 4320), ("planet", 7680, 4320)):
    u = R.uniforms(W, H, .37)
    host = torch.zeros((H, W, 4), dtype=torch.float32).pin_memory()
    host8 = torch.zeros((H, W, 4), dtype=torch.uint8).pin_memory()
    hbm = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
    hbm8 = torch.zeros((H, W, 4), dtype=torch.uint8, device=dev)

    def go(r, fn, ptr):
        r._check(fn(r.ctx, app_id(app), ctypes.byref(u), None, 8, 0, 1, 1, 1, ctypes.c_void_p(ptr), None))
    f4, f3 = R.lib.sbx_render_split_in_place, R.lib.sbx_render_split_in_place_rgb
    for _ in range(3):
        go(R, f4, hbm.data_ptr())
    torch.cuda.synchronize()
    k = med(lambda: go(R, f4, hbm.data_ptr()))
    k3 = med(lambda: go(R, f3, hbm.data_ptr()))
    k8 = med(lambda: go(R8, R8.lib.sbx_render_split_in_place, hbm8.data_ptr()))
    p16 = med(lambda: go(R, f4, host.data_ptr()))
    ref = hbm.cpu()
    same16 = bool((host.view(torch.int32) == ref.view(torch.int32)).all())
    host.zero_()
    host[..., 3] = 1.0
    p12 = med(lambda: go(R, f3, host.data_ptr()))
    same12 = bool((host[..., :3].contiguous().view(torch.int32) == ref[..., :3].contiguous().view(torch.int32)).all())
    # the PACKED 3-float slab of the send / receive forms (sbx_render_split_rgb, rank 0 of 1: the whole frame, 12 bytes per pixel
    # with no holes): the same 12 bytes per lane, but a wave's stores cover whole lines
    hostp = torch.zeros((H, W, 3), dtype=torch.float32).pin_memory()
    fp = R.lib.sbx_render_split_rgb

    def gop(ptr):
        R._check(fp(R.ctx, app_id(app), ctypes.byref(u), None, 8, 0, 1, 1, 1, 0, H, ctypes.c_void_p(ptr), None))
    pp = med(lambda: gop(hostp.data_ptr()))
    samep = bool((hostp.view(torch.int32) == ref[..., :3].contiguous().view(torch.int32)).all())
    p4 = med(lambda: go(R8, R8.lib.sbx_render_split_in_place, host8.data_ptr()))
    same4 = bool((host8 == hbm8.cpu()).all())
    c = med(lambda: host.copy_(hbm, non_blocking=True))
    px = W * H
    print("%-10s %dx%d | into HBM: 16 B %.3f ms, 12 B %.3f, 4 B %.3f | into pinned host memory over PCIe: 16 B pixels %.3f ms (%.1f GB/s), "
          "12 B stores at a 16 B stride %.3f ms (%.1f GB/s of pixel bytes), PACKED 12 B pixels (3-float slab) %.3f ms (%.1f GB/s), RGBA8 %.3f ms (%.1f GB/s) | DMA copy of the float frame %.3f ms "
          "(%.1f GB/s) | same bits: %s %s %s"
          % (app, W, H, k, k3, k8, p16, px * 16 / p16 / 1e6, p12, px * 12 / p12 / 1e6, pp, px * 12 / pp / 1e6, p4, px * 4 / p4 / 1e6, c, px * 16 / c / 1e6,
             "%s %s %s" % (same16, same12, samep), "", same4))
    del host, host8, hbm, hbm8, hostp
    torch.cuda.empty_cache()
