import sys, time, torch
sys.path.insert(0, "/root/repo")
import shaderbox_amd
for fmt in ("rgba32f", "rgba8"):
    R = shaderbox_amd.Renderer(0)
    R.set_output_format(fmt)
    for app, w, h in (("egg", 1920, 1080), ("raytracer", 3840, 2160), ("clouds", 3840, 2160), ("atmosphere", 7680, 4320), ("planet", 7680, 4320), ("sdf_ao", 3840, 2160)):
        out = [torch.empty((h, w, 4), dtype=R.pixel_dtype, device="cuda") for _ in range(3)]
        ss = [torch.cuda.Stream() for _ in range(3)]
        for _ in range(30): R.render(app, w, h, .37, out=out[0])
        torch.cuda.synchronize()
        R.set_timing(True)
        ms = []
        for _ in range(15):
            R.render(app, w, h, .37, out=out[0]); ms.append(R.last_kernel_ms())
        ms.sort()
        R.set_timing(False)
        def pipe(k=60):
            for i in range(9):
                with torch.cuda.stream(ss[i % 3]): R.render(app, w, h, .37, out=out[i % 3])
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(k):
                with torch.cuda.stream(ss[i % 3]): R.render(app, w, h, .37, out=out[i % 3])
            torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3 / k
        p = min(pipe() for _ in range(3))
        print("%-8s %-10s %dx%d: one launch median %.4f ms; 3 in flight %.4f ms/frame" % (fmt, app, w, h, ms[7], p), flush=True)
    R.close()
