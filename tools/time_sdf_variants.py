import sys, time, torch
sys.path.insert(0, "/root/repo")
import os, shaderbox_amd
if os.environ.get("SBX_LIB"): shaderbox_amd.LIB_PATH = os.environ["SBX_LIB"]
R = shaderbox_amd.Renderer(0)
APP = sys.argv[1] if len(sys.argv) > 1 else "egg"
for (w, h) in ((1920, 1080), (3840, 2160)):
    out = [torch.empty((h, w, 4), dtype=torch.float32, device="cuda") for _ in range(3)]
    ss = [torch.cuda.Stream() for _ in range(2)]
    for _ in range(200): R.render(APP, w, h, .37, out=out[0])
    torch.cuda.synchronize()
    for v in (0, 3, 2, 0, 3):
        R.set_variant(v)
        R.set_timing(True)
        ms = []
        for _ in range(31):
            R.render(APP, w, h, .37, out=out[0]); ms.append(R.last_kernel_ms())
        ms.sort()
        R.set_timing(False)
        def pipe(k=200):
            for i in range(20):
                with torch.cuda.stream(ss[i % 2]): R.render(APP, w, h, .37, out=out[i % 2])
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(k):
                with torch.cuda.stream(ss[i % 2]): R.render(APP, w, h, .37, out=out[i % 2])
            torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3 / k
        p = min(pipe() for _ in range(3))
        print(APP + " %dx%d variant %d: one launch median %.4f min %.4f ms; 2 in flight %.4f ms/frame" % (w, h, v, ms[15], ms[0], p), flush=True)
    R.set_variant(0)
