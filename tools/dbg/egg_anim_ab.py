import os, sys, time
sys.path.insert(0, ".")
import torch, shaderbox_amd as sa
R = sa.Renderer(0)
EVERY = int(os.environ.get("SYNC_EVERY", "1"))
W, H = 1920, 1080
out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
res = []
for app, W, H in (("egg", 1920, 1080), ("clouds", 3840, 2160), ("vinyl", 3840, 2160)):
    out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
    for dt in (0.0, 1 / 60.0, 1 / 10.0):
        for k in range(8):
            R.render(app, W, H, 0.37 + dt * k, out=out); torch.cuda.synchronize()
        n = 240
        t0 = time.perf_counter()
        for k in range(n):
            R.render(app, W, H, 0.5 + dt * k, out=out)
            if k % EVERY == EVERY - 1:
                torch.cuda.synchronize()          # a host that looks at its frames
        torch.cuda.synchronize()
        res.append("%s dt=%.3f: %.4f ms/frame" % (app, dt, (time.perf_counter() - t0) * 1e3 / n))
print("SBX_TILE_ORDER=%s animated, one launch at a time: " % os.environ.get("SBX_TILE_ORDER", "1") + " | ".join(res))
