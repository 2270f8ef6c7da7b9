import os, sys, time
sys.path.insert(0, ".")
import torch, shaderbox_amd as sa
R = sa.Renderer(0)
res = []
for app, W, H, ts in (("egg", 1920, 1080, (0.37, 1.0, 2.0, 3.3)), ("clouds", 3840, 2160, (0.37, 2.5, 4.0))):
    out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
    for t in ts:
        for k in range(40):
            R.render(app, W, H, t, out=out)
            if k % 2: torch.cuda.synchronize()
        n = 100
        t0 = time.perf_counter()
        for k in range(n):
            R.render(app, W, H, t, out=out)
        torch.cuda.synchronize()
        res.append("%s t=%.2f %.4f" % (app, t, (time.perf_counter() - t0) * 1e3 / n))
print("SBX_TILE_ORDER=%s static frames back to back, ms: " % os.environ.get("SBX_TILE_ORDER", "1") + " | ".join(res))
