import sys, torch
sys.path.insert(0, ".")
import shaderbox_amd
R = shaderbox_amd.Renderer(0); R.set_timing(True)
buf = torch.empty((4320, 7680, 4), dtype=torch.float32, device="cuda")
for _ in range(10): R.render("atmosphere", 7680, 4320, .37, out=buf)
for t in (0.0, .37, 1.0, 2.0, 2.6, 3.0, 3.1, 3.14159):
    ms = []
    for _ in range(7):
        R.render("atmosphere", 7680, 4320, t, out=buf); ms.append(R.last_kernel_ms())
    ms.sort()
    nan = int(torch.isnan(buf).any(-1).sum()); black = int((buf[..., :3] == 0).all(-1).sum())
    print("t=%.5f  %.3f ms   NaN pixels %d  black %.1f %%" % (t, ms[3], nan, 100.0 * black / (4320 * 7680)))
