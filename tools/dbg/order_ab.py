import os, sys, time
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/shaderbox_amd") else ".")
import torch, shaderbox_amd as sa
if os.environ.get("SBX_AB_LIB"):
    sa.LIB_PATH = os.path.join("build", "ab", "libsbx_%s.so" % os.environ["SBX_AB_LIB"])
R = sa.Renderer(0); R.set_timing(True)
W, H = 3840, 2160
out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
ref = None
for _ in range(40):
    R.render("clouds", W, H, 0.37, out=out)
torch.cuda.synchronize()
ms = []
for _ in range(20):
    R.render("clouds", W, H, 0.37, out=out); ms.append(R.last_kernel_ms())
ms.sort()
R.set_timing(False)
t0 = time.perf_counter()
for _ in range(40):
    R.render("clouds", W, H, 0.37, out=out)
torch.cuda.synchronize()
b2b = (time.perf_counter() - t0) * 1e3 / 40
streams = [torch.cuda.Stream() for _ in range(3)]
outs = [out, torch.empty_like(out), torch.empty_like(out)]
for i in range(12):
    with torch.cuda.stream(streams[i % 3]):
        R.render("clouds", W, H, 0.37, out=outs[i % 3])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(60):
    with torch.cuda.stream(streams[i % 3]):
        R.render("clouds", W, H, 0.37, out=outs[i % 3])
torch.cuda.synchronize()
pipe = (time.perf_counter() - t0) * 1e3 / 60
R.set_variant(1); whole = R.render("clouds", W, H, 0.37); R.set_variant(0)
torch.cuda.synchronize()
diff = int((out.view(torch.int32) != whole.view(torch.int32)).any(dim=-1).sum().item())
print("mode %s " % os.environ.get("SBX_TILE_ORDER_MODE", "1") + "3 in flight %.4f ms/frame | " % pipe, end="")
print("SBX_TILE_ORDER=%s: kernel ms min %.4f median %.4f | back-to-back %.4f ms/frame | pixels differing from the per-lane kernel: %d"
      % (os.environ.get("SBX_TILE_ORDER", "1"), ms[0], ms[len(ms) // 2], b2b, diff))
