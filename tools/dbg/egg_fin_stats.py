import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, shaderbox_amd as sa
sa.LIB_PATH = os.path.join(ROOT, "build", "ab", "libsbx_eggstats.so")
r = sa.Renderer(0)
out = torch.zeros((1080, 1920, 4), dtype=torch.float32, device="cuda")
for k in range(4):
    print("=== frame", k, flush=True)
    r.render("egg", 1920, 1080, 0.37, out=out)
    torch.cuda.synchronize()
