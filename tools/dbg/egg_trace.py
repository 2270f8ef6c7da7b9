import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, shaderbox_amd as sa
if len(sys.argv) > 1 and sys.argv[1] != "base":
    sa.LIB_PATH = os.path.join(ROOT, "build", "ab", "libsbx_%s.so" % sys.argv[1])
r = sa.Renderer(0)
out = torch.zeros((1080, 1920, 4), dtype=torch.float32, device="cuda")
for _ in range(12):
    r.render("egg", 1920, 1080, 0.37, out=out)
    torch.cuda.synchronize()
