import os, sys, time
sys.path.insert(0, ".")
import torch, shaderbox_amd as sa
if os.environ.get("SBX_AB_LIB"):
    sa.LIB_PATH = os.path.join("build", "ab", "libsbx_%s.so" % os.environ["SBX_AB_LIB"])
R = sa.Renderer(0)
W, H, br, world = 3840, 2160, 8, 8
streams = [torch.cuda.Stream() for _ in range(3)]
frames = [torch.zeros((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(3)]
def run(rank, pipelined, n):
    for i in range(n):
        k = i % 3 if pipelined else 0
        with torch.cuda.stream(streams[k]):
            R.render_rank_in_place("clouds", W, H, 0.37, br, rank, world, frames[k])
for _ in range(10):
    R.render("clouds", W, H, 0.37)
torch.cuda.synchronize()
res = []
for rank in (0, 3):
    for pipelined in (False, True):
        run(rank, pipelined, 60); torch.cuda.synchronize()
        t0 = time.perf_counter(); run(rank, pipelined, 300); torch.cuda.synchronize()
        res.append("rank %d %s %.4f ms" % (rank, "3 in flight" if pipelined else "one at a time", (time.perf_counter() - t0) * 1e3 / 300))
print("SBX_TILE_ORDER=%s  eighth-frame launches (rank of 8, in place): " % os.environ.get("SBX_TILE_ORDER", "1") + " | ".join(res), R.tile_order("clouds")[:2])
