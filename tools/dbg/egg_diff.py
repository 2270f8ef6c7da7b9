import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    import numpy as np, torch, shaderbox_amd as sa
    name, variant, W, H = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    if name != "base":
        sa.LIB_PATH = os.path.join(ROOT, "build", "ab", "libsbx_%s.so" % name)
    r = sa.Renderer(0)
    r.set_variant(variant)
    out = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    a = r.render("egg", W, H, 0.37, out=out).cpu().numpy()
    b = r.render("egg", W, H, 0.37, out=out).cpu().numpy()
    print(name, variant, "run-to-run different pixels:", int((a.view(np.uint32) != b.view(np.uint32)).any(axis=2).sum()))
    np.save("/tmp/egg_%s_%d.npy" % (name, variant), a)
    sys.exit(0)
import numpy as np
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
for name, v in [("old", 0), ("base", 0), ("base", 3), ("base", 1), ("base", 2)]:
    subprocess.call([sys.executable, os.path.abspath(__file__), "--one", name, str(v), str(W), str(H)])
ref = np.load("/tmp/egg_old_0.npy")
for name, v in [("base", 0), ("base", 3), ("base", 1), ("base", 2)]:
    a = np.load("/tmp/egg_%s_%d.npy" % (name, v))
    d = (a.view(np.uint32) != ref.view(np.uint32)).any(axis=2)
    ys, xs = np.nonzero(d)
    print("%s variant %d: %d different pixels" % (name, v, d.sum()))
    if d.sum():
        print("   bbox x %d-%d y %d-%d" % (xs.min(), xs.max(), ys.min(), ys.max()))
        for k in range(0, len(ys), max(1, len(ys) // 12)):
            print("   (%4d,%4d) got %s  want %s" % (xs[k], ys[k], a[ys[k], xs[k], :3], ref[ys[k], xs[k], :3]))
        # group structure: per 64x4 group, number of wrong pixels
        g = d[:H // 4 * 4, :W // 64 * 64].reshape(H // 4, 4, W // 64, 64).sum(axis=(1, 3))
        print("   groups with wrong pixels: %d; histogram of wrong pixels per such group:" % (g > 0).sum(), np.bincount(g[g > 0])[:70])
