import sys, time, torch
sys.path.insert(0, "/root/repo")
import shaderbox_amd
R = shaderbox_amd.Renderer(0)
for app, w, h in (("egg", 1920, 1080), ("raytracer", 3840, 2160)):
    outs = [torch.empty((h, w, 4), dtype=torch.float32, device="cuda") for _ in range(3)]
    ss = [torch.cuda.Stream() for _ in range(3)]
    for _ in range(100): R.render(app, w, h, .37, out=outs[0])
    torch.cuda.synchronize()
    def pipe(k=300):
        for i in range(9):
            with torch.cuda.stream(ss[i % 3]): R.render(app, w, h, .37, out=outs[i % 3])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(k):
            with torch.cuda.stream(ss[i % 3]): R.render(app, w, h, .37, out=outs[i % 3])
        torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3 / k
    print(app, "python launches, 3 streams: %.4f ms/frame" % min(pipe() for _ in range(3)))
    # host cost alone: launches of an empty strip
    t0 = time.perf_counter()
    for i in range(300):
        with torch.cuda.stream(ss[i % 3]): R.render(app, w, h, .37, rows=(0, 0), out=outs[i % 3])
    print(app, "host cost of one render() call (empty strip): %.4f ms" % ((time.perf_counter() - t0) * 1e3 / 300))
    # graph: 30 frames on 3 streams forked from a capture stream
    g = torch.cuda.CUDAGraph()
    cs = torch.cuda.Stream()
    with torch.cuda.stream(cs):
        with torch.cuda.graph(g, stream=cs):
            for i in range(30):
                R.render(app, w, h, .37, out=outs[i % 3])
    torch.cuda.synchronize()
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): g.replay()
    torch.cuda.synchronize()
    print(app, "graph of 30 serial launches: %.4f ms/frame" % ((time.perf_counter() - t0) * 1e3 / 300))
