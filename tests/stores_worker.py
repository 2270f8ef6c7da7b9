"""One rank of the store exchange's multi-PROCESS test (tests/test_gpu_round5.py): N processes share the box's one GPU, rendezvous
over gloo, rank 0 exports its frames (hipIpcGetMemHandle), the others map them (hipIpcOpenMemHandle) and render their row-blocks
in place — FramePlan(exchange="stores") exactly as bench.py --gpus N drives it, several frames in flight.  Rank 0 compares every
frame with one launch and writes a JSON verdict.

    RANK=r WORLD_SIZE=n MASTER_ADDR=127.0.0.1 MASTER_PORT=p python tests/stores_worker.py out.json app W H channels fmt nframes [exchange]
(exchange: stores (default), span_stores or packed_stores — the last one shares the owner's LANDING AREA instead of its frame)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out, app, W, H, channels, fmt, nframes = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6], int(sys.argv[7])
    exchange = sys.argv[8] if len(sys.argv) > 8 else "stores"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    import shaderbox_amd
    from shaderbox_amd.distributed import FramePlan, HostStagedDist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    R = shaderbox_amd.Renderer(0)
    R.set_output_format(fmt)
    ns = 2
    streams = [torch.cuda.Stream(device=dev) for _ in range(ns)]
    hd = HostStagedDist(dist, torch)
    plans = [FramePlan(R, hd, W, H, 8, exchange=exchange, channels=channels) for _ in range(ns)]
    times = [0.37 + .5 * i for i in range(nframes)]
    bad, frames = [], []
    for i, t in enumerate(times):
        with torch.cuda.stream(streams[i % ns]):
            f = plans[i % ns].render(app, t)
            if rank == 0:
                frames.append(f.clone())                   # stream-ordered behind the owner's wait: the whole frame
    torch.cuda.synchronize(dev)
    dist.barrier()
    if rank == 0:
        for t, f in zip(times, frames):
            ref = R.render(app, W, H, t)
            torch.cuda.synchronize(dev)
            a = f.view(torch.int32) if fmt == "rgba32f" else f.view(torch.int32)
            b = ref.view(torch.int32)
            bad.append(int((a != b).any(dim=-1).sum().item()) if fmt == "rgba32f" else int((a != b).sum().item()))
        json.dump({"mismatching_pixels": bad, "fault": R.fault_status(), "world": world}, open(out, "w"))
    dist.barrier()
    for p in plans[::-1]:
        if rank != 0 and p.shared is not None:
            p.shared.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
