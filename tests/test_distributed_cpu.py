"""The N>1 path on CPU: world_size 2 (and 3) over gloo, driving shaderbox_amd.distributed.FramePlan — the
same code bench.py runs over RCCL — with an oracle-backed stand-in for the GPU renderer.  Checks that
cyclic row-blocks + ONE exchange (the direct one: root in place, peers' slabs without alpha by grouped
point-to-point; and round 1's dist.gather) + assembly reproduce the single-process frame bit-for-bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class OracleRenderer:
    """CPU stand-in with the duck-typed surface FramePlan needs (render_rank / assemble / empty)."""

    def __init__(self, rgba8=False):
        from oracle.oracle import Oracle
        self.o = Oracle()
        self.rgba8 = bool(rgba8)               # SBX_FORMAT_RGBA8: every buffer holds 4 BYTES per pixel (include/sbx.h)
        self.epp = 4 if rgba8 else 3           # elements per pixel of a span slab

    def empty(self, shape, zero=False):
        return torch.zeros(tuple(shape), dtype=torch.uint8 if self.rgba8 else torch.float32)

    def render(self, app, width, height, time, out=None, mouse=(0.0, 0.0), aux=None):
        """one launch = the whole frame (what tuning.choose_exchange compares every trial frame with)"""
        from oracle.oracle import APP_IDS
        img = self.px(self.o.render(APP_IDS[app], width, height, time, mouse=mouse, aux=aux, threads=2))
        if out is not None:
            out.copy_(img)
            return out
        return img

    def fault_status(self):
        return 0

    def clear_fault(self):
        pass

    def px(self, img):
        """the oracle's float RGBA rows as this renderer's pixels"""
        return torch.from_numpy(pack_unorm8(img)) if self.rgba8 else torch.from_numpy(img)

    def render_rank_rows(self, app, width, height, time, block_rows, rank, nranks, r0, r1, slab, mouse=(0.0, 0.0),
                         aux=None, root_rounds=1, rounds=1):
        from oracle.oracle import APP_IDS
        from shaderbox_amd import shard
        rows = shard.rank_row_indices(height, block_rows, rank, nranks, root_rounds, rounds)[r0:r1]
        if rows:
            img = self.o.render_rows(APP_IDS[app], width, height, time, rows, mouse=mouse, aux=aux, threads=2)
            slab[r0:r0 + len(rows)] = self.px(img)[..., :slab.shape[-1]]      # 3-channel slabs: no alpha
        return slab

    def render_rank_in_place(self, app, width, height, time, block_rows, rank, nranks, frame, mouse=(0.0, 0.0), aux=None,
                             root_rounds=1, rounds=1, channels=4):
        from oracle.oracle import APP_IDS
        from shaderbox_amd import shard
        if isinstance(frame, CpuSharedFrame):                       # the store exchange: a peer writes into the owner's frame
            frame = frame.tensor((height, width, 4), torch.uint8 if self.rgba8 else torch.float32)
        rows = shard.rank_row_indices(height, block_rows, rank, nranks, root_rounds, rounds)
        if rows:
            img = self.px(self.o.render_rows(APP_IDS[app], width, height, time, rows, mouse=mouse, aux=aux, threads=2))
            if channels == 3 and not self.rgba8:
                frame[rows, :, :3] = img[..., :3]                   # sbx_render_split_in_place_rgb: alpha is left as it is
            else:
                frame[rows] = img
        return frame

    # -- the store exchange (include/sbx.h sbx_shared_*): a file-backed stand-in with the same protocol ------------
    def shared_create(self, nbytes, nranks):
        return CpuSharedFrame.create(nbytes, nranks, self.rgba8)

    def shared_open(self, handle):
        return CpuSharedFrame.open(handle)

    def assemble_peers(self, peers, width, height, block_rows, nranks, frame, root_rounds=1, rounds=1):
        from shaderbox_amd import shard          # mirror of k_assemble_peers (kern_util.hip)
        ch = peers.shape[-1]
        for y, (r, local) in enumerate(shard.slab_source(height, block_rows, nranks, root_rounds, rounds)):
            if r > 0:
                frame[y, :, :ch] = peers[r - 1, local]
                if ch == 3:
                    frame[y, :, 3] = 1.0            # (float slabs without alpha only)
        return frame

    # -- the span exchange: mirrors of sbx_render_span_peer / sbx_render_span_root / k_assemble_spans over the REAL span table
    #    (sbx_span_table is host code of libsbx: no GPU needed) ------------------------------------------------
    def span_table(self, app, width, height, time, block_rows, nranks, root_rounds=1, rounds=1, mouse=(0.0, 0.0), aux=None):
        import shaderbox_amd
        return shaderbox_amd.span_table(app, width, height, time, block_rows, nranks, root_rounds, rounds, mouse, aux)

    def render_span_peer(self, app, width, height, time, block_rows, rank, nranks, r0, r1, slab, mouse=(0.0, 0.0), aux=None,
                         root_rounds=1, rounds=1):
        from oracle.oracle import APP_IDS
        from shaderbox_amd import shard
        table, _, _ = self.span_table(app, width, height, time, block_rows, nranks, root_rounds, rounds, mouse, aux)
        if isinstance(slab, tuple):                  # (shared landing area, byte offset): the packed_stores exchange
            sh, off = slab
            item = 1 if self.rgba8 else 4
            slab = sh.tensor((sh.nbytes // item,), torch.uint8 if self.rgba8 else torch.float32)[int(off) // item:]
        rows = shard.rank_row_indices(height, block_rows, rank, nranks, root_rounds, rounds)[r0:r1]
        rows = [y for y in rows if table[y // block_rows][1] > table[y // block_rows][0]]
        if rows:
            img = self.px(self.o.render_rows(APP_IDS[app], width, height, time, rows, mouse=mouse, aux=aux, threads=2))
            e = self.epp
            for k, y in enumerate(rows):
                x0, x1, off, owner = (int(v) for v in table[y // block_rows])
                assert owner == rank
                at = off + (y % block_rows) * (x1 - x0)
                slab[at * e:(at + x1 - x0) * e] = img[k, x0:x1, :e].reshape(-1)
        return slab

    def render_span_peer_in_place(self, app, width, height, time, block_rows, rank, nranks, frame, mouse=(0.0, 0.0), aux=None,
                                  root_rounds=1, rounds=1, channels=4):
        """mirror of sbx_render_span_peer_in_place: the spans of the rank's row-blocks at their place in the owner's frame"""
        from oracle.oracle import APP_IDS
        from shaderbox_amd import shard
        if isinstance(frame, CpuSharedFrame):
            frame = frame.tensor((height, width, 4), torch.uint8 if self.rgba8 else torch.float32)
        table, _, _ = self.span_table(app, width, height, time, block_rows, nranks, root_rounds, rounds, mouse, aux)
        rows = [y for y in shard.rank_row_indices(height, block_rows, rank, nranks, root_rounds, rounds)
                if table[y // block_rows][1] > table[y // block_rows][0]]
        if rows:
            img = self.px(self.o.render_rows(APP_IDS[app], width, height, time, rows, mouse=mouse, aux=aux, threads=2))
            c = 4 if (self.rgba8 or channels == 4) else 3
            for k, y in enumerate(rows):
                x0, x1 = int(table[y // block_rows][0]), int(table[y // block_rows][1])
                frame[y, x0:x1, :c] = img[k, x0:x1, :c]
        return frame

    def render_span_root(self, app, width, height, time, block_rows, nranks, frame, mouse=(0.0, 0.0), aux=None, root_rounds=1,
                         rounds=1):
        from oracle.oracle import APP_IDS
        table, _, _ = self.span_table(app, width, height, time, block_rows, nranks, root_rounds, rounds, mouse, aux)
        rows = [y for y in range(height) if table[y // block_rows][3] == 0 or
                table[y // block_rows][1] - table[y // block_rows][0] < width]
        img = self.px(self.o.render_rows(APP_IDS[app], width, height, time, rows, mouse=mouse, aux=aux, threads=2))
        for k, y in enumerate(rows):
            x0, x1, off, owner = (int(v) for v in table[y // block_rows])
            if owner == 0:
                frame[y] = img[k]
            else:
                frame[y, :x0] = img[k, :x0]
                frame[y, x1:] = img[k, x1:]
        return frame

    def assemble_spans(self, app, width, height, time, block_rows, nranks, peers, stride_pixels, frame, mouse=(0.0, 0.0), aux=None,
                       root_rounds=1, rounds=1):
        table, _, _ = self.span_table(app, width, height, time, block_rows, nranks, root_rounds, rounds, mouse, aux)
        for y in range(height):
            x0, x1, off, owner = (int(v) for v in table[y // block_rows])
            if owner > 0 and x1 > x0:
                at = (owner - 1) * stride_pixels + off + (y % block_rows) * (x1 - x0)
                e = self.epp
                frame[y, x0:x1, :e] = peers[at * e:(at + x1 - x0) * e].reshape(x1 - x0, e)
                if e == 3:
                    frame[y, x0:x1, 3] = 1.0
        return frame

    def assemble(self, gathered, width, height, block_rows, nranks, out=None, root_rounds=1, rounds=1):
        from shaderbox_amd import shard          # mirror of k_assemble (kern_util.hip)
        for y, (r, local) in enumerate(shard.slab_source(height, block_rows, nranks, root_rounds, rounds)):
            out[y] = gathered[r, local]
        return out


class CpuSharedFrame:
    """CPU stand-in of sbx_shared for the gloo tests: the frame and the flag page live in a file under /dev/shm that every rank
    maps; begin / end run the library's protocol on the HOST (owner: publish the frame counter / wait for every peer's; peer: wait
    for the owner's / publish its own), so FramePlan's schedule, its handle broadcast and its channels are what is tested."""
    FLAG_WORDS = 1024

    def __init__(self, path, nbytes, nranks, owner):
        self.path, self.nbytes, self.nranks, self.owner = path, int(nbytes), int(nranks), owner
        self.mm = np.memmap(path, dtype=np.uint8, mode="r+", shape=(self.nbytes + 4 * self.FLAG_WORDS,))
        self.flags = self.mm[self.nbytes:].view(np.uint32)
        self.seq = 0

    @classmethod
    def create(cls, nbytes, nranks, rgba8):
        import tempfile
        fd, path = tempfile.mkstemp(prefix="sbx_cpu_shared_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        os.ftruncate(fd, int(nbytes) + 4 * cls.FLAG_WORDS)
        os.close(fd)
        s = cls(path, nbytes, nranks, True)
        s.rgba8 = bool(rgba8)
        if not rgba8:
            s.mm[:s.nbytes].view(np.float32).reshape(-1, 4)[:] = (0.0, 0.0, 0.0, 1.0)      # as sbx_shared_create: alpha comes with the frame
        return s

    @classmethod
    def open(cls, handle):
        path, nbytes, nranks = handle
        return cls(path, nbytes, nranks, False)

    def export(self):
        return (self.path, self.nbytes, self.nranks)

    def tensor(self, shape, dtype=None):
        a = self.mm[:self.nbytes]
        if dtype is None:
            dtype = torch.uint8 if getattr(self, "rgba8", False) else torch.float32
        a = a if dtype == torch.uint8 else a.view(np.float32)
        return torch.from_numpy(a[:int(np.prod(shape))].reshape(shape))

    def _wait(self, words, what):
        import time as _t
        t0 = _t.time()
        while any(int(np.int32(self.flags[w] - np.uint32(self.seq))) < 0 for w in words):
            if _t.time() - t0 > 120:
                raise RuntimeError("CpuSharedFrame: no signal (%s)" % what)
            _t.sleep(.001)

    def begin(self, rank):
        self.seq += 1
        if rank == 0:
            self.flags[0] = self.seq
        else:
            self._wait([0], "the owner's go")

    def end(self, rank):
        if rank == 0:
            self._wait([16 * r for r in range(1, self.nranks)], "the peers' rows")
        else:
            self.mm.flush()
            self.flags[16 * rank] = self.seq

    def close(self):
        if self.owner and os.path.exists(self.path):
            os.unlink(self.path)


def pack_unorm8(img):
    """numpy statement of the Direct3D float -> UNORM8 rule (sbx_pack_unorm8 / store_rgba's RGBA8 mode): NaN and v <= 0 -> 0,
    v > 1 -> 255, else trunc(v * 255 + .5) in binary32"""
    v = np.asarray(img, dtype=np.float32)
    with np.errstate(invalid="ignore"):
        pos = v > 0
        c = np.where(pos, np.minimum(v, np.float32(1)), np.float32(0)).astype(np.float32)
        return (c * np.float32(255) + np.float32(.5)).astype(np.uint8)


def _worker(rank, world, port, app, w, h, t, br, groups, result_path, relief=(1, 1), exchange="direct", channels=3, rgba8=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from shaderbox_amd.distributed import FramePlan
    plan = FramePlan(OracleRenderer(rgba8), dist, w, h, br, groups=groups, root_rounds=relief[0], rounds=relief[1],
                     exchange=exchange, channels=channels)
    frame = None
    for _ in range(2):                       # buffers are reused across frames
        if rank == 0 and exchange not in ("stores", "span_stores"):
            plan.frame.fill_(7 if rgba8 else -7.0)           # every pixel of the frame must be written again
        frame = plan.render(app, t)
    if rank == 0:
        np.save(result_path, np.array(frame.numpy()))
    else:
        assert frame is None
    dist.barrier()
    if exchange in ("stores", "span_stores", "packed_stores") and rank == 0:
        plan.shared.close()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("exchange,channels", [("direct", 3), ("direct", 4), ("gather", 4)])
@pytest.mark.parametrize("world,app,w,h,br,groups", [(2, "clouds", 96, 54, 8, 1), (2, "egg", 64, 45, 8, 3),
                                                      (3, "raytracer", 64, 50, 5, 2), (2, "egg", 32, 20, 8, 4)])
def test_gather_assembles_the_single_process_frame(tmp_path, oracle, world, app, w, h, br, groups, exchange, channels):
    from oracle.oracle import APP_IDS
    path = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), app, w, h, 0.37, br, groups, path, (1, 1), exchange, channels),
             nprocs=world, join=True)
    got = np.load(path)
    ref = oracle.render(APP_IDS[app], w, h, 0.37)
    assert got.shape == ref.shape
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("exchange", ["direct", "gather"])
@pytest.mark.parametrize("world,app,w,h,br,groups,relief", [(2, "egg", 64, 45, 4, 1, (1, 3)), (3, "clouds", 96, 54, 2, 2, (2, 5)),
                                                             (3, "raytracer", 64, 50, 5, 1, (0, 2))])
def test_gather_with_root_relief(tmp_path, oracle, world, app, w, h, br, groups, relief, exchange):
    """the split that deals the gather's root fewer row-blocks (shard.py) assembles the same frame, incl. a root that
    renders nothing at all"""
    from oracle.oracle import APP_IDS
    path = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), app, w, h, 0.37, br, groups, path, relief, exchange), nprocs=world, join=True)
    got = np.load(path)
    ref = oracle.render(APP_IDS[app], w, h, 0.37)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("world,app,w,h,br,groups,relief", [(2, "clouds", 96, 54, 8, 1, (1, 1)),      # empty spans below the horizon
                                                             (3, "clouds", 128, 72, 4, 3, (1, 2)),
                                                             (2, "atmosphere", 448, 252, 8, 2, (1, 1)),  # partial spans: the dome
                                                             (3, "planet", 448, 96, 8, 1, (2, 3)),
                                                             (2, "egg", 64, 45, 8, 2, (1, 1))])         # no span model: whole rows
def test_span_exchange_assembles_the_single_process_frame(tmp_path, oracle, world, app, w, h, br, groups, relief):
    """exchange='spans': only the expensive interval of every row-block is dealt out and sent; the root renders the rest of
    every block itself; the frame is the single-process frame bit for bit whatever the span table says"""
    import shaderbox_amd
    from oracle.oracle import APP_IDS
    table, pix, maxw = shaderbox_amd.span_table(app, w, h, 0.37, br, world, relief[0], relief[1])
    if app in ("atmosphere", "planet"):
        part = [int(t[1] - t[0]) for t in table if t[3] > 0]
        assert maxw > 0 and any(0 < v < w for v in part), "the case is meant to have partial spans"
    path = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), app, w, h, 0.37, br, groups, path, relief, "spans"), nprocs=world, join=True)
    got = np.load(path)
    ref = oracle.render(APP_IDS[app], w, h, 0.37)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("world,app,w,h,br,groups,relief,exchange", [(2, "atmosphere", 448, 252, 8, 2, (1, 1), "spans"),
                                                                      (3, "clouds", 128, 72, 4, 3, (1, 2), "spans"),
                                                                      (2, "egg", 64, 45, 8, 3, (1, 1), "direct"),
                                                                      (3, "raytracer", 64, 50, 5, 2, (0, 2), "direct"),
                                                                      (2, "egg", 64, 45, 8, 2, (1, 1), "gather")])
def test_rgba8_exchange_assembles_the_packed_frame(tmp_path, oracle, world, app, w, h, br, groups, relief, exchange):
    """SBX_FORMAT_RGBA8: FramePlan with 4-byte pixels everywhere (slabs, landing areas, frame) over gloo == the packed
    single-process frame"""
    from oracle.oracle import APP_IDS
    path = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), app, w, h, 0.37, br, groups, path, relief, exchange, 3, True), nprocs=world, join=True)
    got = np.load(path)
    ref = pack_unorm8(oracle.render(APP_IDS[app], w, h, 0.37))
    assert got.dtype == np.uint8 and got.shape == ref.shape and np.array_equal(got, ref)


@pytest.mark.parametrize("world,app,w,h,br,relief,channels,rgba8", [(2, "clouds", 96, 54, 8, (1, 1), 3, False), (3, "egg", 64, 45, 4, (1, 2), 4, False),
                                                                    (3, "raytracer", 64, 50, 5, (0, 1), 3, False), (2, "egg", 64, 45, 8, (1, 1), 3, True)])
def test_store_exchange_schedule_over_gloo(tmp_path, oracle, world, app, w, h, br, relief, channels, rgba8):
    """exchange='stores' (the peers write their row-blocks IN PLACE into the owner's frame, FramePlan._render_stores): the handle
    broadcast, the begin / end protocol of every rank and the 3-channel stores (alpha untouched) over gloo with a file-backed
    stand-in for sbx_shared == the single-process frame, two frames in a row"""
    from oracle.oracle import APP_IDS
    path = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), app, w, h, 0.37, br, 1, path, relief, "stores", channels, rgba8), nprocs=world, join=True)
    got = np.load(path)
    ref = oracle.render(APP_IDS[app], w, h, 0.37)
    if rgba8:
        assert got.dtype == np.uint8 and np.array_equal(got, pack_unorm8(ref))
    else:
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("world,app,w,h,br,relief,channels", [(2, "atmosphere", 448, 252, 8, (1, 1), 3), (3, "clouds", 128, 72, 4, (1, 2), 4),
                                                              (2, "egg", 64, 45, 8, (1, 1), 3)])
def test_span_store_exchange_schedule_over_gloo(tmp_path, oracle, world, app, w, h, br, relief, channels):
    """exchange='span_stores': the store exchange with only the spans of the peers' row-blocks stored (partial spans, empty spans,
    an app without a span model) over gloo with the file-backed stand-in == the single-process frame"""
    from oracle.oracle import APP_IDS
    path = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), app, w, h, 0.37, br, 1, path, relief, "span_stores", channels, False), nprocs=world, join=True)
    got = np.load(path)
    ref = oracle.render(APP_IDS[app], w, h, 0.37)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("world,app,w,h,br,relief,rgba8", [(2, "atmosphere", 448, 252, 8, (1, 1), False), (3, "clouds", 128, 72, 4, (1, 2), False),
                                                           (2, "egg", 64, 45, 8, (1, 1), True)])
def test_packed_store_exchange_schedule_over_gloo(tmp_path, oracle, world, app, w, h, br, relief, rgba8):
    """exchange='packed_stores': the span exchange whose transport is the peers' own stores — every peer renders its packed spans
    into its stretch of the owner's landing area (the shared object, created and broadcast at the first frame), the owner waits
    for the signals and scatters — over gloo with the file-backed stand-in == the single-process frame (float and RGBA8 pixels)"""
    from oracle.oracle import APP_IDS
    path = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), app, w, h, 0.37, br, 1, path, relief, "packed_stores", 3, rgba8), nprocs=world, join=True)
    got = np.load(path)
    ref = oracle.render(APP_IDS[app], w, h, 0.37)
    if rgba8:
        assert got.dtype == np.uint8 and np.array_equal(got, pack_unorm8(ref))
    else:
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_span_table_is_a_consistent_layout():
    """offsets tile every rank's packed slab exactly; spans are aligned intervals; owners follow the split"""
    import shaderbox_amd
    from shaderbox_amd import shard
    for app, w, h, br, n, m0, m in [("atmosphere", 7680, 4320, 8, 8, 1, 1), ("planet", 7680, 4320, 8, 8, 3, 4),
                                    ("clouds", 3840, 2160, 8, 8, 1, 2), ("clouds", 1000, 333, 5, 3, 1, 1), ("egg", 100, 50, 8, 2, 1, 1)]:
        table, pix, maxw = shaderbox_amd.span_table(app, w, h, 0.37, br, n, m0, m)
        src = shard.slab_source(h, br, n, m0, m)
        for r in range(n):
            at = 0
            for g in shard._rank_blocks(h, br, r, n, m0, m):
                x0, x1, off, owner = (int(v) for v in table[g])
                rows = min(h, (g + 1) * br) - g * br
                assert owner == r == src[g * br][0] and off == at and 0 <= x0 <= x1 <= w
                assert x0 % 64 == 0 and (x1 % 64 == 0 or x1 == w)
                at += rows * (x1 - x0)
            assert at == pix[r]
        assert maxw == max([int(t[1] - t[0]) for t in table if t[3] > 0] + [0])
    # the tables that matter at the BASELINE sizes: what fraction of the peers' pixels still crosses xGMI
    for app, w, h, lo, hi in [("atmosphere", 7680, 4320, .48, .58), ("planet", 7680, 4320, .55, .65), ("clouds", 3840, 2160, .70, .78)]:
        table, pix, _ = shaderbox_amd.span_table(app, w, h, 0.37, 8, 8)
        full = sum(shard.rank_rows(h, 8, r, 8) * w for r in range(1, 8))
        assert lo < sum(int(p) for p in pix[1:]) / full < hi, (app, sum(pix[1:]) / full)


def _choose_worker(rank, world, port, app, w, h, t, br, result_path, allow_stores, budget_s):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from shaderbox_amd import tuning
    from shaderbox_amd.distributed import choose_exchange          # where a host finds it: beside FramePlan
    tuning.CONFIG.coll_dev = torch.device("cpu")
    choice = choose_exchange(OracleRenderer(), dist, torch, torch.device("cpu"), [None, None], app, w, h, t, world=world, rank=rank,
                             block_rows=br, root_rounds="1/1", groups="auto", budget_s=budget_s, preroll_ms=0, allow_stores=allow_stores,
                             trial_frames=2)
    # the chosen plans are ready to run: one frame at another time, compared by the caller
    frame = choice.plans[0].render(app, t + .25)
    if rank == 0:
        import json
        np.save(result_path, np.array(frame.numpy()))
        json.dump({"exchange": choice.exchange, "channels": choice.channels, "relief": list(choice.relief), "groups": choice.groups,
                   "payload": choice.payload_bytes_per_peer, "trials": choice.trials, "notes": choice.notes},
                  open(result_path + ".json", "w"))
    dist.barrier()
    for pl in choice.plans:
        if getattr(pl, "shared", None) is not None and rank == 0:
            pl.shared.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("allow_stores,budget_s", [(False, 600.0), (True, 600.0), (True, 0.0)])
def test_choose_exchange_over_gloo(tmp_path, oracle, allow_stores, budget_s):
    """shaderbox_amd.tuning.choose_exchange with world 2: every candidate is set up on both ranks, soaked against one-launch renders of
    frames at different times, timed; the fastest runs; the store forms are candidates only where allowed (and after the IPC
    pre-flight); a spent time budget cuts the later candidates and says which"""
    import json
    from oracle.oracle import APP_IDS
    app, w, h, br = "egg", 48, 32, 8
    path = str(tmp_path / "frame.npy")
    mp.spawn(_choose_worker, args=(2, _free_port(), app, w, h, 0.37, br, path, allow_stores, budget_s), nprocs=2, join=True)
    rec = json.load(open(path + ".json"))
    forms = ["spans", "direct"] + (["stores", "stores_16B", "span_stores", "span_stores_16B", "packed_stores"] if allow_stores else [])
    assert rec["notes"]["candidates"] == forms
    if budget_s > 0:
        assert rec["notes"]["cut"] == [] and all(isinstance(rec["trials"][f], float) and rec["trials"][f] > 0 for f in forms)
        assert "(first trial, discarded) spans" in rec["trials"]
    else:                                   # no budget: the first candidate (twice: its first reading is discarded) and nothing else
        assert rec["notes"]["cut"] == forms[1:] and rec["exchange"] == "spans" and set(rec["trials"]) == {"(first trial, discarded) spans", "spans"}
    best = min((f for f in forms if isinstance(rec["trials"].get(f), float)), key=lambda f: rec["trials"][f])
    assert form_of(rec["exchange"], rec["channels"]) == best and rec["relief"] == [1, 1] and rec["groups"] >= 1 and rec["payload"] > 0
    got = np.load(path)
    ref = oracle.render(APP_IDS[app], w, h, 0.62)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def form_of(exchange, channels):
    return exchange + ("_16B" if (exchange in ("stores", "span_stores") and channels == 4) else "")
