"""Multi-GPU frame: cyclic row-blocks per rank + ONE gather (optionally pipelined) + root-side assembly.

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, xGMI underneath).  The path has
exactly one exchange step — bringing the rendered rows to the rank that owns the frame — so the only
collective is a gather to rank 0.  Seven peers send over seven distinct xGMI links (point-to-point, no
ring), 16.6 MB each for a 3840x2160 RGBA32F frame.  Everything else is embarrassingly parallel: every
rank computes its pixels from their GLOBAL coordinates, so the assembled frame is bit-identical to a
single-GPU render.

Pipelining: at 8 GPUs a rank's strip takes ~1.5 ms while its 16.6 MB take ~0.2-0.3 ms on one xGMI link, so
the gather is issued in `groups` pieces: slab rows are rendered group by group and each group's gather is
started asynchronously (RCCL runs on its own stream) while the next group renders; only the last piece
and the assembly kernel are exposed.  groups = 1 is the plain single gather.

Two forms of that one exchange (`exchange=`):

* "direct" (default): the root renders its own row-blocks IN PLACE into the frame and posts one grouped receive per piece
  (`batch_isend_irecv`: one ncclGroup, the 7 peers arrive in parallel over their 7 links), the peers send their slabs
  WITHOUT alpha — `channels=3`: alpha is the constant 1 that the caller of mainImage writes (src/main.h:52), so 12 instead
  of 16 bytes per pixel cross xGMI and land in the root's HBM — and the root scatters the peers' rows into the frame
  (`assemble_peers`, which writes the alpha).  No self-copy of the root's slab, no assembly of the root's rows.
* "gather": `dist.gather` of equal RGBA slabs (the root's included) + `assemble` of all of them — round 1's form, kept for
  comparison and as the fallback of backends without grouped point-to-point.

* "spans": the direct exchange with only the EXPENSIVE part of every row-block sharded.  The library's span table
  (`sbx_span_table`, include/sbx.h) gives, per row-block, the interval of columns in which mainImage is expected to get past
  its early exit (APP_CLOUDS above the horizon, APP_ATMOSPHERE where some view sample is above the ground, APP_PLANET where the
  ray meets the atmosphere shell); a peer renders and sends only those intervals, packed; the root renders its own blocks AND
  everything outside the peers' intervals in one in-place launch over the frame (cheap pixels by construction of the table,
  and the same kernel, so the table can only cost balance, never a pixel), then scatters the packed slabs.  Still exactly one
  exchange step.  At 7680x4320 a peer's 49.8 MB become 29.7 MB (ATMOSPHERE, PLANET); at 4K APP_CLOUDS 12.4 -> 9.2 MB.

* "stores": the exchange in which the frame's owner does NOTHING for the others (include/sbx.h sbx_shared_*).  Rank 0 allocates
  the frame through the library and exports it (hipIpcGetMemHandle; the handle travels by one broadcast_object_list when the plan
  is built); every peer maps it (hipIpcOpenMemHandle) and renders its row-blocks IN PLACE into it from its own GPU — the one
  exchange step is the render kernels' own pixel stores over xGMI: 16 bytes per pixel, 12 with `channels=3` (three dwords; the
  constant alpha is written once when the frame is created) or 4 ('rgba8').  No landing area, no RCCL receive kernels on the
  root, no scatter pass: the root is an ordinary rank.  Two flag kernels per frame and rank order it (owner: "frame may be
  overwritten" / wait for every peer's "rows in place"; peer: wait / signal).  Still exactly one exchange step, and since every
  pixel is written by the app's full kernel from its global coordinates the frame is bit-identical to one launch.
* "span_stores": both ideas together, for frames whose pixel stores would bind a link (7680x4320 in float pixels: 50 MB per peer):
  a peer stores only the SPANS of its row-blocks in place (`sbx_render_span_peer_in_place`), the owner renders its blocks and
  everything outside the spans (`sbx_render_span_root`): 30 MB per peer, no landing, no scatter.
* "packed_stores": the span exchange with the peers' own stores as its transport.  Storing R, G, B into 16-byte pixels leaves a hole
  in every pixel, and a link may take such partial stores far below its rate (PCIe: 11.9 against 52 GB/s, profiles/r05_link_stores.txt);
  a PACKED 3-float slab has no holes (52 GB/s on the same link).  So the owner's LANDING AREA is the shared object: a peer renders its
  packed spans straight into its stretch of it (`sbx_render_span_peer` on the mapped pointer), the owner waits for the signals and
  scatters (`sbx_assemble_spans`).  12 contiguous bytes per pixel on the link, no RCCL call, no receive kernels on the owner.

Pixel format: the plan moves whatever pixels its renderer writes.  After `renderer.set_output_format("rgba8")` (include/sbx.h
SBX_FORMAT_RGBA8: the render kernels write one R8G8B8A8_UNORM word per pixel, the reference hosts' display format) every slab,
landing area and frame of the plan is a uint8 tensor with 4 bytes per pixel — a third of the float exchange's bytes on every link
and in the root's HBM; the assembled frame equals pack_unorm8 of the float frame.

`renderer` is duck-typed (render_rank_rows / render_rank_in_place / assemble / assemble_peers / empty):
shaderbox_amd.Renderer on GPUs; the CPU tests drive the same code over gloo with an oracle-backed stand-in.
"""
from . import shard


def choose_exchange(*args, **kwargs):
    """shaderbox_amd.tuning.choose_exchange: the exchange form (and the root's relief, the pieces, the plans) for these ranks, chosen by
    trying the candidates on them.  Here so that a host finds it beside FramePlan."""
    from .tuning import choose_exchange as f
    return f(*args, **kwargs)


class FramePlan:
    """Buffers and schedule of one rank for repeated frames of a fixed size."""

    def __init__(self, renderer, dist, width, height, block_rows=shard.DEFAULT_BLOCK_ROWS, groups=1, root_rounds=1,
                 rounds=1, exchange="direct", channels=3):
        if exchange not in ("direct", "gather", "spans", "stores", "span_stores", "packed_stores"):
            raise ValueError("exchange must be 'direct', 'gather', 'spans', 'stores', 'span_stores' or 'packed_stores'")
        if exchange in ("spans", "packed_stores"):
            channels = 3
        if exchange == "gather":
            channels = 4
        if channels not in (3, 4):
            raise ValueError("channels must be 3 or 4")
        self.r, self.dist = renderer, dist
        self.rgba8 = bool(getattr(renderer, "rgba8", False))
        if self.rgba8:
            channels = 4                                       # 4 BYTES per pixel: one word, whatever the exchange
        self.exchange, self.channels = exchange, int(channels)
        self.span_epp = 4 if self.rgba8 else 3                 # buffer elements per pixel of a span slab
        self.width, self.height, self.block_rows = int(width), int(height), int(block_rows)
        self.world = dist.get_world_size()
        self.rank = dist.get_rank()
        self.root_rounds, self.rounds = int(root_rounds), int(rounds)      # root relief (shard.py); 1, 1 = plain cyclic
        self.rows_max = shard.rank_rows_max(self.height, self.block_rows, self.world, self.root_rounds, self.rounds)
        nblocks = self.rows_max // self.block_rows
        groups = max(1, min(int(groups), nblocks))
        # slab row ranges of the groups: whole blocks, as even as possible
        cuts = [((g * nblocks) // groups) * self.block_rows for g in range(groups + 1)]
        self.ranges = [(cuts[g], cuts[g + 1]) for g in range(groups) if cuts[g + 1] > cuts[g]]
        self.slab = self.gathered = self.peers = self.frame = self.shared = None
        self.glists = [None] * len(self.ranges)
        if exchange in ("stores", "span_stores"):
            # collective: every rank builds its plans in the same order, the handle of the owner's frame goes round once
            box = [None]
            if self.rank == 0:
                try:
                    self.shared = renderer.shared_create(self.height * self.width * (4 if self.rgba8 else 16), self.world)
                    self.frame = self.shared.tensor((self.height, self.width, 4))
                    if self.world > 1:
                        box[0] = self.shared.export()
                except Exception as e:                                   # noqa: BLE001
                    # the peers are about to enter the broadcast: tell them instead of leaving them there
                    box[0] = ("error", "%s: %s" % (type(e).__name__, e))
                    if self.world <= 1:
                        raise
            if self.world > 1:
                dist.broadcast_object_list(box, src=0)
                if isinstance(box[0], tuple) and box[0] and box[0][0] == "error":
                    raise RuntimeError("FramePlan(exchange=%r): rank 0 could not create or export the shared frame: %s" % (exchange, box[0][1]))
                if self.rank != 0:
                    self.shared = renderer.shared_open(box[0])
        elif exchange == "gather":
            self.slab = renderer.empty((self.rows_max, self.width, 4), zero=True)
            if self.rank == 0:
                self.gathered = renderer.empty((self.world, self.rows_max, self.width, 4))
                self.glists = [[self.gathered[i, a:b] for i in range(self.world)] for a, b in self.ranges]
                self.frame = renderer.empty((self.height, self.width, 4))
        elif exchange in ("spans", "packed_stores"):
            if self.rank == 0:
                self.frame = renderer.empty((self.height, self.width, 4))
            self._span_key = None        # buffers and descriptors depend on the app's span table: built at the first render
        elif self.rank == 0:
            self.frame = renderer.empty((self.height, self.width, 4))
            if self.world > 1:
                self.peers = renderer.empty((self.world - 1, self.rows_max, self.width, self.channels), zero=True)
        else:
            self.slab = renderer.empty((self.rows_max, self.width, self.channels), zero=True)
        # the point-to-point descriptors of the direct exchange, built once (they only name fixed buffers and peers)
        self.p2p = []
        if exchange == "direct" and self.world > 1:          # ("spans": built with the layout, _span_layout)
            for a, b in self.ranges:
                if self.rank == 0:
                    self.p2p.append([dist.P2POp(dist.irecv, self.peers[i - 1, a:b], i) for i in range(1, self.world)])
                else:
                    self.p2p.append([dist.P2POp(dist.isend, self.slab[a:b], 0)])

    # -- the span exchange ------------------------------------------------------------------------------------
    def _span_layout(self, app, time, mouse, aux):
        """(Re)build buffers and point-to-point descriptors for the span table of (app, mouse): the table depends on the camera
        and the split only, so an animation keeps one layout.  (That is not a property this code hopes for: the library CACHES the
        device table under exactly (app, u_res, u_mouse, split) — csrc/sbx_capi.hip span_table_device — so the table the kernels
        read for a later u_time or aux block IS the one this layout was sized from.)"""
        key = (str(app), float(mouse[0]), float(mouse[1]))
        if self._span_key == key:
            return
        table, pix, _ = self.r.span_table(app, self.width, self.height, time, self.block_rows, self.world,
                                          self.root_rounds, self.rounds, mouse=mouse, aux=aux)
        self.span_pixels = [int(p) for p in pix]
        # every peer's slab starts at a multiple of `stride` pixels in the root's landing area
        self.span_stride = (max(self.span_pixels[1:] + [0]) + 63) // 64 * 64
        br = self.block_rows

        def pixel_range(rank, a, b):             # packed-pixel interval of slab rows [a, b) of `rank`
            blocks = shard._rank_blocks(self.height, br, rank, self.world, self.root_rounds, self.rounds)
            lo, hi = a // br, b // br

            def at(i):
                return int(table[blocks[i]][2]) if i < len(blocks) else self.span_pixels[rank]
            return at(min(lo, len(blocks))), at(min(hi, len(blocks)))
        d = self.dist
        self.p2p = []
        if self.exchange == "packed_stores":
            # the landing area lives in a shared object of the library: the owner creates and exports it, every peer maps it and
            # will render its packed spans straight into its own stretch of it (collective, like the plan's construction)
            if self.shared is not None:
                self.shared.close()
                self.shared = None
            total = max(self.world - 1, 1) * max(self.span_stride, 1) * self.span_epp
            nbytes = total * (1 if self.rgba8 else 4)
            box = [None]
            if self.rank == 0:
                try:
                    self.shared = self.r.shared_create(nbytes, self.world)
                    self.peers = self.shared.tensor((total,))
                    if self.world > 1:
                        box[0] = self.shared.export()
                except Exception as e:                                   # noqa: BLE001
                    box[0] = ("error", "%s: %s" % (type(e).__name__, e))
                    if self.world <= 1:
                        raise
            if self.world > 1:
                d.broadcast_object_list(box, src=0)
                if isinstance(box[0], tuple) and box[0] and box[0][0] == "error":
                    raise RuntimeError("FramePlan(exchange='packed_stores'): rank 0 could not create or export the landing area: %s" % (box[0][1],))
                if self.rank != 0:
                    self.shared = self.r.shared_open(box[0])
                    self.slab = (self.shared, (self.rank - 1) * self.span_stride * self.span_epp * (1 if self.rgba8 else 4))
            self._span_key = key
            return
        if self.rank == 0:
            self.peers = self.r.empty((max(self.world - 1, 1) * max(self.span_stride, 1) * self.span_epp,), zero=True)
            for a, b in self.ranges:
                ops = []
                for i in range(1, self.world):
                    lo, hi = pixel_range(i, a, b)
                    if hi > lo:
                        base = (i - 1) * self.span_stride
                        ops.append(d.P2POp(d.irecv, self.peers[(base + lo) * self.span_epp:(base + hi) * self.span_epp], i))
                self.p2p.append(ops)
        else:
            self.slab = self.r.empty((max(self.span_pixels[self.rank], 1) * self.span_epp,), zero=True)
            for a, b in self.ranges:
                lo, hi = pixel_range(self.rank, a, b)
                self.p2p.append([d.P2POp(d.isend, self.slab[lo * self.span_epp:hi * self.span_epp], 0)] if hi > lo else [])
        self._span_key = key

    def _render_spans(self, app, time, mouse, aux, mark):
        self._span_layout(app, time, mouse, aux)
        d = self.dist
        works = []
        if self.rank == 0:
            for ops in self.p2p:
                if ops:
                    works += d.batch_isend_irecv(ops)
            self.r.render_span_root(app, self.width, self.height, time, self.block_rows, self.world, self.frame, mouse=mouse,
                                    aux=aux, root_rounds=self.root_rounds, rounds=self.rounds)
            mark("render")
            for w in works:
                w.wait()
            mark("exchange")
            if self.world > 1:
                self.r.assemble_spans(app, self.width, self.height, time, self.block_rows, self.world, self.peers,
                                      self.span_stride, self.frame, mouse=mouse, aux=aux, root_rounds=self.root_rounds,
                                      rounds=self.rounds)
            mark("assemble")
            return self.frame
        for g, (a, b) in enumerate(self.ranges):
            self.r.render_span_peer(app, self.width, self.height, time, self.block_rows, self.rank, self.world, a, b, self.slab,
                                    mouse=mouse, aux=aux, root_rounds=self.root_rounds, rounds=self.rounds)
            if self.p2p[g]:
                works += d.batch_isend_irecv(self.p2p[g])
        mark("render")
        for w in works:
            w.wait()
        mark("exchange")
        return None

    def _render_stores(self, app, time, mouse, aux, mark, phase):
        """the store exchange: begin (owner: release the frame; peer: wait for it), the rank's rows in place, end (peer: signal;
        owner: wait for every peer).  `phase` splits the owner's call for hosts that drive all ranks from ONE thread on one stream
        (LoopbackWorld): "open" = begin, "close" = render + end."""
        if phase in ("all", "open"):
            self.shared.begin(self.rank)
        if phase == "open":
            return None
        ch = 4 if self.rgba8 else self.channels
        if self.exchange == "span_stores" and self.world > 1:
            # with spans: a peer stores only the expensive interval of each of its row-blocks, the owner renders its own blocks and
            # everything outside the spans (one launch over the frame; those pixels leave mainImage through an early exit)
            if self.rank == 0:
                self.r.render_span_root(app, self.width, self.height, time, self.block_rows, self.world, self.frame, mouse=mouse,
                                        aux=aux, root_rounds=self.root_rounds, rounds=self.rounds)
            else:
                self.r.render_span_peer_in_place(app, self.width, self.height, time, self.block_rows, self.rank, self.world,
                                                 self.shared, mouse=mouse, aux=aux, root_rounds=self.root_rounds, rounds=self.rounds,
                                                 channels=ch)
        else:
            self.r.render_rank_in_place(app, self.width, self.height, time, self.block_rows, self.rank, self.world,
                                        self.frame if self.rank == 0 else self.shared, mouse=mouse, aux=aux,
                                        root_rounds=self.root_rounds, rounds=self.rounds, channels=ch)
        mark("render")
        self.shared.end(self.rank)
        mark("exchange")
        if self.rank == 0:
            mark("assemble")                      # nothing to assemble: the frame is complete in stream order
            return self.frame
        return None

    def _render_packed_stores(self, app, time, mouse, aux, mark, phase):
        """the span exchange with the peers' own stores as its transport: a peer renders its packed spans (12 contiguous bytes per
        pixel: whole lines on the link, unlike 12-byte stores into 16-byte pixels) straight into its stretch of the owner's landing
        area; the owner renders its blocks and everything outside the spans, waits for the peers' signals and scatters.  No RCCL
        call and no receive kernels on the owner; one scatter pass more than "span_stores"."""
        self._span_layout(app, time, mouse, aux)
        if phase in ("all", "open"):
            self.shared.begin(self.rank)
        if phase == "open":
            return None
        if self.rank == 0:
            self.r.render_span_root(app, self.width, self.height, time, self.block_rows, self.world, self.frame, mouse=mouse,
                                    aux=aux, root_rounds=self.root_rounds, rounds=self.rounds)
            mark("render")
            self.shared.end(0)                         # every peer's spans are in the landing area
            mark("exchange")
            if self.world > 1:
                self.r.assemble_spans(app, self.width, self.height, time, self.block_rows, self.world, self.peers,
                                      self.span_stride, self.frame, mouse=mouse, aux=aux, root_rounds=self.root_rounds,
                                      rounds=self.rounds)
            mark("assemble")
            return self.frame
        self.r.render_span_peer(app, self.width, self.height, time, self.block_rows, self.rank, self.world, 0, 1 << 30, self.slab,
                                mouse=mouse, aux=aux, root_rounds=self.root_rounds, rounds=self.rounds)
        mark("render")
        self.shared.end(self.rank)
        mark("exchange")
        return None

    def render(self, app, time, mouse=(0.0, 0.0), aux=None, mark=None, phase="all"):
        """All ranks call this; rank 0 returns the assembled [H, W, 4] frame, the others None.  `mark(name)`, if given, is
        called after the rank's rendering ("render"), after the waits of its exchange ("exchange") and after the root's
        assembly ("assemble"): bench.py records stream events there to time the phases of a frame."""
        mark = mark or (lambda name: None)
        if self.exchange in ("stores", "span_stores"):
            return self._render_stores(app, time, mouse, aux, mark, phase)
        if self.exchange == "packed_stores":
            return self._render_packed_stores(app, time, mouse, aux, mark, phase)
        if self.exchange == "gather":
            return self._render_gather(app, time, mouse, aux, mark)
        if self.exchange == "spans":
            return self._render_spans(app, time, mouse, aux, mark)
        d = self.dist
        works = []
        if self.rank == 0:
            # the receives first: they depend only on the previous use of `peers` (earlier on this stream), not on the
            # root's own rendering, which then runs beside them
            for ops in self.p2p:
                works += d.batch_isend_irecv(ops)
            self.r.render_rank_in_place(app, self.width, self.height, time, self.block_rows, 0, self.world, self.frame,
                                        mouse=mouse, aux=aux, root_rounds=self.root_rounds, rounds=self.rounds)
            mark("render")
            for w in works:
                w.wait()
            mark("exchange")
            frame = self.r.assemble_peers(self.peers, self.width, self.height, self.block_rows, self.world, self.frame,
                                          root_rounds=self.root_rounds, rounds=self.rounds)
            mark("assemble")
            return frame
        for g, (a, b) in enumerate(self.ranges):
            self.r.render_rank_rows(app, self.width, self.height, time, self.block_rows, self.rank, self.world,
                                    a, b, self.slab, mouse=mouse, aux=aux, root_rounds=self.root_rounds,
                                    rounds=self.rounds)
            works += d.batch_isend_irecv(self.p2p[g])
        mark("render")
        for w in works:
            w.wait()          # stream-level on GPUs: the next frame's render into this slab is ordered after the send
        mark("exchange")
        return None

    def _render_gather(self, app, time, mouse, aux, mark):
        works = []
        last = len(self.ranges) - 1
        for g, (a, b) in enumerate(self.ranges):
            self.r.render_rank_rows(app, self.width, self.height, time, self.block_rows, self.rank, self.world,
                                    a, b, self.slab, mouse=mouse, aux=aux, root_rounds=self.root_rounds,
                                    rounds=self.rounds)
            # the gather of the path (one logical gather, issued per group so that it overlaps the next render)
            w = self.dist.gather(self.slab[a:b], self.glists[g], dst=0, async_op=(g != last))
            if w is not None:
                works.append(w)
        mark("render")
        for w in works:
            w.wait()          # stream-level wait on GPUs: the slab/gathered buffers are safe to reuse/read after it
        mark("exchange")
        if self.rank == 0:
            frame = self.r.assemble(self.gathered, self.width, self.height, self.block_rows, self.world,
                                    out=self.frame, root_rounds=self.root_rounds, rounds=self.rounds)
            mark("assemble")
            return frame
        return None


class LoopbackWorld:
    """An N-rank world inside ONE process on ONE device: what tests and tools use to run the real FramePlan schedule of every
    rank — the real kernels, buffers, span tables and assembly — where only one GPU exists.  `rank(i)` is a stand-in for
    torch.distributed as FramePlan uses it (get_rank / get_world_size / P2POp / isend / irecv / batch_isend_irecv): a send
    stashes its tensor view, the matching receive copies from the stash on the current stream.  Per frame the callers must run
    the peers' `render` before the root's (the root posts its receives first).  An emulation: it measures nothing about xGMI."""

    class _Work:
        def wait(self):
            return True

    class _Op:
        def __init__(self, op, tensor, peer):
            self.op, self.tensor, self.peer = op, tensor, peer

    class _Rank:
        isend, irecv = "isend", "irecv"

        def __init__(self, world, rank):
            self._w, self._rank = world, rank
            self.P2POp = LoopbackWorld._Op

        def get_world_size(self):
            return self._w.n

        def get_rank(self):
            return self._rank

        def batch_isend_irecv(self, ops):
            for op in ops:
                if op.op == self.isend:
                    self._w.box.setdefault((self._rank, op.peer), []).append(op.tensor)
                else:
                    q = self._w.box.get((op.peer, self._rank))
                    if not q:
                        raise RuntimeError("LoopbackWorld: rank %d receives from %d before it sent (run the peers first)"
                                           % (self._rank, op.peer))
                    src = q.pop(0)
                    if src.numel() != op.tensor.numel():
                        raise RuntimeError("LoopbackWorld: send of %d elements meets a receive of %d" % (src.numel(), op.tensor.numel()))
                    op.tensor.copy_(src.reshape(op.tensor.shape))
                    self._w.bytes_moved += src.numel() * src.element_size()
            return [LoopbackWorld._Work()]

        def broadcast_object_list(self, objs, src=0):
            """FramePlan's one object broadcast (the handle of a shared frame): the source's k-th call feeds every rank's k-th"""
            log = self._w.objs.setdefault(src, [])
            if self._rank == src:
                log.append(list(objs))
            else:
                k = self._w.obj_next.get((src, self._rank), 0)
                if k >= len(log):
                    raise RuntimeError("LoopbackWorld: rank %d reads a broadcast rank %d has not made (build the plans in rank order)"
                                       % (self._rank, src))
                objs[:] = log[k]
                self._w.obj_next[(src, self._rank)] = k + 1

    def __init__(self, n):
        self.n = int(n)
        self.box = {}
        self.objs, self.obj_next = {}, {}
        self.bytes_moved = 0

    def rank(self, i):
        return LoopbackWorld._Rank(self, int(i))

    def plans(self, renderer, width, height, **kw):
        """one FramePlan per rank, all on `renderer`'s device"""
        return [FramePlan(renderer, self.rank(i), width, height, **kw) for i in range(self.n)]

    @staticmethod
    def render(plans, app, time, **kw):
        """one frame through all ranks in the order the emulation needs: peers, then the root; returns the root's frame.  (The
        store exchange runs on ONE stream here, so the owner's "frame may be overwritten" has to be enqueued before the peers'
        waits for it: the owner's call is split, FramePlan._render_stores.)"""
        if plans[0].exchange in ("stores", "span_stores", "packed_stores"):
            plans[0].render(app, time, phase="open", **kw)
            for p in plans[1:]:
                p.render(app, time, **kw)
            return plans[0].render(app, time, phase="close", **kw)
        for p in plans[1:]:
            p.render(app, time, **kw)
        return plans[0].render(app, time, **kw)


class HostStagedDist:
    """torch.distributed as FramePlan uses it, for process groups whose backend cannot move device tensors point to point (gloo):
    every send / receive goes through a pinned host buffer.  It exists so that the WHOLE N > 1 program — one process per rank,
    real rendezvous, real collectives, FramePlan's schedule, bench.py's orchestration — can run where the ranks cannot form an RCCL
    communicator (several ranks on the one GPU of a test box: RCCL refuses duplicate devices).  Not a product path: bench.py uses
    it only under `--backend gloo`, and says so in its line."""

    isend, irecv = "isend", "irecv"

    class _Op:
        def __init__(self, op, tensor, peer):
            self.op, self.tensor, self.peer = op, tensor, peer

    class _Work:
        def __init__(self, fn):
            self._fn = fn

        def wait(self):
            if self._fn is not None:
                self._fn()
                self._fn = None
            return True

    def __init__(self, dist, torch):
        self._d, self._t = dist, torch
        self.P2POp = HostStagedDist._Op

    def get_world_size(self):
        return self._d.get_world_size()

    def get_rank(self):
        return self._d.get_rank()

    def broadcast_object_list(self, objs, src=0):
        return self._d.broadcast_object_list(objs, src=src)

    def batch_isend_irecv(self, ops):
        t = self._t
        works = []
        for op in ops:
            if op.op == self.isend:
                host = op.tensor.detach().to("cpu")                  # (synchronises with the producing launch)
                w = self._d.isend(host, op.peer)
                works.append(HostStagedDist._Work(lambda w=w, host=host: w.wait()))
            else:
                host = t.empty(op.tensor.shape, dtype=op.tensor.dtype, device="cpu")
                w = self._d.irecv(host, op.peer)

                def done(w=w, host=host, dst=op.tensor):
                    w.wait()
                    dst.copy_(host)
                works.append(HostStagedDist._Work(done))
        return works
