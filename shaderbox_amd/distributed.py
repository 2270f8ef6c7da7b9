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

`renderer` is duck-typed (render_rank_rows / render_rank_in_place / assemble / assemble_peers / empty):
shaderbox_amd.Renderer on GPUs; the CPU tests drive the same code over gloo with an oracle-backed stand-in.
"""
from . import shard


class FramePlan:
    """Buffers and schedule of one rank for repeated frames of a fixed size."""

    def __init__(self, renderer, dist, width, height, block_rows=shard.DEFAULT_BLOCK_ROWS, groups=1, root_rounds=1,
                 rounds=1, exchange="direct", channels=3):
        if exchange not in ("direct", "gather"):
            raise ValueError("exchange must be 'direct' or 'gather'")
        if exchange == "gather":
            channels = 4
        if channels not in (3, 4):
            raise ValueError("channels must be 3 or 4")
        self.r, self.dist = renderer, dist
        self.exchange, self.channels = exchange, int(channels)
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
        self.slab = self.gathered = self.peers = self.frame = None
        self.glists = [None] * len(self.ranges)
        if exchange == "gather":
            self.slab = renderer.empty((self.rows_max, self.width, 4), zero=True)
            if self.rank == 0:
                self.gathered = renderer.empty((self.world, self.rows_max, self.width, 4))
                self.glists = [[self.gathered[i, a:b] for i in range(self.world)] for a, b in self.ranges]
                self.frame = renderer.empty((self.height, self.width, 4))
        elif self.rank == 0:
            self.frame = renderer.empty((self.height, self.width, 4))
            if self.world > 1:
                self.peers = renderer.empty((self.world - 1, self.rows_max, self.width, self.channels), zero=True)
        else:
            self.slab = renderer.empty((self.rows_max, self.width, self.channels), zero=True)
        # the point-to-point descriptors of the direct exchange, built once (they only name fixed buffers and peers)
        self.p2p = []
        if exchange == "direct" and self.world > 1:
            for a, b in self.ranges:
                if self.rank == 0:
                    self.p2p.append([dist.P2POp(dist.irecv, self.peers[i - 1, a:b], i) for i in range(1, self.world)])
                else:
                    self.p2p.append([dist.P2POp(dist.isend, self.slab[a:b], 0)])

    def render(self, app, time, mouse=(0.0, 0.0), aux=None, mark=None):
        """All ranks call this; rank 0 returns the assembled [H, W, 4] frame, the others None.  `mark(name)`, if given, is
        called after the rank's rendering ("render"), after the waits of its exchange ("exchange") and after the root's
        assembly ("assemble"): bench.py records stream events there to time the phases of a frame."""
        mark = mark or (lambda name: None)
        if self.exchange == "gather":
            return self._render_gather(app, time, mouse, aux, mark)
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
