"""Multi-GPU frame: cyclic row-blocks per rank + ONE gather + root-side assembly.

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, xGMI underneath).  The path has
exactly one exchange step — bringing the rendered rows to the rank that owns the frame — so there is
exactly one collective per frame: a gather to rank 0.  Seven peers send over seven distinct xGMI links
(point-to-point, no ring), 16.6 MB each for a 3840x2160 RGBA32F frame.  Everything else is
embarrassingly parallel: every rank computes its pixels from their GLOBAL coordinates, so the
assembled frame is bit-identical to a single-GPU render.

`renderer` is duck-typed (render_rank / assemble / empty): shaderbox_amd.Renderer on GPUs; the CPU
tests drive the same code over gloo with an oracle-backed stand-in.
"""
from . import shard


class FramePlan:
    """Buffers of one rank for repeated frames of a fixed size."""

    def __init__(self, renderer, dist, width, height, block_rows=shard.DEFAULT_BLOCK_ROWS):
        self.r, self.dist = renderer, dist
        self.width, self.height, self.block_rows = int(width), int(height), int(block_rows)
        self.world = dist.get_world_size()
        self.rank = dist.get_rank()
        self.rows_max = shard.rank_rows_max(self.height, self.block_rows, self.world)
        self.slab = renderer.empty((self.rows_max, self.width, 4), zero=True)
        if self.rank == 0:
            self.gathered = renderer.empty((self.world, self.rows_max, self.width, 4))
            self.glist = [self.gathered[i] for i in range(self.world)]
            self.frame = renderer.empty((self.height, self.width, 4))
        else:
            self.gathered = self.glist = self.frame = None

    def render(self, app, time, mouse=(0.0, 0.0), aux=None):
        """All ranks call this; rank 0 returns the assembled [H, W, 4] frame, the others None."""
        self.r.render_rank(app, self.width, self.height, time, self.block_rows, self.rank, self.world,
                           mouse=mouse, aux=aux, out=self.slab)
        self.dist.gather(self.slab, self.glist, dst=0)        # the single collective of the path
        if self.rank == 0:
            return self.r.assemble(self.gathered, self.width, self.height, self.block_rows, self.world,
                                   out=self.frame)
        return None
