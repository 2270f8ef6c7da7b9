"""Row-block arithmetic of the multi-GPU split (pure Python mirror of sbx_rank_rows* in sbx_capi.hip).

The frame shards as cyclic row-blocks: blocks of `block_rows` rows, rank r owns blocks r, r+N, ...
Equal contiguous strips do not balance (the bottom ~25 % of an APP_CLOUDS frame never marches,
/root/reference/src/app_clouds.h:212); 8-row cyclic blocks keep every rank
within ~1 % of the mean.  Pixels are computed from their GLOBAL row, so the assembled frame is
bit-identical to a single-GPU render.
"""

DEFAULT_BLOCK_ROWS = 8


def num_blocks(height, block_rows):
    return (height + block_rows - 1) // block_rows


def rank_rows(height, block_rows, rank, nranks):
    """Rows owned by `rank`."""
    rows = 0
    for b in range(rank, num_blocks(height, block_rows), nranks):
        y = b * block_rows
        rows += min(block_rows, height - y)
    return rows


def rank_rows_max(height, block_rows, nranks):
    """Rows every rank's slab holds so that an equal-count gather works."""
    nb = num_blocks(height, block_rows)
    return ((nb + nranks - 1) // nranks) * block_rows


def rank_row_indices(height, block_rows, rank, nranks):
    """Global row index of each local row of `rank`, in slab order."""
    ys = []
    for b in range(rank, num_blocks(height, block_rows), nranks):
        y = b * block_rows
        ys.extend(range(y, min(y + block_rows, height)))
    return ys


def slab_source(height, block_rows, nranks):
    """For every global row y: (rank, local_row) where it lives after the gather."""
    out = []
    for y in range(height):
        blk, in_blk = divmod(y, block_rows)
        out.append((blk % nranks, (blk // nranks) * block_rows + in_blk))
    return out
