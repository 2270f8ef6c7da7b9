"""Row-block arithmetic of the multi-GPU split (pure Python mirror of sbx_split_* in sbx_capi.hip).

The frame shards as cyclic row-blocks: blocks of `block_rows` rows, rank r owns blocks r, r+N, ...
Equal contiguous strips do not balance (the bottom ~25 % of an APP_CLOUDS frame never marches,
/root/reference/src/app_clouds.h:212); 8-row cyclic blocks keep every rank
within ~1 % of the mean.  Pixels are computed from their GLOBAL row, so the assembled frame is
bit-identical to a single-GPU render.

Root relief: rank 0 is also the gather's root (it receives N-1 slabs and assembles the frame), so it can be dealt
fewer blocks: blocks go out in cycles of `rounds` rounds; a round gives one block to every rank, except that rank 0
is left out of the rounds >= root_rounds.  root_rounds = rounds = 1 is the plain cyclic split.
"""

DEFAULT_BLOCK_ROWS = 8


def num_blocks(height, block_rows):
    return (height + block_rows - 1) // block_rows


def cycle_blocks(nranks, root_rounds=1, rounds=1):
    return root_rounds * nranks + (rounds - root_rounds) * (nranks - 1)


def _rank_blocks(height, block_rows, rank, nranks, root_rounds, rounds):
    """Global block indices owned by `rank`, in slab order."""
    assert 0 <= root_rounds <= rounds and rounds >= 1 and (nranks > 1 or root_rounds == rounds)
    nb = num_blocks(height, block_rows)
    V = cycle_blocks(nranks, root_rounds, rounds)
    cnt = root_rounds if rank == 0 else rounds
    out = []
    cycle = 0
    while cycle * V < nb:
        for rnd in range(cnt):
            v = rnd * nranks + rank if rnd < root_rounds else root_rounds * nranks + (rnd - root_rounds) * (nranks - 1) + (rank - 1)
            b = cycle * V + v
            if b < nb:
                out.append(b)
        cycle += 1
    return out


def rank_rows(height, block_rows, rank, nranks, root_rounds=1, rounds=1):
    """Rows owned by `rank`."""
    return sum(min(block_rows, height - b * block_rows) for b in _rank_blocks(height, block_rows, rank, nranks, root_rounds, rounds))


def rank_rows_max(height, block_rows, nranks, root_rounds=1, rounds=1):
    """Rows every rank's slab holds so that an equal-count gather works (the fullest slab, whole blocks)."""
    mx = max(rank_rows(height, block_rows, r, nranks, root_rounds, rounds) for r in range(nranks))
    return ((mx + block_rows - 1) // block_rows) * block_rows


def rank_row_indices(height, block_rows, rank, nranks, root_rounds=1, rounds=1):
    """Global row index of each local row of `rank`, in slab order.  NOTE: a block that falls beyond the frame in the
    last cycle is skipped, but a rank's LATER blocks keep their slab position only if none before them was skipped —
    which holds because blocks of one rank are increasing and the frame ends once."""
    ys = []
    for b in _rank_blocks(height, block_rows, rank, nranks, root_rounds, rounds):
        y = b * block_rows
        ys.extend(range(y, min(y + block_rows, height)))
    return ys


def slab_source(height, block_rows, nranks, root_rounds=1, rounds=1):
    """For every global row y: (rank, local_row) where it lives after the gather."""
    V = cycle_blocks(nranks, root_rounds, rounds)
    out = []
    for y in range(height):
        blk, in_blk = divmod(y, block_rows)
        cycle, v = divmod(blk, V)
        if v < root_rounds * nranks:
            rnd, rank = divmod(v, nranks)
        else:
            q, r = divmod(v - root_rounds * nranks, nranks - 1)
            rnd, rank = root_rounds + q, 1 + r
        cnt = root_rounds if rank == 0 else rounds
        out.append((rank, (cycle * cnt + rnd) * block_rows + in_blk))
    return out


def relief_rounds(nranks, root_cost_ratio, rounds=8):
    """(root_rounds, rounds) that balance the root in the continuous model: every rank renders its share of a frame that
    costs T, the root additionally spends e = root_cost_ratio * T per frame on landing the peers' slabs and assembling.
    Equal finishing times need the root's share s0 = (1 - (N-1) e/T) / N; root_rounds is the nearest count to that share."""
    if nranks <= 1:
        return 1, 1
    s0 = max(0.0, (1.0 - (nranks - 1) * root_cost_ratio) / nranks)
    m0 = int(round(s0 * (nranks - 1) * rounds / (1.0 - s0))) if s0 < 1 else rounds
    return max(0, min(rounds, m0)), rounds


def best_relief(height, block_rows, nranks, root_cost_ratio, max_rounds=16):
    """The (root_rounds, rounds) with rounds <= max_rounds that minimises the modelled frame time
    max(rows_root / H + e/T, max_peer rows_peer / H), using the ACTUAL row counts of the split (partial last cycle
    included).  Ties go to the smaller cycle (finer interleave)."""
    if nranks <= 1:
        return 1, 1
    best = None
    for m in range(1, max_rounds + 1):
        for m0 in range(0, m + 1):
            rows = [rank_rows(height, block_rows, r, nranks, m0, m) for r in range(nranks)]
            cost = max(rows[0] / height + root_cost_ratio, max(rows[1:]) / height)
            key = (round(cost, 6), m)
            if best is None or key < best[0]:
                best = (key, (m0, m))
    return best[1]
