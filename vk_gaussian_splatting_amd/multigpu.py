"""Screen-strip partition across the GPUs of one node + all-gather of the strips.

No reference counterpart (the reference is single-GPU).  One process per GPU; every rank holds the
full (replicated) splat buffers, renders tile rows [row_begin, row_end) of the frame with the whole
path (key/cull -> sort -> project -> bin -> composite) and the strips are exchanged with ONE
collective (`all_gather_into_tensor`; backend "nccl" == RCCL over xGMI on the GPU box, "gloo" in
the CPU tests).  Because every rank applies the same full-frame cull and the same keys, a pixel sees
the same splats in the same order as in the single-GPU frame: the gathered frame is bit-identical.
"""
TILE = 16


def tile_rows(height):
    return (height + TILE - 1) // TILE


def strip_rows(height, world_size, rank):
    """tile-row range [begin, end) of `rank`; equal-sized strips (the last ones may be short/empty)."""
    rows = tile_rows(height)
    per = (rows + world_size - 1) // world_size
    b = min(rank * per, rows)
    e = min(b + per, rows)
    return b, e


def strip_pixel_rows(height, world_size):
    """padded strip height in pixels (identical on every rank, as all_gather needs)"""
    rows = tile_rows(height)
    per = (rows + world_size - 1) // world_size
    return per * TILE


def gather_strips(strip, world_size, group=None):
    """strip: tensor [strip_pixel_rows, width, 4] (this rank's rows, padded).  Returns the tensor
    [world_size*strip_pixel_rows, width, 4]; the frame is its first `height` rows."""
    import torch
    import torch.distributed as dist
    out = torch.empty((world_size * strip.shape[0],) + tuple(strip.shape[1:]), dtype=strip.dtype, device=strip.device)
    if world_size == 1:
        out.copy_(strip)
        return out
    dist.all_gather_into_tensor(out, strip.contiguous(), group=group)
    return out


def balanced_bounds(row_cost, world_size, floor_frac=0.25):
    """tile-row boundaries [world_size + 1] that equalise the per-row cost (mgs_frame_row_costs, averaged over the
    poses of interest) across ranks.  Every row is charged at least floor_frac of the mean row cost: rows with no list
    entries still cost their pixels and the fixed part of every launch.  Deterministic, so every rank that feeds it the
    same costs gets the same table."""
    import numpy as np
    c = np.asarray(row_cost, np.float64)
    rows = c.size
    c = c + floor_frac * max(c.mean(), 1.0)
    cum = np.concatenate([[0.0], np.cumsum(c)])
    bounds = [0]
    for r in range(1, world_size):
        target = cum[-1] * r / world_size
        b = int(np.searchsorted(cum, target, side="left"))
        # the nearer of the two candidate cuts
        if b > 0 and abs(cum[b - 1] - target) <= abs(cum[min(b, rows)] - target):
            b -= 1
        b = max(b, bounds[-1] + (1 if rows - bounds[-1] > world_size - r else 0))
        bounds.append(min(b, rows))
    bounds.append(rows)
    return [int(b) for b in bounds]


def padded_strip_rows(bounds):
    """pixel rows of the tallest strip of a boundary table: every rank's all_gather buffer has this height"""
    return max(bounds[r + 1] - bounds[r] for r in range(len(bounds) - 1)) * TILE


def assemble(gathered, bounds, height):
    """gathered: [world * padded_strip_rows, width, 4] from gather_strips with strips cut by `bounds` (unequal strips
    are padded to the tallest) -> the frame [height, width, 4]"""
    import torch
    world = len(bounds) - 1
    pad = padded_strip_rows(bounds)
    rows = [gathered[r * pad: r * pad + max(0, min(bounds[r + 1] * TILE, height) - bounds[r] * TILE)] for r in range(world)]
    return torch.cat(rows, 0)
