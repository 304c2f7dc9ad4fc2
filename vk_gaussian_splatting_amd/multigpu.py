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
