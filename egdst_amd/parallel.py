"""Sharding of parameter draws over ranks and the final objective reduce.

Draws are independent full solves, so the path shards with NO data-path collective: rank r solves the
contiguous block of draws [r*per, (r+1)*per) (the last ranks take one draw fewer when ndraw % world != 0).
The only collective is the reduce of the per-draw objective contributions at the end: `nccl` (= RCCL over
xGMI) on GPUs, `gloo` in the CPU tests.
"""
from __future__ import annotations

import numpy as np


def shard_bounds(ndraw, world, rank):
    """[lo, hi) of the draws of `rank`; contiguous, sizes differ by at most one."""
    base, extra = divmod(int(ndraw), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def plan_chunks(ndraw_rank, chunk):
    """How a rank walks its shard through ONE handle of `chunk` draws (bench.py --scaling strong): a list of
    (first draw of the shard, draws that count) per chunk; the last chunk may be short -- the handle is then filled up
    with copies of the shard's first draw, which are solved but not counted."""
    ndraw_rank, chunk = int(ndraw_rank), int(chunk)
    if ndraw_rank <= 0:
        return []
    chunk = max(1, min(chunk, ndraw_rank))
    return [(c, min(chunk, ndraw_rank - c)) for c in range(0, ndraw_rank, chunk)]


def reduce_objective(local_obj, ok_mask=None, group=None):
    """Sum and count of the finite per-draw objective contributions over all ranks.

    local_obj: 1-D torch tensor (on the device of the backend).  Returns (sum, count) as python floats.
    """
    import torch
    import torch.distributed as dist
    ok = ~torch.isnan(local_obj) if ok_mask is None else ok_mask
    red = torch.stack([torch.where(ok, local_obj, torch.zeros_like(local_obj)).sum(),
                       ok.sum().to(local_obj.dtype)])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(red, op=dist.ReduceOp.SUM, group=group)
    return float(red[0].item()), float(red[1].item())


def gather_draw_results(local_vals, ndraw, group=None):
    """All-gather per-draw scalars (e.g. objective per draw) into draw order on every rank."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local_vals.clone()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = [shard_bounds(ndraw, world, r)[1] - shard_bounds(ndraw, world, r)[0] for r in range(world)]
    mx = max(sizes)
    pad = torch.full((mx,), float('nan'), dtype=local_vals.dtype, device=local_vals.device)
    pad[:local_vals.numel()] = local_vals
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([b[:n] for b, n in zip(bufs, sizes)])
