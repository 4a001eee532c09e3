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


def shard_indices(ndraw, world, rank, cost=None):
    """The draws of `rank` as an index array, balanced by COST rather than by count ("load balance matters
    more than comms": the cost of a draw varies several-fold with the envelope work its parameters cause, and a draw the reference
    algorithm fails on stops early).

    cost: per-draw cost of an earlier solve of nearby parameters -- an estimation loop moves them slowly, so the last iteration's
    figures order the next one's (`draw_cost` below builds it from a handle's counters) -- or None.  With costs, the longest-
    processing-time rule: draws in order of decreasing cost, each to the rank with the least cost so far (ties: fewer draws, then
    the lower rank), so the shard sizes stay within one draw of each other only by accident but the shard COSTS within the largest
    single draw's.  Without costs: the interleave `d mod world` SURVEY.md offers, which spreads any smooth dependence of the cost on
    the draw index.  Either way every rank computes the same partition from the same arguments (deterministic; no communication),
    and the indices of a rank are ascending."""
    ndraw, world, rank = int(ndraw), int(world), int(rank)
    if cost is None:
        return np.arange(rank, ndraw, world, dtype=np.int64)
    cost = np.asarray(cost, dtype=np.float64)
    if cost.shape != (ndraw,):
        raise ValueError('cost must have one entry per draw')
    cost = np.where(np.isfinite(cost) & (cost > 0), cost, 0.0)
    order = np.argsort(-cost, kind='stable')          # (stable: equal costs keep the draw order -- the same on every rank)
    load = np.zeros(world)
    count = np.zeros(world, dtype=np.int64)
    cap = -(-ndraw // world)                          # a handle holds whole chunks: no rank takes more than ceil(ndraw / world) draws
    owner = np.empty(ndraw, dtype=np.int64)
    for d in order:
        free = np.nonzero(count < cap)[0]
        r = free[np.lexsort((free, count[free], load[free]))[0]]
        owner[d] = r
        load[r] += cost[d]
        count[r] += 1
    return np.nonzero(owner == rank)[0].astype(np.int64)


def draw_cost(evals, work=None, failed=None):
    """Cost figure of every draw from a handle's counters after a solve: the EGM evaluations the draw caused (egdst_get_evals: the
    grid kernel's time, the dominant kernel of the stress configurations, is proportional to them, and so is the number of candidate
    points its envelopes sort and walk) plus 32 evaluations' worth per re-basing call of its guess streams (egdst_get_work: strictly
    sequential calls); a failed draw keeps the count it reached before it stopped."""
    c = np.asarray(evals, dtype=np.float64).copy()
    if work is not None:
        c += 32.0 * np.asarray(work, dtype=np.float64)
    return c


def shard_balance(cost, world, by_cost=True):
    """(max shard cost / mean shard cost) of the partition `shard_indices` makes -- 1.0 is perfect"""
    cost = np.asarray(cost, dtype=np.float64)
    tot = [cost[shard_indices(len(cost), world, r, cost if by_cost else None)].sum() for r in range(world)]
    return max(tot) / (sum(tot) / world)


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
