"""N>1 path on CPU: two gloo ranks shard the draws, 'solve' them with the CPU oracle as a stand-in for the
device solve (the sharding/reduce plumbing is what is under test), and reduce the objective."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from egdst_amd import parallel

HERE = os.path.dirname(os.path.abspath(__file__))


def test_shard_bounds_cover_all_draws():
    for nd in (1, 7, 16, 1024):
        for w in (1, 2, 3, 8):
            b = [parallel.shard_bounds(nd, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == nd
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def test_strong_scaling_plan_covers_the_job_exactly_once():
    """bench.py --scaling strong: the job's draws are split over the ranks (shard_bounds) and every rank walks its shard
    in chunks through one handle (plan_chunks): every draw of the job is counted exactly once, whatever the sizes."""
    for job, world, chunk in ((1024, 8, 128), (1024, 1, 128), (256, 8, 32), (1000, 3, 128), (5, 8, 128), (64, 2, 16), (130, 1, 128)):
        seen = np.zeros(job, dtype=int)
        for r in range(world):
            lo, hi = parallel.shard_bounds(job, world, r)
            plan = parallel.plan_chunks(hi - lo, chunk)
            assert sum(v for _, v in plan) == hi - lo
            for c0, valid in plan:
                assert 0 < valid <= max(1, min(chunk, hi - lo))
                seen[lo + c0:lo + c0 + valid] += 1
        assert np.all(seen == 1), (job, world, chunk)
    assert parallel.plan_chunks(0, 128) == []


def _worker(rank, world, port, ndraw, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from egdst_amd import workloads
    from oracle_harness import Oracle
    m, gen = workloads.c2(a0=0, ngridm=40, T=6, ny=3)
    P = gen(ndraw)
    lo, hi = parallel.shard_bounds(ndraw, world, rank)
    orc = Oracle(m)
    vals = []
    for p in P[lo:hi]:
        s = orc.solve(p)
        vals.append(s.V[0, 0, 1] if s.rc == 0 else float('nan'))
    local = torch.tensor(vals, dtype=torch.float64)
    tot, cnt = parallel.reduce_objective(local)
    allv = parallel.gather_draw_results(local, ndraw)
    if rank == 0:
        q.put((tot, cnt, allv.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_reduce_equals_single_process():
    sys.path.insert(0, HERE)
    from egdst_amd import workloads
    from oracle_harness import Oracle
    ndraw = 5
    m, gen = workloads.c2(a0=0, ngridm=40, T=6, ny=3)
    orc = Oracle(m)
    ref = np.array([orc.solve(p).V[0, 0, 1] for p in gen(ndraw)])
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ndraw, q)) for r in range(2)]
    for p in procs:
        p.start()
    tot, cnt, allv = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert cnt == ndraw and abs(tot - ref.sum()) <= 1e-12 * abs(ref.sum())
    assert np.array_equal(allv, ref)


def test_cost_aware_shards_partition_the_job_and_balance_the_recorded_costs():
    """parallel.shard_indices: every draw to exactly one rank, no rank above ceil(n / world) draws, the same partition on every rank
    (it is a pure function of its arguments), and balanced COST: on the per-draw figures recorded on an MI355X for the north_star's
    batches (tests/golden/draw_costs_*.npz, make_draw_costs.py: evaluations and re-basing calls of the 1024 C5 and 256 C4 draws; 62
    of the C5 draws fail and cost a sixteenth of the others) the most loaded of 8 ranks carries at most 1 % more than the mean --
    SURVEY.md section 8(e)'s "load balance matters more than comms", the bar of the round-3 verdict being 1.1 -- and on a
    heavy-tailed synthetic cost (a tenth of the draws ten times dearer, in a block: what contiguous shards handle worst) it stays
    within 2 % where contiguous shards are off by more than half."""
    rng = np.random.default_rng(5)
    for n, w in ((1024, 8), (1000, 3), (7, 8), (256, 2)):
        c = rng.lognormal(0, 1, n)
        parts = [parallel.shard_indices(n, w, r, c) for r in range(w)]
        assert sorted(np.concatenate(parts).tolist()) == list(range(n))
        assert max(len(p) for p in parts) <= -(-n // w)
        assert all(np.all(np.diff(p) > 0) for p in parts if len(p) > 1)
        assert all(np.array_equal(p, parallel.shard_indices(n, w, r, c.copy())) for r, p in enumerate(parts))
        inter = [parallel.shard_indices(n, w, r) for r in range(w)]
        assert sorted(np.concatenate(inter).tolist()) == list(range(n)) and all(np.array_equal(p, np.arange(r, n, w)) for r, p in enumerate(inter))
    for wl, n in (('C5', 1024), ('C4', 256)):
        d = np.load(os.path.join(HERE, 'golden', 'draw_costs_%s.npz' % wl))
        c = parallel.draw_cost(d['evals'], d['work'])
        assert len(c) == n
        for w in (2, 4, 8):
            assert parallel.shard_balance(c, w) <= 1.01, (wl, w, parallel.shard_balance(c, w))
    c = np.ones(1024)
    c[300:400] = 10.0
    contiguous = max(c[slice(*parallel.shard_bounds(1024, 8, r))].sum() for r in range(8)) / (c.sum() / 8)
    assert contiguous > 1.5 and parallel.shard_balance(c, 8) <= 1.02 and parallel.shard_balance(c, 8, by_cost=False) <= 1.05
    with pytest.raises(ValueError):
        parallel.shard_indices(8, 2, 0, np.ones(7))
