#!/usr/bin/env python
"""bench.py -- EGM grid-point x shock evals/s of the batched backward induction on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one full backward induction (terminal period + T-t0 EGM periods, envelopes included) of a
batch of `--ndraw` independent parameter draws of the workload, inputs (parameter vectors, quadrature)
resident in HBM when the timed region starts.  Weak scaling: every rank solves `--ndraw` draws of its own
(draws are independent); the only collective is the RCCL all-reduce of the per-draw
objective contributions after the timed steps.  Rank 0 prints ONE JSON line.

value       = evals the REFERENCE would execute for these draws (counted on device, equal to the oracle's
              count in the parity tests) * ranks / max-over-ranks wall time
roofline    = algorithmic table bytes per launch of the dominant kernel / its mean HIP-event duration,
              against the 8 TB/s HBM3E peak (the path is NOT bandwidth bound at these sizes,
              the fraction is reported honestly)
cpu_baseline= the CPU oracle (oracle/egdst_oracle.c, glibc math, gcc -O2, 1 thread) on a bounded sample
              of the same draws, timed in this run on the GPU box's host
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import egdst_amd  # noqa: E402,F401  (before torch starts the HIP runtime: sets GPU_MAX_HW_QUEUES, see egdst_amd/__init__.py)


def cpu_baseline(model, draws, budget_s=12.0):
    """Single-thread oracle (the 'port' of the reference CPU path) on a bounded sample of the draws."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from oracle_harness import Oracle
    orc = Oracle(model, native_math=True)
    t0 = time.perf_counter()
    evals, n = 0, 0
    for p in draws:
        sol = orc.solve(p)
        evals += sol.nevals
        n += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {'value': evals / dt, 'unit': 'evals/s', 'cores': 1, 'kind': 'port',
            'sample': '%d draws of the same workload, %.1f s, oracle/egdst_oracle.c gcc -O2 glibc math' % (n, dt),
            'wall_s_per_solve': dt / max(n, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--workload', default='C2')
    ap.add_argument('--ndraw', type=int, default=4096, help='parameter draws per GPU (weak scaling)')
    ap.add_argument('--rows-cap', type=int, default=0, help='compact physical row capacity (0 = exact, ngridmax rows)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-single-solve', action='store_true', help='skip the one-draw latency leg (profiling runs)')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from egdst_amd import build, runtime, workloads

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the egdst hot path has no CPU fallback')
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or 'TORCHELASTIC_RUN_ID' in os.environ   # under torchrun also with one rank
    if use_dist:
        # Rendezvous and the timing barriers go over gloo (host side).  The RCCL communicator for the path's one
        # collective is created AFTER the timed region: an initialised RCCL communicator holds hardware queues of its
        # own, which the 21 streams of the solver then have to share (measured with one rank: 346 -> 450 ms per step).
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('gloo')

    model, drawgen = workloads.WORKLOADS[args.workload]()
    lib = build.build_model(model)  # prebuilt in-tree by __graft_entry__.build(); rebuilds if stale
    desc = model.descriptor()
    ndraw = args.ndraw
    all_draws = drawgen(ndraw * world) if drawgen else np.tile(model.param_vector(), (ndraw * world, 1))
    mine = np.ascontiguousarray(all_draws[rank * ndraw:(rank + 1) * ndraw])

    stream = torch.cuda.Stream()
    solver = runtime.Solver(lib, desc, ndraw=ndraw, keep_history=False, stream=stream.cuda_stream, rows_cap=args.rows_cap)
    params_dev = torch.from_numpy(mine).cuda()          # inputs resident in HBM before the timed region
    obj = torch.zeros(ndraw, 2, dtype=torch.float64, device='cuda')
    torch.cuda.synchronize()
    solver.set_params_dev(params_dev.data_ptr())

    def sync_all():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        solver.solve_async()
    solver.sync(raise_on_error=False)
    solver.set_profile(True)

    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        solver.solve_async()
    solver.sync(raise_on_error=False)
    sync_all()
    dt = time.perf_counter() - t0

    status, _ = solver.status()
    evals_step, _ = solver.evals()
    kms, klaunch, algbytes = solver.profile()       # of the LAST step (events are re-recorded every step)
    # final objective reduce (the only collective of the path, RCCL over xGMI when world > 1)
    if solver.rows_cap:
        obj = torch.from_numpy(solver.objective()).cuda()   # includes the draws redone with exact capacities
    else:
        with torch.cuda.stream(stream):
            solver.objective_dev(obj.data_ptr())
        stream.synchronize()
    okmask = ~torch.isnan(obj[:, 0])
    red = torch.stack([torch.where(okmask, obj[:, 0], torch.zeros_like(obj[:, 0])).sum(), okmask.sum().double()])
    tt = torch.tensor([dt], dtype=torch.float64)
    ev = torch.tensor([float(evals_step)], dtype=torch.float64)
    if use_dist:
        rccl = dist.new_group(backend='nccl', device_id=torch.device('cuda', local_rank))
        dist.all_reduce(red, op=dist.ReduceOp.SUM, group=rccl)   # objective contributions: RCCL over xGMI
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)                # timing bookkeeping: host
        dist.all_reduce(ev, op=dist.ReduceOp.SUM)
    dt_max = float(tt.item())
    evals_all = float(ev.item())

    if rank == 0:
        ms_step = dt_max / args.steps * 1e3
        names = ['probe', 'grid', 'envelope']
        dom = int(np.argmax(kms))
        avg_launch_s = (kms[dom] / max(klaunch[dom], 1)) * 1e-3
        bytes_per_launch = algbytes / max(klaunch[dom], 1)
        achieved = bytes_per_launch / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        # single-solve latency (one draw) for the "full backward-induction wall time" half of the metric
        single_ms = None
        if not args.no_single_solve and world == 1:
            s1 = runtime.Solver(lib, desc, ndraw=1, keep_history=False)
            s1.set_params(mine[:1])
            s1.solve()
            t1 = time.perf_counter()
            for _ in range(3):
                s1.solve()
            single_ms = (time.perf_counter() - t1) / 3 * 1e3
            s1.close()
        # HBM traffic of the dominant kernel: PMC counters cannot be read from inside the process; the figure is
        # the one measured by `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` on this command and committed
        # under profiles/ (see profiles/README.md), used only when it was taken on the same configuration
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')))
            if (tj['workload'], tj['ndraw'], tj['rows_cap'], tj['kernel'], tj['launches_per_step']) == (
                    args.workload, ndraw, args.rows_cap, 'k_' + names[dom], int(klaunch[dom])):
                traffic = tj['fetch_bytes_per_launch'] + tj['write_bytes_per_launch']
        except (OSError, KeyError, ValueError):
            pass
        out = {
            'metric': 'EGM grid-point x shock evals/sec (batched backward induction, all draws, all periods)',
            'value': evals_all / (dt_max / args.steps), 'unit': 'evals/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': ms_step, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': '%s: %s, T=%d, ngridm=%d, ny=%d, nd=%d, nst=%d, a0=%g, mmax=%g' % (
                args.workload, model.label, desc['T'], desc['ngridm'], desc['ny'], lib.info.nd, lib.info.nst,
                desc['a0'], desc['mmax']), 'ndraw_per_gpu': ndraw, 'ndraw_total': ndraw * world,
                'rows_cap': args.rows_cap,
                'parallelism': 'draws sharded over %d rank(s), no data-path collective' % world},
            'evals_per_step': evals_all, 'failed_draws_rank0': int((status != 0).sum()),
            'single_solve_ms': single_ms, 'capacity_retries': solver.capacity_retries,
            'schedule': dict(zip(('groups', 'straggler_lanes', 'straggler_draws'), solver.schedule())),
            'objective_mean': float(red[0].item() / max(red[1].item(), 1.0)),
            'kernel_ms_per_step': {n: float(v) for n, v in zip(names, kms)},
            'roofline': {'bound': 'hbm', 'kernel': 'k_' + names[dom], 'achieved': achieved, 'peak': 8000.0,
                         'unit': 'GB/s', 'frac': achieved / 8000.0, 'traffic': traffic,
                         'algorithmic_bytes_per_launch': bytes_per_launch, 'avg_launch_ms': avg_launch_s * 1e3,
                         'launches': int(klaunch[dom])},
        }
        if not args.no_cpu_baseline and world == 1:   # reported baseline: rank 0 at N=1 only
            out['cpu_baseline'] = cpu_baseline(model, mine)
            out['speedup_vs_cpu_1thread'] = out['value'] / out['cpu_baseline']['value']
            if single_ms:
                out['single_solve_speedup_vs_cpu'] = out['cpu_baseline']['wall_s_per_solve'] * 1e3 / single_ms
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
