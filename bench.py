#!/usr/bin/env python
"""bench.py -- EGM grid-point x shock evals/s of the batched backward induction on MI355X.

    python bench.py --gpus N --steps K --warmup W                       (N > 1 without torchrun: this script starts the ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one full backward induction (terminal period + T-t0 EGM periods, envelopes included) of a batch of
independent parameter draws of the workload, parameter vectors and quadrature resident in HBM when the timed region
starts.  The draws move a little from step to step (each parameter times 1 + 0.5 % U(-1,1), as the iterations of an
estimation loop would move them), so the handle's history-based straggler schedule is never trained on the very draws
it then meets (--no-perturb: the same draws every step).

  --scaling weak   (default) every rank solves --ndraw draws per step (BASELINE configs[1]: C2, 4096 draws per GPU)
  --scaling strong the job is --ndraw-total draws per step (north_star: the 1024-draw C5 batch, 256 draws of C4), rank r
                   takes the contiguous shard egdst_amd.parallel.shard_bounds gives it and loops chunks of --chunk draws
                   that fit HBM through ONE handle
Draws are independent: no data-path collective; after the timed steps the per-draw objective
contributions are reduced with one RCCL all-reduce.  Rank 0 prints ONE JSON line.

value       = evaluations EXECUTED for the draws that solved (failed draws and the evaluations the device credits
              without executing -- include/egdst.h egdst_get_evals_credited -- are left out), all ranks, all timed
              steps / max-over-ranks wall time.  `evals_reference_per_step` is what the reference would count.
roofline    = for the dominant kernel (largest total duration in the committed rocprofv3 trace of this configuration,
              profiles/r04_kernel_stats_<key>.csv): the algorithmic table bytes of the cells one launch handles / its mean
              HIP-event duration in one more solve of the last step's draws with ONE draw group (events on the launching stream,
              the kernel alone on the GPU; `timed_step` holds the queue-inclusive figure of the 16-stream solve), against the
              8 TB/s HBM3E peak; `traffic`, `valu_util`, `valu_fp64_util`, `lds_GBps` from the rocprofv3 --pmc passes committed
              under profiles/ for this very configuration (counters cannot be read from inside the process); `step` = the
              whole step: algorithmic bytes / ms_per_step, and the summed counter traffic of all its kernels.  The path is NOT
              HBM-bound at these sizes (SURVEY.md section 8d says so): the fraction is reported as it is, and `ceilings` says
              what does bind:
              `egm_only_evals_per_s` = the evaluations of the step / the summed device time of the grid kernel alone (what
              the step would deliver if the EGM evaluations were all there is), `pipeline_frac_of_egm_only` = value / that,
              and the issue-slot figures of the grid kernel from the counter passes.
legs        = (N = 1) one warm-up and one timed step each of the per-GPU shares of the stress configurations BASELINE.json
              names (C4 x 32 draws, C5 x 128 draws at full size) and of C2 with a0 = 0 (the headline of rounds 1-3; the headline is
              now C2 on the surveyed credit limit a0 = -5), each with its own roofline; (N > 1) `strong`: one timed step of the north_star's batches
              (C5 x 1024, C4 x 256 draws) sharded over the N ranks.
cpu_baseline= the CPU oracle (oracle/egdst_oracle.c, glibc math, gcc -O2) on a bounded sample of the same draws, timed in
              this run on the GPU box's host: one thread (`value`, `cores` = 1) and every core this process may use
              (scheduler affinity and cgroup quota), one process per core.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import egdst_amd  # noqa: E402,F401  (before torch starts the HIP runtime: sets GPU_MAX_HW_QUEUES, see egdst_amd/__init__.py)

DEFAULT_CHUNK = {'C1': 4096, 'C2': 4096, 'C3': 256, 'C4': 32, 'C5': 128}   # draws per handle that fit 288 GB comfortably


def _oracle_worker(args):
    """(module-level for multiprocessing) solve draws with the oracle until the budget is spent; returns (evals, n, seconds)"""
    wl, kw, draws, budget_s = args
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from egdst_amd import workloads
    from oracle_harness import Oracle
    model = workloads.WORKLOADS[wl](**kw)[0]
    orc = Oracle(model, native_math=True)
    t0 = time.perf_counter()
    evals = n = 0
    for p in draws:
        evals += orc.solve(p).nevals
        n += 1
        if time.perf_counter() - t0 > budget_s:
            break
    return evals, n, time.perf_counter() - t0


def cpu_model_name():
    try:
        for ln in open('/proc/cpuinfo'):
            if ln.startswith('model name'):
                return ln.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def usable_cores():
    """cores this process may really use: scheduler affinity, capped by the cgroup CPU quota when there is one"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            txt = open(path).read().split()
            if path.endswith('cpu.max'):
                if txt[0] != 'max':
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(wl, kw, draws, budget_s=10.0, cores=None):
    """Oracle (the 'port' of the reference CPU path) on a bounded sample: one thread, then one process per core."""
    import multiprocessing as mp
    ev, n, dt = _oracle_worker((wl, kw, draws, budget_s))
    out = {'value': ev / dt, 'unit': 'evals/s', 'cores': 1, 'kind': 'port',
           'sample': '%d draws of the same workload, %.1f s, oracle/egdst_oracle.c gcc -O2 glibc math' % (n, dt),
           'wall_s_per_solve': dt / max(n, 1), 'cpu_model': cpu_model_name(), 'host_cores': os.cpu_count()}
    cores = cores or max(1, min(usable_cores(), len(draws)))
    out['usable_cores'] = usable_cores()
    if cores > 1:
        parts = [draws[i::cores] for i in range(cores)]
        t0 = time.perf_counter()
        with mp.get_context('spawn').Pool(cores) as pool:
            res = pool.map(_oracle_worker, [(wl, kw, p, budget_s * 0.6) for p in parts])
        wall = time.perf_counter() - t0
        busy = max(r[2] for r in res)
        out['all_cores'] = {'value': sum(r[0] for r in res) / busy, 'unit': 'evals/s', 'cores': cores,
                            'sample': '%d draws, one process per core, %.1f s of solving (%.1f s with start-up)' % (
                                sum(r[1] for r in res), busy, wall)}
    return out


def spawn_ranks(n):
    """`python bench.py --gpus N` without torchrun: start the ranks as a child process BEFORE anything touches the GPU."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    return subprocess.call(cmd, env=env)


CLASS_NAMES = ['probe', 'grid', 'k_envelope', 'regeneration', 'tp_prep', 'tp_sort0', 'tp_sort1', 'tp_walk0', 'tp_walk1']
# the kernels of a period (profile classes of each; the two stages of k_tp_sort / k_tp_walk are one kernel) and what a launch of
# theirs covers: 'cells' = every cell of the group (dense kernels), 'left' = the cells the throughput path left to k_envelope,
# 'regen' = the guess streams k_fixup regenerated
KERNELS = [('k_probe', ['k_probe'], [0], 'cells'),
           ('grid', ['k_grid_lds_cv', 'k_grid_lds', 'k_grid_wide', 'k_grid'], [1], 'cells'),
           ('k_envelope', ['k_env1', 'k_envelope'], [2], 'left'),   # (single-choice models: k_env1 does the regular cells of the class)
           ('k_fixup', ['k_fixup'], [3], 'regen'),
           ('k_tp_prep', ['k_tp_prep'], [4], 'cells'),
           ('k_tp_sort', ['k_tp_sort', 'k_tp_sort_big'], [5, 6], 'cells'),
           ('k_tp_walk', ['k_tp_walk', 'k_tp_big'], [7, 8], 'cells'),
           ('k_sortcheck', ['k_sortcheck'], [], 'cells')]   # (no profile class of its own: launched once per period like k_probe)
TRACE_ROUND = 'r04'


def trace_stats(key):
    """{kernel base name: (calls, total ns, average ns)} from the committed rocprofv3 --kernel-trace --stats summary of this
    configuration (profiles/<round>_kernel_stats_<key>.csv), or {}"""
    import csv
    f = os.path.join(ROOT, 'profiles', '%s_kernel_stats_%s.csv' % (TRACE_ROUND, key))
    out = {}
    try:
        for row in csv.DictReader(open(f)):
            out[row['Name'].split('(')[0]] = (int(row['Calls']), float(row['TotalDurationNs']), float(row['AverageNs']))
    except (OSError, KeyError, ValueError):
        return {}, None
    return out, os.path.relpath(f, ROOT)


def roofline_record(key, kms, klaunch, algbytes, evals_step, value, ms_step, cells, left, regen, small=False, serial_ms=None, serial_launch=None):
    """The `roofline` object of one configuration.

    key: '<workload>_ndraw<N>' -- names the committed trace (profiles/r04_kernel_stats_<key>.csv) and counters (pmc_metrics.json).
    Dominant kernel: the one with the largest total duration in that trace (deterministic: a committed file), else in the one-group
    HIP-event profile of this run.  Its `achieved` = the algorithmic bytes of the cells ONE LAUNCH of it handles / its average launch
    duration, both taken from the one-group solve of this run (every kernel with the GPU to itself: HIP events on the launching
    stream bracket the kernel, not its wait behind fifteen other groups); `timed_step` holds the queue-inclusive figure of the
    16-stream step beside it and `trace_avg_launch_ms` the committed trace's.  `step` is the whole step against the same peak."""
    tr, tr_file = ({}, None) if small else trace_stats(key)
    pmc_cfg = {}
    if not small:
        try:
            pmc_cfg = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_metrics.json'))).get(key, {})
        except (OSError, ValueError):
            pmc_cfg = {}
    per = {}
    for name, knames, classes, covers in KERNELS:
        ms16, n16 = sum(kms[c] for c in classes), sum(int(klaunch[c]) for c in classes)
        ms1 = sum(serial_ms[c] for c in classes) if serial_ms is not None else None
        n1 = sum(int(serial_launch[c]) for c in classes) if serial_launch is not None else 0
        tcalls = sum(tr[k][0] for k in knames if k in tr)
        tns = sum(tr[k][1] for k in knames if k in tr)
        pm = next((pmc_cfg[k] for k in knames if k in pmc_cfg), {})
        kname = next((k for k in knames if k in tr or k in pmc_cfg), knames[0])
        per[name] = dict(kernel=kname, knames=knames, covers=covers, ms16=ms16, n16=n16, ms1=ms1, n1=n1, trace_calls=tcalls, trace_ns=tns, pm=pm)
    if any(v['trace_ns'] for v in per.values()):
        dom, dom_by = max(per, key=lambda k: per[k]['trace_ns']), 'largest total duration in ' + tr_file
    elif serial_ms is not None:
        dom, dom_by = max(per, key=lambda k: per[k]['ms1'] or 0.0), 'largest device time in the one-group solve of this run (no committed trace of this configuration)'
    else:
        dom, dom_by = max(per, key=lambda k: per[k]['ms16']), 'largest summed HIP-event time of the last step'
    d = per[dom]
    # share of the step's algorithmic bytes the dominant kernel's launches handle: all cells (dense kernels), or the cells / streams
    # it was actually given
    share = {'cells': 1.0, 'left': left / max(cells, 1), 'regen': regen / max(2 * cells, 1)}[d['covers']]
    bytes_dom = algbytes * share

    def rate(ms, n):
        if not ms or not n:
            return None, None, None
        avg = ms / n
        ach = bytes_dom / n / (avg * 1e-3) / 1e9
        return avg, bytes_dom / n, ach
    avg1, bpl1, ach1 = rate(d['ms1'], d['n1'])
    avg16, bpl16, ach16 = rate(d['ms16'], d['n16'])
    achieved, avg, bpl = (ach1, avg1, bpl1) if ach1 is not None else (ach16, avg16, bpl16)
    pm = d['pm']
    # whole step: algorithmic bytes and, from the counter passes, the HBM-side traffic of all kernels of a step
    traffic_step = None
    if pmc_cfg and any(v['pm'] for v in per.values()):
        traffic_step = 0.0
        for name, v in per.items():
            for k in v['knames']:   # every kernel of the class with its own counters (k_env1 and k_envelope; k_tp_walk and k_tp_big)
                if k in pmc_cfg:
                    n = pmc_cfg[k]['dispatches'] / max(pmc_cfg.get('k_probe', {}).get('dispatches', 0), 1) * per['k_probe']['n16']
                    traffic_step += pmc_cfg[k]['hbm_bytes_per_launch'] * n
    grid_alone = float(serial_ms[1]) if serial_ms is not None else None
    egm_only = evals_step / (grid_alone * 1e-3) if grid_alone else None
    pg = per['grid']['pm']
    dense_sum = sum(float(serial_ms[c]) for c in (0, 1, 4, 5, 6, 7, 8)) if serial_ms is not None else None
    return {'bound': 'hbm', 'bound_note': 'reported against HBM as the contract asks; the path is latency / issue bound, see ceilings',
            'kernel': d['kernel'], 'kernel_class': dom, 'dominant_by': dom_by,
            'achieved': achieved, 'peak': 8000.0, 'unit': 'GB/s', 'frac': (achieved / 8000.0) if achieved is not None else None,
            # the counter passes ran the 16-group schedule (a launch = one group's cells); `achieved` is per ONE-GROUP launch (all cells):
            # the same units, i.e. the counters' bytes per launch x the launches of the 16-group step per one-group launch
            'traffic': (pm['hbm_bytes_per_launch'] * (d['n16'] / d['n1'] if (ach1 is not None and d['n1']) else 1.0)) if pm.get('hbm_bytes_per_launch') else None,
            'traffic_per_launch_in_counter_pass': pm.get('hbm_bytes_per_launch'),
            'algorithmic_bytes_per_launch': bpl, 'avg_launch_ms': avg, 'launches': int(d['n1'] or d['n16']),
            'measured': 'HIP events around the launches of one more solve of the same draws with ONE draw group (the kernel has the GPU to itself), right after the timed region',
            'share_of_step_bytes': share,
            'timed_step': {'note': 'the same kernel in the 16-stream step after the timed region: HIP events on the group streams, the wait in the hardware queue included',
                           'avg_launch_ms': avg16, 'launches': int(d['n16']), 'achieved': ach16},
            'trace_avg_launch_ms': (d['trace_ns'] / d['trace_calls'] * 1e-6) if d['trace_calls'] else None,
            'valu_util': pm.get('valu_util'), 'valu_fp64_util': pm.get('valu_fp64_util'), 'lds_GBps': pm.get('lds_GBps'),
            'wave_cycles_waiting_frac': pm.get('wave_cycles_waiting_frac'), 'counters_from': pm.get('source'),
            'step': {'algorithmic_bytes': float(algbytes), 'ms_per_step': ms_step, 'achieved': algbytes / (ms_step * 1e-3) / 1e9,
                     'frac': algbytes / (ms_step * 1e-3) / 1e9 / 8000.0, 'traffic_bytes': traffic_step,
                     'traffic_over_algorithmic': (traffic_step / algbytes) if (traffic_step and algbytes) else None,
                     'traffic_GBps': (traffic_step / (ms_step * 1e-3) / 1e9) if traffic_step else None, 'unit': 'GB/s',
                     'note': 'traffic_bytes = sum over the kernels of (FETCH_SIZE + WRITE_SIZE per launch in the committed counter passes) x their launches per k_probe launch there x this run\'s k_probe launches'},
            'ceilings': {'note': 'egm_only: the evaluations of a step / the device time of the grid kernel when it has the GPU to itself '
                                 '(one draw group, no concurrent streams); kernel_ms_one_group: every class measured that way',
                         'egm_only_evals_per_s': egm_only,
                         'pipeline_frac_of_egm_only': (value / egm_only) if egm_only else None,
                         'kernel_ms_one_group': {n: float(v) for n, v in zip(CLASS_NAMES, serial_ms)} if serial_ms is not None else None,
                         'dense_kernels_ms_one_group': dense_sum,
                         'step_over_dense_one_group': (ms_step / dense_sum) if dense_sum else None,
                         'grid_kernel_issue_util': pg.get('valu_util'), 'grid_kernel_fp64_util': pg.get('valu_fp64_util'),
                         'grid_kernel_fp64_share_of_valu': (pg['valu_fp64_util'] / pg['valu_util']) if pg.get('valu_util') else None}}


def profiled_step(solver):
    """one more solve of the handle's current draws with HIP events around every launch, OUTSIDE every timed region (the events
    are created when first recorded: inside a timed step that cost would be in `value`)"""
    solver.set_profile(True)
    solver.solve(raise_on_error=False)
    out = solver.profile()
    solver.set_profile(False)
    return out


def serial_profile(solver):
    """one more solve of the handle's current draws with ONE draw group, HIP events on: device time per kernel class without
    concurrent streams (outside every timed region); the grouping is restored afterwards"""
    groups = solver.schedule()[0]
    solver.set_groups(1)
    solver.set_profile(True)
    solver.solve(raise_on_error=False)
    ms, launches, _ = solver.profile()
    solver.set_profile(False)
    solver.set_groups(groups)
    return ms, launches


def timed_leg(workload, ndraw, model=None, drawgen=None, params=None, label=None, chunk=None, sync=None, counters_as=None, shard=None):
    """One warm-up and one timed step of `ndraw` draws of a workload through one handle of `chunk` draws (default: all of
    them): the per-GPU share of a stress configuration.  Returns the leg's record.

    shard = (world, rank, all_gather_object): `params` holds the draws of the WHOLE job and this rank solves its shard of them --
    the interleave d mod world in the warm-up step, whose per-draw counters (evaluations, re-basing calls) are then exchanged over the
    host-side group and give the cost-balanced shards of the timed step (parallel.shard_indices: as an estimation loop would
    re-balance from its previous iteration)."""
    import torch
    from egdst_amd import build, parallel, runtime, workloads
    if model is None:
        model, drawgen = workloads.WORKLOADS[workload]()
    chunk = min(chunk or ndraw, ndraw)
    flags = workloads.BATCH_BUILD_FLAGS.get(workload, []) if chunk >= workloads.BATCH_BUILD_MIN_DRAWS.get(workload, 1 << 30) else []
    lib = build.build_model(model, extra_flags=flags)
    desc = model.descriptor()
    if params is None:
        params = drawgen(ndraw) if drawgen else np.tile(model.param_vector(), (ndraw, 1))
    params_all, my_idx, balance = params, None, None
    if shard is not None:
        world_, rank_, gather_ = shard
        my_idx = parallel.shard_indices(len(params_all), world_, rank_)
        params = params_all[my_idx]
    plan = parallel.plan_chunks(ndraw, chunk)
    solver = runtime.Solver(lib, desc, ndraw=chunk, keep_history=False)
    rec = {}
    for timed in (False, True):
        if timed and shard is not None:
            # the warm-up's counters of every draw of the job -> cost-balanced shards for the timed step
            mine = np.stack([my_idx.astype(np.float64), parallel.draw_cost(cost_ev, cost_wk)]) if len(my_idx) else np.zeros((2, 0))
            parts = gather_(mine)
            cost = np.zeros(len(params_all))
            for part in parts:
                cost[part[0].astype(np.int64)] = part[1]
            my_idx = parallel.shard_indices(len(params_all), world_, rank_, cost)
            params = params_all[my_idx]
            ndraw = len(my_idx)
            plan = parallel.plan_chunks(ndraw, chunk)
            balance = {'max_over_mean_shard_cost_interleaved': parallel.shard_balance(cost, world_, by_cost=False),
                       'max_over_mean_shard_cost_balanced': parallel.shard_balance(cost, world_, by_cost=True)}
        cost_ev, cost_wk = np.zeros(ndraw), np.zeros(ndraw)
        if sync:
            sync()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev_ref = ev_exec = nfail = 0
        for c0, valid in plan:
            p = params[c0:c0 + valid]
            if valid < chunk:
                p = np.concatenate([p, np.repeat(p[:1], chunk - valid, axis=0)])
            solver.set_params(p)
            solver.solve_async()
            solver.sync(raise_on_error=False)
            st, per, cred = solver.status()[0], solver.evals()[1], solver.evals_credited()
            ok = st[:valid] == 0
            ev_ref += int(per[:valid].sum())
            ev_exec += int((per[:valid] - cred[:valid])[ok].sum())
            nfail += int((~ok).sum())
            if shard is not None and not timed:
                cost_ev[c0:c0 + valid], cost_wk[c0:c0 + valid] = per[:valid], solver.work()[:valid]
        if sync:
            sync()
        dt = time.perf_counter() - t0
        rec = {'dt': dt, 'ev_ref': ev_ref, 'ev_exec': ev_exec, 'nfail': nfail}
    kms, klaunch, algbytes = profiled_step(solver)   # the last chunk's draws once more, HIP events on, outside the timed step
    tps, regen = solver.tp_stats(), solver.regenerations()
    nst, nt = lib.info.nst, desc['T'] - desc['t0'] + 1
    serial_ms, serial_launch = serial_profile(solver)
    solver.close()
    value = rec['ev_exec'] / rec['dt']
    out = {'workload': label or '%s x %d draws' % (workload, ndraw), 'ndraw': ndraw, 'chunk': chunk, 'build_flags': flags,
           'ms_per_step': rec['dt'] * 1e3, 'value': value, 'unit': 'evals/s', 'evals_executed': rec['ev_exec'],
           'evals_reference': rec['ev_ref'], 'failed_draws': rec['nfail'],
           'kernel_ms_last_solve_summed_over_concurrent_streams': {n: float(v) for n, v in zip(CLASS_NAMES, kms)},
           'config': 'T=%d, ngridm=%d, ny=%d, nd=%d, nst=%d, a0=%g, mmax=%g' % (desc['T'], desc['ngridm'], desc['ny'], lib.info.nd,
                                                                                   lib.info.nst, desc['a0'], desc['mmax'])}
    out['roofline'] = roofline_record('%s_ndraw%d' % (counters_as or workload, chunk), kms, klaunch, algbytes, rec['ev_exec'] / max(len(plan), 1), value,
                                      rec['dt'] * 1e3 / max(len(plan), 1), chunk * nst * nt, int(tps[:, 1].sum()) if (lib.info.nd > 1 and int(tps.sum()) > 0) else chunk * nst * nt,   # (no throughput path: every cell is k_envelope's)
                                      int(regen.sum()), serial_ms=serial_ms, serial_launch=serial_launch)
    if balance:
        out['shard_balance'] = balance
    out['_raw'] = rec
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--workload', default='C2')
    ap.add_argument('--scaling', choices=('weak', 'strong'), default='weak')
    ap.add_argument('--ndraw', type=int, default=4096, help='weak scaling: parameter draws per GPU and step')
    ap.add_argument('--ndraw-total', type=int, default=1024, help='strong scaling: draws of the whole job per step')
    ap.add_argument('--chunk', type=int, default=0, help='draws solved per handle at a time (0: per workload)')
    ap.add_argument('--rows-cap', type=int, default=0, help='compact physical row capacity (0 = exact, ngridmax rows)')
    ap.add_argument('--small', action='store_true', help='reduced grid of the workload (plumbing rehearsals, not a result)')
    ap.add_argument('--no-perturb', action='store_true', help='the same draws every step')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-single-solve', action='store_true', help='skip the one-draw latency legs (profiling runs)')
    ap.add_argument('--no-extras', action='store_true', help='skip the copy-peak, export and estimation legs')
    ap.add_argument('--rehearse-legs', action='store_true', help='N > 1: the strong-scaling sub-records on reduced batches (C5 x 32, C4 x 8 draws): a plumbing rehearsal on one card, not a result')
    ap.add_argument('--no-legs', action='store_true', help='skip the legs of the stress configurations (C4 x 32, C5 x 128, C2 a0=0; N > 1: the strong-scaling batches)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))

    import torch
    import torch.distributed as dist
    from egdst_amd import build, parallel, runtime, workloads

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus and rank == 0:
        print('bench.py: --gpus %d but WORLD_SIZE=%d; reporting n_gpus=%d' % (args.gpus, world, world), file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the egdst hot path has no CPU fallback')
    torch.cuda.set_device(local_rank % torch.cuda.device_count())   # (rehearsals put several ranks on one card)
    use_dist = world > 1 or 'TORCHELASTIC_RUN_ID' in os.environ   # under torchrun also with one rank
    if use_dist:
        # Rendezvous and the timing barriers go over gloo (host side).  The RCCL communicator for the path's one
        # collective is created AFTER the timed region: an initialised RCCL communicator holds hardware queues of its
        # own, which the streams of the solver then have to share (measured with one rank: 346 -> 450 ms per step).
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        # (gloo announces its connections on the C++ stdout; the contract is ONE JSON line there: send them to stderr)
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group('gloo')
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    wl_kw = {}
    if args.small:
        wl_kw = {'C2': dict(ngridm=200, T=20), 'C4': dict(ngridm=2048, T=20), 'C5': dict(ngridm=500, T=20)}.get(args.workload, {})
    model, drawgen = workloads.WORKLOADS[args.workload](**wl_kw)
    lib = build.build_model(model)  # prebuilt in-tree by __graft_entry__.build(); rebuilds if stale
    desc = model.descriptor()
    nparam = lib.info.nparam

    # ---- the draws of this rank, per step ---------------------------------------------------------------------
    if args.scaling == 'weak':
        job = args.ndraw * world
        lo, hi = rank * args.ndraw, (rank + 1) * args.ndraw
    else:
        job = args.ndraw_total
        lo, hi = parallel.shard_bounds(job, world, rank)
    mine_n = hi - lo
    chunk = args.chunk or DEFAULT_CHUNK.get(args.workload, 256)
    chunk = max(1, min(chunk, max(mine_n, 1)))
    plan = parallel.plan_chunks(mine_n, chunk)     # [(first draw of the shard, draws that count)] per chunk
    nsteps_all = args.warmup + args.steps

    def step_draws(s):
        """the job's draws, moved a little from step to step as an estimation loop would move them (each parameter times
        1 + 0.5 % * U(-1, 1), seeded by the step): close enough for the handle's history to mean something, never equal"""
        seed = {'C4': 20240}.get(args.workload, 20241)
        allp = drawgen(job, seed=seed) if drawgen else np.tile(model.param_vector(), (job, 1))
        if not args.no_perturb:
            allp = allp * (1.0 + 0.005 * (2.0 * np.random.default_rng(7919 + s).random(allp.shape) - 1.0))
        return np.ascontiguousarray(allp[lo:hi])

    host_draws = [step_draws(s) for s in range(nsteps_all)]
    # inputs resident in HBM before the timed region: [step][draw][param], padded to whole chunks with the shard's first draw
    nchunks = len(plan)
    padded = nchunks * chunk
    dev = torch.zeros(nsteps_all, max(padded, 1), max(nparam, 1), dtype=torch.float64, device='cuda')
    for s, p in enumerate(host_draws):
        if mine_n:
            pp = np.concatenate([p, np.repeat(p[:1], padded - mine_n, axis=0)]) if padded > mine_n else p
            dev[s, :padded] = torch.from_numpy(pp).cuda()

    stream = torch.cuda.Stream()
    # large batches of a workload may have a build variant of their own (workloads.BATCH_BUILD_FLAGS: other register
    # budgets for k_envelope or k_grid_lds); the one-draw latency legs below use the default build
    batch_flags = workloads.BATCH_BUILD_FLAGS.get(args.workload, []) if (chunk >= workloads.BATCH_BUILD_MIN_DRAWS.get(args.workload, 1 << 30) and not args.small) else []
    lib_batch = build.build_model(model, extra_flags=batch_flags) if batch_flags else lib
    solver = runtime.Solver(lib_batch, desc, ndraw=chunk, keep_history=False, stream=stream.cuda_stream, rows_cap=args.rows_cap) if mine_n else None
    obj = torch.full((nsteps_all, max(padded, 1), 2), float('nan'), dtype=torch.float64, device='cuda')
    torch.cuda.synchronize()

    def sync_all():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    def run_step(s):
        """one step of this rank: every chunk of its shard through the one handle; returns the counters"""
        ev_ref = ev_exec = nfail = ndone = 0
        for c, (c0, valid) in enumerate(plan):                  # (the last chunk may be padded: valid < chunk)
            solver.set_params_dev(dev[s, c * chunk:(c + 1) * chunk].data_ptr())
            solver.solve_async()
            solver.sync(raise_on_error=False)
            st, _ = solver.status()
            per = solver.evals()[1]
            cred = solver.evals_credited()
            ok = st[:valid] == 0
            ev_ref += int(per[:valid].sum())
            ev_exec += int((per[:valid] - cred[:valid])[ok].sum())
            nfail += int((~ok).sum())
            ndone += valid
            if not solver.rows_cap:
                with torch.cuda.stream(stream):
                    solver.objective_dev(obj[s, c * chunk:(c + 1) * chunk].data_ptr())
            else:
                obj[s, c * chunk:(c + 1) * chunk] = torch.from_numpy(solver.objective()).cuda()
        return ev_ref, ev_exec, nfail, ndone

    for s in range(args.warmup):
        if solver:
            run_step(s)

    sync_all()
    t0 = time.perf_counter()
    tot = np.zeros(4, dtype=np.int64)
    step_ms = []
    for s in range(args.warmup, nsteps_all):
        if solver:
            ts_ = time.perf_counter()
            tot += np.array(run_step(s), dtype=np.int64)
            step_ms.append((time.perf_counter() - ts_) * 1e3)   # (every step ends with a host sync: status and counts are read back)
    sync_all()
    dt = time.perf_counter() - t0

    # the last step's draws once more with HIP events around every launch (16 streams), then with one draw group: outside the timed region
    kms, klaunch, algbytes = profiled_step(solver) if solver else (np.zeros(9), np.zeros(9, dtype=np.int32), 0)
    tps_last = solver.tp_stats() if solver else np.zeros((1, 2))
    regen_last = int(solver.regenerations().sum()) if solver else 0
    serial_ms, serial_launch = serial_profile(solver) if (solver and rank == 0 and world == 1 and not args.no_extras) else (None, None)
    # ---- final objective reduce: the only collective of the path (RCCL over xGMI when world > 1) -----------------
    last = obj[nsteps_all - 1, :mine_n, 0] if mine_n else torch.zeros(0, dtype=torch.float64, device='cuda')
    okmask = ~torch.isnan(last)
    red = torch.stack([torch.where(okmask, last, torch.zeros_like(last)).sum(), okmask.sum().double()])
    tt = torch.tensor([dt], dtype=torch.float64)
    cnt = torch.tensor(tot.astype(np.float64))
    if use_dist:
        if world <= torch.cuda.device_count():
            rccl = dist.new_group(backend='nccl', device_id=torch.device('cuda', torch.cuda.current_device()))
            dist.all_reduce(red, op=dist.ReduceOp.SUM, group=rccl)   # objective contributions: RCCL over xGMI
        else:   # rehearsal with several ranks on one card (RCCL refuses duplicate devices): the same reduce over gloo
            red_h = red.cpu()
            dist.all_reduce(red_h, op=dist.ReduceOp.SUM)
            red = red_h
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)                # timing bookkeeping: host
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    dt_max = float(tt.item())
    ev_ref_all, ev_exec_all, nfail_all, ndone_all = [float(x) for x in cnt.tolist()]
    main_extra = {'capacity_retries': solver.capacity_retries if solver else 0,
                  'schedule': dict(zip(('groups', 'straggler_lanes', 'straggler_draws'), solver.schedule())) if solver else {},
                  'envelope_cells_by_throughput_path': solver.tp_stats().sum(axis=0).tolist() if solver else None}

    # ---- N > 1: the north_star's batches, strong scaling -- one timed step of C5 x 1024 and of C4 x 256 draws sharded over the
    # ranks (contiguous shards, chunks that fit HBM through one handle, no collective but the timing barriers over gloo)
    want_legs = not args.no_legs and (not args.small or args.rehearse_legs) and args.workload == 'C2' and args.scaling == 'weak' and args.rows_cap == 0
    strong = None
    if want_legs and world > 1:
        if solver:
            solver.close()
            solver = None
        torch.cuda.empty_cache()
        strong = {}
        for wl, total, ch in ((('C5', 32, 8), ('C4', 8, 4)) if args.rehearse_legs else (('C5', 1024, 128), ('C4', 256, 32))):
            lo2, hi2 = parallel.shard_bounds(total, world, rank)   # (the shard SIZES; which draws: parallel.shard_indices, see timed_leg)
            m2, gen2 = workloads.WORKLOADS[wl]()

            def gather_obj(x):
                outl = [None] * world
                dist.all_gather_object(outl, x)   # (host side, gloo: between the warm-up and the timed step)
                return outl
            rec = timed_leg(wl, len(parallel.shard_indices(total, world, rank)), model=m2, params=gen2(total), chunk=ch, sync=sync_all,
                            shard=(world, rank, gather_obj)) if total >= world else None
            raw = rec['_raw'] if rec else {'dt': 0.0, 'ev_ref': 0, 'ev_exec': 0, 'nfail': 0}
            t2 = torch.tensor([raw['dt']], dtype=torch.float64)
            c2 = torch.tensor([float(raw['ev_ref']), float(raw['ev_exec']), float(raw['nfail'])], dtype=torch.float64)
            dist.all_reduce(t2, op=dist.ReduceOp.MAX)
            dist.all_reduce(c2, op=dist.ReduceOp.SUM)
            if rank == 0:
                rec.pop('_raw', None)
                strong['%sx%d' % (wl, total)] = dict(rec, workload='%s x %d draws over %d GPUs (strong scaling, %d per handle)' % (wl, total, world, ch),
                                                     ndraw=total, draws_rank0=hi2 - lo2, ms_per_step=float(t2.item()) * 1e3,
                                                     value=float(c2[1].item()) / float(t2.item()), evals_executed=float(c2[1].item()),
                                                     evals_reference=float(c2[0].item()), failed_draws=float(c2[2].item()), scaling='strong')

    if rank == 0:
        ms_step = dt_max / args.steps * 1e3
        out = {
            'metric': 'EGM grid-point x shock evals/sec (batched backward induction, all draws, all periods)',
            'value': ev_exec_all / dt_max, 'unit': 'evals/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': ms_step, 'higher_is_better': True, 'scaling': args.scaling,
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': '%s%s: %s, T=%d, ngridm=%d, ny=%d, nd=%d, nst=%d, a0=%g, mmax=%g' % (
                args.workload, ' (REDUCED GRID, rehearsal only)' if args.small else '', model.label, desc['T'], desc['ngridm'],
                desc['ny'], lib.info.nd, lib.info.nst, desc['a0'], desc['mmax']),
                'draws_per_step_all_gpus': job, 'draws_per_gpu': args.ndraw if args.scaling == 'weak' else '%d..%d' % (job // world, -(-job // world)),
                'chunk_draws_per_handle': chunk, 'chunks_per_step_rank0': nchunks, 'rows_cap': args.rows_cap, 'batch_build_flags': batch_flags,
                'draws_perturbed_every_step': not args.no_perturb,
                'parallelism': 'draws sharded over %d rank(s), no data-path collective, one RCCL all-reduce of the objective' % world},
            'evals_executed_per_step': ev_exec_all / args.steps, 'evals_reference_per_step': ev_ref_all / args.steps,
            'failed_draws_per_step': nfail_all / args.steps, 'draws_per_step': ndone_all / args.steps,
            'step_ms_rank0': [round(t, 2) for t in step_ms],
            'capacity_retries': main_extra['capacity_retries'], 'schedule': main_extra['schedule'],
            'objective_mean': float(red[0].item() / max(red[1].item(), 1.0)),
            'kernel_ms_last_solve_summed_over_concurrent_streams': {n: float(v) for n, v in zip(CLASS_NAMES, kms)},
            'envelope_cells_by_throughput_path': main_extra['envelope_cells_by_throughput_path'],
            'roofline': roofline_record('%s_ndraw%d' % (args.workload, chunk), kms, klaunch, algbytes, ev_exec_all / args.steps / max(world * nchunks, 1),
                                        ev_exec_all / dt_max, ms_step / max(nchunks, 1), chunk * lib.info.nst * (desc['T'] - desc['t0'] + 1),
                                        int(tps_last[:, 1].sum()) if (lib.info.nd > 1 and int(tps_last.sum()) > 0) else chunk * lib.info.nst * (desc['T'] - desc['t0'] + 1),
                                        regen_last, small=args.small, serial_ms=serial_ms, serial_launch=serial_launch),
        }
        one = np.asarray(model.param_vector(), dtype=np.float64)[None] if mine_n else None   # fixed parameters: comparable from run to run
        if not args.no_single_solve and world == 1 and mine_n:
            # single-solve latency (one draw): the "full backward-induction wall time" half of the metric
            s1 = runtime.Solver(lib, desc, ndraw=1, keep_history=False)
            s1.set_params(one)
            s1.solve(raise_on_error=False)
            t1 = time.perf_counter()
            for _ in range(3):
                s1.solve(raise_on_error=False)
            out['single_solve_ms'] = (time.perf_counter() - t1) / 3 * 1e3
            s1.close()
            if not args.no_extras:
                # the same including the D2H copy of the full solution (SURVEY.md section 8d metric 2)
                s2 = runtime.Solver(lib, desc, ndraw=1, keep_history=True)
                s2.set_params(one)
                s2.solve(raise_on_error=False)
                t1 = time.perf_counter()
                s2.solve(raise_on_error=False)
                sol = s2.solution(0)
                out['single_solve_plus_export_ms'] = (time.perf_counter() - t1) * 1e3
                out['export_bytes'] = int(sol.len.sum()) * 24 + int(sol.thlen.sum()) * 16
                s2.close()
        if not args.no_extras and not args.no_single_solve and world == 1 and args.workload == 'C2' and not args.small:
            # the other one-draw latencies the verdicts follow: C2 with a0 = 0 (the configuration of rounds 1-3) and C3
            def one_solve(mk):
                mm = mk()
                ss = runtime.Solver(build.build_model(mm), mm.descriptor(), ndraw=1, keep_history=False)
                ss.set_params(mm.param_vector()[None])
                ss.solve(raise_on_error=False)
                t_ = time.perf_counter()
                for _ in range(3):
                    ss.solve(raise_on_error=False)
                r_ = {'single_solve_ms': (time.perf_counter() - t_) / 3 * 1e3, 'evals': int(ss.evals()[0]), 'status': int(ss.status()[0][0])}
                ss.close()
                return r_
            out['single_solve_other'] = {'c2_a0_0': one_solve(lambda: workloads.c2(a0=0)[0]), 'c3': one_solve(lambda: workloads.c3()[0])}
        if not args.no_extras and world == 1:
            # measured streaming-copy rate of this box (read + write), the practical HBM ceiling beside the 8 TB/s spec
            a = torch.empty(1 << 28, dtype=torch.float64, device='cuda')   # 2 GiB
            b_ = torch.empty_like(a)
            b_.copy_(a)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(5):
                b_.copy_(a)
            torch.cuda.synchronize()
            out['roofline']['hbm_copy_GBps_measured'] = 5 * 2 * a.numel() * 8 / (time.perf_counter() - t1) / 1e9
            del a, b_
        if strong is not None:
            out['strong'] = strong
        if want_legs and world == 1:
            # the per-GPU shares of the stress configurations and C2 on the surveyed credit limit (see the module docstring)
            if solver:
                solver.close()
                solver = None
            torch.cuda.empty_cache()
            from egdst_amd import examples
            legs = {}
            legs['C4x32'] = timed_leg('C4', 32, label='C4 x 32 draws: the per-GPU share of BASELINE configs[3] (256 draws over 8 GPUs)')
            legs['C5x128'] = timed_leg('C5', 128, label='C5 x 128 draws at full size: the per-GPU share of BASELINE configs[4] (1024 draws over 8 GPUs)')
            m0 = workloads.c2(a0=0)[0]
            legs['c2_a0_0_batch'] = timed_leg('C2', args.ndraw, model=m0, params=host_draws[-1], counters_as='C2a0',
                                              label='C2 with a0=0 (the headline configuration of rounds 1-3), the same %d draws' % args.ndraw)
            for v in legs.values():
                v.pop('_raw', None)
            out['legs'] = legs
        if not args.no_cpu_baseline and world == 1 and mine_n:   # reported baseline: rank 0 at N=1 only
            out['cpu_baseline'] = cpu_baseline(args.workload, wl_kw, host_draws[-1])
            out['speedup_vs_cpu_1thread'] = out['value'] / out['cpu_baseline']['value']
            if out.get('single_solve_ms'):
                out['single_solve_speedup_vs_cpu'] = out['cpu_baseline']['wall_s_per_solve'] * 1e3 / out['single_solve_ms']
            if 'c3' in out.get('single_solve_other', {}) and args.workload == 'C2' and not args.small:
                # BASELINE configs[2] (model_occ3, one solve on one GPU): the oracle's single-thread time for that very solve (~2.6 s)
                m3 = workloads.c3()[0]
                ev3, n3, dt3 = _oracle_worker(('C3', {}, [m3.param_vector()], 0.0))
                c3 = out['single_solve_other']['c3']
                c3['cpu_ms'], c3['cpu_evals'] = dt3 / max(n3, 1) * 1e3, int(ev3)
                c3['speedup_vs_cpu_1thread'] = c3['cpu_ms'] / c3['single_solve_ms']
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
