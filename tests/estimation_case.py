"""Shared by the CPU-harness test and the GPU test of the on-device estimation step (egdst_simulate_batch_moments):
the host replay of the uniforms, the oracle's moments per draw, and the comparison."""
import warnings

import numpy as np

MASK = (1 << 64) - 1


def uniforms(seed, n):
    """egdst_uniform(seed, 0..n-1) in numpy (splitmix64, include/egdst.h)"""
    k = np.arange(1, n + 1, dtype=np.uint64)
    with np.errstate(over='ignore'):
        z = np.uint64(seed) + k * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * 2.0 ** -53


def oracle_moments(orc, params, init, rs, rndtype):
    """(means, counts, solved) of the oracle's simulated paths for one parameter vector"""
    sol = orc.solve(params)
    if sol.rc != 0:
        return None, None, False
    sims = orc.sim(sol, init, rs, rndtype=rndtype, params=params)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        means = np.nanmean(sims, axis=0)
    return means, (~np.isnan(sims)).sum(axis=0), True


def check(solver, orc, P, init, seed, rndtype, lib=None):
    """solve must have run on `solver` with the parameter rows P.  Returns a list of problems (empty = pass)."""
    nt = solver.nt
    nsim = len(init)
    rs = uniforms(seed, 4 * nt * (1 if rndtype == 1 else nsim))
    if lib is not None:   # the library's own host replay
        assert all(lib.lib.egdst_uniform(seed, int(k)) == rs[k] for k in (0, 1, 17, len(rs) - 1))
    info = solver.lib.info
    nout = 11 + info.nnst + info.nnd + info.neq
    rng = np.random.default_rng(3)
    target = rng.uniform(0, 2, (nt, nout))
    weight = np.zeros((nt, nout))
    weight[1:, 1] = 1.0     # consumption by period
    weight[1:, 4] = 4.0     # share working (decision index) by period
    if 'cpuemu' in solver.lib.path:   # the harness has no device: "device" buffers are host arrays
        means = np.zeros((solver.ndraw, nt, nout))
        counts = np.zeros((solver.ndraw, nt, nout), dtype=np.int32)
        obj = np.zeros(solver.ndraw)
        solver.simulate_batch_moments(init, seed=seed, rndtype=rndtype, target=target, weight=weight,
                                      means_dev=means.ctypes.data, counts_dev=counts.ctypes.data, obj_dev=obj.ctypes.data)
    else:
        means, counts, obj = solver.simulate_batch_moments(init, seed=seed, rndtype=rndtype, target=target, weight=weight)
    problems = []
    st = solver.status()[0]
    for d in range(solver.ndraw):
        rm, rc, ok = oracle_moments(orc, P[d], init, rs, rndtype)
        if not ok:
            if st[d] == 0 or not np.isnan(obj[d]):
                problems.append('draw %d: oracle fails, device status %d objective %r' % (d, st[d], obj[d]))
            continue
        if not np.array_equal(counts[d], rc):
            problems.append('draw %d: counts differ' % d)
        fin = np.isfinite(rm)
        if not np.array_equal(np.isnan(means[d]), ~fin):
            problems.append('draw %d: NaN pattern differs' % d)
        elif np.any(np.abs(means[d][fin] - rm[fin]) > 1e-13 * np.maximum(1, np.abs(rm[fin]))):
            problems.append('draw %d: means differ by %g' % (d, np.abs(means[d][fin] - rm[fin]).max()))
        w = weight != 0
        if np.any(rc[w] == 0):
            ro = np.nan
        else:
            ro = float((weight[w] * (rm[w] - target[w]) ** 2).sum())
        if not (np.isnan(ro) and np.isnan(obj[d])) and abs(obj[d] - ro) > 1e-11 * max(1.0, abs(ro)):
            problems.append('draw %d: objective %r vs %r' % (d, obj[d], ro))
    return problems
