"""Diagnostic: what the slow paths of a C2 batch are given to do (-DEGDST_CENSUS build: k_envelope's cells and jobs with the reason
the throughput path left them, k_fixup's streams, slow k_probe waves).
   python tests/diag/gpu_census.py [a0=-5] [ndraw=4096] [flags: batch|default] [perturbation seed]"""
import ctypes as C
import os, sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from egdst_amd import build, examples, runtime, workloads

a0 = float(sys.argv[1]) if len(sys.argv) > 1 else -5.0
nd = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
var = sys.argv[3] if len(sys.argv) > 3 else 'batch'
pert = int(sys.argv[4]) if len(sys.argv) > 4 else -1   # >= 0: the draws moved by 0.5 % with this seed (gpu_groups_sweep.py, bench.py)
wl = os.environ.get('EGDST_DIAG_WL', 'C2')   # C2 (the a0 of argv), or a stress workload: C5, C4, C3 with its own draws
if wl == 'C2':
    m = examples.retirement_sig(T=60, ngridm=1000, ngridmax=10000, nthrhmax=1000, ny=10, a0=a0)
    _, gen = workloads.c2(a0=0)
else:
    m, gen = workloads.WORKLOADS[wl]()
P = gen(nd) if gen else np.tile(m.param_vector(), (nd, 1))
if pert >= 0:
    P = P * (1 + 0.005 * (2 * np.random.default_rng(pert).random(P.shape) - 1))
flags = (workloads.BATCH_BUILD_FLAGS.get(wl, []) if var == 'batch' else []) + ['-DEGDST_CENSUS']
lib = build.build_model(m, extra_flags=flags)
L = lib.lib
L.egdst_census_read.argtypes = [C.c_void_p, C.c_int, C.c_int]
s = runtime.Solver(lib, m.descriptor(), ndraw=nd, keep_history=False)
s.set_params(P)
s.solve(raise_on_error=False)
L.egdst_census_read(None, 0, 1)
t = time.perf_counter(); s.solve(raise_on_error=False); ms = (time.perf_counter() - t) * 1e3
cap = 1 << 20
buf = np.zeros((cap, 8), dtype=np.int32)
n = L.egdst_census_read(buf.ctypes.data, cap, 1)
rec = buf[:min(n, cap)]
st = s.status()[0]
print(wl, 'a0=%g ndraw=%d flags=%s: %.1f ms, failed %d, census records %d, tp done/left %s' % (a0, nd, flags, ms, int((st != 0).sum()), n, s.tp_stats().sum(axis=0).tolist()))
codes, cnts = np.unique(st[st != 0], return_counts=True)
print('failure codes:', dict(zip(codes.tolist(), cnts.tolist())))
us = lambda x: x * 0.01   # ticks of 10 ns -> us
for kind, name in ((2, 'k_envelope cells'), (1, 'k_envelope jobs'), (3, 'k_fixup streams'), (4, 'slow k_probe waves'), (5, 'tp walks that gave up')):
    r = rec[rec[:, 0] == kind]
    print('--- kind %d %s: %d records, sum %.1f ms' % (kind, name, len(r), us(r[:, 7].astype(np.int64).sum()) / 1e3))
    if not len(r):
        continue
    if kind == 2:
        for why in np.unique(r[:, 6]):
            q = r[r[:, 6] == why]
            print('   why=%2d: %6d cells, sum %9.1f ms, median %7.1f us, max %8.1f us; errors %s' % (
                why, len(q), us(q[:, 7].astype(np.int64).sum()) / 1e3, us(np.median(q[:, 7])), us(q[:, 7].max()),
                dict(zip(*[x.tolist() for x in np.unique(q[:, 5], return_counts=True)]))))
        top = r[np.argsort(-r[:, 7])[:15]]
        for x in top:
            print('   top: draw %5d it %2d rows %5d th %3d err %4d why %2d  %9.1f us' % (x[1], x[2], x[3], x[4], x[5], x[6], us(x[7])))
        # per period: sum and max
        per = [(it, us(r[r[:, 2] == it][:, 7].astype(np.int64).sum()), us(r[r[:, 2] == it][:, 7].max()), int((r[:, 2] == it).sum())) for it in np.unique(r[:, 2])]
        print('   per period (it: cells, sum us, max us):', ' '.join('%d:%d,%.0f,%.0f' % (it, c, sm, mx) for it, sm, mx, c in per))
    if kind == 1:
        why_, bad_, sort_us = r[:, 6] >> 24, (r[:, 6] >> 20) & 15, r[:, 6] & 0xfffff
        r = r.copy(); r[:, 6] = why_
        for bb in np.unique(bad_):
            q = r[bad_ == bb]
            print('   lists %s: %6d jobs, sum %9.1f ms (sorts %9.1f ms), median %7.1f us (sort %6.1f us), max %8.1f us' % (
                {0: 'in order   ', 1: 'network    ', 2: 'counted    '}.get(int(bb), str(bb)), len(q), us(q[:, 7].astype(np.int64).sum()) / 1e3,
                sort_us[bad_ == bb].astype(np.int64).sum() / 1e3, us(np.median(q[:, 7])), np.median(sort_us[bad_ == bb]), us(q[:, 7].max())))
        for jb in np.unique(r[:, 3]):
            q = r[r[:, 3] == jb]
            print('   job %d (%s): %6d jobs, sum %9.1f ms, median %7.1f us, out of order %d' % (jb, 'primary' if jb == 2 else 'secondary', len(q),
                  us(q[:, 7].astype(np.int64).sum()) / 1e3, us(np.median(q[:, 7])), int((bad_[r[:, 3] == jb] != 0).sum())))
        for path in (0, 1):
            q = r[(r[:, 5] >> 16) == path]
            if len(q):
                print('   %s: %6d jobs, sum %9.1f ms, median %7.1f us, max %8.1f us, points median %d max %d' % (
                    'LDS   ' if path == 0 else 'global', len(q), us(q[:, 7].astype(np.int64).sum()) / 1e3, us(np.median(q[:, 7])), us(q[:, 7].max()),
                    int(np.median(q[:, 4])), int(q[:, 4].max())))
        top = r[np.argsort(-r[:, 7])[:15]]
        for x in top:
            print('   top: draw %5d it %2d job %d points %6d nf %4d path %d why %2d  %9.1f us' % (x[1], x[2], x[3], x[4], x[5] & 0xffff, x[5] >> 16, x[6], us(x[7])))
        h, e = np.histogram(r[:, 4], bins=[0, 500, 1100, 2100, 2300, 3000, 4200, 6000, 8000, 10001, 1 << 30])
        print('   points histogram:', list(zip(e[:-1].tolist(), h.tolist())))
    if kind == 3:
        print('   median %.1f us, max %.1f us; kept points median %d max %d; calls median %d max %d; fast-forwarded calls >0 in %d' % (
            us(np.median(r[:, 7])), us(r[:, 7].max()), int(np.median(r[:, 4])), int(r[:, 4].max()), int(np.median(r[:, 5])), int(r[:, 5].max()), int((r[:, 6] > 0).sum())))
        top = r[np.argsort(-r[:, 7])[:10]]
        for x in top:
            print('   top: draw %5d it %2d id %d kept %5d calls %5d skipped %5d  %9.1f us' % (x[1], x[2], x[3], x[4], x[5], x[6], us(x[7])))
        dr = np.unique(r[:, 1])
        print('   draws with regenerated streams: %d, of which failed %d' % (len(dr), int((st[dr] != 0).sum())))
    if kind == 4:
        top = r[np.argsort(-r[:, 7])[:10]]
        for x in top:
            print('   top: draw %5d it %2d id %d calls %5d skipped %5d kept %d  %9.1f us' % (x[1], x[2], x[3], x[4], x[5], x[6], us(x[7])))
    if kind == 5:
        print('   by (stage, error):', dict(zip(*[x.tolist() for x in np.unique(r[:, 3] * 100000 + r[:, 4], return_counts=True)])))
# the walks of the throughput path: how much longer than its average workgroup does a launch's slowest one take?
r = rec[rec[:, 0] == 6]
if len(r):
    ng = s.schedule()[0]
    grp = (np.searchsorted(np.arange(ng + 1) * nd // ng, r[:, 1], side='right') - 1)
    for stage in (0, 1, 17):
        q = r[r[:, 3] == stage]
        if not len(q):
            continue
        g = grp[r[:, 3] == stage]
        key = g.astype(np.int64) * 1000 + q[:, 2]
        order = np.argsort(key, kind='stable')
        ks, tk = key[order], us(q[order, 7].astype(np.float64))
        bounds = np.flatnonzero(np.r_[True, ks[1:] != ks[:-1], True])
        mx = np.array([tk[a:b].max() for a, b in zip(bounds[:-1], bounds[1:])])
        mean = np.array([tk[a:b].mean() for a, b in zip(bounds[:-1], bounds[1:])])
        cnt = np.diff(bounds)
        print('--- k_tp_walk stage %d%s: %d walks, per walk median %.1f us, mean %.1f, p99 %.1f, max %.1f; per (group, period) launch: walks %.0f, slowest walk median %.0f us, mean %.0f, p90 %.0f, max %.0f (its mean walk %.0f us); sum over periods of the slowest, per group: %.1f ms' % (
            stage & 15, ' second tier' if stage & 16 else '', len(q), np.median(tk), tk.mean(), np.percentile(tk, 99), tk.max(), cnt.mean(), np.median(mx), mx.mean(),
            np.percentile(mx, 90), mx.max(), mean.mean(), mx.sum() / ng / 1e3))
        fb = (q[:, 5] >> 16) & 3
        for code, name in ((0, 'one wave'), (1, 'segments merged'), (2, 'segments, then one wave again')):
            w = q[fb == code]
            if len(w):
                print('   %-32s %7d walks, mean %.1f us, sum %.1f ms, rows out mean %.0f' % (name, len(w), us(w[:, 7].mean()), us(w[:, 7].astype(np.int64).sum()) / 1e3, w[:, 6].mean()))
        top = q[np.argsort(-q[:, 7])[:8]]
        for x in top:
            print('   top: draw %5d it %2d points %5d functions %3d rows %5d mode %d  %8.1f us' % (x[1], x[2], x[4], x[5] & 0xffff, x[6], (x[5] >> 16) & 3, us(x[7])))
        slow = q[us(q[:, 7]) > 150]
        if len(slow):
            print('   walks over 150 us: %d, functions median %d, rows out median %d, modes %s' % (len(slow), np.median(slow[:, 5] & 0xffff), np.median(slow[:, 6]),
                  dict(zip(*[x.tolist() for x in np.unique((slow[:, 5] >> 16) & 3, return_counts=True)]))))
# which draws own the k_envelope time
r = rec[rec[:, 0] == 2]
if len(r):
    tot = np.zeros(nd)
    np.add.at(tot, r[:, 1], us(r[:, 7].astype(np.float64)))
    o = np.argsort(-tot)[:20]
    print('draws by k_envelope time (us): ' + ' '.join('%d:%.0f%s' % (d, tot[d], '*' if st[d] else '') for d in o) + '   (* = failed)')
    print('k_envelope time of failed draws %.1f ms of %.1f ms' % (tot[st != 0].sum() / 1e3, tot.sum() / 1e3))
s.close()
