"""Validation of a C2 batch build on the shipped path (throughput path of the envelope step, k_grid_lds_cv) against the oracle:
every draw's status, error text and evaluation count, and the checksums of EVERY cell of `ntab` draws (the draws that give
k_envelope and k_fixup most to do first, then the rest in order), the oracle in a pool of processes.
   python tests/diag/gpu_r04_parity.py [a0=-5] [ndraw=4096] [ntab=256] [variant: batch|default] [procs=16]"""
import os, sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, 'tests/golden'); sys.path.insert(0, '.')
import multiprocessing as mp
import numpy as np

a0 = float(sys.argv[1]) if len(sys.argv) > 1 else -5.0
nd = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
ntab = int(sys.argv[3]) if len(sys.argv) > 3 else 256
var = sys.argv[4] if len(sys.argv) > 4 else 'batch'
procs = int(sys.argv[5]) if len(sys.argv) > 5 else 16


def _model():
    from egdst_amd import examples
    return examples.retirement_sig(T=60, ngridm=1000, ngridmax=10000, nthrhmax=1000, ny=10, a0=a0)


def _oracle(args):
    idx, P, want_tab = args
    from oracle_harness import Oracle
    from make_golden_big import cell_sums
    orc = Oracle(_model())
    out = []
    for i, p, wt in zip(idx, P, want_tab):
        r = orc.solve(p)
        out.append((i, r.rc, r.err.strip(), r.nevals, cell_sums(r) if wt else None, r.len.copy(), r.thlen.copy()))
    return out


if __name__ == '__main__':
    from egdst_amd import build, runtime, workloads
    m = _model()
    _, gen = workloads.c2(a0=0)
    P = gen(nd)
    flags = workloads.BATCH_BUILD_FLAGS['C2'] if var == 'batch' else []
    lib = build.build_model(m, extra_flags=flags)
    print('library:', lib.path, flush=True)
    s = runtime.Solver(lib, m.descriptor(), ndraw=nd, keep_history=True)
    s.set_params(P)
    s.solve(raise_on_error=False)
    t = time.perf_counter(); s.solve(raise_on_error=False); ms = (time.perf_counter() - t) * 1e3
    st, wh = s.status(); ev = s.evals()[1]
    tps = s.tp_stats()
    work, regen = s.work(), s.regenerations()
    print('a0=%g ndraw=%d %s: %.1f ms (history kept), failed %d, tp done/left %s, regenerated streams %d' % (
        a0, nd, flags, ms, int((st != 0).sum()), tps.sum(axis=0).tolist(), int(regen.sum())), flush=True)
    # table checks: the draws with most leftover cells / regenerations / re-basing work first
    score = tps[:, 1].astype(np.int64) * 1000 + regen.astype(np.int64) * 10 + (work > 0)
    order = np.argsort(-score, kind='stable')
    want = np.zeros(nd, dtype=bool)
    want[order[:ntab]] = True
    chunks = [(list(range(k, nd, procs)), P[k::procs], want[k::procs]) for k in range(procs)]
    t = time.time()
    with mp.get_context('spawn').Pool(procs) as pool:
        res = [r for part in pool.map(_oracle, chunks) for r in part]
    print('oracle: %d draws in %.0f s on %d processes' % (nd, time.time() - t, procs), flush=True)
    bad = tb = ntb = 0
    for i, rc, err, nev, sums, ln, th in res:
        ok = ((st[i] == 0) == (rc == 0)) and (st[i] != 0 or ev[i] == nev)
        if st[i] != 0 and rc != 0:
            ok = ok and lib.lib.egdst_strerror(int(st[i])).decode().strip() == err
        if not ok:
            bad += 1
            print('MISMATCH draw', i, P[i].round(4).tolist(), 'gpu', st[i], wh[i].tolist(), ev[i], '| oracle', rc, err[:50], nev, flush=True)
        if sums is not None:
            ntb += 1
            gl, gt = s.dims(i)
            gs = s.checksums(i)
            # (a failed draw: the cells before the failure compare; the oracle stops where the device stops)
            same = np.array_equal(gl, ln) and np.array_equal(gt, th) and np.array_equal(gs, sums)
            if not same:
                tb += 1
                d = np.argwhere((gl != ln) | (gt != th) | (gs != sums).any(axis=2))
                c = tuple(d[-1])
                print('TABLE MISMATCH draw %d status %d: %d cells differ, first (it, ist) %s: rows %s vs %s, thresholds %s vs %s, columns M C V TH D differ: %s; rows of the period before: %s' % (
                    i, st[i], len(d), d[-1].tolist(), gl[c], ln[c], gt[c], th[c], (gs[c] != sums[c]).astype(int).tolist(),
                    ln[c[0] + 1, c[1]] if c[0] + 1 < ln.shape[0] else -1), flush=True)
    print('%d draws: %d failed on both sides, %d status/evals mismatches; %d draws compared cell by cell (checksums of M, C, V, TH, D): %d mismatches' % (
        nd, int((st != 0).sum()), bad, ntb, tb), flush=True)
    s.close()
    sys.exit(1 if (bad or tb) else 0)
