"""Parity protocol between two solutions in the shared table layout (SURVEY.md §8d 'Parity protocol').

* identical row counts per (it, ist) cell and identical threshold counts,
* D sequences exact, |dTH| <= 1e-9*max(1,|TH|),
* M, C, A, V row-wise within RTOL=1e-10 relative (north_star: "policy/value outputs match the
  reference CPU MEX within 1e-10 relative fp64"); -inf must match exactly.
"""
import numpy as np

RTOL = 1e-10
TH_TOL = 1e-9


def rel_err(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    both_inf = np.isinf(a) & np.isinf(b) & (np.sign(a) == np.sign(b))
    both_nan = np.isnan(a) & np.isnan(b)
    with np.errstate(invalid='ignore'):
        e = np.abs(a - b) / np.maximum(1.0, np.abs(b))
    e = np.where(both_inf | both_nan, 0.0, e)
    e = np.where(np.isnan(e), np.inf, e)
    return e


def compare(sol, ref, rtol=RTOL, th_tol=TH_TOL):
    """Returns (ok, report dict). `sol` and `ref` expose len, thlen, M, C, V, D, TH arrays [nt, nst, ...]."""
    rep = {'max_rel': 0.0, 'max_dth': 0.0, 'cells': 0, 'rows': 0, 'problems': []}
    if sol.len.shape != ref.len.shape:
        rep['problems'].append('shape %s vs %s' % (sol.len.shape, ref.len.shape))
        return False, rep
    nt, nst = ref.len.shape
    for it in range(nt - 1, -1, -1):
        for ist in range(nst):
            n, nr = int(sol.len[it, ist]), int(ref.len[it, ist])
            m, mr = int(sol.thlen[it, ist]), int(ref.thlen[it, ist])
            if n != nr or m != mr:
                rep['problems'].append('it=%d ist=%d rows %d vs %d, thresholds %d vs %d' % (it, ist, n, nr, m, mr))
                if len(rep['problems']) > 8:
                    return False, rep
                continue
            if n == 0:
                continue
            rep['cells'] += 1
            rep['rows'] += n
            if not np.array_equal(sol.D[it, ist, :m], ref.D[it, ist, :m]):
                rep['problems'].append('it=%d ist=%d D differs' % (it, ist))
            dth = np.abs(sol.TH[it, ist, :m] - ref.TH[it, ist, :m]) / np.maximum(1.0, np.abs(ref.TH[it, ist, :m]))
            rep['max_dth'] = max(rep['max_dth'], float(dth.max()))
            for name in ('M', 'C', 'V'):
                a, b = getattr(sol, name)[it, ist, :n], getattr(ref, name)[it, ist, :n]
                e = rel_err(a, b)
                if e.max() > rep['max_rel']:
                    rep['max_rel'] = float(e.max())
                    rep['worst'] = (name, it, ist, int(e.argmax()), float(a[e.argmax()]), float(b[e.argmax()]))
            ea = rel_err(sol.M[it, ist, :n] - sol.C[it, ist, :n], ref.M[it, ist, :n] - ref.C[it, ist, :n])
            rep['max_rel'] = max(rep['max_rel'], float(ea.max()))
    ok = not rep['problems'] and rep['max_rel'] <= rtol and rep['max_dth'] <= th_tol
    return ok, rep
