"""ctypes binding of the per-model C ABI (include/egdst.h) + the solve/sim plumbing of the class.

No torch types cross the boundary; torch is only used by callers that want device-resident
parameter batches (``Solver.set_params_torch``).  There is no CPU fallback: if the HIP library
cannot be loaded, or no GPU is visible, every entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np


class EgdstDesc(C.Structure):
    _fields_ = [('t0', C.c_int), ('T', C.c_int), ('ngridm', C.c_int), ('ngridmax', C.c_int),
                ('nthrhmax', C.c_int), ('ny', C.c_int), ('mmax', C.c_double), ('a0', C.c_double),
                ('quadrature', C.POINTER(C.c_double))]


class EgdstModelInfo(C.Structure):
    _fields_ = [('nst', C.c_int), ('nd', C.c_int), ('nnst', C.c_int), ('nnd', C.c_int), ('nparam', C.c_int),
                ('neq', C.c_int), ('distrib', C.c_int), ('optim_MUnoD', C.c_int), ('optim_UnoD', C.c_int),
                ('optim_UasD', C.c_int), ('optim_TRPRnoSH', C.c_int), ('tolerance', C.c_double),
                ('zeroconsumption', C.c_double), ('doublepoint_delta', C.c_double), ('label', C.c_char_p)]


# every symbol include/egdst.h declares (checked by the CPU test-suite against the header)
ABI_SYMBOLS = ['egdst_get_model_info', 'egdst_strerror', 'egdst_last_error', 'egdst_create', 'egdst_destroy',
               'egdst_set_params', 'egdst_set_params_dev', 'egdst_solve_async', 'egdst_sync', 'egdst_solve',
               'egdst_get_status', 'egdst_get_evals', 'egdst_cell_dims', 'egdst_get_cell_M', 'egdst_get_cell_D',
               'egdst_get_solution', 'egdst_simulate', 'egdst_device_tables', 'egdst_get_debug', 'egdst_set_profile',
               'egdst_get_profile', 'egdst_objective_dev', 'egdst_get_objective', 'egdst_get_params',
               'egdst_create_compact', 'egdst_geometry', 'egdst_set_groups', 'egdst_set_adaptive', 'egdst_get_schedule', 'egdst_get_work', 'egdst_get_regenerations', 'egdst_call', 'egdst_simulate_moments',
               'egdst_get_checksums', 'egdst_math_eval', 'egdst_get_evals_credited', 'egdst_simulate_batch_moments',
               'egdst_uniform', 'egdst_set_dbgout', 'egdst_get_dbgout', 'egdst_get_walk_stats',
               'egdst_set_cell_M', 'egdst_set_cell_D', 'egdst_set_solution', 'egdst_get_tp_stats', 'egdst_get_group_profile']


class EgdstRuntimeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__('egdst error %d: %s' % (code, msg))
        self.code = code


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


class ModelLibrary:
    """One loaded per-model shared library."""

    def __init__(self, path, tag=''):
        if not os.path.exists(path):
            raise EgdstRuntimeError(3, 'HIP library %s is missing: run model.compile() (no CPU fallback exists)' % path)
        self.path, self.tag = path, tag
        self.lib = C.CDLL(path, mode=getattr(os, 'RTLD_LOCAL', 0) | getattr(os, 'RTLD_NOW', 2))
        L = self.lib
        for s in ABI_SYMBOLS:
            getattr(L, s)  # AttributeError if the library does not export what the header declares
        L.egdst_strerror.restype = C.c_char_p
        L.egdst_last_error.restype = C.c_char_p
        L.egdst_create.argtypes = [C.POINTER(EgdstDesc), C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
        L.egdst_create_compact.argtypes = [C.POINTER(EgdstDesc), C.c_int, C.c_int, C.c_int, C.c_void_p,
                                           C.POINTER(C.c_void_p)]
        L.egdst_set_groups.argtypes = [C.c_void_p, C.c_int]
        L.egdst_set_adaptive.argtypes = [C.c_void_p, C.c_int]
        L.egdst_call.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.egdst_get_schedule.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.egdst_geometry.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.egdst_get_objective.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.egdst_get_params.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.egdst_destroy.argtypes = [C.c_void_p]
        L.egdst_set_params.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
        L.egdst_set_params_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.egdst_solve_async.argtypes = [C.c_void_p]
        L.egdst_sync.argtypes = [C.c_void_p]
        L.egdst_solve.argtypes = [C.c_void_p]
        L.egdst_get_status.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.egdst_get_evals.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
        L.egdst_get_evals_credited.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
        L.egdst_simulate_batch_moments.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_void_p, C.c_longlong, C.c_ulonglong,
                                                   C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p, C.c_void_p, C.c_void_p]
        L.egdst_uniform.restype = C.c_double
        L.egdst_uniform.argtypes = [C.c_ulonglong, C.c_ulonglong]
        L.egdst_set_dbgout.argtypes = [C.c_void_p, C.c_int]
        L.egdst_get_dbgout.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]
        L.egdst_cell_dims.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.egdst_get_cell_M.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
        L.egdst_get_cell_D.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
        L.egdst_get_solution.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)] + \
            [C.POINTER(C.c_double)] * 5
        L.egdst_set_cell_M.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
        L.egdst_set_cell_D.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
        L.egdst_set_solution.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)] + \
            [C.POINTER(C.c_double)] * 5
        L.egdst_simulate.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double),
                                     C.c_longlong, C.c_int, C.POINTER(C.c_double)]
        L.egdst_simulate_moments.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double),
                                             C.c_longlong, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]
        L.egdst_get_debug.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.egdst_objective_dev.argtypes = [C.c_void_p, C.c_void_p]
        L.egdst_set_profile.argtypes = [C.c_void_p, C.c_int]
        L.egdst_get_profile.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_longlong)]
        L.egdst_get_checksums.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.egdst_math_eval.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.egdst_device_tables.argtypes = [C.c_void_p, C.c_int] + [C.POINTER(C.c_void_p)] * 4
        info = EgdstModelInfo()
        L.egdst_get_model_info(C.byref(info))
        self.info = info

    def check(self, rc):
        if rc != 0:
            raise EgdstRuntimeError(rc, (self.lib.egdst_last_error() or b'').decode(errors='replace'))

    def lerp_eval(self, x, g0, g1, f0, f1, shared=True):
        """linter's interpolation on the device: the grid kernels' shared-reciprocal form or the plain one (egdst_math_eval 3 / 4)."""
        xs = np.ascontiguousarray(np.concatenate([np.asarray(a, dtype=np.float64).ravel() for a in (x, g0, g1)]))
        ys = np.ascontiguousarray(np.concatenate([np.asarray(a, dtype=np.float64).ravel() for a in (f0, f1)]))
        n = xs.size // 3
        out = np.zeros(n)
        self.check(self.lib.egdst_math_eval(3 if shared else 4, n, _dp(xs), _dp(ys), _dp(out)))
        return out

    def math_eval(self, fn, x, y=None):
        """The device's exp ('exp'), log ('log'), pow ('pow') on host arrays (egdst_math_eval)."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.ascontiguousarray(x if y is None else y, dtype=np.float64)
        out = np.zeros_like(x)
        self.check(self.lib.egdst_math_eval({'exp': 0, 'log': 1, 'pow': 2}[fn], x.size, _dp(x), _dp(y), _dp(out)))
        return out


class Solution:
    """Host copy of one draw's solution in the table layout shared with the oracle harness."""

    def __init__(self, nt, nst, ngridmax, nthrhmax):
        self.nt, self.nst = nt, nst
        self.M = np.zeros((nt, nst, ngridmax + 1))
        self.C = np.zeros((nt, nst, ngridmax + 1))
        self.V = np.zeros((nt, nst, ngridmax + 1))
        self.D = np.zeros((nt, nst, nthrhmax))
        self.TH = np.zeros((nt, nst, nthrhmax))
        self.len = np.zeros((nt, nst), dtype=np.int32)
        self.thlen = np.zeros((nt, nst), dtype=np.int32)
        self.nevals = 0
        self.status = 0
        self.where = (0, 0)
        self.err = ''

    def cell_M(self, it, ist):
        n = self.len[it, ist]
        m, c, v = self.M[it, ist, :n], self.C[it, ist, :n], self.V[it, ist, :n]
        return np.stack([m, c, m - c, v], axis=1)

    def cell_D(self, it, ist):
        n = self.thlen[it, ist]
        return np.stack([self.D[it, ist, :n], self.TH[it, ist, :n]], axis=1)

    def total_rows(self):
        return int(self.len.sum())

    def cells(self):
        """(M, D) as nst x nt nested lists of matrices: the MATLAB cell arrays of solve (empty = None)."""
        M = [[self.cell_M(it, ist) if self.len[it, ist] else None for it in range(self.nt)] for ist in range(self.nst)]
        D = [[self.cell_D(it, ist) if self.len[it, ist] else None for it in range(self.nt)] for ist in range(self.nst)]
        return M, D


class Solver:
    """A device-resident batch of `ndraw` independent solves of one compiled model.

    rows_cap > 0 makes the handle COMPACT (include/egdst.h: egdst_create_compact): device lists and tables hold
    rows_cap rows instead of ngridmax, which keeps thousands of resident draws dense in memory.  A draw that needs
    more rows stops with EGDST_E_CAPACITY on the device; `sync` then solves exactly those draws again on a second,
    exact handle and every accessor below answers for them from there, so callers see the reference's semantics.
    """
    E_CAPACITY = 28

    def __init__(self, lib: ModelLibrary, desc: dict, ndraw=1, keep_history=True, stream=None, rows_cap=0):
        self.lib, self.ndraw, self.keep_history = lib, int(ndraw), bool(keep_history)
        self._quad = np.ascontiguousarray(desc['quadrature'], dtype=np.float64)
        ngridmax = desc['ngridmax'] if desc['ngridmax'] > desc['ngridm'] else 2 * desc['ngridm']
        self.desc = dict(desc, ngridmax=ngridmax)
        d = EgdstDesc(desc['t0'], desc['T'], desc['ngridm'], ngridmax, desc['nthrhmax'], desc['ny'],
                      desc['mmax'], desc['a0'], _dp(self._quad))
        self.nt = desc['T'] - desc['t0'] + 1
        self.h = C.c_void_p()
        self.rows_cap = int(rows_cap) if 0 < int(rows_cap) < ngridmax else 0
        self._exact = None          # exact handle for the draws that overflowed the compact one
        self._redo = np.zeros(0, dtype=np.int64)   # their draw indices, in the order of self._exact
        self.capacity_retries = 0   # draws solved again since creation
        lib.check(lib.lib.egdst_create_compact(C.byref(d), self.ndraw, int(self.keep_history), self.rows_cap,
                                               C.c_void_p(stream) if stream else None, C.byref(self.h)))

    def close(self):
        if self._exact is not None:
            self._exact.close()
            self._exact = None
        if self.h:
            self.lib.lib.egdst_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_groups(self, ngroups):
        """Split the draws into `ngroups` ranges that run on their own streams (egdst_set_groups)."""
        self.lib.check(self.lib.lib.egdst_set_groups(self.h, int(ngroups)))

    def set_adaptive(self, on=True):
        """History-based scheduling of straggler draws on lanes of their own (egdst_set_adaptive)."""
        self.lib.check(self.lib.lib.egdst_set_adaptive(self.h, int(bool(on))))

    def schedule(self):
        """(regular groups, straggler lanes, straggler draws) of the next solve"""
        a, b, c = C.c_int(0), C.c_int(0), C.c_int(0)
        self.lib.check(self.lib.lib.egdst_get_schedule(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def walk_stats(self):
        """[ndraw, 2]: envelope walks cut into segments and merged / fallen back to one wave (egdst_get_walk_stats)"""
        out = np.zeros((self.ndraw, 2), dtype=np.uint32)
        self.lib.check(self.lib.lib.egdst_get_walk_stats(self.h, out.ctypes.data_as(C.c_void_p)))
        return out

    def tp_stats(self):
        """[ndraw, 2]: cells of the last solve completed by the throughput path of the envelope step / left to k_envelope"""
        out = np.zeros((self.ndraw, 2), dtype=np.uint32)
        self.lib.check(self.lib.lib.egdst_get_tp_stats(self.h, out.ctypes.data_as(C.c_void_p)))
        return out

    def work(self):
        """re-basing calls per draw in the last solve (egdst_get_work)"""
        out = np.zeros(self.ndraw, dtype=np.uint32)
        self.lib.check(self.lib.lib.egdst_get_work(self.h, out.ctypes.data_as(C.c_void_p)))
        return out

    def regenerations(self):
        """regenerated guess streams per draw in the last solve (egdst_get_regenerations)"""
        out = np.zeros(self.ndraw, dtype=np.uint32)
        self.lib.check(self.lib.lib.egdst_get_regenerations(self.h, out.ctypes.data_as(C.c_void_p)))
        return out

    def geometry(self):
        """(physical rows per list, row stride of the device tables)"""
        a, b = C.c_int(0), C.c_int(0)
        self.lib.check(self.lib.lib.egdst_geometry(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def _route(self, draw):
        """(solver, local draw) that holds the results of `draw`"""
        if len(self._redo):
            j = np.nonzero(self._redo == draw)[0]
            if len(j):
                return self._exact, int(j[0])
        return self, draw

    def set_params(self, params):
        p = np.ascontiguousarray(np.asarray(params, dtype=np.float64).reshape(self.ndraw, -1))
        if p.shape[1] != self.lib.info.nparam:
            raise EgdstRuntimeError(1, 'expected %d parameters per draw, got %d' % (self.lib.info.nparam, p.shape[1]))
        self.lib.check(self.lib.lib.egdst_set_params(self.h, _dp(p), self.ndraw))

    def set_params_dev(self, dev_ptr):
        self.lib.check(self.lib.lib.egdst_set_params_dev(self.h, C.c_void_p(dev_ptr), self.ndraw))

    def solve_async(self):
        self.lib.check(self.lib.lib.egdst_solve_async(self.h))

    def sync(self, raise_on_error=True):
        rc = self.lib.lib.egdst_sync(self.h)
        self._redo = np.zeros(0, dtype=np.int64)
        if rc and self.rows_cap:
            rc = self._redo_overflowed()
        if rc and (raise_on_error or rc < 10):
            self.lib.check(rc)
        return rc

    def _redo_overflowed(self):
        """Solve the draws that stopped with EGDST_E_CAPACITY again with exact capacities; returns the first
        non-zero per-draw status of the batch after that (what egdst_sync would have returned)."""
        st, _ = self._status_raw()
        idx = np.nonzero(st == self.E_CAPACITY)[0]
        if len(idx):
            par = np.zeros((self.ndraw, max(self.lib.info.nparam, 1)))
            self.lib.check(self.lib.lib.egdst_get_params(self.h, _dp(par)))
            if self._exact is None or self._exact.ndraw != len(idx):
                if self._exact is not None:
                    self._exact.close()
                self._exact = Solver(self.lib, self.desc, ndraw=len(idx), keep_history=self.keep_history)
            self._exact.set_params(par[idx, :self.lib.info.nparam])
            self._exact.solve(raise_on_error=False)
            self._redo = idx.astype(np.int64)
            self.capacity_retries += len(idx)
        st, _ = self.status()
        bad = st[st != 0]
        return int(bad[0]) if len(bad) else 0

    def solve(self, raise_on_error=True):
        self.solve_async()
        return self.sync(raise_on_error)

    def _status_raw(self):
        st = np.zeros(self.ndraw, dtype=np.int32)
        wh = np.zeros(2 * self.ndraw, dtype=np.int32)
        self.lib.check(self.lib.lib.egdst_get_status(self.h, _ip(st), _ip(wh)))
        return st, wh.reshape(self.ndraw, 2)

    def status(self):
        st, wh = self._status_raw()
        if len(self._redo):
            st[self._redo], wh[self._redo] = self._exact.status()
        return st, wh

    def evals(self):
        tot = C.c_longlong(0)
        per = np.zeros(self.ndraw, dtype=np.int64)
        self.lib.check(self.lib.lib.egdst_get_evals(self.h, C.byref(tot), per.ctypes.data_as(C.POINTER(C.c_longlong))))
        if len(self._redo):
            per[self._redo] = self._exact.evals()[1]
            return int(per.sum()), per
        return int(tot.value), per

    def evals_credited(self):
        """per-draw evaluations that were accounted for without being executed (egdst_get_evals_credited)"""
        tot = C.c_longlong(0)
        per = np.zeros(self.ndraw, dtype=np.int64)
        self.lib.check(self.lib.lib.egdst_get_evals_credited(self.h, C.byref(tot), per.ctypes.data_as(C.POINTER(C.c_longlong))))
        if len(self._redo):
            per[self._redo] = self._exact.evals_credited()
        return per

    def objective(self):
        """[ndraw, 2] host array of the objective contributions (egdst_objective_dev), NaN for failed draws."""
        out = np.zeros((self.ndraw, 2))
        self.lib.check(self.lib.lib.egdst_get_objective(self.h, _dp(out)))
        if len(self._redo):
            out[self._redo] = self._exact.objective()
        return out

    def solution(self, draw=0):
        if len(self._redo):
            s_, j_ = self._route(draw)
            if s_ is not self:
                return s_.solution(j_)
        d = self.desc
        sol = Solution(self.nt, self.lib.info.nst, d['ngridmax'], d['nthrhmax'])
        self.lib.check(self.lib.lib.egdst_get_solution(self.h, draw, _ip(sol.len), _ip(sol.thlen), _dp(sol.M),
                                                       _dp(sol.C), _dp(sol.V), _dp(sol.D), _dp(sol.TH)))
        st, wh = self.status()
        sol.status, sol.where = int(st[draw]), tuple(int(x) for x in wh[draw])
        sol.err = self.lib.lib.egdst_strerror(sol.status).decode() if sol.status else ''
        sol.nevals = int(self.evals()[1][draw])
        return sol

    def set_solution(self, sol, draw=0):
        """Import a host solution (the arrays `solution()` returns) into `draw`: egdst_set_solution, the inverse of the
        export.  The handle then simulates / evaluates `call` from it without having solved (egdst_simulator.c:61-68)."""
        ln = np.ascontiguousarray(sol.len, dtype=np.int32)
        th = np.ascontiguousarray(sol.thlen, dtype=np.int32)
        arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (sol.M, sol.C, sol.V, sol.D, sol.TH)]
        d = self.desc
        if arrs[0].shape != (self.nt, self.lib.info.nst, d['ngridmax'] + 1) or arrs[3].shape != (self.nt, self.lib.info.nst, d['nthrhmax']):
            raise EgdstRuntimeError(1, 'set_solution: the arrays do not have the shapes of this handle')
        self.lib.check(self.lib.lib.egdst_set_solution(self.h, draw, _ip(ln), _ip(th), *[_dp(a) for a in arrs]))

    def set_cells(self, M, D, draw=0):
        """Import the MATLAB-style cell arrays (nst x nt nested lists of (len x 4) [M C A V] and (thlen x 2) [D TH]
        matrices, None / empty = unsolved cell) cell by cell: egdst_set_cell_M / egdst_set_cell_D, what a MEX shim of the
        simulator gateway does with mxGetCell (shims/egdst_simulator_hip.c)."""
        for ist in range(self.lib.info.nst):
            for it in range(self.nt):
                m = None if M[ist][it] is None else np.asfortranarray(np.asarray(M[ist][it], dtype=np.float64))
                dd = None if D[ist][it] is None else np.asfortranarray(np.asarray(D[ist][it], dtype=np.float64))
                nm = 0 if m is None or m.size == 0 else m.shape[0]
                nd_ = 0 if dd is None or dd.size == 0 else dd.shape[0]
                if nm and m.shape[1] != 4 or nd_ and dd.shape[1] != 2:
                    raise EgdstRuntimeError(1, 'set_cells: cell (%d, %d) is not (len x 4) / (thlen x 2)' % (ist, it))
                self.lib.check(self.lib.lib.egdst_set_cell_M(self.h, draw, it, ist, nm, _dp(m) if nm else None))
                self.lib.check(self.lib.lib.egdst_set_cell_D(self.h, draw, it, ist, nd_, _dp(dd) if nd_ else None))

    def set_dbgout(self, on=True):
        """keep the kink log of the next solves (third output of the solver gateway)"""
        self.lib.check(self.lib.lib.egdst_set_dbgout(self.h, int(bool(on))))

    def dbgout(self, draw=0):
        """(matrix [cap, 7] as the gateway returns it, number of recorded rows)"""
        info = self.lib.info
        cap = self.nt * info.nst * info.nd * 2 * self.nt
        out = np.zeros((cap, 7), order='F')
        n = C.c_int(0)
        self.lib.check(self.lib.lib.egdst_get_dbgout(self.h, draw, _dp(out), C.byref(n)))
        return out, n.value

    def checksums(self, draw=0):
        """[nt, nst, 5] uint64 checksums of the draw's cells (columns M, C, V, TH, D), computed on the device."""
        if len(self._redo) and self._route(draw)[0] is not self:
            s_, j_ = self._route(draw)
            return s_.checksums(j_)
        out = np.zeros((self.nt, self.lib.info.nst, 5), dtype=np.uint64)
        self.lib.check(self.lib.lib.egdst_get_checksums(self.h, draw, out.ctypes.data_as(C.c_void_p)))
        return out

    def dims(self, draw=0):
        """(len, thlen) [nt, nst] of the draw's cells without copying the tables."""
        ln = np.zeros((self.nt, self.lib.info.nst), dtype=np.int32)
        th = np.zeros((self.nt, self.lib.info.nst), dtype=np.int32)
        self.lib.check(self.lib.lib.egdst_get_solution(self.h, draw, _ip(ln), _ip(th), None, None, None, None, None))
        return ln, th

    def objective_dev(self, dev_ptr):
        """Device-side objective of THIS handle's draws (draws redone after a capacity overflow read NaN here;
        use objective() when rows_cap is set and capacity_retries may be non-zero)."""
        self.lib.check(self.lib.lib.egdst_objective_dev(self.h, C.c_void_p(dev_ptr)))

    def set_profile(self, on=True):
        self.lib.check(self.lib.lib.egdst_set_profile(self.h, int(on)))

    PROFILE_CLASSES = ['probe', 'grid', 'k_envelope', 'regeneration', 'tp_prep', 'tp_sort0', 'tp_sort1', 'tp_walk0', 'tp_walk1']

    def profile(self):
        """(ms[9], launches[9], algorithmic bytes) of the last solve by kernel class (PROFILE_CLASSES; include/egdst.h)."""
        ms = np.zeros(9)
        ln = np.zeros(9, dtype=np.int32)
        ab = C.c_longlong(0)
        self.lib.check(self.lib.lib.egdst_get_profile(self.h, _dp(ms), _ip(ln), C.byref(ab)))
        return ms, ln, int(ab.value)

    def group_profile(self):
        """(finish[g], class_ms[g, 9]) of the last solve with profiling on: ms from the first group's first kernel to the end of group
        g's last kernel, and the group's summed HIP-event time per kernel class (egdst_get_group_profile)"""
        fin = np.zeros(32)
        cls = np.zeros((32, 9))
        self.lib.lib.egdst_get_group_profile.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        n = self.lib.lib.egdst_get_group_profile(self.h, 32, _dp(fin), _dp(cls))
        return fin[:n], cls[:n]

    def debug(self, draw=0):
        out = np.zeros(16, dtype=np.int32)
        self.lib.check(self.lib.lib.egdst_get_debug(self.h, draw, _ip(out)))
        return out

    def cell_M(self, draw, it, ist):
        if len(self._redo) and self._route(draw)[0] is not self:
            s_, j_ = self._route(draw)
            return s_.cell_M(j_, it, ist)
        n, nth = C.c_int(0), C.c_int(0)
        self.lib.check(self.lib.lib.egdst_cell_dims(self.h, draw, it, ist, C.byref(n), C.byref(nth)))
        out = np.zeros((4, n.value))
        if n.value:
            self.lib.check(self.lib.lib.egdst_get_cell_M(self.h, draw, it, ist, _dp(out)))
        return np.ascontiguousarray(out.T)  # column-major (len x 4) -> rows

    def cell_D(self, draw, it, ist):
        if len(self._redo) and self._route(draw)[0] is not self:
            s_, j_ = self._route(draw)
            return s_.cell_D(j_, it, ist)
        n, nth = C.c_int(0), C.c_int(0)
        self.lib.check(self.lib.lib.egdst_cell_dims(self.h, draw, it, ist, C.byref(n), C.byref(nth)))
        out = np.zeros((2, nth.value))
        if nth.value:
            self.lib.check(self.lib.lib.egdst_get_cell_D(self.h, draw, it, ist, _dp(out)))
        return np.ascontiguousarray(out.T)

    def simulate_moments(self, init, randstream, rndtype=0, draw=0):
        """(means [nt, nout], counts [nt, nout]) of the simulated paths over the agents that have a value in a period;
        the paths themselves stay on the device (egdst_simulate_moments)."""
        if len(self._redo) and self._route(draw)[0] is not self:
            s_, j_ = self._route(draw)
            return s_.simulate_moments(init, randstream, rndtype, j_)
        init = np.asfortranarray(np.atleast_2d(np.asarray(init, dtype=np.float64)))
        info = self.lib.info
        nout = 11 + info.nnst + info.nnd + info.neq
        means = np.zeros((self.nt, nout))
        counts = np.zeros((self.nt, nout), dtype=np.int32)
        rs = np.ascontiguousarray(randstream, dtype=np.float64)
        self.lib.check(self.lib.lib.egdst_simulate_moments(self.h, draw, _dp(init), init.shape[0], _dp(rs), rs.size, int(rndtype),
                                                           _dp(means), _ip(counts)))
        return means, counts

    def simulate_batch_moments(self, init, seed=0, rndtype=0, target=None, weight=None, randstream_dev=None, nrand=0,
                               means_dev=None, counts_dev=None, obj_dev=None):
        """egdst_simulate_batch_moments: all draws of the handle, common random numbers; the *_dev arguments are device
        pointers (ints).  With no device pointers given, returns (means [ndraw, nt, nout], counts, objective [ndraw]) through
        torch tensors allocated here."""
        init = np.asfortranarray(np.atleast_2d(np.asarray(init, dtype=np.float64)))
        info = self.lib.info
        nout = 11 + info.nnst + info.nnd + info.neq
        ncell = nout * self.nt
        t = np.ascontiguousarray(np.zeros(ncell) if target is None else target, dtype=np.float64).reshape(-1)
        w = np.ascontiguousarray(np.zeros(ncell) if weight is None else weight, dtype=np.float64).reshape(-1)
        own = means_dev is None and counts_dev is None and obj_dev is None
        if own:
            import torch
            tm = torch.zeros(self.ndraw, self.nt, nout, dtype=torch.float64, device='cuda')
            tc = torch.zeros(self.ndraw, self.nt, nout, dtype=torch.int32, device='cuda')
            to = torch.zeros(self.ndraw, dtype=torch.float64, device='cuda')
            # torch fills them on ITS stream; the library writes them on the handle's: without this the fill may land after the
            # results (seen once in a few hundred runs: all moments and objectives zero)
            torch.cuda.current_stream().synchronize()
            means_dev, counts_dev, obj_dev = tm.data_ptr(), tc.data_ptr(), to.data_ptr()
        self.lib.check(self.lib.lib.egdst_simulate_batch_moments(
            self.h, _dp(init), init.shape[0], C.c_void_p(randstream_dev) if randstream_dev else None, int(nrand), int(seed),
            int(rndtype), _dp(t), _dp(w), C.c_void_p(means_dev) if means_dev else None,
            C.c_void_p(counts_dev) if counts_dev else None, C.c_void_p(obj_dev) if obj_dev else None))
        if own:
            return tm.cpu().numpy(), tc.cpu().numpy(), to.cpu().numpy()
        return None

    def call(self, sw, args, draw=0):
        """egdst_call gateway (egdst_call.c:17-164): sw 1 utility, 2 marginal utility, 3 discount, 4 budget, 5 marginal
        budget, 6 value function; args [narg x ncol] with 1-based it/ist/id as in MATLAB.  Returns [narg]."""
        if len(self._redo) and self._route(draw)[0] is not self:
            s_, j_ = self._route(draw)
            return s_.call(sw, args, j_)
        a = np.asfortranarray(np.atleast_2d(np.asarray(args, dtype=np.float64)))
        res = np.zeros(a.shape[0])
        self.lib.check(self.lib.lib.egdst_call(self.h, draw, int(sw), a.shape[0], a.shape[1], _dp(a), _dp(res)))
        return res

    def simulate(self, init, randstream, rndtype=0, draw=0):
        if len(self._redo) and self._route(draw)[0] is not self:
            s_, j_ = self._route(draw)
            return s_.simulate(init, randstream, rndtype, j_)
        init = np.asfortranarray(np.atleast_2d(np.asarray(init, dtype=np.float64)))
        nsim = init.shape[0]
        info = self.lib.info
        nout = 11 + info.nnst + info.nnd + info.neq
        sims = np.zeros((nsim, self.nt, nout))  # C order == column-major [nout x nt x nsim]
        rs = np.ascontiguousarray(randstream, dtype=np.float64)
        self.lib.check(self.lib.lib.egdst_simulate(self.h, draw, _dp(init), nsim, _dp(rs), rs.size, int(rndtype),
                                                   _dp(sims)))
        return sims


def solve_model(model, dbgout=False):
    """model.solve(): one draw with the model's current parameters, history kept for sim/export."""
    desc = model.descriptor()
    if model.__dict__.get('_solver') is not None:
        model._solver.close()
    s = Solver(model._lib, desc, ndraw=1, keep_history=True)
    model.__dict__['_solver'] = s
    s.set_params(model.param_vector())
    if dbgout:
        s.set_dbgout(True)
    rc = s.solve(raise_on_error=False)
    sol = s.solution(0)
    if dbgout:
        sol.dbgout, sol.dbgn = s.dbgout(0)
    if rc:
        # the reference warns and returns the partially filled cells (egdst_solver.c:237)
        import warnings
        warnings.warn(sol.err)
    return sol


def _cells_stamp(M, D):
    """A cheap fingerprint of the cell arrays (shape and the bytes of the first, middle and last entry of every cell), so that a cell
    REPLACED inside model.M / model.D -- which keeps the outer objects -- is noticed and the cells are uploaded again: the reference
    reads them from the model object on every call.  (Bytes, not floats: a NaN entry must compare equal to itself.)"""
    out = []
    for cells in (M, D):
        for row in cells:
            for c in row:
                if c is None:
                    out.append(None)
                    continue
                a = np.asarray(c)
                n = a.size
                out.append((a.shape, a.flat[[0, n // 2, n - 1]].tobytes() if n else b''))
    return tuple(out)


def _solver_with_solution(model):
    """The handle that serves `sim` and `call`.  The reference's gateways take the solution from the model object on every
    call (egdst_simulator.c:61-68, egdst_call.c:28-34), not from the solve that produced it: when the handle of the last
    `solve` still holds exactly model.M / model.D it is reused (nothing to upload), otherwise -- the cells were assigned,
    e.g. restored from a saved model -- a fresh handle imports them (egdst_set_cell_M / egdst_set_cell_D)."""
    if model.M is None or model.D is None:
        raise EgdstRuntimeError(40, 'Error: the model has not yet been solved!')
    s = model.__dict__.get('_solver')
    held = model.__dict__.get('_solver_cells')   # the very objects the live handle holds (strong references: an id can be reused)
    if s is not None and held is not None and held[0] is model.M and held[1] is model.D and held[2] == _cells_stamp(model.M, model.D):
        s.set_params(model.param_vector())   # loadparameters() of the gateways: the CURRENT values (after a setparam too)
        return s
    if s is not None:
        s.close()
    s = Solver(model._lib, model.descriptor(), ndraw=1, keep_history=True)
    s.set_params(model.param_vector())   # (loadparameters() of the gateways: the CURRENT parameter values)
    s.set_cells(model.M, model.D, draw=0)
    model.__dict__['_solver'] = s
    model.__dict__['_solver_cells'] = (model.M, model.D, _cells_stamp(model.M, model.D))
    return s


def call_model(model, sw, args):
    return _solver_with_solution(model).call(sw, args, draw=0)


def simulate_model(model, rndtype):
    return _solver_with_solution(model).simulate(model.init, model.randstream, rndtype, draw=0)
