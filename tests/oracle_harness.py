"""ctypes harness around the CPU oracle (oracle/egdst_oracle.c).  Test infrastructure only."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))

from egdst_amd import codegen  # noqa: E402
import build_oracle  # noqa: E402


class OrcDesc(C.Structure):
    _fields_ = [('t0', C.c_int), ('T', C.c_int), ('ngridm', C.c_int), ('ngridmax', C.c_int),
                ('nthrhmax', C.c_int), ('ny', C.c_int), ('mmax', C.c_double), ('a0', C.c_double),
                ('quadrature', C.POINTER(C.c_double))]


class OrcSolution(C.Structure):
    _fields_ = [('M', C.POINTER(C.c_double)), ('C', C.POINTER(C.c_double)), ('V', C.POINTER(C.c_double)),
                ('D', C.POINTER(C.c_double)), ('TH', C.POINTER(C.c_double)), ('len', C.POINTER(C.c_int)),
                ('thlen', C.POINTER(C.c_int)), ('nevals', C.c_longlong), ('err', C.c_char * 300),
                ('dbgout', C.POINTER(C.c_double)), ('dbgcap', C.c_int), ('dbgn', C.c_int)]


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class OracleSolution:
    """Host copy of a solved model in the oracle's (and the product's) table layout."""

    def __init__(self, nt, nst, ngridmax, nthrhmax):
        self.nt, self.nst, self.stride, self.nthrhmax = nt, nst, ngridmax + 1, nthrhmax
        self.M = np.zeros((nt, nst, ngridmax + 1))
        self.C = np.zeros((nt, nst, ngridmax + 1))
        self.V = np.zeros((nt, nst, ngridmax + 1))
        self.D = np.zeros((nt, nst, nthrhmax))
        self.TH = np.zeros((nt, nst, nthrhmax))
        self.len = np.zeros((nt, nst), dtype=np.int32)
        self.thlen = np.zeros((nt, nst), dtype=np.int32)
        self.nevals = 0
        self.err = ''

    def cell_M(self, it, ist):
        """(len x 4) matrix [M C A V] as saveoutput builds it (egdst_solver.c:917-941)."""
        n = self.len[it, ist]
        m, c, v = self.M[it, ist, :n], self.C[it, ist, :n], self.V[it, ist, :n]
        return np.stack([m, c, m - c, v], axis=1)

    def cell_D(self, it, ist):
        n = self.thlen[it, ist]
        return np.stack([self.D[it, ist, :n], self.TH[it, ist, :n]], axis=1)

    def total_rows(self):
        return int(self.len.sum())


class Oracle:
    def __init__(self, model, build_dir=None, native_math=False):
        """native_math=True reproduces the reference's glibc arithmetic (known-answer tests); the default uses
        the bit-reproducible functions of include/egdst_math.h, exactly as the GPU path does."""
        text = codegen.generate_modelspec(model)
        tag = ''.join(ch for ch in model.label if ch.isalnum())[:16] + '_' + codegen.spec_hash(text)
        d = build_dir or os.path.join(ROOT, 'oracle', '_build', tag)
        os.makedirs(d, exist_ok=True)
        spec = os.path.join(d, 'modelspec.h')
        if not os.path.exists(spec) or open(spec).read() != text:
            with open(spec, 'w') as f:
                f.write(text)
        self.lib = C.CDLL(build_oracle.build(d, native_math=native_math))
        self.model = model
        self.lib.egdst_oracle_solve.restype = C.c_int
        self.lib.egdst_oracle_sim.restype = C.c_int
        info = (C.c_int * 6)()
        self.lib.egdst_oracle_info(info)
        self.nst, self.nd, self.nnst, self.nnd, self.nparam, self.neq = list(info)

    def _desc(self):
        d = self.model.descriptor()
        self._quad = np.ascontiguousarray(d['quadrature'], dtype=np.float64)
        return OrcDesc(d['t0'], d['T'], d['ngridm'], d['ngridmax'], d['nthrhmax'], d['ny'], d['mmax'], d['a0'],
                       _dp(self._quad)), d

    def solve(self, params=None, dbgout=False):
        """dbgout=True: also the third output of the solver gateway, sol.dbgout [cap x 7] (zero rows past sol.dbgn)."""
        desc, d = self._desc()
        nt = d['T'] - d['t0'] + 1
        sol = OracleSolution(nt, self.nst, d['ngridmax'], d['nthrhmax'])
        par = np.ascontiguousarray(self.model.param_vector() if params is None else params, dtype=np.float64)
        cs = OrcSolution(_dp(sol.M), _dp(sol.C), _dp(sol.V), _dp(sol.D), _dp(sol.TH),
                         sol.len.ctypes.data_as(C.POINTER(C.c_int)), sol.thlen.ctypes.data_as(C.POINTER(C.c_int)))
        if dbgout:
            cap = nt * self.nst * self.nd * 2 * nt                      # egdst_solver.c:178
            dbg = np.zeros((cap, 7), order='F')
            cs.dbgout, cs.dbgcap = _dp(dbg), cap
        rc = self.lib.egdst_oracle_solve(C.byref(desc), _dp(par), C.byref(cs))
        if dbgout:
            sol.dbgout, sol.dbgn = dbg, int(cs.dbgn)
        sol.nevals = int(cs.nevals)
        sol.err = cs.err.decode(errors='replace')
        sol.rc = rc
        return sol

    def call(self, sol, sw, args, params=None):
        desc, d = self._desc()
        a = np.asfortranarray(np.atleast_2d(np.asarray(args, dtype=np.float64)))
        res = np.zeros(a.shape[0])
        par = np.ascontiguousarray(self.model.param_vector() if params is None else params, dtype=np.float64)
        cs = OrcSolution(_dp(sol.M), _dp(sol.C), _dp(sol.V), _dp(sol.D), _dp(sol.TH),
                         sol.len.ctypes.data_as(C.POINTER(C.c_int)), sol.thlen.ctypes.data_as(C.POINTER(C.c_int)))
        self.lib.egdst_oracle_call.restype = C.c_int
        rc = self.lib.egdst_oracle_call(C.byref(desc), _dp(par), C.byref(cs), C.c_int(int(sw)), C.c_int(a.shape[0]),
                                        C.c_int(a.shape[1]), _dp(a), _dp(res))
        if rc != 0:
            raise RuntimeError('oracle call failed rc=%d' % rc)
        return res

    def sim(self, sol, init, randstream, rndtype=0, params=None):
        desc, d = self._desc()
        nt = d['T'] - d['t0'] + 1
        init = np.asfortranarray(np.atleast_2d(np.asarray(init, dtype=np.float64)))
        nsim = init.shape[0]
        nout = 11 + self.nnst + self.nnd + self.neq
        sims = np.zeros((nsim, nt, nout))  # C-order [nsim][nt][nout] == column-major [nout x nt x nsim]
        par = np.ascontiguousarray(self.model.param_vector() if params is None else params, dtype=np.float64)
        rs = np.ascontiguousarray(randstream, dtype=np.float64)
        cs = OrcSolution(_dp(sol.M), _dp(sol.C), _dp(sol.V), _dp(sol.D), _dp(sol.TH),
                         sol.len.ctypes.data_as(C.POINTER(C.c_int)), sol.thlen.ctypes.data_as(C.POINTER(C.c_int)))
        rc = self.lib.egdst_oracle_sim(C.byref(desc), _dp(par), C.byref(cs), _dp(init), C.c_int(nsim), _dp(rs),
                                       C.c_longlong(rs.size), C.c_int(rndtype), _dp(sims))
        if rc != 0:
            raise RuntimeError('oracle sim failed rc=%d' % rc)
        return sims
