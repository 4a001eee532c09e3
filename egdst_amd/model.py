"""Host-side mirror of the reference's ``@egdstmodel`` class surface.

The reference host language is MATLAB (absent in this image), so the host side above
the C-ABI is written in Python and keeps the reference's names, argument meaning and
error behaviour (class header: @egdstmodel/egdstmodel.m:1-351):

    m = egdstmodel('retire2')
    m.t0 = 1; m.T = 25; m.mmax = 10; m.ngridm = 100; ...
    m.s = ('Singleton state', [0, 'dummy state'])       # MATLAB: m.s={'..',{0,'dummy state'}}
    m.trpr = ('true', [[1]])
    m.u = ('utility', 'log(consumption)+duw*(id==0)')
    m.param = ('duw', 'disutility of work', 0.5)
    m.compile(); m.solve(); m.sim([[1, 0.25]])

Assignments to the DSL properties *append* like the MATLAB ``set.*`` methods do
(egdstmodel.m:573-1024); assigning ``None`` clears.  ``compile`` generates the model
plugin (``modelspec.h``, the counterpart of compile.m's modelspec.c/.h) and builds the
per-model HIP library; ``solve``/``sim`` call through the C-ABI of ``include/egdst.h``.
There is no CPU fallback: without the HIP library ``solve`` raises.
"""
from __future__ import annotations

import math
import re
from dataclasses import dataclass, field
from typing import Any, List, Optional, Sequence

import numpy as np

from .quadrature import quadrature_array

GLOBALS = ['t0', 'T', 'ngridm', 'ngridmax', 'nthrhmax', 'ny', 'nd', 'nnd', 'nst', 'nnst', 'mmax', 'a0']
RESERVED = (['age', 'ist', 'id1', 'id'] + ['st%d' % i for i in range(1, 16)] +
            ['st%dgrid' % i for i in range(1, 16)] + ['dc%d' % i for i in range(1, 16)] +
            ['consumption', 'mutility', 'discount', 'survival', 'utility', 'utility_marginal',
             'utility_marginal_inverse', 'trpr', 'feasible', 'inchoiceset', 'cashinhand',
             'cashinhand_marginal', 'shock', 'mu_param', 'sigma_param', 'sigma', 'mu', 'cash',
             'savings', 'min', 'max'])
SIMLABELS_DEFAULT = ['1  cash-in-hand (M)', '2  optimal consumption (C)', '3  optimal saving (A)',
                     '4  value function', '5  current discrete decision index (id)',
                     '6  current period state index (ist)', '7  mu parameter of shock distribution',
                     '8  sigma parameter of shock distribution', '9  income shock', '10 utility',
                     '11 discount factor']


class EgdstError(Exception):
    """Counterpart of MATLAB ``error('egdstmodel:...')``."""


@dataclass
class _Var:
    index: int
    name: str
    type: str
    values: List[float]
    descriptions: List[str]
    gridlimits: Optional[Sequence[float]] = None
    gridpoints: Optional[int] = None

    @property
    def discrete(self):
        return self.type == 'discrete'

    @property
    def continuous(self):
        return self.type == 'continuous'


@dataclass
class _Case:
    condition: str
    prob: Any  # list-of-lists of exec strings (discrete) or one exec string (continuous)


@dataclass
class _Trpr:
    varindex: int
    cases: List[_Case] = field(default_factory=list)


@dataclass
class _Eq:
    ref: str
    type: str
    expression: Any  # str or list of lines
    description: str


@dataclass
class _Coef:
    ref: str
    array: np.ndarray
    description: str


@dataclass
class _Param:
    ref: str
    description: str
    value: float


def _num2execstr(v):
    # set.discount / set.survival / set.shock with numeric value (egdstmodel.m:546-572, 791-797)
    return '0.0' if v == 0 else '%25.25f' % v


def _stepmult(a):
    # egdstmodel.m:1432-1437
    return [int(np.prod(a[i + 1:])) for i in range(len(a))]


class egdstmodel:  # noqa: N801  (reference class name)
    def __init__(self, label='<no name>', dirname=None):
        d = self.__dict__
        d['label'] = label
        d['dir'] = dirname
        d['t0'] = float('nan')
        d['T'] = float('nan')
        d['_s'] = []
        d['_d'] = []
        d['mmax'] = float('nan')
        d['_ngridm'] = 10
        d['_ngridmax'] = 100
        d['nthrhmax'] = 100
        d['ny'] = 1
        d['a0'] = 0.0
        d['_discount'] = ''
        d['_survival'] = '1.0'
        d['_u'] = {}
        d['_transform'] = {'direct': 'log(x+1)', 'inverse': 'exp(x)-1'}
        d['_budget'] = {}
        d['_shock'] = {'type': 'lognormal'}
        d['_trpr'] = []
        d['_choiceset'] = {'defaultallow': True, 'rules': []}
        d['_feasible'] = {'defaultfeasible': True, 'rules': []}
        d['_eq'] = []
        d['_coef'] = []
        d['_param'] = []
        d['_cflags'] = {'TOLERANCE': '1e-10', 'ZEROCONSUMPTION': '1e-10',
                        'DOUBLEPOINT_DELTA': '1e-10', 'VERBOSE': '0'}
        d['nd'] = 0
        d['nnd'] = 0
        d['nst'] = 0
        d['nnst'] = 0
        d['stm'] = []
        d['dm'] = []
        d['M'] = None
        d['D'] = None
        d['states'] = np.zeros((0, 0))
        d['decisions'] = np.zeros((0, 0))
        d['sims'] = None
        d['simlabels'] = []
        d['init'] = None
        d['randstream'] = None
        d['optim'] = {'optim_MUnoD': False, 'optim_UnoD': False, 'optim_UasD': False,
                      'optim_TRPRnoSH': False}
        d['quadrature'] = None
        d['quiet'] = True
        d['needtocompile'] = True
        d['lastrun_solver'] = None
        d['_lib'] = None       # loaded per-model HIP library (runtime.ModelLibrary)
        d['_solution'] = None  # device-resident solution handle

    # ------------------------------------------------------------------ simple props
    @property
    def nt(self):
        res = int(self.T) - int(self.t0) + 1
        if res < 1:
            raise EgdstError('Error: t0>T!')
        return res

    @property
    def ngridm(self):
        return self._ngridm

    @ngridm.setter
    def ngridm(self, v):  # egdstmodel.m:532-538
        self.__dict__['_ngridm'] = int(v)
        if self._ngridm * 1.5 > self._ngridmax:
            self.__dict__['_ngridmax'] = 2 * self._ngridm

    @property
    def ngridmax(self):
        return self._ngridmax

    @ngridmax.setter
    def ngridmax(self, v):  # egdstmodel.m:539-545
        self.__dict__['_ngridmax'] = int(v)
        if self._ngridm * 1.5 > self._ngridmax:
            self.__dict__['_ngridmax'] = 2 * self._ngridm

    @property
    def cflags(self):
        return self._cflags

    @cflags.setter
    def cflags(self, v):
        self.__dict__['_cflags'] = dict(v)
        self.needtocompile = True

    @property
    def discount(self):
        return self._discount

    @discount.setter
    def discount(self, v):
        self.__dict__['_discount'] = v if isinstance(v, str) else _num2execstr(float(v))
        self.needtocompile = True

    @property
    def survival(self):
        return self._survival

    @survival.setter
    def survival(self, v):
        self.__dict__['_survival'] = v if isinstance(v, str) else _num2execstr(float(v))
        self.needtocompile = True

    # ------------------------------------------------------------------ states / decisions
    def _add_var(self, lst, value, what):
        if (isinstance(value, (tuple, list)) and len(value) == 2 and isinstance(value[0], str)
                and isinstance(value[1], (list, tuple)) and len(value[1]) > 0
                and any(isinstance(x, str) for x in value[1])):
            pairs = list(value[1])
            if len(pairs) % 2:
                raise EgdstError('Unrecognized structure for %s variable!' % what)
            vals, descr = [], []
            for i in range(0, len(pairs), 2):
                if isinstance(pairs[i], str) or not np.isscalar(pairs[i]):
                    raise EgdstError('Non-numeric value of the %s variable detected!..' % what)
                vals.append(float(pairs[i]))
                descr.append(str(pairs[i + 1]))
            lst.append(_Var(len(lst) + 1, value[0], 'discrete', vals, descr))
        elif (isinstance(value, (tuple, list)) and len(value) == 2 and isinstance(value[0], str)):
            vals = [float(x) for x in np.atleast_1d(value[1])]
            lst.append(_Var(len(lst) + 1, value[0], 'discrete', vals, ['value %1.3f' % x for x in vals]))
        elif (isinstance(value, (tuple, list)) and len(value) == 3 and isinstance(value[0], str)
              and len(np.atleast_1d(value[1])) == 2 and np.isscalar(value[2])):
            # name + grid limits + number of grid points: a CONTINUOUS variable on a linspace grid whose points double
            # as its "values" (egdstmodel.m:628-646); decisions of this type are not part of the hot path
            if what != 'state':
                raise EgdstError('Continuous decision variables are not implemented')
            lim = [float(x) for x in np.atleast_1d(value[1])]
            n = int(value[2])
            if n < 2:
                raise EgdstError('A continuous state variable needs at least two grid points')
            grid = [float(x) for x in np.linspace(lim[0], lim[1], n)]
            lst.append(_Var(len(lst) + 1, value[0], 'continuous', grid, ['grid point'] * n, lim, n))
        else:
            raise EgdstError('Unrecognized structure for %s variable!' % what)

    @staticmethod
    def _buildstates(vars_):
        # egdstmodel.m:1439-1461: Cartesian product, first variable slowest
        if not vars_:
            return np.zeros((0, 0))
        grids = np.meshgrid(*[np.asarray(v.values, dtype=float) for v in vars_], indexing='ij')
        return np.stack([g.reshape(-1) for g in grids], axis=1)

    @property
    def s(self):
        return self._s

    @s.setter
    def s(self, value):
        self.needtocompile = True
        if value is None or (hasattr(value, '__len__') and len(value) == 0):
            self.__dict__['_s'] = []
            self.__dict__['_trpr'] = []
            self.nnst = 0
            return
        self._add_var(self._s, value, 'state')
        self.nnst = len(self._s)
        sizes = [len(v.values) for v in self._s]
        self.stm = sizes + _stepmult(sizes)  # egdstmodel.m:651
        self.nst = int(np.prod(sizes))
        self.states = self._buildstates(self._s)

    @property
    def d(self):
        return self._d

    @d.setter
    def d(self, value):
        self.needtocompile = True
        if value is None or (hasattr(value, '__len__') and len(value) == 0):
            self.__dict__['_d'] = []
            self.nnd = 0
            return
        self._add_var(self._d, value, 'decision')
        self.nnd = len(self._d)
        sizes = [len(v.values) for v in self._d]
        self.dm = sizes + _stepmult(sizes)
        self.nd = int(np.prod(sizes))
        self.decisions = self._buildstates(self._d)

    # ------------------------------------------------------------------ exec-string parts
    def _set_keyed(self, store, value, keys, what):
        self.needtocompile = True
        if value is None:
            store.clear()
            return
        if (isinstance(value, (tuple, list)) and len(value) == 2 and value[0] in keys
                and isinstance(value[1], (str, list, tuple))):
            store[value[0]] = value[1] if isinstance(value[1], str) else list(value[1])
        else:
            raise EgdstError('Unrecognized structure for %s definition!' % what)

    @property
    def u(self):
        return self._u

    @u.setter
    def u(self, value):
        self._set_keyed(self._u, value, ('utility', 'marginal', 'marginalinverse', 'extrap'), 'utility')

    @property
    def budget(self):
        return self._budget

    @budget.setter
    def budget(self, value):
        self._set_keyed(self._budget, value, ('cashinhand', 'marginal'), 'budget')

    @property
    def transform(self):
        return self._transform

    @transform.setter
    def transform(self, value):
        self.needtocompile = True
        if value is None:
            self.__dict__['_transform'] = {'direct': 'log(x+1)', 'inverse': 'exp(x)-1'}
        elif (isinstance(value, (tuple, list)) and len(value) == 2
              and all(isinstance(x, str) for x in value)):
            self.__dict__['_transform'] = {'direct': value[0], 'inverse': value[1]}
        else:
            raise EgdstError('Unrecognized structure for transformation function!')

    @property
    def shock(self):
        return self._shock

    @shock.setter
    def shock(self, value):
        self.needtocompile = True
        if value is None:
            self.__dict__['_shock'] = {'type': 'lognormal'}
        elif isinstance(value, str) and value in ('lognormal', 'normal'):
            self._shock['type'] = value
        elif (isinstance(value, (tuple, list)) and len(value) == 2 and value[0] in ('mu', 'sigma')):
            v = value[1]
            self._shock[value[0]] = v if isinstance(v, (str, list, tuple)) else _num2execstr(float(v))
        else:
            raise EgdstError('Unrecognized structure for shock definition!')

    @property
    def eq(self):
        return self._eq

    @eq.setter
    def eq(self, value):
        self.needtocompile = True
        if value is None:
            self.__dict__['_eq'] = []
            return
        if not (isinstance(value, (tuple, list)) and len(value) in (3, 4)
                and isinstance(value[0], str) and isinstance(value[1], str)):
            raise EgdstError('Unrecognized structure for equation definition!')
        typ = value[3] if len(value) == 4 else 'current'
        if typ not in ('current', 'next'):
            raise EgdstError('Unrecognized structure for equation definition!')
        expr = value[2] if isinstance(value[2], str) else list(value[2])
        new = _Eq(value[0], typ, expr, value[1])
        for i, e in enumerate(self._eq):
            if e.ref == new.ref:
                self._eq[i] = new  # replace (egdstmodel.m:817-829)
                break
        else:
            self._eq.append(new)
        self._checkrefs()

    @property
    def coef(self):
        return self._coef

    @coef.setter
    def coef(self, value):
        self.needtocompile = True
        if value is None:
            self.__dict__['_coef'] = []
            return
        if not (isinstance(value, (tuple, list)) and len(value) == 3 and isinstance(value[0], str)
                and isinstance(value[1], str)):
            raise EgdstError('Unrecognized structure for coefficient definition!')
        self._coef.append(_Coef(value[0], np.atleast_2d(np.asarray(value[2], dtype=float)), value[1]))
        self._checkrefs()

    @property
    def param(self):
        return self._param

    @param.setter
    def param(self, value):
        # parameters are run-time inputs: no recompile (compile.m:469-475)
        if value is None:
            self.__dict__['_param'] = []
            return
        if not (isinstance(value, (tuple, list)) and len(value) == 3 and isinstance(value[0], str)
                and isinstance(value[1], str) and np.isscalar(value[2])):
            raise EgdstError("Unrecognized structure for parameter definition! Need ('ref','description',start value)!")
        self._param.append(_Param(value[0], value[1], float(value[2])))
        self.needtocompile = True  # a *new* parameter changes the generated code
        self._checkrefs()

    @property
    def choiceset(self):
        return self._choiceset

    @choiceset.setter
    def choiceset(self, value):
        self._set_rule(self._choiceset, 'defaultallow', value, 'choiceset')

    @property
    def feasible(self):
        return self._feasible

    @feasible.setter
    def feasible(self, value):
        self._set_rule(self._feasible, 'defaultfeasible', value, 'feasibility')

    def _set_rule(self, store, defkey, value, what):
        self.needtocompile = True
        if value is None:
            store[defkey] = True
            store['rules'] = []
        elif isinstance(value, (tuple, list)) and len(value) == 2 and value[0] == defkey:
            store[defkey] = bool(value[1])
        elif (isinstance(value, (tuple, list)) and len(value) == 2
              and isinstance(value[0], str) and isinstance(value[1], str)):
            store['rules'].append({'condition': value[0], 'description': value[1]})
        else:
            raise EgdstError('Unrecognized structure for %s definition!' % what)

    @property
    def trpr(self):
        return self._trpr

    @trpr.setter
    def trpr(self, value):
        # egdstmodel.m:971-1024
        self.needtocompile = True
        if value is None:
            self.__dict__['_trpr'] = []
            return
        value = list(value)
        if len(value) == 2 and isinstance(value[0], str):
            value = [len(self._s)] + value  # varindex skipped: last added variable
        if not (len(value) == 3 and isinstance(value[0], (int, np.integer)) and 1 <= value[0] <= self.nnst
                and isinstance(value[1], str)):
            raise EgdstError('Unrecognized structure for trpr definition!')
        vi, cond, mat = value
        n = self.stm[vi - 1]
        if isinstance(mat, str):
            # varindex + condition + executable string: the deterministic motion rule of a continuous state
            # (egdstmodel.m:997-1003)
            while len(self._trpr) < vi:
                self._trpr.append(_Trpr(0))
            self._trpr[vi - 1].varindex = vi
            self._trpr[vi - 1].cases.append(_Case(cond, mat))
            return
        rows = [list(r) for r in mat]
        if len(rows) != n or any(len(r) != n for r in rows):
            raise EgdstError('Unrecognized structure for trpr definition!')
        prob = []
        for r in rows:
            pr = []
            for x in r:
                if isinstance(x, str):
                    pr.append(x if x != '' else '0.0')
                else:
                    pr.append('%10.10f' % float(x))  # numbers are rewritten as strings (egdstmodel.m:1002)
            prob.append(pr)
        while len(self._trpr) < vi:
            self._trpr.append(_Trpr(0))
        self._trpr[vi - 1].varindex = vi
        self._trpr[vi - 1].cases.append(_Case(cond, prob))

    def _checkrefs(self):
        # egdstmodel.m:1463-1494: refs must be unique and not shadow globals / reserved words
        refs = GLOBALS + RESERVED + [e.ref for e in self._eq] + [c.ref for c in self._coef] + \
            [p.ref for p in self._param]
        seen, dup = set(), []
        for r in refs:
            if r in seen:
                dup.append(r)
            seen.add(r)
        if dup:
            raise EgdstError('Not uniques refs! %s' % dup)

    # ------------------------------------------------------------------ public methods
    def quietly(self, *args):
        self.quiet = not args

    def reswords(self):
        return GLOBALS + RESERVED

    def setparam(self, *args):
        # egdstmodel.m:1077-1113
        if len(args) == 0:
            return self._listparam()
        if len(args) == 1:
            vec = np.atleast_1d(np.asarray(args[0], dtype=float))
            if vec.size != len(self._param):
                raise EgdstError('Passed vector does not match the dimentionality of param vector in the model!')
            for p, v in zip(self._param, vec):
                p.value = float(v)
            return None
        if len(args) % 2:
            raise EgdstError("Expected pairs 'name',value,.. or index,value,.. !")
        for k in range(0, len(args), 2):
            key, val = args[k], args[k + 1]
            if isinstance(key, (int, np.integer)) and 1 <= key <= len(self._param):
                self._param[key - 1].value = float(val)  # base-1 like MATLAB
            elif isinstance(key, str) and key in [p.ref for p in self._param]:
                for p in self._param:
                    if p.ref == key:
                        p.value = float(val)
            else:
                raise EgdstError("Unrecognized pairs 'name',value,.. or index,value,.. !")
        return None

    def getparam(self, *args):
        # egdstmodel.m:1115-1138
        if len(args) == 0:
            return np.array([p.value for p in self._param])
        if len(args) == 1:
            val = args[0]
            if isinstance(val, (int, np.integer)) and val <= len(self._param):
                return self._param[val - 1].value
            if isinstance(val, str) and val in [p.ref for p in self._param]:
                return [p.value for p in self._param if p.ref == val][-1]
            raise EgdstError('Unrecognized parameter name or parameter index out of bounds')
        raise EgdstError('Please call getparam with index or param ref')

    def _listparam(self):
        lines = ['%3s%20s%50s%12s' % ('No', 'Name', 'Description', 'Value')]
        for k, p in enumerate(self._param, 1):
            lines.append('%3d%20s%50s%12.4f' % (k, p.ref[:19], p.description[:49], p.value))
        return '\n'.join(lines)

    def param_vector(self):
        return np.array([p.value for p in self._param], dtype=np.float64)

    def indx(self, v):
        """Index (base-1, like the reference) of a state/decision vector (egdstmodel.m:1296-1330)."""
        v = np.atleast_2d(np.asarray(v, dtype=float))
        if v.shape[1] == len(self._s):
            table = self.states
        elif v.shape[1] == len(self._d):
            table = self.decisions
        else:
            raise EgdstError('Wrong dimention of the vector!')
        out = []
        for row in v:
            hit = np.where((table == row).all(axis=1))[0]
            out.append(int(hit[0]) + 1 if hit.size else float('nan'))
        return out if len(out) > 1 else out[0]

    # -- compile / solve / sim are implemented in terms of codegen + runtime ---------------------
    def analyse_optim(self):
        from .codegen import analyse_optim
        self.optim = analyse_optim(self)
        return self.optim

    def descriptor(self):
        """Run-time scalars read by parseModel (egdst_lib.c:34-62) + the quadrature array."""
        if self.ngridmax <= self.ngridm:  # solve: egdstmodel.m:1145-1147
            self.__dict__['_ngridmax'] = 2 * self.ngridm
        self.quadrature = quadrature_array(int(self.ny))
        return dict(t0=int(self.t0), T=int(self.T), ngridm=int(self.ngridm), ngridmax=int(self.ngridmax),
                    nthrhmax=int(self.nthrhmax), ny=int(self.ny), mmax=float(self.mmax), a0=float(self.a0),
                    quadrature=np.ascontiguousarray(self.quadrature, dtype=np.float64))

    def compile(self, build_dir=None, force=False):  # noqa: A003  (reference method name)
        """Counterpart of compile.m: generate the model plugin and build the per-model HIP library."""
        from . import build as _build
        self.analyse_optim()
        self._lib = _build.build_model(self, build_dir=build_dir, force=force)
        self.M = None
        self.D = None
        self.sims = None
        self.needtocompile = False
        return self._lib

    def solve(self, keep_on_device=False, dbgout=False):
        """``[M,D,dbgout]=egdst_solver(model)`` (egdstmodel.m:1141-1178) through the C-ABI.  The class drops the third
        output (egdstmodel.m:1170); dbgout=True keeps it in ``self.dbgout`` ([nt*nst*nd*2*nt, 7])."""
        if self.needtocompile or self._lib is None:
            raise EgdstError('The model needs to be compiled first!\nRun <model>.compile')
        from . import runtime
        import time
        t = time.perf_counter()
        sol = runtime.solve_model(self, dbgout=dbgout)
        self.lastrun_solver = time.perf_counter() - t
        self.dbgout = getattr(sol, 'dbgout', None)
        self._solution = sol
        self.M, self.D = sol.cells()
        from . import runtime as _rt
        self.__dict__['_solver_cells'] = (self.M, self.D, _rt._cells_stamp(self.M, self.D))  # the live handle holds exactly these cells (runtime.py)
        return sol

    def sim(self, *args):
        """``sims=egdst_simulator(model,rndtype)`` (egdstmodel.m:1210-1276); sims is [nsim x nt x nvar]."""
        if self.needtocompile or self.M is None:
            raise EgdstError('The model needs to be compiled and solved first!\nRun <model>.compile and <model>.sim')
        rndtype = 0
        for a in args:
            if isinstance(a, str):
                if a not in ('own_shocks', 'same_shocks'):
                    raise EgdstError('Could not recognize argumend!')
                rndtype = 1 if a == 'same_shocks' else 0
            else:
                arr = np.atleast_2d(np.asarray(a, dtype=float))
                if arr.shape[1] != 2:
                    raise EgdstError('Could not recognize argumend!')
                self.init = arr
        if self.init is None or len(self.init) == 0:
            self.init = np.array([[1.0, 0.0]])
        self.init = np.array(self.init, dtype=float)
        self.init[:, 1] = np.maximum(self.init[:, 1], self.a0)  # egdstmodel.m:1237
        if self.randstream is None or len(self.randstream) == 0:
            # reference: rand(max(nsim,100)*nt*100,1) (egdstmodel.m:1254); seeded here for reproducibility
            n = max(self.init.shape[0], 100) * self.nt * 100
            self.randstream = np.random.default_rng(0).random(n)
        from . import runtime
        sims = runtime.simulate_model(self, rndtype)  # [nsimout x nt x nsim] column-major semantics
        self.sims = sims
        return sims

    CALL_NAMES = {'utility': 1, 'util': 1, 'u': 1, 'mutility': 2, 'mu': 2, 'discount': 3, 'df': 3, 'budget': 4, 'b': 4,
                  'mbudget': 5, 'mb': 5, 'value': 6, 'vf': 6}

    def call(self, func, funcargs):
        """``res=model.call(funcname,funcargs)`` (egdstmodel.m:1181-1207 -> egdst_call.c): runs internal functions of the
        solved model -- 'utility'|'util'|'u' (it,ist,id,consumption), 'mutility'|'mu', 'discount'|'df' (it,ist),
        'budget'|'b' (it,ist,id,savings,ist(t+1),shock), 'mbudget'|'mb', 'value'|'vf' (it,ist,cash); rows of funcargs
        are evaluated independently (vector input)."""
        if self.needtocompile or self.M is None:
            raise EgdstError('The model needs to be compiled and solved first!\nRun <model>.compile')
        if func not in self.CALL_NAMES:
            raise EgdstError('Unknown internal model function to call!')
        from . import runtime
        return runtime.call_model(self, self.CALL_NAMES[func], funcargs)

    def sims2panel(self):
        # egdstmodel.m:1279-1292
        if self.sims is None:
            raise EgdstError('The model needs to be simulated first!\nRun <model>.sim')
        nsim, nt, nv = self.sims.shape
        rows = []
        for i in range(nsim):
            for t in range(nt):
                if not math.isnan(self.sims[i, t, 0]):
                    rows.append(np.concatenate([[i + 1, t + 1], self.sims[i, t, :]]))
        labels = ['i', 't'] + list(self.simlabels)
        return np.array(rows), labels

    def make_simlabels(self):
        # compile.m:633-648
        labels = list(SIMLABELS_DEFAULT)
        for i, v in enumerate(self._s, 1):
            labels.append('%2d %s (st%d)' % (11 + i, v.name, i))
        for i, v in enumerate(self._d, 1):
            labels.append('%2d %s (dc%d)' % (11 + len(self._s) + i, v.name, i))
        for i, e in enumerate(self._eq, 1):
            labels.append('%2d %s (eq%d)' % (11 + len(self._s) + len(self._d) + i, e.description, i))
        self.simlabels = labels
        return labels


_token_re = re.compile(r'[A-Za-z_][A-Za-z_0-9]*')
