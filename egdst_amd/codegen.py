"""Model-plugin generator: the ``compile`` counterpart of @egdstmodel/compile.m.

compile.m turns the user's executable C strings into ``modelspec.c/.h`` (compile.m:171-660)
and analyses four optimisation flags (compile.m:669-747).  This module emits ONE header,
``modelspec.h``, with the same ~20 model functions as ``static`` inline functions whose
qualifier is a macro (``MS_FN``), so the very same generated plugin is compiled

  * as ``__device__`` code into the per-model HIP library (the product), and
  * as plain C into the CPU oracle (test infrastructure, ``oracle/``).

Differences from the reference's generated code, all behaviour-preserving:
  * parameters are read from ``E->par[k]`` (one vector per parameter draw) instead of
    process globals filled by ``loadparameters()`` (compile.m:469-475);
  * ``states``/``decisions``/``stm`` are baked as constant tables (they only change with
    ``m.s``/``m.d``, which force a recompile in the reference too: egdstmodel.m:575,659);
  * identifier rewriting is done on a token stream, not with ordered regexes
    (compile.m:12-64 ``StdConvertN``): same substitutions, no accidental partial matches;
  * an incomplete ``trpr`` case list raises a device-side error flag instead of
    ``mexErrMsgTxt`` (compile.m:541-544).
"""
from __future__ import annotations

import hashlib
import re

import numpy as np

_TOK = re.compile(r'(?P<num>(?:\d+\.?\d*|\.\d+)(?:[eE][+-]?\d+)?[fFlLuU]*)|(?P<id>[A-Za-z_][A-Za-z_0-9]*)|'
                  r'(?P<str>"(?:\\.|[^"\\])*")')

_C_KEEP = {'return', 'if', 'else', 'for', 'while', 'int', 'double', 'float', 'const', 'static', 'switch',
           'case', 'break', 'default', 'do', 'long', 'unsigned', 'void', 'sizeof',
           # libm (double overloads exist in both C and HIP device code)
           'log', 'exp', 'pow', 'sqrt', 'fabs', 'floor', 'ceil', 'fmin', 'fmax', 'log1p', 'expm1', 'tanh',
           'sin', 'cos', 'atan', 'erf', 'erfc', 'fmod', 'isfinite', 'isnan', 'INFINITY', 'NAN',
           'consumption', 'mutility', 'x'}


class CodegenError(Exception):
    pass


def _lines(expr):
    return [expr] if isinstance(expr, str) else list(expr)


class _Rewriter:
    def __init__(self, model):
        self.m = model
        self.param_idx = {p.ref: i for i, p in enumerate(model.param)}
        self.coefs = {c.ref for c in model.coef}
        self.eqs = {e.ref: e for e in model.eq}

    def convert(self, text, allow_next, where, banned=()):
        """Token rewrite of one executable string (StdConvertN, compile.m:12-64)."""
        m = self.m
        # ProhibitString (compile.m:65-84): banned word followed by optional digits, case-insensitive
        for b in banned:
            if re.search(r'\b' + b + r'\d*\b', text, flags=re.IGNORECASE):
                raise CodegenError('Error in %s: use of `%s` is not allowed in %s!' % (where, b, where))

        def need_next(tok):
            if not allow_next:
                raise CodegenError('`%s` refers to the next period and cannot be used in %s' % (tok, where))

        def sub(mo):
            if mo.group('num') or mo.group('str'):
                return mo.group(0)
            t = mo.group('id')
            if t == 'min':
                return 'MS_MIN'
            if t == 'max':
                return 'MS_MAX'
            if t == 'true':
                return '1'
            if t == 'false':
                return '0'
            if t == 'it':
                return 'curr->it'
            if t == 'age':
                return '(curr->it+E->t0)'
            if t == 'id':
                return 'curr->id'
            if t == 'ist':
                return 'curr->ist'
            if t == 'ist1':
                need_next(t)
                return 'next->ist'
            # with continuous states the simulator evaluates the model functions BY VALUE (byval, egdst_lib.h:63,
            # compile.m:26-34): states and decisions are then read from the period's own st[]/dc[] instead of the tables
            byval = any(v.type == 'continuous' for v in m.s)
            mm = re.fullmatch(r'dc(\d+)', t)
            if mm and 1 <= int(mm.group(1)) <= m.nnd:
                k = int(mm.group(1)) - 1
                if byval:
                    return '(curr->byval>0?curr->dc[%d]:ms_decisions[curr->id+%d*MS_ND])' % (k, k)
                return 'ms_decisions[curr->id+%d*MS_ND]' % k
            mm = re.fullmatch(r'st(\d+)(n?)', t)
            if mm and 1 <= int(mm.group(1)) <= m.nnst:
                k = int(mm.group(1)) - 1
                if mm.group(2):
                    need_next(t)
                    if byval:
                        return '(next->byval>0?next->st[%d]:ms_states[next->ist+%d*MS_NST])' % (k, k)
                    return 'ms_states[next->ist+%d*MS_NST]' % k
                if byval:
                    return '(curr->byval>0?curr->st[%d]:ms_states[curr->ist+%d*MS_NST])' % (k, k)
                return 'ms_states[curr->ist+%d*MS_NST]' % k
            if t in self.eqs:
                if self.eqs[t].type == 'next':
                    need_next(t)
                    return 'ms_eq_%s(E,curr,next)' % t
                return 'ms_eq_%s(E,curr)' % t
            if t == 'cash':
                return 'curr->cash'
            if t == 'savings':
                need_next(t)
                return 'next->savings'
            if t == 'shock':
                need_next(t)
                return 'next->shock'
            if t == 'sigma':
                need_next(t)
                return 'ms_sigma(E,curr,next)'
            if t == 'mu':
                need_next(t)
                return 'ms_mu(E,curr,next)'
            if t == 'discount':
                return 'ms_discount(E,curr)'
            if t == 'survival':
                return 'ms_survival(E,curr)'
            if t in self.param_idx:
                return 'E->par[%d]' % self.param_idx[t]
            if t in self.coefs:
                return 'ms_coef_' + t
            if t in ('t0', 'T', 'ngridm', 'ngridmax', 'nthrhmax', 'ny', 'mmax', 'a0'):
                return 'E->' + t
            if t in ('nd', 'nnd', 'nst', 'nnst'):
                return 'MS_' + t.upper()
            if t in ('exp', 'log', 'pow'):
                return 'MS_' + t.upper()  # include/egdst_math.h: bit-reproducible or native libm
            if t in _C_KEEP:
                return t
            raise CodegenError('Unknown identifier `%s` in %s: %s' % (t, where, text))

        return _TOK.sub(sub, text)



def _mentions(model, expr, words, seen=None):
    """Does the executable string (or any equation it references, recursively) use one of `words` as an identifier?"""
    seen = set() if seen is None else seen
    eqs = {e.ref: e for e in model.eq}
    for ln in _lines(expr):
        for mo in _TOK.finditer(ln):
            t = mo.group('id')
            if not t:
                continue
            if t in words:
                return True
            if t in eqs and t not in seen:
                seen.add(t)
                if _mentions(model, eqs[t].expression, words, seen):
                    return True
            if t in ('mu', 'sigma') and t not in seen:   # mu may be written in terms of sigma (and vice versa)
                seen.add(t)
                if _mentions(model, model.shock[t], words, seen):
                    return True
    return False


def analyse_optim(model):
    """Optimisation flags exactly as compile.m:669-747 derives them (regex on the raw strings)."""
    def joined(x):
        return x if isinstance(x, str) else ''.join(x)

    dcpat = 'dc[' + '  '.join(str(i) for i in range(1, model.nnd + 1)) + ']'
    marg = joined(model.u.get('marginal', ''))
    util = joined(model.u.get('utility', ''))
    optim = {}
    optim['optim_MUnoD'] = not (re.search(dcpat, marg) or 'id' in marg)
    optim['optim_UnoD'] = not (re.search(dcpat, util) or 'id' in util)
    # additive separability (compile.m:690-727)
    uasd = True
    dcpat0 = 'dc[' + '  '.join(str(i) for i in range(0, model.nnd)) + ']'
    for tmp in _lines(model.u.get('utility', '')):
        if not uasd:
            break
        while re.search(r'\([^+\-()]*\)', tmp):
            tmp = re.sub(r'\(([^+\-()]*)\)', r'[\1]', tmp)
        while re.search(r'\([^()]*(\+|\-)[^()]*\)', tmp):
            tmp = re.sub(r'(\([^()]*)(\+|\-)([^()]*\))', r'\1#\3', tmp)
        for sub in re.split(r'\+|-', tmp):
            if (re.search(dcpat0, sub) or 'id' in sub) and 'consumption' in sub:
                uasd = False
                break
    if not uasd:
        raise CodegenError('Utility is not additively separable in consumption and discrete choices. '
                           'This case is not yet implemented!')
    optim['optim_UasD'] = True
    allpr = ''
    for tr in model.trpr:
        for case in tr.cases:
            allpr += '#' + ''.join(''.join(r) for r in case.prob)
    optim['optim_TRPRnoSH'] = 'shock' not in allpr
    return optim


def _fmt(v):
    return repr(float(v))


def generate_modelspec(model):
    """Return the text of modelspec.h for ``model`` (an ``egdstmodel``)."""
    m = model
    if m.nnst == 0 or m.nnd == 0:
        raise CodegenError('Model needs at least one state and one decision variable')
    for part, key, name in ((m.u, 'utility', '.u.utility'), (m.u, 'marginal', '.u.marginal'),
                            (m.u, 'marginalinverse', '.u.marginalinverse'),
                            (m.budget, 'cashinhand', '.budget.cashinhand'),
                            (m.budget, 'marginal', '.budget.marginal'),
                            (m.shock, 'mu', '.shock.mu'), (m.shock, 'sigma', '.shock.sigma')):
        if not part.get(key):
            raise CodegenError('Missing %s, can not proceed with compile!' % name)
    if not m.discount:
        raise CodegenError('Missing .discount, can not proceed with compile!')
    if len(m.trpr) < m.nnst or any(t.varindex == 0 or not t.cases for t in m.trpr):
        raise CodegenError('Missing .trpr, can not proceed with compile!')
    optim = analyse_optim(m)
    rw = _Rewriter(m)
    L = []
    w = L.append
    w("/* Model plugin generated by egdst_amd.codegen for the model '%s' */" % m.label.replace('*/', ''))
    w('#ifndef EGDST_MODELSPEC_H')
    w('#define EGDST_MODELSPEC_H')
    w('#ifndef MS_FN')
    w('#error "define MS_FN (function qualifier) and MS_TABLE (constant-table qualifier) before including modelspec.h"')
    w('#endif')
    w('#include "egdst_math.h"')
    w('#define MS_LABEL "%s"' % re.sub(r'[^A-Za-z0-9 _.-]', '', m.label))
    w('#define MS_NNST %d' % m.nnst)
    w('#define MS_NND %d' % m.nnd)
    w('#define MS_NST %d' % m.nst)
    w('#define MS_ND %d' % m.nd)
    w('#define MS_NPARAM %d' % len(m.param))
    w('#define MS_NEQ %d' % len(m.eq))
    w('#define MS_DISTRIB %d' % (1 if m.shock['type'] == 'lognormal' else 2))
    for k in ('optim_MUnoD', 'optim_UnoD', 'optim_UasD', 'optim_TRPRnoSH'):
        w('#define MS_%s %d' % (k.upper(), int(optim[k])))
    # 1: mu and sigma of the shock do not depend on savings (nor on cash): the nodes exp(mu + z*sigma) of a
    # (current state, decision, next state) are the same for every end-of-period asset point (device: computed once per workgroup)
    w('#define MS_SHOCK_NODES_SHARED %d' % int(not (_mentions(m, m.shock['mu'], ('savings', 'cash', 'shock')) or
                                                   _mentions(m, m.shock['sigma'], ('savings', 'cash', 'shock')))))
    for k in ('TOLERANCE', 'ZEROCONSUMPTION', 'DOUBLEPOINT_DELTA'):
        w('#define MS_%s (%s)' % (k, m.cflags[k]))
    w('#define MS_MAX(X,Y) (((X)>(Y))?(X):(Y))')
    w('#define MS_MIN(X,Y) (((X)<(Y))?(X):(Y))')
    cont = [i for i, v in enumerate(m.s) if v.type == 'continuous']
    w('#define MS_NCONT %d  /* continuous state variables (SURVEY 8f N4) */' % len(cont))
    if cont:
        # PeriodVars with the by-value fields of the simulator (compile.m:192); C++ gives byval its default, C code sets it
        w('#ifdef __cplusplus')
        w('struct ms_pv {int it; int ist; int id; double cash; double savings; double shock; int byval = 0; '
          'double st[MS_NNST]; double dc[MS_NND];};')
        w('#else')
        w('typedef struct ms_pv {int it; int ist; int id; double cash; double savings; double shock; int byval; '
          'double st[MS_NNST]; double dc[MS_NND];} ms_pv;')
        w('#endif')
    else:
        w('typedef struct ms_pv {int it; int ist; int id; double cash; double savings; double shock;} ms_pv;')
    w('typedef struct ms_env {int t0; int T; int ngridm; int ngridmax; int nthrhmax; int ny; '
      'double mmax; double a0; const double* par;} ms_env;')
    # constant tables
    sizes = [int(x) for x in m.stm[:m.nnst]]
    strides = [int(x) for x in m.stm[m.nnst:]]
    w('MS_TABLE int ms_stsize[%d] = {%s};' % (m.nnst, ','.join(map(str, sizes))))
    w('MS_TABLE int ms_ststride[%d] = {%s};' % (m.nnst, ','.join(map(str, strides))))
    st = np.asarray(m.states, dtype=float)  # [nst x nnst]; C code indexes states[ist + k*nst]
    w('MS_TABLE double ms_states[%d] = {%s};' % (m.nst * m.nnst, ','.join(_fmt(v) for v in st.T.reshape(-1))))
    dc = np.asarray(m.decisions, dtype=float)
    w('MS_TABLE double ms_decisions[%d] = {%s};' % (m.nd * m.nnd, ','.join(_fmt(v) for v in dc.T.reshape(-1))))
    w('MS_TABLE int ms_stcont[%d] = {%s};  /* 1: continuous */' % (max(m.nnst, 1), ','.join('1' if v.type == 'continuous' else '0' for v in m.s) or '0'))
    for i in cont:
        w('MS_TABLE double ms_stgrid%d[%d] = {%s};' % (i + 1, len(m.s[i].values), ','.join(_fmt(v) for v in m.s[i].values)))
    if cont:
        # bxsearch (egdst_lib.c:123-176, type 0): left index for interpolation with extrapolation on both sides
        w('MS_FN int ms_bxsearch(double x, const double* g, int n) {')
        w('  int lo = 1, hi = n - 2, mid;')
        w('  if (x < g[1]) return 0;')
        w('  if (x >= g[n-2]) return n - 2;')
        w('  while (hi - lo > 1) { mid = (hi + lo) / 2; if (g[0] <= g[n-1] && g[mid] > x) hi = mid; else lo = mid; }')
        w('  return lo;')
        w('}')
        w('MS_FN const double* ms_stgrid(int k) {')
        for i in cont:
            w('  if (k == %d) return ms_stgrid%d;' % (i, i + 1))
        w('  return 0;')
        w('}')
    for c in m.coef:
        # base-1 indexing kept by padding row/column 0 (compile.m:199-219); values printed %18.15f there
        r, cc = c.array.shape
        rows = ['{' + ','.join(['0.0'] * (cc + 1)) + '}']
        for i in range(r):
            rows.append('{0.0,' + ','.join('%18.15f' % v for v in c.array[i]) + '}')
        w('MS_TABLE double ms_coef_%s[%d][%d] = {%s};' % (c.ref, r + 1, cc + 1, ','.join(rows)))

    def emit(sig, expr, allow_next, where, banned=()):
        w('MS_FN ' + sig + ' {')
        ls = _lines(expr)
        if isinstance(expr, str):
            w('  return ' + rw.convert(expr, allow_next, where, banned) + ';')
        else:
            for ln in ls:
                w('  ' + rw.convert(ln, allow_next, where, banned))
        w('}')

    C1 = 'const ms_env* E, const ms_pv* curr'
    C2 = 'const ms_env* E, const ms_pv* curr, const ms_pv* next'
    # forward declarations (equations may be chained, mu/sigma may reference each other)
    for e in m.eq:
        w('MS_FN double ms_eq_%s(%s);' % (e.ref, C2 if e.type == 'next' else C1))
    w('MS_FN double ms_mu(%s);' % C2)
    w('MS_FN double ms_sigma(%s);' % C2)
    w('MS_FN double ms_discount(%s);' % C1)
    w('MS_FN double ms_survival(%s);' % C1)
    emit('double ms_discount(%s)' % C1, m.discount, False, 'discount', ('id', 'dc', 'cash'))
    emit('double ms_survival(%s)' % C1, m.survival, False, 'survival', ('id', 'dc', 'cash'))
    emit('double ms_utility(%s, double consumption)' % C1, m.u['utility'], False, 'utility', ('cash',))
    emit('double ms_utility_marginal(%s, double consumption)' % C1, m.u['marginal'], False,
         'marginal utility', ('cash',))
    emit('double ms_utility_marginal_inverse(%s, double mutility)' % C1, m.u['marginalinverse'], False,
         'marginal utility inverse', ('cash',))
    tb = ('id', 'dc', 'cash', 'savings', 'shock')
    emit('double ms_tr(%s, double x)' % C1, m.transform['direct'], False, 'extrapolation function', tb)
    emit('double ms_trinv(%s, double x)' % C1, m.transform['inverse'], False, 'extrapolation function', tb)
    emit('double ms_sigma(%s)' % C2, m.shock['sigma'], True, 'sigma parameter', ('shock',))
    emit('double ms_mu(%s)' % C2, m.shock['mu'], True, 'mu paremeter', ('shock',))
    for e in m.eq:
        emit('double ms_eq_%s(%s)' % (e.ref, C2 if e.type == 'next' else C1), e.expression,
             e.type == 'next', 'equation ' + e.ref)
    emit('double ms_cashinhand(%s)' % C2, m.budget['cashinhand'], True, 'cashinhand', ('cash',))
    emit('double ms_cashinhand_marginal(%s)' % C2, m.budget['marginal'], True, 'cashinhand marginal', ('cash',))
    # choiceset / feasible (compile.m:403-445)
    w('MS_FN int ms_inchoiceset(%s) {' % C1)
    w('  int res = %d;' % int(m.choiceset['defaultallow']))
    for r in m.choiceset['rules']:
        w('  if (%s) res = %d;' % (rw.convert(r['condition'], False, '.choiceset', ('cash',)),
                                  int(not m.choiceset['defaultallow'])))
    w('  return res;')
    w('}')
    w('MS_FN int ms_feasible(%s) {' % C1)
    w('  int res = %d;' % int(m.feasible['defaultfeasible']))
    for r in m.feasible['rules']:
        w('  if (%s) res = %d;' % (rw.convert(r['condition'], False, '.feasible', ('id', 'dc', 'cash')),
                                  int(not m.feasible['defaultfeasible'])))
    w('  return res;')
    w('}')
    # transition probabilities (compile.m:476-551); *err is raised on an incomplete case list.  `all` (compile.m:482):
    # 1 = the solver's call, continuous states contribute their interpolation weights; 0 = the simulator's call, discrete
    # variables only.  Models without continuous states keep the four-argument form (ms_trpr_all is then ms_trpr).
    def trpr_body(with_all):
        w('  double res = 1.0;')
        w('  int varindex, varindex1;')
        if cont:
            w('  double nval; (void)nval;')
        for tr in m.trpr:
            k = tr.varindex - 1
            n = sizes[k]
            is_cont = m.s[k].type == 'continuous'
            w('  varindex = (curr->ist/%d)%%%d; varindex1 = (next->ist/%d)%%%d;' % (strides[k], n, strides[k], n))
            first = True
            for case in tr.cases:
                w('  %sif (%s) {' % ('' if first else 'else ', rw.convert(case.condition, True, 'trpr condition')))
                first = False
                if not is_cont:
                    w('    switch (varindex*%d+varindex1) {' % n)
                    for i in range(n):
                        for j in range(n):
                            w('      case %d: res *= %s; break;' % (i * n + j, rw.convert(case.prob[i][j], True, 'trpr')))
                    w('      default: break;')
                    w('    }')
                elif with_all:
                    # deterministic motion rule: linear-interpolation weights of the two neighbouring grid points (:527-538)
                    g = 'ms_stgrid%d' % (k + 1)
                    w('    nval = %s;' % rw.convert(case.prob, True, 'motion rules', ('ist1',)))
                    w('    varindex = ms_bxsearch(nval, %s, %d);' % (g, n))
                    w('    if (varindex==varindex1) res *= (%s[varindex+1]-nval)/(%s[varindex+1]-%s[varindex]);' % (g, g, g))
                    w('    else if (varindex+1==varindex1) res *= (nval-%s[varindex])/(%s[varindex+1]-%s[varindex]);' % (g, g, g))
                    w('    else return 0.0;')
                w('  }')
            w('  else { *err = 1; return 0.0; }')
            w('  if (res==0.0) return 0.0;')
        w('  return res;')

    w('MS_FN double ms_trpr(%s, int* err) {' % C2)
    trpr_body(True)
    w('}')
    if cont:
        w('MS_FN double ms_trpr_discrete(%s, int* err) {  /* all == 0 */' % C2)
        trpr_body(False)
        w('}')
        # exact next-period values of the continuous states (trpr_cont, compile.m:553-575)
        w('MS_FN void ms_trpr_cont(const ms_env* E, const ms_pv* curr, ms_pv* next) {')
        for tr in m.trpr:
            k = tr.varindex - 1
            if m.s[k].type != 'continuous':
                continue
            first = True
            for case in tr.cases:
                w('  %sif (%s) {' % ('' if first else 'else ', rw.convert(case.condition, True, 'trpr condition')))
                first = False
                w('    next->st[%d] = %s;' % (k, rw.convert(case.prob, True, 'motion rules')))
                w('  }')
        w('}')
    else:
        w('#define ms_trpr_discrete ms_trpr')
    # equations for the simulator output (compile.m:629-649)
    w('MS_FN void ms_eqs_sim(%s, int has_next, double* out) {' % C2)
    w('  int i = 0; (void)i; (void)has_next; (void)out; (void)next;')
    for e in m.eq:
        if e.type == 'next':
            w('  if (!has_next) out[i++] = NAN; else out[i++] = ms_eq_%s(E,curr,next);' % e.ref)
        else:
            w('  out[i++] = ms_eq_%s(E,curr);' % e.ref)
    w('}')
    w('#endif')
    return '\n'.join(L) + '\n'


def spec_hash(text):
    return hashlib.sha1(text.encode()).hexdigest()[:12]
