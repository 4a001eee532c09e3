"""The model plugin against an INDEPENDENT evaluation of the user's strings (tests/plugin_eval.py).

egdst_amd/codegen.py restates compile.m's string -> C rewriting once, and the same generated modelspec.h is compiled into
the device library AND into the CPU oracle: a rewriting bug would be common to both and no parity test could see it.  Here
every model function is computed a third way -- the user's strings parsed and evaluated in Python with the meanings the
reference's DSL gives its identifiers (compile.m:12-64) -- and compared with

  * the oracle (CPU tests, all 12 models) and the device (`-m gpu`): utility, marginal utility, discount, budget and
    marginal budget through the accessor gateway (egdst_call.c switches 1-5) on random arguments;
  * the simulator's output columns (mu, sigma, shock -> cash-in-hand, u(c), discount, equations, states, decisions) and
    the transitions it draws (trpr, feasible, survival), replayed from the uniforms;
  * committed expected texts of modelspec.h for the five shipped models (tests/golden/modelspec_*.h, read against
    compile.m:255-551 by hand);
and a deliberate one-token bug in the generator is shown to fail the comparison.
"""
import math
import os

import numpy as np
import pytest

from egdst_amd import codegen, examples
from oracle_harness import Oracle
from plugin_eval import PluginEval, parse

HERE = os.path.dirname(os.path.abspath(__file__))
SMALL = {'deaton1': dict(T=6, ngridm=30), 'deaton2': dict(T=6, ngridm=30), 'deaton_sig': dict(T=6, ngridm=30),
         'retirement1': dict(T=6, ngridm=30), 'retirement2': dict(T=6, ngridm=30), 'retirement_sig': dict(T=6, ngridm=30),
         'occ3': dict(T=6, ngridm=30, ngridmax=100), 'model2': dict(T=4, ngridm=30), 'retirement8': dict(T=5, ngridm=30, ny=3),
         'cake_normal': dict(), 'retirement_mortal': dict(), 'retirement_hc': dict()}
MODELS = sorted(examples.REGISTRY)


def make(name):
    kw = dict(SMALL[name])
    try:
        return examples.REGISTRY[name](**kw)
    except TypeError:
        return examples.REGISTRY[name]()


def ulps(a, b):
    if a == b or (a != a and b != b):
        return 0.0
    if not (math.isfinite(a) and math.isfinite(b)):
        return math.inf
    return abs(a - b) / max(math.ulp(max(abs(a), abs(b))), 5e-324)


def call_args(m, nt, nst, nd, n=60, seed=5):
    rng = np.random.default_rng(seed)
    it = rng.integers(m.t0, m.t0 + nt - 1, n).astype(float)          # (budget needs a next period)
    ist = rng.integers(1, nst + 1, n).astype(float)
    idc = rng.integers(1, nd + 1, n).astype(float)
    cons = rng.uniform(0.05, 0.95 * (m.mmax - m.a0), n)
    sav = rng.uniform(m.a0, m.mmax, n)
    ist1 = rng.integers(1, nst + 1, n).astype(float)
    shock = rng.uniform(0.4, 1.8, n)
    return it, ist, idc, cons, sav, ist1, shock


def check_call_switches(m, call, nt, nst, nd, tol_ulp=0.0):
    """call(sw, args) -> values of the accessor gateway; compares with the Python evaluation of the strings"""
    pe = PluginEval(m)
    it, ist, idc, cons, sav, ist1, shock = call_args(m, nt, nst, nd)
    n = len(it)
    worst = 0.0
    got = {1: call(1, np.column_stack([it, ist, idc, cons])), 2: call(2, np.column_stack([it, ist, idc, cons])),
           3: call(3, np.column_stack([it, ist])), 4: call(4, np.column_stack([it, ist, idc, sav, ist1, shock])),
           5: call(5, np.column_stack([it, ist, idc, sav, ist1, shock]))}
    for k in range(n):
        cur = {'it': int(it[k]) - int(m.t0), 'ist': int(ist[k]) - 1, 'id': int(idc[k]) - 1, 'cash': 0.0}
        nxt = {'it': cur['it'] + 1, 'ist': int(ist1[k]) - 1, 'savings': float(sav[k]), 'shock': float(shock[k])}
        want = {1: pe.utility(cur, float(cons[k])), 2: pe.utility_marginal(cur, float(cons[k])),
                3: pe.discount(dict(cur, id=0)), 4: pe.cashinhand(cur, nxt), 5: pe.cashinhand_marginal(cur, nxt)}
        for sw in want:
            u = ulps(float(got[sw][k]), want[sw])
            assert u <= tol_ulp, (m.label, 'switch', sw, 'row', k, float(got[sw][k]), want[sw], u)
            worst = max(worst, u)
    return worst


def check_simulated_columns(m, sims, init, rs, rndtype, neq):
    """sims [nsim, nt, nout] of the simulator against the strings: every column that is a model function of the period's
    variables, and the state transitions and deaths replayed from the uniform numbers (egdst_simulator.c:204-383)."""
    pe = PluginEval(m)
    nsim, nt, nout = sims.shape
    nnst, nnd = m.nnst, m.nnd
    cont = any(v.type == 'continuous' for v in m.s)
    checked = 0
    for i in range(nsim):
        base = 0 if rndtype == 1 else 4 * nt * i
        prev = None
        irnd = 0
        for t in range(nt):
            row = sims[i, t]
            if np.isnan(row[0]):
                if prev is not None and t >= 1 and not cont:
                    # the agent died here: the third uniform of the period exceeds the survival probability (:265)
                    assert rs[base + irnd + 2] > pe.survival(prev), (m.label, i, t)
                break
            cur = {'it': t, 'ist': int(row[5]), 'id': int(row[4]), 'cash': float(row[0])}
            c = float(row[1])
            assert ulps(float(row[2]), cur['cash'] - c) <= 0
            if not cont:
                assert ulps(float(row[9]), pe.utility(cur, c)) <= 0, (m.label, 'u(c)', i, t)
                assert ulps(float(row[10]), pe.discount(cur)) <= 0, (m.label, 'discount', i, t)
                for k in range(nnst):
                    assert row[11 + k] == float(m.states[cur['ist']][k])
                for k in range(nnd):
                    assert row[11 + nnst + k] == float(m.decisions[cur['id']][k])
            if t == 0:
                assert np.isnan(row[6]) and np.isnan(row[7]) and np.isnan(row[8])
                ecur, enxt, has_next = cur, cur, False
                # (period 0: the gateway evaluates the equations before the policy, with id = 0)
                ecur = dict(cur, id=0)
            else:
                r0, r1, r2 = rs[base + irnd], rs[base + irnd + 1], rs[base + irnd + 2]
                irnd += 3
                nxt = {'it': t, 'ist': cur['ist'], 'savings': prev['cash'] - prev['c'], 'shock': float(row[8])}
                if not cont:
                    assert r2 <= pe.survival(prev), (m.label, 'survival', i, t)
                    # the state drawn by the cumulative transition probabilities (:267-290)
                    r = r0
                    drawn = None
                    for s1 in range(m.nst):
                        cand = dict(nxt, ist=s1)
                        if not pe.feasible({'it': t, 'ist': s1, 'id': 0, 'cash': 0.0}):
                            continue
                        r -= pe.trpr(prev, cand)
                        if r <= 0:
                            drawn = s1
                            break
                    assert drawn == cur['ist'], (m.label, 'transition', i, t, drawn, cur['ist'])
                    mu, sigma = pe.mu(prev, nxt), pe.sigma(prev, nxt)
                    assert ulps(float(row[6]), mu) <= 0 and ulps(float(row[7]), sigma) <= 0, (m.label, 'mu/sigma', i, t)
                    assert ulps(cur['cash'], pe.cashinhand(prev, nxt)) <= 0, (m.label, 'cash', i, t)
                ecur, enxt, has_next = prev, nxt, True
            if not cont:
                for k, e in enumerate(m.eq):
                    got = float(row[11 + nnst + nnd + k])
                    if e.type == 'next' and not has_next:
                        assert np.isnan(got)
                    else:
                        want = pe.equation(e.ref, ecur, enxt if e.type == 'next' else None)
                        assert ulps(got, want) <= 0, (m.label, 'eq', e.ref, i, t, got, want)
            prev = dict(cur, c=c)
            checked += 1
    return checked


def sim_inputs(m, nst, nt, nsim=40, seed=9):
    rng = np.random.default_rng(seed)
    init = np.column_stack([rng.integers(1, nst + 1, nsim).astype(float), rng.uniform(max(m.a0, 0.1), 0.8 * m.mmax, nsim)])
    return init, rng.random(4 * nt * nsim)


# ---- CPU: the oracle's plugin ------------------------------------------------------------------------------------------
@pytest.mark.parametrize('name', MODELS)
def test_oracle_plugin_equals_python_evaluation_of_the_strings(name):
    m = make(name)
    orc = Oracle(m, native_math=True)      # glibc build: Python's math module is the same libm
    ref = orc.solve()
    nt = int(m.T) - int(m.t0) + 1
    worst = check_call_switches(m, lambda sw, a: orc.call(ref, sw, a), nt, orc.nst, orc.nd)
    assert worst == 0.0
    if ref.rc == 0:
        feas = [s for s in range(orc.nst) if ref.len[0, s] > 0]
        init, rs = sim_inputs(m, orc.nst, nt)
        init[:, 0] = np.asarray(feas)[np.asarray(init[:, 0], dtype=int) % len(feas)] + 1
        if name == 'retirement_hc':
            init[:, 0] = 1 + np.asarray(init[:, 0], dtype=int) % 3
        for rndtype in (0, 1):
            sims = orc.sim(ref, init, rs, rndtype=rndtype)
            assert check_simulated_columns(m, sims, init, rs, rndtype, orc.neq) > len(init)


def test_a_one_token_bug_in_the_generator_is_caught(monkeypatch):
    """`savings` rewritten to the CURRENT period's field instead of the next period's (one token of StdConvertN): the
    generated plugin still compiles, oracle and device would still agree with each other -- and this check fails."""
    m = make('retirement2')
    real = codegen._Rewriter.convert

    def buggy(self, text, allow_next, where, banned=()):
        return real(self, text, allow_next, where, banned).replace('next->savings', 'curr->savings')
    monkeypatch.setattr(codegen._Rewriter, 'convert', buggy)
    orc = Oracle(m, native_math=True)      # (keyed by the generated text: a build of its own)
    ref = orc.solve()
    nt = int(m.T) - int(m.t0) + 1
    with pytest.raises(AssertionError):
        check_call_switches(m, lambda sw, a: orc.call(ref, sw, a), nt, orc.nst, orc.nd)


def test_parser_and_c_semantics():
    pe = PluginEval(make('occ3'))
    cur = {'it': 2, 'ist': 0, 'id': 2, 'cash': 1.0}
    assert pe.eval('7/2', cur) == 3 and pe.eval('-7/2', cur) == -3 and pe.eval('7/2.0', cur) == 3.5
    assert pe.eval('(int)2.9+1', cur) == 3 and pe.eval('1<2 && 3>4 || !0', cur) == 1
    assert pe.eval('id?10:20', cur) == 10 and pe.eval('(id==0)*5', cur) == 0
    assert pe.eval('sigs[1][(int)dc1+1]', cur) == 0.75 and pe.eval('disutility[1][2]', cur) == 1.0
    assert pe.eval('max(2,3)+min(2,3)', cur) == 5 and pe.eval('age', cur) == 2
    assert pe.eval('2-3-4', cur) == -5 and pe.eval('2*3%4', cur) == 2 and pe.eval('1?2:0?3:4', cur) == 2
    assert parse('a*(b+c)') == ('bin', '*', ('id', 'a'), ('bin', '+', ('id', 'b'), ('id', 'c')))


SHIPPED = ['deaton1', 'deaton2', 'retirement1', 'retirement2', 'occ3']


@pytest.mark.parametrize('name', SHIPPED)
def test_generated_plugin_text_equals_the_reviewed_golden_file(name):
    """tests/golden/modelspec_<name>.h: the plugin of the shipped example as generated, committed after reading it against
    the templates of compile.m:255-551 (signatures, the token meanings of :12-64, flags :669-747, trpr :476-551).  Any change
    of the generator's output for these models shows up here and has to be re-read."""
    text = codegen.generate_modelspec(examples.REGISTRY[name]())
    path = os.path.join(HERE, 'golden', 'modelspec_%s.h' % name)
    assert os.path.exists(path), 'generate with: python tests/golden/make_modelspec_golden.py'
    assert text == open(path).read()


# ---- GPU: the device's plugin --------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize('name', MODELS)
def test_device_plugin_equals_python_evaluation_of_the_strings(name):
    from egdst_amd import build, runtime
    m = make(name)
    lib = build.build_model(m)
    s = runtime.Solver(lib, m.descriptor(), ndraw=1, keep_history=True)
    s.set_params(m.param_vector())
    rc = s.solve(raise_on_error=False)
    worst = check_call_switches(m, lambda sw, a: s.call(sw, a), s.nt, lib.info.nst, lib.info.nd)
    assert worst == 0.0
    if rc == 0:
        sol = s.solution(0)
        feas = [k for k in range(lib.info.nst) if sol.len[0, k] > 0]
        init, rs = sim_inputs(m, lib.info.nst, s.nt)
        init[:, 0] = np.asarray(feas)[np.asarray(init[:, 0], dtype=int) % len(feas)] + 1
        if name == 'retirement_hc':
            init[:, 0] = 1 + np.asarray(init[:, 0], dtype=int) % 3
        for rndtype in (0, 1):
            sims = s.simulate(init, rs, rndtype)
            assert check_simulated_columns(m, sims, init, rs, rndtype, lib.info.neq) > len(init)
    s.close()
