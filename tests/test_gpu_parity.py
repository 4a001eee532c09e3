"""GPU parity tests proper: every call goes through the C ABI of the per-model HIP library.

* shipped example models and scaled configs: bit-for-bit equal to the CPU oracle -- rows, thresholds, decisions,
  M/C/V and the evaluation count (the oracle build that shares include/egdst_math.h with the device; that header
  restates glibc's exp/log/pow bit for bit, so the glibc build of the oracle gives the same numbers);
* committed golden fixtures: tests/golden/*_portable.npz and *_native.npz (generated with the platform libm, the
  arithmetic the reference MEX runs on): both exact (full sizes: tests/test_gpu_big.py);
* simulator: bit-for-bit equal to the oracle's;
* batched draws (with ping-pong tables): every draw equals its single-draw solve;
* full-size properties (C2 batch): finite, monotone grids; evaluation counts equal the oracle's on a sample.
"""
import os

import numpy as np
import pytest

from egdst_amd import build, examples, runtime, workloads
from make_golden_cases import CASES
from oracle_harness import Oracle
from parity import compare

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def gpu_solve(model, params=None, keep_history=True):
    lib = build.build_model(model)
    P = model.param_vector()[None] if params is None else np.atleast_2d(params)
    s = runtime.Solver(lib, model.descriptor(), ndraw=len(P), keep_history=keep_history)
    s.set_params(P)
    s.solve(raise_on_error=False)
    return s


SCALED = {
    'C1': lambda: workloads.c1()[0],
    'C2': lambda: workloads.c2(a0=0)[0],
    'C2_a0neg_T60': lambda: examples.retirement2(T=60, ngridm=1000, ngridmax=10000, nthrhmax=1000, ny=10),
    'occ3_n400': lambda: examples.occ3(ngridm=400, ngridmax=4000, nthrhmax=400, ny=15),
    # 3 x 1400 points do not fit the LDS stream buffers: exercises the global-memory sort and walk
    'occ3_n1400_global_path': lambda: examples.occ3(T=12, ngridm=1400, ngridmax=14000, nthrhmax=1400, ny=10),
    # BASELINE configs[3] and [4] at the sizes SURVEY.md §8d pins them on (C5 at T=60, n=2000; tests/diag/gpu_big_configs.py
    # runs C5 at T=100, n=32768 against the oracle as well: bit-exact, 216 s of CPU)
    'C4_deaton_T80_n65536_ny21': lambda: workloads.c4()[0],
    'C5_8states_T60_n2000': lambda: workloads.c5(ngridm=2000, T=60)[0],
    'deaton_n4096': lambda: examples.deaton_sig(a0=0, mmax=50, t0=1, T=30, ngridm=4096, ngridmax=8192, ny=21),
}


@pytest.mark.parametrize('name', sorted(CASES) + sorted(SCALED))
def test_solver_bit_exact_vs_oracle(name):
    m = (CASES.get(name) or SCALED[name])()
    s = gpu_solve(m)
    sol, ref = s.solution(0), Oracle(m).solve()
    assert ref.rc == 0 and sol.status == 0, (sol.status, sol.err, sol.where)
    ok, rep = compare(sol, ref, rtol=0.0, th_tol=0.0)
    assert ok, rep
    assert sol.nevals == ref.nevals


@pytest.mark.parametrize('name', sorted(CASES))
def test_golden_fixtures(name):
    from test_oracle_math_and_golden import check_against_golden
    m = CASES[name]()
    s = gpu_solve(m)
    sol = s.solution(0)
    check_against_golden(sol, np.load(os.path.join(HERE, 'golden', name + '_portable.npz')), exact=True)
    g = np.load(os.path.join(HERE, 'golden', name + '_native.npz'))     # the reference's own arithmetic
    check_against_golden(sol, g, exact=True)
    sims = s.simulate(g['sim_init'], g['sim_rand'])
    gp = np.load(os.path.join(HERE, 'golden', name + '_portable.npz'))
    assert np.array_equal(sims, gp['sims'], equal_nan=True)
    # id/ist exact and continuous columns within 1e-10 of the glibc run (SURVEY §8d parity protocol, sim)
    assert np.array_equal(np.isnan(sims), np.isnan(g['sims']))
    assert np.array_equal(np.nan_to_num(sims[:, :, 4:6]), np.nan_to_num(g['sims'][:, :, 4:6]))
    fin = np.isfinite(g['sims'])
    assert np.all(np.abs(sims[fin] - g['sims'][fin]) <= 1e-10 * np.maximum(1, np.abs(g['sims'][fin])))


def test_cell_export_layout_matches_saveoutput():
    m = examples.retirement2()
    s = gpu_solve(m)
    sol = s.solution(0)
    M, D = s.cell_M(0, 3, 0), s.cell_D(0, 3, 0)
    assert M.shape == (sol.len[3, 0], 4) and D.shape == (sol.thlen[3, 0], 2)
    assert np.array_equal(M, sol.cell_M(3, 0), equal_nan=True) and np.array_equal(D, sol.cell_D(3, 0))
    assert M[0, 0] == m.a0 and M[0, 1] == 0 and M[0, 2] == m.a0          # row 0 = (a0, 0, a0, evf(a0))
    assert np.array_equal(M[:, 2], M[:, 0] - M[:, 1])


def test_class_surface_solve_and_sim():
    m = examples.retirement2()
    m.compile()
    sol = m.solve()
    assert len(m.M) == 1 and len(m.M[0]) == m.nt and m.M[0][0].shape[1] == 4
    sims = m.sim([[1, 0.25], [1, 5.0]])
    assert sims.shape == (2, m.nt, 14) and len(m.simlabels) == 14
    ref = Oracle(m)
    rs = ref.sim(ref.solve(), m.init, m.randstream)
    assert np.array_equal(sims, rs, equal_nan=True)
    m.setparam('duw', 0.7)                        # parameters are run-time: no recompile (compile.m:469-475)
    sol2 = m.solve()
    ok, rep = compare(sol2, Oracle(m).solve(), 0.0, 0.0)
    assert ok, rep


def test_batched_draws_equal_single_draw_solves():
    m, gen = workloads.c2(a0=0, ngridm=300, T=40)
    P = gen(12)
    orc = Oracle(m)
    for keep in (True, False):
        s = gpu_solve(m, P, keep_history=keep)
        st, _ = s.status()
        ev = s.evals()[1]
        assert np.all(st == 0)
        for i in range(len(P)):
            ref = orc.solve(P[i])
            assert ev[i] == ref.nevals
            if keep:
                ok, rep = compare(s.solution(i), ref, 0.0, 0.0)
                assert ok, (i, rep)


def test_full_size_batch_properties():
    """C2 at full size, 64 draws: every draw succeeds or fails as in the oracle (about 1 % of the parameter draws break
    the reference algorithm itself), eval counts equal the oracle's, grids monotone."""
    m, gen = workloads.c2(a0=0)
    P = gen(64)
    s = gpu_solve(m, P, keep_history=True)
    st, _ = s.status()
    ev = s.evals()[1]
    orc = Oracle(m)
    refs = [orc.solve(p) for p in P]
    assert [int(x != 0) for x in st] == [int(r.rc != 0) for r in refs], st
    assert sum(r.rc != 0 for r in refs) <= 3
    for i, r in enumerate(refs):
        if r.rc == 0:
            assert ev[i] == r.nevals, i
    sol = s.solution(33)
    assert refs[33].rc == 0
    for it in range(sol.nt):
        n = sol.len[it, 0]
        assert n >= 2 and sol.M[it, 0, 0] == m.a0 and sol.C[it, 0, 0] == 0
        assert np.all(np.diff(sol.M[it, 0, :n]) >= 0)


def test_errors_are_reported_not_hidden():
    """mmax far too small: the reference aborts with 'Could not complete initial stage in adraw()' (:1018)."""
    m = examples.deaton2(mmax=0.5, a0=0)
    s = gpu_solve(m)
    sol, ref = s.solution(0), Oracle(m).solve()
    assert ref.rc != 0 and sol.status != 0
    assert sol.err.strip() == ref.err.strip()
    assert sol.len[sol.nt - 1, 0] > 0 and sol.len[0, 0] == 0       # partial result: terminal period only
    with pytest.raises(runtime.EgdstRuntimeError):
        s.solve(raise_on_error=True)


DEGENERATE = (67, 9, 771)   # draws of workloads.c2(a0=0) on which the reference algorithm itself breaks down


def test_degenerate_draws_fail_like_the_oracle_with_pingpong_tables():
    """Draws whose guess stream re-bases at every point end in a one-row table; the reference then reads one row
    past the table's end and finds zeros (fresh matrix per period).  With two ping-pong periods the device must
    zero the rows past the new length to see the same: status, failing period and message equal the oracle's."""
    m, gen = workloads.c2(a0=0)
    P = gen(1024)[list(DEGENERATE) + [0]]
    orc = Oracle(m)
    refs = [orc.solve(p) for p in P]
    for keep in (True, False):
        s = gpu_solve(m, P, keep_history=keep)
        st, wh = s.status()
        ev = s.evals()[1]
        for i, r in enumerate(refs):
            assert (st[i] == 0) == (r.rc == 0), (keep, i, st[i], r.err)
            if r.rc:
                assert s.lib.lib.egdst_strerror(int(st[i])).decode().strip() == r.err.strip(), (keep, i)
            else:
                assert ev[i] == r.nevals


def test_draw_groups_and_compact_capacity_do_not_change_results():
    """Grouping draws on concurrent streams and the compact physical row capacity (with the exact redo of the
    draws that overflow it) are layout and scheduling choices only: per-draw status, evaluation counts,
    objective values and exported cells are bit-identical to the plain handle's."""
    m, gen = workloads.c2(a0=0, ngridm=300, T=30)
    P = gen(24)
    lib = build.build_model(m)
    base = runtime.Solver(lib, m.descriptor(), ndraw=len(P), keep_history=True)
    base.set_params(P)
    base.solve(raise_on_error=False)
    st0, ev0, ob0 = base.status()[0], base.evals()[1], base.objective()
    rows = max(int(base.solution(i).len.max()) for i in range(len(P)))
    for groups, cap in ((5, 0), (1, rows + 8), (3, rows - 4)):   # the last capacity forces redo of some draws
        s = runtime.Solver(lib, m.descriptor(), ndraw=len(P), keep_history=True, rows_cap=cap)
        s.set_groups(groups)
        s.set_params(P)
        s.solve(raise_on_error=False)
        assert np.array_equal(s.status()[0], st0) and np.array_equal(s.evals()[1], ev0)
        assert np.array_equal(s.objective(), ob0, equal_nan=True)
        if cap and cap < rows:
            assert s.capacity_retries > 0
        for i in (0, 7, 23):
            a, b = s.solution(i), base.solution(i)
            assert np.array_equal(a.len, b.len) and np.array_equal(a.M, b.M) and np.array_equal(a.V, b.V, equal_nan=True)
        s.close()


def test_draws_with_many_monotone_pieces_bit_exact():
    """C2 draws with a high disutility of work fold the worker's choice list into dozens of monotone pieces: the
    secondary envelope then walks 20-50 functions, which exercises the wave-cooperative generic step (a function per
    lane, the reference's tie rules) and the range shortcuts of the rank-merge sort.  Tables bit for bit."""
    m, gen = workloads.c2(a0=0)
    P = gen(64)[[57, 13, 11, 37]]
    s = gpu_solve(m, P, keep_history=True)
    orc = Oracle(m)
    for i in range(len(P)):
        ref = orc.solve(P[i])
        sol = s.solution(i)
        assert ref.rc == 0 and sol.status == 0
        ok, rep = compare(sol, ref, 0.0, 0.0)
        assert ok, (i, rep)
        assert sol.nevals == ref.nevals


def test_history_based_schedule_does_not_change_results():
    """After a solve the handle moves draws whose guess streams re-based many times to lanes of their own
    (egdst_set_adaptive); solves with and without it must give the same per-draw status, evaluation counts and objective
    values.  (The draws that used to dominate -- the resend fixed point, ~9000 sequential calls -- are fast-forwarded
    since round 2 and no longer register as work: their credited evaluations are checked instead.)"""
    m, gen = workloads.c2(a0=0)
    P = gen(1024)[[0, 771, 3, 982, 5, 7, 11, 13]]
    lib = build.build_model(m)
    s = runtime.Solver(lib, m.descriptor(), ndraw=len(P), keep_history=False)
    s.set_groups(2)
    s.set_params(P)
    s.solve(raise_on_error=False)
    first = (s.status()[0].copy(), s.evals()[1].copy(), s.objective().copy())
    cred = s.evals_credited()
    assert cred[1] > 50000 and cred[3] > 50000 and cred[0] == 0          # the two degenerate draws were fast-forwarded
    # (their guess streams leave a one-row table; the next period stops in valuefunc's linter_extrap, egdst_lib.c:183 -- until round 4
    #  oracle and device missed that check, went on with the one-row table and failed later with error 15)
    assert first[0][1] == 10 and first[0][3] == 10 and first[0][0] == 0
    assert s.work()[1] < 1000 and s.work()[3] < 1000
    for _ in range(2):
        s.solve(raise_on_error=False)
        assert np.array_equal(s.status()[0], first[0]) and np.array_equal(s.evals()[1], first[1])
        assert np.array_equal(s.objective(), first[2], equal_nan=True)
    s.set_adaptive(False)
    assert s.schedule()[2] == 0
    s.solve(raise_on_error=False)
    assert np.array_equal(s.status()[0], first[0]) and np.array_equal(s.evals()[1], first[1])


TP_CASES = {
    'retirement2': lambda: examples.retirement2(),
    'C2': lambda: workloads.c2(a0=0)[0],
    'C2_a0neg_T60': lambda: examples.retirement2(T=60, ngridm=1000, ngridmax=10000, nthrhmax=1000, ny=10),
    'occ3_n400': lambda: examples.occ3(ngridm=400, ngridmax=4000, nthrhmax=400, ny=15),
    'retire8': lambda: examples.retirement8(T=12, ngridm=150, ny=5),
    'model2': lambda: examples.model2(),
    'retirement_hc': lambda: examples.retirement_hc(),
    'three_points_per_choice': lambda: examples.retirement2(T=3, ngridm=3, ngridmax=40),
    'ngridmax_just_above_ngridm': lambda: examples.occ3(T=6, ngridm=30, ngridmax=31),
}


@pytest.mark.parametrize('name', sorted(TP_CASES))
@pytest.mark.parametrize('keys', ['lds_keys', 'sampled_keys'])
def test_throughput_envelope_path_bit_exact(name, keys, monkeypatch):
    """The envelope step of big batches runs as five lean kernels with one wave per walk (k_tp_prep / k_tp_sort / k_tp_walk,
    egdst_kernels.hip) and hands the cells it does not take to k_envelope.  Forced on here for single solves (EGDST_ENV_TP=1):
    tables, thresholds and evaluation counts equal the oracle's bit for bit, with the sort's M keys whole in LDS and with so
    little LDS that it works on a sampled index of them; and the path really does the cells (egdst_get_tp_stats).  (The walks over
    global memory, k_tp_walk_g, were measured slower in round 3 and are compiled into diagnostic builds only since round 4.)"""
    monkeypatch.setenv('EGDST_ENV_TP', '1')
    if keys != 'lds_keys':
        monkeypatch.setenv('EGDST_TP_SORT_LKCAP', '96')
    m = TP_CASES[name]()
    s = gpu_solve(m)
    sol = s.solution(0)
    ref = Oracle(m).solve()
    assert (sol.status == 0) == (ref.rc == 0)
    ok, rep = compare(sol, ref, rtol=0.0, th_tol=0.0)
    assert ok and sol.nevals == ref.nevals, (name, rep)
    done, left = s.tp_stats()[0]
    feasible_cells = int((sol.len[:-1] > 0).sum())          # (the terminal period is k_envelope's)
    assert done + left == s.lib.info.nst * (s.nt - 1)       # every cell of every other period went through the path's last stage once
    if name not in ('ngridmax_just_above_ngridm',):
        assert done >= 0.9 * feasible_cells, (done, left, feasible_cells)
    s.close()


def test_throughput_envelope_path_in_a_batch_with_failing_draws(monkeypatch):
    """600 draws of C2 at ngridm=300, T=30 (>= 512 cells per period: the throughput path is the default) and the same draws
    through k_envelope alone (EGDST_ENV_TP=0): status, failing period, evaluation counts, objective and -- for every draw
    -- the checksums of all cells are identical; a sample of the draws equals the oracle's tables bit for bit, and the
    draws on which the reference algorithm breaks down fail with the oracle's message."""
    m, gen = workloads.c2(a0=0, ngridm=300, T=30)
    P = gen(600)
    lib = build.build_model(m)
    res = {}
    for tp in ('default', '0'):
        if tp == '0':
            monkeypatch.setenv('EGDST_ENV_TP', '0')
        s = runtime.Solver(lib, m.descriptor(), ndraw=len(P), keep_history=True)
        s.set_params(P)
        s.solve(raise_on_error=False)
        res[tp] = (s.status(), s.evals()[1], s.objective(), [s.checksums(i) for i in range(0, len(P), 7)], s.tp_stats())
        if tp == 'default':
            orc = Oracle(m)
            st = s.status()[0]
            for i in list(range(0, 40)) + [int(k) for k in np.nonzero(st)[0][:5]]:
                ref = orc.solve(P[i])
                assert (st[i] == 0) == (ref.rc == 0), i
                if ref.rc:
                    assert s.lib.lib.egdst_strerror(int(st[i])).decode().strip() == ref.err.strip()
                else:
                    ok, rep = compare(s.solution(i), ref, 0.0, 0.0)
                    assert ok and s.evals()[1][i] == ref.nevals, (i, rep)
        s.close()
    a, b_ = res['default'], res['0']
    assert np.array_equal(a[0][0], b_[0][0]) and np.array_equal(a[0][1], b_[0][1])
    assert np.array_equal(a[1], b_[1]) and np.array_equal(a[2], b_[2], equal_nan=True)
    assert all(np.array_equal(x, y) for x, y in zip(a[3], b_[3]))
    assert a[4][:, 0].sum() > 0.95 * (a[4].sum()) and b_[4].sum() == 0      # the path did the cells / was off


EDGE = {
    'single_period_T_equals_t0': lambda: examples.deaton2(T=1, t0=1),
    'two_periods_8_points': lambda: examples.retirement2(T=2, ngridm=8),
    'three_points_per_choice': lambda: examples.retirement2(T=3, ngridm=3, ngridmax=40),
    'one_quadrature_node_8_states': lambda: examples.retirement8(T=2, ngridm=12, ny=1),
    'ngridmax_just_above_ngridm': lambda: examples.occ3(T=6, ngridm=30, ngridmax=31),
}


@pytest.mark.parametrize('name', sorted(EDGE))
def test_edge_sizes_bit_exact(name):
    m = EDGE[name]()
    s = gpu_solve(m)
    sol, ref = s.solution(0), Oracle(m).solve()
    assert ref.rc == 0 and sol.status == 0, (sol.status, sol.err, ref.err)
    ok, rep = compare(sol, ref, rtol=0.0, th_tol=0.0)
    assert ok, rep
    assert sol.nevals == ref.nevals


@pytest.mark.parametrize('nthrhmax', [1, 2])
def test_threshold_capacity_error_matches_the_reference(nthrhmax):
    """'Not enough space for thresholds' (egdst_solver.c:1327,1891): same failing period, same message, and the
    periods solved before it are exported exactly as the oracle's (the reference returns partial cells, :237)."""
    m = examples.retirement2(T=8, ngridm=60, nthrhmax=nthrhmax)
    s = gpu_solve(m)
    sol, ref = s.solution(0), Oracle(m).solve()
    assert ref.rc != 0 and sol.status == 20
    assert sol.err.strip() == ref.err.strip()
    ok, rep = compare(sol, ref, rtol=0.0, th_tol=0.0)
    assert ok, rep


@pytest.mark.parametrize('name', ['retirement2', 'occ3', 'retire8'])
def test_model_function_accessor_matches_the_oracle(name):
    """model.call / egdst_call (egdst_call.c:17-164): utility, marginal utility, discount, budget, marginal budget and
    the value function of the solved tables, with vector input, out-of-domain values and bad indices."""
    from call_cases import call_cases
    m = {'retirement2': lambda: examples.retirement2(), 'occ3': lambda: examples.occ3(),
         'retire8': lambda: examples.retirement8(T=12, ngridm=150, ny=5)}[name]()
    s = gpu_solve(m)
    orc = Oracle(m)
    ref = orc.solve()
    for sw, args in call_cases(m, s.nt, s.lib.info.nst, s.lib.info.nd):
        assert np.array_equal(s.call(sw, args), orc.call(ref, sw, args), equal_nan=True), sw


def test_class_surface_call():
    m = examples.retirement2()
    m.compile()
    m.solve()
    ref = Oracle(m)
    rsol = ref.solve()
    args = [[m.t0 + 3, 1, 2.5], [m.t0 + 3, 1, 7.0]]
    assert np.array_equal(m.call('vf', args), ref.call(rsol, 6, args))
    assert np.array_equal(m.call('u', [[m.t0, 1, 2, 1.5]]), ref.call(rsol, 1, [[m.t0, 1, 2, 1.5]]))
    from egdst_amd import EgdstError
    with pytest.raises(EgdstError):
        m.call('nonsense', args)


@pytest.mark.parametrize('name', ['retirement2', 'retire8', 'retirement_hc'])
@pytest.mark.parametrize('how', ['cells', 'bulk'])
def test_import_solution_then_simulate_and_call_without_solving(name, how):
    """Row (b) of SURVEY section 8: the reference's simulator and accessor gateways take M and D from the model object
    (egdst_simulator.c:61-68, egdst_call.c:28-34), not from a solve in the same process.  A handle that NEVER solved is given
    the exported cells (egdst_set_cell_M / _D, the calls shims/egdst_simulator_hip.c makes; or egdst_set_solution) and must
    then return the simulated paths, every accessor value, the cell export and the checksums of the handle that solved."""
    from call_cases import call_cases
    m = {'retirement2': lambda: examples.retirement2(), 'retire8': lambda: examples.retirement8(T=12, ngridm=150, ny=5),
         'retirement_hc': lambda: examples.retirement_hc()}[name]()
    s = gpu_solve(m)
    sol = s.solution(0)
    assert sol.status == 0
    f = runtime.Solver(s.lib, m.descriptor(), ndraw=1, keep_history=True)
    with pytest.raises(runtime.EgdstRuntimeError):
        f.simulate(np.array([[1, 1.0]]), np.random.default_rng(0).random(4 * s.nt))   # not solved, nothing imported
    f.set_params(m.param_vector())
    if how == 'cells':
        M, D = sol.cells()
        f.set_cells(M, D)
    else:
        f.set_solution(sol)
    nst = s.lib.info.nst
    rng = np.random.default_rng(11)
    nsim = 64
    feas = [ist for ist in range(nst) if sol.len[0, ist] > 0]
    init = np.column_stack([rng.choice(feas, nsim) + 1.0, rng.uniform(m.a0, m.mmax, nsim)])
    if name == 'retirement_hc':   # (initial states in the lower half of the continuous grid: see DESIGN.md section 8, N4)
        init = np.column_stack([rng.integers(1, 4, nsim), rng.uniform(0.2, 9, nsim)])
    rs = rng.random(4 * s.nt * nsim)
    for rndtype in (0, 1):
        a, b = s.simulate(init, rs, rndtype), f.simulate(init, rs, rndtype)
        assert np.array_equal(a, b, equal_nan=True) and np.isfinite(a[:, 0, 0]).all()
    for sw, args in call_cases(m, s.nt, nst, s.lib.info.nd):
        assert np.array_equal(s.call(sw, args), f.call(sw, args), equal_nan=True), sw
    assert np.array_equal(s.checksums(0), f.checksums(0))
    for it in (0, s.nt // 2, s.nt - 1):
        for ist in range(nst):
            assert np.array_equal(s.cell_M(0, it, ist), f.cell_M(0, it, ist)) and np.array_equal(s.cell_D(0, it, ist), f.cell_D(0, it, ist))
    # rows past a table's end are zero on the importing handle too (the reference reads one past a one-row table)
    g = f.solution(0)
    for it in range(s.nt):
        for ist in range(nst):
            assert not g.M[it, ist, g.len[it, ist]:].any() and not g.TH[it, ist, g.thlen[it, ist]:].any()
    # importing over a solved handle replaces the solution of that draw (and a later solve replaces the import)
    s2 = gpu_solve(m, params=m.param_vector() * 1.01)
    s2.set_solution(sol)
    assert np.array_equal(s2.checksums(0), s.checksums(0))
    s2.solve()
    assert not np.array_equal(s2.checksums(0), s.checksums(0))
    for h in (s, f, s2):
        h.close()


def test_import_state_is_per_draw():
    """A handle that was never solved and received the cells of ONE draw simulates that draw only: the others keep status
    EGDST_E_NOT_SOLVED (they would otherwise simulate from empty tables); and the first import into a draw whose solve FAILED clears
    the failed solve's other cells, so a half-imported draw is the import plus empty cells, never a mixture."""
    m = examples.retirement2()
    s = gpu_solve(m)
    sol = s.solution(0)
    f = runtime.Solver(s.lib, m.descriptor(), ndraw=3, keep_history=True)
    f.set_params(np.tile(m.param_vector(), (3, 1)))
    f.set_solution(sol, draw=1)
    assert f.status()[0].tolist() == [40, 0, 40]
    assert np.array_equal(f.checksums(1), s.checksums(0))
    f.close()
    s.close()
    m, gen = workloads.c2(a0=0)
    P = gen(1024)[[0, DEGENERATE[0]]]
    g = gpu_solve(m, P)
    nt, nst = g.nt, g.lib.info.nst
    assert g.status()[0][0] == 0 and g.status()[0][1] != 0
    good, bad = g.solution(0), g.solution(1)
    assert bad.len.sum() > 0        # (the failed draw stopped in the middle: its later periods hold tables)
    before = g.checksums(0).copy()
    it, ist = nt - 1, 0
    c = np.asfortranarray(g.cell_M(0, it, ist))
    g.lib.check(g.lib.lib.egdst_set_cell_M(g.h, 1, it, ist, c.shape[0], c.ctypes.data_as(runtime.C.POINTER(runtime.C.c_double))))
    assert g.status()[0].tolist() == [0, 0]
    h = g.solution(1)
    assert h.len[it, ist] == c.shape[0] and h.len.sum() == c.shape[0] and h.thlen.sum() == 0
    assert np.array_equal(g.checksums(0), before)   # the other draw is untouched
    g.close()


def test_import_rejects_what_does_not_fit():
    m = examples.retirement2()
    s = gpu_solve(m)
    f = runtime.Solver(s.lib, m.descriptor(), ndraw=1, keep_history=False)
    with pytest.raises(runtime.EgdstRuntimeError):
        f.set_solution(s.solution(0))   # needs keep_history=1
    f.close()
    f = runtime.Solver(s.lib, m.descriptor(), ndraw=1, keep_history=True)
    big = np.zeros((m.ngridmax + 5, 4), order='F')
    with pytest.raises(runtime.EgdstRuntimeError):
        f.lib.check(f.lib.lib.egdst_set_cell_M(f.h, 0, 0, 0, big.shape[0], big.ctypes.data_as(runtime.C.POINTER(runtime.C.c_double))))
    with pytest.raises(runtime.EgdstRuntimeError):
        f.lib.check(f.lib.lib.egdst_set_cell_M(f.h, 0, s.nt, 0, 0, None))   # period out of range
    f.close()
    s.close()


def test_class_surface_sim_and_call_from_assigned_cells():
    """egdstmodel is ConstructOnLoad (egdstmodel.m:1): a model whose M / D were restored (assigned) simulates and answers
    `call` without solving; and `sim` after `setparam` uses the NEW parameter values with the OLD solution, as the
    reference's gateway does (loadparameters at every call, egdst_simulator.c:60)."""
    m = examples.retirement2()
    m.compile()
    sol = m.solve()
    init = np.array([[1, 0.25], [1, 3.0], [1, 8.0]])
    m.randstream = np.random.default_rng(7).random(4 * m.nt * len(init))
    sims = m.sim(init)
    vf = m.call('vf', [[m.t0 + 3, 1, 2.5]])
    m2 = examples.retirement2()
    m2.compile()
    m2.M, m2.D = sol.cells()        # e.g. read back from disk
    m2.randstream = m.randstream
    assert np.array_equal(m2.sim(init), sims, equal_nan=True)
    assert np.array_equal(m2.call('vf', [[m.t0 + 3, 1, 2.5]]), vf)
    # a cell replaced INSIDE the restored lists is noticed too (the reference reads the cells on every call)
    it, ist = 3, 0
    old = m2.M[ist][it]
    changed = old.copy()
    changed[:, 3] += 1.0            # the value column of one table
    m2.M[ist][it] = changed
    assert not np.array_equal(m2.call('vf', [[m.t0 + 3, 1, 2.5]]), vf)
    m2.M[ist][it] = old
    assert np.array_equal(m2.call('vf', [[m.t0 + 3, 1, 2.5]]), vf)
    ref = Oracle(m)
    rsol = ref.solve()
    m.setparam('wage', 1.2)         # new wage, old solution
    refm = examples.retirement2()
    refm.setparam('wage', 1.2)
    expect = Oracle(refm).sim(rsol, init, m.randstream)
    assert np.array_equal(m.sim(init), expect, equal_nan=True)
    assert not np.array_equal(expect, sims, equal_nan=True)


def test_simulated_moments_on_device():
    """egdst_simulate_moments: per-period means of the simulated columns computed on the device equal
    the means of the oracle's simulated paths (summation order differs: 1e-13 relative), the counts exactly."""
    m = examples.retirement2()
    s = gpu_solve(m)
    ref = Oracle(m)
    rsol = ref.solve()
    rng = np.random.default_rng(21)
    nsim = 5000
    init = np.column_stack([np.ones(nsim), rng.uniform(m.a0 - 1, m.mmax + 1, nsim)])   # a few outside [a0, mmax]
    rs = rng.random(4 * s.nt * nsim)
    means, counts = s.simulate_moments(init, rs)
    rsims = ref.sim(rsol, init, rs)
    rcounts = (~np.isnan(rsims)).sum(axis=0)
    with np.errstate(all='ignore'):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            rmeans = np.nanmean(rsims, axis=0)
    assert np.array_equal(counts, rcounts)
    assert np.array_equal(np.isnan(means), np.isnan(rmeans))
    fin = np.isfinite(rmeans)
    assert np.all(np.abs(means[fin] - rmeans[fin]) <= 1e-13 * np.maximum(1, np.abs(rmeans[fin])))
    assert 0 < counts[0, 0] < nsim and counts[-1, 0] <= counts[0, 0]


def test_device_math_equals_host_libm():
    """exp / log / pow on the device == glibc's (x86-64 FMA variant), bit for bit: against vectors committed from this
    image's libm (tests/golden/math_vectors.npz) and against the GPU box's own libm through ctypes."""
    from test_math_vs_libm import _glibc_fma_variant
    sys_path = os.path.join(HERE, 'golden')
    import sys
    sys.path.insert(0, sys_path)
    from make_math_vectors import libm
    lib = build.build_model(examples.occ3())
    g = np.load(os.path.join(HERE, 'golden', 'math_vectors.npz'))
    e, l, p = lib.math_eval('exp', g['exp_x']), lib.math_eval('log', g['log_x']), lib.math_eval('pow', g['pow_a'], g['pow_b'])
    assert np.array_equal(e, g['exp_y'], equal_nan=True)
    assert np.array_equal(l, g['log_y'], equal_nan=True)
    assert np.array_equal(p, g['pow_y'], equal_nan=True)
    assert np.array_equal(np.signbit(p), np.signbit(g['pow_y']))       # (-0.0 vs 0.0)
    if _glibc_fma_variant():
        m = libm()
        x = np.random.default_rng(5).uniform(-30, 30, 20000)
        assert np.array_equal(lib.math_eval('exp', x), np.array([m.exp(v) for v in x]))
        assert np.array_equal(lib.math_eval('log', np.abs(x)), np.array([m.log(abs(v)) for v in x]))
        assert np.array_equal(lib.math_eval('pow', np.abs(x), x / 7), np.array([m.pow(abs(v), v / 7) for v in x]))


def test_shared_reciprocal_interpolation_is_the_ieee_expression():
    """The grid kernels interpolate with ONE refined reciprocal for the two quotients of linter's expression (eg_lerp_fast, round 4).
    On 4 million operand sets -- node spacings from 1e-13 to 1e4, abscissae inside and far outside the bracket, ordinates of both
    signs up to 1e300, exact hits of a node (zero numerators) -- the device's result equals the host's IEEE evaluation of
    f1 (x - g0) / (g1 - g0) + f0 (g1 - x) / (g1 - g0) (numpy: correctly rounded divisions, no contraction) bit for bit; and for
    divisors outside the fast form's domain (below 2^-300, zero, negative, NaN) and results beyond the finite range it returns
    what the plain device expression returns."""
    lib = build.build_model(examples.retirement2())
    rng = np.random.default_rng(2024)
    n = 1 << 22
    g0 = rng.uniform(-5, 50, n)
    d = 10.0 ** rng.uniform(-13, 4, n)
    g1 = g0 + d
    x = np.where(rng.random(n) < 0.7, g0 + rng.random(n) * (g1 - g0), g0 + rng.normal(0, 5, n) * (g1 - g0))
    mag = 10.0 ** rng.uniform(-8, 8, n)
    f0, f1 = rng.normal(0, 1, n) * mag, rng.normal(0, 1, n) * mag
    k = n // 16
    x[:k] = g0[:k]                      # exact node: a zero numerator
    x[k:2 * k] = g1[k:2 * k]
    f0[2 * k:3 * k] *= 1e292            # large ordinates (finite results)
    f1[2 * k:3 * k] *= 1e292
    with np.errstate(all='ignore'):
        want = f1 * (x - g0) / (g1 - g0) + f0 * (g1 - x) / (g1 - g0)
    ok = np.isfinite(want) & ((g1 - g0) >= 2.0 ** -300)
    got = lib.lerp_eval(x, g0, g1, f0, f1, shared=True)
    assert ok.mean() > 0.99
    assert np.array_equal(got[ok], want[ok])
    plain = lib.lerp_eval(x, g0, g1, f0, f1, shared=False)
    assert np.array_equal(plain[ok], want[ok])
    # outside the domain of the fast form: whatever the plain expression gives on the device
    m = 4096
    g0 = rng.uniform(-5, 50, m)
    g1 = g0.copy()
    g1[: m // 4] = g0[: m // 4]                                  # zero divisor
    g1[m // 4: m // 2] = g0[m // 4: m // 2] - 10.0 ** rng.uniform(-10, 2, m // 4)   # negative divisor
    g0[m // 2: 3 * m // 4] = 0.0
    g1[m // 2: 3 * m // 4] = 2.0 ** rng.uniform(-1070, -301, m // 4)   # tiny divisor
    g1[3 * m // 4:] = np.nan
    x = g0 + rng.normal(0, 1, m)
    f0, f1 = rng.normal(0, 1e300, m), rng.normal(0, 1e300, m)
    a, b = lib.lerp_eval(x, g0, g1, f0, f1, shared=True), lib.lerp_eval(x, g0, g1, f0, f1, shared=False)
    assert np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize('name', ['retirement2', 'occ3', 'retire8', 'deaton2', 'model2', 'C2', 'C2_a0neg_T60', 'occ3_n400'])
@pytest.mark.parametrize('lds', ['lds', 'lds_cv', 'sampled', 'general'])
def test_batch_grid_kernels_on_single_draws(name, lds, monkeypatch):
    """A single draw normally takes k_grid_wide (16 lanes per point).  Forced through the kernels big batches use --
    k_grid_lds (searched columns in LDS, one bracket search per evaluation, shared shock nodes) and the general k_grid --
    every model must still equal the oracle bit for bit."""
    monkeypatch.setenv('EGDST_GRID_WIDE', '0')
    # 2048 rows of LDS hold whole columns; 64 force the sampled index (every k-th row in LDS, the search finished in the
    # global column: the form C4 and C5 take); 0 is the general kernel with two searches in global memory
    # lds_cv: the C and V columns in LDS as well (k_grid_lds_cv: no global read in the loop over the shock nodes)
    monkeypatch.setenv('EGDST_GRID_LDS', {'general': '0', 'sampled': '64', 'lds': '2048', 'lds_cv': '2048'}[lds])
    monkeypatch.setenv('EGDST_GRID_CV', '1' if lds == 'lds_cv' else '0')
    m = (CASES.get(name) or SCALED[name])()
    s = gpu_solve(m)
    sol, ref = s.solution(0), Oracle(m).solve()
    assert ref.rc == 0 and sol.status == 0, (sol.status, sol.err, sol.where)
    ok, rep = compare(sol, ref, rtol=0.0, th_tol=0.0)
    assert ok, rep
    assert sol.nevals == ref.nevals


@pytest.mark.parametrize('name', ['retirement2', 'occ3_n400', 'retire8', 'C2', 'deaton_n4096', 'cake_normal'])
def test_branch_free_bracket_search_is_the_binary_search(name, monkeypatch):
    """k_grid_lds on single draws (EGDST_GRID_WIDE=0): the bracket of every evaluation comes from the branch-free bisection with a
    uniform step count (eg_bracket_sorted, eg_bracket_sampled: round 4), which on an ordered column must land where the reference's
    bxsearch_common lands (egdst_lib.c:136-166) -- and the two divisions of every interpolation share one refined reciprocal
    (eg_lerp_fast).  Tables, thresholds and evaluation counts equal the oracle's bit for bit, with whole columns in LDS and with
    the sampled index (the search is then finished in the window of the global column).  (The neighbour-hinted search this test
    covered until round 3, k_grid_lds_n, was measured slower and is compiled into diagnostic builds only.)"""
    monkeypatch.setenv('EGDST_GRID_WIDE', '0')
    m = {'retirement2': lambda: examples.retirement2(), 'occ3_n400': SCALED['occ3_n400'], 'retire8': lambda: examples.retirement8(T=12, ngridm=150, ny=5),
         'C2': SCALED['C2'], 'deaton_n4096': SCALED['deaton_n4096'], 'cake_normal': lambda: examples.cake_normal()}[name]()
    ref = Oracle(m).solve()
    for lds in ('', '96'):
        if lds:
            monkeypatch.setenv('EGDST_GRID_LDS', lds)
        s = gpu_solve(m)
        sol = s.solution(0)
        ok, rep = compare(sol, ref, rtol=0.0, th_tol=0.0)
        assert ok and sol.nevals == ref.nevals, (name, lds, rep)
        s.close()


@pytest.mark.parametrize('name', ['cake_normal', 'retirement_mortal'])
@pytest.mark.parametrize('rndtype', [0, 1])
def test_normal_shocks_mortality_and_shared_streams(name, rndtype):
    """The branches no shipped script reaches: additive normal shocks (DISTRIB=2: rescale = mu + x*sigma, egdst_lib.c:84-100;
    model_cake1.m forms with sigma > 0), a survival probability below one (death draw, egdst_simulator.c:265: the rest of
    the path stays NaN), and rndtype=1 (every agent re-uses the head of the stream, egdst_simulator.c:109-114).
    Solution and simulated panel bit for bit against the oracle."""
    m = examples.REGISTRY[name]()
    s = gpu_solve(m)
    orc = Oracle(m)
    ref = orc.solve()
    sol = s.solution(0)
    assert ref.rc == 0 and sol.status == 0
    ok, rep = compare(sol, ref, rtol=0.0, th_tol=0.0)
    assert ok, rep
    assert sol.nevals == ref.nevals
    nsim = 300
    rng = np.random.default_rng(4)
    init = np.column_stack([np.ones(nsim), rng.uniform(m.a0, m.mmax, nsim)])
    rs = rng.random(4 * m.nt * (1 if rndtype == 1 else nsim))
    sims, rsims = s.simulate(init, rs, rndtype=rndtype), orc.sim(ref, init, rs, rndtype=rndtype)
    assert np.array_equal(sims, rsims, equal_nan=True)
    alive = (~np.isnan(sims[:, :, 0])).sum(axis=0)
    if name == 'retirement_mortal':
        assert alive[0] == nsim and (alive[-1] < alive[0] if rndtype == 0 else alive[-1] in (0, nsim))  # shared stream: all die together
    else:
        assert np.all(alive == nsim)
    if rndtype == 1:   # too short a stream is refused with the gateway's message (egdst_simulator.c:71-75)
        with pytest.raises(runtime.EgdstRuntimeError) as e:
            s.simulate(init, rs[:4 * m.nt - 1], rndtype=1)
        assert e.value.code == 41


@pytest.mark.parametrize('rndtype', [0, 1])
def test_estimation_step_on_device(rndtype):
    """egdst_simulate_batch_moments: every draw of a batch simulated on the device with generated uniforms
    (common random numbers), moments reduced per draw, objective = weighted distance to target moments -- against the
    oracle fed with the host replay of the same uniforms: counts exact, means 1e-13, objective 1e-11; a draw that fails to
    solve has a NaN objective."""
    import estimation_case
    m, gen = workloads.c2(a0=0, ngridm=300, T=30)
    P = gen(1024)[[0, 1, 2, 3, 5, 8, 13, 771]]
    s = gpu_solve(m, P)
    rng = np.random.default_rng(5)
    nsim = 2000
    init = np.column_stack([np.ones(nsim), rng.uniform(m.a0 - 0.5, m.mmax + 0.5, nsim)])
    bad = estimation_case.check(s, Oracle(m), P, init, seed=99 + rndtype, rndtype=rndtype, lib=s.lib)
    assert not bad, bad


@pytest.mark.parametrize('name', ['retirement2', 'occ3', 'retire8', 'model2', 'C2', 'occ3_n400'])
def test_solver_gateway_third_output_dbgout(name):
    """[M,D,dbgout] = egdst_solver(model): the kink log of thresholds() (egdst_solver.c:178-181,1866-1879) -- period,
    state, choice of the secondary envelope or -1, threshold, consumption either side, jump -- row for row equal to the
    oracle's, zero rows after the recorded ones, and the solve itself unchanged by the logging."""
    m = (CASES.get(name) or SCALED[name])()
    lib = build.build_model(m)
    s = runtime.Solver(lib, m.descriptor(), ndraw=1, keep_history=True)
    s.set_dbgout(True)
    s.set_params(m.param_vector()[None])
    s.solve(raise_on_error=False)
    ref = Oracle(m).solve(dbgout=True)
    out, n = s.dbgout(0)
    assert n == ref.dbgn and n > 0
    assert out.shape == ref.dbgout.shape and np.array_equal(out, ref.dbgout)
    assert np.all(out[n:] == 0)
    ok, rep = compare(s.solution(0), ref, rtol=0.0, th_tol=0.0)
    assert ok, rep
    th = np.concatenate([ref.TH[it, ist, 1:ref.thlen[it, ist]] for it in range(ref.len.shape[0] - 1, -1, -1) for ist in range(ref.len.shape[1])])
    assert np.array_equal(out[:n][out[:n, 2] == -1, 3], th)   # the primary rows are the thresholds of the D cells


def test_class_surface_dbgout():
    m = examples.retirement2()
    m.compile()
    m.solve(dbgout=True)
    assert m.dbgout.shape == (m.nt * 1 * 2 * 2 * m.nt, 7) and m.dbgout[0, 0] == m.nt - 2 and m.dbgout[0, 2] == -1


def test_continuous_state_model_solver_simulator_accessor():
    """a continuous state on a grid with deterministic motion rules.  Solver: the transition weights of
    trpr(..., all=1) (compile.m:527-538).  Simulator: exact state values, consumption blended over the 2^k grid corners,
    the (state, decision) pair drawn by the weights (egdst_simulator.c:313-372), including the gateway's failure for
    initial states beyond the middle of the grid.  Bit for bit against the oracle."""
    m = examples.retirement_hc()
    s = gpu_solve(m)
    orc = Oracle(m)
    ref = orc.solve()
    sol = s.solution(0)
    assert ref.rc == 0 and sol.status == 0
    ok, rep = compare(sol, ref, rtol=0.0, th_tol=0.0)
    assert ok, rep
    assert sol.nevals == ref.nevals
    rng = np.random.default_rng(3)
    nsim = 500
    init = np.column_stack([rng.integers(1, 4, nsim), rng.uniform(0.2, 9, nsim)])
    rs = rng.random(4 * m.nt * nsim)
    for rndtype in (0, 1):
        a, b_ = s.simulate(init, rs, rndtype=rndtype), orc.sim(ref, init, rs, rndtype=rndtype)
        assert np.array_equal(a, b_, equal_nan=True)
    hc = a[:, :, 11]
    assert hc.min() >= 0 and hc.max() <= 2 and len(np.unique(hc)) > 10       # the state really is continuous (off-grid values)
    with pytest.raises(runtime.EgdstRuntimeError):
        s.simulate(np.array([[5, 2.0]]), rs)
    from call_cases import call_cases
    for sw, args in call_cases(m, s.nt, s.lib.info.nst, s.lib.info.nd):
        assert np.array_equal(s.call(sw, args), orc.call(ref, sw, args), equal_nan=True), sw


def test_segmented_envelope_walks_are_used_and_exact(monkeypatch):
    """The envelope walk of a cell is cut at predictable points into one segment per wave, the predictions are checked
    and the segments merged (run_walk in egdst_kernels.hip); EGDST_NOSEG=1 keeps the one-wave walk.  Both ways equal the
    oracle bit for bit (C2 and C3 at full size: LDS-resident and global-memory streams), the statistics show that the
    cutting really happens, and fallbacks (a failed prediction is not an error) stay rare."""
    for name in ('C2', 'occ3_n1400_global_path'):
        m = SCALED[name]()
        ref = Oracle(m).solve()
        for noseg in ('0', '1'):
            monkeypatch.setenv('EGDST_NOSEG', noseg)
            s = gpu_solve(m)
            sol = s.solution(0)
            ok, rep = compare(sol, ref, rtol=0.0, th_tol=0.0)
            assert ok and sol.nevals == ref.nevals, (name, noseg, rep)
            merged, fallback = s.walk_stats()[0]
            if noseg == '1':
                assert merged == 0 and fallback == 0
            else:
                assert merged >= sol.nt // 2 and fallback <= merged // 4, (name, merged, fallback)
            s.close()


# the build variants the tests of this file load besides the default libraries: (model factory, extra hipcc flags).
# __graft_entry__.build() compiles them in-tree beforehand, so that the GPU lease never runs hipcc.
BUILD_VARIANTS = [(lambda: TP_CASES['C2'](), ['-DEGDST_RANKCHK']), (lambda: TP_CASES['occ3_n400'](), ['-DEGDST_RANKCHK']),
                  (lambda: TP_CASES['occ3_n400'](), ['-DEGDST_RANKCHK', '-DENV_RK=4']),
                  (lambda: workloads.c2(a0=0, ngridm=300, T=30)[0], ['-DENV_LANE_STEP=0', '-DENV_CDEFER_ON=0'])]


@pytest.mark.parametrize('name,tp,rk', [('C2', '0', 1), ('occ3_n400', '1', 1), ('occ3_n400', '0', 1), ('occ3_n400', '0', 4)])
def test_every_rank_of_the_sort_equals_a_plain_count(name, tp, rk, monkeypatch):
    """The checking build of the sort (-DEGDST_RANKCHK): every rank that the rank-merge of the envelope step assigns from
    binary searches (eg_rank_classify_run: comp1 order of egdst_solver.c:1570-1582 extended by the input index) is compared on
    the device with a plain count over the whole stream -- C2 at full size through k_envelope (the secondary envelopes of
    lists with several folds), a three-choice model through the throughput path and through k_envelope: thousands of ranks
    checked, none differs, and the solution is still the oracle's.  (Two-list merges take the merge path, which assigns
    positions, not ranks.)  rk = 4: the C5 batch build's form (-DENV_RK=4: a thread ranks four consecutive points, the later ones
    galloping on from their predecessor's count) with an LDS budget so small that the keys are a sampled index and the searches end
    in global memory, as on C5's 65 536-point streams."""
    monkeypatch.setenv('EGDST_ENV_TP', tp)
    if rk > 1:
        monkeypatch.setenv('EGDST_LCAP', '200')
    m = TP_CASES[name]()
    lib = build.build_model(m, extra_flags=['-DEGDST_RANKCHK'] + (['-DENV_RK=%d' % rk] if rk > 1 else []))
    s = runtime.Solver(lib, m.descriptor(), ndraw=1, keep_history=True)
    s.set_params(m.param_vector()[None])
    assert s.solve(raise_on_error=False) == 0
    dbg = s.debug(0)
    assert dbg[12] > 1000 and dbg[13] == 0, dbg.tolist()
    ref = Oracle(m).solve()
    ok, rep = compare(s.solution(0), ref, rtol=0.0, th_tol=0.0)
    assert ok, rep
    s.close()


def test_walk_variants_agree_on_a_batch(monkeypatch):
    """The envelope walk's generic step with the functions' state on the lanes (env_step_lanes) and the consumption column copied
    after the walk (ENV_CDEFER) -- the default build -- against the build with env_step_wave and the copy inside the batches
    (-DENV_LANE_STEP=0 -DENV_CDEFER_ON=0): 600 draws of C2 at ngridm=300, T=30, through the throughput path and through
    k_envelope alone: status, failing period, evaluation counts, objective and the checksums of every cell of a sample of draws."""
    m, gen = workloads.c2(a0=0, ngridm=300, T=30)
    P = gen(600)
    res = {}
    for name, flags in (('default', []), ('old', ['-DENV_LANE_STEP=0', '-DENV_CDEFER_ON=0'])):
        lib = build.build_model(m, extra_flags=flags)
        for tp in ('1', '0'):
            monkeypatch.setenv('EGDST_ENV_TP', tp)
            s = runtime.Solver(lib, m.descriptor(), ndraw=len(P), keep_history=True)
            s.set_params(P)
            s.solve(raise_on_error=False)
            res[name, tp] = (s.status(), s.evals()[1], s.objective(), [s.checksums(i) for i in range(0, len(P), 5)])
            s.close()
    a = res['default', '1']
    for k, b_ in res.items():
        assert np.array_equal(a[0][0], b_[0][0]) and np.array_equal(a[0][1], b_[0][1]), k
        assert np.array_equal(a[1], b_[1]) and np.array_equal(a[2], b_[2], equal_nan=True), k
        for x, y in zip(a[3], b_[3]):
            assert np.array_equal(x, y), k


@pytest.mark.parametrize('mode', ['fast', 'defer_all', 'defer_late', 'off'])
def test_single_choice_envelope_tiles_and_hand_over(mode, monkeypatch):
    """k_env1 (round 4: one pass, tiles of 1024 candidates with decoupled look-back) on a batch of single-choice draws with five
    tiles per cell and on deaton2 as shipped (regenerated streams: k_envelope's from the start): tables equal the oracle's bit for
    bit with the history kept, and with two ping-pong periods status, evaluation counts and the checksums of the two live periods
    equal the kept-history solve's -- on the fast path, with every cell handed to k_envelope before (EGDST_E1_DEFER_ALL=1) and AFTER
    (=2) its tiles wrote rows into the table (the high-water marks must then cover them), and with k_env1 off."""
    if mode == 'defer_all':
        monkeypatch.setenv('EGDST_E1_DEFER_ALL', '1')
    if mode == 'defer_late':
        monkeypatch.setenv('EGDST_E1_DEFER_ALL', '2')
    if mode == 'off':
        monkeypatch.setenv('EGDST_NO_ENV1', '1')
    m, gen = workloads.c4(ngridm=5000, T=12, ny=7)
    P = gen(6)
    orc = Oracle(m)
    s = gpu_solve(m, P)
    for d in range(len(P)):
        ref = orc.solve(P[d])
        sol = s.solution(d)
        assert ref.rc == 0 and sol.status == 0
        ok, rep = compare(sol, ref, rtol=0.0, th_tol=0.0)
        assert ok, (d, rep)
        assert sol.nevals == ref.nevals
    p = gpu_solve(m, P, keep_history=False)     # two ping-pong slots: a slot's rows past the new end must be cleared every period
    assert np.array_equal(p.status()[0], s.status()[0]) and np.array_equal(p.evals()[1], s.evals()[1])
    assert np.array_equal(p.objective(), s.objective(), equal_nan=True)
    p.solve(raise_on_error=False)               # once more on the same slots (leftovers of the previous solve)
    assert np.array_equal(p.objective(), s.objective(), equal_nan=True) and np.array_equal(p.evals()[1], s.evals()[1])
    p.close()
    s.close()
    m2 = examples.deaton2()
    s2 = gpu_solve(m2)
    ok, rep = compare(s2.solution(0), Oracle(m2).solve(), rtol=0.0, th_tol=0.0)
    assert ok, rep
    s2.close()


def test_cycle_of_the_zero_consumption_resend_is_accounted_for_not_executed():
    """C2 on the surveyed credit limit: a draw whose guess stream (it=14, worker) enters a CYCLE of the resend -- the re-sent guess
    signals c1<=0 at another shock node, the guess prepared for that node signals it at the first one again -- which the reference
    follows for ngridmax calls until its runaway guard (egdst_solver.c:963-978).  The device recognises the period (two to four
    visits) and credits the remaining calls; status, error text and the solved cells must be the oracle's, which executes them."""
    from oracle_harness import Oracle
    from make_golden_big import cell_sums
    m, gen = workloads.c2()
    P = gen(4096)
    P = P * (1 + 0.005 * (2 * np.random.default_rng(3).random(P.shape) - 1))
    idx = [2833, 0, 1, 2]
    lib = build.build_model(m)
    s = runtime.Solver(lib, m.descriptor(), ndraw=len(idx), keep_history=True)
    s.set_params(P[idx])
    s.solve(raise_on_error=False)
    st = s.status()[0]
    assert s.evals_credited()[0] > 50000 and s.work()[0] < 1000      # ~10 000 calls of ~10 evaluations credited, none executed
    orc = Oracle(m)
    for k, d in enumerate(idx):
        r = orc.solve(P[d])
        assert (r.rc == 0) == (st[k] == 0), (d, st[k], r.err)
        if r.rc:
            assert lib.lib.egdst_strerror(int(st[k])).decode().strip() == r.err.strip(), d
        else:
            assert s.evals()[1][k] == r.nevals, d
        ln, th = s.dims(k)
        assert np.array_equal(ln, r.len) and np.array_equal(th, r.thlen), d
        assert np.array_equal(s.checksums(k), cell_sums(r)), d
    s.close()
