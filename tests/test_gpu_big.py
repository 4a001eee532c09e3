"""GPU parity at BASELINE.json's full sizes against the committed fixtures of the glibc oracle (tests/golden/big_*.npz):
rows and thresholds of every cell, evaluation count and the per-cell checksums of M, C, V, TH, D -- i.e. the complete
solution bit for bit -- without running the oracle on the GPU box (C5 at T=100, n=32768 takes it 220 s per draw).
Checksums are computed on the device (egdst_get_checksums); for the smaller cases the exported tables are checksummed on
the host as well, which ties the device kernel to numpy."""
import os
import sys

import numpy as np
import pytest

from egdst_amd import build, runtime, workloads

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
from make_golden_big import BIG, cell_sums  # noqa: E402
from test_big_fixtures import check_solution_against_fixture, load  # noqa: E402


class _Dims:
    def __init__(self, ln, th, nevals):
        self.len, self.thlen, self.nevals = ln, th, nevals


def solve_batch(m, P, keep_history=True):
    lib = build.build_model(m)
    P = np.atleast_2d(np.asarray(P, dtype=np.float64))
    s = runtime.Solver(lib, m.descriptor(), ndraw=len(P), keep_history=keep_history)
    s.set_params(P)
    s.solve(raise_on_error=False)
    return s


def check_draw(s, draw, g):
    assert s.status()[0][draw] == 0
    ln, th = s.dims(draw)
    check_solution_against_fixture(_Dims(ln, th, s.evals()[1][draw]), g, sums=s.checksums(draw))


@pytest.mark.parametrize('name', ['C1', 'C2', 'C2_a0m5', 'C3', 'C4', 'C5_T60_n2000'])
def test_full_size_configs_equal_the_glibc_fixtures(name):
    g = load(name)
    m, par = BIG[name][0]()
    s = solve_batch(m, m.param_vector() if par is None else par)
    check_draw(s, 0, g)
    if name in ('C1', 'C2', 'C2_a0m5', 'C3'):
        sol = s.solution(0)
        assert np.array_equal(cell_sums(sol), s.checksums(0))   # the device checksum is the host's
        nt, nst = sol.len.shape
        for it in range(nt):
            for ist in range(nst):
                if sol.len[it, ist]:
                    assert sol.M[it, ist, sol.len[it, ist] - 1] == g['lastM'][it, ist]
    s.close()


def test_c5_full_size_four_draws_one_batch():
    """BASELINE configs[4] at T=100, n=32768, ny=15: draws 0..3 of the 1024-draw batch solved together."""
    m, gen = workloads.c5()
    P = gen(1024)[:4]
    s = solve_batch(m, P)
    for i in range(4):
        g = load('C5_full_draw%d' % i)
        assert np.array_equal(g['params'], P[i])
        check_draw(s, i, g)
    s.close()


def test_c4_per_gpu_share_32_draws():
    """BASELINE configs[3]: 256 draws over 8 GPUs = 32 per GPU.  Draw 5 equals its fixture; every draw solves, and the
    batch result of a draw equals its single-draw solve (checksums)."""
    m, gen = workloads.c4()
    P = gen(256)[:32]
    s = solve_batch(m, P)
    assert np.all(s.status()[0] == 0)
    check_draw(s, 5, load('C4_draw5'))
    one = solve_batch(m, P[17])
    assert np.array_equal(one.checksums(0), s.checksums(17)) and one.evals()[1][0] == s.evals()[1][17]
    one.close()
    s.close()


def test_c5_batch_16_draws_full_size():
    """A 16-draw slice of the C5 estimation batch at full size: draws 0..3 equal their fixtures inside the bigger batch
    (other grouping, other neighbours), and a draw without a fixture equals its single-draw solve."""
    m, gen = workloads.c5()
    P = gen(1024)[:16]
    s = solve_batch(m, P)
    st = s.status()[0]
    for i in range(4):
        check_draw(s, i, load('C5_full_draw%d' % i))
    j = int(np.nonzero(st == 0)[0][-1])
    one = solve_batch(m, P[j])
    assert one.status()[0][0] == 0
    assert np.array_equal(one.checksums(0), s.checksums(j)) and one.evals()[1][0] == s.evals()[1][j]
    one.close()
    s.close()


def test_batch_build_variant_equals_the_fixture_and_the_default_build():
    """bench.py solves large C2 batches with a build variant of its own (workloads.BATCH_BUILD_FLAGS: k_envelope compiled
    for fewer registers).  Same source, same results: draw 0 against the glibc fixture, 48 perturbed draws against the
    default build -- status, evaluation counts and the checksums of every cell."""
    g = load('C2')
    m, gen = workloads.c2(a0=0)
    flags = workloads.BATCH_BUILD_FLAGS['C2']
    P = np.concatenate([m.param_vector()[None], gen(48)])
    lib_v = build.build_model(m, extra_flags=flags)
    assert lib_v.path != build.build_model(m).path
    sv = runtime.Solver(lib_v, m.descriptor(), ndraw=len(P), keep_history=True)
    sv.set_params(P)
    sv.solve(raise_on_error=False)
    check_draw(sv, 0, g)
    sd = solve_batch(m, P)
    assert np.array_equal(sv.status()[0], sd.status()[0])
    assert np.array_equal(sv.evals()[1], sd.evals()[1])
    for d in range(len(P)):
        assert np.array_equal(sv.checksums(d), sd.checksums(d)), d
    sv.close()
    sd.close()


@pytest.mark.parametrize('name,wl', [('C4', 'C4'), ('C5_T60_n2000', 'C5')])
def test_batch_build_variants_of_the_stress_configs_equal_the_fixtures(name, wl):
    """C4 and C5 batches are solved with k_grid_lds compiled for eight waves per SIMD (workloads.BATCH_BUILD_FLAGS): the
    variant against the glibc fixture of the configuration, and against the default build on a few draws."""
    g = load(name)
    m, par = BIG[name][0]()
    flags = workloads.BATCH_BUILD_FLAGS[wl]
    lib_v = build.build_model(m, extra_flags=flags)
    assert lib_v.path != build.build_model(m).path
    P = np.atleast_2d(np.asarray(m.param_vector() if par is None else par, dtype=np.float64))
    gen = workloads.WORKLOADS[wl]()[1]
    P = np.concatenate([P[:1], gen(3)])
    sv = runtime.Solver(lib_v, m.descriptor(), ndraw=len(P), keep_history=True)
    sv.set_params(P)
    sv.solve(raise_on_error=False)
    check_draw(sv, 0, g)
    sd = solve_batch(m, P)
    assert np.array_equal(sv.status()[0], sd.status()[0]) and np.array_equal(sv.evals()[1], sd.evals()[1])
    for d in range(len(P)):
        assert np.array_equal(sv.checksums(d), sd.checksums(d)), d
    sv.close()
    sd.close()


def _solve_with_env(lib, m, P, env):
    """a kept-history solve with environment switches of the library set for the duration of the solve (they are read at solve time)"""
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        s = runtime.Solver(lib, m.descriptor(), ndraw=len(P), keep_history=True)
        s.set_params(P)
        s.solve(raise_on_error=False)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return s


@pytest.mark.parametrize('a0', [0.0, -5.0])
def test_c2_batch_build_on_the_path_the_bench_runs(a0):
    """The headline batch is solved by the C2 build variant on kernels a 49-draw batch never reaches: k_grid_lds_cv at 512 threads and
    64 VGPRs, and the throughput path of the envelope step (k_tp_* with its second tier for long lists).  1024 draws on exactly that
    path -- asserted from the path's own counters -- against (i) the default build with the envelope step on k_envelope alone and the
    general grid kernel's LDS form (another kernel for every phase): status, evaluation counts and the checksums of every cell of
    every draw; (ii) the oracle, cell by cell, on the draws that give the slow paths most to do (left-over cells, regenerated guess
    streams) and on failing draws, whose error texts must be the oracle's."""
    from egdst_amd import examples
    from oracle_harness import Oracle
    nd = 1024
    m = examples.retirement_sig(T=60, ngridm=1000, ngridmax=10000, nthrhmax=1000, ny=10, a0=a0)
    P = workloads.c2(a0=0)[1](nd)
    lib_v = build.build_model(m, extra_flags=workloads.BATCH_BUILD_FLAGS['C2'])
    sv = _solve_with_env(lib_v, m, P, {})
    tps = sv.tp_stats()
    done, left = int(tps[:, 0].sum()), int(tps[:, 1].sum())
    nt = m.descriptor()['T'] - m.descriptor()['t0'] + 1
    assert done > 0.9 * nd * (nt - 1) and done + left == nd * (nt - 1), (done, left)   # the throughput path did the cells
    sd = _solve_with_env(build.build_model(m), m, P, {'EGDST_ENV_TP': '0', 'EGDST_GRID_CV': '0'})
    assert int(sd.tp_stats().sum()) == 0                                               # ... and the other side did not use it
    st, ev = sv.status()[0], sv.evals()[1]
    assert np.array_equal(st, sd.status()[0]) and np.array_equal(sv.status()[1], sd.status()[1])
    assert np.array_equal(ev, sd.evals()[1])
    for d in range(nd):
        assert np.array_equal(sv.checksums(d), sd.checksums(d)), d
        lv, tv = sv.dims(d)
        ld, td = sd.dims(d)
        assert np.array_equal(lv, ld) and np.array_equal(tv, td), d
    sd.close()
    # the oracle on the interesting draws
    score = tps[:, 1].astype(np.int64) * 1000 + sv.regenerations().astype(np.int64)
    pick = list(np.argsort(-score, kind='stable')[:24]) + list(np.nonzero(st != 0)[0][:16])
    orc = Oracle(m)
    for d in dict.fromkeys(int(x) for x in pick):
        r = orc.solve(P[d])
        assert (r.rc == 0) == (st[d] == 0), d
        if r.rc:
            assert lib_v.lib.egdst_strerror(int(st[d])).decode().strip() == r.err.strip(), d
        else:
            assert ev[d] == r.nevals, d
        ln, th = sv.dims(d)
        assert np.array_equal(ln, r.len) and np.array_equal(th, r.thlen), d
        assert np.array_equal(sv.checksums(d), cell_sums(r)), d
    sv.close()
