"""Device code under AddressSanitizer/UBSan on the CPU (one-thread workgroups, tests/cpu_emu) -- sanitizers are
not available on the GPU pool.  The same kernels, compiled by g++, must be memory-clean and reproduce the
oracle bit for bit, including batched draws with ping-pong tables."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _asan():
    r = subprocess.run(['gcc', '-print-file-name=libasan.so'], capture_output=True, text=True)
    p = r.stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.skipif(_asan() is None, reason='libasan not found')
@pytest.mark.parametrize('args', [['retirement2', 'T=8, ngridm=60'], ['retirement8', 'T=5, ngridm=30, ny=3'],
                                  ['occ3', 'T=6, ngridm=30, ngridmax=100']])
def test_device_code_is_asan_clean_and_bit_exact(args):
    env = dict(os.environ, LD_PRELOAD=_asan(), ASAN_OPTIONS='detect_leaks=0', EMU_SANITIZE='address')
    r = subprocess.run([sys.executable, os.path.join(HERE, 'cpu_emu', 'run_emu.py')] + args, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert 'ok=True' in r.stdout and 'max_rel=0.00e+00' in r.stdout, r.stdout + r.stderr[-2000:]
    assert 'ERROR: AddressSanitizer' not in r.stderr and 'runtime error' not in r.stderr, r.stderr[-3000:]


@pytest.mark.skipif(_asan() is None, reason='libasan not found')
@pytest.mark.parametrize('args,lcap', [(['retirement2', 'T=8, ngridm=60'], '16'), (['retirement8', 'T=5, ngridm=30, ny=3'], '64'),
                                       (['occ3', 'T=6, ngridm=30, ngridmax=100'], '64')])
def test_global_memory_streams_are_asan_clean_and_bit_exact(args, lcap):
    """EGDST_LCAP: so little LDS that every stream sorts and walks in global memory -- the merge path of two lists, the rank
    merge with its keys (or, beyond their capacity, a sampled index of them) in LDS, multi-lane waves, segmented walks."""
    env = dict(os.environ, LD_PRELOAD=_asan(), ASAN_OPTIONS='detect_leaks=0', EMU_SANITIZE='address', EGDST_LCAP=lcap,
               EMU_WAVE='4', EMU_ENV_BS='8')
    r = subprocess.run([sys.executable, os.path.join(HERE, 'cpu_emu', 'run_emu.py')] + args, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert 'ok=True' in r.stdout and 'max_rel=0.00e+00' in r.stdout, r.stdout + r.stderr[-2000:]
    assert 'ERROR: AddressSanitizer' not in r.stderr and 'runtime error' not in r.stderr, r.stderr[-3000:]


@pytest.mark.skipif(_asan() is None, reason='libasan not found')
def test_the_round1_fault_case_is_clean_with_every_array_in_its_own_allocation():
    """Round 1 saw a GPU memory fault in k_envelope on this very input (8-state model, T=12, ngridm=150, ny=5: 14 492
    rows, 401 608 evaluations) in builds whose walk was NOT inlined.  Under the harness every device array is its own
    heap block and the LDS regions are separated by poisoned gaps, so an overrun from one array into its neighbour --
    invisible while everything was carved from one pool -- would trap here.  It does not."""
    env = dict(os.environ, LD_PRELOAD=_asan(), ASAN_OPTIONS='detect_leaks=0', EMU_SANITIZE='address')
    r = subprocess.run([sys.executable, os.path.join(HERE, 'cpu_emu', 'run_emu.py'), 'retirement8', 'T=12, ngridm=150, ny=5'],
                       env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    assert 'ok=True' in r.stdout and 'rows=14492/14492 evals=401608/401608 max_rel=0.00e+00' in r.stdout, r.stdout
    assert 'ERROR: AddressSanitizer' not in r.stderr and 'runtime error' not in r.stderr, r.stderr[-3000:]


@pytest.mark.skipif(_asan() is None, reason='libasan not found')
def test_batched_draws_and_pingpong_tables():
    env = dict(os.environ, LD_PRELOAD=_asan(), ASAN_OPTIONS='detect_leaks=0', EMU_SANITIZE='address')
    r = subprocess.run([sys.executable, os.path.join(HERE, 'cpu_emu', 'run_emu_batch.py'), '60', '12', '4'], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('keep_history')]
    assert len(lines) == 8
    for ln in lines:
        st, where, ev, ev_ref, same = eval(ln.split(' ', 2)[2])
        assert st == 0 and ev == ev_ref and same in (True, None), ln
    assert 'ERROR: AddressSanitizer' not in r.stderr


@pytest.mark.skipif(_asan() is None, reason='libasan not found')
def test_compact_capacity_overflow_is_redone_exactly():
    """rows_cap below what the model needs: the device reports EGDST_E_CAPACITY without touching memory past the
    compact arrays (ASan), the host solves the draw again on an exact handle, the result equals the oracle's."""
    env = dict(os.environ, LD_PRELOAD=_asan(), ASAN_OPTIONS='detect_leaks=0', EMU_SANITIZE='address', EMU_ROWS_CAP='64')
    r = subprocess.run([sys.executable, os.path.join(HERE, 'cpu_emu', 'run_emu.py'), 'retirement2', 'T=8, ngridm=60'],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert 'ok=True' in r.stdout and 'max_rel=0.00e+00' in r.stdout and 'capacity_retries 1' in r.stdout, r.stdout
    assert 'ERROR: AddressSanitizer' not in r.stderr and 'runtime error' not in r.stderr, r.stderr[-3000:]


@pytest.mark.skipif(_asan() is None, reason='libasan not found')
@pytest.mark.parametrize('args', [['retirement2', 'T=8, ngridm=60'], ['retirement8', 'T=5, ngridm=30, ny=3']])
def test_solution_import_serves_simulator_and_accessor(args):
    """egdst_set_cell_M/_D and egdst_set_solution (the inverse of the export: what the simulator and accessor gateways of
    the reference read from the model object, egdst_simulator.c:61-68) on a handle that never solved: simulated paths
    and accessor values equal the oracle's, tables and checksums equal the solving handle's, ASan clean."""
    env = dict(os.environ, LD_PRELOAD=_asan(), ASAN_OPTIONS='detect_leaks=0', EMU_SANITIZE='address')
    r = subprocess.run([sys.executable, os.path.join(HERE, 'cpu_emu', 'run_emu_import.py')] + args, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert 'import mismatches: 0' in r.stdout, r.stdout + r.stderr[-2000:]
    assert 'ERROR: AddressSanitizer' not in r.stderr and 'runtime error' not in r.stderr, r.stderr[-3000:]


@pytest.mark.skipif(_asan() is None, reason='libasan not found')
@pytest.mark.parametrize('args,env', [
    (['retirement2', 'T=8, ngridm=200, nthrhmax=200'], {}),                         # walks cut into segments (four walking waves)
    (['occ3', 'T=6, ngridm=30, ngridmax=100'], {'EGDST_TP_SORT_LKCAP': '16'}),       # sampled key index, permutation through global memory
    (['retirement8', 'T=5, ngridm=30, ny=3'], {'EGDST_TP_LKCAP': '40', 'EGDST_TP_BIG': '0'}),   # streams beyond the walk's LDS: left to k_envelope
    (['retirement8', 'T=5, ngridm=30, ny=3'], {'EGDST_TP_LKCAP': '40'}),             # ... done by the second tier of stage 1 (k_tp_big)
    (['retirement8', 'T=5, ngridm=30, ny=3'], {'EGDST_TP_LKCAP': '40', 'EGDST_TP_BIGCAP': '52'}),  # ... some by the second tier, the longest by k_envelope
])
def test_throughput_path_of_the_envelope_step(args, env):
    """EGDST_ENV_TP=1: the envelope step as the lean kernels of big batches (k_tp_prep / k_tp_sort / k_tp_walk) with multi-lane
    waves and multi-wave workgroups -- bit-exact against the oracle, ASan clean, with the LDS regions of every phase separated
    by poisoned gaps; cells the path does not take go through k_envelope's list pass."""
    e = dict(os.environ, LD_PRELOAD=_asan(), ASAN_OPTIONS='detect_leaks=0', EMU_SANITIZE='address', EGDST_ENV_TP='1', EMU_WAVE='4',
             EMU_ENV_BS='16', **env)
    r = subprocess.run([sys.executable, os.path.join(HERE, 'cpu_emu', 'run_emu.py')] + args, env=e, capture_output=True, text=True,
                       timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    assert 'ok=True' in r.stdout and 'max_rel=0.00e+00' in r.stdout, r.stdout + r.stderr[-2000:]
    done, left = eval(r.stdout.split('tp done/left')[1].strip())
    assert done + left > 0 and (done > 0 or env.get('EGDST_TP_BIG') == '0'), r.stdout
    if env.get('EGDST_TP_LKCAP') and (env.get('EGDST_TP_BIG') == '0' or env.get('EGDST_TP_BIGCAP')):
        assert left > 0
    if env.get('EGDST_TP_LKCAP') and env.get('EGDST_TP_BIG') != '0':
        assert done > 0   # (with a 40-point budget the regular launch completes next to nothing: these are the second tier's)
    assert 'ERROR: AddressSanitizer' not in r.stderr and 'runtime error' not in r.stderr, r.stderr[-3000:]


@pytest.mark.skipif(_asan() is None, reason='libasan not found')
@pytest.mark.parametrize('args,lds', [(['retirement2', 'T=8, ngridm=60'], ''), (['occ3', 'T=6, ngridm=30, ngridmax=100'], '24')])
def test_grid_kernel_with_the_branch_free_searches(args, lds):
    """EGDST_GRID_WIDE=0: k_grid_lds on a single draw -- the branch-free bisection over whole columns in LDS (eg_bracket_sorted) and,
    with EGDST_GRID_LDS=24, over the sampled index and the window of the global column (eg_bracket_sampled): every read in bounds,
    results the oracle's."""
    e = dict(os.environ, LD_PRELOAD=_asan(), ASAN_OPTIONS='detect_leaks=0', EMU_SANITIZE='address', EGDST_GRID_WIDE='0')
    if lds:
        e['EGDST_GRID_LDS'] = lds
    r = subprocess.run([sys.executable, os.path.join(HERE, 'cpu_emu', 'run_emu.py')] + args, env=e, capture_output=True, text=True,
                       timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    assert 'ok=True' in r.stdout and 'max_rel=0.00e+00' in r.stdout, r.stdout + r.stderr[-2000:]
    assert 'ERROR: AddressSanitizer' not in r.stderr and 'runtime error' not in r.stderr, r.stderr[-3000:]


def test_one_row_tables_read_zeros_past_their_end_in_pingpong_mode():
    """Full-size C2, a draw on which the reference algorithm degenerates (one-row table at it=27): with ping-pong
    tables the failure must be the oracle's, not a success built on stale rows.  (Since round 4 that failure is the reference's
    own: valuefunc's linter_extrap refuses a table with one row beside the a0 row, egdst_lib.c:183 -- error 10 at it=26; before,
    oracle and device both went on with the one-row table and stopped in the same period with error 15.)"""
    r = subprocess.run([sys.executable, os.path.join(HERE, 'cpu_emu', 'run_emu_draws.py'), '67'],
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('keep_history')]
    assert len(lines) == 2, r.stdout
    for ln in lines:
        st, where, ev, ref_rc = eval(ln.split(' ', 3)[3])
        assert st == 10 and where == (26, 0) and ref_rc != 0, ln


def test_plain_sequential_walk_build_agrees():
    """EGDST_SEQ_WALK builds the envelope walk as the literal one-point-at-a-time restatement of the reference loop
    (env_step / env_crossing) instead of the wave-cooperative one; both must reproduce the oracle bit for bit."""
    env = dict(os.environ, EMU_SANITIZE='0', EMU_SEQ_WALK='1')
    for args in (['retirement2', 'T=8, ngridm=60'], ['occ3', 'T=6, ngridm=30, ngridmax=100']):
        r = subprocess.run([sys.executable, os.path.join(HERE, 'cpu_emu', 'run_emu.py')] + args, env=env,
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        assert 'ok=True' in r.stdout and 'max_rel=0.00e+00' in r.stdout, r.stdout + r.stderr[-2000:]


@pytest.mark.skipif(_asan() is None, reason='libasan not found')
@pytest.mark.parametrize('mode', ['fast', 'defer_all', 'defer_late', 'off'])
def test_single_choice_envelope_kernels(mode):
    """Single-choice models compact a cell with many workgroups (k_env1: tiles with look-back); irregular cells are handed to
    k_envelope.  The fast path, the hand-over (forced for every cell, before and after the tiles have written rows) and the general
    kernel alone must all equal the oracle."""
    env = dict(os.environ, LD_PRELOAD=_asan(), ASAN_OPTIONS='detect_leaks=0', EMU_SANITIZE='address')
    if mode == 'defer_all':
        env['EGDST_E1_DEFER_ALL'] = '1'
    if mode == 'defer_late':   # every cell handed over AFTER its tiles wrote rows into the table: the marks must cover them
        env['EGDST_E1_DEFER_ALL'] = '2'
    if mode == 'off':
        env['EGDST_NO_ENV1'] = '1'
    r = subprocess.run([sys.executable, os.path.join(HERE, 'cpu_emu', 'run_emu.py'), 'deaton2', ''], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert 'ok=True' in r.stdout and 'max_rel=0.00e+00' in r.stdout and 'evals=24474/24474' in r.stdout, r.stdout + r.stderr[-2000:]
    assert 'ERROR: AddressSanitizer' not in r.stderr and 'runtime error' not in r.stderr, r.stderr[-3000:]


@pytest.mark.skipif(_asan() is None, reason='libasan not found')
@pytest.mark.parametrize('args', [['retirement2', 'T=8, ngridm=60'], ['retirement8', 'T=5, ngridm=30, ny=3']])
def test_model_function_accessor(args):
    """egdst_call (egdstmodel.call): every switch, vector input, out-of-domain values and the gateway's bad-index rules
    -- device code under ASan equals the oracle's restatement of egdst_call.c bit for bit."""
    env = dict(os.environ, LD_PRELOAD=_asan(), ASAN_OPTIONS='detect_leaks=0', EMU_SANITIZE='address')
    r = subprocess.run([sys.executable, os.path.join(HERE, 'cpu_emu', 'run_emu_call.py')] + args, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert 'call mismatches: 0' in r.stdout, r.stdout + r.stderr[-2000:]
    assert 'ERROR: AddressSanitizer' not in r.stderr and 'runtime error' not in r.stderr, r.stderr[-3000:]


@pytest.mark.skipif(_asan() is None, reason='libasan not found')
def test_simulated_moments_kernel():
    """k_moments under ASan: counts and means of the simulated columns equal numpy's over the same simulated paths."""
    code = r'''
import os, sys
ROOT = %r
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tests', 'cpu_emu'))
import warnings
import numpy as np
import build_emu
from egdst_amd import build, codegen, examples, runtime
m = examples.retirement2(T=8, ngridm=60)
text = codegen.generate_modelspec(m)
d = os.path.join(build.MODELS_DIR, build.model_tag(m, text))
os.makedirs(d, exist_ok=True); open(os.path.join(d, 'modelspec.h'), 'w').write(text)
lib = runtime.ModelLibrary(build_emu.build(d, 'address', 1, False, 1))
s = runtime.Solver(lib, m.descriptor(), ndraw=1, keep_history=True)
s.set_params(m.param_vector()); s.solve()
rng = np.random.default_rng(1)
nsim = 200
init = np.column_stack([np.ones(nsim), rng.uniform(-6, 11, nsim)])
rs = rng.random(4 * s.nt * nsim)
means, counts = s.simulate_moments(init, rs)
sims = s.simulate(init, rs)
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    ref = np.nanmean(sims, axis=0)
ok = np.array_equal(counts, (~np.isnan(sims)).sum(axis=0)) and np.array_equal(np.isnan(means), np.isnan(ref))
fin = np.isfinite(ref)
ok = ok and np.all(np.abs(means[fin] - ref[fin]) <= 1e-13 * np.maximum(1, np.abs(ref[fin])))
print('moments ok', ok)
''' % os.path.dirname(HERE)
    env = dict(os.environ, LD_PRELOAD=_asan(), ASAN_OPTIONS='detect_leaks=0')
    r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert 'moments ok True' in r.stdout, r.stdout + r.stderr[-2000:]
    assert 'ERROR: AddressSanitizer' not in r.stderr, r.stderr[-3000:]


@pytest.mark.skipif(_asan() is None, reason='libasan not found')
def test_estimation_step_kernels():
    """egdst_simulate_batch_moments under ASan: batch simulation with device-generated uniforms (own and shared streams),
    per-draw moments and the moment objective equal the oracle's on the host replay of the same uniforms."""
    env = dict(os.environ, LD_PRELOAD=_asan(), ASAN_OPTIONS='detect_leaks=0', EMU_SANITIZE='address')
    r = subprocess.run([sys.executable, os.path.join(HERE, 'cpu_emu', 'run_emu_estimation.py')], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert 'estimation problems: 0' in r.stdout, r.stdout + r.stderr[-2000:]
    assert 'ERROR: AddressSanitizer' not in r.stderr and 'runtime error' not in r.stderr, r.stderr[-3000:]


@pytest.mark.skipif(_asan() is None, reason='libasan not found')
def test_grid_kernel_with_whole_tables_in_lds():
    """EGDST_GRID_CV=1: k_grid_lds_cv -- M, C and V of the next-period table staged in LDS, no global read in the loop over the shock
    nodes -- bit-exact against the oracle, ASan clean."""
    e = dict(os.environ, LD_PRELOAD=_asan(), ASAN_OPTIONS='detect_leaks=0', EMU_SANITIZE='address', EGDST_GRID_CV='1', EGDST_GRID_WIDE='0')
    r = subprocess.run([sys.executable, os.path.join(HERE, 'cpu_emu', 'run_emu.py'), 'retirement2', 'T=8, ngridm=60'], env=e,
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    assert 'ok=True' in r.stdout and 'max_rel=0.00e+00' in r.stdout, r.stdout + r.stderr[-2000:]
    assert 'ERROR: AddressSanitizer' not in r.stderr and 'runtime error' not in r.stderr, r.stderr[-3000:]
