"""Diagnostic: the generic step of the envelope walk with the functions' state on the lanes (default) against env_step_wave
   (-DENV_LANE_STEP=0): single solve and a batch, same bits.   python tests/diag/gpu_step_ab.py [NDRAW]"""
import os, sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from egdst_amd import build, runtime, workloads
nd = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
AB = len(sys.argv) > 2 and sys.argv[2] == 'ab'
m, gen = workloads.c2(a0=0)
P = gen(nd)
ref = {}
for name, extra, bdir in [('lanes', [], None)] + ([('wave', ['-DENV_LANE_STEP=0'], 'egdst_amd/_models/_c2_oldstep')] if AB else []):
    lib = build.build_model(m, build_dir=bdir, extra_flags=extra)
    s = runtime.Solver(lib, m.descriptor(), ndraw=1, keep_history=True)
    s.set_params(m.param_vector()[None]); s.solve()
    ts = []
    for _ in range(5):
        t = time.perf_counter(); s.solve(); ts.append((time.perf_counter() - t) * 1e3)
    ck = s.checksums(0)
    print('%s single solve ms %s evals %d same=%s' % (name, ['%.2f' % t for t in ts], s.evals()[0], None if 'one' not in ref else bool(np.array_equal(ck, ref['one']))), flush=True)
    ref.setdefault('one', ck)
    s.close()
for name, extra, bdir in [('lanes', [], None)] + ([('wave', ['-DENV_LANE_STEP=0'], 'egdst_amd/_models/_c2_oldstep_b')] if AB else []):
    lib = build.build_model(m, build_dir=bdir, extra_flags=workloads.BATCH_BUILD_FLAGS['C2'] + extra + os.environ.get('EGDST_DIAG_FLAGS', '').split())
    s = runtime.Solver(lib, m.descriptor(), ndraw=nd, keep_history=False)
    if os.environ.get('EGDST_DIAG_GROUPS'): s.set_groups(int(os.environ['EGDST_DIAG_GROUPS']))
    s.set_params(P); s.solve(raise_on_error=False)
    ts = []
    for _ in range(3):
        t = time.perf_counter(); s.solve(raise_on_error=False); ts.append((time.perf_counter() - t) * 1e3)
    r = (s.status()[0].copy(), s.evals()[1].copy(), s.objective().copy())
    same = None
    if 'b' in ref:
        same = bool(np.array_equal(r[0], ref['b'][0]) and np.array_equal(r[1], ref['b'][1]) and np.array_equal(r[2], ref['b'][2], equal_nan=True))
    ref.setdefault('b', r)
    s.set_profile(True); s.solve(raise_on_error=False)
    print('%s batch x %d ms %s failed %d tp %s kernel ms %s same=%s' % (name, nd, ['%.1f' % t for t in ts], int((r[0] != 0).sum()), s.tp_stats().sum(axis=0).tolist(), np.round(s.profile()[0], 1).tolist(), same), flush=True)
    s.close()
