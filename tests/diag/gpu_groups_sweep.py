"""Diagnostic: step time of a C2 batch against the number of draw groups and of hardware queues (one child process per setting:
GPU_MAX_HW_QUEUES is read when the HIP runtime starts).   python tests/diag/gpu_groups_sweep.py [a0=-5] [ndraw=4096] [groups,hwq ...]"""
import os, subprocess, sys
a0 = sys.argv[1] if len(sys.argv) > 1 else '-5'
nd = sys.argv[2] if len(sys.argv) > 2 else '4096'
combos = [tuple(x.split(',')) for x in sys.argv[3:]] or [('16', '24'), ('20', '24'), ('24', '32'), ('32', '40'), ('32', '64'), ('12', '24')]
child = r'''
import os, sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from egdst_amd import build, runtime, workloads
a0, nd = float(sys.argv[1]), int(sys.argv[2])
m, gen = workloads.c2(a0=a0)
lib = build.build_model(m, extra_flags=workloads.BATCH_BUILD_FLAGS['C2'])
s = runtime.Solver(lib, m.descriptor(), ndraw=nd, keep_history=False)
P = gen(nd)
s.set_params(P); s.solve(raise_on_error=False); s.solve(raise_on_error=False)
ts = []
for k in range(4):
    s.set_params(P * (1 + 0.005 * (2 * np.random.default_rng(k).random(P.shape) - 1)))
    t = time.perf_counter(); s.solve(raise_on_error=False); ts.append((time.perf_counter() - t) * 1e3)
print('groups %s hwq %s: %s ms  schedule %s' % (os.environ.get('EGDST_GROUPS'), os.environ.get('GPU_MAX_HW_QUEUES'), ['%.1f' % t for t in ts], s.schedule()), flush=True)
'''
for g, q in combos:
    env = dict(os.environ, EGDST_GROUPS=g, GPU_MAX_HW_QUEUES=q)
    r = subprocess.run([sys.executable, '-c', child, a0, nd], env=env, capture_output=True, text=True, timeout=300)
    print(r.stdout.strip() or r.stderr[-500:], flush=True)
