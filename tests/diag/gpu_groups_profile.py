import sys, os
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from egdst_amd import build, runtime, workloads
wl, nd = sys.argv[1], int(sys.argv[2])
m, gen = workloads.WORKLOADS[wl]()
flags = workloads.BATCH_BUILD_FLAGS.get(wl, [])
lib = build.build_model(m, extra_flags=flags)
s = runtime.Solver(lib, m.descriptor(), ndraw=nd, keep_history=False)
s.set_params(gen(nd)); s.solve(raise_on_error=False)
print('solved', s.schedule())
os.environ['EGDST_DEBUG_SYNC'] = sys.argv[3] if len(sys.argv) > 3 else ''
if not os.environ['EGDST_DEBUG_SYNC']: del os.environ['EGDST_DEBUG_SYNC']
s.set_profile(True); s.solve(raise_on_error=False); print('profiled', np.round(s.profile()[0], 1))
s.set_groups(1); print('groups 1', s.schedule())
s.solve(raise_on_error=False); print('serial', np.round(s.profile()[0], 1))
