import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from egdst_amd import build, runtime, workloads, examples
for m in (examples.retirement2(T=8, ngridm=60), workloads.WORKLOADS['C2']()[0], examples.occ3(ngridm=500, ngridmax=5000, nthrhmax=500, ny=5)):
  for rk in (4,):  # (the default)
      lib = build.build_model(m, build_dir='egdst_amd/_models/_rk%d_%s' % (rk, m.label[:6] + str(m.ngridm)), extra_flags=['-DENV_RK=%d' % rk, '-DEGDST_RANKCHK'])
      s = runtime.Solver(lib, m.descriptor(), ndraw=1, keep_history=False)
      s.set_params(m.param_vector()[None]); rc = s.solve(raise_on_error=False)
      print(m.label, 'RK', rk, 'rc', rc, 'dbg', s.debug(0).tolist()[:8], flush=True)
