"""Sanitizer build: a handle that never solved is given another handle's exported cells (egdst_set_cell_M / _D and
egdst_set_solution) and must simulate, answer egdst_call and export exactly as the handle that solved them."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tests', 'cpu_emu'))
import numpy as np
import build_emu
from egdst_amd import build, codegen, examples, runtime
from oracle_harness import Oracle
from call_cases import call_cases

if __name__ == '__main__':
    name = sys.argv[1]
    kw = eval('dict(%s)' % (sys.argv[2] if len(sys.argv) > 2 else ''))
    m = examples.REGISTRY[name](**kw)
    text = codegen.generate_modelspec(m)
    d = os.path.join(build.MODELS_DIR, build.model_tag(m, text))
    os.makedirs(d, exist_ok=True)
    open(os.path.join(d, 'modelspec.h'), 'w').write(text)
    san = os.environ.get('EMU_SANITIZE', 'address')
    lib = runtime.ModelLibrary(build_emu.build(d, {'0': False}.get(san, san), 1, False, 1))
    s = runtime.Solver(lib, m.descriptor(), ndraw=1, keep_history=True)
    s.set_params(m.param_vector()); s.solve()
    sol = s.solution(0)
    orc = Oracle(m); ref = orc.solve()
    rng = np.random.default_rng(5)
    nsim = 12
    feas = [ist for ist in range(lib.info.nst) if sol.len[0, ist] > 0]
    init = np.column_stack([rng.choice(feas, nsim) + 1.0, rng.uniform(m.a0, m.mmax, nsim)])
    rs = rng.random(4 * s.nt * nsim)
    bad = 0
    for how in ('cells', 'bulk'):
        f = runtime.Solver(lib, m.descriptor(), ndraw=1, keep_history=True)
        f.set_params(m.param_vector())
        if how == 'cells':
            f.set_cells(*sol.cells())
        else:
            f.set_solution(sol)
        same_sim = np.array_equal(f.simulate(init, rs), orc.sim(ref, init, rs), equal_nan=True)
        same_call = all(np.array_equal(f.call(sw, a), orc.call(ref, sw, a), equal_nan=True) for sw, a in call_cases(m, s.nt, lib.info.nst, lib.info.nd))
        same_sum = np.array_equal(f.checksums(0), s.checksums(0))
        g = f.solution(0)
        same_tab = all(np.array_equal(getattr(g, k), getattr(sol, k)) for k in ('M', 'C', 'V', 'D', 'TH', 'len', 'thlen'))
        print(how, 'sim', same_sim, 'call', same_call, 'checksums', same_sum, 'tables', same_tab)
        bad += not (same_sim and same_call and same_sum and same_tab)
        f.close()
    print('import mismatches:', bad)
