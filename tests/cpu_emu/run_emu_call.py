"""Sanitizer build: the model-function accessor (egdst_call) against the oracle, every switch and the bad-argument rules."""
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
    orc = Oracle(m); ref = orc.solve()
    bad = 0
    for sw, args in call_cases(m, s.nt, lib.info.nst, lib.info.nd):
        a, b = s.call(sw, args), orc.call(ref, sw, args)
        same = np.array_equal(a, b, equal_nan=True)
        bad += not same
        print('sw', sw, 'rows', len(a), 'nan', int(np.isnan(a).sum()), 'identical', same)
    print('call mismatches:', bad)
