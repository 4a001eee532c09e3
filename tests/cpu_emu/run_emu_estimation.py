"""Harness build: the on-device estimation step (batch simulation with generated uniforms, moments, objective) of a few
C2-like draws against the oracle fed with the host replay of the same uniforms."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tests', 'cpu_emu'))
import numpy as np
import build_emu
from egdst_amd import build, codegen, runtime, workloads
from oracle_harness import Oracle
import estimation_case

if __name__ == '__main__':
    san = os.environ.get('EMU_SANITIZE', 'address')
    m, gen = workloads.c2(a0=0, ngridm=60, T=10, ny=5)
    P = gen(5)
    text = codegen.generate_modelspec(m)
    d = os.path.join(build.MODELS_DIR, build.model_tag(m, text))
    os.makedirs(d, exist_ok=True)
    open(os.path.join(d, 'modelspec.h'), 'w').write(text)
    lib = runtime.ModelLibrary(build_emu.build(d, {'0': False}.get(san, san), 1, False, 1))
    s = runtime.Solver(lib, m.descriptor(), ndraw=len(P), keep_history=True)
    s.set_params(P)
    s.solve(raise_on_error=False)
    rng = np.random.default_rng(5)
    nsim = 40
    init = np.column_stack([np.ones(nsim), rng.uniform(m.a0 - 0.5, m.mmax + 0.5, nsim)])
    bad = []
    for rndtype in (0, 1):
        bad += estimation_case.check(s, Oracle(m), P, init, seed=12345 + rndtype, rndtype=rndtype, lib=lib)
    print('estimation problems: %d %s' % (len(bad), bad[:3]))
