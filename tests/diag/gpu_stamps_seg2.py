"""Diagnostic (-DEGDST_STAMPS -DEGDST_STAMPS2): inside run_walk -- planning the cuts, the segments (wall), check + gather."""
import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from egdst_amd import build, runtime, workloads
extra = [a for a in sys.argv[1:] if a.startswith('-')]
only = [a for a in sys.argv[1:] if not a.startswith('-')]
for wl, tag in ([(w, '_stamps2_' + w) for w in only] or (('C2', '_stamps2'), ('C3', '_stamps2_c3'))[:1 if extra else 2]):
    m = workloads.WORKLOADS[wl]()[0]
    lib = build.build_model(m, build_dir='egdst_amd/_models/' + tag + ''.join(extra).replace('-D', '_').replace('=', ''), extra_flags=['-DEGDST_STAMPS', '-DEGDST_STAMPS2'] + extra)
    import os
    nd = int(os.environ.get('ND', '1'))  # (ND > 1: a batch of identical draws; the figures are of draw 0)
    s = runtime.Solver(lib, m.descriptor(), ndraw=nd, keep_history=False)
    s.set_params(np.tile(m.param_vector()[None], (nd, 1))); s.solve()
    b0 = s.debug(0).view(np.uint64).copy()
    s.solve()
    raw = s.debug(0).view(np.uint64) - b0
    one = int(raw[0])
    print(wl, 'walks by one wave: %d (%d of them not short lists), %.2f ms' % (
        (one >> 40) & 4095, one >> 52, (one & ((1 << 40) - 1)) * 1e-5))
    if '-DEGDST_STAMPS2_WHY' in extra:
        print(wl, 'one-wave walks by reason (kink log, no scratch, too many pieces, short list, scratch too small, other):', [(int(raw[1]) >> (10 * k)) & 1023 for k in range(6)])
    d = raw.astype(np.float64) * 1e-5
    print(extra, wl, 'walk phase %.2f ms = plan %.2f + segments %.2f + check/gather %.2f | sum of segment times %.2f ms | sort %.2f stop+compact %.2f | walks %s' % (
        d[6], d[3], d[4], d[7], d[1], d[5], d[2], s.walk_stats()[0].tolist()), flush=True)
