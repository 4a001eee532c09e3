"""BASELINE.json configs restated concretely + parameter-draw generators.

C2 is pinned by SURVEY.md §8(d) on the forms and values of model_retirement2.m, credit limit a0 = -5 included
(egdst_examples/model_retirement2.m:36-41): that is `c2()`, the configuration bench.py's headline runs.  The default draw solves
(1 180 889 evaluations, the survey's probe number); about 12 % of the 4096 perturbed draws of the batch do not -- the reference
algorithm fails on them (stage 0 of adraw finds no savings with positive consumption; a guess stream that ends in the resend fixed
point leaves a one-row table and the next period stops with "At least two points are required for interpolation!") and so do the
oracle and the device, with the same texts.  `c2(a0=0)` is the easier parameterisation rounds 1-3 reported (1.5 % failing draws);
it stays as a leg of the bench and as the configuration of most full-size fixtures.
"""
from __future__ import annotations

import numpy as np

from . import examples


def _draws(seed, ndraw, lo, hi):
    rng = np.random.default_rng(seed)
    lo, hi = np.asarray(lo, float), np.asarray(hi, float)
    return lo + (hi - lo) * rng.random((ndraw, len(lo)))


def c1():
    return examples.deaton_sig(a0=0, mmax=50, t0=1, T=20, ngridm=100, ny=5), None


def c2(ngridm=1000, T=60, ny=10, a0=-5.0):
    """retirement2 forms, 2 discrete choices, T=60, 1000 grid points, 10 nodes, a0=-5 (BASELINE configs[1])."""
    m = examples.retirement_sig(T=T, ngridm=ngridm, ngridmax=10 * ngridm, nthrhmax=ngridm, ny=ny, a0=a0)

    def draws(ndraw, seed=20241):
        # params: duw, interest, wage, sig  -- draws over (duw, wage, sigma), interest fixed (SURVEY §8d C5 ranges)
        d = _draws(seed, ndraw, [0.3, 0.9, 0.15], [0.8, 1.3, 0.35])
        out = np.tile(m.param_vector(), (ndraw, 1))
        out[:, 0], out[:, 2], out[:, 3] = d[:, 0], d[:, 1], d[:, 2]
        return out
    return m, draws


def c3():
    return examples.occ3(ngridm=4000, ngridmax=40000, nthrhmax=4000, ny=15), None


def c4(ngridm=65536, T=80, ny=21):
    """Deaton stress: draws over (interest, income, sigma), rng(20240)."""
    m = examples.deaton_sig(a0=0, mmax=50, t0=1, T=T, ngridm=ngridm, ngridmax=2 * ngridm, nthrhmax=10, ny=ny)

    def draws(ndraw, seed=20240):
        return _draws(seed, ndraw, [0.005, 1.0, 0.5], [0.03, 1.5, 0.9])  # interest, income, sig
    return m, draws


def c5(ngridm=32768, T=100, ny=15):
    m = examples.retirement8(T=T, ngridm=ngridm, ny=ny)
    m.ngridmax = 4 * ngridm

    def draws(ndraw, seed=20241):
        d = _draws(seed, ndraw, [0.3, 0.9, 0.15], [0.8, 1.3, 0.35])  # duw, wage, sigma
        out = np.tile(m.param_vector(), (ndraw, 1))  # duw, interest, wage, sig
        out[:, 0], out[:, 2], out[:, 3] = d[:, 0], d[:, 1], d[:, 2]
        return out
    return m, draws


WORKLOADS = {'C1': c1, 'C2': c2, 'C3': c3, 'C4': c4, 'C5': c5}

# Build variants for large batches (same source, other register budgets; `bench.py` uses them for handles of at least
# BATCH_BUILD_MIN_DRAWS[workload] draws, the one-draw latency legs use the default build; tests/test_gpu_big.py holds them to
# the full-size fixtures).  Measured on MI355X:
#  * -DENV_MINW=3: k_envelope is compiled for 256 VGPRs by default, and a 512-thread workgroup of it then owns the whole
#    register file of its CU: no wave of the grid kernels runs beside it, although its waves wait ~75% of their cycles (PMC).
#    168 VGPRs (more spills) leave room for two grid-kernel waves per SIMD: C2 x 4096 draws 243 -> 230 ms per batch, but one
#    C2 solve 9.1 -> 9.6 ms, one C3 solve 21.0 -> 23.4 ms, C3 x 64 42.5 -> 46.2 ms, C5 x 128 unchanged.
#  * -DGRID_MINW=8: k_grid_lds for 64 VGPRs (44-84 B of spills) instead of 77-92, eight waves per SIMD instead of five or
#    six, for the long tables of C4 and C5 whose bracket searches finish in global memory: C4 x 32 draws 63.6 -> 58.8 ms,
#    C5 x 128 draws 4.32 -> 3.76 s; C2 unchanged, C3 x 64 (pow-heavy) 42.5 -> 48.0 ms.
#  * -DGRID_BS=1024: workgroups of 1024 grid points share one staged index (C4 x 32: 58.8 -> 54.3 ms; 512: 55.5; C5 x 128 with
#    512: 3.84 s against 3.78, not used there).
#  * C2: -DGRID_BS=512 -DEG_GRID_CV_DEFAULT=1: the grid kernel stages the whole next-period table (M, C and V: 37.5 KB) for 512 points
#    instead of the M column for 256, and reads nothing from global memory in its loop over the shock nodes (k_grid_lds_cv):
#    C2 x 4096 164 -> 161 ms (with 256 points the extra LDS costs occupancy: 168 ms).
#    With it -DGRID_MINW=8 pays for C2 as well (64 VGPRs, four 8-wave workgroups per CU): grid kernel 49.9 -> 45.1 ms, solve 162 -> 158 ms.
#    (Round 4: 1024-point workgroups, one per stream, which then list the stream for k_fixup themselves instead of a k_fixup_scan
#    launch: 169.8 ms against 167.1 for C2 a0=-5 x 4096 -- the scan launch costs nothing a chain notices; not kept.)
#  * C5, round 4: -DGRID_BS=1024 as well (with the branch-free searches the longer staging of a larger index is shared by four times the
#    points: C5 x 128 3.28 -> 3.22 s; 512 points per workgroup: 3.36 s).
#  * C5, round 4: -DENV_RK=4: a thread of the global-memory sort ranks four consecutive points, the later ones galloping on from their
#    predecessor's count (one or two dependent global reads instead of five; on C2's LDS-resident keys the same lost): C5 x 128
#    3.17 -> 3.12-3.14 s (2: 3.15, 8: 3.15, 16: 3.27).
#  * C2, round 4: -DENV_SEG_MINPTS=128 (default 192, tuned for k_envelope's eight walking waves): a stage-0 stream of the throughput
#    path (~1050 points) is cut into eight segments for its four walking waves instead of five; C2 a0=-5 x 4096 159.8 -> 157.3 ms per
#    solve (112 ... 160: 156.5 ... 157.7; a0=0 unchanged; the event cost of the planner 160 ... 1280: no difference).
BATCH_BUILD_FLAGS = {'C2': ['-DENV_MINW=3', '-DGRID_BS=512', '-DEG_GRID_CV_DEFAULT=1', '-DGRID_MINW=8', '-DENV_SEG_MINPTS=128'], 'C4': ['-DGRID_MINW=8', '-DGRID_BS=1024'], 'C5': ['-DGRID_MINW=8', '-DGRID_BS=1024', '-DENV_RK=4']}
BATCH_BUILD_MIN_DRAWS = {'C2': 1024, 'C4': 16, 'C5': 64}
