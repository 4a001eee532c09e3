"""Argument sets for the model-function accessor (egdst_call.c), shared by the CPU-harness and the GPU parity tests."""
import numpy as np


def call_cases(model, nt, nst, nd, seed=3):
    """[(sw, args)] covering every function, vector input, values outside the domains and bad indices."""
    rng = np.random.default_rng(seed)
    t0, a0, mmax = model.t0, model.a0, model.mmax
    n = 40
    it = rng.integers(t0, t0 + nt, n).astype(float)
    ist = rng.integers(1, nst + 1, n).astype(float)
    idc = rng.integers(1, nd + 1, n).astype(float)
    cons = rng.uniform(0.05, (mmax - a0) * 1.1, n)           # some above mmax-a0 -> NaN
    sav = rng.uniform(a0 - 0.5, mmax, n)                      # some below a0 -> NaN
    ist1 = rng.integers(1, nst + 1, n).astype(float)
    shock = rng.uniform(0.5, 1.5, n)
    cash = rng.uniform(a0, mmax * 1.05, n)                    # some above mmax -> NaN
    cases = [
        (1, np.column_stack([it, ist, idc, cons])),
        (2, np.column_stack([it, ist, idc, cons])),
        (3, np.column_stack([it, ist])),
        (4, np.column_stack([it, ist, idc, sav, ist1, shock])),
        (5, np.column_stack([it, ist, idc, sav, ist1, shock])),
        (6, np.column_stack([it, ist, cash])),
        (6, np.column_stack([np.full(n, float(t0 + nt - 1)), ist, cash])),   # terminal period: utility of cash
    ]
    bad = np.column_stack([it, ist, idc, cons]); bad[17, 1] = nst + 3            # bad ist: rows 17.. are NaN
    cases.append((1, bad))
    bad2 = np.column_stack([it, ist, idc, sav, ist1, shock]); bad2[:, 3] = np.abs(bad2[:, 3]) + a0
    bad2[:, 0] = np.minimum(bad2[:, 0], t0 + nt - 2); bad2[9, 4] = 0              # bad ist1: row 9 stays 0, later NaN
    cases.append((4, bad2))
    cases.append((3, np.column_stack([it, ist, idc])))                             # wrong column count: zeros
    cases.append((9, np.column_stack([it, ist])))                                  # unknown switch: NaN
    badit = np.column_stack([it, ist, cash]); badit[0, 0] = t0 - 1                 # first row bad: all NaN
    cases.append((6, badit))
    return cases
