"""Gauss-Legendre nodes/weights on [x1,x2].

Host-side counterpart of the local function ``quadpoints`` in the reference class
(@egdstmodel/egdstmodel.m:1504-1529, attributed there to John Rust): Newton
iteration on the Legendre polynomial roots until ``abs(z-z1) <= eps`` with the
machine epsilon.  ``solve`` stores ``quadrature=[qw qx]`` (ny x 2, column-major:
the ny weights first, then the ny abscissae on [0,1]; egdstmodel.m:1157-1160);
the solver maps the abscissae through the inverse normal cdf itself
(egdst_solver.c:162-164).
"""
import math

import numpy as np

_EPS = 2.0 ** -52


def quadpoints(n, x1=0.0, x2=1.0):
    """Return (x, w): n abscissae and weights over [x1, x2]."""
    n = int(n)
    x = np.zeros(n)
    w = np.zeros(n)
    m = (n + 1) / 2.0  # MATLAB `for i=1:m` runs i=1..floor(m)
    xm = 0.5 * (x2 + x1)
    xl = 0.5 * (x2 - x1)
    for i in range(1, int(math.floor(m)) + 1):
        z = math.cos(math.pi * (i - 0.25) / (n + 0.5))
        z1 = 2.0
        pp = 1.0
        while abs(z - z1) > _EPS:
            p1 = 1.0
            p2 = 0.0
            for j in range(1, n + 1):
                p3 = p2
                p2 = p1
                p1 = ((2.0 * j - 1.0) * z * p2 - (j - 1.0) * p3) / j
            pp = n * (z * p1 - p2) / (z * z - 1.0)
            z1 = z
            z = z1 - p1 / pp
        x[i - 1] = xm - xl * z
        x[n - i] = xm + xl * z
        w[i - 1] = 2.0 * xl / ((1.0 - z * z) * pp * pp)
        w[n - i] = w[i - 1]
    return x, w


def quadrature_array(ny):
    """``[qw qx]`` flattened column-major (weights then abscissae), length 2*ny.

    For ny==1 the reference leaves ``quadrature`` empty (egdstmodel.m:1157) and the
    solver never reads it (niy==1 branch, egdst_solver.c:510,525-529); we return a
    harmless [1, .5] so the C-ABI always receives 2*ny doubles.
    """
    if ny <= 1:
        return np.array([1.0, 0.5])
    qx, qw = quadpoints(ny, 0.0, 1.0)
    return np.concatenate([qw, qx])
