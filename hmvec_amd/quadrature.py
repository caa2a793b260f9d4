"""Quadrature rules of the path expressed as weight vectors.

The reference integrates with ``np.trapz`` over the (non-uniform) mass grid
(hmvec/hmvec.py:466,526,533,957) and with ``scipy.integrate.simpson(y, x=k)`` over
the geometric sigma^2 wavenumber grid (hmvec/cosmology.py:265).  Both rules are
linear in y with weights that depend on the abscissae only, so the device kernels
take a weight vector and do a fused weighted reduction; nothing of the integrand's
shape is materialised.
"""
import numpy as np


def trapz_weights(x):
    """w such that sum(w*y) == np.trapz(y, x) up to rounding."""
    x = np.asarray(x, dtype=np.float64)
    w = np.zeros_like(x)
    if x.size < 2:
        return w
    d = np.diff(x)
    w[:-1] += 0.5 * d
    w[1:] += 0.5 * d
    return w


def simpson_weights(x):
    """w such that sum(w*y) == scipy.integrate.simpson(y, x=x) (scipy >= 1.11).

    Composite Simpson for irregular spacing over point triples; when the number of
    points is even the last interval gets Cartwright's correction, which is what
    scipy does (the path's default grid has 10000 points, i.e. this branch).
    """
    x = np.asarray(x, dtype=np.float64)
    n = x.size
    w = np.zeros(n)
    if n < 2:
        return w
    if n == 2:
        w[:] = 0.5 * (x[1] - x[0])
        return w
    h = np.diff(x)
    last = n - 1 if n % 2 == 1 else n - 2      # Simpson pairs cover points [0, last]
    i = np.arange(0, last - 1, 2)
    h0, h1 = h[i], h[i + 1]
    hs = h0 + h1
    np.add.at(w, i, hs / 6.0 * (2.0 - h1 / h0))
    np.add.at(w, i + 1, hs / 6.0 * (hs * hs / (h0 * h1)))
    np.add.at(w, i + 2, hs / 6.0 * (2.0 - h0 / h1))
    if n % 2 == 0:
        a, b = h[-2], h[-1]
        w[-1] += (2.0 * b ** 2 + 3.0 * a * b) / (6.0 * (a + b))
        w[-2] += (b ** 2 + 3.0 * a * b) / (6.0 * a)
        w[-3] -= b ** 3 / (6.0 * a * (a + b))
    return w


def gradient_is_uniform(x):
    """np.gradient switches to its uniform-spacing stencil when every diff is bit-equal."""
    d = np.diff(np.asarray(x, dtype=np.float64))
    return bool(d.size and (d == d[0]).all()), (float(d[0]) if d.size else 0.0)
