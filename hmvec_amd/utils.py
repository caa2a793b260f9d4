"""``hmvec.utils`` mirror.  ``vectorized_bisection_search`` is host control flow around a
caller-supplied function (hmvec/utils.py:9-42): the function is where the arithmetic is, and
for the path's only caller - ``HaloModel.add_hod(ngal=...)`` - that function is the device HOD
kernel.  What matters for parity is the loop's GLOBAL stop test: every element keeps bisecting
until all of them meet rtol, so results depend on the whole input vector (SURVEY 8e exception).
"""
import numpy as np


def vectorized_bisection_search(x, inv_func, ybounds, monotonicity, rtol=1e-4, verbose=True, hang_check_num_iter=20):
    """Solve inv_func(y) = x for y in ybounds by bisection, to a relative tolerance rtol on x."""
    assert monotonicity in ["increasing", "decreasing"]
    x = np.asarray(x, dtype=np.float64)
    lo = x * 0 + ybounds[0]
    hi = x * 0 + ybounds[1]
    rising = monotonicity == "increasing"
    miss = np.inf
    n_iter, warned = 0, False
    while np.any(np.abs(miss) > rtol):
        mid = (lo + hi) / 2.0
        miss = (inv_func(mid) - x) / x
        over, under = miss > 0, miss <= 0
        if rising:
            hi[over] = mid[over]
            lo[under] = mid[under]
        else:
            lo[over] = mid[over]
            hi[under] = mid[under]
        n_iter += 1
        if n_iter > hang_check_num_iter and not warned:
            print("WARNING: Bisection search has done more than ", hang_check_num_iter, " loops. Still searching...")
            warned = True
    if verbose:
        print("Bisection search converged in ", n_iter, " iterations.")
    return mid


def interp(x, y, bounds_error=False, fill_value=0.0, **kwargs):
    """hmvec/utils.py:6-7: thin scipy wrapper used by the cosmology layer."""
    from scipy.interpolate import interp1d
    return interp1d(x, y, bounds_error=bounds_error, fill_value=fill_value, **kwargs)


__all__ = ["vectorized_bisection_search", "interp"]
