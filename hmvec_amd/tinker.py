"""``hmvec.tinker`` mirror: Tinker et al. 2010 halo bias and multiplicity function
(hmvec/tinker.py:26-67), evaluated on the GPU through ``hmg_fn2d``.

The alpha(z) normalisation table is the VALUES of the reference's data file
(hmvec/data/alpha_consistency.txt), shipped as hmvec_amd/data/tinker10_alpha_of_z.npz: the
reference's generator script does not reproduce it to better than 1 % (SURVEY 8a, A4 iii)."""
import os

import numpy as np

from .functions import FN_TINKER_BIAS, FN_TINKER_FNU, fn2d

constants = {"deltac": 1.686}
default_params = {"tinker_f_nu_alpha_z0_delta_200": 0.368}   # Tinker et al 2010 table 4

_TABLE = None


def _alpha_table():
    global _TABLE
    if _TABLE is None:
        d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "tinker10_alpha_of_z.npz"))
        _TABLE = (np.ascontiguousarray(d["z"], dtype=np.float64), np.ascontiguousarray(d["alpha"], dtype=np.float64))
    return _TABLE


def bias(nu, delta=200.0):
    """Eq. 6 of Tinker et al. 2010 (hmvec/tinker.py:26-40)."""
    return fn2d(FN_TINKER_BIAS, [nu], [delta])


def f_nu(nu, zs, delta=200.0, norm_consistency=True, alpha=default_params["tinker_f_nu_alpha_z0_delta_200"]):
    """f(nu) of Tinker et al. 2010 with the redshift scalings of its eqs. 9-12; z is clamped with the
    reference's heaviside expression (exactly z = 3 maps to z = 0, z > 3 to 3; hmvec/tinker.py:53).
    With norm_consistency the normalisation alpha(z) is interpolated linearly in the table and, as
    in the reference (interp1d(bounds_error=True)), z outside the table raises ValueError."""
    assert np.isclose(delta, 200.0), "delta!=200 note implemented yet."
    zs = np.asarray(zs, dtype=np.float64)
    tz, ta = _alpha_table()
    if norm_consistency:
        zc = zs * np.heaviside(3 - zs, 0) + 3 * np.heaviside(zs - 3, 0)
        if np.any(zc < tz[0]):
            raise ValueError("A value in x_new is below the interpolation range.")
        if np.any(zc > tz[-1]):
            raise ValueError("A value in x_new is above the interpolation range.")
    return fn2d(FN_TINKER_FNU, [nu, zs], [1.0 if norm_consistency else 0.0, float(alpha), float(tz.size)],
                tables=(tz, ta))


def simple_f_nu(nu, delta=200.0):
    raise NotImplementedError("simple_f_nu (hmvec/tinker.py:70-78) is not on the accelerated path")


def NlnMsub(Msubs, Mhosts):
    raise NotImplementedError("NlnMsub (hmvec/tinker.py:81-) is not on the accelerated path")


__all__ = ["bias", "f_nu", "simple_f_nu", "NlnMsub", "constants", "default_params"]
