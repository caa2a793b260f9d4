"""hmvec_amd — MI355X-native implementation of hmvec's halo-model power-spectrum hot path.

Drop-in surface (reference: hmvec/__init__.py:1 ``from .hmvec import *``): ``HaloModel`` with
the reference's constructor / ``add_*`` / ``get_power_*`` API, the parameter tables and the
small host helpers users import by name.  The compute path is hand-written HIP behind the
C ABI in ``include/hmgrid.h``; importing this package does not need the GPU library, the
first kernel call does (and fails loudly if it is not built).
"""
from .params import battaglia_defaults, default_params  # noqa: F401
from .cosmology import Cosmology  # noqa: F401
from .background import AnalyticBackground, CambBackground, TabulatedBackground  # noqa: F401
from .halomodel import HaloModel  # noqa: F401
from .functions import *  # noqa: F401,F403  (the reference's free functions, GPU-backed)
from .fft import generic_profile_fft  # noqa: F401  (hmvec/hmvec.py:3 star-imports it into the package)
from . import cosmology, fft, functions, params, quadrature, tinker, utils  # noqa: F401
from .functions import __all__ as _fn_all

__all__ = ["HaloModel", "Cosmology", "default_params", "battaglia_defaults", "generic_profile_fft",
           "fft", "tinker", "utils", "cosmology", "params"] + list(_fn_all)
