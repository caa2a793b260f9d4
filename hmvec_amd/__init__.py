"""hmvec_amd — MI355X-native implementation of hmvec's halo-model power-spectrum hot path.

Drop-in surface (reference: hmvec/__init__.py:1 ``from .hmvec import *``): ``HaloModel`` with
the reference's constructor / ``add_*`` / ``get_power_*`` API, the parameter tables and the
small host helpers users import by name.  The compute path is hand-written HIP behind the
C ABI in ``include/hmgrid.h``; importing this package does not need the GPU library, the
first kernel call does (and fails loudly if it is not built).
"""
from .params import battaglia_defaults, default_params  # noqa: F401
from .cosmology import Cosmology  # noqa: F401
from .halomodel import HaloModel, R_from_M, duffy_concentration  # noqa: F401
from . import cosmology, params, quadrature  # noqa: F401

__all__ = ["HaloModel", "Cosmology", "default_params", "battaglia_defaults",
           "duffy_concentration", "R_from_M"]
