import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as f:
        d = {k: f[k] for k in f.files}
    if "meta_json" in d:
        d["meta"] = json.loads(str(d.pop("meta_json")))
    return d


@pytest.fixture(scope="session")
def alpha_table():
    g = load_golden("unit_pins")
    return g["tinker_alpha_z"], g["tinker_alpha"]


def merged_params(overrides=None):
    from hmvec_amd.params import default_params
    p = dict(default_params)
    p.update(overrides or {})
    return p


def cosmo_inputs_from_golden(g, p):
    """Build the oracle's CosmoInputs from the arrays the reference saw."""
    from oracle.hmref import CosmoInputs
    ksig = np.geomspace(p["sigma2_kmin"], p["sigma2_kmax"], p["sigma2_numks"])
    return CosmoInputs(h=float(g["in_h"]), omm0=float(g["in_omm0"]), ombh2=p["ombh2"],
                       rho_crit_0=float(np.ravel(g["in_rho_crit_0"])[0]),
                       rho_crit_zs=g["in_rho_crit_zs"], Pzk=g["in_Pzk"], sPzk=g["in_sPzk"],
                       ks_sigma2=ksig, h_of_z_zs=g["in_h_of_z_zs"])


def rel_err(a, b):
    a, b = np.asarray(a), np.asarray(b)
    scale = np.maximum(np.abs(b), 1e-300)
    return float(np.max(np.abs(a - b) / scale))


def power_close(P, Pref, rtol=1e-8, atol_frac=1e-12):
    """Parity gate of SURVEY §8(d): |dP| <= rtol*|Pref| + atol_frac*max_k|Pref(z,.)|."""
    P, Pref = np.asarray(P), np.asarray(Pref)
    tol = rtol * np.abs(Pref) + atol_frac * np.max(np.abs(Pref), axis=-1, keepdims=True)
    bad = np.abs(P - Pref) > tol
    return not bad.any(), float(np.max(np.abs(P - Pref) / np.maximum(tol, 1e-300)))


ROUTE_SWITCHES = ("HMG_FUSED_FFT", "HMG_PRUNED_FFT", "HMG_BAND_FFT", "HMG_CHIRP", "HMG_FUSED_GENERIC", "HMG_NO_HINTS",
                  "HMG_PRUNED_LP_MIN", "HMG_FUSED_MAX_M", "HMG_FUSED_PREFER_M")


@pytest.fixture
def default_routes(monkeypatch):
    """For tests that are ABOUT one route of hmg_profile_fft (its bound, its hints, its fault word): the suite may run
    under a switch that reroutes that very part (tools/env_matrix.sh); such a test pins the default routes first."""
    for sw in ROUTE_SWITCHES:
        monkeypatch.delenv(sw, raising=False)
    return monkeypatch
