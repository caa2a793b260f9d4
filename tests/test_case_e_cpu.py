"""Row N3's host seam pinned to the reference.  tests/golden/case_e.npz was written by
the UNMODIFIED reference running accuracy='medium' and 'high' on the tabulated, non-separable P(k,z) of
tests/helpers/pk_table.py (served to it through a stand-in camb.get_matter_power_interpolator,
tools/make_golden.py).  Here, on CPU:
  * this repo's host code - Cosmology.P_lin / P_lin_slow / _get_matter_power over TabulatedBackground
    (hmvec/cosmology.py:227-229,353-389,772-786) - reproduces the reference's sPzk and Pzk, and
  * the oracle fed those arrays reproduces the reference's sigma2, n(z,m), b(z,m) and spectra.
Also: Cosmology.sigma_crit and Cosmology.bias_fnl against tests/golden/extra_pins.npz."""
import os
import sys

import numpy as np
import pytest

from conftest import cosmo_inputs_from_golden, load_golden, merged_params, power_close, rel_err

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers"))
from pk_table import table  # noqa: E402


@pytest.mark.parametrize("accuracy", ["medium", "high"])
def test_host_seam_reproduces_the_reference_spectra_inputs(accuracy):
    import hmvec_amd as hm
    g = load_golden("case_e")
    p = merged_params()
    cos = hm.Cosmology(p, accuracy=accuracy, background=hm.TabulatedBackground(p, *table(p["ns"])))
    ksig = np.geomspace(p["sigma2_kmin"], p["sigma2_kmax"], p["sigma2_numks"])
    sP = cos.P_lin(ksig, g["zs"]) if accuracy == "medium" else cos.P_lin_slow(ksig, g["zs"], kmax=p["sigma2_kmax"])
    assert rel_err(sP, g[f"{accuracy}_sPzk"]) < 1e-13
    assert rel_err(cos._get_matter_power(g["zs"], g["ks"]), g[f"{accuracy}_Pzk"]) < 1e-13
    assert abs(cos.h - float(g["in_h"])) < 1e-15 and abs(cos.omm0 - float(g["in_omm0"])) < 1e-15


@pytest.mark.parametrize("accuracy", ["medium", "high"])
@pytest.mark.parametrize("mf,tag", [("sheth-torman", "st"), ("tinker", "tk")])
def test_oracle_reproduces_the_reference_downstream_of_the_seam(accuracy, mf, tag, alpha_table):
    from hmvec_amd.params import battaglia_defaults
    from oracle import hmref
    g = load_golden("case_e")
    p = merged_params()
    gi = dict(g, in_Pzk=g[f"{accuracy}_Pzk"], in_sPzk=g[f"{accuracy}_sPzk"])
    ci = cosmo_inputs_from_golden(gi, p)
    o = hmref.RefHaloModel(ci, g["zs"], g["ks"], g["ms"], p, mass_function=mf, alpha_table=alpha_table)
    pre = f"{accuracy}_{tag}_"
    assert rel_err(o.sigma2, g[pre + "sigma2"]) < 1e-12
    assert rel_err(o.nzm, g[pre + "nzm"]) < 1e-11
    assert rel_err(o.bh, g[pre + "bh"]) < 1e-12
    meta = g["meta"]
    o.add_battaglia_profile("electron", "AGN", p["battaglia_gas_gamma"], battaglia_defaults["AGN"], meta["nxs"], meta["xmax"])
    o.add_hod("g", mthresh=10 ** 10.5 + g["zs"] * 0.0)
    for a, b in (("nfw", "nfw"), ("electron", "electron"), ("g", "g")):
        ok, w = power_close(o.get_power(a, b), g[pre + f"P_{a}_{b}"])
        assert ok, (a, b, w)


def test_sigma_crit_and_bias_fnl_against_the_reference():
    import hmvec_amd as hm
    g = load_golden("extra_pins")
    cos = hm.Cosmology(merged_params(), accuracy="low", engine="analytic")
    assert rel_err(cos.sigma_crit(g["zs"], 2.0), g["sigma_crit"]) < 1e-12
    for z in (0.0, 0.8):
        assert rel_err(cos.bias_fnl(1.8, 25.0, z, g["ks"]), g[f"bias_fnl_z{z}"]) < 1e-12
    assert rel_err(cos.bias_fnl(2.4, -10.0, 0.5, g["ks"], deltac=1.686), g["bias_fnl_deltac"]) < 1e-12
