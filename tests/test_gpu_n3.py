"""Row N3: sigma^2 and everything downstream for accuracy='medium' and 'high', where the
linear power spectrum comes from a Boltzmann-code provider instead of the closed-form Eisenstein-Hu
spectrum (hmvec/cosmology.py:245-269,353-382,772-786).  CAMB itself cannot be installed in this
image, so the provider is `TabulatedBackground`: a P(k,z) TABLE (what one would save from a CAMB
run) whose z and k dependence does not factorise - the case the (z x k') . (k' x m) sigma^2
contraction must handle in general.  The GPU path is compared with the UNMODIFIED reference
run on the same table (fixture case_e) and with the oracle fed the same sPzk / Pzk arrays; what CAMB itself
would have put in the table stays an unpinned input."""
import numpy as np
import pytest

import os
import sys

from conftest import load_golden, merged_params, power_close, rel_err

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers"))
import pk_table  # noqa: E402

pytestmark = pytest.mark.gpu


def table(p):
    return pk_table.table(p["ns"])


@pytest.mark.parametrize("accuracy", ["medium", "high"])
@pytest.mark.parametrize("mf,tag", [("sheth-torman", "st"), ("tinker", "tk")])
def test_default_accuracies_against_the_reference_on_a_tabulated_spectrum(accuracy, mf, tag):
    """The drop-in constructor with a Boltzmann-code provider, end to end against the UNMODIFIED reference
    (tests/golden/case_e.npz: accuracy='medium'/'high' on the same table): the host seam feeds the device
    path, sigma2 / n / b and three spectra come out as the reference's."""
    import hmvec_amd as hm
    g = load_golden("case_e")
    p = merged_params()
    h = hm.HaloModel(g["zs"], g["ks"], ms=g["ms"], mass_function=mf, accuracy=accuracy,
                     background=hm.TabulatedBackground(p, *table(p)))
    pre = f"{accuracy}_{tag}_"
    assert rel_err(h.sPzk, g[f"{accuracy}_sPzk"]) < 1e-13 and rel_err(h.Pzk, g[f"{accuracy}_Pzk"]) < 1e-13
    assert rel_err(h.sigma2, g[pre + "sigma2"]) < 1e-12
    assert rel_err(h.bh, g[pre + "bh"]) < 1e-12
    assert np.allclose(h.nzm, g[pre + "nzm"], rtol=1e-10, atol=1e-300)
    meta = g["meta"]
    h.add_battaglia_profile("electron", family="AGN", xmax=meta["xmax"], nxs=meta["nxs"])
    h.add_hod("g", mthresh=10 ** 10.5 + g["zs"] * 0.0)
    for a, b in (("nfw", "nfw"), ("electron", "electron"), ("g", "g")):
        ok, w = power_close(h.get_power(a, b), g[pre + f"P_{a}_{b}"])
        assert ok, (a, b, w)


@pytest.mark.parametrize("accuracy", ["medium", "high"])
@pytest.mark.parametrize("mass_function", ["sheth-torman", "tinker"])
def test_sigma2_and_spectra_with_a_tabulated_power_spectrum(accuracy, mass_function, alpha_table):
    import hmvec_amd as hm
    from hmvec_amd.params import battaglia_defaults
    from oracle import hmref
    p = merged_params()
    zt, kt, P = table(p)
    prov = hm.TabulatedBackground(p, zt, kt, P)
    zs = np.array([0.05, 0.6, 1.4, 2.2, 3.0])
    ms = np.geomspace(1e10, 1e16, 72)
    ks = np.geomspace(1e-4, 80.0, 90)
    h = hm.HaloModel(zs, ks, ms=ms, mass_function=mass_function, accuracy=accuracy, background=prov)
    # the inputs really are non-separable: rows are not multiples of each other ('medium' rescales the
    # Eisenstein-Hu shape per redshift, so only its P(k) on the user grid carries the table's k-z mixing)
    ratio = (h.sPzk if accuracy == "high" else h.Pzk)[3] / (h.sPzk if accuracy == "high" else h.Pzk)[0]
    assert ratio.max() / ratio.min() > 1.05
    if accuracy == "high":
        ksig = np.geomspace(p["sigma2_kmin"], p["sigma2_kmax"], p["sigma2_numks"])
        assert rel_err(h.sPzk, prov.pk_interpolator(zs, 2000.0).P(zs, ksig)) < 1e-14
    h.add_battaglia_profile("electron", family="AGN", nxs=400, xmax=20)
    h.add_hod("g", mthresh=10 ** 10.5 + zs * 0.0)

    ksig = np.geomspace(p["sigma2_kmin"], p["sigma2_kmax"], p["sigma2_numks"])
    ci = hmref.CosmoInputs(h=h.h, omm0=h.omm0, ombh2=p["ombh2"], rho_crit_0=float(h.rho_critical_z(0.0)),
                           rho_crit_zs=h.rho_critical_z(zs), Pzk=h.Pzk, sPzk=h.sPzk, ks_sigma2=ksig,
                           h_of_z_zs=h.h_of_z(zs))
    o = hmref.RefHaloModel(ci, zs, ks, ms, p, mass_function=mass_function, alpha_table=alpha_table)
    assert rel_err(h.sigma2, o.sigma2) < 1e-12
    assert rel_err(h.bh, o.bh) < 1e-12
    assert np.allclose(h.nzm, o.nzm, rtol=1e-10, atol=1e-300)
    o.add_battaglia_profile("electron", "AGN", p["battaglia_gas_gamma"], battaglia_defaults["AGN"], 400, 20)
    o.add_hod("g", mthresh=10 ** 10.5 + zs * 0.0)
    for a, b in (("nfw", "nfw"), ("electron", "electron"), ("g", "g"), ("g", "electron"), ("nfw", "electron")):
        ok, w = power_close(h.get_power(a, b), o.get_power(a, b))
        assert ok, (a, b, w)


def test_get_sigma2_R_high_accuracy_matches_the_oracle_window_integral():
    """Cosmology.get_sigma2_R (hmvec/cosmology.py:245-269) on its own, accuracy='high', a handful of radii."""
    import hmvec_amd as hm
    from oracle import hmref
    p = merged_params()
    zt, kt, P = table(p)
    cos = hm.Cosmology(p, accuracy="high", background=hm.TabulatedBackground(p, zt, kt, P))
    zs = np.array([0.0, 1.0, 2.5])
    R = np.array([0.3, 2.0, 8.0 / 0.6766, 40.0])
    got = cos.get_sigma2_R(R, zs)
    ksig = np.geomspace(p["sigma2_kmin"], p["sigma2_kmax"], p["sigma2_numks"])
    w2 = hmref.tophat_window(ksig[None, None, :] * R[None, :, None], p["Wkr_taylor_switch"]) ** 2
    from scipy.integrate import simpson
    want = simpson(cos.sPzk[:, None, :] * w2 * ksig ** 2 / 2 / np.pi ** 2, x=ksig, axis=-1)
    assert rel_err(got, want) < 1e-12
