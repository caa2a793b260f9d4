"""The oracle's restatements of the reference's free functions, pinned against
tests/golden/func_pins.npz (outputs of the unmodified reference, tools/make_golden.py
run_function_pins).  CPU only; the GPU mirrors are checked against the same file in
tests/test_gpu_functions.py."""
import numpy as np

from conftest import load_golden, rel_err
from oracle import hmref

G = load_golden("func_pins")
Z, MS, RHOC, RHOM = G["z"], G["ms"], G["rhoc"], G["rhom"]
OMB, OMM = 0.049, 0.315


def test_mass_conversion_secant():
    cs = G["duffy_default"]
    got = hmref.mdelta_from_mdelta(MS, cs, 200.0 * RHOM, 200.0 * RHOC)
    assert rel_err(got, G["mdelta"]) < 1e-13
    # the reference's own vectorised/unvectorised cross-check (bin/tests.py:293-295): secant tolerance
    assert rel_err(G["mdelta_unvec"], G["mdelta"]) < 1e-7


def test_battaglia_fits_and_profiles():
    from hmvec_amd.params import battaglia_defaults, default_params
    m3, z3, rc3 = MS[None, :, None], Z[:, None, None], RHOC[:, None, None]
    x = G["x"][None, None]
    assert rel_err(hmref.battaglia_fit(m3, z3, 4000.0, 0.29, -0.66), G["batt_fit"]) < 1e-14
    got = hmref.rho_gas_x(x, m3, z3, OMB, OMM, rc3, default_params["battaglia_gas_gamma"], battaglia_defaults["AGN"])
    assert rel_err(got, G["rho_gas_generic_x"]) < 1e-13
    got = hmref.rho_gas_x(x, m3, z3, OMB, OMM, rc3, -0.25, battaglia_defaults["SH"])
    assert rel_err(got, G["rho_gas_generic_x_SH"]) < 1e-13
    r200 = hmref.lagrangian_radius(m3, rc3, 200.0)
    got = hmref.pressure_x(x, m3, r200, z3, OMB, OMM, rc3, default_params["battaglia_pres_alpha"],
                           default_params["battaglia_pres_gamma"], battaglia_defaults["pres"],
                           default_params["parsec"], default_params["mSun"])
    assert rel_err(got, G["P_e_generic_x"]) < 1e-13


def test_hod_helpers():
    from hmvec_amd.params import default_params
    p = dict(default_params)
    Nc, Ns, nn, cn = hmref.hod_occupations(MS, Z, G["hod_thr"][:, 0], p, "max")
    assert rel_err(Nc, G["avg_Nc"]) < 1e-13 and rel_err(Ns, G["avg_Ns"]) < 1e-13
    assert np.allclose(nn, G["NsNsm1_max"], rtol=1e-13, atol=0) and np.array_equal(cn, G["NcNs_max"])
    _, _, nn, cn = hmref.hod_occupations(MS, Z, G["hod_thr"][:, 0], p, "min")
    assert np.allclose(nn, G["NsNsm1_min"], rtol=1e-13, atol=0) and np.allclose(cn, G["NcNs_min"], rtol=1e-13, atol=0)


def test_profile_fft_with_user_callables():
    cmax, rss, ks = G["gpf_cmax"], G["gpf_rss"][..., 0], G["ks"]
    u = hmref.profile_fft(lambda xx: 1.0 / xx / (1.0 + xx) ** 2, cmax, rss, Z, ks, 60.0, 3000)
    assert np.max(np.abs(u - G["gpf_shared"])) < 1e-14
    slope = G["gpf_slope"]
    u = hmref.profile_fft(lambda xx: xx ** -0.5 * (1.0 + xx) ** -slope, cmax, rss, Z, ks, 30.0, 1001)
    assert np.max(np.abs(u - G["gpf_rows_odd_nxs"])) < 1e-14
    u = hmref.profile_fft(lambda xx: np.exp(-xx) + 0 * slope, 0 * cmax + 4.0, rss, Z, ks, 12.0, 640, do_mass_norm=False)
    assert rel_err(u, G["gpf_nonorm"]) < 1e-12


def test_sine_transform_2d():
    kt, u = hmref.sine_transform(G["fi_x"], G["fi_y"])
    assert np.array_equal(kt[:300], G["fi_k"]) and np.max(np.abs(u[:, :300] - G["fi_u"])) < 1e-15
