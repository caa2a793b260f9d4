"""Pin the CPU oracle (oracle/hmref.py) to the reference's own outputs.

The fixtures in tests/golden were produced by running the unmodified reference
(tools/make_golden.py).  Everything here runs on CPU.
"""
import numpy as np
import pytest
import scipy.constants as sc

from conftest import cosmo_inputs_from_golden, load_golden, merged_params, power_close, rel_err
from oracle import hmref
from hmvec_amd.params import battaglia_defaults


def build_oracle(g, alpha_table):
    meta = g["meta"]
    p = merged_params(meta["params"])
    ci = cosmo_inputs_from_golden(g, p)
    om = hmref.RefHaloModel(ci, g["zs"], g["ks"], g["ms"], p, mass_function=meta["mass_function"],
                            mdef=meta["mdef"], alpha_table=alpha_table)
    return om, p, meta


def add_everything(om, g, p, meta):
    fitp = dict(battaglia_defaults[meta["family"]])
    gamma = p["battaglia_gas_gamma"]
    for k, v in (meta["batt_override"] or {}).items():
        if k == "battaglia_gas_gamma":
            gamma = v
        elif k in fitp:
            fitp[k] = v
    om.add_battaglia_profile("electron", meta["family"], gamma, fitp, meta["nxs"], meta["xmax"])
    central = "electron" if meta["central"] else None
    if meta["ngal_mode"]:
        om.add_hod("g", ngal=g["ngal_target"], corr=meta["corr"], central_profile_name=central)
    else:
        om.add_hod("g", mthresh=10 ** 10.5 + g["zs"] * 0.0, corr=meta["corr"], central_profile_name=central)
    if meta["pres"]:
        sigT = sc.physical_constants["Thomson cross section"][0]
        me = sc.physical_constants["electron mass"][0] / p["mSun"]
        om.add_battaglia_pres_profile("y", p["battaglia_pres_alpha"], p["battaglia_pres_gamma"],
                                      battaglia_defaults["pres"], meta["nxs"], meta["xmax"], sigT, me, sc.c)


@pytest.fixture(scope="module", params=["case_a", "case_b", "case_c"])
def case(request, alpha_table):
    g = load_golden(request.param)
    om, p, meta = build_oracle(g, alpha_table)
    add_everything(om, g, p, meta)
    return g, om, p, meta


def test_mass_function(case):
    g, om, _, _ = case
    assert rel_err(om.sigma2, g["sigma2"]) < 1e-12
    assert rel_err(om.nzm, g["nzm"]) < 1e-11
    assert rel_err(om.bh, g["bh"]) < 1e-12
    assert rel_err(om.cs, g["cs"]) < 1e-14
    assert rel_err(om.rvirs, g["rvir"]) < 1e-14


def test_profiles(case):
    g, om, _, meta = case
    assert np.max(np.abs(om.uk_profiles["nfw"] - g["uk_nfw"])) < 1e-14
    assert rel_err(om.m200c, g["m200c"]) < 1e-13
    assert np.max(np.abs(om.uk_profiles["electron"] - g["uk_electron"])) < 1e-13
    if meta["pres"]:
        assert rel_err(om.pk_profiles["y"], g["pk_y"]) < 1e-10


def test_numeric_nfw(alpha_table):
    g = load_golden("case_a")
    om, p, meta = build_oracle(g, alpha_table)
    nn, xm = meta["numeric_nfw"]
    _, u = om.add_nfw_profile("nfwnum", numeric=True, nxs=nn, xmax=xm)
    assert np.max(np.abs(u - g["uk_nfwnum"])) < 1e-13


def test_hod(case):
    g, om, _, _ = case
    for k in ("Nc", "Ns", "NsNsm1", "NcNs", "ngal", "bg", "log10mthresh"):
        ref = g["hod_" + k]
        got = om.hods["g"][k]
        assert np.allclose(got, ref, rtol=1e-11, atol=1e-300), k


def test_spectra(case):
    g, om, _, meta = case
    names = ["nfw", "electron", "g"] + (["y"] if meta["pres"] else [])
    for i, a in enumerate(names):
        for b in names[i:]:
            ok, worst = power_close(om.get_power_1halo(a, b), g[f"P1h_{a}_{b}"], rtol=1e-10)
            assert ok, ("1h", a, b, worst)
            ok, worst = power_close(om.get_power_2halo(a, b), g[f"P2h_{a}_{b}"], rtol=1e-10)
            assert ok, ("2h", a, b, worst)
    ok, _ = power_close(om.get_power_1halo("electron", "nfw"), g["P1h_electron_nfw"], rtol=1e-10)
    assert ok
    ok, _ = power_close(om.get_power_2halo("g", "nfw", b1_in=g["b1_in"], b2_in=g["b2_in"]),
                        g["P2h_g_nfw_bin"], rtol=1e-10)
    assert ok
    ok, _ = power_close(om.get_power("g", "electron"), g["P_tot_g_electron"], rtol=1e-10)
    assert ok


def test_unit_pins(alpha_table):
    u = load_golden("unit_pins")
    kt, uk = hmref.sine_transform(u["fftint_x"], np.exp(-u["fftint_x"] ** 2 / 2.0))
    assert np.allclose(kt[:400], u["fftint_k"], rtol=1e-15)
    assert np.max(np.abs(uk[:400] - u["fftint_u"])) < 1e-15
    # the reference's own analytic check (fft.py:36-43): loose because of the phase quirk
    sel = (kt > 0) & (kt < 3.0)
    ana = np.sqrt(np.pi / 2.0) * np.exp(-kt[sel] ** 2 / 2.0) * kt[sel]
    assert np.max(np.abs(uk[sel] / ana - 1)) < 2e-2
    y, _ = hmref.bisection(u["bisect_x"], lambda v: np.sqrt(v), (1, 40), "increasing", rtol=1e-4)
    assert np.array_equal(y, u["bisect_y"])
    assert np.allclose(y, [4.0, 16.0, 36.0], rtol=1e-3)          # hmvec/utils.py:45-51
    assert rel_err(hmref.tinker_bias(u["tinker_nu"]), u["tinker_bias"]) < 1e-14
    assert rel_err(hmref.tinker_fnu(u["tinker_nu"], u["tinker_z"], alpha_table), u["tinker_fnu"]) < 1e-13
    assert abs(alpha_table[1][0] - 0.368) < 1e-3                 # Tinker+10 table 4 anchor
    assert rel_err(hmref.mhalo_of_mstellar(u["shmr_z"], u["shmr_logmstar"]), u["shmr_Mhalo_stellar"]) < 1e-14
    assert rel_err(hmref.mstellar_of_mhalo(u["shmr_z"], u["shmr_logmhalo"]), u["shmr_Mstellar_halo"]) < 1e-13


def test_limber(alpha_table):
    g = load_golden("case_c")
    meta = g["meta"]
    lz, gzs = meta["limber"]["lzs"], meta["limber"]["gzs"]
    from hmvec_amd.background import AnalyticBackground
    p = merged_params(meta["params"])
    bg = AnalyticBackground(p["H0"], p["ombh2"], p["omch2"])
    zs, ks = g["zs"], g["ks"]
    chis, hz = bg.comoving_radial_distance(zs), bg.h_of_z(zs)
    W = hmref.lensing_window(zs, lz, bg.h_of_z(0.0), hz, chis, bg.comoving_radial_distance(lz), float(g["in_omm0"]))
    assert rel_err(W, g["lensing_window"]) < 1e-13
    Pmm = g["P1h_nfw_nfw"] + g["P2h_nfw_nfw"]
    ckk = hmref.limber_integral(g["ells"], zs, ks, Pmm, zs, W, W, hz, chis)
    assert rel_err(ckk, g["C_kk"]) < 1e-12
    Pgm = g["P1h_nfw_g"] + g["P2h_nfw_g"]
    gz = np.array([gzs])
    Wg = hmref.lensing_window(gz, lz, bg.h_of_z(0.0), bg.h_of_z(gz), bg.comoving_radial_distance(gz),
                              bg.comoving_radial_distance(lz), float(g["in_omm0"]))
    ckg = hmref.limber_integral(g["ells"], zs, ks, Pgm, gz, Wg, 1.0, bg.h_of_z(gz), bg.comoving_radial_distance(gz))
    assert rel_err(ckg, g["C_kg"]) < 1e-12


def test_second_tracers_and_tsz_projections(alpha_table):
    """case_d: the 1-halo term of two DIFFERENT HOD names / two different pressure names takes the square
    term of the first name only (hmvec/hmvec.py:510-513) - order-dependent by tens of per cent - and the
    tSZ Limber projections C_yy / C_ky (hmvec/cosmology.py:585-597)."""
    g = load_golden("case_d")
    om, p, meta = build_oracle(g, alpha_table)
    add_everything(om, g, p, meta)
    zs, ks = g["zs"], g["ks"]
    om.add_hod("g2", mthresh=10 ** 11.0 + zs * 0.0, corr="min")
    sigT = sc.physical_constants["Thomson cross section"][0]
    me = sc.physical_constants["electron mass"][0] / p["mSun"]
    fit2 = dict(battaglia_defaults["pres"], P0_A0=25.0, xc_A0=0.6)
    om.add_battaglia_pres_profile("y2", p["battaglia_pres_alpha"], -0.4, fit2, meta["nxs"], meta["xmax"], sigT, me, sc.c)
    assert np.max(np.abs(g["P1h_g_g2"] / g["P1h_g2_g"] - 1)) > 0.1      # the fixture does exercise the rule
    for a, b in (("g", "g2"), ("g2", "g"), ("y", "y2"), ("y2", "y")):
        ok, w = power_close(om.get_power_1halo(a, b), g[f"P1h_{a}_{b}"])
        assert ok, (a, b, w)
        ok, w = power_close(om.get_power_2halo(a, b), g[f"P2h_{a}_{b}"])
        assert ok, (a, b, w)
    from hmvec_amd.background import AnalyticBackground
    bg = AnalyticBackground(p["H0"], p["ombh2"], p["omch2"])
    chis, hz = bg.comoving_radial_distance(zs), bg.h_of_z(zs)
    lz = meta["limber"]["lzs"]
    W = hmref.lensing_window(zs, lz, bg.h_of_z(0.0), hz, chis, bg.comoving_radial_distance(lz), float(g["in_omm0"]))
    Pyy = g["P1h_y_y"] + g["P2h_y_y"]
    Pym = g["P1h_nfw_y"] + g["P2h_nfw_y"]
    one = np.ones(zs.size)
    assert rel_err(hmref.limber_integral(g["ells"], zs, ks, Pyy, zs, one, one, hz, chis), g["C_yy"]) < 1e-12
    assert rel_err(hmref.limber_integral(g["ells"], zs, ks, Pym, zs, W, one, hz, chis), g["C_ky"]) < 1e-12


def test_long_radial_grids_case_f():
    """case_f: the reference run with the radial grids its own callers use (nxs = 30000 / xmax = 50 gas,
    nxs = 40000 / xmax = 200 numeric NFW, nxs = 30000 / xmax = 2 pressure) pins the oracle at those lengths."""
    g = load_golden("case_f")
    p = merged_params()
    ci = cosmo_inputs_from_golden(g, p)
    om = hmref.RefHaloModel(ci, g["zs"], g["ks"], g["ms"], p)
    om.add_battaglia_profile("electron", "AGN", p["battaglia_gas_gamma"], battaglia_defaults["AGN"], 30000, 50)
    assert np.max(np.abs(om.uk_profiles["electron"] - g["uk_electron"])) < 1e-13
    _, u = om.add_nfw_profile("nfwnum", numeric=True)
    assert np.max(np.abs(u - g["uk_nfwnum"])) < 1e-13
    sigT = sc.physical_constants["Thomson cross section"][0]
    me = sc.physical_constants["electron mass"][0] / p["mSun"]
    om.add_battaglia_pres_profile("y", p["battaglia_pres_alpha"], p["battaglia_pres_gamma"],
                                  battaglia_defaults["pres"], 30000, 2, sigT, me, sc.c)
    assert rel_err(om.pk_profiles["y"], g["pk_y"]) < 1e-10
    om.add_hod("g", mthresh=10 ** 10.5 + g["zs"] * 0.0, satellite_profile_name="nfwnum")
    for a, b in (("electron", "electron"), ("nfw", "electron"), ("nfwnum", "nfwnum"), ("g", "electron"), ("y", "y"),
                 ("nfw", "y")):
        ok, w = power_close(om.get_power_1halo(a, b), g[f"P1h_{a}_{b}"])
        assert ok, (a, b, w)
        ok, w = power_close(om.get_power_2halo(a, b), g[f"P2h_{a}_{b}"])
        assert ok, (a, b, w)
