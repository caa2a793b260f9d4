#!/usr/bin/env python3
"""Generate golden fixtures by running the UNMODIFIED reference (simonsobs/hmvec,
mounted read-only at /root/reference) in this container.

Only runs where /root/reference exists (never on the GPU box).  Output: small
``tests/golden/*.npz`` files holding inputs + every intermediate + outputs of the
hot path.  Fixtures are data; no reference source is written anywhere.

How the reference is made importable: it does ``import camb`` at
module scope (hmvec/hmvec.py:5, hmvec/cosmology.py:5) and camb is not installed
(no network).  camb only supplies *inputs* to the path (H(z), distances,
Omega_nu), so a stand-in module is registered in ``sys.modules`` whose
``get_background`` returns this repo's ``AnalyticBackground``; with
``accuracy='low'`` P(k) comes from the reference's own Eisenstein-Hu code.  The
path under test (hmvec/hmvec.py, fft.py, tinker.py, utils.py and
Cosmology.get_sigma2_R) runs exactly as shipped.

Two environment shims, both recorded in DESIGN.md:
  * tinker.py:64 reads ``<pkg>/../data/alpha_consistency.txt`` which does not
    exist in the checkout (the file is at ``<pkg>/data/``): np.loadtxt is
    redirected for that one path.
  * Limber (cosmology.py:890-899) uses scipy.interpolate.interp2d and
    si.dfitpack.bispeu, both removed in scipy 1.15: a bilinear shim is injected
    for the Limber fixtures only.

Usage:  python tools/make_golden.py [--out tests/golden]
"""
import argparse
import json
import os
import sys
import types
import warnings

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)

from hmvec_amd.background import AnalyticBackground  # noqa: E402


def install_standin_camb():
    camb = types.ModuleType("camb")
    model = types.ModuleType("camb.model")
    model.NonLinear_none, model.NonLinear_both = 0, 3

    class Pars:
        pass

    def set_params(**kw):
        p = Pars()
        p.kw = dict(kw)
        p.YHe = kw.get("YHe")
        if p.YHe is None:
            p.YHe = 0.2454
        return p

    def get_background(p):
        k = p.kw
        return AnalyticBackground(H0=k["H0"], ombh2=k["ombh2"], omch2=k["omch2"],
                                  omk=k.get("omk", 0.0), w0=k.get("w", -1.0),
                                  wa=k.get("wa", 0.0), YHe=p.YHe)

    def get_matter_power_interpolator(pars, nonlinear=False, hubble_units=True, k_hunit=True, kmax=None,
                                      var1=None, var2=None, zmax=None, **kw):
        """What hmvec/cosmology.py:783-786 asks CAMB for with accuracy='medium'/'high': an object with
        .P(z, k, grid=True) in Mpc units.  Served from the table of tests/helpers/pk_table.py through this
        repo's TabulatedPowerInterpolator (a spline of ln P in (z, ln k), as CAMB's own interpolator is)."""
        assert not nonlinear and hubble_units is False and k_hunit is False and var1 == var2 == "delta_tot"
        sys.path.insert(0, os.path.join(REPO, "tests", "helpers"))
        from pk_table import table
        from hmvec_amd.background import TabulatedPowerInterpolator
        return TabulatedPowerInterpolator(*table(pars.kw["ns"]))

    camb.set_params = set_params
    camb.get_background = get_background
    camb.get_matter_power_interpolator = get_matter_power_interpolator
    camb.model = model
    sys.modules["camb"] = camb
    sys.modules["camb.model"] = model


def install_loadtxt_redirect():
    real = np.loadtxt
    bad = os.path.normpath(os.path.join(REF, "data", "alpha_consistency.txt"))
    good = os.path.join(REF, "hmvec", "data", "alpha_consistency.txt")

    def loadtxt(fname, *a, **k):
        if isinstance(fname, str) and os.path.normpath(fname) == bad:
            fname = good
        return real(fname, *a, **k)

    np.loadtxt = loadtxt


def install_limber_shims():
    """interp2d / dfitpack.bispeu were removed from scipy>=1.14; the reference's
    Limber code needs both.  Bilinear spline with the same (kx=ky=1) semantics."""
    import scipy.interpolate as si
    from scipy.interpolate import RectBivariateSpline

    class interp2d:  # noqa: N801
        def __init__(self, x, y, z, bounds_error=False):
            self._s = RectBivariateSpline(np.asarray(x), np.asarray(y), np.asarray(z).T,
                                          kx=1, ky=1, s=0)
            tx, ty, c = self._s.tck
            self.tck = (tx, ty, c, 1, 1)

    class _Dfit:
        @staticmethod
        def bispeu(tx, ty, c, kx, ky, x, y):
            from scipy.interpolate import _dfitpack
            return _dfitpack.bispeu(tx, ty, c, kx, ky, x, y)

    si.interp2d = interp2d
    si.dfitpack = _Dfit
    import hmvec.cosmology as rc
    rc.interp2d = interp2d


def cosmo_inputs(h, zs):
    """Everything the path consumes from the cosmology layer, as arrays."""
    d = dict(
        h=h.h, omm0=h.omm0, YHe=h.YHe,
        Pzk=h.Pzk, hubble_zs=h.hubble_parameter(h.zs), hubble_0=h.hubble_parameter(0.0),
        h_of_z_zs=h.h_of_z(h.zs), rho_crit_zs=h.rho_critical_z(h.zs),
        rho_crit_0=h.rho_critical_z(0.0), rho_matter_0=h.rho_matter_z(0),
        chi_zs=h.comoving_radial_distance(h.zs),
    )
    if hasattr(h, "sPzk"):
        d["sPzk"] = h.sPzk
    return d


def run_case(hm, tag, zs, ks, ms, *, params=None, mass_function="sheth-torman", mdef="vir",
             family="AGN", nxs=512, xmax=20.0, corr="max", ngal_mode=False,
             central=False, pres=True, numeric_nfw=None, batt_override=None, limber=None, second_tracers=False):
    params = dict(params or {})
    out = dict(zs=zs, ks=ks, ms=ms)
    meta = dict(params=params, mass_function=mass_function, mdef=mdef, family=family,
                nxs=nxs, xmax=xmax, corr=corr, ngal_mode=ngal_mode, central=central,
                pres=pres, numeric_nfw=numeric_nfw, batt_override=batt_override,
                limber=limber, second_tracers=second_tracers)
    h = hm.HaloModel(zs, ks, ms=ms, params=dict(params), mass_function=mass_function,
                     mdef=mdef, accuracy="low")
    for k, v in cosmo_inputs(h, zs).items():
        out["in_" + k] = np.asarray(v)
    out["sigma2"], out["nzm"], out["bh"] = h.sigma2, h.nzm, h.bh
    out["cs"] = h.concentration()
    out["rvir"] = h.rvir(ms[None, :], zs[:, None])
    out["uk_nfw"] = h.uk_profiles["nfw"]

    # mass conversion intermediates (hmvec.py:216-225)
    rhoc = h.rho_critical_z(zs)
    d1 = rhoc * h.deltav(zs) if mdef == "vir" else h.rho_matter_z(zs) * 200.0
    out["m200c"] = hm.mdelta_from_mdelta(ms, out["cs"], d1, 200.0 * rhoc)

    h.add_battaglia_profile("electron", family=family, xmax=xmax, nxs=nxs,
                            param_override=batt_override)
    out["uk_electron"] = h.uk_profiles["electron"]

    if numeric_nfw is not None:
        nn, xm = numeric_nfw
        _, u = h.add_nfw_profile("nfwnum", numeric=True, nxs=nn, xmax=xm)
        out["uk_nfwnum"] = u

    if ngal_mode:
        h.add_hod("g0", mthresh=10 ** 10.5 + zs * 0.0)
        target = h.hods["g0"]["ngal"] * 1.37
        h.add_hod("g", ngal=target, corr=corr,
                  central_profile_name="electron" if central else None)
        out["ngal_target"] = target
    else:
        h.add_hod("g", mthresh=10 ** 10.5 + zs * 0.0, corr=corr,
                  central_profile_name="electron" if central else None)
    for key in ("Nc", "Ns", "NsNsm1", "NcNs", "ngal", "bg", "log10mthresh"):
        out["hod_" + key] = np.asarray(h.hods["g"][key])

    names = ["nfw", "electron", "g"]
    if pres:
        h.add_battaglia_pres_profile("y", nxs=nxs, xmax=xmax)
        out["pk_y"] = h.pk_profiles["y"]
        names.append("y")
    for i, a in enumerate(names):
        for b in names[i:]:
            out[f"P1h_{a}_{b}"] = h.get_power_1halo(a, b)
            out[f"P2h_{a}_{b}"] = h.get_power_2halo(a, b)
    # argument-order quirk (hmvec.py:510-511): (g, g) only; cross order symmetric
    out["P1h_electron_nfw"] = h.get_power_1halo("electron", "nfw")
    b1 = 1.0 + 0.1 * np.arange(zs.size)
    b2 = 2.0 - 0.05 * np.arange(zs.size)
    out["b1_in"], out["b2_in"] = b1, b2
    out["P2h_g_nfw_bin"] = h.get_power_2halo("g", "nfw", b1_in=b1, b2_in=b2)
    out["P_tot_g_electron"] = h.get_power("g", "electron")

    if second_tracers:
        # two DIFFERENT HOD names / two different pressure names: the 1-halo term takes the square
        # term of the FIRST name only (hmvec/hmvec.py:510-513), so the result depends on the order
        h.add_hod("g2", mthresh=10 ** 11.0 + zs * 0.0, corr="min" if corr == "max" else "max")
        h.add_battaglia_pres_profile("y2", param_override=dict(P0_A0=25.0, xc_A0=0.6, battaglia_pres_gamma=-0.4),
                                     nxs=nxs, xmax=xmax)
        for a, b in (("g", "g2"), ("g2", "g"), ("y", "y2"), ("y2", "y")):
            out[f"P1h_{a}_{b}"] = h.get_power_1halo(a, b)
            out[f"P2h_{a}_{b}"] = h.get_power_2halo(a, b)

    if limber is not None:
        ells = np.asarray(limber["ells"], dtype=float)
        Pmm = out["P1h_nfw_nfw"] + out["P2h_nfw_nfw"]
        Pgm = out["P1h_nfw_g"] + out["P2h_nfw_g"]
        Pgg = out["P1h_g_g"] + out["P2h_g_g"]
        out["ells"] = ells
        out["C_kk"] = h.C_kk(ells, zs, ks, Pmm, lzs1=limber["lzs"], lzs2=limber["lzs"])
        out["C_kg"] = h.C_kg(ells, zs, ks, Pgm, gzs=limber["gzs"], lzs=limber["lzs"])
        gz = np.linspace(0.3, 1.2, 7)
        gd = np.exp(-0.5 * ((gz - 0.7) / 0.2) ** 2)
        out["gz_dndz"] = np.stack([gz, gd])
        out["C_kg_dndz"] = h.C_kg(ells, zs, ks, Pgm, gzs=gz, gdndz=gd, lzs=limber["lzs"])
        out["C_gg_dndz"] = h.C_gg(ells, zs, ks, Pgg, gzs=gz, gdndz=gd)
        out["lensing_window"] = h.lensing_window(zs, limber["lzs"])
        if pres:      # tSZ projections (hmvec/cosmology.py:585-597)
            Pyy = out["P1h_y_y"] + out["P2h_y_y"]
            Pym = out["P1h_nfw_y"] + out["P2h_nfw_y"]
            out["C_yy"] = h.C_yy(ells, zs, ks, Pyy)
            out["C_ky"] = h.C_ky(ells, zs, ks, Pym, lzs1=limber["lzs"])
    out["meta_json"] = np.array(json.dumps(meta, default=lambda o: np.asarray(o).tolist()))
    return out


def run_readme_anchor(hm):
    """Config 1/2 (README grid) at full size; only a strided sub-sample is stored."""
    zs = np.linspace(0.0, 3.0, 20)
    ms = np.geomspace(2e10, 1e17, 200)
    ks = np.geomspace(1e-4, 100, 1001)
    h = hm.HaloModel(zs, ks, ms=ms, accuracy="low")
    h.add_battaglia_profile("electron", family="AGN", xmax=20, nxs=5000)
    h.add_hod("g", mthresh=10 ** 10.5 + zs * 0.0)
    zi, mi, ki = slice(None, None, 3), slice(None, None, 13), slice(None, None, 20)
    out = dict(zs=zs, ms=ms, ks=ks, zstride=3, mstride=13, kstride=20)
    out["sigma2"] = h.sigma2[zi, mi]
    out["nzm"] = h.nzm[zi, mi]
    out["bh"] = h.bh[zi, mi]
    out["uk_nfw"] = h.uk_profiles["nfw"][zi, mi, ki]
    out["uk_electron"] = h.uk_profiles["electron"][zi, mi, ki]
    for k in ("ngal", "bg"):
        out["hod_" + k] = h.hods["g"][k]
    names = ["nfw", "electron", "g"]
    for i, a in enumerate(names):
        for b in names[i:]:
            out[f"P1h_{a}_{b}"] = h.get_power_1halo(a, b)[:, ki]
            out[f"P2h_{a}_{b}"] = h.get_power_2halo(a, b)[:, ki]
    out["meta_json"] = np.array(json.dumps(dict(config="README C1/C2", nxs=5000, xmax=20)))
    return out


def run_unit_pins(hm):
    """Known-answer pins the reference itself carries."""
    import hmvec.fft as rfft
    import hmvec.utils as rutils
    import hmvec.tinker as rtinker
    out = {}
    # fft_integral on the authors' test input (bin/tests.py:8-11): dx=1e-3, x<20
    x = np.arange(1e-3, 20.0, 1e-3)
    kt, ukt = rfft.fft_integral(x, np.exp(-x ** 2 / 2.0))
    out["fftint_x"] = x
    out["fftint_k"], out["fftint_u"] = kt[:400], ukt[:400]
    # bisection self-test (hmvec/utils.py:45-51)
    xs = np.array([2.0, 4.0, 6.0])
    out["bisect_x"] = xs
    out["bisect_y"] = rutils.vectorized_bisection_search(xs, lambda y: np.sqrt(y), (1, 40),
                                                         "increasing", rtol=1e-4, verbose=False)
    nu = np.geomspace(0.2, 6.0, 40)[None, :] + np.zeros((5, 1))
    z = np.array([0.0, 0.7, 2.2, 3.0, 4.5])[:, None]
    out["tinker_nu"], out["tinker_z"] = nu, z
    out["tinker_bias"] = rtinker.bias(nu)
    out["tinker_fnu"] = rtinker.f_nu(nu, z)
    izs, ial = np.loadtxt(os.path.join(REF, "data", "alpha_consistency.txt"), unpack=True)
    out["tinker_alpha_z"], out["tinker_alpha"] = izs, ial
    # HOD helper functions on a grid
    zz = np.array([0.0, 0.8, 0.80001, 2.0])[:, None]
    lm = np.linspace(8.0, 13.0, 30)[None, :]
    out["shmr_z"], out["shmr_logmstar"] = zz, lm
    out["shmr_Mhalo_stellar"] = hm.Mhalo_stellar(zz, lm)
    lmh = np.linspace(10.0, 16.0, 25)[None, :]
    out["shmr_logmhalo"] = lmh
    out["shmr_Mstellar_halo"] = hm.Mstellar_halo(zz, lmh)
    return out


def run_function_pins(hm):
    """Inputs and outputs of the reference's module-level helpers on the path (SURVEY 8a rows
    A5, A7, A8, F1, F2, H1-H3, X1), on small grids that exercise their broadcasting."""
    import hmvec.fft as rfft
    out = {}
    h = 0.673
    rho_c0 = 2.77536627e11 * h ** 2
    z1 = np.array([0.0, 0.6, 1.3, 2.9])
    E2 = 0.315 * (1 + z1) ** 3 + 0.685
    rhoc = rho_c0 * E2
    rhom = rho_c0 * 0.315 * (1 + z1) ** 3
    ms = np.geomspace(1e11, 3e15, 12)
    out["z"], out["ms"], out["rhoc"], out["rhom"] = z1, ms, rhoc, rhom
    cs = hm.duffy_concentration(ms[None, :], z1[:, None])
    out["duffy_default"] = cs
    out["duffy_vir"] = hm.duffy_concentration(ms[None, :], z1[:, None], 7.85, -0.081, -0.71, 0.7)
    out["R_from_M"] = hm.R_from_M(ms[None, :], rhoc[:, None], delta=200.0)
    out["Fcon"] = hm.Fcon(cs)
    r = np.geomspace(1e-3, 5.0, 17)
    out["r"] = r
    out["rho_nfw"] = hm.rho_nfw(r, 3.3e14, 0.31)
    out["rho_nfw_x"] = hm.rho_nfw_x(r, 2.0)
    out["a2z"] = hm.a2z(np.array([1.0, 0.5, 0.25]))
    out["mdelta"] = hm.mdelta_from_mdelta(ms, cs, 200.0 * rhom, 200.0 * rhoc)
    out["mdelta_unvec"] = hm.mdelta_from_mdelta(ms, cs, 200.0 * rhom, 200.0 * rhoc, vectorized=False)
    # HOD helpers with the shapes add_hod uses
    lmh = np.log10(ms)[None, :]
    zc = z1[:, None]
    thr = np.array([10.2, 10.5, 10.9, 11.3])[:, None]
    out["hod_thr"] = thr
    Nc = hm.avg_Nc(lmh, zc, thr, 0.2)
    Ns = hm.avg_Ns(lmh, zc, thr, Nc, 0.2, 1.0, 9.04, 0.74, 1.65, 0.59)
    out["avg_Nc"], out["avg_Ns"] = Nc, Ns
    out["avg_Ns_noNc"] = hm.avg_Ns(lmh, zc, thr, None, 0.2, 1.1, 9.0, 0.7, 1.6, 0.6)
    out["hod_mfunc"] = hm.hod_default_mfunc(hm.Mhalo_stellar(zc, thr), 9.04, 0.74)
    for corr in ("max", "min"):
        out[f"NsNsm1_{corr}"] = hm.avg_NsNsm1(Nc, Ns, corr)
        out[f"NcNs_{corr}"] = hm.avg_NcNs(Nc, Ns, corr)
    nzm = 1e-3 * (ms[None, :] / 1e13) ** -1.9 * np.exp(-ms[None, :] / 1e15) / ms[None, :] * (1 + zc) ** -0.5
    out["nzm"] = nzm
    out["ngal_from_NcNs"] = hm.ngal_from_mthresh(nzm=nzm, ms=ms, Ncs=Nc, Nss=Ns)
    out["ngal_from_thr"] = hm.ngal_from_mthresh(thr[:, 0], z1, nzm, ms, 0.2, alphasat=1.0, Bsat=9.04, betasat=0.74,
                                                Bcut=1.65, betacut=0.59)
    # Battaglia profiles with the (nz,nm,nxs) broadcasting add_battaglia_profile uses
    omb, omm = 0.049, 0.315
    x = np.linspace(0.0, 8.0, 9)[1:]
    out["x"] = x
    m3, z3, rc3 = ms[None, :, None], z1[:, None, None], rhoc[:, None, None]
    out["batt_fit"] = hm.battaglia_gas_fit(m3, z3, 4000.0, 0.29, -0.66)
    out["rho_gas_generic_x"] = hm.rho_gas_generic_x(x[None, None], m3, z3, omb, omm, rc3)
    sh = hm.battaglia_defaults["SH"]
    out["rho_gas_generic_x_SH"] = hm.rho_gas_generic_x(x[None, None], m3, z3, omb, omm, rc3, gamma=-0.25, **sh)
    out["rho_gas_generic"] = hm.rho_gas_generic(r[None, None], m3, z3, omb, omm, rc3)
    out["rho_gas_AGN"] = hm.rho_gas(r, 1e13, 1.0, omb, omm, rhoc[1], profile="AGN")
    out["rho_gas_SH"] = hm.rho_gas(r, 1e13, 1.0, omb, omm, rhoc[1], profile="SH")
    r200 = hm.R_from_M(m3, rc3, delta=200.0)
    out["P_e_generic_x"] = hm.P_e_generic_x(x[None, None], m3, r200, z3, omb, omm, rc3)
    out["P_e_generic"] = hm.P_e_generic(r[None, None], m3, z3, omb, omm, rc3, alpha=1.1, gamma=-0.35)
    out["P_e"] = hm.P_e(r, 2e14, 0.5, omb, omm, rhoc[1])
    # generic_profile_fft with user callables: shared 1-D profile, per-(z,m) 3-D profile, no mass norm
    ks = np.geomspace(1e-3, 30.0, 21)
    out["ks"] = ks
    cmax = cs
    rss = (hm.R_from_M(ms[None, :], rhoc[:, None], delta=200.0) / cs)[..., None]
    out["gpf_cmax"], out["gpf_rss"] = cmax, rss
    _, u1 = rfft.generic_profile_fft(lambda xx: 1.0 / xx / (1.0 + xx) ** 2, cmax, rss, z1, ks, 60.0, 3000)
    out["gpf_shared"] = u1
    slope = (2.0 + 0.1 * np.arange(ms.size))[None, :, None] + 0.05 * z1[:, None, None]
    out["gpf_slope"] = slope
    _, u3 = rfft.generic_profile_fft(lambda xx: xx ** -0.5 * (1.0 + xx) ** -slope, cmax, rss, z1, ks, 30.0, 1001)
    out["gpf_rows_odd_nxs"] = u3
    _, u4 = rfft.generic_profile_fft(lambda xx: np.exp(-xx) + 0 * slope, 0 * cmax + 4.0, rss, z1, ks, 12.0, 640,
                                     do_mass_norm=False)
    out["gpf_nonorm"] = u4
    # fft_integral with a 2-D integrand
    xx = np.arange(2e-3, 9.0, 2e-3)
    yy = np.exp(-xx[None, :] ** 2 / np.array([2.0, 1.0, 0.5])[:, None])
    kt, ukt = rfft.fft_integral(xx, yy)
    out["fi_x"], out["fi_y"], out["fi_k"], out["fi_u"] = xx, yy, kt[:300], ukt[:, :300]
    # uk_fft on an NFW profile (bin/tests.py:45)
    kf, uf = rfft.uk_fft(lambda rr: 1.0 / (rr / 0.2) / (1.0 + rr / 0.2) ** 2, 1.5, dr=0.01, rmax=40)
    out["ukfft_k"], out["ukfft_u"] = kf[1:200], uf[1:200]
    return out


def run_case_e(hm):
    """Row N3: the UNMODIFIED reference with accuracy='medium' and 'high' (hmvec/hmvec.py:96-102,
    hmvec/cosmology.py:255-260,353-389,772-786) on the tabulated P(k,z) the stand-in camb serves - the host
    seam (P_lin, P_lin_slow, _get_matter_power) and everything downstream of it."""
    zs = np.array([0.05, 0.6, 1.4, 2.2, 3.0])
    ms = np.geomspace(1e10, 1e16, 36)
    ks = np.geomspace(1e-4, 80.0, 45)
    out = dict(zs=zs, ms=ms, ks=ks)
    for acc in ("medium", "high"):
        for mf, tag in (("sheth-torman", "st"), ("tinker", "tk")):
            h = hm.HaloModel(zs, ks, ms=ms, mass_function=mf, accuracy=acc)
            pre = f"{acc}_{tag}_"
            if "in_h" not in out:
                for k, v in cosmo_inputs(h, zs).items():
                    if k not in ("Pzk", "sPzk"):
                        out["in_" + k] = np.asarray(v)
            if tag == "st":        # the spectra the path consumes do not depend on the mass function
                out[f"{acc}_Pzk"], out[f"{acc}_sPzk"] = h.Pzk, h.sPzk
            out[pre + "sigma2"], out[pre + "nzm"], out[pre + "bh"] = h.sigma2, h.nzm, h.bh
            h.add_battaglia_profile("electron", family="AGN", xmax=20, nxs=400)
            h.add_hod("g", mthresh=10 ** 10.5 + zs * 0.0)
            for a, b in (("nfw", "nfw"), ("electron", "electron"), ("g", "g")):
                out[pre + f"P_{a}_{b}"] = h.get_power(a, b)
    out["meta_json"] = np.array(json.dumps(dict(nxs=400, xmax=20, table="tests/helpers/pk_table.py")))
    return out


def run_case_f(hm):
    """The radial grids the reference's own callers use (too long for one LDS row: the pruned long-grid route):
    add_battaglia_profile(xmax=50, nxs=30000) as in examples/lensing_baryons.py:27 and bin/tests.py:308, the numeric
    NFW branch at its defaults nxs=40000 / xmax=200 (hmvec/params.py:59-60), and the tSZ notebook's
    add_battaglia_pres_profile(xmax=2, nxs=30000) (support not short: the rocFFT route).  Tiny (z, m) grid."""
    zs = np.array([0.3, 1.1, 2.4])
    ms = np.geomspace(2e10, 1e17, 10)
    ks = np.geomspace(1e-4, 100, 60)
    out = dict(zs=zs, ks=ks, ms=ms)
    h = hm.HaloModel(zs, ks, ms=ms, accuracy="low")
    for k, v in cosmo_inputs(h, zs).items():
        out["in_" + k] = np.asarray(v)
    out["cs"] = h.concentration()
    out["uk_nfw"] = h.uk_profiles["nfw"]
    h.add_battaglia_profile("electron", family="AGN", xmax=50, nxs=30000)
    out["uk_electron"] = h.uk_profiles["electron"]
    _, u = h.add_nfw_profile("nfwnum", numeric=True)
    out["uk_nfwnum"] = u
    h.add_battaglia_pres_profile("y", xmax=2, nxs=30000)
    out["pk_y"] = h.pk_profiles["y"]
    h.add_hod("g", mthresh=10 ** 10.5 + zs * 0.0, satellite_profile_name="nfwnum")
    for a, b in (("electron", "electron"), ("nfw", "electron"), ("nfwnum", "nfwnum"), ("g", "electron"), ("y", "y"),
                 ("nfw", "y")):
        out[f"P1h_{a}_{b}"] = h.get_power_1halo(a, b)
        out[f"P2h_{a}_{b}"] = h.get_power_2halo(a, b)
    out["meta_json"] = np.array(json.dumps(dict(
        electron=dict(family="AGN", xmax=50, nxs=30000), nfwnum=dict(nxs=40000, xmax=200),
        y=dict(xmax=2, nxs=30000), hod=dict(mthresh="10**10.5", satellite="nfwnum"))))
    return out


def run_extra_pins(hm):
    """Helpers off the grid path that the reference exports and its own scripts use: Cosmology.sigma_crit
    (hmvec/hmvec.py:595), Cosmology.bias_fnl (examples/fnl.py) and fft.uk_brute_force (bin/tests.py:36)."""
    import hmvec.fft as rfft
    out = {}
    zs = np.linspace(0.1, 1.5, 6)
    ks = np.geomspace(1e-3, 5.0, 40)
    h = hm.HaloModel(zs, ks, ms=np.geomspace(1e11, 1e15, 8), accuracy="low", skip_nfw=True)
    out["zs"], out["ks"] = zs, ks
    out["sigma_crit"] = h.sigma_crit(zs, 2.0)
    for z in (0.0, 0.8):
        out[f"bias_fnl_z{z}"] = h.bias_fnl(1.8, 25.0, z, ks)
    out["bias_fnl_deltac"] = h.bias_fnl(2.4, -10.0, 0.5, ks, deltac=1.686)
    r = np.arange(0.01, 4.0, 0.01)
    rho = 1.0 / (r / 0.2) / (1.0 + r / 0.2) ** 2
    kb = np.geomspace(0.05, 40.0, 25)
    out["ub_r"], out["ub_rho"], out["ub_k"] = r, rho, kb
    out["ub_u"] = rfft.uk_brute_force(r, rho, 1.5, kb)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    ap.add_argument("--skip-readme", action="store_true")
    ap.add_argument("--only", default=None, help="comma-separated fixture names to (re)generate")
    args = ap.parse_args()
    if not os.path.isdir(REF):
        sys.exit("reference checkout not present; goldens can only be generated in the build container")
    warnings.filterwarnings("ignore")
    install_standin_camb()
    install_loadtxt_redirect()
    sys.path.insert(0, REF)
    import hmvec as hm  # the unmodified reference
    install_limber_shims()
    os.makedirs(args.out, exist_ok=True)

    only = set(args.only.split(",")) if args.only else None

    def want(name):
        return only is None or name in only

    def save(name, d):
        path = os.path.join(args.out, name + ".npz")
        np.savez_compressed(path, **d)
        print(f"wrote {path}  ({os.path.getsize(path)/1024:.0f} KiB)")

    if want("unit_pins"):
        save("unit_pins", run_unit_pins(hm))
    if want("func_pins"):
        save("func_pins", run_function_pins(hm))

    ks = np.geomspace(1e-4, 100, 40)
    ms = np.geomspace(2e10, 1e17, 32)
    # A: defaults (ST, vir, AGN, corr=max, mthresh); z grid has z=0, both SHMR branches, z=3.0
    if want("case_a"):
        save("case_a", run_case(hm, "a", np.array([0.0, 0.5, 1.2, 3.0]), ks, ms, nxs=1000,
                                numeric_nfw=(4000, 50.0),
                                limber=None))
    # B: tinker + mean + SH + corr=min + ngal bisection + miscentred central + odd sigma2_numks
    if want("case_b"):
        save("case_b", run_case(hm, "b", np.array([0.1, 0.8, 1.7, 3.0, 3.4]),
                                np.geomspace(2e-4, 50, 33), np.geomspace(1e11, 5e15, 24),
                                params=dict(sigma2_numks=2001, omch2=0.125, H0=70.0, ns=0.97),
                                mass_function="tinker", mdef="mean", family="SH", nxs=600,
                                xmax=15.0, corr="min", ngal_mode=True, central=True,
                                batt_override=dict(battaglia_gas_gamma=-0.25, rho0_A0=4100.0)))
    # C: Limber fixtures (no z=0: chi=0 makes the reference NaN), ST/vir
    if want("case_c"):
        save("case_c", run_case(hm, "c", np.linspace(0.05, 3.0, 9), np.geomspace(1e-4, 100, 48),
                                np.geomspace(2e10, 1e17, 28), nxs=400, pres=False,
                                params=dict(sigma2_numks=4000),
                                limber=dict(ells=np.linspace(100, 6000, 12), lzs=2.5, gzs=0.8)))
    # D: pressure + Limber (C_yy, C_ky) and the first-name-only rule for two HODs / two pressure profiles
    if want("case_d"):
        save("case_d", run_case(hm, "d", np.linspace(0.1, 2.8, 7), np.geomspace(1e-4, 100, 44),
                                np.geomspace(2e10, 1e17, 26), nxs=400, pres=True, second_tracers=True,
                                params=dict(sigma2_numks=3000),
                                limber=dict(ells=np.linspace(100, 6000, 10), lzs=2.5, gzs=0.8)))
    # E: accuracy='medium' / 'high' on a tabulated, non-separable P(k,z) (row N3's host seam)
    if want("case_e"):
        save("case_e", run_case_e(hm))
    # F: the long radial grids of the reference's own callers (nxs = 30000 / 40000)
    if want("case_f"):
        save("case_f", run_case_f(hm))
    if want("extra_pins"):
        save("extra_pins", run_extra_pins(hm))
    if not args.skip_readme:
        if want("readme_c1"):
            save("readme_c1", run_readme_anchor(hm))


if __name__ == "__main__":
    main()
