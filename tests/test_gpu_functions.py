"""GPU mirrors of the reference's free functions (hmvec_amd.functions / .fft / .tinker / .utils)
against outputs of the unmodified reference (tests/golden/func_pins.npz, unit_pins.npz), plus the
reference's own self-checks that use them (bin/tests.py: test_fft_integral, test_battaglia,
test_mcon; hmvec/utils.py:45-51)."""
import numpy as np
import pytest

from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu

G = load_golden("func_pins")
U = load_golden("unit_pins")
Z, MS, RHOC, RHOM = G["z"], G["ms"], G["rhoc"], G["rhom"]
OMB, OMM = 0.049, 0.315
TOL = 2e-13          # generic pow/exp/log on two different libms


def test_package_exports_reference_names():
    import hmvec_amd as hm
    for name in ("HaloModel", "duffy_concentration", "R_from_M", "Mstellar_halo", "Mhalo_stellar", "Mhalo_stellar_core",
                 "avg_Nc", "avg_Ns",
                 "avg_NsNsm1", "avg_NcNs", "hod_default_mfunc", "Fcon", "rhoscale_nfw", "rho_nfw_x", "rho_nfw",
                 "mdelta_from_mdelta", "mdelta_from_mdelta_unvectorized", "battaglia_gas_fit", "rho_gas",
                 "rho_gas_generic", "rho_gas_generic_x", "P_e", "P_e_generic", "P_e_generic_x", "a2z",
                 "ngal_from_mthresh", "default_params", "battaglia_defaults", "generic_profile_fft", "Cosmology"):
        assert hasattr(hm, name), name
    for mod, names in (("fft", ("fft_integral", "generic_profile_fft", "uk_fft", "analytic_fft_integral")),
                       ("tinker", ("bias", "f_nu")), ("utils", ("vectorized_bisection_search",))):
        for n in names:
            assert hasattr(getattr(hm, mod), n), (mod, n)
    with pytest.raises(NameError):       # the reference's rhoscale_nfw reads an undefined global
        hm.rhoscale_nfw(1e14, 1.0, 5.0)


def test_halo_structure_functions():
    import hmvec_amd as hm
    cs = hm.duffy_concentration(MS[None, :], Z[:, None])
    assert cs.shape == (Z.size, MS.size) and rel_err(cs, G["duffy_default"]) < TOL
    assert rel_err(hm.duffy_concentration(MS[None, :], Z[:, None], 7.85, -0.081, -0.71, 0.7), G["duffy_vir"]) < TOL
    assert rel_err(hm.R_from_M(MS[None, :], RHOC[:, None], delta=200.0), G["R_from_M"]) < TOL
    assert rel_err(hm.Fcon(G["duffy_default"]), G["Fcon"]) < TOL
    assert rel_err(hm.rho_nfw(G["r"], 3.3e14, 0.31), G["rho_nfw"]) < TOL
    assert rel_err(hm.rho_nfw_x(G["r"], 2.0), G["rho_nfw_x"]) < TOL
    assert rel_err(hm.a2z(np.array([1.0, 0.5, 0.25])), G["a2z"]) < TOL
    assert isinstance(hm.Fcon(5.0), float) and abs(hm.Fcon(5.0) - (np.log(6.0) - 5.0 / 6.0)) < 1e-15   # scalars stay scalars


def test_mass_conversion():
    """mdelta_from_mdelta: the reference stops its secant at 1.5e-8 in ln M2 (scipy newton default tol), so
    its own output is only that close to the root; the device solve is exact to rounding."""
    import hmvec_amd as hm
    cs = G["duffy_default"]
    got = hm.mdelta_from_mdelta(MS, cs, 200.0 * RHOM, 200.0 * RHOC)
    assert got.shape == cs.shape and rel_err(got, G["mdelta"]) < 1e-7
    # residual of the defining equation M1 F(c1) = M2 F(c2) at the returned root
    F = lambda c: 1.0 / (np.log(1 + c) - c / (1 + c))    # noqa: E731
    c2 = cs * ((got / MS[None, :]) * (RHOM / RHOC)[:, None]) ** (1.0 / 3.0)
    assert np.max(np.abs(MS[None, :] * F(cs) / (got * F(c2)) - 1.0)) < 1e-14
    # bin/tests.py test_mcon: vectorised and unvectorised agree
    un = hm.mdelta_from_mdelta(MS, cs, 200.0 * RHOM, 200.0 * RHOC, vectorized=False)
    assert np.array_equal(un, got)
    el = hm.mdelta_from_mdelta_unvectorized(MS[None, :] + cs * 0, cs, 200.0 * RHOM[:, None], 200.0 * RHOC[:, None])
    assert np.array_equal(el, got)


def test_hod_functions():
    import hmvec_amd as hm
    lmh, zc, thr = np.log10(MS)[None, :], Z[:, None], G["hod_thr"]
    assert rel_err(hm.Mhalo_stellar(U["shmr_z"], U["shmr_logmstar"]), U["shmr_Mhalo_stellar"]) < TOL
    assert np.max(np.abs(hm.Mstellar_halo(U["shmr_z"], U["shmr_logmhalo"]) - U["shmr_Mstellar_halo"])) < 1e-12
    # the explicit-parameter core (hmvec/hmvec.py:648-657) with the two table-2 sets reproduces the same golden rows
    zz = np.asarray(U["shmr_z"]).reshape(-1, 1)
    lo = (10.72, 0.55, 12.35, 0.28, 0.44, 0.18, 1.56, 2.51, 0.57, 0.17)
    hi = (11.09, 0.56, 12.27, -0.84, 0.65, 0.31, 1.12, -0.53, 0.56, -0.12)
    core = np.where(zz <= 0.8, hm.Mhalo_stellar_core(U["shmr_logmstar"], 1.0 / (1.0 + zz), *lo),
                    hm.Mhalo_stellar_core(U["shmr_logmstar"], 1.0 / (1.0 + zz), *hi))
    assert rel_err(core, U["shmr_Mhalo_stellar"]) < TOL
    Nc = hm.avg_Nc(lmh, zc, thr, 0.2)
    assert np.allclose(Nc, G["avg_Nc"], rtol=1e-11, atol=1e-300)
    Ns = hm.avg_Ns(lmh, zc, thr, G["avg_Nc"], 0.2, 1.0, 9.04, 0.74, 1.65, 0.59)
    assert np.allclose(Ns, G["avg_Ns"], rtol=1e-12, atol=0)
    assert np.allclose(hm.avg_Ns(lmh, zc, thr, None, 0.2, 1.1, 9.0, 0.7, 1.6, 0.6), G["avg_Ns_noNc"], rtol=1e-11, atol=1e-300)
    assert rel_err(hm.hod_default_mfunc(hm.Mhalo_stellar(zc, thr), 9.04, 0.74), G["hod_mfunc"]) < 1e-12
    for corr in ("max", "min"):
        assert np.allclose(hm.avg_NsNsm1(G["avg_Nc"], G["avg_Ns"], corr), G[f"NsNsm1_{corr}"], rtol=1e-14, atol=0)
        assert np.allclose(hm.avg_NcNs(G["avg_Nc"], G["avg_Ns"], corr), G[f"NcNs_{corr}"], rtol=1e-14, atol=0)
    assert hm.avg_NcNs(G["avg_Nc"], G["avg_Ns"], "other") is None
    ng = hm.ngal_from_mthresh(nzm=G["nzm"], ms=MS, Ncs=G["avg_Nc"], Nss=G["avg_Ns"])
    assert rel_err(ng, G["ngal_from_NcNs"]) < 1e-13
    ng = hm.ngal_from_mthresh(thr[:, 0], Z, G["nzm"], MS, 0.2, alphasat=1.0, Bsat=9.04, betasat=0.74, Bcut=1.65, betacut=0.59)
    assert rel_err(ng, G["ngal_from_thr"]) < 1e-11
    with pytest.raises(AssertionError):
        hm.ngal_from_mthresh(thr[:, 0], nzm=G["nzm"], ms=MS, Ncs=G["avg_Nc"], Nss=G["avg_Ns"])


def test_battaglia_functions():
    import hmvec_amd as hm
    m3, z3, rc3 = MS[None, :, None], Z[:, None, None], RHOC[:, None, None]
    x, r = G["x"], G["r"]
    assert rel_err(hm.battaglia_gas_fit(m3, z3, 4000.0, 0.29, -0.66), G["batt_fit"]) < TOL
    got = hm.rho_gas_generic_x(x[None, None], m3, z3, OMB, OMM, rc3)
    assert got.shape == (Z.size, MS.size, x.size) and rel_err(got, G["rho_gas_generic_x"]) < 1e-12
    sh = hm.battaglia_defaults["SH"]
    assert rel_err(hm.rho_gas_generic_x(x[None, None], m3, z3, OMB, OMM, rc3, gamma=-0.25, **sh), G["rho_gas_generic_x_SH"]) < 1e-12
    assert rel_err(hm.rho_gas_generic(r[None, None], m3, z3, OMB, OMM, rc3), G["rho_gas_generic"]) < 1e-12
    assert rel_err(hm.rho_gas(r, 1e13, 1.0, OMB, OMM, RHOC[1], profile="AGN"), G["rho_gas_AGN"]) < 1e-12
    assert rel_err(hm.rho_gas(r, 1e13, 1.0, OMB, OMM, RHOC[1], profile="SH"), G["rho_gas_SH"]) < 1e-12
    r200 = hm.R_from_M(m3, rc3, delta=200.0)
    assert rel_err(hm.P_e_generic_x(x[None, None], m3, r200, z3, OMB, OMM, rc3), G["P_e_generic_x"]) < 1e-12
    assert rel_err(hm.P_e_generic(r[None, None], m3, z3, OMB, OMM, rc3, alpha=1.1, gamma=-0.35), G["P_e_generic"]) < 1e-12
    assert rel_err(hm.P_e(r, 2e14, 0.5, OMB, OMM, RHOC[1]), G["P_e"]) < 1e-12
    with pytest.raises(TypeError):
        hm.rho_gas_generic_x(x, 1e13, 1.0, OMB, OMM, RHOC[1], not_a_parameter=1.0)


def test_battaglia_gas_mass_closure():
    """bin/tests.py test_battaglia: the AGN gas profile integrated to R200c holds about the cosmic
    baryon fraction of M200c (the reference prints this ratio)."""
    import hmvec_amd as hm
    r = np.geomspace(1e-4, 20.0, 10000)
    rhocz = RHOC[1]
    rhos = hm.rho_gas(r, 1e13, 1.0, OMB, OMM, rhocz, profile="AGN")
    r200 = hm.R_from_M(1e13, rhocz, delta=200)
    integrand = rhos * 4.0 * np.pi * r ** 2
    integrand[r > r200] = 0
    ratio = hm.functions.trapz_lastaxis(integrand, r) / (1e13 * OMB / OMM)
    assert 0.3 < ratio < 1.2


def test_tinker_functions():
    import hmvec_amd as hm
    assert rel_err(hm.tinker.bias(U["tinker_nu"]), U["tinker_bias"]) < TOL
    assert rel_err(hm.tinker.f_nu(U["tinker_nu"], U["tinker_z"]), U["tinker_fnu"]) < 1e-12
    with pytest.raises(ValueError):
        hm.tinker.f_nu(U["tinker_nu"], U["tinker_z"] - 1.0)      # below the alpha(z) table (bounds_error=True)
    raw = hm.tinker.f_nu(U["tinker_nu"], U["tinker_z"], norm_consistency=False)
    assert np.all(np.isfinite(raw)) and raw.shape == U["tinker_nu"].shape


def test_fft_integral_pins():
    import hmvec_amd as hm
    kt, u = hm.fft.fft_integral(U["fftint_x"], np.exp(-U["fftint_x"] ** 2 / 2.0))
    assert np.array_equal(kt[:400], U["fftint_k"])
    assert np.max(np.abs(u[:400] - U["fftint_u"])) < 1e-14
    # the authors' known-answer check (bin/tests.py:8-18): sqrt(pi/2) k exp(-k^2/2) to the quirk level
    sel = (kt > 0.5) & (kt < 3)
    assert np.max(np.abs(u[sel] / hm.fft.analytic_fft_integral(kt[sel]) - 1)) < 2e-2
    kt, u = hm.fft.fft_integral(G["fi_x"], G["fi_y"])
    assert u.shape == (3, G["fi_x"].size // 2 + 1)
    assert np.array_equal(kt[:300], G["fi_k"]) and np.max(np.abs(u[:, :300] - G["fi_u"])) < 1e-14


def test_generic_profile_fft_with_user_callables():
    import hmvec_amd as hm
    cmax, rss, ks = G["gpf_cmax"], G["gpf_rss"], G["ks"]
    k, u = hm.generic_profile_fft(lambda xx: 1.0 / xx / (1.0 + xx) ** 2, cmax, rss, Z, ks, 60.0, 3000)
    assert k is ks or np.array_equal(k, ks)
    assert np.max(np.abs(u - G["gpf_shared"])) < 1e-12
    slope = G["gpf_slope"]
    _, u = hm.fft.generic_profile_fft(lambda xx: xx ** -0.5 * (1.0 + xx) ** -slope, cmax, rss, Z, ks, 30.0, 1001)
    assert np.max(np.abs(u - G["gpf_rows_odd_nxs"])) < 1e-12
    _, u = hm.fft.generic_profile_fft(lambda xx: np.exp(-xx) + 0 * slope, 0 * cmax + 4.0, rss, Z, ks, 12.0, 640,
                                      do_mass_norm=False)
    assert rel_err(u, G["gpf_nonorm"]) < 1e-11
    with pytest.raises(AssertionError):      # rhos.ndim must be 1 or 3 (hmvec/fft.py:75-78)
        hm.generic_profile_fft(lambda xx: xx[None, :] + 0 * cmax[:, :1], cmax, rss, Z, ks, 10.0, 64)


@pytest.mark.parametrize("nxs,xmax", [(5000, 20.0), (3000, 20.0), (4000, 6.0), (30000, 50.0), (40000, 200.0)])
def test_table_route_runs_the_row_kernels_of_the_builtin_families(nxs, xmax, monkeypatch):
    """generic_profile_fft with a user's callable (hmvec/fft.py:56-94) takes the in-LDS row kernels too (round 5): the
    table row stands where the family's integrand is evaluated, everything behind it is the same code.  (a) Bit
    equality with the built-in route on a profile both evaluate to the same bits: gamma = 0, exponent 0, amplitude 1 is
    the constant 1 exactly (exp(0) = 1), and so is a table of ones - one-row kernels (compile-time and run-time plans)
    and the long-grid kernel (decomposition and chirp rows).  (b) A callable with structure against the table -> rocFFT
    chain the route replaces (HMG_FUSED_FFT=0), to the 1e-12 gate on u."""
    import ctypes as C
    import hmvec_amd as hm
    from hmvec_amd import _native as nat
    for sw in ("HMG_FUSED_FFT", "HMG_PRUNED_FFT", "HMG_FUSED_GENERIC"):     # (the suite may run under a route switch)
        monkeypatch.delenv(sw, raising=False)
    nz, nm, nk = 2, 12, 150
    xs = np.linspace(0.0, xmax, nxs + 1)[1:]
    step = (xs[-1] - xs[0]) / nxs
    kts = np.fft.rfftfreq(nxs, step) * 2 * np.pi
    cmax = np.linspace(0.9, 2.9, nz * nm).reshape(nz, nm)
    rss = np.geomspace(0.03, 2.5, nz * nm).reshape(nz, nm)
    zs, ks = np.array([0.2, 1.7]), np.geomspace(1e-3, 80, nk)
    ctx = nat.Context(0)
    d_xs, d_kts, d_cmax, d_rss, d_zs, d_ks = (ctx.upload(a) for a in (xs, kts, cmax, rss, zs, ks))
    out_f, out_t = ctx.empty((nz, nm, nk)), ctx.empty((nz, nm, nk))
    ctx.call("hmg_profile_fft", nz, nm, nk, nxs, step, d_xs.ptr, d_kts.ptr, None, None, None, None,
             1.0, 1.0, 1.0, 0.0, 0.0, d_cmax.ptr, d_rss.ptr, d_zs.ptr, d_ks.ptr, 1, None, out_f.ptr, None, None, None)
    d_ones = ctx.upload(np.ones(nxs))
    ctx.call("hmg_profile_fft_table", nz, nm, nk, nxs, step, d_xs.ptr, d_kts.ptr, d_ones.ptr, 1, d_cmax.ptr, d_rss.ptr,
             d_zs.ptr, d_ks.ptr, 1, out_t.ptr)
    a, b = out_f.numpy(), out_t.numpy()
    assert np.all(np.isfinite(a)) and np.array_equal(a, b)
    # per-row tables, a profile with structure
    rho = (xs[None, None, :] / 0.7) ** -0.4 * (1.0 + (xs[None, None, :] / 0.7) ** (1.0 + 0.1 * np.arange(nz * nm).reshape(nz, nm, 1) / 24)) ** -2.5
    d_rho = ctx.upload(rho.reshape(-1))
    ctx.call("hmg_profile_fft_table", nz, nm, nk, nxs, step, d_xs.ptr, d_kts.ptr, d_rho.ptr, nz * nm, d_cmax.ptr, d_rss.ptr,
             d_zs.ptr, d_ks.ptr, 1, out_t.ptr)
    fused = out_t.numpy()
    ctx.close()
    monkeypatch.setenv("HMG_FUSED_FFT", "0")
    ctx = nat.Context(0)
    d_xs, d_kts, d_cmax, d_rss, d_zs, d_ks = (ctx.upload(a) for a in (xs, kts, cmax, rss, zs, ks))
    d_rho, out_r = ctx.upload(rho.reshape(-1)), ctx.empty((nz, nm, nk))
    ctx.call("hmg_profile_fft_table", nz, nm, nk, nxs, step, d_xs.ptr, d_kts.ptr, d_rho.ptr, nz * nm, d_cmax.ptr, d_rss.ptr,
             d_zs.ptr, d_ks.ptr, 1, out_r.ptr)
    assert np.max(np.abs(fused - out_r.numpy())) < 1e-12
    ctx.close()


def test_uk_fft():
    import hmvec_amd as hm
    k, u = hm.fft.uk_fft(lambda rr: 1.0 / (rr / 0.2) / (1.0 + rr / 0.2) ** 2, 1.5, dr=0.01, rmax=40)
    assert np.allclose(k[1:200], G["ukfft_k"], rtol=1e-15, atol=0)
    assert np.max(np.abs(u[1:200] - G["ukfft_u"])) < 1e-12


def test_bisection_self_test(capsys):
    """hmvec/utils.py:45-51 plus the pinned iterate."""
    import hmvec_amd as hm
    xs = np.array([2.0, 4.0, 6.0])
    d = hm.utils.vectorized_bisection_search(xs, lambda y: np.sqrt(y), (1, 40), "increasing", rtol=1e-4, verbose=True)
    assert np.all(np.isclose(d, np.array([4.0, 16.0, 36.0]), rtol=1e-3))
    assert np.array_equal(d, U["bisect_y"])
    assert "Bisection search converged in" in capsys.readouterr().out


def test_get_ngal_get_bg_methods():
    import hmvec_amd as hm
    zs = np.array([0.3, 1.1])
    ms = np.geomspace(1e11, 1e16, 48)
    h = hm.HaloModel(zs, np.geomspace(1e-3, 10, 8), ms=ms, accuracy="low", engine="analytic")
    h.add_hod("g", mthresh=10 ** 10.5 + zs * 0.0)
    hod = h.hods["g"]
    ng = h.get_ngal(hod["Nc"], hod["Ns"])
    assert rel_err(ng, hod["ngal"]) < 1e-13
    assert rel_err(h.get_bg(hod["Nc"], hod["Ns"], ng), hod["bg"]) < 1e-13


@pytest.mark.parametrize("mode", ["sheth-torman", "tinker"])
def test_get_fsigmaz_reproduces_the_mass_function(mode):
    """n(z,m) = rho_m0 f(sigma,z) (dln sigma^-1 / dln m) / m^2 (hmvec/hmvec.py:178-185) with f from the
    device-backed getter and numpy's gradient."""
    import hmvec_amd as hm
    zs = np.array([0.0, 0.9, 2.4])
    ms = np.geomspace(1e11, 1e16, 40)
    h = hm.HaloModel(zs, np.geomspace(1e-3, 10, 8), ms=ms, mass_function=mode, accuracy="low", engine="analytic",
                     skip_nfw=True)
    f = h.get_fsigmaz()
    assert f.shape == (3, 40) and np.all(f > 0)
    ln_sigma_inv = -0.5 * np.log(h.sigma2)
    n = h.rho_matter_z(0) * f * np.gradient(ln_sigma_inv, np.log(ms), axis=-1) / ms[None] ** 2
    assert np.allclose(n, h.nzm, rtol=1e-11, atol=0)
    assert np.array_equal(h.get_nzm(), h.nzm) and np.array_equal(h.get_bh(), h.bh)


def test_cosmology_layer_names():
    """Module-level Wkr / Wkr_taylor / limber_integral and the small Cosmology methods consumers use
    (examples/lensing_baryons.py: total_matter_power_spectrum, total_matter_galaxy_power_spectrum)."""
    import hmvec_amd as hm
    from hmvec_amd import cosmology as hc
    k = np.geomspace(1e-4, 50, 200)[None, :]
    R = np.array([0.5, 3.0, 20.0])[:, None]
    kR = k * R
    ref = 3.0 * (np.sin(kR) - kR * np.cos(kR)) / kR ** 3.0
    small = kR < 0.01
    ref[small] = 1 - 0.1 * kR[small] ** 2 + 0.00357142857143 * kR[small] ** 4
    assert np.allclose(hc.Wkr(k, R), ref, rtol=1e-13, atol=1e-15)
    assert np.allclose(hc.Wkr_taylor(kR[small]), ref[small], rtol=1e-15, atol=0)
    zs = np.linspace(0.1, 2.0, 6)
    ks = np.geomspace(1e-3, 20, 40)
    h = hm.HaloModel(zs, ks, ms=np.geomspace(1e11, 1e16, 32), accuracy="low", engine="analytic")
    h.add_battaglia_profile("electron", nxs=200, xmax=20)
    h.add_hod("g", mthresh=10 ** 10.5 + zs * 0.0)
    Pnn, Pne, Pee = h.get_power("nfw"), h.get_power("nfw", "electron"), h.get_power("electron")
    fc = h.p["omch2"] / (h.p["omch2"] + h.p["ombh2"])
    fb = 1 - fc
    assert np.allclose(h.total_matter_power_spectrum(Pnn, Pne, Pee), fc ** 2 * Pnn + 2 * fc * fb * Pne + fb ** 2 * Pee,
                       rtol=1e-14, atol=0)
    Pgn, Pge = h.get_power("g", "nfw"), h.get_power("g", "electron")
    assert np.allclose(h.total_matter_galaxy_power_spectrum(Pgn, Pge), fc * Pgn + fb * Pge, rtol=1e-14, atol=0)
    s8 = h.get_sigma8(zs)
    assert s8.shape == (6, 1) and np.all(np.diff(s8[:, 0]) < 0) and 0.5 < s8[0, 0] < 1.2   # (nz,1) as the reference
    assert np.allclose(s8 ** 2, h.get_sigma2_R(8.0 / h.p["H0"] * 100.0, zs, kmin=1e-4, numks=1000), rtol=1e-14)
    ells = np.linspace(100, 3000, 50)
    chis, hzs = h.comoving_radial_distance(zs), h.h_of_z(zs)
    w = h.lensing_window(zs, 2.5)
    a = hc.limber_integral(ells, zs, ks, Pnn, zs, w, w, hzs, chis)
    assert np.array_equal(a, h.C_kk(ells, zs, ks, Pnn, lzs1=2.5, lzs2=2.5))
    with pytest.raises(NameError):      # the reference's C_gy reads undefined names (cosmology.py:570-583)
        h.C_gy(ells, zs, ks, Pge, zs, gdndz=np.ones(zs.size))
    with pytest.raises(NameError):
        h.C_gy(ells, zs, ks, Pge, 0.8, zmin=0.7, zmax=0.9)
    assert h.P_mm_linear(zs, ks) is None


def test_uk_brute_force_against_the_reference():
    """fft.uk_brute_force (hmvec/fft.py:22-33; bin/tests.py:36): direct quadrature of the profile transform."""
    from hmvec_amd import fft as hfft
    g = load_golden("extra_pins")
    got = hfft.uk_brute_force(g["ub_r"], g["ub_rho"], 1.5, g["ub_k"])
    assert rel_err(got, g["ub_u"]) < 1e-12
