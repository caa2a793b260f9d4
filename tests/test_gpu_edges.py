"""Edge cases of the HIP path against the CPU oracle: degenerate and ragged grid sizes, FFT
lengths that take the workgroup-FFT path, the rocFFT fallback (odd length, prime factor > 5,
too long for LDS), user-supplied profiles, and C-ABI argument validation."""
import ctypes as C

import numpy as np
import pytest

from conftest import merged_params, power_close

pytestmark = pytest.mark.gpu


def oracle_for(h, zs, ks, ms, nxs, xmax):
    from hmvec_amd.params import battaglia_defaults
    from oracle import hmref
    p = merged_params()
    ksig = np.geomspace(p["sigma2_kmin"], p["sigma2_kmax"], p["sigma2_numks"])
    ci = hmref.CosmoInputs(h=h.h, omm0=h.omm0, ombh2=p["ombh2"], rho_crit_0=float(h.rho_critical_z(0.0)),
                           rho_crit_zs=h.rho_critical_z(zs), Pzk=h.Pzk, sPzk=h.sPzk, ks_sigma2=ksig,
                           h_of_z_zs=h.h_of_z(zs))
    o = hmref.RefHaloModel(ci, zs, ks, ms, p)
    o.add_battaglia_profile("electron", "AGN", p["battaglia_gas_gamma"], battaglia_defaults["AGN"], nxs, xmax)
    o.add_hod("g", mthresh=10 ** 10.5 + zs * 0.0)
    return o


@pytest.mark.parametrize("nz,nm,nk", [(1, 2, 1), (1, 5, 3), (3, 7, 65), (2, 64, 130), (5, 3, 2)])
def test_ragged_grid_sizes(nz, nm, nk):
    import hmvec_amd as hm
    zs = np.linspace(0.2, 2.2, nz)
    ms = np.geomspace(1e11, 1e16, nm)
    ks = np.geomspace(1e-3, 30, nk) if nk > 1 else np.array([0.3])
    h = hm.HaloModel(zs, ks, ms=ms, accuracy="low", engine="analytic")
    h.add_battaglia_profile("electron", nxs=200, xmax=20)
    h.add_hod("g", mthresh=10 ** 10.5 + zs * 0.0)
    o = oracle_for(h, zs, ks, ms, 200, 20)
    assert np.max(np.abs(h.uk_profiles["nfw"] - o.uk_profiles["nfw"])) < 1e-12
    assert np.max(np.abs(h.uk_profiles["electron"] - o.uk_profiles["electron"])) < 1e-12
    for a, b in (("nfw", "nfw"), ("g", "electron"), ("g", "g"), ("electron", "nfw")):
        ok, w = power_close(h.get_power(a, b), o.get_power(a, b))
        assert ok, (a, b, w)
    tot = h.get_power_all([("nfw", "electron"), ("g", "g")])
    ok, _ = power_close(tot[("g", "g")], o.get_power("g", "g"))
    assert ok


@pytest.mark.parametrize("nxs", [16, 30, 100, 600, 14, 45, 22, 25000, 5400, 6000, 7000, 12288])
def test_fft_lengths_fused_and_fallback(nxs):
    """16, 30, 100, 600: workgroup FFT (radix 5/4/3/2, incl. an odd half-length); 14, 22: prime
    factor 7/11 -> rocFFT; 45: odd length -> rocFFT; 25000: too long for LDS -> the long-grid route (10 x 1250);
    6000: fits LDS as one row but is longer than M = 2500 -> the long-grid route first (3 x 1000); 5400: longer than
    M = 2500 with no compiled sub-transform length as a divisor -> one row in LDS, run-time plan; 7000: factor 7 ->
    rocFFT; 12288: the longest one-row length (M = 6144 = 3 x 2048 -> the long-grid route)."""
    import hmvec_amd as hm
    zs = np.array([0.3, 1.4])
    ms = np.geomspace(1e12, 1e15, 6)
    ks = np.geomspace(1e-3, 30, 33)
    h = hm.HaloModel(zs, ks, ms=ms, accuracy="low", engine="analytic")
    h.add_battaglia_profile("electron", nxs=nxs, xmax=20)
    o = oracle_for(h, zs, ks, ms, nxs, 20)
    assert np.max(np.abs(h.uk_profiles["electron"] - o.uk_profiles["electron"])) < 1e-12


def test_user_supplied_profile_and_ms_none():
    import hmvec_amd as hm
    zs = np.array([0.5, 1.0])
    ms = np.geomspace(1e12, 1e15, 8)
    ks = np.geomspace(1e-2, 10, 16)
    h = hm.HaloModel(zs, ks, ms=ms, accuracy="low", engine="analytic")
    u = np.exp(-np.outer(np.geomspace(0.1, 2.0, 8), ks) ** 2)[None] * np.ones((2, 1, 1))
    h.uk_profiles["custom"] = u                      # numpy in -> uploaded
    assert "custom" in h.uk_profiles and np.array_equal(h.uk_profiles["custom"], u)
    trapz = getattr(np, "trapezoid", None) or np.trapz
    w = (ms[:, None] * u / h.rho_matter_z(0)) ** 2
    ref = trapz(h.nzm[..., None] * w, ms[:, None], axis=-2) * (1 - np.exp(-(ks / 0.01) ** 2))
    assert np.allclose(h.get_power_1halo("custom"), ref, rtol=1e-12)
    h2 = hm.HaloModel(zs, ks, ms=None, skip_nfw=True, accuracy="low", engine="analytic")
    assert len(h2.uk_profiles) == 0 and h2.Pzk.shape == (2, 16)


def test_c_abi_rejects_bad_arguments():
    from hmvec_amd import _native as nat
    ctx = nat.Context(0)
    lib = ctx.lib
    assert lib.hmg_sigma2(ctx.handle, 2, 2, 8, None, None, None, None, 0.01, None) != 0
    assert b"NULL" in lib.hmg_last_error()
    d = ctx.empty((4,))
    assert lib.hmg_massfn(ctx.handle, 0, 4, None, d.ptr, d.ptr, d.ptr, None, d.ptr, d.ptr) != 0
    par = nat.MassFnParams(mode=7)
    assert lib.hmg_massfn(ctx.handle, 1, 4, C.byref(par), d.ptr, d.ptr, d.ptr, None, d.ptr, d.ptr) != 0
    assert b"unknown mass function" in lib.hmg_last_error()
    assert lib.hmg_lane_set(ctx.handle, 99) != 0
    # the one-launch constructor stage: both halves are validated before anything is launched
    ok = nat.MassFnParams(mode=nat.MF_SHETH_TORMEN, deltac=1.686, st_A=0.3222, st_a=0.707, st_p=0.3, rho_m0=1.0)
    sig = (1, 4, 8, d.ptr, d.ptr, d.ptr, d.ptr, 0.01, C.byref(ok), d.ptr, d.ptr, None, d.ptr, d.ptr, d.ptr)
    assert lib.hmg_sigma2_massfn_halo(ctx.handle, *sig, None) != 0
    assert b"NULL" in lib.hmg_last_error()
    halo = nat.HaloStageArgs(d.ptr, d.ptr, d.ptr, 1.0, 0.1, 0.1, 0.7, d.ptr, d.ptr, d.ptr, None, None, 200.0, None,
                             d.ptr, None)
    assert lib.hmg_sigma2_massfn_halo(ctx.handle, *sig, C.byref(halo)) != 0
    assert b"both d_m2 and d_r2" in lib.hmg_last_error()
    halo.d_r2 = d.ptr
    assert lib.hmg_sigma2_massfn_halo(ctx.handle, *sig, C.byref(halo)) != 0
    assert b"d_drho1 and d_rho2" in lib.hmg_last_error()
    h = C.c_void_p()
    assert lib.hmg_ctx_create(12345, C.byref(h)) != 0
    assert b"out of range" in lib.hmg_last_error()
    with pytest.raises(nat.NativeError):
        ctx.call("hmg_event_record", -5)
    ctx.close()


def test_spectrum_cache_is_invalidated_by_state_changes():
    """get_power_1halo(a,b) + get_power_2halo(a,b) share one fused launch; any change of the
    profiles / HOD / mass function must drop the cached pair."""
    import hmvec_amd as hm
    zs = np.array([0.4, 1.1])
    ms = np.geomspace(1e12, 1e15, 12)
    ks = np.geomspace(1e-2, 10, 20)
    h = hm.HaloModel(zs, ks, ms=ms, accuracy="low", engine="analytic")
    h.add_battaglia_profile("electron", nxs=200, xmax=20)
    p1 = h.get_power_1halo("electron")
    v = h._version
    p2 = h.get_power_2halo("electron")
    assert h._version == v and ("electron", "electron") in h._pcache
    assert np.array_equal(h.get_power("electron"), p1 + p2)
    h.add_battaglia_profile("electron", family="SH", nxs=200, xmax=20, ignore_existing=True)
    q1 = h.get_power_1halo("electron")
    assert not np.allclose(q1, p1)                       # recomputed with the new profile
    h.uk_profiles["electron"] = np.ones((2, 12, 20))
    assert not np.allclose(h.get_power_1halo("electron"), q1)
    h.add_hod("g", mthresh=10 ** 10.5 + zs * 0.0)
    g1 = h.get_power_1halo("g")
    h.add_hod("g", mthresh=10 ** 11.5 + zs * 0.0, ignore_existing=True)
    assert not np.allclose(h.get_power_1halo("g"), g1)


@pytest.mark.parametrize("small_grid_rule", [False, True])
def test_free_rider_batching_matches_single_pair_kernel(small_grid_rule, monkeypatch):
    """A get_power_* call computes, in the same pass, every other pair whose tensors are already
    being streamed - on a small grid (round 5: tensors up to 32 MB, where another tensor in the batch costs less than
    another launch and result copy) every registered tracer; all of them must agree with the single-pair kernel,
    including for two different HOD names (kept on the per-pair path) and the (b, a) ordering."""
    import hmvec_amd as hm
    if not small_grid_rule:
        monkeypatch.setattr(hm.HaloModel, "_SMALL_GRID_BYTES", 0)      # the large-grid rule on this little grid
    zs = np.array([0.3, 0.9, 1.6])
    ms = np.geomspace(1e11, 1e16, 24)
    ks = np.geomspace(1e-3, 20, 40)
    h = hm.HaloModel(zs, ks, ms=ms, accuracy="low", engine="analytic")
    h.add_battaglia_profile("electron", nxs=300, xmax=20)
    h.add_hod("g", mthresh=10 ** 10.5 + zs * 0.0)
    h.add_hod("g2", mthresh=10 ** 11.0 + zs * 0.0, corr="min")
    names = ["nfw", "electron", "g", "g2"]
    ref = {}
    for a in names:
        for b in names:
            d1, d2 = h.power_device(a, b)            # single-pair kernel, no cache
            ref[(a, b)] = (d1.numpy(), d2.numpy())
    v = h._version
    p = h.get_power_1halo("nfw")                      # streams only the nfw tensor ...
    assert ("g", "nfw") in h._pcache and ("g", "g") in h._pcache      # ... g rides along for free
    assert (("electron", "electron") in h._pcache) == small_grid_rule   # a second tensor only on a small grid
    for a in names:
        for b in names:
            assert np.allclose(h.get_power_1halo(a, b), ref[(a, b)][0], rtol=1e-12, atol=0), (a, b)
            assert np.allclose(h.get_power_2halo(a, b), ref[(a, b)][1], rtol=1e-12, atol=0), (a, b)
    assert h._version == v and np.array_equal(p, h.get_power_1halo("nfw"))


def test_results_handed_out_are_the_callers_own_and_shared_inputs_survive_eviction():
    """Round 5 host-side sharing.  (a) A batch of spectra reaches the host in one copy and every get_power_* call hands
    out an array of the caller's own, as the reference does: writing into one result changes no later one.  (b) The device
    layout of P(k',z) is shared by models that are handed the same read-only host array (a loop over one cosmology) and
    the context keeps a few of them: six cosmologies in turn on one context, then the first again - every model equals
    what a fresh context computes."""
    import hmvec_amd as hm
    from hmvec_amd import _native as nat
    zs = np.array([0.2, 1.1]); ms = np.geomspace(1e11, 1e16, 24); ks = np.geomspace(1e-3, 20, 40)
    h = hm.HaloModel(zs, ks, ms=ms, accuracy="low", engine="analytic")
    a = h.get_power_1halo("nfw")
    keep = a.copy()
    a *= 0.0                                            # the caller scribbles over its array ...
    b = h.get_power_1halo("nfw")
    assert np.array_equal(b, keep) and b is not a        # ... the next call is not affected
    b += 1.0
    assert np.array_equal(h.get_power_1halo("nfw"), keep) and np.array_equal(h.get_power("nfw"), keep + h.get_power_2halo("nfw"))

    def spectrum(ctx, omch2):
        m = hm.HaloModel(zs, ks, ms=ms, params={"omch2": omch2}, accuracy="low", engine="analytic", ctx=ctx)
        return m.get_power("nfw"), m
    fresh = {}
    for w in (0.10, 0.11, 0.12, 0.13, 0.14, 0.15):
        c = nat.Context(0)
        fresh[w] = spectrum(c, w)[0]
        c.close()
    ctx = nat.Context(0)
    models = []
    for w in (0.10, 0.11, 0.12, 0.13, 0.14, 0.15, 0.10, 0.13):
        p, m = spectrum(ctx, w)
        models.append(m)                                 # (earlier models stay alive while entries are evicted)
        assert np.array_equal(p, fresh[w]), w
    assert len(ctx.shared) <= 4
    assert np.array_equal(models[0].get_power("nfw"), fresh[0.10])
    ctx.close()


def test_ksz_consumer_call_sequence():
    """Row N4: the call sequence of the largest in-repo consumer of the path, kSZ.__init__
    (hmvec/ksz.py:123-141,161-162,196-197): ctor with params=None, add_battaglia_profile by
    keyword with nxs/xmax=None defaults, add_hod by number density (global bisection), and
    get_power with name2=/verbose=/b1=/b2= keywords; checked against the oracle."""
    import hmvec_amd as hm
    from hmvec_amd.params import battaglia_defaults
    from oracle import hmref
    zs = np.array([0.4, 0.8, 1.3])
    ms = np.geomspace(1e11, 1e16, 40)
    ks = np.geomspace(0.1, 10.0, 31)
    ngals = np.array([3e-3, 2e-3, 1e-3])
    h = hm.HaloModel(zs, ks, ms=ms, params=None, mass_function="sheth-torman", halofit=None, mdef="vir",
                     nfw_numeric=False, skip_nfw=False, accuracy="low", engine="analytic")
    h.add_battaglia_profile(name="e", family="AGN", param_override=None, nxs=None, xmax=None,
                            ignore_existing=False)
    h.add_hod("g", mthresh=None, ngal=ngals, corr="max", satellite_profile_name="nfw",
              central_profile_name=None, ignore_existing=False, param_override=None)
    b1 = np.array([1.2, 1.5, 1.9])
    sPgg = h.get_power("g", name2="g", verbose=False, b1=b1, b2=b1)
    sPge = h.get_power("g", name2="e", verbose=False, b1=b1)
    aPgg = h.get_power("g", "g", verbose=False)
    aPge = h.get_power("g", "e", verbose=False)
    bg = np.array([h.hods["g"]["bg"][i] for i in range(zs.size)])

    p = merged_params()
    ksig = np.geomspace(p["sigma2_kmin"], p["sigma2_kmax"], p["sigma2_numks"])
    ci = hmref.CosmoInputs(h=h.h, omm0=h.omm0, ombh2=p["ombh2"], rho_crit_0=float(h.rho_critical_z(0.0)),
                           rho_crit_zs=h.rho_critical_z(zs), Pzk=h.Pzk, sPzk=h.sPzk, ks_sigma2=ksig,
                           h_of_z_zs=h.h_of_z(zs))
    o = hmref.RefHaloModel(ci, zs, ks, ms, p)
    o.add_battaglia_profile("e", "AGN", p["battaglia_gas_gamma"], battaglia_defaults["AGN"],
                            p["electron_density_profile_integral_numxs"],
                            p["electron_density_profile_integral_xmax"])
    o.add_hod("g", ngal=ngals)
    assert np.allclose(h.hods["g"]["log10mthresh"], o.hods["g"]["log10mthresh"], rtol=1e-13)
    assert np.allclose(bg, o.hods["g"]["bg"], rtol=1e-9)
    for got, (a, b, kw) in ((sPgg, ("g", "g", dict(b1=b1, b2=b1))), (sPge, ("g", "e", dict(b1=b1))),
                            (aPgg, ("g", "g", {})), (aPge, ("g", "e", {}))):
        ok, w = power_close(got, o.get_power(a, b, **kw))
        assert ok, (a, b, kw.keys(), w)


def test_lensing_baryons_example_sequence():
    """examples/lensing_baryons.py, the reference's worked example, line for line (minus the plots;
    `accuracy='low'` because CAMB is not in this image): HOD by number density on a 20-redshift grid,
    a Battaglia profile with nxs=30000 / xmax=50 (too long for the workgroup FFT: the rocFFT route at
    full size), total-matter combinations, and galaxy-galaxy lensing / cosmic shear Limber ratios -
    each quantity against the oracle."""
    import hmvec_amd as hm
    from hmvec_amd.params import battaglia_defaults
    from oracle import hmref
    zgalaxy, zsource, ngal = 0.6, 1.0, 1e-4
    zs = np.linspace(0.01, zsource + 1, 20)
    ms = np.geomspace(2e10, 1e17, 100)
    ks = np.geomspace(1e-4, 100, 1001)
    hcos = hm.HaloModel(zs, ks, ms=ms, accuracy="low", engine="analytic")
    hcos.add_hod("g", ngal=ngal + zs * 0.0, corr="max")
    hcos.add_battaglia_profile("electron", family="AGN", xmax=50, nxs=30000)
    Pgn = hcos.get_power("g", "nfw", verbose=False)
    Pge = hcos.get_power("g", "electron", verbose=False)
    Pgm = hcos.total_matter_galaxy_power_spectrum(Pgn, Pge)
    ells = np.linspace(80, 6000, 100)
    Ckg0 = hcos.C_kg(ells, zs, ks, Pgn, gzs=zgalaxy, lzs=zsource)
    Ckg = hcos.C_kg(ells, zs, ks, Pgm, gzs=zgalaxy, lzs=zsource)
    Pnn = hcos.get_power("nfw", verbose=False)
    Pne = hcos.get_power("nfw", "electron", verbose=False)
    Pee = hcos.get_power("electron", "electron", verbose=False)
    Pmm = hcos.total_matter_power_spectrum(Pnn, Pne, Pee)
    Ckk0 = hcos.C_kk(ells, zs, ks, Pnn, lzs1=zsource, lzs2=zsource)
    Ckk = hcos.C_kk(ells, zs, ks, Pmm, lzs1=zsource, lzs2=zsource)

    p = merged_params()
    ksig = np.geomspace(p["sigma2_kmin"], p["sigma2_kmax"], p["sigma2_numks"])
    ci = hmref.CosmoInputs(h=hcos.h, omm0=hcos.omm0, ombh2=p["ombh2"], rho_crit_0=float(hcos.rho_critical_z(0.0)),
                           rho_crit_zs=hcos.rho_critical_z(zs), Pzk=hcos.Pzk, sPzk=hcos.sPzk, ks_sigma2=ksig,
                           h_of_z_zs=hcos.h_of_z(zs))
    o = hmref.RefHaloModel(ci, zs, ks, ms, p)
    o.add_hod("g", ngal=ngal + zs * 0.0)
    assert np.allclose(hcos.hods["g"]["log10mthresh"], o.hods["g"]["log10mthresh"], rtol=1e-13)
    assert np.allclose(hcos.hods["g"]["bg"], o.hods["g"]["bg"], rtol=1e-9)
    o.add_battaglia_profile("electron", "AGN", p["battaglia_gas_gamma"], battaglia_defaults["AGN"], 30000, 50)
    assert np.max(np.abs(hcos.uk_profiles["electron"] - o.uk_profiles["electron"])) < 1e-12
    fc = p["omch2"] / (p["omch2"] + p["ombh2"])
    fb = 1.0 - fc
    oPgn, oPge = o.get_power("g", "nfw"), o.get_power("g", "electron")
    oPnn, oPne, oPee = o.get_power("nfw"), o.get_power("nfw", "electron"), o.get_power("electron")
    oPgm = fc * oPgn + fb * oPge
    oPmm = fc ** 2 * oPnn + 2 * fc * fb * oPne + fb ** 2 * oPee
    for got, want in ((Pgn, oPgn), (Pge, oPge), (Pgm, oPgm), (Pnn, oPnn), (Pne, oPne), (Pee, oPee), (Pmm, oPmm)):
        ok, w = power_close(got, want)
        assert ok, w
    H0, chis, hz = hcos.h_of_z(0.0), hcos.comoving_radial_distance(zs), hcos.h_of_z(zs)
    chig, hg = hcos.comoving_radial_distance(np.array([zgalaxy])), hcos.h_of_z(np.array([zgalaxy]))
    chistar = hcos.comoving_radial_distance(np.array([zsource]))
    wg = hmref.lensing_window(np.array([zgalaxy]), zsource, H0, hg, chig, chistar, hcos.omm0)
    wz = hmref.lensing_window(zs, zsource, H0, hz, chis, chistar, hcos.omm0)
    oCkg0 = hmref.limber_integral(ells, zs, ks, oPgn, zgalaxy, wg, 1.0, hg, chig)
    oCkg = hmref.limber_integral(ells, zs, ks, oPgm, zgalaxy, wg, 1.0, hg, chig)
    oCkk0 = hmref.limber_integral(ells, zs, ks, oPnn, zs, wz, wz, hz, chis)
    oCkk = hmref.limber_integral(ells, zs, ks, oPmm, zs, wz, wz, hz, chis)
    for got, want in ((Ckg0, oCkg0), (Ckg, oCkg), (Ckk0, oCkk0), (Ckk, oCkk)):
        assert np.allclose(got, want, rtol=1e-8, atol=0)
    # the example's plotted quantities: baryonic feedback suppresses small-scale lensing by a few per cent
    assert np.all(np.abs(Ckg / Ckg0 - 1) < 0.2) and np.all(np.abs(Ckk / Ckk0 - 1) < 0.2)
    assert (Ckk / Ckk0)[-1] < 1.0


def test_numeric_nfw_constructor_default_length():
    """bin/test_generic_fft.py: HaloModel(..., nfw_numeric=True) transforms the NFW profile with the
    default nxs=40000 / xmax=200 (hmvec/params.py:59-60; rocFFT route, 20000 complex points per row)
    and must agree with the analytic profile at the level of the reference's FFT conventions."""
    import hmvec_amd as hm
    from oracle import hmref
    zs = np.array([3.0])
    ms = np.geomspace(2e10, 1e17, 50)
    ks = np.geomspace(1e-4, 100, 301)
    num = hm.HaloModel(zs, ks, ms=ms, nfw_numeric=True, accuracy="low", engine="analytic")
    ana = hm.HaloModel(zs, ks, ms=ms, nfw_numeric=False, accuracy="low", engine="analytic")
    un, ua = num.uk_profiles["nfw"], ana.uk_profiles["nfw"]
    cs, rss = ana.concentration(), ana._d_rvir.numpy() / ana.concentration()
    want = hmref.profile_fft(lambda x: 1.0 / x / (1.0 + x) ** 2.0, cs, rss, zs, ks, 200, 40000)
    assert np.max(np.abs(un - want)) < 1e-12
    sel = ks < 5.0
    assert np.max(np.abs(un[..., sel] - ua[..., sel])) < 2e-2          # the step/phase quirks of fft_integral
    p1n, p1a = num.get_power_1halo("nfw"), ana.get_power_1halo("nfw")
    assert np.allclose(p1n[:, sel], p1a[:, sel], rtol=5e-2)


def test_constant_prefix_hints_change_nothing(monkeypatch):
    """hmg_profile_fft's per-row hints (np.interp left-fill region) let hmg_power_batch skip parts of a
    tensor; the spectra must be bit-identical with and without them, a descending k grid must get no
    hints, and redefining the tensor by hand must drop them."""
    import hmvec_amd as hm
    monkeypatch.delenv("HMG_NO_HINTS", raising=False)      # (the suite may run under the switch: tools/env_matrix.sh)
    zs = np.array([0.2, 1.0, 2.5])
    ms = np.geomspace(1e10, 1e16, 96)
    ks = np.geomspace(1e-4, 50, 512)
    pairs = [("nfw", "nfw"), ("electron", "electron"), ("g", "g"), ("nfw", "electron"), ("g", "electron"), ("y", "y"),
             ("y", "electron")]

    def build(kgrid):
        h = hm.HaloModel(zs, kgrid, ms=ms, accuracy="low", engine="analytic")
        h.add_battaglia_profile("electron", nxs=1000, xmax=20)
        h.add_battaglia_pres_profile("y", nxs=1000, xmax=20)
        h.add_hod("g", mthresh=10 ** 10.5 + zs * 0.0, central_profile_name="electron")
        return h

    h = build(ks)
    hint = h.uk_profiles.hint("electron")
    assert hint[0] is not None and h.pk_profiles.hint("y")[0] is not None and h.uk_profiles.hint("nfw")[0] is None
    n, c = hint[0].numpy().view(np.int32)[:zs.size * ms.size].reshape(zs.size, ms.size), hint[1].numpy()
    ue = h.uk_profiles["electron"]
    assert n.min() >= 0 and n.max() > 128                       # whole 128-k tiles are skippable on this grid
    for iz, im in ((0, 0), (1, 40), (2, 95)):
        k = int(n[iz, im])
        assert np.all(ue[iz, im, :k] == c[iz, im]) and (k == ks.size or ue[iz, im, k] != c[iz, im])
    with_hints = h.get_power_all(pairs[:5])
    monkeypatch.setenv("HMG_NO_HINTS", "1")
    h0 = build(ks)
    assert h0.uk_profiles.hint("electron")[0] is None
    without = h0.get_power_all(pairs[:5])
    for p in pairs[:5]:
        assert np.array_equal(with_hints[p], without[p]), p
    monkeypatch.delenv("HMG_NO_HINTS")
    hd = build(ks[::-1].copy())                                 # descending grid: no prefix notion
    assert hd.uk_profiles.hint("electron")[0] is None
    assert np.allclose(hd.get_power("g", "electron")[:, ::-1], with_hints[("g", "electron")], rtol=1e-12, atol=0)
    h.uk_profiles["electron"] = ue * 2.0                        # redefined by hand: the hint must go
    assert h.uk_profiles.hint("electron")[0] is None
    p1_before = build(ks).get_power_1halo("electron")
    o1, _ = h.power_device_batch([("electron", "electron")])   # batched kernel: the one that uses hints
    assert np.allclose(o1[0].numpy(), 4.0 * p1_before, rtol=1e-12, atol=0)       # 1-halo is quadratic in u


@pytest.mark.parametrize("nm", [5, 16, 23, 64, 100])
def test_mass_integral_launch_shapes_agree_bit_for_bit(monkeypatch, nm):
    """hmg_power_batch has two launch shapes - 8 wavefronts walking two of the 16 virtual mass slices each
    (full grids) and 16 wavefronts with one slice each (thin z-slabs) - that realise the SAME summation
    order, so that a slab run reproduces the full grid's bits.  Force either shape on the same
    grid, including mass-bin counts that leave slices empty or ragged, and compare bitwise."""
    import hmvec_amd as hm
    zs = np.array([0.3, 0.9, 1.6])
    ms = np.geomspace(1e11, 1e16, nm)
    ks = np.geomspace(1e-3, 20, 130)
    h = hm.HaloModel(zs, ks, ms=ms, accuracy="low", engine="analytic")
    h.add_battaglia_profile("electron", nxs=300, xmax=20)
    h.add_hod("g", mthresh=10 ** 10.5 + zs * 0.0)
    pairs = [("nfw", "nfw"), ("electron", "electron"), ("g", "g"), ("nfw", "electron"), ("g", "nfw"), ("g", "electron")]
    out = {}
    for thin in ("0", "1", "3"):
        monkeypatch.setenv("HMG_PB_THIN", thin)
        o1, o2 = h.power_device_batch(pairs)
        out[thin] = [a.numpy() for a in o1] + [a.numpy() for a in o2]
    for a, b, c3 in zip(out["0"], out["1"], out["3"]):
        assert np.array_equal(a, b) and np.array_equal(a, c3)
    monkeypatch.delenv("HMG_PB_THIN")
    o = oracle_for(h, zs, ks, ms, 300, 20)
    for (a, b), p1, p2 in zip(pairs, out["0"][:6], out["0"][6:]):
        ok, w = power_close(p1 + p2, o.get_power(a, b))
        assert ok, (a, b, w)


def test_compile_time_plan_equals_run_time_plan(monkeypatch):
    """nxs = 5000 (the Battaglia default) runs a build of the fused profile kernel whose FFT plan is a
    compile-time constant; the run-time-plan build of the same kernel must give the same bits, for rows
    that take the pruned first pass and rows that do not (xmax small enough that cmax > xmax/4)."""
    import hmvec_amd as hm
    zs = np.array([0.2, 1.1, 2.7])
    ms = np.geomspace(2e10, 1e17, 40)
    ks = np.geomspace(1e-4, 100, 300)
    out = {}
    for generic in ("0", "1"):
        if generic == "1":
            monkeypatch.setenv("HMG_FUSED_GENERIC", "1")
        h = hm.HaloModel(zs, ks, ms=ms, accuracy="low", engine="analytic")
        h.add_battaglia_profile("e20", nxs=5000, xmax=20)          # truncation at ~2.6 of 20: first pass pruned
        h.add_battaglia_profile("e8", nxs=5000, xmax=8)            # 2.6 > 8/4: not pruned
        h.add_battaglia_profile("e12", nxs=5000, xmax=12)          # pruned, but not zero from sample 375 on: the
        h.add_battaglia_pres_profile("y", nxs=5000, xmax=20)       # full butterfly behind the pruned pass
        out[generic] = (h.uk_profiles["e20"].copy(), h.uk_profiles["e8"].copy(), h.pk_profiles["y"].copy(),
                        h.uk_profiles["e12"].copy())
    monkeypatch.delenv("HMG_FUSED_GENERIC")
    for a, b in zip(out["0"], out["1"]):
        assert np.array_equal(a, b)
    o = oracle_for(h, zs, ks, ms, 5000, 8)
    assert np.max(np.abs(out["0"][1] - o.uk_profiles["electron"])) < 1e-12
    o = oracle_for(h, zs, ks, ms, 5000, 12)
    assert np.max(np.abs(out["0"][3] - o.uk_profiles["electron"])) < 1e-12


@pytest.mark.parametrize("nk,klo,khi", [(301, 0.5, 8.0), (64, 1e-4, 1e-2), (130, 40.0, 900.0), (257, 1e-3, 3000.0)])
def test_left_fill_prefix_of_every_length(default_routes, nk, klo, khi):
    """The fused profile kernel finds the end of np.interp's left-fill prefix of a row by a search and writes the
    prefix as a plain fill: rows whose targets all lie below the first FFT mode (prefix = the whole row), rows with
    none below it, odd row lengths (every other row starts off a 16-byte boundary) and targets beyond the last mode."""
    import hmvec_amd as hm
    zs = np.array([0.1, 0.9, 2.4])
    ms = np.geomspace(2e10, 1e17, 37)
    ks = np.geomspace(klo, khi, nk)
    h = hm.HaloModel(zs, ks, ms=ms, accuracy="low", engine="analytic")
    h.add_battaglia_profile("electron", nxs=5000, xmax=20)
    h.add_battaglia_profile("e1000", nxs=1000, xmax=20)            # run-time plan
    ue = h.uk_profiles["electron"]
    n = h.uk_profiles.hint("electron")[0].numpy().view(np.int32)[:zs.size * ms.size].reshape(zs.size, ms.size)
    c = h.uk_profiles.hint("electron")[1].numpy()
    for iz in range(zs.size):
        for im in range(ms.size):
            k = int(n[iz, im])
            assert 0 <= k <= nk and np.all(ue[iz, im, :k] == c[iz, im]), (iz, im)
    if (nk, klo) == (301, 0.5):
        assert n.min() == 0 and n.max() == nk                   # both extremes occur on this grid
    o = oracle_for(h, zs, ks, ms, 5000, 20)
    assert np.max(np.abs(ue - o.uk_profiles["electron"])) < 1e-12
    o = oracle_for(h, zs, ks, ms, 1000, 20)
    assert np.max(np.abs(h.uk_profiles["e1000"] - o.uk_profiles["electron"])) < 1e-12


@pytest.mark.parametrize("nz,nm", [(3, 7), (2, 64), (5, 130), (1, 65)])
def test_constructor_stage_in_one_launch_equals_the_two_launches(nz, nm, monkeypatch):
    """hmg_sigma2_massfn_halo (mass function and halo stage side by side in one launch, what the
    constructor issues) against hmg_sigma2_massfn + hmg_halo_stage (what the two-lane scheme issues):
    every (z,m) array bit for bit, across partial 64-mass tiles."""
    import hmvec_amd as hm
    zs = np.linspace(0.1, 2.5, nz)
    ms = np.geomspace(3e10, 1e16, nm)
    ks = np.geomspace(1e-3, 30, 17)
    one = hm.HaloModel(zs, ks, ms=ms, accuracy="low", engine="analytic")
    monkeypatch.setenv("HMG_LANES", "1")
    two = hm.HaloModel(zs, ks, ms=ms, accuracy="low", engine="analytic")
    assert not one._use_lanes and two._use_lanes
    for name in ("sigma2", "nzm", "bh"):
        assert np.array_equal(getattr(one, name), getattr(two, name)), name
    for name in ("_d_cs", "_d_rvir", "_d_rs", "_d_nfw_series"):
        assert np.array_equal(getattr(one, name).numpy(), getattr(two, name).numpy()), name
    for name in ("m200c", "r200c"):
        assert np.array_equal(one._buf(name, (nz, nm)).numpy(), two._buf(name, (nz, nm)).numpy()), name
    assert np.array_equal(one.uk_profiles["nfw"], two.uk_profiles["nfw"])


def test_a_spectrum_does_not_depend_on_what_was_asked_before():
    """get_power(a,b) comes from the batched kernel whether or not other pairs ride along, and a pair's
    sums do not depend on the rest of the batch: the same bits for every call history."""
    import hmvec_amd as hm
    zs = np.linspace(0.2, 2.0, 5)
    ms = np.geomspace(1e11, 1e16, 70)
    ks = np.geomspace(1e-3, 30, 150)

    def model():
        h = hm.HaloModel(zs, ks, ms=ms, accuracy="low", engine="analytic")
        h.add_battaglia_profile("electron", nxs=400, xmax=20)
        h.add_battaglia_pres_profile("y", nxs=400, xmax=20)
        h.add_hod("g", mthresh=10 ** 10.5 + zs * 0.0)
        return h

    pairs = [("electron", "electron"), ("nfw", "nfw"), ("g", "g"), ("g", "electron"), ("nfw", "electron"),
             ("y", "y"), ("g", "y")]
    a = model()
    first = {p: (a.get_power_1halo(*p), a.get_power_2halo(*p)) for p in pairs}
    b = model()
    for p in reversed(pairs):                       # other order: other batches, other free riders
        one, two = b.get_power_1halo(*p), b.get_power_2halo(*p)
        assert np.array_equal(one, first[p][0]) and np.array_equal(two, first[p][1]), p
    c = model()
    blk = c.spectra_block(pairs[:5])                # and the explicit batch of bench.py / ShardedSpectra
    blk.compute()
    for p, (one, two) in blk.fetch().items():
        assert np.array_equal(one, first[p][0]) and np.array_equal(two, first[p][1]), p


def test_structure_compiled_mass_integrals_equal_the_generic_forms(monkeypatch):
    """hmg_power_batch runs a kernel compiled for the batch's structure (which coefficients of the linear
    forms are structural zeros or ones) where it has one, the generic forms otherwise (HMG_PB_GENERIC=1
    forces them): the same sums bit for bit, for every batch the facade issues here - one to four tracers,
    one to three tensors, tracers in any order, every launch shape."""
    import hmvec_amd as hm
    zs = np.linspace(0.2, 2.0, 3)
    ms = np.geomspace(1e11, 1e16, 70)
    ks = np.geomspace(1e-3, 30, 130)

    def model():
        h = hm.HaloModel(zs, ks, ms=ms, accuracy="low", engine="analytic")
        h.add_battaglia_profile("electron", nxs=400, xmax=20)
        h.add_battaglia_pres_profile("y", nxs=400, xmax=20)
        h.add_hod("g", mthresh=10 ** 10.5 + zs * 0.0)
        h.add_hod("gc", mthresh=10 ** 10.8 + zs * 0.0, central_profile_name="electron")   # no compact form
        return h

    batches = [[("nfw", "nfw")], [("electron", "electron")], [("nfw", "nfw"), ("g", "g"), ("g", "nfw")],
               [("electron", "nfw"), ("nfw", "nfw")], [("g", "electron"), ("electron", "electron"), ("nfw", "g")],
               [("y", "y"), ("nfw", "y"), ("electron", "y")], [("g", "y"), ("y", "electron"), ("nfw", "nfw"), ("g", "g")],
               [("gc", "gc"), ("gc", "nfw")]]
    for thin in ("0", "1", "3"):
        monkeypatch.setenv("HMG_PB_THIN", thin)
        for pairs in batches:
            monkeypatch.delenv("HMG_PB_GENERIC", raising=False)
            a = model().spectra_block(pairs)
            a.compute()
            got = {p: (x.copy(), y.copy()) for p, (x, y) in a.fetch().items()}
            monkeypatch.setenv("HMG_PB_GENERIC", "1")
            b = model().spectra_block(pairs)
            b.compute()
            for p, (x, y) in b.fetch().items():
                assert np.array_equal(got[p][0], x) and np.array_equal(got[p][1], y), (thin, pairs, p)


def test_a_bad_registered_tracer_cannot_break_an_unrelated_request():
    """on small grids every registered tracer rides in the first batch of mass integrals.  A tracer whose
    tensor is not in this model's (nz, nm, nk) shape - a hand-assigned uk_profiles entry - must neither ride nor make the
    request it would have ridden with fail; the requested spectrum equals the one of a model without that entry."""
    import hmvec_amd as hm
    zs = np.array([0.2, 1.0])
    ms = np.geomspace(1e11, 1e16, 40)
    ks = np.geomspace(1e-3, 30, 64)
    ref = hm.HaloModel(zs, ks, ms=ms, accuracy="low", engine="analytic")
    want = ref.get_power("nfw")
    h = hm.HaloModel(zs, ks, ms=ms, accuracy="low", engine="analytic")
    h.uk_profiles["broken"] = np.ones((zs.size, ms.size, ks.size // 2))        # wrong k length
    h.uk_profiles["good"] = np.ones((zs.size, ms.size, ks.size)) * 0.5
    got = h.get_power("nfw")
    assert np.array_equal(got, want)
    assert ("good", "nfw") in h._pcache or ("nfw", "good") in h._pcache          # the valid rider did ride ...
    assert not any("broken" in k for k in h._pcache)                              # ... the bad one did not
    with pytest.raises(ValueError):
        h.get_power("broken")                                                     # asked for by name: refused BEFORE any launch
