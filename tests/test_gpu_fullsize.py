"""Config 3 (BASELINE.json configs[2]: 32 x 512 x 4096, NFW + electron + HOD) at FULL size on
the GPU, checked through size-independent properties of the path plus an oracle sub-sample:
batched == per-pair, z-slab == full grid, the 2-halo consistency limit, 1-halo damping,
affine dependence on an injected bias, profile limits, sortedness of inputs irrelevant."""
import numpy as np
import pytest

from conftest import merged_params, power_close

pytestmark = pytest.mark.gpu

NZ, NM, NK, NXS = 32, 512, 4096, 5000
PAIRS = [("nfw", "nfw"), ("electron", "electron"), ("g", "g"),
         ("nfw", "electron"), ("g", "nfw"), ("g", "electron")]


def grids():
    return np.linspace(0.01, 3.0, NZ), np.geomspace(2e10, 1e17, NM), np.geomspace(1e-4, 100, NK)


def build(zs, ms, ks):
    import hmvec_amd as hm
    h = hm.HaloModel(zs, ks, ms=ms, accuracy="low", engine="analytic")
    h.add_battaglia_profile("electron", family="AGN", xmax=20, nxs=NXS)
    h.add_hod("g", mthresh=10 ** 10.5 + zs * 0.0)
    return h


@pytest.fixture(scope="module")
def full():
    zs, ms, ks = grids()
    return build(zs, ms, ks)


def test_batched_equals_per_pair_full_size(full):
    o1, o2 = full.power_device_batch(PAIRS)
    for (a, b), d1, d2 in zip(PAIRS, o1, o2):
        assert np.allclose(d1.numpy(), full.get_power_1halo(a, b), rtol=1e-12, atol=0), (a, b)
        assert np.allclose(d2.numpy(), full.get_power_2halo(a, b), rtol=1e-12, atol=0), (a, b)


def test_zslab_equals_full_grid(full):
    """Every stage is independent along z: a slab model reproduces its rows."""
    zs, ms, ks = grids()
    lo, hi = 12, 16
    slab = build(zs[lo:hi], ms, ks)
    # a fresh full-grid model with the same call sequence as the slab: a spectrum can come from the one-pair
    # kernel or from a batched launch depending on what was asked before (per-pair cache, free riders), and the
    # two sum in different orders (~1e-16), so the module fixture with its history is not the like-for-like partner
    full = build(zs, ms, ks)
    assert np.array_equal(slab.sigma2, full.sigma2[lo:hi])
    assert np.array_equal(slab.nzm, full.nzm[lo:hi])
    # (the reference's own slab runs differ by ~7e-15 in the electron spectra - its secant solver stops on a
    # grid-wide test; here the mass conversion is solved per (z,m), so these are bit-equal too)
    assert np.array_equal(slab.uk_profiles["electron"], full.uk_profiles["electron"][lo:hi])
    for a, b in (("nfw", "nfw"), ("g", "g"), ("electron", "electron"), ("g", "electron"), ("nfw", "electron")):
        assert np.array_equal(slab.get_power(a, b), full.get_power(a, b)[lo:hi]), (a, b)


def test_two_halo_consistency_and_damping(full):
    zs, ms, ks = grids()
    p1, p2 = full.get_power_1halo("nfw"), full.get_power_2halo("nfw")
    assert np.allclose(p2[:, 0] / full.Pzk[:, 0], 1.0, rtol=1e-5)              # P2h -> P_lin (hmvec.py:566-572)
    bg = full.hods["g"]["bg"]
    pg = full.get_power_2halo("g")
    assert np.allclose(pg[:, 0] / full.Pzk[:, 0], bg ** 2, rtol=1e-5)          # -> b_g^2 P_lin
    assert np.all(p1[:, ks < 1e-3] < 1e-3 * p2[:, ks < 1e-3])                   # 1h damped below k* (hmvec.py:526)
    assert np.all(np.isfinite(p1)) and np.all(np.isfinite(p2)) and np.all(p1 >= 0)


def test_two_halo_affine_in_injected_bias(full):
    zs = grids()[0]
    one = np.ones(zs.size)
    P = [full.get_power_2halo("g", "nfw", b1_in=b * one, b2_in=one) for b in (1.0, 2.0, 3.0)]
    d1, d2 = P[1] - P[0], P[2] - P[1]
    assert np.allclose(d1, d2, rtol=1e-10, atol=1e-12 * np.max(np.abs(P[2])))


def test_profile_limits(full):
    u = full.uk_profiles["nfw"]
    assert np.allclose(u[:, :, 0], 1.0, atol=1e-5)            # u(k->0) = 1
    assert np.max(np.abs(u)) <= 1.0 + 1e-12
    ue = full.uk_profiles["electron"]
    assert np.all(np.isfinite(ue)) and np.max(np.abs(ue)) < 1.1
    # mass-normalised; the k->0 value is the first FFT mode (left=u[first k>0], hmvec/fft.py:107),
    # i.e. u at k x ~ 0.3, a little below 1
    assert np.all(ue[:, :, 0] > 0.5) and np.all(ue[:, :, 0] < 1.05)


def test_against_oracle_subsample(full):
    """Two redshifts of the full grid through the CPU oracle (seconds) vs the same rows on the GPU."""
    from hmvec_amd.params import battaglia_defaults
    from oracle import hmref
    zs, ms, ks = grids()
    sel = np.array([3, 27])
    z = zs[sel]
    p = merged_params()
    ksig = np.geomspace(p["sigma2_kmin"], p["sigma2_kmax"], p["sigma2_numks"])
    ci = hmref.CosmoInputs(h=full.h, omm0=full.omm0, ombh2=p["ombh2"],
                           rho_crit_0=float(full.rho_critical_z(0.0)), rho_crit_zs=full.rho_critical_z(z),
                           Pzk=full.Pzk[sel], sPzk=full.sPzk[sel], ks_sigma2=ksig, h_of_z_zs=full.h_of_z(z))
    o = hmref.RefHaloModel(ci, z, ks, ms, p)
    o.add_battaglia_profile("electron", "AGN", p["battaglia_gas_gamma"], battaglia_defaults["AGN"], NXS, 20)
    o.add_hod("g", mthresh=10 ** 10.5 + z * 0.0)
    assert np.max(np.abs(full.uk_profiles["nfw"][sel] - o.uk_profiles["nfw"])) < 1e-12
    assert np.max(np.abs(full.uk_profiles["electron"][sel] - o.uk_profiles["electron"])) < 1e-12
    for a, b in PAIRS:
        ok, w = power_close(full.get_power(a, b)[sel], o.get_power(a, b))
        assert ok, (a, b, w)


def test_ngal_mode_on_a_zslab_uses_the_global_bisection(capsys):
    """SURVEY 8e exception: add_hod(ngal=...) on a slab must use thresholds bisected on the
    FULL z grid (the reference's stop test is global over z)."""
    import hmvec_amd as hm
    from hmvec_amd.dist import mthresh_from_ngal_global
    zs = np.linspace(0.05, 2.5, 8)
    ms = np.geomspace(1e11, 1e16, 40)
    ks = np.geomspace(1e-3, 10, 16)
    kw = dict(accuracy="low", engine="analytic")
    full = hm.HaloModel(zs, ks, ms=ms, **kw)
    full.add_hod("g0", mthresh=10 ** 10.5 + zs * 0.0)
    target = full.hods["g0"]["ngal"] * 1.37
    full.add_hod("g", ngal=target)
    mthr = mthresh_from_ngal_global(zs, ks, ms, target, **kw)
    assert np.array_equal(np.log10(mthr), full.hods["g"]["log10mthresh"][:, 0])
    lo, hi = 2, 4
    slab = hm.HaloModel(zs[lo:hi], ks, ms=ms, **kw)
    slab.add_hod("g", mthresh=mthr[lo:hi])
    assert np.array_equal(slab.get_power("g"), full.get_power("g")[lo:hi])


def test_config5_limber_at_full_size(full):
    """BASELINE.json configs[4]: Config 3's grid + C_kk and C_kg at 2000 multipoles (lzs=2.5, gzs=0.8;
    README.rst:106,123, hmvec/cosmology.py:536-568,867-904), from the device-resident GPU spectra
    (P_1h + P_2h summed inside the Limber kernel) against the oracle's limber_integral on the same
    (32 x 4096) arrays."""
    from oracle import hmref
    zs, ms, ks = grids()
    ells = np.linspace(100, 6000, 2000)
    lzs, gzs = 2.5, 0.8
    ckk = full.C_kk(ells, zs, ks, full.power_device("nfw", "nfw"), lzs1=lzs, lzs2=lzs)
    ckg = full.C_kg(ells, zs, ks, full.power_device("g", "nfw"), gzs=gzs, lzs=lzs)
    Pmm, Pgm = full.get_power("nfw"), full.get_power("g", "nfw")
    H0, chis, hz = full.h_of_z(0.0), full.comoving_radial_distance(zs), full.h_of_z(zs)
    chistar = full.comoving_radial_distance(np.array([lzs]))
    wz = hmref.lensing_window(zs, lzs, H0, hz, chis, chistar, full.omm0)
    gz = np.array([gzs])
    chig, hg = full.comoving_radial_distance(gz), full.h_of_z(gz)
    wg = hmref.lensing_window(gz, lzs, H0, hg, chig, chistar, full.omm0)
    okk = hmref.limber_integral(ells, zs, ks, Pmm, zs, wz, wz, hz, chis)
    okg = hmref.limber_integral(ells, zs, ks, Pgm, gzs, wg, 1.0, hg, chig)
    assert ckk.shape == ckg.shape == (2000,)
    assert np.allclose(ckk, okk, rtol=1e-8, atol=0) and np.allclose(ckg, okg, rtol=1e-8, atol=0)
    # the host-array route (what the reference's signature takes) gives the same numbers
    assert np.allclose(full.C_kk(ells, zs, ks, Pmm, lzs1=lzs, lzs2=lzs), ckk, rtol=1e-13, atol=0)
    assert np.all(ckk > 0) and np.all(np.diff(ckk) < 0)            # a lensing power spectrum: positive, falling


def test_spectra_block_hand_over(full):
    """All twelve (nz,nk) outputs through one device block and one pinned host block."""
    blk = full.spectra_block(PAIRS)
    blk.compute()
    got = blk.fetch()
    for a, b in PAIRS:
        assert np.allclose(got[(a, b)][0], full.get_power_1halo(a, b), rtol=1e-12, atol=0), (a, b)
        assert np.allclose(got[(a, b)][1], full.get_power_2halo(a, b), rtol=1e-12, atol=0), (a, b)


def test_captured_step_replays_identically():
    """A pass captured as a HIP graph reproduces the eager pass bit for bit, also after the inputs of
    the pass changed in place (same buffers, new HOD thresholds)."""
    import hmvec_amd as hm
    zs = np.linspace(0.1, 2.5, 8)
    ms = np.geomspace(2e10, 1e17, 128)
    ks = np.geomspace(1e-4, 100, 512)
    h = hm.HaloModel(zs, ks, ms=ms, accuracy="low", engine="analytic")
    h.add_battaglia_profile("electron", family="AGN", xmax=20, nxs=1000)
    thr = 10 ** 10.5 + zs * 0.0
    h.add_hod("g", mthresh=thr)
    blk = h.spectra_block(PAIRS)

    def step():
        h.init_mass_function(ms)
        h.add_nfw_profile("nfw", ignore_existing=True)
        h.add_battaglia_profile("electron", family="AGN", xmax=20, nxs=1000, ignore_existing=True)
        h.add_hod("g", mthresh=thr, ignore_existing=True)
        blk.compute()

    step()
    eager = {p: (a.copy(), b.copy()) for p, (a, b) in blk.fetch().items()}
    ctx = h._ctx()
    gid = ctx.capture(step)
    for _ in range(3):
        ctx.replay(gid)
    replayed = blk.fetch()
    for p in PAIRS:
        assert np.array_equal(replayed[p][0], eager[p][0]) and np.array_equal(replayed[p][1], eager[p][1]), p
    with pytest.raises(Exception):              # a step that would allocate cannot be captured
        ctx.capture(lambda: h.add_battaglia_profile("other", family="SH", xmax=20, nxs=1000))
    ctx.replay(gid)                             # ... and the failed capture leaves the context usable
    again = blk.fetch()
    assert np.array_equal(again[PAIRS[2]][1], eager[PAIRS[2]][1])
    ctx.call("hmg_graph_destroy", gid)


def test_no_block_changes_hands_inside_a_capture():
    """The allocator recycles blocks by size.  Inside a capture that would bake a recycled address into the
    graph while the block goes back to the list when its owner dies: EVERY allocation is refused there, also
    one the free list could serve, and a block freed during a capture stays out of the list as long as the
    graph that may use it exists."""
    from hmvec_amd import _native as nat
    ctx = nat.Context(0)
    a = ctx.empty((1234,))
    ptr_a = a.ptr
    a.free()                                    # a same-size block now sits in the free list
    keep = ctx.empty((777,))
    ptr_k = keep.ptr

    def body():
        with pytest.raises(nat.NativeError, match="captured step"):
            ctx.empty((1234,))                  # would have been served from the list
        keep.free()                             # deferred: not dropped, not recycled yet

    gid = ctx.capture(body)
    b = ctx.empty((1234,))
    assert b.ptr == ptr_a                       # outside the capture the list serves it again
    c = ctx.empty((777,))
    assert c.ptr != ptr_k                       # the block freed during the capture is still pinned to the graph
    ctx.call("hmg_graph_destroy", gid)
    c.free()
    d, e = ctx.empty((777,)), ctx.empty((777,))
    assert ptr_k in (d.ptr, e.ptr)              # ... and back in circulation once the graph is gone
    ctx.close()


def test_a_recorded_call_list_reissues_the_pass(monkeypatch):
    """bench.py's default launch mode (round 6): the native calls of a pass recorded once (Context.trace) and re-issued
    eagerly without the facade (Context.run_trace).  The list holds raw pointers: re-issued after the CONTENTS of an input
    changed in place it must give what a fresh eager pass gives on those contents, bit for bit; and it must hold exactly
    the three launches of a grouped pass."""
    import hmvec_amd as hm
    monkeypatch.setenv("HMG_NO_GROUPS", "0")      # (the launch count below is the grouped pass's)
    zs = np.linspace(0.1, 2.5, 6)
    ms = np.geomspace(2e10, 1e17, 96)
    ks = np.geomspace(1e-4, 100, 384)
    h = hm.HaloModel(zs, ks, ms=ms, accuracy="low", engine="analytic")
    h.add_battaglia_profile("electron", family="AGN", xmax=20, nxs=1000)
    thr = 10 ** 10.5 + zs * 0.0
    h.add_hod("g", mthresh=thr)
    blk = h.spectra_block(PAIRS)

    def step():
        h.init_mass_function(ms)
        h.add_nfw_profile("nfw", ignore_existing=True)
        h.add_battaglia_profile("electron", family="AGN", xmax=20, nxs=1000, ignore_existing=True)
        h.add_hod("g", mthresh=thr, ignore_existing=True)
        blk.compute()

    step()
    eager = {p: (a.copy(), b.copy()) for p, (a, b) in blk.fetch().items()}
    ctx = h._ctx()
    calls = ctx.trace(step)
    launches = [name for name, _ in calls if name in ("hmg_sigma2_halo_front", "hmg_group_rows", "hmg_group_profile",
                                                       "hmg_group_tensors", "hmg_power_batch_run")]
    assert launches == ["hmg_sigma2_halo_front", "hmg_group_tensors", "hmg_power_batch_run"], calls
    for _ in range(3):
        ctx.run_trace(calls)
    got = blk.fetch()
    for p in PAIRS:
        assert np.array_equal(got[p][0], eager[p][0]) and np.array_equal(got[p][1], eager[p][1]), p
    # new contents in the SAME device buffer (the thresholds the occupations read): the list sees them
    key = [k for k in h._dcache if isinstance(k, tuple) and k[0] == "thr"][0]
    host_thr, d_thr = h._dcache[key]
    ctx.write(d_thr, host_thr + 0.25)
    ctx.run_trace(calls)
    listed = {p: (a.copy(), b.copy()) for p, (a, b) in blk.fetch().items()}
    assert not np.array_equal(listed[PAIRS[2]][0], eager[PAIRS[2]][0])
    h2 = hm.HaloModel(zs, ks, ms=ms, accuracy="low", engine="analytic")
    h2.add_battaglia_profile("electron", family="AGN", xmax=20, nxs=1000)
    h2.add_hod("g", mthresh=thr * 10 ** 0.25)
    for p in PAIRS:
        assert np.allclose(listed[p][0], h2.get_power_1halo(*p), rtol=1e-11, atol=0), p
        assert np.allclose(listed[p][1], h2.get_power_2halo(*p), rtol=1e-11, atol=0), p
