"""Grouped launches (include/hmgrid.h "grouped launches"): independent stages of a pass
as block ranges of one grid.  Every role runs the device function of its stand-alone kernel, so the results
must equal the one-launch-per-stage path BIT FOR BIT - state arrays, tensors, hints and spectra - on full
grids, thin z-slabs, ragged mass grids, both mass functions, and whatever order the stages are queued in."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PAIRS = [("nfw", "nfw"), ("electron", "electron"), ("g", "g"), ("nfw", "electron"), ("g", "nfw"), ("g", "electron")]


def build(monkeypatch, grouped, zs, ms, ks, nxs=1000, mass_function="sheth-torman", pressure=False, corr="max", ctx=None):
    import hmvec_amd as hm
    monkeypatch.setenv("HMG_NO_GROUPS", "0" if grouped else "1")
    h = hm.HaloModel(zs, ks, ms=ms, accuracy="low", engine="analytic", mass_function=mass_function, ctx=ctx)
    assert h._groups == grouped
    h.add_battaglia_profile("electron", family="AGN", xmax=20, nxs=nxs)
    if pressure:
        h.add_battaglia_pres_profile("y", family="pres", xmax=5, nxs=nxs)
    h.add_hod("g", mthresh=10 ** 10.5 + zs * 0.0, corr=corr)
    return h


def state(h, pairs=PAIRS):
    out = {"sigma2": h.sigma2, "nzm": h.nzm, "bh": h.bh, "cs": h._d_cs.numpy(), "rvir": h._d_rvir.numpy(),
           "uk_nfw": h.uk_profiles["nfw"], "uk_e": h.uk_profiles["electron"]}
    for k in ("Nc", "Ns", "NsNsm1", "NcNs", "ngal", "bg"):
        out["hod_" + k] = h.hods["g"][k]
    hint = h.uk_profiles.hint("electron")
    if hint[0] is not None:
        out["nconst"] = hint[0].numpy().view(np.int32)[:h.zs.size * h.ms.size]
        out["cconst"] = hint[1].numpy()
    return out


def assert_same(a, b):
    assert a.keys() == b.keys()
    for k in a:
        assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("nz,nm,nk,mf", [(3, 48, 96, "sheth-torman"), (4, 200, 257, "tinker"), (5, 77, 130, "sheth-torman"),
                                         (2, 300, 64, "sheth-torman")])
def test_grouped_pass_equals_one_launch_per_stage(monkeypatch, nz, nm, nk, mf):
    zs = np.linspace(0.1, 2.9, nz)
    ms = np.geomspace(2e10, 1e17, nm)
    ks = np.geomspace(1e-4, 100, nk)
    e = build(monkeypatch, False, zs, ms, ks, mass_function=mf)
    e1, e2 = e.power_device_batch(PAIRS)
    g = build(monkeypatch, True, zs, ms, ks, mass_function=mf)
    # spectra first: on the grouped model this is what issues the queue (chain + rows + NFW, then the
    # coefficient rows of the batch beside the profile FFT)
    assert len(g._stages) > 0
    g1, g2 = g.power_device_batch(PAIRS)
    assert g._stages == []
    for p, a, b, c, d in zip(PAIRS, g1, g2, e1, e2):
        assert np.array_equal(a.numpy(), c.numpy()), p
        assert np.array_equal(b.numpy(), d.numpy()), p
    assert_same(state(g), state(e))


@pytest.mark.parametrize("nz,nm", [(3, 64), (20, 100), (2, 96), (8, 128), (17, 513)])
def test_hod_right_behind_the_constructor_rides_with_the_front(monkeypatch, nz, nm):
    """Constructor + add_hod with nothing read in between: one front launch (contraction | halo stage | HOD
    occupations), one rows group (n, b tiles | NFW rows; hmg_group_rows rejects a mass function and an HOD
    together) and the n_gal, b_g sums as the per-z chain of hmg_group_profile.  On fresh buffers, so that
    a sum taken from stale memory cannot pass (a repeated pass would hide it: the stale values are the right ones)."""
    import hmvec_amd as hm
    zs = np.linspace(0.01, 2.0, nz)
    ms = np.geomspace(2e10, 1e17, nm)
    ks = np.geomspace(1e-4, 100, 70)
    thr = 10 ** 10.5 + zs * 0.0
    out = []
    for grouped in (False, True):
        monkeypatch.setenv("HMG_NO_GROUPS", "0" if grouped else "1")
        h = hm.HaloModel(zs, ks, ms=ms, accuracy="low", engine="analytic")
        h.add_hod("g", mthresh=thr)
        if grouped:
            assert [s[0] for s in h._stages] == ["front", "massfn", "nfw", "hod"]
        out.append({k: h.hods["g"][k] for k in ("ngal", "bg", "Nc", "Ns", "NsNsm1", "NcNs")} | {"nzm": h.nzm, "bh": h.bh})
    assert_same(out[0], out[1])
    assert np.all(out[1]["ngal"] > 0)


def test_reads_issue_the_queue_in_any_order(monkeypatch):
    """Reading any piece of state first (instead of asking for spectra) must give the same numbers."""
    zs = np.array([0.0, 0.6, 1.5, 3.0])
    ms = np.geomspace(2e10, 1e17, 64)
    ks = np.geomspace(1e-4, 100, 80)
    e = build(monkeypatch, False, zs, ms, ks, pressure=True, corr="min")
    want = state(e)
    want_y = e.pk_profiles["y"]
    for first in ("hod_ngal", "uk_e", "nzm", "uk_nfw", "cs"):
        g = build(monkeypatch, True, zs, ms, ks, pressure=True, corr="min")
        got_first = state(g)[first] if first != "hod_ngal" else g.hods["g"]["ngal"]
        assert np.array_equal(got_first, want[first])
        assert_same(state(g), want)
        assert np.array_equal(g.pk_profiles["y"], want_y)
        for a, b in (("y", "y"), ("g", "y"), ("nfw", "y")):
            assert np.array_equal(g.get_power(a, b), e.get_power(a, b)), (a, b)


def test_requeueing_a_stage_issues_the_earlier_one_first(monkeypatch):
    """The same stage twice (a pass repeated without reading anything) and a producer queued behind its consumer."""
    zs = np.array([0.3, 1.1])
    ms = np.geomspace(2e10, 1e17, 96)
    ks = np.geomspace(1e-4, 100, 64)
    g = build(monkeypatch, True, zs, ms, ks)
    e = build(monkeypatch, False, zs, ms, ks)
    for h in (g, e):
        for _ in range(2):
            h.init_mass_function(ms)
            h.add_nfw_profile("nfw", ignore_existing=True)
            h.add_battaglia_profile("electron", family="AGN", xmax=20, nxs=1000, ignore_existing=True)
            h.add_hod("g", mthresh=10 ** 10.8 + zs * 0.0, ignore_existing=True)
        # a second mass function behind a queued HOD: the HOD must still see the n, b of before
        h.add_hod("g2", mthresh=10 ** 11.0 + zs * 0.0)
        h.init_mass_function(ms)
    assert_same(state(g), state(e))
    assert np.array_equal(g.hods["g2"]["ngal"], e.hods["g2"]["ngal"])
    assert np.array_equal(g.get_power("g", "g2"), e.get_power("g", "g2"))


def test_a_captured_pass_with_groups_replays_identically(monkeypatch):
    zs = np.linspace(0.2, 2.0, 4)
    ms = np.geomspace(2e10, 1e17, 128)
    ks = np.geomspace(1e-4, 100, 256)
    g = build(monkeypatch, True, zs, ms, ks)
    e = build(monkeypatch, False, zs, ms, ks)
    ctx = g._ctx()
    blk = g.spectra_block(PAIRS)
    thr = 10 ** 10.5 + zs * 0.0

    def one_pass():
        g.init_mass_function(ms)
        g.add_nfw_profile("nfw", ignore_existing=True)
        g.add_battaglia_profile("electron", family="AGN", xmax=20, nxs=1000, ignore_existing=True)
        g.add_hod("g", mthresh=thr, ignore_existing=True)
        blk.compute()

    one_pass()
    ctx.sync()
    gid = ctx.capture(one_pass)
    assert g._stages == []
    for _ in range(3):
        ctx.replay(gid)
    got = blk.fetch()
    e1, e2 = e.power_device_batch(PAIRS)
    for p, a, b in zip(PAIRS, e1, e2):
        assert np.array_equal(got[p][0], a.numpy()), p
        assert np.array_equal(got[p][1], b.numpy()), p


def test_group_entry_points_reject_bad_arguments():
    from hmvec_amd import _native as nat
    ctx = nat.default_context()
    lib = ctx.lib
    assert lib.hmg_group_rows(ctx.handle, 2, 8, 8, 16, None, None, None, None) != 0
    assert b"empty group" in lib.hmg_last_error()
    d = ctx.empty((64,))
    par = nat.MassFnParams(mode=0, deltac=1.686, st_A=0.3, st_a=0.7, st_p=0.3, rho_m0=1.0, lnm_uniform=0, lnm_step=0.0)
    mf = nat.MassFnPart(C.pointer(par), d.ptr, d.ptr, None, d.ptr, d.ptr, d.ptr)
    # no contraction of this shape has left partial sums in the context
    assert lib.hmg_group_rows(ctx.handle, 3, 7, 8, 12345, C.byref(mf), None, None, None) != 0
    assert b"hmg_sigma2_halo_front" in lib.hmg_last_error()
    nfw = nat.NfwPart(d.ptr, d.ptr, d.ptr, d.ptr, None, d.ptr)
    assert lib.hmg_group_rows(ctx.handle, 2, 4, 8, 16, None, None, None, C.byref(nfw)) != 0
    assert b"NFW part" in lib.hmg_last_error()
    hp = nat.HodParams(0.2, 1.0, 10.0, 1.0, 1.0, 1.0, 0)
    hod = nat.HodPart(nat.HOD_ALL, C.pointer(hp), *([d.ptr] * 12))
    assert lib.hmg_group_rows(ctx.handle, 3, 7, 8, 16, C.byref(mf), C.byref(hod), None, None) != 0
    assert b"cannot share its launch" in lib.hmg_last_error()
    assert lib.hmg_group_rows(ctx.handle, 3, 7, 8, 16, None, C.byref(hod), None, None) != 0
    assert b"sums of an HOD" in lib.hmg_last_error()
    assert lib.hmg_group_profile(ctx.handle, 2, 4, 8, None, None, None) != 0
    assert lib.hmg_power_batch_run(ctx.handle, 2, 4, 8, None, 0) != 0


@pytest.mark.parametrize("nm_new", [150, 40])
def test_queue_built_for_one_mass_grid_is_issued_before_the_grid_changes(monkeypatch, nm_new):
    """the constructor queues front, massfn and nfw with raw pointers into (nz, nm) buffers.
    init_mass_function(ms2) with another length, and no read in between, must issue that queue with the sizes it
    was built for BEFORE the inputs are released and the pool buffers replaced - with the model's current nm
    the stale pass would write nz*nm_new(*nk) elements into nz*nm_old(*nk) allocations."""
    import hmvec_amd as hm
    zs = np.array([0.2, 0.9, 1.7])
    ks = np.geomspace(1e-4, 100, 70)
    ms1, ms2 = np.geomspace(2e10, 1e17, 64), np.geomspace(1e11, 5e16, nm_new)
    out = []
    for grouped in (False, True):
        monkeypatch.setenv("HMG_NO_GROUPS", "0" if grouped else "1")
        h = hm.HaloModel(zs, ks, ms=ms1, accuracy="low", engine="analytic")
        if grouped:
            assert [s[0] for s in h._stages] == ["front", "massfn", "nfw"] and h._stage_dims[1] == 64
        h.init_mass_function(ms2)              # nothing was read: the first queue is still pending here
        if grouped:
            assert h._stage_dims[1] == nm_new
        h.add_nfw_profile("nfw", ignore_existing=True)
        h.add_battaglia_profile("electron", family="AGN", xmax=20, nxs=1000)
        h.add_hod("g", mthresh=10 ** 10.5 + zs * 0.0)
        st = state(h)
        assert st["uk_nfw"].shape == (3, nm_new, 70) and st["nzm"].shape == (3, nm_new)
        st["P"] = h.get_power("g", "electron")
        out.append(st)
    assert_same(out[0], out[1])
    # and a fresh model on the second grid gives the same numbers as the re-initialised one
    monkeypatch.setenv("HMG_NO_GROUPS", "0")
    f = build(monkeypatch, True, zs, ms2, ks)
    ref = state(f)
    ref["P"] = f.get_power("g", "electron")
    assert_same(out[1], ref)


@pytest.mark.parametrize("nxs,xmax", [(1000, 20.0), (5000, 20.0), (3000, 7.5)])
def test_row_scalars_left_by_the_rows_stage_equal_what_a_row_workgroup_works_out(default_routes, nxs, xmax):
    """ABI 8 (hmg_rows_part.d_rowsc): the thread that computes a row's length scale also leaves the output-side scalars
    of the row's transform - 1/(r(1+z)), k_lo, k_hi, 1/k_lo, 1/kt_1, reachable modes, left-fill count (hmvec/fft.py:96-107) -
    and the row kernel reads them instead of having one wavefront divide and search.  The record is checked against
    numpy, and gas and pressure tensors, hints and spectra against the path without it (HMG_NO_ROWSC=1), bit for bit."""
    import hmvec_amd as hm
    from hmvec_amd import _native as nat
    monkeypatch = default_routes              # (this test is ABOUT the grouped route with hints: pin it under tools/env_matrix.sh)
    monkeypatch.setenv("HMG_NO_GROUPS", "0")
    zs = np.linspace(0.05, 2.5, 4)
    ms = np.geomspace(2e10, 1e17, 90)
    ks = np.geomspace(1e-4, 100, 300)
    got = {}
    for off in ("1", "0"):
        monkeypatch.setenv("HMG_NO_ROWSC", off)
        h = hm.HaloModel(zs, ks, ms=ms, accuracy="low", engine="analytic")
        h.add_battaglia_profile("electron", family="AGN", xmax=xmax, nxs=nxs)
        h.add_battaglia_pres_profile("y", family="pres", xmax=xmax, nxs=nxs)
        p1, p2 = h.power_device_batch([("electron", "electron"), ("y", "y"), ("nfw", "y")])
        hint = h.uk_profiles.hint("electron")
        got[off] = dict(uk=h.uk_profiles["electron"], pk=h.pk_profiles["y"],
                        n=hint[0].numpy().view(np.int32)[:zs.size * ms.size].copy(), c=hint[1].numpy(),
                        P=[a.numpy() for a in p1] + [a.numpy() for a in p2])
        rec = h._pool.get((("uk", "electron"), "rowsc"))
        assert (rec is None) == (off == "1")
        if rec is not None:
            r = rec.numpy().reshape(zs.size * ms.size, nat.ROWSC_STRIDE)
            rscale = h._pool[(("uk", "electron"), "rowp", 5)].numpy().reshape(-1)
            z1 = np.repeat(1.0 + zs, ms.size)
            xs = np.linspace(0.0, xmax, nxs + 1)[1:]
            kts = np.fft.rfftfreq(xs.size, (xs[-1] - xs[0]) / xs.size) * 2 * np.pi
            isc = 1.0 / (rscale * z1)
            assert np.array_equal(r[:, 0], isc) and np.array_equal(r[:, 1], kts[1] * isc)
            assert np.array_equal(r[:, 2], kts[nxs // 2] * isc) and np.array_equal(r[:, 3], 1.0 / (kts[1] * isc))
            packed = r[:, 5].copy().view(np.int32).reshape(-1, 2)          # little endian: low word first
            nleft = np.array([np.searchsorted(ks, klo, side="left") for klo in kts[1] * isc])
            assert np.array_equal(packed[:, 0], nleft) and np.array_equal(packed[:, 0], got[off]["n"])
            tmax = ks[-1] * r[:, 3]                                    # the kernel's own product
            assert np.all(packed[:, 1] == np.where(tmax < nxs // 2 - 4, tmax.astype(int) + 3, nxs // 2))
    for k in ("uk", "pk", "n", "c"):
        assert np.array_equal(got["0"][k], got["1"][k]), k
    for a, b in zip(got["0"]["P"], got["1"]["P"]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("nz,nm,nk,mf", [(4, 512, 130, "sheth-torman"), (3, 62, 96, "sheth-torman"), (2, 63, 70, "tinker"),
                                         (5, 125, 257, "sheth-torman"), (1, 513, 64, "tinker"), (6, 187, 64, "sheth-torman"),
                                         (2, 1100, 40, "sheth-torman"), (2, 1022, 40, "tinker")])
def test_tensor_group_equals_the_two_groups_and_the_separate_launches(default_routes, monkeypatch, nz, nm, nk, mf):
    """hmg_group_tensors: sigma^2 -> n, b as the first link of the per-z chain (all masses of a redshift by the chain's own
    workgroup, one per thread) | profile rows | NFW rows in ONE launch, against the rows group followed by the profile
    group (HMG_NO_TENSOR_GROUP=1: the same entry point, two launches) and against one launch per stage - state arrays,
    tensors, hints and spectra bit for bit; mass grids shorter than, as long as and longer than a workgroup (513: two
    chunks with their stencil neighbours), lengths around the 62-mass tiles of the other path."""
    zs = np.linspace(0.1, 2.9, nz)
    ms = np.geomspace(2e10, 1e17, nm)
    ks = np.geomspace(1e-4, 100, nk)
    def one_pass(h):        # a pass repeated on a model that exists (the first one after the constructor reads tables in between)
        h.init_mass_function(ms)
        h.add_nfw_profile("nfw", ignore_existing=True)
        h.add_battaglia_profile("electron", family="AGN", xmax=20, nxs=1000, ignore_existing=True)
        h.add_hod("g", mthresh=10 ** 10.5 + zs * 0.0, ignore_existing=True)
        return h.power_device_batch(PAIRS)

    from hmvec_amd import _native as nat
    monkeypatch.delenv("HMG_NO_ROWSC", raising=False)       # (the suite may run under a switch: this test is about the default routes)
    ectx = nat.Context(0)                                   # (route switches are read when a context is created)
    e = build(monkeypatch, False, zs, ms, ks, mass_function=mf, ctx=ectx)
    e1, e2 = one_pass(e)
    want = state(e)
    for no_tensor in ("0", "1"):
        if no_tensor == "1":
            monkeypatch.setenv("HMG_NO_TENSOR_GROUP", "1")
        else:
            monkeypatch.delenv("HMG_NO_TENSOR_GROUP", raising=False)
        ctx, names = nat.Context(0), []                      # (the switch is read when a context is created)
        g = build(monkeypatch, True, zs, ms, ks, mass_function=mf, ctx=ctx)
        g.power_device_batch(PAIRS)
        issue = ctx.call_now
        monkeypatch.setattr(ctx, "call_now", lambda name, *a: (names.append(name), issue(name, *a))[1], raising=False)
        g1, g2 = one_pass(g)                             # issues the queue: front, tensor group, integrals
        monkeypatch.setattr(ctx, "call_now", issue, raising=False)
        assert [n for n in names if n.startswith("hmg_group") or n == "hmg_sigma2_halo_front"] == \
            ["hmg_sigma2_halo_front", "hmg_group_tensors"], names
        for p, a, b, c, d in zip(PAIRS, g1, g2, e1, e2):
            assert np.array_equal(a.numpy(), c.numpy()), (no_tensor, p)
            assert np.array_equal(b.numpy(), d.numpy()), (no_tensor, p)
        assert_same(state(g), want)
        blk = g.spectra_block(PAIRS)

        def captured_pass():
            g.init_mass_function(ms)
            g.add_nfw_profile("nfw", ignore_existing=True)
            g.add_battaglia_profile("electron", family="AGN", xmax=20, nxs=1000, ignore_existing=True)
            g.add_hod("g", mthresh=10 ** 10.5 + zs * 0.0, ignore_existing=True)
            blk.compute()
        captured_pass()
        gid = ctx.capture(captured_pass)                     # the kernels of a pass: front, tensor group, mass integrals
        assert ctx.graph_kernel_nodes(gid) == (4 if no_tensor == "1" else 3)
        ctx.call("hmg_graph_destroy", gid)
        del g
        ctx.close()
    monkeypatch.delenv("HMG_NO_TENSOR_GROUP", raising=False)
    del e
    ectx.close()


def test_tensor_group_rejects_bad_arguments_and_falls_back_for_other_routes(default_routes, monkeypatch):
    """No transform part: refused.  A radial grid the one-row transform does not take (nxs = 30000: long-grid route) behind the same
    call: the groups run one after the other and the results equal the separate launches."""
    from hmvec_amd import _native as nat
    ctx = nat.Context(0)
    assert ctx.lib.hmg_group_tensors(ctx.handle, 2, 8, 8, 16, None, None, None, None, None) != 0
    assert b"profile transform" in ctx.lib.hmg_last_error()
    ctx.close()
    zs = np.array([0.2, 1.4])
    ms = np.geomspace(2e10, 1e17, 40)
    ks = np.geomspace(1e-4, 100, 96)
    e = build(monkeypatch, False, zs, ms, ks, nxs=30000)
    g = build(monkeypatch, True, zs, ms, ks, nxs=30000)
    for p in PAIRS:
        assert np.array_equal(g.get_power(*p), e.get_power(*p)), p
    assert_same(state(g), state(e))
