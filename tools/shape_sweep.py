"""Shape sweep: milliseconds per stage of one pass for the shapes beside Config 3 that users of
the reference reach - other radial grids (compile-time plan vs run-time plan vs pruned long-grid route), a
pressure + tSZ batch, an HOD with a central profile (generic mass-integral forms), the README grid.
HIP-event times (hmg_bracket_next brackets around the profile stage and the mass integrals; events around the
whole pass), eager launches, median of `reps` passes after warm-up.
Usage (GPU box): python tools/shape_sweep.py > profiles/rNN/shape_sweep.txt"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REPS = 15
EV = dict(p0=40, p1=41, f0=44, f1=45, a=50, b=51)


def run_case(label, zs, ms, ks, profile, pairs, env=None, central=False, pressure=None, reps=REPS):
    """profile: (nxs, xmax) of the Battaglia gas profile; pressure: (nxs, xmax) or None."""
    for k, v in (env or {}).items():
        os.environ[k] = v
    try:
        import hmvec_amd as hm
        from hmvec_amd import _native as nat
        ctx = nat.Context(0)                       # env switches are read when a context is created
        h = hm.HaloModel(zs, ks, ms=ms, accuracy="low", engine="analytic", ctx=ctx)
        nxs, xmax = profile
        mthr = 10 ** 10.5 + zs * 0.0

        def one_pass(timed):
            if timed:
                ctx.record(EV["a"])
            h.init_mass_function(ms)
            h.add_nfw_profile("nfw", ignore_existing=True)
            if timed:
                ctx.call("hmg_bracket_next", nat.KERNEL_PROFILE_FFT, EV["f0"], EV["f1"])
            h.add_battaglia_profile("electron", family="AGN", xmax=xmax, nxs=nxs, ignore_existing=True)
            if pressure is not None:
                h.add_battaglia_pres_profile("y", xmax=pressure[1], nxs=pressure[0], ignore_existing=True)
            h.add_hod("g", mthresh=mthr, ignore_existing=True,
                      central_profile_name="electron" if central else None)
            if timed:
                ctx.call("hmg_bracket_next", nat.KERNEL_POWER, EV["p0"], EV["p1"])
            o1, o2 = h.power_device_batch(pairs)
            if timed:
                ctx.record(EV["b"])
            return o1, o2

        for _ in range(3):
            one_pass(False)
        ctx.sync()
        t = []
        for _ in range(reps):
            one_pass(True)
            ctx.sync()
            t.append((ctx.elapsed_ms(EV["a"], EV["b"]), ctx.elapsed_ms(EV["f0"], EV["f1"]), ctx.elapsed_ms(EV["p0"], EV["p1"])))
        t = np.median(np.array(t), axis=0)
        print(f"{label:74s} pass {t[0]:7.3f}  profile stage {t[1]:7.3f}  mass integrals {t[2]:7.3f}   ms", flush=True)
        ctx.close()
    finally:
        for k in (env or {}):
            os.environ.pop(k, None)


def table_case(label, zs, ms, ks, nxs, xmax, env=None, reps=7):
    """generic_profile_fft with a user's callable (hmvec/fft.py:56-94) at the size of the bench grid: HIP-event time of
    hmg_profile_fft_table on a per-row table that is already resident (the profile evaluated by the user on the host
    and its upload are not the path), next to the built-in family through hmg_profile_fft on the same grid."""
    for k, v in (env or {}).items():
        os.environ[k] = v
    try:
        from hmvec_amd import _native as nat
        ctx = nat.Context(0)
        nz, nm, nk = zs.size, ms.size, ks.size
        xs = np.linspace(0.0, xmax, nxs + 1)[1:]
        step = (xs[-1] - xs[0]) / nxs
        kts = np.fft.rfftfreq(nxs, step) * 2 * np.pi
        cmax = np.linspace(1.9, 2.9, nz * nm).reshape(nz, nm)
        rss = np.geomspace(0.03, 2.5, nm)[None, :] * (1.0 + 0.0 * zs[:, None])
        d_xs, d_kts, d_cmax, d_rss, d_zs, d_ks = (ctx.upload(a) for a in (xs, kts, cmax, np.ascontiguousarray(rss), zs, ks))
        out = ctx.empty((nz, nm, nk))
        d_rho = ctx.empty((nz * nm, nxs))
        row = (xs / 0.5) ** -0.2 * (1.0 + (xs / 0.5) ** 1.1) ** -2.6
        d_row = ctx.upload(row)
        for r in range(nz * nm):                       # (device-side replication: the table's contents do not matter for time)
            ctx.call("hmg_memcpy_d2d", d_rho.ptr + r * nxs * 8, d_row.ptr, nxs * 8)
        t_tab, t_fam = [], []
        for i in range(reps + 2):
            ctx.call("hmg_bracket_next", nat.KERNEL_PROFILE_FFT, EV["f0"], EV["f1"])
            ctx.call("hmg_profile_fft_table", nz, nm, nk, nxs, step, d_xs.ptr, d_kts.ptr, d_rho.ptr, nz * nm, d_cmax.ptr,
                     d_rss.ptr, d_zs.ptr, d_ks.ptr, 1, out.ptr)
            ctx.sync()
            if i >= 2:
                t_tab.append(ctx.elapsed_ms(EV["f0"], EV["f1"]))
            ctx.call("hmg_bracket_next", nat.KERNEL_PROFILE_FFT, EV["f0"], EV["f1"])
            ctx.call("hmg_profile_fft", nz, nm, nk, nxs, step, d_xs.ptr, d_kts.ptr, None, None, None, None,
                     1.0, 0.5, 1.1, 2.6, -0.2, d_cmax.ptr, d_rss.ptr, d_zs.ptr, d_ks.ptr, 1, None, out.ptr, None, None, None)
            ctx.sync()
            if i >= 2:
                t_fam.append(ctx.elapsed_ms(EV["f0"], EV["f1"]))
        print(f"{label:74s} table {np.median(t_tab):7.3f}  built-in family {np.median(t_fam):7.3f}   ms (profile stage)", flush=True)
        ctx.close()
    finally:
        for k in (env or {}):
            os.environ.pop(k, None)


def main():
    zs = np.linspace(0.01, 3.0, 32); ms = np.geomspace(2e10, 1e17, 512); ks = np.geomspace(1e-4, 100, 4096)
    six = [("nfw", "nfw"), ("electron", "electron"), ("g", "g"), ("nfw", "electron"), ("g", "nfw"), ("g", "electron")]
    print("# Config-3 grid 32 x 512 x 4096, six spectra, eager launches (a pass as a HIP graph is ~0.02 ms shorter)")
    print("# 'profile stage' = the launch that writes the Battaglia tensor; for one-row lengths with a compile-time plan that is the")
    print("# tensor group, which also writes the analytic NFW tensor (~0.14 ms of it); other routes: the profile rows alone")
    run_case("Config 3: nxs=5000 xmax=20 (compile-time plan 2500)", zs, ms, ks, (5000, 20), six)
    for nxs in (1000, 2000, 4000):
        run_case(f"nxs={nxs} xmax=20: compile-time plan M={nxs // 2}", zs, ms, ks, (nxs, 20), six)
        run_case(f"nxs={nxs} xmax=20: run-time plan (HMG_FUSED_GENERIC=1)", zs, ms, ks, (nxs, 20), six,
                 env={"HMG_FUSED_GENERIC": "1"})
    run_case("nxs=3000 xmax=20: compile-time plan M=1500", zs, ms, ks, (3000, 20), six)
    run_case("nxs=3000 xmax=20: run-time plan (HMG_FUSED_GENERIC=1)", zs, ms, ks, (3000, 20), six, env={"HMG_FUSED_GENERIC": "1"})
    run_case("nxs=10000 xmax=20: long-grid route (LP=1000, R=5; one 80-KB row in LDS measured slower)", zs, ms, ks,
             (10000, 20), six)
    run_case("nxs=10000 xmax=20: rocFFT route (HMG_PRUNED_FFT=0)", zs, ms, ks, (10000, 20), six,
             env={"HMG_PRUNED_FFT": "0"}, reps=5)
    run_case("nxs=30000 xmax=50: long-grid route (LP=1000, R=15; chirp route for rows with jn <= 590)", zs, ms, ks,
             (30000, 50), six)
    run_case("nxs=30000 xmax=50: long-grid route without the chirp route (HMG_CHIRP=0)", zs, ms, ks, (30000, 50), six,
             env={"HMG_CHIRP": "0"})
    run_case("nxs=30000 xmax=50: rocFFT route (HMG_PRUNED_FFT=0)", zs, ms, ks, (30000, 50), six,
             env={"HMG_PRUNED_FFT": "0"}, reps=5)
    names = ["nfw", "electron", "g", "y"]
    ten = [(a, b) for i, a in enumerate(names) for b in names[i:]]
    run_case("pressure + tSZ batch: + add_battaglia_pres_profile(nxs=5000, xmax=5), 10 spectra", zs, ms, ks, (5000, 20),
             ten, pressure=(5000, 5))
    run_case("HOD with central_profile_name='electron' (generic mass-integral forms), six spectra", zs, ms, ks,
             (5000, 20), six, central=True)
    run_case("tSZ notebook's pressure profile (nxs=30000, xmax=2) + 10 spectra: narrow-band route", zs, ms, ks, (5000, 20),
             ten, pressure=(30000, 2))
    run_case("the same through rocFFT (HMG_BAND_FFT=0)", zs, ms, ks, (5000, 20), ten, pressure=(30000, 2),
             env={"HMG_BAND_FFT": "0"}, reps=5)
    print("# generic_profile_fft with a user's callable (per-row table resident in HBM), Config-3 grid")
    table_case("table nxs=5000 xmax=20: one-row kernel (table build of the 2500 plan)", zs, ms, ks, 5000, 20.0)
    table_case("table nxs=5000 xmax=20: table -> rocFFT chain (HMG_FUSED_FFT=0)", zs, ms, ks, 5000, 20.0, env={"HMG_FUSED_FFT": "0"}, reps=3)
    table_case("table nxs=3000 xmax=20: one-row kernel, run-time plan", zs, ms, ks, 3000, 20.0)
    table_case("table nxs=30000 xmax=50: long-grid kernel", zs, ms, ks, 30000, 50.0, reps=3)
    print("# README grid 20 x 200 x 1001 (BASELINE configs 1/2), six spectra")
    zs = np.linspace(0., 3., 20); ms = np.geomspace(2e10, 1e17, 200); ks = np.geomspace(1e-4, 100, 1001)
    run_case("README grid: nxs=5000 xmax=20", zs, ms, ks, (5000, 20), six)
    run_case("README grid: nxs=30000 xmax=50 (examples/lensing_baryons.py:27), pruned route", zs, ms, ks, (30000, 50), six)


if __name__ == "__main__":
    main()
