"""``hmvec.fft`` mirror: the radial-profile sine transform and its interpolation onto a target
k grid (hmvec/fft.py), on the GPU.

``generic_profile_fft`` takes the reference's arguments, including an arbitrary Python callable
for the profile: the callable is evaluated once on the x grid (that is user code, on the host),
everything after it - truncation at cmax, trapezoid mass normalisation, the length-nxs real
FFT of every (z,m) row, the k scaling and the per-row interpolation that the reference does in a
Python double loop - runs in ``hmg_profile_fft_table``.  ``HaloModel.add_*_profile`` does not go
through here: for the NFW/Battaglia families the integrand is evaluated inside the fused kernel.
"""
import numpy as np

from .functions import _ctx, trapz_lastaxis


def fft_integral(x, y, axis=-1):
    """int dx x sin(kx) y(|x|) by FFT with the reference's conventions (hmvec/fft.py:35-51):
    step = (x[-1]-x[0])/N, 0-based phase.  Returns (ks, uk)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    assert x.ndim == 1
    y = np.asarray(y, dtype=np.float64)
    y = np.broadcast_to(y, np.broadcast_shapes(y.shape, x.shape))     # integrand = x*y aligns x with the last axis
    if axis not in (-1, y.ndim - 1):
        raise NotImplementedError("fft_integral: only the last axis (the reference's own usage) is supported")
    lead = y.shape[:-1]
    rows = int(np.prod(lead, dtype=np.int64)) if lead else 1
    N = x.size
    step = (x[-1] - x[0]) / N
    ks = np.fft.rfftfreq(N, step) * 2 * np.pi
    ctx = _ctx()
    d_x, d_y = ctx.upload(x), ctx.upload(np.ascontiguousarray(y).reshape(-1))
    out = ctx.empty((rows, N // 2 + 1))
    ctx.call("hmg_sine_transform", rows, N, d_x.ptr, d_y.ptr, out.ptr)
    return ks, out.numpy().reshape(lead + (N // 2 + 1,))


def uk_brute_force(r, rho, rvir, ks):
    """The same transform by direct quadrature, u(k) = (4 pi / m) int_0^rvir dr r sin(kr) rho(r) / k with
    m = 4 pi int_0^rvir dr r^2 rho (hmvec/fft.py:22-33; bin/tests.py:36 checks uk_fft against it).  r, rho:
    (nr,) samples of the profile, ks: (nk,).  The (nk, nr) integrand and its trapezoid sums are evaluated on
    the device (HMG_FN_BRUTE_INTEGRAND + hmg_trapz_rows)."""
    from .functions import FN_BRUTE_INTEGRAND, fn2d
    r, rho, ks = (np.asarray(a, dtype=np.float64) for a in (r, rho, ks))
    keep = r < rvir
    rs, rhos = r[keep], rho[keep]
    mass = 4.0 * np.pi * float(trapz_lastaxis((rhos * rs ** 2.0)[None, :], rs)[0])
    integrand = fn2d(FN_BRUTE_INTEGRAND, [rs[None, :], rhos[None, :], ks[:, None]])       # [k][r]
    return trapz_lastaxis(integrand, rs) / mass


def analytic_fft_integral(ks):
    """Closed form of fft_integral for y = exp(-x^2/2) (hmvec/fft.py:53); a check function."""
    ks = np.asarray(ks, dtype=np.float64)
    return np.sqrt(np.pi / 2.0) * np.exp(-ks ** 2.0 / 2.0) * ks


def generic_profile_fft(rhofunc_x, cmaxs, rss, zs, ks, xmax, nxs, do_mass_norm=True):
    """u(k|m,z) of a radial profile rho(x) truncated at x = cmax (hmvec/fft.py:56-94).

    rhofunc_x maps xs = linspace(0,xmax,nxs+1)[1:] to a (nxs,) or (nz,nm,nxs) array; cmaxs (nz,nm);
    rss (nz,nm,1) scale radii; zs (nz,).  Returns (ks, uk[nz,nm,nk])."""
    xs = np.linspace(0.0, xmax, nxs + 1)[1:]
    rhos = np.asarray(rhofunc_x(xs), dtype=np.float64)
    cmaxs = np.ascontiguousarray(cmaxs, dtype=np.float64)
    if rhos.ndim == 1:
        rho_rows = 1
    else:
        assert rhos.ndim == 3
        rho_rows = None
    nz, nm = cmaxs.shape
    if rho_rows is None:
        rhos = np.ascontiguousarray(np.broadcast_to(rhos, (nz, nm, xs.size)))
        rho_rows = nz * nm
    zs = np.ascontiguousarray(zs, dtype=np.float64).reshape(-1)
    ks = np.ascontiguousarray(ks, dtype=np.float64)
    rss = np.asarray(rss, dtype=np.float64)
    rss2 = np.ascontiguousarray(np.broadcast_to(rss[..., 0] if rss.ndim == 3 else rss, (nz, nm)))
    step = (xs[-1] - xs[0]) / xs.size
    kts = np.fft.rfftfreq(xs.size, step) * 2 * np.pi
    ctx = _ctx()
    d = [ctx.upload(a) for a in (xs, kts, rhos.reshape(-1), cmaxs.reshape(-1), rss2.reshape(-1), zs, ks)]
    out = ctx.empty((nz, nm, ks.size))
    ctx.call("hmg_profile_fft_table", nz, nm, ks.size, int(nxs), float(step), d[0].ptr, d[1].ptr, d[2].ptr,
             int(rho_rows), d[3].ptr, d[4].ptr, d[5].ptr, d[6].ptr, int(bool(do_mass_norm)), out.ptr)
    return ks, out.numpy()


def uk_fft(rhofunc, rvir, dr=0.001, rmax=100):
    """FFT of a truncated profile on r = arange(dr, rmax, dr), normalised by its mass
    (hmvec/fft.py:8-19); used by the reference's NFW cross-checks (bin/tests.py:45,147)."""
    rvir = np.asarray(rvir, dtype=np.float64)
    rs = np.arange(dr, rmax, dr)
    rhos = np.asarray(rhofunc(np.abs(rs)), dtype=np.float64)
    theta = np.ones(rhos.shape)
    theta[np.abs(rs) > rvir[..., None]] = 0
    integrand = rhos * theta
    m = trapz_lastaxis(integrand * rs ** 2.0, rs) * 4.0 * np.pi
    ks, ukt = fft_integral(rs, integrand)
    with np.errstate(divide="ignore", invalid="ignore"):
        uk = 4.0 * np.pi * ukt / ks / np.asarray(m)[..., None]
    return ks, uk


__all__ = ["fft_integral", "analytic_fft_integral", "generic_profile_fft", "uk_fft", "uk_brute_force"]
