"""Free-function mirrors of the reference's module-level helpers on the hot path.

The reference star-exports its building blocks (hmvec/__init__.py:1) and its own tests call
them by name (bin/tests.py:11,27,268-295): ``hmvec.rho_gas``, ``hmvec.R_from_M``,
``hmvec.duffy_concentration``, ``hmvec.mdelta_from_mdelta``, ``hmvec.avg_Nc`` ...  Each function
here keeps the reference's name, argument order, defaults, broadcasting and return shape; the
arithmetic runs on the GPU through ``hmg_fn2d`` / ``hmg_mstellar_halo`` / ``hmg_trapz_rows`` /
``hmg_mdelta_convert`` (include/hmgrid.h).  numpy arrays in, numpy arrays out.  There is no CPU
implementation behind these names: without libhmgrid the first call raises.

``HaloModel`` does not call these - its stages use the fused kernels - so they cost nothing on
the hot path; they exist so that code written against the reference keeps working.
"""
import contextlib
import ctypes as C

import numpy as np
import scipy.constants as constants

from . import _native as nat
from .params import battaglia_defaults, default_params

# hmg_fn2d function ids (include/hmgrid.h)
(FN_TINKER_BIAS, FN_TINKER_FNU, FN_MHALO_STELLAR, FN_HOD_NC, FN_HOD_NS, FN_HOD_MFUNC, FN_HOD_NSNSM1,
 FN_HOD_NCNS, FN_FCON, FN_RHO_NFW, FN_R_FROM_M, FN_DUFFY, FN_BATT_FIT, FN_RHO_GAS_X, FN_RHO_GAS_R,
 FN_PE_X, FN_PE_R, FN_NGAL_INTEGRAND, FN_A2Z, FN_MDELTA, FN_BG_INTEGRAND, FN_ST_FSIGMA, FN_TINKER_FSIGMA, FN_WKR,
 FN_LINCOMB3, FN_MHALO_STELLAR_CORE, FN_BRUTE_INTEGRAND) = range(27)

_ctx_override = None


def use_context(ctx):
    """Run the free functions on this Context (default: the process-wide device-0 context)."""
    global _ctx_override
    _ctx_override = ctx


@contextlib.contextmanager
def context(ctx):
    """``with functions.context(ctx):`` - run the free functions on ``ctx`` inside the block."""
    global _ctx_override
    prev, _ctx_override = _ctx_override, ctx
    try:
        yield ctx
    finally:
        _ctx_override = prev


def _ctx():
    return _ctx_override if _ctx_override is not None else nat.default_context(0)


def _operand(a, shape):
    """Classify one operand against the broadcast result shape -> (flat array, row stride, col stride)."""
    nd = len(shape)
    a = np.asarray(a, dtype=np.float64)
    a = a.reshape((1,) * (nd - a.ndim) + a.shape)
    cols = shape[-1]
    if a.size == 1:
        return a.reshape(1), 0, 0
    if a.shape[-1] == 1 and cols != 1:                        # constant along the last axis
        return np.ascontiguousarray(np.broadcast_to(a, shape[:-1] + (1,))).reshape(-1), 1, 0
    if all(s == 1 for s in a.shape[:-1]):                     # varies along the last axis only
        return np.ascontiguousarray(np.broadcast_to(a, (1,) * (nd - 1) + (cols,))).reshape(-1), 0, 1
    return np.ascontiguousarray(np.broadcast_to(a, shape)).reshape(-1), cols, 1


def fn2d(op, inputs, par=(), tables=()):
    """Evaluate function ``op`` over the numpy-broadcast of ``inputs`` on the device."""
    arrs = [np.asarray(a, dtype=np.float64) for a in inputs]
    shape = np.broadcast_shapes(*[a.shape for a in arrs])
    scalar = shape == ()
    full = (1,) if scalar else shape
    rows = int(np.prod(full[:-1], dtype=np.int64))
    cols = int(full[-1])
    if rows * cols == 0:
        return np.empty(shape, dtype=np.float64)
    ctx = _ctx()
    ops = [_operand(a, full) for a in arrs] + [(np.ascontiguousarray(t, dtype=np.float64).reshape(-1), 0, 0)
                                               for t in tables]
    dev = [ctx.upload(o[0]) for o in ops]
    n = len(ops)
    ptrs = (C.c_void_p * n)(*[d.ptr for d in dev])
    sr = (C.c_int * n)(*[o[1] for o in ops])
    sc = (C.c_int * n)(*[o[2] for o in ops])
    hp = (C.c_double * max(len(par), 1))(*[float(p) for p in par])
    out = ctx.empty((rows, cols))
    ctx.call("hmg_fn2d", int(op), rows, cols, n, ptrs, sr, sc, hp, len(par), out.ptr)
    res = out.numpy().reshape(full)
    return np.float64(res[0]) if scalar else res


# ------------------------------------------------------------------ halo structure (A5, A7)
def duffy_concentration(m, z, A=None, alpha=None, beta=None, h=None):
    """c = A (h m / 2e12)^alpha (1+z)^beta, mean-density Duffy set by default (hmvec/hmvec.py:68-73)."""
    A = default_params["duffy_A_mean"] if A is None else A
    alpha = default_params["duffy_alpha_mean"] if alpha is None else alpha
    beta = default_params["duffy_beta_mean"] if beta is None else beta
    h = default_params["H0"] / 100.0 if h is None else h
    return fn2d(FN_DUFFY, [m, z], [A, alpha, beta, h])


def R_from_M(M, rho, delta):
    """(3M / 4 pi delta rho)^(1/3)  (hmvec/hmvec.py:627-628)."""
    return fn2d(FN_R_FROM_M, [M, rho, delta])


def Fcon(c):
    """ln(1+c) - c/(1+c)  (hmvec/hmvec.py:737)."""
    return fn2d(FN_FCON, [c])


def rhoscale_nfw(mdelta, rdelta, cdelta):
    """hmvec/hmvec.py:739-742 reads a module global ``pref`` that the reference never defines, so
    the reference raises NameError on every call; the mirror does the same rather than guess."""
    raise NameError("name 'pref' is not defined")


def rho_nfw(r, rhoscale, rs):
    """rhoscale / x / (1+x)^2 at x = r/rs  (hmvec/hmvec.py:744-746)."""
    return fn2d(FN_RHO_NFW, [r, rhoscale, rs])


def rho_nfw_x(x, rhoscale):
    return fn2d(FN_RHO_NFW, [x, rhoscale, 1.0])


def a2z(a):
    return fn2d(FN_A2Z, [a])


def mdelta_from_mdelta(M1, C1, delta_rhos1, delta_rhos2, vectorized=True):
    """M1(m) -> M2(z,m) between two spherical-overdensity definitions assuming NFW
    (hmvec/hmvec.py:748-798).  M1 (nm,), C1 (nz,nm), delta_rhos* (nz,).  The reference finds the
    root of M1 F(c1) = M2 F(c2(M2)) in ln M2 by a secant iteration stopped at 1.5e-8; the device
    solver runs Newton to machine precision ("solve to 1e-14 with any method"), so
    ``vectorized`` changes nothing here and exists for signature parity."""
    M1 = np.asarray(M1, dtype=np.float64)
    d1 = np.asarray(delta_rhos1, dtype=np.float64)
    d2 = np.asarray(delta_rhos2, dtype=np.float64)
    return fn2d(FN_MDELTA, [M1[None, :], C1, d1[:, None], d2[:, None]])


def mdelta_from_mdelta_unvectorized(M1, C1, delta_rhos1, delta_rhos2):
    """Element-wise form (hmvec/hmvec.py:770-798): the four operands broadcast against each other."""
    return fn2d(FN_MDELTA, [M1, C1, delta_rhos1, delta_rhos2])


# ------------------------------------------------------------------ HOD (H1-H3)
def _z_column(z):
    z = np.asarray(z, dtype=np.float64)
    return z.reshape(-1, 1)


def Mhalo_stellar(z, log10mstellar):
    """log10 M_halo(log10 M*, z): Behroozi+10 table 2, the two parameter sets split at z = 0.8
    (hmvec/hmvec.py:648-695).  z (nz,1); log10mstellar (1,n) or (nz,n) -> (nz,n)."""
    z = _z_column(z)
    lms = np.asarray(log10mstellar, dtype=np.float64)
    lms = lms.reshape(-1, lms.shape[-1]) if lms.ndim else lms.reshape(1, 1)
    return fn2d(FN_MHALO_STELLAR, [z, lms]).reshape(z.size, lms.shape[-1])


def Mhalo_stellar_core(log10mstellar, a, Mstar00, Mstara, M1, M1a, beta0, beta_a, gamma0, gamma_a, delta0, delta_a):
    """The Behroozi+10 relation with explicit parameters (hmvec/hmvec.py:648-657); operands broadcast."""
    return fn2d(FN_MHALO_STELLAR_CORE, [log10mstellar, a],
                [Mstar00, Mstara, M1, M1a, beta0, beta_a, gamma0, gamma_a, delta0, delta_a])


def Mstellar_halo(z, log10mhalo):
    """Inverse of Mhalo_stellar through the reference's 4000-point table and np.interp
    (hmvec/hmvec.py:634-646).  z (nz,1), log10mhalo (1,nm) [row 0 is used for every z] -> (nz,nm)."""
    z = _z_column(z)
    lmh = np.asarray(log10mhalo, dtype=np.float64)
    lmh0 = np.ascontiguousarray(lmh[0] if lmh.ndim > 1 else lmh).reshape(-1)
    ctx = _ctx()
    d_z, d_l = ctx.upload(np.ascontiguousarray(z.reshape(-1))), ctx.upload(lmh0)
    out = ctx.empty((z.size, lmh0.size))
    ctx.call("hmg_mstellar_halo", z.size, lmh0.size, d_z.ptr, d_l.ptr, out.ptr)
    return out.numpy()


def avg_Nc(log10mhalo, z, log10mstellar_thresh, sig_log_mstellar):
    """<Nc(m)> = (1 - erf((log M*_thr - log M*(m)) / (sqrt2 sigma))) / 2  (hmvec/hmvec.py:698-703)."""
    log10mstar = Mstellar_halo(z, log10mhalo)
    return fn2d(FN_HOD_NC, [log10mstar, log10mstellar_thresh], [sig_log_mstellar])


def hod_default_mfunc(mthresh, Bamp, Bind):
    """1e12 B 10^((log10 M_thr - 12) beta)  (hmvec/hmvec.py:706)."""
    return fn2d(FN_HOD_MFUNC, [mthresh], [Bamp, Bind])


def avg_Ns(log10mhalo, z, log10mstellar_thresh, Nc=None, sig_log_mstellar=None, alphasat=None, Bsat=None,
           betasat=None, Bcut=None, betacut=None, Msat_override=None, Mcut_override=None):
    """<Ns(m)> = Nc (m/Msat)^alpha exp(-Mcut/m)  (hmvec/hmvec.py:708-716)."""
    mthresh = Mhalo_stellar(z, log10mstellar_thresh)
    Msat = Msat_override if Msat_override is not None else hod_default_mfunc(mthresh, Bsat, betasat)
    Mcut = Mcut_override if Mcut_override is not None else hod_default_mfunc(mthresh, Bcut, betacut)
    if Nc is None:
        Nc = avg_Nc(log10mhalo, z, log10mstellar_thresh, sig_log_mstellar=sig_log_mstellar)
    return fn2d(FN_HOD_NS, [Nc, log10mhalo, Msat, Mcut], [alphasat])


def _corr_id(corr):
    if corr == "max":
        return 0.0
    if corr == "min":
        return 1.0
    return None


def avg_NsNsm1(Nc, Ns, corr="max"):
    """hmvec/hmvec.py:719-725 (returns None for an unknown corr, as the reference falls through)."""
    cid = _corr_id(corr)
    return None if cid is None else fn2d(FN_HOD_NSNSM1, [Nc, Ns], [cid])


def avg_NcNs(Nc, Ns, corr="max"):
    """hmvec/hmvec.py:727-731."""
    cid = _corr_id(corr)
    return None if cid is None else fn2d(FN_HOD_NCNS, [Nc, Ns], [cid])


def trapz_lastaxis(y, x):
    """np.trapz(y, x, axis=-1) on the device."""
    y = np.ascontiguousarray(y, dtype=np.float64)
    x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1)
    if y.shape[-1] != x.size:
        raise ValueError("x must match the last axis of y")
    rows = int(np.prod(y.shape[:-1], dtype=np.int64))
    ctx = _ctx()
    d_y, d_x = ctx.upload(y.reshape(-1)), ctx.upload(x)
    out = ctx.empty((max(rows, 1),))
    ctx.call("hmg_trapz_rows", max(rows, 1), x.size, d_y.ptr, d_x.ptr, out.ptr)
    res = out.numpy()
    return res.reshape(y.shape[:-1]) if y.ndim > 1 else np.float64(res[0])


def ngal_from_mthresh(log10mthresh=None, zs=None, nzm=None, ms=None, sig_log_mstellar=None, Ncs=None, Nss=None,
                      alphasat=None, Bsat=None, betasat=None, Bcut=None, betacut=None, Msat_override=None,
                      Mcut_override=None):
    """n_gal(z) = int dm n(z,m) (Nc + Ns)  (hmvec/hmvec.py:936-957)."""
    if (Ncs is None) and (Nss is None):
        log10mstellar_thresh = np.asarray(log10mthresh)[:, None]
        log10mhalo = np.log10(np.asarray(ms, dtype=np.float64))[None, :]     # input transform of the mass grid
        Ncs = avg_Nc(log10mhalo, np.asarray(zs)[:, None], log10mstellar_thresh, sig_log_mstellar)
        Nss = avg_Ns(log10mhalo, np.asarray(zs)[:, None], log10mstellar_thresh, Ncs, sig_log_mstellar, alphasat,
                     Bsat, betasat, Bcut, betacut, Msat_override=Msat_override, Mcut_override=Mcut_override)
    else:
        assert log10mthresh is None
        assert zs is None
        assert sig_log_mstellar is None
    integrand = fn2d(FN_NGAL_INTEGRAND, [nzm, Ncs, Nss])
    return trapz_lastaxis(integrand, ms)


# ------------------------------------------------------------------ Battaglia profiles (A8, X1)
_GAS_KEYS = ("rho0_A0", "rho0_alpham", "rho0_alphaz", "alpha_A0", "alpha_alpham", "alpha_alphaz",
             "beta_A0", "beta_alpham", "beta_alphaz")
_PRES_KEYS = ("P0_A0", "P0_alpham", "P0_alphaz", "xc_A0", "xc_alpham", "xc_alphaz",
              "beta_A0", "beta_alpham", "beta_alphaz")
_GAS_DEF = battaglia_defaults[default_params["battaglia_gas_family"]]
_PRES_DEF = battaglia_defaults[default_params["battaglia_pres_family"]]


def battaglia_gas_fit(m200critz, z, A0x, alphamx, alphazx):
    """A0 (M200c/1e14)^alpha_m (1+z)^alpha_z  (hmvec/hmvec.py:800-802)."""
    return fn2d(FN_BATT_FIT, [m200critz, z], [A0x, alphamx, alphazx])


def _gas(op, x, m200critz, z, omb, omm, rhocritz, gamma, fit):
    return fn2d(op, [x, m200critz, z, rhocritz], [omb, omm, gamma] + [fit[k] for k in _GAS_KEYS])


def _gas_kwargs(kw, defaults):
    fit = dict(defaults)
    for k in list(kw):
        if k in fit:
            fit[k] = kw.pop(k)
    if kw:
        raise TypeError(f"unexpected keyword argument {sorted(kw)[0]!r}")
    return fit


def rho_gas_generic_x(x, m200critz, z, omb, omm, rhocritz, gamma=default_params["battaglia_gas_gamma"], *fitpos, **kw):
    """(Ob/Om) rho_c(z) rho0 x^gamma (1+x^alpha)^(-(beta+gamma)/alpha), x = r/(R200c/2)
    (hmvec/hmvec.py:844-860; the nine fit numbers follow positionally or by the reference's keywords)."""
    fit = _gas_kwargs(kw, _GAS_DEF)
    fit.update(dict(zip(_GAS_KEYS, fitpos)))
    return _gas(FN_RHO_GAS_X, x, m200critz, z, omb, omm, rhocritz, gamma, fit)


def rho_gas_generic(r, m200critz, z, omb, omm, rhocritz, gamma=default_params["battaglia_gas_gamma"], *fitpos, **kw):
    """Same at physical radius r: x = 2 r / R200c  (hmvec/hmvec.py:819-842)."""
    fit = _gas_kwargs(kw, _GAS_DEF)
    fit.update(dict(zip(_GAS_KEYS, fitpos)))
    return _gas(FN_RHO_GAS_R, r, m200critz, z, omb, omm, rhocritz, gamma, fit)


def rho_gas(r, m200critz, z, omb, omm, rhocritz, gamma=default_params["battaglia_gas_gamma"], profile="AGN"):
    """hmvec/hmvec.py:804-817."""
    return _gas(FN_RHO_GAS_R, r, m200critz, z, omb, omm, rhocritz, gamma, battaglia_defaults[profile])


def _G_newt():
    return constants.G / (default_params["parsec"] * 1e6) ** 3 * default_params["mSun"]


def P_e_generic_x(x, m200critz, R200critz, z, omb, omm, rhocritz, alpha=default_params["battaglia_pres_alpha"],
                  gamma=default_params["battaglia_pres_gamma"], *fitpos, **kw):
    """Electron pressure at x = r/R200c  (hmvec/hmvec.py:906-927)."""
    fit = _gas_kwargs(kw, battaglia_defaults["pres"])
    fit.update(dict(zip(_PRES_KEYS, fitpos)))
    par = [omb, omm, alpha, gamma] + [fit[k] for k in _PRES_KEYS] + [_G_newt()]
    return fn2d(FN_PE_X, [x, m200critz, R200critz, z, rhocritz], par)


def P_e_generic(r, m200critz, z, omb, omm, rhocritz, alpha=default_params["battaglia_pres_alpha"],
                gamma=default_params["battaglia_pres_gamma"], *fitpos, **kw):
    """hmvec/hmvec.py:881-904."""
    fit = _gas_kwargs(kw, _PRES_DEF)
    fit.update(dict(zip(_PRES_KEYS, fitpos)))
    par = [omb, omm, alpha, gamma] + [fit[k] for k in _PRES_KEYS] + [_G_newt()]
    return fn2d(FN_PE_R, [r, m200critz, z, rhocritz], par)


def P_e(r, m200critz, z, omb, omm, rhocritz, alpha=default_params["battaglia_pres_alpha"],
        gamma=default_params["battaglia_pres_gamma"], profile="pres"):
    """hmvec/hmvec.py:864-879."""
    fit = battaglia_defaults[profile]
    par = [omb, omm, alpha, gamma] + [fit[k] for k in _PRES_KEYS] + [_G_newt()]
    return fn2d(FN_PE_R, [r, m200critz, z, rhocritz], par)


__all__ = ["duffy_concentration", "R_from_M", "Fcon", "rhoscale_nfw", "rho_nfw", "rho_nfw_x", "a2z",
           "mdelta_from_mdelta", "mdelta_from_mdelta_unvectorized", "Mhalo_stellar", "Mhalo_stellar_core",
           "Mstellar_halo", "avg_Nc",
           "avg_Ns", "avg_NsNsm1", "avg_NcNs", "hod_default_mfunc", "ngal_from_mthresh", "battaglia_gas_fit",
           "rho_gas", "rho_gas_generic", "rho_gas_generic_x", "P_e", "P_e_generic", "P_e_generic_x"]
