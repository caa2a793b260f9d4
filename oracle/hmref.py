"""ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product.

A CPU/numpy restatement of the reference's (simonsobs/hmvec) halo-model hot
path, used ONLY by ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` as the checker for the HIP kernels.  Nothing
under ``hmvec_amd/`` imports it.

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks every function
here against fixtures produced by the unmodified reference itself
(``tools/make_golden.py`` -> ``tests/golden/*.npz``).

The restatement leans on the same third-party numerics the reference calls
(numpy ``trapz``/``gradient``/``interp``/``fft.rfft``; scipy ``special.sici``,
``special.erf``, ``integrate.simpson``, ``optimize.newton`` — numpy 2.2.6 /
scipy 1.15.3 in this image), so it reproduces the reference to rounding.  The
cosmology layer (CAMB / Eisenstein-Hu P(k), H(z)) is *input* to the path and is
passed in as arrays (``CosmoInputs``).

Every function cites the reference lines it follows.
"""
from dataclasses import dataclass

import numpy as np
from scipy.integrate import simpson
from scipy.optimize import newton
from scipy.special import erf, sici

try:  # numpy>=2 renamed trapz
    _trapz = np.trapezoid
except AttributeError:  # pragma: no cover
    _trapz = np.trapz


@dataclass
class CosmoInputs:
    """What the path takes from the cosmology layer (hmvec/hmvec.py:96-124)."""
    h: float
    omm0: float
    ombh2: float
    rho_crit_0: float          # Msun/Mpc^3
    rho_crit_zs: np.ndarray    # (nz,)
    Pzk: np.ndarray            # (nz, nk) linear P(k) on the user grid
    sPzk: np.ndarray           # (nz, nK) linear P(k) on the sigma2 grid
    ks_sigma2: np.ndarray      # (nK,)
    h_of_z_zs: np.ndarray = None   # (nz,) 1/Mpc, pressure profile only

    @property
    def rho_matter_0(self):
        return self.rho_crit_0 * self.omm0


# ----------------------------------------------------------------- mass function
def tophat_window(kR, switch):
    """hmvec/cosmology.py:30-38 — Fourier top-hat with a Taylor branch below `switch`."""
    with np.errstate(invalid="ignore", divide="ignore"):
        w = 3.0 * (np.sin(kR) - kR * np.cos(kR)) / kR ** 3.0
    small = kR < switch
    x2 = kR[small] ** 2
    w[small] = 1.0 - 0.1 * x2 + 0.00357142857143 * x2 * x2
    return w


def lagrangian_radius(m, rho, delta):
    """hmvec/hmvec.py:627-628."""
    return (3.0 * m / 4.0 / np.pi / delta / rho) ** (1.0 / 3.0)


def sigma2(ci, ms, switch):
    """hmvec/cosmology.py:245-269 with R from hmvec/hmvec.py:117-124.
    Integrator: scipy simpson on the irregular (geomspace) grid."""
    R = lagrangian_radius(ms, ci.rho_matter_0, 1.0)[None, :, None]
    kk = ci.ks_sigma2[None, None, :]
    w2 = tophat_window(kk * R, switch) ** 2.0
    integrand = ci.sPzk[:, None, :] * w2 * kk ** 2.0 / 2.0 / np.pi ** 2
    return simpson(integrand, x=kk, axis=-1)


def tinker_bias(nu, delta=200.0):
    """hmvec/tinker.py:26-40 (Tinker+10 eq. 6)."""
    dc = 1.686
    y = np.log10(delta)
    ey = np.exp(-(4.0 / y) ** 4.0)
    A = 1.0 + 0.24 * y * ey
    a = 0.44 * y - 0.88
    C = 0.019 + 0.107 * y + 0.19 * ey
    nua = nu ** a
    return 1.0 - A * nua / (nua + dc ** a) + 0.183 * nu ** 1.5 + C * nu ** 2.4


def tinker_fnu(nu, zs, alpha_table):
    """hmvec/tinker.py:43-67.  zs broadcastable to nu; alpha_table = (z_i, alpha_i).
    z clamp via heaviside(.,0): z==3 exactly maps to 0, z>3 to 3 (tinker.py:53)."""
    zc = zs * np.heaviside(3.0 - zs, 0.0) + 3.0 * np.heaviside(zs - 3.0, 0.0)
    beta = 0.589 * (1 + zc) ** 0.20
    phi = -0.729 * (1 + zc) ** (-0.08)
    eta = -0.243 * (1 + zc) ** 0.27
    gamma = 0.864 * (1 + zc) ** (-0.01)
    shape = (1.0 + (beta * nu) ** (-2.0 * phi)) * nu ** (2 * eta) * np.exp(-gamma * nu ** 2.0 / 2.0)
    tz, ta = alpha_table
    if np.any(zc < tz[0]) or np.any(zc > tz[-1]):
        raise ValueError("z outside alpha(z) table")  # interp1d(bounds_error=True)
    alpha = np.interp(zc, tz, ta)
    return alpha * shape


def mass_function(s2, zs, ms, rho_m0, mode, p, alpha_table=None):
    """nzm, bh from sigma2: hmvec/hmvec.py:133-161,178-185."""
    dc = p["st_deltac"]
    if mode == "sheth-torman":
        A, a, pp = p["st_A"], p["st_a"], p["st_p"]
        sig = np.sqrt(s2)
        f = (A * np.sqrt(2.0 * a / np.pi) * (1 + (s2 / a / dc ** 2.0) ** pp)
             * (dc / sig) * np.exp(-a * dc ** 2.0 / 2.0 / s2))
        bh = (1.0 + (1.0 / dc) * ((a * dc ** 2.0 / s2) - 1.0)
              + (2.0 * pp / dc) / (1.0 + (a * dc ** 2.0 / s2) ** pp))
    elif mode == "tinker":
        nu = dc / np.sqrt(s2)
        f = nu * tinker_fnu(nu, zs[:, None], alpha_table)
        bh = tinker_bias(nu)
    else:
        raise NotImplementedError(mode)
    dlns = np.gradient(-0.5 * np.log(s2), np.log(ms), axis=-1)
    nzm = rho_m0 * f * dlns / ms[None, :] ** 2.0
    return nzm, bh


# ----------------------------------------------------------------- halo structure
def delta_vir(omz):
    """Bryan & Norman: hmvec/hmvec.py:105-109."""
    x = omz - 1.0
    return 18.0 * np.pi ** 2 + 82.0 * x - 39.0 * x ** 2


def concentration(ms, zs, h, p, mdef):
    """Duffy+08: hmvec/hmvec.py:68-73,163-176."""
    sfx = {"vir": "vir", "mean": "mean"}[mdef]
    A, al, be = p["duffy_A_" + sfx], p["duffy_alpha_" + sfx], p["duffy_beta_" + sfx]
    return A * ((h * ms[None, :] / 2.0e12) ** al) * (1 + zs[:, None]) ** be


def delta_rho_of_mdef(ci, zs, mdef):
    """Density threshold Delta*rho(z) of the halo mass definition (hmvec.py:111-115,217-220)."""
    rho_m = ci.rho_matter_0 * (1 + zs) ** 3.0
    if mdef == "vir":
        return ci.rho_crit_zs * delta_vir(rho_m / ci.rho_crit_zs)
    if mdef == "mean":
        return rho_m * 200.0
    raise NotImplementedError(mdef)


def rvir(ci, ms, zs, mdef):
    return lagrangian_radius(ms[None, :], delta_rho_of_mdef(ci, zs, mdef)[:, None], 1.0)


def _fcon(c):
    return np.log(1.0 + c) - c / (1.0 + c)


def mdelta_from_mdelta(ms, cs, drho1, drho2):
    """hmvec/hmvec.py:748-798 — scipy vectorised secant in ln M2 from x0 = ln M1."""
    M1 = ms[None, :] + cs * 0.0
    d1, d2 = drho1[:, None], drho2[:, None]
    F1 = 1.0 / _fcon(cs)

    def resid(lm2):
        c2 = cs * (np.exp(lm2 - np.log(M1)) * (d1 / d2)) ** (1.0 / 3.0)
        return M1 * F1 - np.exp(lm2) / _fcon(c2)

    return np.exp(newton(resid, np.log(M1)))


# ----------------------------------------------------------------- profiles
def nfw_analytic(ks, cs, rss, zs):
    """hmvec/hmvec.py:346-353.  cs, rss: (nz, nm)."""
    c = cs[..., None]
    mc = np.log(1 + c) - c / (1.0 + c)
    x = ks[None, None] * rss[..., None] * (1 + zs[:, None, None])
    si1, ci1 = sici(x)
    si2, ci2 = sici((1.0 + c) * x)
    return (np.sin(x) * (si2 - si1) - np.sin(c * x) / ((1 + c) * x) + np.cos(x) * (ci2 - ci1)) / mc


def sine_transform(x, y):
    """hmvec/fft.py:35-51.  Quirks kept: step=(x[-1]-x[0])/N (not dx) and the DFT
    phase index starts at 0 while x starts at dx."""
    N = x.size
    step = (x[-1] - x[0]) / N
    uk = -np.fft.rfft(x * y, axis=-1).imag * step
    kt = np.fft.rfftfreq(N, step) * 2 * np.pi
    return kt, uk


def profile_fft(rho_x, cmaxs, rss, zs, ks, xmax, nxs, do_mass_norm=True):
    """hmvec/fft.py:56-115.  rho_x: callable xs->(nz,nm,nxs) or (nxs,); cmaxs,rss: (nz,nm)."""
    xs = np.linspace(0.0, xmax, nxs + 1)[1:]
    rho = rho_x(xs)
    if rho.ndim == 1:
        rho = rho[None, None]
    rho = rho + cmaxs[..., None] * 0.0
    theta = np.where(np.abs(xs) > cmaxs[..., None], 0.0, 1.0)
    mnorm = _trapz(theta * rho * xs ** 2.0, xs) if do_mass_norm else np.ones(cmaxs.shape)
    kt, ukt = sine_transform(xs, rho * theta)
    with np.errstate(invalid="ignore", divide="ignore"):
        uk = ukt / kt[None, None, :] / mnorm[..., None]
    kout = kt / rss[..., None] / (1 + zs[:, None, None])
    out = np.zeros(cmaxs.shape + (ks.size,))
    for i in range(out.shape[0]):
        for j in range(out.shape[1]):
            sel = kout[i, j] > 0
            pu = uk[i, j][sel]
            out[i, j] = np.interp(ks, kout[i, j][sel], pu, left=pu[0], right=0)
    return out


def battaglia_fit(m200c, z, A0, am, az):
    """hmvec/hmvec.py:800-802."""
    return A0 * (m200c / 1.0e14) ** am * (1.0 + z) ** az


def rho_gas_x(x, m200c, z, omb, omm, rhocz, gamma, pp):
    """hmvec/hmvec.py:844-860 (note the deliberate sign of the second gamma)."""
    rho0 = battaglia_fit(m200c, z, pp["rho0_A0"], pp["rho0_alpham"], pp["rho0_alphaz"])
    al = battaglia_fit(m200c, z, pp["alpha_A0"], pp["alpha_alpham"], pp["alpha_alphaz"])
    be = battaglia_fit(m200c, z, pp["beta_A0"], pp["beta_alpham"], pp["beta_alphaz"])
    return (omb / omm) * rhocz * rho0 * (x ** gamma) * (1.0 + x ** al) ** (-(be + gamma) / al)


G_NEWTON_SI = 6.6743e-11  # scipy.constants.G (CODATA 2018), used at hmvec.py:926


def pressure_x(x, m200c, r200c, z, omb, omm, rhocz, alpha, gamma, pp, parsec, msun):
    """hmvec/hmvec.py:906-927."""
    P0 = battaglia_fit(m200c, z, pp["P0_A0"], pp["P0_alpham"], pp["P0_alphaz"])
    xc = battaglia_fit(m200c, z, pp["xc_A0"], pp["xc_alpham"], pp["xc_alphaz"])
    be = battaglia_fit(m200c, z, pp["beta_A0"], pp["beta_alpham"], pp["beta_alphaz"])
    XH = 0.76
    efrac = 2.0 * (XH + 1.0) / (5.0 * XH + 3.0)
    G = G_NEWTON_SI / (parsec * 1e6) ** 3 * msun
    return (efrac * (omb / omm) * 200 * m200c * G * rhocz / (2 * r200c) * P0
            * (x / xc) ** gamma * (1.0 + (x / xc) ** alpha) ** (-be))


# ----------------------------------------------------------------- HOD
_SHMR_LO = (10.72, 0.55, 12.35, 0.28, 0.44, 0.18, 1.56, 2.51, 0.57, 0.17)     # z <= 0.8
_SHMR_HI = (11.09, 0.56, 12.27, -0.84, 0.65, 0.31, 1.12, -0.53, 0.56, -0.12)  # z  > 0.8


def mhalo_of_mstellar(z, log10mstar):
    """Behroozi+10 SHMR: hmvec/hmvec.py:648-695.  z: (nz,1); log10mstar: (1|nz, n)."""
    z = np.asarray(z, dtype=float)
    lm = log10mstar + z * 0
    a = 1.0 / (1 + z)
    out = np.zeros((z.size, lm.shape[-1]))
    for sel, (Ms0, Msa, M1, M1a, b0, ba, g0, ga, d0, da) in (
            (z.reshape(-1) <= 0.8, _SHMR_LO), (z.reshape(-1) > 0.8, _SHMR_HI)):
        aa = a[sel] - 1
        d = lm[sel] - (Ms0 + Msa * aa)
        out[sel] = (-0.5 + (M1 + M1a * aa) + (b0 + ba * aa) * d
                    + 10 ** ((d0 + da * aa) * d) / (1.0 + 10 ** (-(g0 + ga * aa) * d)))
    return out


def mstellar_of_mhalo(z, log10mhalo):
    """hmvec/hmvec.py:634-646: invert on a 4000-pt table per z with np.interp."""
    grid = np.linspace(-18, 18, 4000)[None, :]
    mh = mhalo_of_mstellar(z, grid)
    out = np.zeros((z.shape[0], log10mhalo.shape[-1]))
    for i in range(z.size):
        out[i] = np.interp(log10mhalo[0], mh[i], grid[0])
    return out


def hod_occupations(ms, zs, log10mstar_thresh, p, corr):
    """<Nc>, <Ns>, <Ns(Ns-1)>, <NcNs>: hmvec/hmvec.py:698-731."""
    l10m = np.log10(ms[None, :])
    z = zs[:, None]
    thr = log10mstar_thresh[:, None]
    Nc = 0.5 * (1.0 - erf((thr - mstellar_of_mhalo(z, l10m)) / (np.sqrt(2.0) * p["hod_sig_log_mstellar"])))
    mthr_halo = mhalo_of_mstellar(z, thr)
    scale = lambda B, be: (10.0 ** 12.0) * B * 10 ** ((mthr_halo - 12) * be)   # noqa: E731
    Msat = scale(p["hod_Bsat"], p["hod_betasat"])
    Mcut = scale(p["hod_Bcut"], p["hod_betacut"])
    mm = 10 ** l10m
    Ns = Nc * ((mm / Msat) ** p["hod_alphasat"]) * np.exp(-Mcut / mm)
    if corr == "max":
        with np.errstate(invalid="ignore", divide="ignore"):
            NsNsm1 = Ns ** 2.0 / Nc
        NsNsm1[np.isclose(Nc, 0.0)] = 0
        NcNs = Ns
    elif corr == "min":
        NsNsm1, NcNs = Ns ** 2.0, Ns * Nc
    else:
        raise ValueError(corr)
    return Nc, Ns, NsNsm1, NcNs


def bisection(x, inv_func, ybounds, monotonicity, rtol=1e-4):
    """hmvec/utils.py:9-42.  The stop test is GLOBAL over the vector (all entries
    keep bisecting until every entry meets rtol)."""
    lo = x * 0 + ybounds[0]
    hi = x * 0 + ybounds[1]
    mtol = np.inf
    n = 0
    while np.any(np.abs(mtol) > rtol):
        mid = (lo + hi) / 2.0
        mtol = (inv_func(mid) - x) / x
        up = mtol > 0
        if monotonicity == "decreasing":
            lo[up], hi[~up] = mid[up], mid[~up]
        else:
            hi[up], lo[~up] = mid[up], mid[~up]
        n += 1
    return mid, n


# ----------------------------------------------------------------- the model
class RefHaloModel:
    """Array-in/array-out restatement of hmvec.HaloModel for the hot path
    (hmvec/hmvec.py:75-572).  `p` is the merged parameter dict."""

    def __init__(self, ci, zs, ks, ms, p, mass_function="sheth-torman", mdef="vir",
                 alpha_table=None, skip_nfw=False):
        self.ci, self.p = ci, p
        self.zs, self.ks, self.ms = np.asarray(zs, float), np.asarray(ks, float), np.asarray(ms, float)
        self.mode, self.mdef = mass_function, mdef
        self.uk_profiles, self.pk_profiles, self.hods = {}, {}, {}
        self.Pzk = ci.Pzk
        self.sigma2 = sigma2(ci, self.ms, p["Wkr_taylor_switch"])
        self.nzm, self.bh = mass_function_(self, alpha_table)
        self.cs = concentration(self.ms, self.zs, ci.h, p, mdef)
        self.rvirs = rvir(ci, self.ms, self.zs, mdef)
        if not skip_nfw:
            self.add_nfw_profile("nfw")

    # -- profiles
    def add_nfw_profile(self, name, numeric=False, nxs=None, xmax=None):
        rss = self.rvirs / self.cs
        if numeric:
            nxs = self.p["nfw_integral_numxs"] if nxs is None else nxs
            xmax = self.p["nfw_integral_xmax"] if xmax is None else xmax
            u = profile_fft(lambda x: 1.0 / x / (1.0 + x) ** 2.0, self.cs, rss, self.zs, self.ks, xmax, nxs)
        else:
            u = nfw_analytic(self.ks, self.cs, rss, self.zs)
        self.uk_profiles[name] = u
        return self.ks, u

    def _m200c(self):
        d1 = delta_rho_of_mdef(self.ci, self.zs, self.mdef)
        self.m200c = mdelta_from_mdelta(self.ms, self.cs, d1, 200.0 * self.ci.rho_crit_zs)
        self.r200c = lagrangian_radius(self.m200c, self.ci.rho_crit_zs[:, None], 200.0)
        return self.m200c, self.r200c

    def add_battaglia_profile(self, name, family, gamma, fitp, nxs, xmax):
        """hmvec/hmvec.py:188-250."""
        ci = self.ci
        m200c, r200c = self._m200c()
        omb = ci.ombh2 / ci.h ** 2.0
        rgs = r200c / 2.0
        rho = lambda x: rho_gas_x(x, m200c[..., None], self.zs[:, None, None], omb, ci.omm0,   # noqa: E731
                                  ci.rho_crit_zs[:, None, None], gamma, fitp)
        self.uk_profiles[name] = profile_fft(rho, self.rvirs / rgs, rgs, self.zs, self.ks, xmax, nxs)

    def add_battaglia_pres_profile(self, name, alpha, gamma, fitp, nxs, xmax, sigmaT, m_e_msun, c_si):
        """hmvec/hmvec.py:252-316."""
        ci = self.ci
        m200c, r200c = self._m200c()
        omb = ci.ombh2 / ci.h ** 2.0
        pf = lambda x: pressure_x(x, m200c[..., None], r200c[..., None], self.zs[:, None, None],  # noqa: E731
                                  omb, ci.omm0, ci.rho_crit_zs[:, None, None], alpha, gamma, fitp,
                                  self.p["parsec"], self.p["mSun"])
        pk = profile_fft(pf, self.rvirs / r200c, r200c, self.zs, self.ks, xmax, nxs, do_mass_norm=False)
        pref = 4 * np.pi * (sigmaT / (m_e_msun * c_si ** 2))
        self.pk_profiles[name] = pk * pref * (r200c ** 3 * ((1 + self.zs) ** 2 / ci.h_of_z_zs)[:, None])[..., None]

    # -- HOD
    def _ngal_bg(self, Nc, Ns):
        ngal = _trapz(self.nzm * (Nc + Ns), self.ms, axis=-1)
        bg = _trapz(self.nzm * (Nc + Ns) * self.bh, self.ms, axis=-1) / ngal
        return ngal, bg

    def add_hod(self, name, mthresh=None, ngal=None, corr="max", satellite_profile_name="nfw",
                central_profile_name=None):
        """hmvec/hmvec.py:357-460."""
        p = self.p
        if ngal is not None:
            def nfunc(l10):
                Nc, Ns, _, _ = hod_occupations(self.ms, self.zs, l10, p, "max")
                return _trapz(self.nzm * (Nc + Ns), self.ms, axis=-1)
            l10, self.bisect_iters = bisection(
                np.asarray(ngal, float), nfunc,
                (p["hod_bisection_search_min_log10mthresh"], p["hod_bisection_search_max_log10mthresh"]),
                "decreasing", rtol=p["hod_bisection_search_rtol"])
            mthresh = 10 ** (l10 * p["hod_A_log10mthresh"])
        l10thr = np.log10(np.asarray(mthresh, float))
        Nc, Ns, NsNsm1, NcNs = hod_occupations(self.ms, self.zs, l10thr, p, corr)
        ng, bg = self._ngal_bg(Nc, Ns)
        self.hods[name] = dict(Nc=Nc, Ns=Ns, NsNsm1=NsNsm1, NcNs=NcNs, ngal=ng, bg=bg,
                               satellite_profile=satellite_profile_name,
                               central_profile=central_profile_name, log10mthresh=l10thr[:, None])

    # -- tracer weights (hmvec/hmvec.py:469-497)
    def _hod_parts(self, name):
        hod = self.hods[name]
        uc = 1 if hod["central_profile"] is None else self.uk_profiles[hod["central_profile"]]
        return hod, uc, self.uk_profiles[hod["satellite_profile"]]

    def _w_hod(self, name, lowk=False):
        hod, uc, us = self._hod_parts(name)
        if lowk:
            uc = us = 1
        return (uc * hod["Nc"][..., None] + us * hod["Ns"][..., None]) / hod["ngal"][..., None, None]

    def _w_hod_sq(self, name):
        hod, uc, us = self._hod_parts(name)
        return ((2.0 * uc * us * hod["NcNs"][..., None] + hod["NsNsm1"][..., None] * us ** 2.0)
                / hod["ngal"][..., None, None] ** 2.0)

    def _w_matter(self, name, lowk=False):
        u = 1 if lowk else self.uk_profiles[name]
        return self.ms[..., None] * u / self.ci.rho_matter_0

    def _w_pres(self, name, lowk=False):
        pk = self.pk_profiles[name].copy()
        if lowk:
            pk[:, :, :] = pk[:, :, 0][..., None]
        return pk

    def _w(self, nm):
        if nm in self.hods:
            return self._w_hod(nm)
        if nm in self.uk_profiles:
            return self._w_matter(nm)
        if nm in self.pk_profiles:
            return self._w_pres(nm)
        raise ValueError(nm)

    # -- spectra
    def get_power_1halo(self, name="nfw", name2=None):
        """hmvec/hmvec.py:504-526 (incl. the first-name-only quirk for hod/hod, pres/pres)."""
        name2 = name if name2 is None else name2
        if name in self.hods and name2 in self.hods:
            sq = self._w_hod_sq(name)
        elif name in self.pk_profiles and name2 in self.pk_profiles:
            sq = self._w_pres(name) ** 2
        else:
            sq = self._w(name) * self._w(name2)
        integ = _trapz(self.nzm[..., None] * sq, self.ms[..., None], axis=-2)
        return integ * (1 - np.exp(-(self.ks / self.p["kstar_damping"]) ** 2.0))

    def get_power_2halo(self, name="nfw", name2=None, b1_in=None, b2_in=None):
        """hmvec/hmvec.py:528-572."""
        name2 = name if name2 is None else name2
        msx = self.ms[..., None]

        def integral(term):
            return _trapz(self.nzm[..., None] * term * self.bh[..., None], msx, axis=-2)

        def parts(nm):
            if nm in self.uk_profiles:
                return integral(self._w_matter(nm)), integral(self._w_matter(nm, True)), 1
            if nm in self.pk_profiles:
                return integral(self._w_pres(nm)), integral(0), 0
            if nm in self.hods:
                hod = self.hods[nm]
                _, bg = self._ngal_bg(hod["Nc"], hod["Ns"])
                return integral(self._w_hod(nm)), integral(self._w_hod(nm, True)), bg[:, None]
            raise ValueError(nm)

        I1, C1, b1 = parts(name)
        I2, C2, b2 = parts(name2)
        if b1_in is not None:
            b1 = b1_in.reshape((b1_in.shape[0], 1))
        if b2_in is not None:
            b2 = b2_in.reshape((b1_in.shape[0], 1))
        return self.Pzk * (I1 + b1 - C1) * (I2 + b2 - C2)

    def get_power(self, name, name2=None, b1=None, b2=None):
        return self.get_power_1halo(name, name2) + self.get_power_2halo(name, name2, b1, b2)


def mass_function_(model, alpha_table):
    return mass_function(model.sigma2, model.zs, model.ms, model.ci.rho_matter_0, model.mode,
                         model.p, alpha_table)


# ----------------------------------------------------------------- Limber (row N1)
def lensing_window(ezs, zsrc, H0_invMpc, H_invMpc, chis, chistar, omm0):
    """Delta-function source branch of hmvec/cosmology.py:506-534."""
    w = (chistar - chis) / chistar
    w = np.where(ezs > zsrc, 0.0, w)
    return 1.5 * omm0 * H0_invMpc ** 2.0 * (1.0 + ezs) * chis / H_invMpc * w


def bilinear_clamped(xg, yg, f, x, y):
    """P(z,k) lookup used by limber_integral (hmvec/cosmology.py:890-899): degree-1
    spline evaluated pointwise, clamped to the grid box (fitpack bispeu clamps)."""
    x = np.clip(x, xg[0], xg[-1])
    y = np.clip(y, yg[0], yg[-1])
    i = np.clip(np.searchsorted(xg, x, side="right") - 1, 0, xg.size - 2)
    j = np.clip(np.searchsorted(yg, y, side="right") - 1, 0, yg.size - 2)
    tx = (x - xg[i]) / (xg[i + 1] - xg[i])
    ty = (y - yg[j]) / (yg[j + 1] - yg[j])
    return ((1 - tx) * (1 - ty) * f[j, i] + tx * (1 - ty) * f[j, i + 1]
            + (1 - tx) * ty * f[j + 1, i] + tx * ty * f[j + 1, i + 1])


def limber_integral(ells, zs, ks, Pzks, gzs, W1, W2, hzs, chis):
    """hmvec/cosmology.py:867-904."""
    gzs = np.atleast_1d(np.asarray(gzs, float))
    pref = np.reshape(hzs, -1) * np.reshape(W1, -1) * np.reshape(W2, -1) / np.reshape(chis, -1) ** 2.0
    out = np.zeros(len(ells))
    for i, ell in enumerate(ells):
        kev = (ell + 0.5) / np.reshape(chis, -1)
        val = bilinear_clamped(ks, zs, Pzks, kev, gzs)
        out[i] = (val * pref)[0] if gzs.size == 1 else _trapz(val * pref, gzs)
    return out
