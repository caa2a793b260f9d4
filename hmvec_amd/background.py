"""Background-cosmology providers (the "cosmology seam").

The reference builds a CAMB (or CLASS) object in ``Cosmology._init_cosmology``
(hmvec/cosmology.py:161-211) and asks it for H(z), distances and Omega_nu.  Those
are *inputs* to the halo-model hot path, not part of it, so here they sit behind
a tiny provider interface:

    hubble_parameter(z)  [km/s/Mpc]      h_of_z(z)  [1/Mpc]
    comoving_radial_distance(z) [Mpc]    angular_diameter_distance(z) [Mpc]
    angular_diameter_distance2(z1, z2)   get_Omega(name)       YHe

``AnalyticBackground`` is a closed-form (no radiation, no massive neutrinos)
w0-wa CDM background; it is what the golden fixtures were generated with (the
generator registers it as the stand-in ``camb`` results object), so both sides
of every parity test see bit-identical inputs.  ``CambBackground`` forwards to a
real CAMB install when one is importable.
"""
import numpy as np

C_KMS = 299792.458  # speed of light [km/s]; same constant as hmvec/cosmology.py:27

# Fixed 96-node Gauss-Legendre rule for chi(z); deterministic, vectorised.
_GL_X, _GL_W = np.polynomial.legendre.leggauss(96)


class AnalyticBackground:
    def __init__(self, H0, ombh2, omch2, omk=0.0, w0=-1.0, wa=0.0, YHe=None):
        self.H0 = float(H0)
        h = self.H0 / 100.0
        self.omm = (ombh2 + omch2) / h ** 2
        self.omk = float(omk)
        self.omde = 1.0 - self.omm - self.omk
        self.w0 = float(w0)
        self.wa = float(wa)
        self.YHe = 0.2454 if YHe is None else YHe
        self._chi_memo = {}

    def _E(self, z):
        z = np.asarray(z, dtype=np.float64)
        a1 = 1.0 + z
        de = a1 ** (3.0 * (1.0 + self.w0 + self.wa)) * np.exp(-3.0 * self.wa * z / a1)
        return np.sqrt(self.omm * a1 ** 3 + self.omk * a1 ** 2 + self.omde * de)

    def hubble_parameter(self, z):
        return self.H0 * self._E(z)

    def h_of_z(self, z):
        return self.hubble_parameter(z) / C_KMS

    def _chi_flat(self, z):
        z = np.asarray(z, dtype=np.float64)
        # chi(z) is asked for the same few redshift vectors over and over (every lensing window, every
        # Limber projection): the last results are kept, keyed by the redshifts and the parameters
        key = None
        if z.size <= 4096:
            key = (z.shape, z.tobytes(), self.H0, self.omm, self.omk, self.w0, self.wa)
            hit = self._chi_memo.get(key)
            if hit is not None:
                return hit.copy()
        zz = np.atleast_1d(z)[..., None]
        nodes = 0.5 * zz * (_GL_X + 1.0)
        val = 0.5 * zz[..., 0] * np.sum(_GL_W / self._E(nodes), axis=-1)
        out = (C_KMS / self.H0) * val.reshape(z.shape)
        if key is not None:
            if len(self._chi_memo) >= 64:
                self._chi_memo.pop(next(iter(self._chi_memo)))
            self._chi_memo[key] = np.array(out, copy=True)
        return out

    def comoving_radial_distance(self, z):
        return self._chi_flat(z)

    def _transverse(self, chi):
        if self.omk == 0.0:
            return chi
        rk = (C_KMS / self.H0) / np.sqrt(abs(self.omk))
        return rk * (np.sinh(chi / rk) if self.omk > 0 else np.sin(chi / rk))

    def angular_diameter_distance(self, z):
        z = np.asarray(z, dtype=np.float64)
        return self._transverse(self._chi_flat(z)) / (1.0 + z)

    def angular_diameter_distance2(self, z1, z2):
        z1 = np.asarray(z1, dtype=np.float64)
        z2 = np.asarray(z2, dtype=np.float64)
        return self._transverse(self._chi_flat(z2) - self._chi_flat(z1)) / (1.0 + z2)

    def get_Omega(self, name):
        if name == "nu":
            return 0.0
        if name in ("baryon", "cdm", "de", "K"):
            raise NotImplementedError(name)
        raise ValueError(name)


class CambBackground:
    """Thin forwarder to a real CAMB results object (only if camb is installed).

    Mirrors the ``camb.set_params`` call at hmvec/cosmology.py:161-176 keyword for keyword, including the
    ``theta100`` parameterisation (hmvec/cosmology.py:140-143: ``cosmomc_theta = theta100/100`` with ``H0=None``).
    CAMB is not installable in this image; tests/test_host_cpu.py drives this class through a recording stand-in
    ``camb`` module, so the forwarding itself is exercised.
    """

    def __init__(self, params, halofit=None):
        import camb  # noqa: F401  (ImportError propagates: caller decides)

        if "theta100" in params:
            theta, H0 = params["theta100"] / 100.0, None
        else:
            theta, H0 = None, params["H0"]
        YHe = params.get("YHe")
        rTensors = params.get("r", 0.0)
        self.pars = camb.set_params(
            ns=params["ns"], As=params["As"], r=rTensors, H0=H0, cosmomc_theta=theta,
            ombh2=params["ombh2"], omch2=params["omch2"], mnu=params["mnu"],
            omk=params["omk"], tau=params["tau"], nnu=params["nnu"],
            num_massive_neutrinos=params["num_massive_neutrinos"],
            w=params["w0"], wa=params["wa"], dark_energy_model="ppf",
            halofit_version=params["default_halofit"] if halofit is None else halofit,
            AccuracyBoost=2, pivot_scalar=params["pivot_scalar"], YHe=YHe)
        self.pars.WantTransfer = True
        if rTensors is not None:
            self.pars.WantTensors = True
        self.results = camb.get_background(self.pars)
        self.YHe = self.pars.YHe

    def pk_interpolator(self, zs, kmax, var="total", nonlinear=False):
        """camb.get_matter_power_interpolator with the reference's arguments
        (hmvec/cosmology.py:775-786)."""
        import camb
        from camb import model
        cvar = {"weyl": model.Transfer_Weyl, "total": "delta_tot", "cb": "delta_nonu"}[var]
        return camb.get_matter_power_interpolator(self.pars, nonlinear=nonlinear, hubble_units=False,
                                                  k_hunit=False, kmax=kmax, var1=cvar, var2=cvar,
                                                  zmax=zs[-1])

    def __getattr__(self, name):
        return getattr(self.results, name)


class TabulatedPowerInterpolator:
    """``.P(zs, ks, grid=True)`` over a table P[z, k], the one call the path makes on CAMB's
    ``get_matter_power_interpolator`` result (hmvec/cosmology.py:229,369,381).  Like CAMB's
    interpolator this is a bivariate spline of ln P in (z, ln k) - cubic where the table has at
    least four nodes along an axis, lower order otherwise."""

    def __init__(self, zs, ks, P):
        from scipy.interpolate import RectBivariateSpline
        self.zs = np.asarray(zs, dtype=np.float64)
        self.ks = np.asarray(ks, dtype=np.float64)
        P = np.asarray(P, dtype=np.float64)
        if P.shape != (self.zs.size, self.ks.size) or np.any(P <= 0):
            raise ValueError("P must be a positive (nz, nk) table")
        kz, kk = min(3, self.zs.size - 1), min(3, self.ks.size - 1)
        if kz < 1 or kk < 1:
            raise ValueError("the table needs at least two redshifts and two wavenumbers")
        self._spl = RectBivariateSpline(self.zs, np.log(self.ks), np.log(P), kx=kz, ky=kk, s=0)

    def P(self, z, k, grid=True):
        z = np.atleast_1d(np.asarray(z, dtype=np.float64))
        k = np.atleast_1d(np.asarray(k, dtype=np.float64))
        if z.min() < self.zs[0] or z.max() > self.zs[-1] or k.min() < self.ks[0] or k.max() > self.ks[-1]:
            raise ValueError("requested (z, k) outside the tabulated P(k, z)")
        if grid:
            zi, ki = np.argsort(z), np.argsort(k)          # the spline wants ascending axes
            out = np.empty((z.size, k.size))
            out[np.ix_(zi, ki)] = np.exp(self._spl(z[zi], np.log(k[ki]), grid=True))
            return out
        return np.exp(self._spl(z, np.log(k), grid=False))


class TabulatedBackground(AnalyticBackground):
    """Background + a tabulated linear (and optionally non-linear) matter power spectrum: the
    provider to use when the Boltzmann code ran elsewhere (CAMB is not installable on the GPU
    boxes) and its ``P(k, z)`` table travels as data.  ``accuracy='medium'`` then normalises the
    Eisenstein-Hu shape to the table at ``knorm`` and ``accuracy='high'`` integrates the table
    itself, exactly the two code paths of hmvec/cosmology.py:255-260,353-382.  Distances and
    H(z) come from the closed-form background unless ``base`` supplies another provider."""

    def __init__(self, params, zs, ks, P_lin, P_nonlin=None, base=None):
        AnalyticBackground.__init__(self, params["H0"], params["ombh2"], params["omch2"], params.get("omk", 0.0),
                                    params.get("w0", -1.0), params.get("wa", 0.0), params.get("YHe"))
        self._base = base
        self._lin = TabulatedPowerInterpolator(zs, ks, P_lin)
        self._nonlin = None if P_nonlin is None else TabulatedPowerInterpolator(zs, ks, P_nonlin)

    def pk_interpolator(self, zs, kmax, var="total", nonlinear=False):
        if nonlinear:
            if self._nonlin is None:
                raise NotImplementedError("no non-linear table was given to this provider")
            return self._nonlin
        return self._lin

    def __getattribute__(self, name):
        if name in ("hubble_parameter", "h_of_z", "comoving_radial_distance", "angular_diameter_distance",
                    "angular_diameter_distance2", "get_Omega"):
            base = object.__getattribute__(self, "_base")
            if base is not None:
                return getattr(base, name)
        return object.__getattribute__(self, name)
