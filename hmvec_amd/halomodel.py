"""``HaloModel`` — the drop-in boundary of the hot path (hmvec/hmvec.py:75-572).

Same constructor, ``add_*`` and ``get_power_*`` surface, argument meaning, error types
and stdout messages as the reference class, but every array lives in HBM and every
stage is a HIP kernel behind the C ABI in ``include/hmgrid.h``.  Host Python here only
wires pointers and scalars.  There is no CPU fallback.

State mirrors the reference: ``uk_profiles`` / ``pk_profiles`` map name -> (nz,nm,nk)
array, ``hods`` maps name -> dict, plus ``sigma2``, ``nzm``, ``bh``, ``Pzk``.  Reads of
those return numpy arrays (copied from the device on first access, then cached);
assigning a numpy array to ``uk_profiles[name]`` uploads it.
"""
import ctypes as C
import itertools
import os
from collections.abc import MutableMapping

import numpy as np
import scipy.constants as constants

from . import _native as nat
from .cosmology import Cosmology, _is_shared_product, sigma2_kgrid, sigma2_weights
from .params import battaglia_defaults, default_params
from .quadrature import gradient_is_uniform, simpson_weights, trapz_weights
from .functions import FN_BG_INTEGRAND, FN_ST_FSIGMA, FN_TINKER_FSIGMA, fn2d, ngal_from_mthresh, trapz_lastaxis
from .functions import context as fn_context
from .utils import vectorized_bisection_search

_trapz = getattr(np, "trapezoid", None) or np.trapz
_EPOCHS = itertools.count(1)      # content tags of the per-model (z,m) arrays (hmg_profile_support_epoch)
_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def _tinker_alpha_table():
    """alpha(z) normalisation table of Tinker+10 f(nu) (values of the table the
    reference ships as hmvec/data/alpha_consistency.txt, read at hmvec/tinker.py:64-66)."""
    with np.load(os.path.join(_DATA, "tinker10_alpha_of_z.npz")) as f:
        return f["z"], f["alpha"]


class DeviceDict(MutableMapping):
    """name -> DeviceArray with numpy reads (lazy D2H, cached) and numpy writes (H2D)."""

    def __init__(self, ctx_getter, on_change=None):
        self._ctx_getter = ctx_getter
        self._on_change = on_change or (lambda: None)
        self._dev = {}
        self._host = {}
        self._hint = {}       # name -> (d_nconst, d_cconst): constant-prefix hint emitted with the tensor

    def dev(self, name):
        return self._dev[name]

    def hint(self, name):
        """(nconst, cconst) device arrays of a profile produced by hmg_profile_fft, else (None, None)."""
        return self._hint.get(name, (None, None))

    def set_dev(self, name, darr, hint=None):
        self._dev[name] = darr
        self._host.pop(name, None)
        if hint is None:
            self._hint.pop(name, None)      # any other way of (re)defining the tensor drops the hint
        else:
            self._hint[name] = hint
        self._on_change()

    def __getitem__(self, name):
        if name not in self._host:
            self._host[name] = self._dev[name].numpy()
        return self._host[name]

    def __setitem__(self, name, value):
        if isinstance(value, nat.DeviceArray):
            self.set_dev(name, value)
        else:
            self.set_dev(name, self._ctx_getter().upload(np.asarray(value, dtype=np.float64)))

    def __delitem__(self, name):
        del self._dev[name]
        self._host.pop(name, None)
        self._hint.pop(name, None)
        self._on_change()

    def __contains__(self, name):      # Mapping's default would call __getitem__ (a D2H copy)
        return name in self._dev

    def __iter__(self):
        return iter(self._dev)

    def __len__(self):
        return len(self._dev)


class HodEntry(MutableMapping):
    """``hods[name]`` — same keys as the reference (hmvec/hmvec.py:452-460); array values
    are fetched from the device on first read."""

    _ARR = ("Nc", "Ns", "NsNsm1", "NcNs", "ngal", "bg")

    def __init__(self, dev, meta):
        self.dev = dev          # key -> DeviceArray
        self._vals = dict(meta)  # python-side values

    def __getitem__(self, k):
        if k not in self._vals:
            self._vals[k] = self.dev[k].numpy()
        return self._vals[k]

    def __setitem__(self, k, v):
        self._vals[k] = v

    def __delitem__(self, k):
        self._vals.pop(k, None)
        self.dev.pop(k, None)

    def __contains__(self, k):
        return k in self._vals or k in self.dev

    def __iter__(self):
        seen = list(self.dev) + [k for k in self._vals if k not in self.dev]
        return iter(seen)

    def __len__(self):
        return len(set(self.dev) | set(self._vals))


def _R_from_M(M, rho, delta):
    """hmvec/hmvec.py:627-628 for the handful of grid radii the constructor needs (host input
    preparation; the public, device-backed R_from_M lives in functions.py)."""
    return (3.0 * M / 4.0 / np.pi / delta / rho) ** (1.0 / 3.0)


class HaloModel(Cosmology):
    def __init__(self, zs, ks, ms=None, params={}, mass_function="sheth-torman",
                 halofit=None, mdef="vir", nfw_numeric=False, skip_nfw=False, accuracy="medium",
                 engine="camb", *, device=0, ctx=None, background=None):
        self.zs = np.asarray(zs, dtype=np.float64)
        self.ks = np.asarray(ks, dtype=np.float64)
        self._device = device
        self.ctx = ctx
        Cosmology.__init__(self, params, halofit, engine=engine, accuracy=accuracy, background=background)

        self.mdef = mdef
        self.mode = mass_function
        self.hods = {}
        self._dcache = {}
        self._pool = {}
        self._use_lanes = os.environ.get("HMG_LANES", "0") == "1"   # two-stream overlap
        # Grouped launches: the launch-only stages of a pass are queued and issued when
        # something consumes them, so that stages which do not depend on each other share a launch.
        # HMG_NO_GROUPS=1 issues every stage as its own launch at call time (A/B timing, debugging).
        self._groups = os.environ.get("HMG_NO_GROUPS", "0") != "1" and not self._use_lanes
        self._stages = []
        self._stage_dims = None

        # (name, name2) -> (state version, P1h, P2h): a fused launch yields both terms, so the
        # usual get_power_1halo(a,b) followed by get_power_2halo(a,b) streams the tensors once
        self._pcache = {}
        self._version = 0
        self.uk_profiles = DeviceDict(self._ctx, self._bump)
        self.pk_profiles = DeviceDict(self._ctx, self._bump)

        if ms is not None:
            self.ms = np.asarray(ms, dtype=np.float64)
            self.init_mass_function(self.ms)

        if not skip_nfw:
            self.add_nfw_profile("nfw", numeric=nfw_numeric)

    # ------------------------------------------------------------------ cosmology inputs
    def _init_cosmology(self, params, halofit):
        """hmvec/hmvec.py:96-102."""
        Cosmology._init_cosmology(self, params, halofit)
        if self.accuracy == "low":
            self.Pzk = self.P_lin_approx(self.ks, self.zs)
        else:
            self.Pzk = self._get_matter_power(self.zs, self.ks, nonlinear=False)
        if halofit is not None:
            self.nPzk = self._get_matter_power(self.zs, self.ks, nonlinear=True)

    def deltav(self, z):
        """Bryan & Norman virial overdensity (hmvec/hmvec.py:105-109)."""
        x = self.omz(z) - 1.0
        return 18.0 * np.pi ** 2.0 + 82.0 * x - 39.0 * x ** 2.0

    def _mdef_delta_rho(self):
        """(delta[z], rho[z]) of the halo mass definition (hmvec/hmvec.py:111-115)."""
        if self.mdef == "vir":
            return self.deltav(self.zs), self.rho_critical_z(self.zs)
        if self.mdef == "mean":
            return 200.0 + 0.0 * self.zs, self.rho_matter_z(self.zs)
        raise NotImplementedError(self.mdef)

    def rvir(self, m, z):
        """Host helper with the reference's signature (hmvec/hmvec.py:111-115)."""
        if self.mdef == "vir":
            return _R_from_M(m, self.rho_critical_z(z), delta=self.deltav(z))
        elif self.mdef == "mean":
            return _R_from_M(m, self.rho_matter_z(z), delta=200.0)

    def R_of_m(self, ms):
        return _R_from_M(ms, self.rho_matter_z(0), delta=1.0)

    # ------------------------------------------------------------------ device plumbing
    def _bump(self):
        """Any change of profiles / HODs / mass function invalidates cached spectra."""
        self._version = getattr(self, "_version", 0) + 1

    def _release_inputs(self, keys):
        """Drop cached device inputs.  Stages queued earlier (the constructor queues three) hold raw pointers to
        them and to pool buffers of the shapes they were built for: the queue is issued before anything is
        released, with the grid sizes recorded when it was built (_stage_dims)."""
        if self._stages:
            self._ctx().flush()
        for k in keys:
            self._dcache.pop(k, None)

    def _dev(self, key, builder):
        """Host array -> device, uploaded once per key (inputs of the path stay resident)."""
        if key not in self._dcache:
            self._dcache[key] = self._ctx().upload(builder())
        return self._dcache[key]

    # -- lanes (two-stream dataflow, HMG_LANES=1).  The main lane (0) carries the chip-filling kernels
    # in sequence - c, rvir -> NFW -> mass conversion + rows -> FFT -> spectra - and the auxiliary lane
    # (1) the short latency-bound chain sigma2 -> n(z,m), b(z,m) -> HOD, which only needs a few CUs
    # for a few microseconds at a time and so hides inside the big kernels' ramps and tails.  Ordering:
    # an auxiliary launch waits for the last SYNC POINT of the main lane (start of a pass / end of the
    # last spectra launch: whatever still reads n, b or the HOD arrays was enqueued before it), the
    # spectra wait for the auxiliary lane.  All events used inside a capture are recorded inside it.
    # Event slots 0..3 are reserved for this class (ShardedSpectra: 8-9, bench.py: 40 and up).
    _AUX = 1
    _EV_SYNC, _EV_AUX = 0, 1

    def _sync_point(self):
        if self._use_lanes:
            ctx = self._ctx()
            ctx.lane(0)
            ctx.record(self._EV_SYNC)
            self._sync_serial = ctx.capture_serial

    def _aux(self):
        """Route the next launches to the auxiliary lane (no-op without lanes)."""
        ctx = self._ctx()
        if self._use_lanes:
            if getattr(self, "_sync_serial", None) != ctx.capture_serial:
                self._sync_point()          # first auxiliary launch of this capture / eager epoch
            ctx.lane(self._AUX)
            ctx.wait(self._EV_SYNC)
        return ctx

    def _aux_done(self):
        if self._use_lanes:
            ctx = self._ctx()
            ctx.record(self._EV_AUX)
            self._aux_pending = True
            ctx.lane(0)

    def _main(self, needs_aux=False):
        """Main lane; with needs_aux the launches that follow also wait for the auxiliary lane."""
        ctx = self._ctx()
        if self._use_lanes:
            ctx.lane(0)
            if needs_aux and getattr(self, "_aux_pending", False):
                ctx.wait(self._EV_AUX)
        return ctx

    def _buf(self, key, shape):
        """Output/workspace buffer allocated once per key, so that re-running a stage is
        launch-only: no allocation, no free, no host synchronisation."""
        shape = tuple(int(x) for x in shape)
        b = self._pool.get(key)
        if b is None or b.shape != shape:
            if b is not None:
                self._ctx().flush()       # a queued stage may still point at the block being replaced
            b = self._ctx().empty(shape)
            self._pool[key] = b
        return b

    @property
    def _nz(self):
        return self.zs.size

    @property
    def _nm(self):
        return self.ms.size

    @property
    def _nk(self):
        return self.ks.size

    def _d_zs(self):
        return self._dev("zs", lambda: self.zs)

    def _d_ks(self):
        return self._dev("ks", lambda: self.ks)

    def _d_ms(self):
        return self._dev("ms", lambda: self.ms)

    def _d_wm(self):
        return self._dev("wm", lambda: trapz_weights(self.ms))

    def _d_Pzk(self):
        return self._dev("Pzk", lambda: self.Pzk)

    def _lazy_host(self, name):
        h = "_h_" + name
        if getattr(self, h, None) is None:
            setattr(self, h, getattr(self, "_d_" + name).numpy())
        return getattr(self, h)

    sigma2 = property(lambda self: self._lazy_host("sigma2"))
    nzm = property(lambda self: self._lazy_host("nzm"))
    bh = property(lambda self: self._lazy_host("bh"))

    # ------------------------------------------------------------------ grouped launches
    # Which queued stage must not run before which (a stage kind appears at most once in the queue):
    # the HOD reads n, b; the profile FFT reads the row parameters.
    _RUNS_AFTER = {"hod": ("massfn",), "fft": ("rows",)}

    def _queue(self, kind, part, keep=()):
        """Queue a launch-only stage.  Anything that reads device state goes through the context, which
        issues the queue first (Context.flush), so deferral is invisible to callers."""
        pend = [st[0] for st in self._stages]
        # The grid sizes a stage was built for travel with it: the raw pointers inside `part` belong to buffers
        # of exactly these sizes, whatever self.ms / self.ks are by the time the queue is issued.
        dims = (self._nz, self._nm if getattr(self, "ms", None) is not None else 0, self._nk, getattr(self, "_nq", 0))
        # same kind twice, a producer queued behind its own consumer (it would overwrite what the consumer
        # still has to read), the front of a new pass (it rewrites what every queued stage reads), or a stage
        # built for another grid than the queued ones: issue what is queued first
        if pend and (kind == "front" or kind in pend or any(kind in self._RUNS_AFTER.get(k, ()) for k in pend)
                     or dims != self._stage_dims):
            self._ctx().flush()
        self._stage_dims = dims
        self._stages.append((kind, part, keep))
        self._ctx().defer(self)

    def _flush(self, prep=None):
        """Issue the queued stages as grouped launches: front (sigma^2 contraction | halo stage | HOD occupations |
        Battaglia row parameters: inputs only), then the tensor group (per-z chain: sigma^2 second stage + n, b -> HOD sums ->
        coefficient rows of the mass integrals `prep` describes | profile FFT rows | NFW rows) - or, when no profile
        transform is queued or its row parameters could not ride with a front, the rows group (n, b tiles | row parameters |
        NFW rows) followed by the profile group (chain | profile FFT rows).  Returns True if `prep` was issued
        (hmg_power_batch_run then skips its preparation launch)."""
        stages, self._stages = self._stages, []
        st = {k: p for k, p, _ in stages}
        if not st:
            return False
        ctx = self._ctx()
        nz, nm, nk, nq = self._stage_dims          # the sizes at queue time, not the model's current ones
        ref = lambda k: C.byref(st[k]) if k in st else None      # noqa: E731
        x = os.environ.get("HMG_X", "")          # experiment switches (tuning only)
        if "prep_alone" in x:
            prep = None
        hod_sums = False
        if "front" in st:
            occ = None
            if "hod" in st:      # the occupation numbers need inputs only: they ride in the front launch ...
                occ = nat.HodPart.from_buffer_copy(st["hod"])
                occ.stage, st["hod"].stage = nat.HOD_OCCUPATIONS, nat.HOD_SUMS
                hod_sums = True  # ... and n_gal, b_g, which need the n, b of this pass, in the chain of the tensor (or profile) group
            args = st.pop("front")
            # ... and so do the Battaglia row parameters: the thread that solves for M_200c goes on to them
            rows = st.pop("rows") if "rows" in st and "rows_alone" not in x else None
            ctx.call_now("hmg_sigma2_halo_front", *args[:-1], C.byref(args[-1]), C.byref(occ) if occ is not None else None,
                         C.byref(rows) if rows is not None else None)
        elif "hod" in st:        # no front to ride with: the HOD's own kernel (n, b are there already)
            h = st.pop("hod")
            ctx.call_now("hmg_hod", nz, nm, h.h_par, h.d_zs, h.d_ms, h.d_log10mstar_thresh, h.d_nzm, h.d_bh, h.d_wm,
                         h.d_Nc, h.d_Ns, h.d_NsNsm1, h.d_NcNs, h.d_ngal, h.d_bg)
        if "fft" in st and ("massfn" in st or "nfw" in st) and "rows" not in st and not x:
            # tensor group: chain (sigma^2 -> n, b -> HOD sums -> coefficient rows) | profile rows | NFW rows, one launch
            ctx.call_now("hmg_profile_support_epoch", getattr(self, "_epoch", 0))
            try:
                ctx.call_now("hmg_group_tensors", nz, nm, nk, nq, ref("massfn"), ref("hod") if hod_sums else None,
                             C.byref(prep) if prep is not None else None, ref("nfw"), ref("fft"))
            finally:
                ctx.call_now("hmg_profile_support_epoch", 0)
            return prep is not None
        nfw_alone = st.pop("nfw") if "nfw_alone" in x and "nfw" in st else None
        if any(k in st for k in ("massfn", "rows", "nfw")):
            ctx.call_now("hmg_group_rows", nz, nm, nk, nq, ref("massfn"), None, ref("rows"), ref("nfw"))
        if nfw_alone is not None:
            ctx.call_now("hmg_nfw_analytic", nz, nm, nk, nfw_alone.d_cs, nfw_alone.d_rs, nfw_alone.d_zs, nfw_alone.d_ks,
                         nfw_alone.d_nfw_series, nfw_alone.d_uk)
        if "fft" in st or hod_sums:
            if "chain_alone" in x and "fft" in st and (hod_sums or prep is not None):
                # experiment: the per-z chain as a launch of its own in front of the stand-alone
                # row kernel (no private segment) instead of a role of the profile group
                ctx.call_now("hmg_group_profile", nz, nm, nk, None, ref("hod") if hod_sums else None,
                             C.byref(prep) if prep is not None else None)
                ctx.call_now("hmg_group_profile", nz, nm, nk, ref("fft"), None, None)
                return prep is not None
            # the support tag is context state: it is set right around the ONE call it describes and restored whatever
            # that call does (a call that raises inside a capture must not leave this model's tag behind: ADVICE r05)
            ctx.call_now("hmg_profile_support_epoch", getattr(self, "_epoch", 0))
            try:
                ctx.call_now("hmg_group_profile", nz, nm, nk, ref("fft"), ref("hod") if hod_sums else None,
                             C.byref(prep) if prep is not None else None)
            finally:
                ctx.call_now("hmg_profile_support_epoch", 0)
            return prep is not None
        return False

    # ------------------------------------------------------------------ mass function
    def get_sigma2(self):
        """hmvec/hmvec.py:121-124 — evaluated by hmg_sigma2."""
        return self.sigma2

    def _tinker_z_params(self):
        """Per-z scalars of Tinker+10 f(nu) (hmvec/tinker.py:53-66): z clamp with the
        heaviside(.,0) quirk (z == 3 -> 0, z > 3 -> 3) and alpha(z) from the table."""
        zs = self.zs
        zc = zs * np.heaviside(3 - zs, 0) + 3 * np.heaviside(zs - 3, 0)
        tz, ta = _tinker_alpha_table()
        if np.any(zc < tz[0]) or np.any(zc > tz[-1]):
            raise ValueError("A value in x_new is outside the interpolation range.")
        alpha = np.interp(zc, tz, ta)
        beta = 0.589 * (1 + zc) ** 0.20
        phi = -0.729 * (1 + zc) ** (-0.08)
        eta = -0.243 * (1 + zc) ** 0.27
        gamma = 0.864 * (1 + zc) ** (-0.01)
        return np.stack([alpha, beta, phi, eta, gamma], axis=1)

    def init_mass_function(self, ms):
        """sigma2, n(z,m), b(z,m), c(z,m), rvir(z,m) on the device (hmvec/hmvec.py:127-185).
        Inputs are uploaded on the first call; later calls only launch kernels."""
        ms = np.asarray(ms, dtype=np.float64)
        if getattr(self, "_ms_key", None) is None or not np.array_equal(ms, self._ms_key):
            self._release_inputs([k for k in self._dcache if k != "zs" and k != "ks" and k != "Pzk"
                                  and not (isinstance(k, tuple) and k[0] == "fftgrid")])
            self._ms_key = ms.copy()
            # concentrations, radii and with them the support of every profile row and the modes it needs are
            # functions of (cosmology, zs, ks, ms): fixed per model until the mass grid changes.  The tag lets the
            # long-grid profile routes reuse the bound they measured once (hmg_profile_support_epoch) instead of
            # synchronising the stream in every call.
            self._epoch = next(_EPOCHS)
        self.ms = ms
        self._bump()
        if self.mode not in ("sheth-torman", "tinker"):
            raise NotImplementedError
        if self.mdef not in ("vir", "mean"):
            raise NotImplementedError
        ctx = self._ctx()
        nz, nm = self._nz, self._nm
        self._h_sigma2 = self._h_nzm = self._h_bh = None
        # sigma^2: inputs (P(k) on the sigma2 grid, Simpson weights, Lagrangian radii)
        if "sig_in" not in self._dcache:
            kmin, kmax, numks = self.p["sigma2_kmin"], self.p["sigma2_kmax"], self.p["sigma2_numks"]
            kq = sigma2_kgrid(kmin, kmax, numks)
            if self.accuracy == "high":
                self.sPzk = self.P_lin_slow(kq, self.zs, kmax=kmax)
            elif self.accuracy == "medium":
                self.sPzk = self.P_lin(kq, self.zs)
            else:
                self.sPzk = self._P_lin_approx_shared(kq, self.zs)
            wq = sigma2_weights(kq)
            # P(k',z) is an input of the path: uploaded and laid out once as the contraction reads it - once per
            # MODEL, or once per host array when the provider hands out the same read-only array again (the analytic
            # provider does for a repeated cosmology: Cosmology.P_lin_approx), together with the grid and its weights
            owned = _is_shared_product(self.sPzk)       # (the provider's own cached product: nothing modifies it)
            shared = ctx.shared.get(id(self.sPzk)) if owned else None
            if shared is None or shared[0] is not self.sPzk:
                d_sP = ctx.upload(self.sPzk)
                n = C.c_size_t()
                nat.check(ctx.lib.hmg_sigma2_layout_size(nz, kq.size, C.byref(n)))
                d_PT = ctx.empty((n.value,))
                ctx.call("hmg_sigma2_prepare", nz, kq.size, d_sP.ptr, d_PT.ptr)
                shared = (self.sPzk, d_PT, ctx.upload(kq), ctx.upload(wq))
                if owned:
                    if len(ctx.shared) >= 4:
                        ctx.flush()           # (queued stages of other models may hold addresses of an evicted entry)
                        ctx.shared.clear()
                    ctx.shared[id(self.sPzk)] = shared
            self._dcache["sig_in"] = shared[1:] + (ctx.upload(self.R_of_m(ms)),)
        d_PT, d_kq, d_wq, d_R = self._dcache["sig_in"]
        self._sync_point()               # start of a pass: everything launched so far precedes it
        ctx = self._aux()
        self._d_sigma2 = self._buf("sigma2", (nz, nm))
        # n(z,m), b(z,m): same launch as the second stage of the sigma^2 contraction
        if "mf_in" not in self._dcache:
            lnm = np.log(ms)
            uniform, step = gradient_is_uniform(lnm)
            par = nat.MassFnParams(
                mode=nat.MF_SHETH_TORMEN if self.mode == "sheth-torman" else nat.MF_TINKER10,
                deltac=self.p["st_deltac"], st_A=self.p["st_A"], st_a=self.p["st_a"], st_p=self.p["st_p"],
                rho_m0=float(self.rho_matter_z(0)[0]), lnm_uniform=int(uniform), lnm_step=step)
            d_tz = ctx.upload(self._tinker_z_params()) if self.mode == "tinker" else None
            delta, rho = self._mdef_delta_rho()
            self._dcache["mf_in"] = (par, ctx.upload(lnm), d_tz, ctx.upload(delta), ctx.upload(rho))
            ctx = self._aux()            # (the uploads above synchronise on whatever lane is current)
        par, d_lnm, d_tz, d_delta, d_rho = self._dcache["mf_in"]
        self._d_nzm, self._d_bh = self._buf("nzm", (nz, nm)), self._buf("bh", (nz, nm))
        d_ms = self._d_ms()
        # c(z,m), rvir(z,m), rs(z,m), the series rows of the analytic NFW kernel and the vir -> 200c mass
        # conversion the Battaglia profiles need: one thread per (z,m), independent of sigma2
        sfx = self.mdef
        self._d_cs, self._d_rvir, self._d_rs = (self._buf(k, (nz, nm)) for k in ("cs", "rvir", "rs"))
        self._d_nfw_series = self._buf("nfw_series", (nz, nm, nat.NFW_SERIES_STRIDE))
        m2, r2 = self._buf("m200c", (nz, nm)), self._buf("r200c", (nz, nm))
        d_drho1 = self._d_drho1()
        d_rhoc = self._dev("rhocz", lambda: self.rho_critical_z(self.zs))
        sig_args = (nz, nm, d_kq.size, d_PT.ptr, d_kq.ptr, d_wq.ptr, d_R.ptr,
                    float(self.p["Wkr_taylor_switch"]), C.byref(par), d_ms.ptr, d_lnm.ptr, nat.ptr(d_tz),
                    self._d_sigma2.ptr, self._d_nzm.ptr, self._d_bh.ptr)
        duffy = tuple(float(self.p[f"duffy_{k}_{sfx}"]) for k in ("A", "alpha", "beta"))
        if self._groups:
            # front group: the contraction beside the halo stage (both need inputs only); the second stage of
            # the contraction and n(z,m), b(z,m) are queued as the first link of the per-z chain
            halo = nat.HaloStageArgs(self._d_zs().ptr, d_delta.ptr, d_rho.ptr, *duffy, float(self.h),
                                     self._d_cs.ptr, self._d_rvir.ptr, self._d_rs.ptr, self._d_nfw_series.ptr,
                                     d_drho1.ptr, 200.0, d_rhoc.ptr, m2.ptr, r2.ptr)
            self._nq = int(d_kq.size)
            self._queue("front", sig_args[:8] + (d_ms.ptr, halo))
            self._queue("massfn", nat.MassFnPart(C.pointer(par), d_ms.ptr, d_lnm.ptr, nat.ptr(d_tz), self._d_sigma2.ptr,
                                                 self._d_nzm.ptr, self._d_bh.ptr), keep=(par,))
        elif not self._use_lanes:
            # one launch behind the contraction: mass function and halo stage side by side
            halo = nat.HaloStageArgs(self._d_zs().ptr, d_delta.ptr, d_rho.ptr, *duffy, float(self.h),
                                     self._d_cs.ptr, self._d_rvir.ptr, self._d_rs.ptr, self._d_nfw_series.ptr,
                                     d_drho1.ptr, 200.0, d_rhoc.ptr, m2.ptr, r2.ptr)
            ctx.call("hmg_sigma2_massfn_halo", *sig_args, C.byref(halo))
        else:
            # two-lane scheme: n, b on the auxiliary lane, the halo stage on the main lane ahead of the profiles
            ctx.call("hmg_sigma2_massfn", *sig_args)
            self._aux_done()
            ctx = self._main()
            ctx.call("hmg_halo_stage", nz, nm, d_ms.ptr, self._d_zs().ptr, d_delta.ptr, d_rho.ptr,
                     *duffy, float(self.h), self._d_cs.ptr, self._d_rvir.ptr, self._d_rs.ptr,
                     self._d_nfw_series.ptr, d_drho1.ptr, 200.0, d_rhoc.ptr, m2.ptr, r2.ptr)
        self._m200c_valid = True

    def get_fsigmaz(self):
        """Multiplicity function f(sigma, z) on the (z,m) grid (hmvec/hmvec.py:133-147).  The path
        itself never materialises it (hmg_massfn goes from sigma^2 straight to n and b); this
        getter evaluates it on the device for callers that ask."""
        dc = self.p["st_deltac"]
        with fn_context(self._ctx()):
            if self.mode == "sheth-torman":
                return fn2d(FN_ST_FSIGMA, [self.sigma2], [self.p["st_A"], self.p["st_a"], self.p["st_p"], dc])
            if self.mode == "tinker":
                tz, ta = _tinker_alpha_table()
                return fn2d(FN_TINKER_FSIGMA, [self.sigma2, self.zs[:, None]], [1.0, 0.368, float(tz.size), dc],
                            tables=(tz, ta))
        raise NotImplementedError

    def get_bh(self):
        return self.bh

    def get_nzm(self):
        return self.nzm

    def concentration(self, mode="duffy"):
        """hmvec/hmvec.py:163-176."""
        if mode != "duffy":
            raise NotImplementedError
        return self._d_cs.numpy()

    # ------------------------------------------------------------------ profiles
    def _m200c(self):
        """(m200c, r200c) on the device (hmvec/hmvec.py:216-225)."""
        ctx = self._ctx()
        nz, nm = self._nz, self._nm
        m2, r2 = self._buf("m200c", (nz, nm)), self._buf("r200c", (nz, nm))
        if not getattr(self, "_m200c_valid", False):
            d1 = self._d_drho1()
            d_rhoc = self._dev("rhocz", lambda: self.rho_critical_z(self.zs))
            ctx.call("hmg_mdelta_convert", nz, nm, self._d_ms().ptr, self._d_cs.ptr, d1.ptr, 200.0,
                     d_rhoc.ptr, m2.ptr, r2.ptr)
            self._m200c_valid = True
        return m2, r2

    def _d_drho1(self):
        """Delta * rho(z) of the model's own mass definition (hmvec/hmvec.py:217-220)."""
        def drho1():
            delta, rho = self._mdef_delta_rho()
            return rho * delta if self.mdef == "vir" else rho * 200.0
        return self._dev("drho1", drho1)

    def _fft_grids(self, xmax, nxs):
        """x grid and FFT wavenumber grid exactly as hmvec/fft.py:45-50,73 build them."""
        key = ("fftgrid", float(xmax), int(nxs))
        if key not in self._dcache:
            xs = np.linspace(0.0, xmax, nxs + 1)[1:]
            step = (xs[-1] - xs[0]) / xs.size
            kts = np.fft.rfftfreq(xs.size, step) * 2 * np.pi
            ctx = self._ctx()
            d_xs = ctx.upload(xs)
            d_lx = ctx.empty((xs.size,))            # ln x_n, shared by every row of every pass
            ctx.call("hmg_profile_fft_logx", xs.size, d_xs.ptr, d_lx.ptr)
            self._dcache[key] = (d_xs, ctx.upload(kts), float(step), d_lx)
        return self._dcache[key]

    def _profile_fft(self, key, nxs, xmax, rowp, consts, gamma, d_cmax, d_rss, do_mass_norm, d_post=None, d_rowsc=None):
        """One hmg_profile_fft launch; returns (tensor, hint).  The hint - how many leading target
        wavenumbers of each row lie below the row's first FFT mode, and the value np.interp's left
        fill gives them all - lets the batched mass integrals skip those parts of the tensor.  It
        is only requested when the target k grid is ascending."""
        ctx = self._ctx()
        nz, nm, nk = self._nz, self._nm, self._nk
        d_xs, d_kts, step, d_lx = self._fft_grids(xmax, nxs)
        out = self._buf(key, (nz, nm, nk))
        if "ks_ascending" not in self._dcache:
            self._dcache["ks_ascending"] = (bool(np.all(np.diff(self.ks) > 0))
                                            and os.environ.get("HMG_NO_HINTS", "0") != "1")   # debugging switch
        hint = None
        if self._dcache["ks_ascending"]:
            # nconst is an int32 array: allocated as (nz*nm+1)//2 doubles, viewed as integers on read
            hint = (self._buf((key, "nconst"), ((nz * nm + 1) // 2,)), self._buf((key, "cconst"), (nz, nm)))
        amp, xc, alpha, expo = rowp
        args = (int(nxs), step, d_xs.ptr, d_kts.ptr,
                nat.ptr(amp), nat.ptr(xc), nat.ptr(alpha), nat.ptr(expo),
                float(consts[0]), float(consts[1]), float(consts[2]), float(consts[3]), float(gamma),
                d_cmax.ptr, d_rss.ptr, self._d_zs().ptr, self._d_ks().ptr, int(do_mass_norm),
                nat.ptr(d_post), out.ptr, nat.ptr(hint[0] if hint else None), nat.ptr(hint[1] if hint else None),
                d_lx.ptr)
        if self._groups:
            # (the row scalars only travel with the hint arrays: they carry the left-fill count, which needs ks ascending)
            self._queue("fft", nat.ProfileFftPart(*args, nat.ptr(d_rowsc if hint else None)))
        else:
            ctx.flush()          # deferred stages of ANY model on this context first: their own flush resets the tag
            ctx.call("hmg_profile_support_epoch", getattr(self, "_epoch", 0))
            try:
                ctx.call("hmg_profile_fft", nz, nm, nk, *args)
            finally:
                ctx.call("hmg_profile_support_epoch", 0)
        return out, hint

    def _battaglia_rowparams(self, key, kind, fit9, gamma, alpha_const, pref, post_pref, nxs=None, xmax=None):
        """Row parameters of a Battaglia family; returns (outs, d_rowsc): with a radial grid (nxs, xmax) and an ascending
        k grid the queued stage also leaves the output-side scalars of every row for the transform (hmg_rows_part,
        ABI 8), d_rowsc is then the buffer to hand to _profile_fft, else None."""
        ctx = self._ctx()
        nz, nm = self._nz, self._nm
        outs = [self._buf((key, "rowp", i), (nz, nm)) for i in range(7)]
        fit = (C.c_double * 9)(*fit9)
        d_hz = self._dev("hz", lambda: self.h_of_z(self.zs))
        d_rhoc = self._dev("rhocz", lambda: self.rho_critical_z(self.zs))
        if not getattr(self, "_m200c_valid", False):
            # mass conversion and row parameters in one launch
            m2, r2 = self._buf("m200c", (nz, nm)), self._buf("r200c", (nz, nm))
            ctx.call("hmg_profile_rows_from_mvir", kind, nz, nm, self._d_ms().ptr, self._d_cs.ptr,
                     self._d_rvir.ptr, self._d_zs().ptr, self._d_drho1().ptr, 200.0, d_rhoc.ptr, d_hz.ptr,
                     C.byref(fit), float(gamma), float(alpha_const), float(pref), float(post_pref),
                     m2.ptr, r2.ptr, *[o.ptr for o in outs])
            self._m200c_valid = True
            return outs, None
        m200c, r200c = self._m200c()
        if self._groups:
            d_rowsc, extra = None, (None, None, 0, 0, None)
            if nxs is not None and nxs % 2 == 0 and os.environ.get("HMG_NO_ROWSC", "0") != "1":
                if "ks_ascending" not in self._dcache:
                    self._dcache["ks_ascending"] = (bool(np.all(np.diff(self.ks) > 0))
                                                    and os.environ.get("HMG_NO_HINTS", "0") != "1")
                if self._dcache["ks_ascending"]:
                    d_kts = self._fft_grids(xmax, nxs)[1]
                    d_rowsc = self._buf((key, "rowsc"), (nz, nm, nat.ROWSC_STRIDE))
                    extra = (self._d_ks().ptr, d_kts.ptr, self._nk, int(nxs) // 2, d_rowsc.ptr)
            self._queue("rows", nat.RowsPart(kind, m200c.ptr, r200c.ptr, self._d_rvir.ptr, self._d_zs().ptr, d_rhoc.ptr,
                                             d_hz.ptr, fit, float(gamma), float(alpha_const), float(pref),
                                             float(post_pref), *[o.ptr for o in outs], *extra))
            return outs, d_rowsc
        ctx.call("hmg_profile_rowparams", kind, nz, nm, m200c.ptr, r200c.ptr, self._d_rvir.ptr,
                 self._d_zs().ptr, d_rhoc.ptr, d_hz.ptr, C.byref(fit), float(gamma), float(alpha_const),
                 float(pref), float(post_pref), *[o.ptr for o in outs])
        return outs, None

    def add_battaglia_profile(self, name, family=None, param_override=None, nxs=None, xmax=None,
                              ignore_existing=False):
        """Battaglia+16 gas density profile -> uk_profiles[name] (hmvec/hmvec.py:188-250)."""
        if not (ignore_existing):
            assert name not in self.uk_profiles.keys(), "Profile name already exists."
        assert name != "nfw", "Name nfw is reserved."
        if nxs is None:
            nxs = self.p["electron_density_profile_integral_numxs"]
        if xmax is None:
            xmax = self.p["electron_density_profile_integral_xmax"]
        if family is None:
            family = self.p["battaglia_gas_family"]
        pparams = {"battaglia_gas_gamma": self.p["battaglia_gas_gamma"]}
        pparams.update(battaglia_defaults[family])
        if param_override is not None:
            print(param_override)
            for key in param_override.keys():
                if key == "battaglia_gas_gamma" or key in battaglia_defaults[family]:
                    pparams[key] = param_override[key]
        omb = self.p["ombh2"] / self.h ** 2.0
        gamma = pparams["battaglia_gas_gamma"]
        fit9 = [pparams[a + b] for a in ("rho0_", "alpha_", "beta_") for b in ("A0", "alpham", "alphaz")]
        key = ("uk", name)
        self._main()
        (amp, xc, alpha, expo, cmax, rscale, _post), rowsc = self._battaglia_rowparams(
            key, nat.PROF_BATTAGLIA_GAS, fit9, gamma, 0.0, omb / self.omm0, 0.0, nxs=nxs, xmax=xmax)
        out, hint = self._profile_fft(key, nxs, xmax, (amp, None, alpha, expo), (0.0, 1.0, 0.0, 0.0), gamma,
                                      cmax, rscale, True, d_rowsc=rowsc)
        self.uk_profiles.set_dev(name, out, hint)

    def add_battaglia_pres_profile(self, name, family=None, param_override=None, nxs=None, xmax=None,
                                   ignore_existing=False):
        """Battaglia+12 electron pressure profile -> pk_profiles[name] (hmvec/hmvec.py:252-316)."""
        if not (ignore_existing):
            assert name not in self.pk_profiles.keys(), "Profile name already exists."
        assert name != "nfw", "Name nfw is reserved."
        if nxs is None:
            nxs = self.p["electron_pressure_profile_integral_numxs"]
        if xmax is None:
            xmax = self.p["electron_pressure_profile_integral_xmax"]
        if family is None:
            family = self.p["battaglia_pres_family"]
        pparams = {"battaglia_pres_gamma": self.p["battaglia_pres_gamma"],
                   "battaglia_pres_alpha": self.p["battaglia_pres_alpha"]}
        pparams.update(battaglia_defaults[family])
        if param_override is not None:
            for key in param_override.keys():
                if key in ("battaglia_pres_gamma", "battaglia_pres_alpha") or key in battaglia_defaults[family]:
                    pparams[key] = param_override[key]
        omb = self.p["ombh2"] / self.h ** 2.0
        gamma, alpha = pparams["battaglia_pres_gamma"], pparams["battaglia_pres_alpha"]
        fit9 = [pparams[a + b] for a in ("P0_", "xc_", "beta_") for b in ("A0", "alpham", "alphaz")]
        XH = 0.76
        eFrac = 2.0 * (XH + 1.0) / (5.0 * XH + 3.0)
        G_newt = constants.G / (default_params["parsec"] * 1e6) ** 3 * default_params["mSun"]
        pref = eFrac * (omb / self.omm0) * 200 * G_newt
        sigmaT = constants.physical_constants["Thomson cross section"][0]
        mElect = constants.physical_constants["electron mass"][0] / default_params["mSun"]
        post_pref = 4 * np.pi * (sigmaT / (mElect * constants.c ** 2))
        key = ("pk", name)
        self._main()
        (amp, xc, _alpha, expo, cmax, rscale, post), rowsc = self._battaglia_rowparams(
            key, nat.PROF_BATTAGLIA_PRES, fit9, gamma, alpha, pref, post_pref, nxs=nxs, xmax=xmax)
        out, hint = self._profile_fft(key, nxs, xmax, (amp, xc, None, expo), (0.0, 0.0, alpha, 0.0), gamma,
                                      cmax, rscale, False, d_post=post, d_rowsc=rowsc)
        self.pk_profiles.set_dev(name, out, hint)

    def add_nfw_profile(self, name, numeric=False, nxs=None, xmax=None, ignore_existing=False):
        """NFW u(k|m,z): analytic Si/Ci or numeric FFT branch (hmvec/hmvec.py:318-355).
        Returns (ks, uk) like the reference."""
        if not (ignore_existing):
            assert name not in self.uk_profiles.keys(), "Profile name already exists."
        if nxs is None:
            nxs = self.p["nfw_integral_numxs"]
        if xmax is None:
            xmax = self.p["nfw_integral_xmax"]
        ctx = self._ctx()
        nz, nm, nk = self._nz, self._nm, self._nk
        hint = None
        ctx = self._main()
        if numeric:
            # rho = 1/x/(1+x)^2 is the gamma=-1, alpha=1, expo=2 member of the family
            out, hint = self._profile_fft(("uk", name), nxs, xmax, (None, None, None, None), (1.0, 1.0, 1.0, 2.0),
                                          -1.0, self._d_cs, self._d_rs, True)
        else:
            out = self._buf(("uk", name), (nz, nm, nk))
            if self._groups:
                self._queue("nfw", nat.NfwPart(self._d_cs.ptr, self._d_rs.ptr, self._d_zs().ptr, self._d_ks().ptr,
                                               self._d_nfw_series.ptr, out.ptr))
            else:
                ctx.call("hmg_nfw_analytic", nz, nm, nk, self._d_cs.ptr, self._d_rs.ptr, self._d_zs().ptr,
                         self._d_ks().ptr, self._d_nfw_series.ptr, out.ptr)
        self.uk_profiles.set_dev(name, out, hint)
        return self.ks, _LazyArray(self.uk_profiles, name)

    # ------------------------------------------------------------------ HOD
    _HOD_PARAMS = ["hod_sig_log_mstellar", "hod_bisection_search_min_log10mthresh",
                   "hod_bisection_search_max_log10mthresh", "hod_bisection_search_rtol",
                   "hod_bisection_search_warn_iter", "hod_alphasat", "hod_Bsat",
                   "hod_betasat", "hod_Bcut", "hod_betacut", "hod_A_log10mthresh"]

    def _hod_device(self, key, log10mstar_thresh, pparams, corr):
        """One hmg_hod launch into the buffers of `key`; returns dict of DeviceArrays."""
        ctx = self._ctx()
        nz, nm = self._nz, self._nm
        par = nat.HodParams(pparams["hod_sig_log_mstellar"], pparams["hod_alphasat"], pparams["hod_Bsat"],
                            pparams["hod_betasat"], pparams["hod_Bcut"], pparams["hod_betacut"],
                            {"max": 0, "min": 1}[corr])
        thr = np.ascontiguousarray(log10mstar_thresh, dtype=np.float64)
        cached = self._dcache.get(("thr", key))
        if cached is None or cached[0].shape != thr.shape:
            if cached is not None:
                self._release_inputs([("thr", key)])
            cached = (thr.copy(), ctx.upload(thr))
            self._dcache[("thr", key)] = cached
        elif not np.array_equal(cached[0], thr):
            # new thresholds go into the SAME device buffer (stream-ordered behind the launches that
            # read the old ones): nothing is freed, so nothing synchronises the other lanes
            ctx.write(cached[1], thr)
            cached[0][...] = thr
        d_thr = cached[1]
        out = {k: self._buf((key, k), (nz, nm)) for k in ("Nc", "Ns", "NsNsm1", "NcNs")}
        out["ngal"], out["bg"] = self._buf((key, "ngal"), (nz,)), self._buf((key, "bg"), (nz,))
        d_wm, d_ms, d_zs = self._d_wm(), self._d_ms(), self._d_zs()     # (uploads, if any, before the lane switch)
        if self._groups:
            self._queue("hod", nat.HodPart(nat.HOD_ALL, C.pointer(par), d_zs.ptr, d_ms.ptr, d_thr.ptr, self._d_nzm.ptr,
                                           self._d_bh.ptr, d_wm.ptr, out["Nc"].ptr, out["Ns"].ptr, out["NsNsm1"].ptr,
                                           out["NcNs"].ptr, out["ngal"].ptr, out["bg"].ptr), keep=(par,))
            return out
        ctx = self._aux()
        ctx.call("hmg_hod", nz, nm, C.byref(par), d_zs.ptr, d_ms.ptr, d_thr.ptr,
                 self._d_nzm.ptr, self._d_bh.ptr, d_wm.ptr, out["Nc"].ptr, out["Ns"].ptr,
                 out["NsNsm1"].ptr, out["NcNs"].ptr, out["ngal"].ptr, out["bg"].ptr)
        self._aux_done()
        return out

    def add_hod(self, name, mthresh=None, ngal=None, corr="max", satellite_profile_name="nfw",
                central_profile_name=None, ignore_existing=False, param_override=None):
        """Specify an HOD by stellar-mass threshold or by number density (hmvec/hmvec.py:357-460)."""
        if not (ignore_existing):
            assert name not in self.uk_profiles.keys(), "HOD name already used by profile."
        assert satellite_profile_name in self.uk_profiles.keys(), "No matter profile by that name exists."
        if central_profile_name is not None:
            assert central_profile_name in self.uk_profiles.keys(), "No matter profile by that name exists."
        if not (ignore_existing):
            assert name not in self.hods.keys(), "HOD with that name already exists."
        pparams = {ip: self.p[ip] for ip in self._HOD_PARAMS}
        if param_override is not None:
            for key in param_override.keys():
                if key in self._HOD_PARAMS:
                    pparams[key] = param_override[key]
                else:
                    raise ValueError  # not an HOD parameter
        if corr not in ("max", "min"):
            raise ValueError(corr)

        if ngal is not None:
            ngal = np.asarray(ngal)
            if ngal.size != self.zs.size:
                raise ValueError("ngal has to be a vector of size self.zs")
            assert mthresh is None
            log10mthresh = self._bisect_mthresh(ngal.astype(np.float64), pparams)
            mthresh = 10 ** (log10mthresh * pparams["hod_A_log10mthresh"])
        try:
            assert mthresh.size == self.zs.size
        except Exception:
            raise ValueError("mthresh has to be a vector of size self.zs")

        l10 = np.log10(np.asarray(mthresh, dtype=np.float64))
        dev = self._hod_device(("hod", name), l10, pparams, corr)
        self._bump()
        self.hods[name] = HodEntry(dev, dict(satellite_profile=satellite_profile_name,
                                             central_profile=central_profile_name,
                                             log10mthresh=np.log10(mthresh[:, None])))

    def _bisect_mthresh(self, ngal, pparams, full_model=None):
        """Bisection on log10 mthresh with the reference's GLOBAL stop test over z
        (hmvec/utils.py:9-42 called at hmvec/hmvec.py:426-433).  Every z keeps bisecting
        until all z meet rtol, so the answer depends on the whole z vector."""
        def ngal_of(log10mthresh):       # the device HOD kernel's n_gal(z) for trial thresholds
            return self._hod_device(("hod", "_bisect"), log10mthresh, pparams, "max")["ngal"].numpy()

        return vectorized_bisection_search(ngal, ngal_of,
                                           [pparams["hod_bisection_search_min_log10mthresh"],
                                            pparams["hod_bisection_search_max_log10mthresh"]],
                                           "decreasing", rtol=pparams["hod_bisection_search_rtol"], verbose=True,
                                           hang_check_num_iter=pparams["hod_bisection_search_warn_iter"])

    def get_ngal(self, Nc, Ns):
        """hmvec/hmvec.py:462."""
        with fn_context(self._ctx()):
            return ngal_from_mthresh(nzm=self.nzm, ms=self.ms, Ncs=Nc, Nss=Ns)

    def get_bg(self, Nc, Ns, ngal):
        """hmvec/hmvec.py:464-466."""
        with fn_context(self._ctx()):
            return trapz_lastaxis(fn2d(FN_BG_INTEGRAND, [self.nzm, Nc, Ns, self.bh]), self.ms) / ngal

    # ------------------------------------------------------------------ spectra
    def _tracer(self, name, order):
        """Resolve a tracer name to an hmg_tracer.  `order` is the reference's lookup
        order, which differs between the 1-halo (hods, uk, pk: hmvec.py:516-523) and the
        2-halo (uk, pk, hods: hmvec.py:537-550) code paths."""
        def tensor(dd, nm_):
            # (a kernel trusts the pointer it is handed: a hand-assigned entry of another shape must stop here - numpy
            # would refuse to broadcast it in the reference, hmvec/hmvec.py:516-550)
            d_ = dd.dev(nm_)
            if tuple(d_.shape) != (self._nz, self._nm, self._nk):
                raise ValueError(f"profile {nm_!r} has shape {tuple(d_.shape)}, the model's grid is "
                                 f"{(self._nz, self._nm, self._nk)}")
            return d_
        for kind in order:
            if kind == "h" and name in self.hods:
                hod = self.hods[name]
                cn = hod["central_profile"]
                d = hod.dev
                sn = hod["satellite_profile"]
                hs = self.uk_profiles.hint(sn)
                hc = self.uk_profiles.hint(cn) if cn is not None else (None, None)
                t = nat.Tracer(nat.TRACER_HOD, tensor(self.uk_profiles, sn).ptr,
                               None if cn is None else tensor(self.uk_profiles, cn).ptr,
                               d["Nc"].ptr, d["Ns"].ptr, d["NcNs"].ptr, d["NsNsm1"].ptr, d["ngal"].ptr, None,
                               nat.ptr(hs[0]), nat.ptr(hs[1]), nat.ptr(hc[0]), nat.ptr(hc[1]))
                return t, "h"
            if kind == "m" and name in self.uk_profiles:
                hp = self.uk_profiles.hint(name)
                return nat.Tracer(nat.TRACER_MATTER, tensor(self.uk_profiles, name).ptr, None, None, None, None, None,
                                  None, None, nat.ptr(hp[0]), nat.ptr(hp[1])), "m"
            if kind == "p" and name in self.pk_profiles:
                hp = self.pk_profiles.hint(name)
                return nat.Tracer(nat.TRACER_PRESSURE, tensor(self.pk_profiles, name).ptr, None, None, None, None, None,
                                  None, None, nat.ptr(hp[0]), nat.ptr(hp[1])), "p"
        raise ValueError

    def _power_launch(self, ta, tb, want1, want2, out1=None, out2=None):
        ctx = self._main(needs_aux=True)
        nz, nm, nk = self._nz, self._nm, self._nk
        d1 = (out1 if out1 is not None else ctx.empty((nz, nk))) if want1 else None
        d2 = (out2 if out2 is not None else ctx.empty((nz, nk))) if want2 else None
        ctx.call("hmg_power", nz, nm, nk, C.byref(ta), C.byref(tb), self._d_nzm.ptr, self._d_bh.ptr,
                 self._d_ms().ptr, self._d_wm().ptr, self._d_ks().ptr, self._d_Pzk().ptr,
                 self._rho_m0(), float(self.p["kstar_damping"]), nat.ptr(d1), nat.ptr(d2))
        self._sync_point()
        return d1, d2

    def _rho_m0(self):
        if getattr(self, "_rho_m0_val", None) is None:
            self._rho_m0_val = float(self.rho_matter_z(0)[0])
        return self._rho_m0_val

    def power_device(self, name, name2=None, b1_in=None, b2_in=None, want=("1h", "2h"), out1=None, out2=None):
        """Device-resident (P1h, P2h) DeviceArrays of shape (nz, nk) — one fused pass over the
        profile tensors when the 1-halo and 2-halo code paths resolve the names identically."""
        name2 = name if name2 is None else name2
        keep = []
        want1, want2 = "1h" in want, "2h" in want
        a1, ka1 = self._tracer(name, "hmp")
        b1, kb1 = self._tracer(name2, "hmp")
        a2, ka2 = self._tracer(name, "mph") if want2 else (a1, ka1)
        b2, kb2 = self._tracer(name2, "mph") if want2 else (b1, kb1)
        if want2:
            if b1_in is not None:
                keep.append(self._ctx().upload(np.asarray(b1_in, dtype=np.float64).reshape(-1)))
                a2.d_bias_override = keep[-1].ptr
            if b2_in is not None:
                n = b1_in.shape[0]     # reference reshapes b2_in with b1_in's length (hmvec.py:561)
                keep.append(self._ctx().upload(np.asarray(b2_in, dtype=np.float64).reshape((n, 1))))
                b2.d_bias_override = keep[-1].ptr
        same = (ka1 == ka2) and (kb1 == kb2)
        if want1 and want2 and same:
            d1, d2 = self._power_launch(a2, b2, True, True, out1, out2)
        else:
            d1 = self._power_launch(a1, b1, True, False, out1, None)[0] if want1 else None
            d2 = self._power_launch(a2, b2, False, True, None, out2)[1] if want2 else None
        if keep:
            self._ctx().sync()
        return d1, d2

    def power_device_batch(self, pairs, outs1=None, outs2=None):
        """(P1h, P2h) DeviceArrays for SEVERAL (name, name2) pairs with every distinct profile
        tensor streamed from HBM once for the whole batch (hmg_power_batch) instead of once
        per pair.  Falls back to one fused launch per pair when the batch cannot express the
        reference's semantics (two different HOD / two different pressure names, a name that
        the 1-halo and 2-halo code paths resolve differently, more than 4 tracers)."""
        ctx = self._ctx()
        nz, nm, nk = self._nz, self._nm, self._nk
        pairs = [(a, a if b is None else b) for a, b in pairs]
        names = []
        for a, b in pairs:
            for n_ in (a, b):
                if n_ not in names:
                    names.append(n_)
        res1 = [self._tracer(n_, "hmp") for n_ in names]
        res2 = [self._tracer(n_, "mph") for n_ in names]
        kinds = {n_: r[1] for n_, r in zip(names, res1)}
        batchable = len(names) <= 4 and all(r1[1] == r2[1] for r1, r2 in zip(res1, res2))
        batchable = batchable and not any(a != b and kinds[a] == kinds[b] and kinds[a] in "hp"
                                          for a, b in pairs)
        o1 = [outs1[i] if outs1 is not None else ctx.empty((nz, nk)) for i in range(len(pairs))]
        o2 = [outs2[i] if outs2 is not None else ctx.empty((nz, nk)) for i in range(len(pairs))]
        if not batchable:
            for i, (a, b) in enumerate(pairs):
                self.power_device(a, b, out1=o1[i], out2=o2[i])
            return o1, o2
        # (a,b) and (b,a) are the same spectrum here: compute each unordered pair once
        uniq, alias = [], []
        for a, b in pairs:
            key = tuple(sorted((names.index(a), names.index(b))))
            if key not in uniq:
                uniq.append(key)
            alias.append(uniq.index(key))
        first = [alias.index(u) for u in range(len(uniq))]
        n = len(uniq)
        ctx = self._main(needs_aux=True)
        tr = (nat.Tracer * len(names))(*[r[0] for r in res1])
        pa = (C.c_int * n)(*[u[0] for u in uniq])
        pb = (C.c_int * n)(*[u[1] for u in uniq])
        p1 = (C.c_void_p * n)(*[o1[i].ptr for i in first])
        p2 = (C.c_void_p * n)(*[o2[i].ptr for i in first])
        d_wm, d_Pzk = self._d_wm(), self._d_Pzk()         # (uploads, if any, before the queue is issued)
        desc = nat.PowerBatchDesc(len(names), tr, n, pa, pb, self._d_nzm.ptr, self._d_bh.ptr, self._d_ms().ptr,
                                  d_wm.ptr, self._d_ks().ptr, d_Pzk.ptr, self._rho_m0(),
                                  float(self.p["kstar_damping"]), p1, p2)
        # the coefficient rows of this batch ride in the tensor / profile group of the queued stages, if there is one
        prepared = self._flush(prep=desc) if self._stages else False
        ctx.call("hmg_power_batch_run", nz, nm, nk, C.byref(desc), nat.PB_PREPARED if prepared else 0)
        self._sync_point()
        for i, u in enumerate(alias):
            if first[u] != i:
                ctx.call("hmg_memcpy_d2d", o1[i].ptr, o1[first[u]].ptr, o1[i].nbytes)
                ctx.call("hmg_memcpy_d2d", o2[i].ptr, o2[first[u]].ptr, o2[i].nbytes)
        return o1, o2

    def spectra_block(self, pairs, nbuf=1):
        """A reusable result block for several (name, name2) spectra: one contiguous device buffer
        for the 2*len(pairs) (nz,nk) outputs and one page-locked host buffer of the same shape.
        ``compute()`` enqueues the batched mass integrals into it, ``fetch()`` brings all of them to
        the host with ONE asynchronous copy at link speed and returns numpy views (valid until the
        next ``fetch`` of the same block).  For parameter sweeps that read every spectrum back."""
        return SpectraBlock(self, pairs, nbuf)

    def get_power_all(self, pairs):
        """Extension of the reference API: {(name, name2): P_1h + P_2h} for several pairs in one
        pass over the profile tensors."""
        seen = []
        for a, b in pairs:
            for nm_ in (a, a if b is None else b):
                if nm_ not in seen:
                    seen.append(nm_)
        self._tsz_notice(*seen)
        o1, o2 = self.power_device_batch(pairs)
        return {tuple(p): self._sum_on_device(a, b) for p, a, b in zip(pairs, o1, o2)}

    def _tsz_notice(self, *names):
        """The reference prints this once per pressure tracer in every 2-halo evaluation
        (hmvec/hmvec.py:544)."""
        for nm_ in names:
            if nm_ not in self.uk_profiles and nm_ in self.pk_profiles:
                print("Check the consistency relation for tSZ")

    def _tensor_names(self, nm_):
        """Profile tensors a tracer streams, or None if the 1-halo and 2-halo code paths would
        resolve the name differently (not batchable)."""
        try:
            k1, k2 = self._tracer(nm_, "hmp")[1], self._tracer(nm_, "mph")[1]
        except (ValueError, KeyError):
            return None
        if k1 != k2:
            return None
        if k1 == "h":
            hod = self.hods[nm_]
            return {("uk", hod["satellite_profile"])} | (
                {("uk", hod["central_profile"])} if hod["central_profile"] is not None else set())
        return {("uk" if k1 == "m" else "pk", nm_)}

    _SMALL_GRID_BYTES = 32 << 20       # tensors up to this size: every registered tracer rides in the first batch

    def _tensors_valid(self, tn):
        shape = (self._nz, self._nm, self._nk)
        for kind, nm_ in tn:
            d = (self.uk_profiles if kind == "uk" else self.pk_profiles)._dev.get(nm_)
            if d is None or tuple(d.shape) != shape:
                return False
        return True

    def _free_riders(self, name, name2):
        """Other registered tracers whose tensors are a subset of what (name, name2) streams
        anyway: their spectra with each other and with the requested pair cost no extra HBM
        traffic in the batched kernel, so they are computed in the same pass and cached."""
        need = self._tensor_names(name)
        need2 = self._tensor_names(name2)
        if need is None or need2 is None:
            return None
        need = need | need2
        kinds = {}
        names = [name] + ([name2] if name2 != name else [])
        # On a small grid (a tensor below 32 MB: the README grid's are 32 MB for all three) a tensor more in the batch
        # costs microseconds while a batch more costs a launch and a result copy: every registered tracer rides along
        # in the first request.  On a large grid only those whose tensors are streamed anyway.
        small = self._nz * self._nm * self._nk * 8 <= self._SMALL_GRID_BYTES
        for cand in list(self.hods) + list(self.uk_profiles) + list(self.pk_profiles):
            if cand in names or len(names) >= 4:
                continue
            tn = self._tensor_names(cand)
            # a rider must not be able to break the request it rides with: only tracers whose tensors are on the
            # device in this model's (nz, nm, nk) shape (a hand-assigned uk_profiles entry can be anything)
            if tn is not None and (tn <= need or small) and self._tensors_valid(tn):
                names.append(cand)
        for n_ in names:
            kinds[n_] = self._tracer(n_, "hmp")[1]
        pairs = []
        for i, a in enumerate(names):
            for b in names[i:]:
                # two different HOD (or pressure) names use the first name's square term in the
                # reference (hmvec.py:510-513): order-dependent, leave those to the per-pair kernel
                if a != b and kinds[a] == kinds[b] and kinds[a] in "hp":
                    continue
                pairs.append((a, b))
        return pairs

    def _power_cached(self, name, name2):
        name2 = name if name2 is None else name2
        ent = self._pcache.get((name, name2))
        if ent is not None and ent[0] == self._version:
            return ent[1], ent[2]
        pairs = self._free_riders(name, name2)
        # Batchable requests always go through the batched kernel, also when nothing rides along: a pair's
        # sums do not depend on what else is in the batch (foreign tensors enter with exact-zero coefficients),
        # so a spectrum has the same bits whatever was asked for before it.  The one-pair kernel (another
        # summation order, ~1e-16 away) is left with what the batch cannot do: bias overrides, the verbose
        # terms, names the 1-halo and 2-halo lookups resolve differently, two different HOD/pressure names.
        if pairs and ((name, name2) in pairs or (name2, name) in pairs):
            # every spectrum of the batch in ONE device block: the host side then fetches the block in one copy the
            # first time any of them is asked for (_HostBlock) instead of one synchronising copy per spectrum
            n = len(pairs)
            blk = self._ctx().empty((2 * n, self._nz, self._nk))
            per = self._nz * self._nk
            o1 = [blk.view(i * per, (self._nz, self._nk)) for i in range(n)]
            o2 = [blk.view((n + i) * per, (self._nz, self._nk)) for i in range(n)]
            try:
                self.power_device_batch(pairs, outs1=o1, outs2=o2)
            except Exception:
                # a rider turned out bad after all: the request itself must not fail for it - the minimal batch
                minimal = [(a, b) for a, b in pairs if {a, b} <= {name, name2}]
                if len(minimal) == len(pairs):
                    raise
                pairs, n = minimal, len(minimal)
                o1 = [blk.view(i * per, (self._nz, self._nk)) for i in range(n)]
                o2 = [blk.view((n + i) * per, (self._nz, self._nk)) for i in range(n)]
                self.power_device_batch(pairs, outs1=o1, outs2=o2)
            hb = _HostBlock(blk)
            for i, ((a, b), d1, d2) in enumerate(zip(pairs, o1, o2)):
                self._pcache[(a, b)] = (self._version, d1, d2, hb, i, n + i)
                self._pcache[(b, a)] = (self._version, d1, d2, hb, i, n + i)
        else:
            d1, d2 = self.power_device(name, name2)
            self._pcache[(name, name2)] = (self._version, d1, d2)
        ent = self._pcache[(name, name2)]
        return ent[1], ent[2]

    def _power_host(self, name, name2, term):
        """Host copy of P_1h (term 0) or P_2h (term 1) of a cached pair."""
        d = self._power_cached(name, name2)
        ent = self._pcache[(name, name if name2 is None else name2)]
        if len(ent) > 3:
            return ent[3].take(ent[4 + term])
        return d[term].numpy()

    def get_power(self, name, name2=None, verbose=False, b1=None, b2=None):
        """P_1h + P_2h in one pass (hmvec/hmvec.py:500-502)."""
        self._tsz_notice(name, name if name2 is None else name2)
        if b1 is None and b2 is None:
            d1, d2 = self._power_cached(name, name2)
        else:
            d1, d2 = self.power_device(name, name2, b1, b2)
        if verbose:
            self._print_consistency(name, name2)
        return self._sum_on_device(d1, d2)

    def _sum_on_device(self, d1, d2):
        """P_1h + P_2h added on the device: one array crosses PCIe (hmvec/hmvec.py:500-502)."""
        ctx = self._ctx()
        out = ctx.empty(d1.shape)
        ctx.call("hmg_add", d1.size, d1.ptr, d2.ptr, out.ptr)
        return out.numpy()

    def get_power_1halo(self, name="nfw", name2=None):
        """hmvec/hmvec.py:504-526."""
        return self._power_host(name, name2, 0)

    def get_power_2halo(self, name="nfw", name2=None, verbose=False, b1_in=None, b2_in=None):
        """hmvec/hmvec.py:528-572."""
        self._tsz_notice(name, name if name2 is None else name2)
        if b1_in is None and b2_in is None:
            out = self._power_host(name, name2, 1)
        else:
            out = self.power_device(name, name2, b1_in, b2_in, want=("2h",))[1].numpy()
        if verbose:
            self._print_consistency(name, name2)
        return out

    def two_halo_terms(self, name, name2=None):
        """(I_1, C_1, I_2, C_2) of the 2-halo term (hmvec/hmvec.py:563-568): the mass integrals
        I(z,k) = int dm n b W(k), shape (nz,nk), and their k -> 0 consistency limits C(z), shape (nz,1)."""
        name2 = name if name2 is None else name2
        ctx = self._main(needs_aux=True)
        nz, nm, nk = self._nz, self._nm, self._nk
        ta, tb = self._tracer(name, "mph")[0], self._tracer(name2, "mph")[0]
        d_i1, d_i2, d_c = ctx.empty((nz, nk)), ctx.empty((nz, nk)), ctx.empty((nz, 2))
        ctx.call("hmg_power_2halo_terms", nz, nm, nk, C.byref(ta), C.byref(tb), self._d_nzm.ptr, self._d_bh.ptr,
                 self._d_ms().ptr, self._d_wm().ptr, self._d_ks().ptr, self._rho_m0(), d_i1.ptr, d_i2.ptr, d_c.ptr)
        c = d_c.numpy()
        return d_i1.numpy(), c[:, 0:1].copy(), d_i2.numpy(), c[:, 1:2].copy()

    def _print_consistency(self, name, name2):
        """The two lines get_power_2halo(verbose=True) prints (hmvec/hmvec.py:569-571)."""
        i1, c1, i2, c2 = self.two_halo_terms(name, name2)
        print("Two-halo consistency1: ", c1, i1)
        print("Two-halo consistency2: ", c2, i2)


class _HostBlock:
    """The spectra of one batched launch on the host: fetched in one copy on first use.  Every array handed out is the
    caller's own, as in the reference (a fresh array per call): a slice of the fetched block the first time a spectrum
    is asked for, a new copy from the device after that (the first caller may have written into its slice)."""

    def __init__(self, dev):
        self.dev, self.host, self.given = dev, None, set()

    def take(self, i):
        if i in self.given:
            n = self.dev.shape[1] * self.dev.shape[2]
            return self.dev.view(i * n, self.dev.shape[1:]).numpy()
        if self.host is None:
            self.host = self.dev.numpy()
        self.given.add(i)
        return self.host[i]


class SpectraBlock:
    """See HaloModel.spectra_block.  With ``nbuf=2`` the block is double-buffered for streams of passes (parameter
    sweeps): ``compute(slot)`` enqueues pass i into device block i % 2, ``fetch_async(slot)`` sends it to its pinned
    twin on the context's copy lane behind an event, and pass i+1's kernels run while that copy is in flight;
    ``wait(slot)`` blocks the host on that one copy only and returns the views.  The next ``compute(slot)`` of the
    same slot is ordered behind the copy on the device, so no result is overwritten before it has left."""

    _COPY_LANE = 2
    _EV_READY, _EV_DONE = 10, 12       # event slots 10-11, 12-13 (HaloModel: 0-3, ShardedSpectra: 8-9, bench.py: 40+)

    def __init__(self, model, pairs, nbuf=1):
        self.model = model
        self.pairs = [(a, a if b is None else b) for a, b in pairs]
        ctx = model._ctx()
        nz, nk = model.zs.size, model.ks.size
        n = 2 * len(self.pairs)
        self.nbuf = int(nbuf)
        self._devs = [ctx.empty((n, nz, nk)) for _ in range(self.nbuf)]
        self._views = [[d.view(i * nz * nk, (nz, nk)) for i in range(n)] for d in self._devs]
        self._hosts = [nat.PinnedArray(ctx, (n, nz, nk)) for _ in range(self.nbuf)]
        self._inflight = [False] * self.nbuf
        self.dev, self.views, self.host = self._devs[0], self._views[0], self._hosts[0]
        self.nbytes = self.dev.nbytes

    def compute(self, slot=0):
        """Launch-only: the spectra of all pairs into device block `slot` (behind the copy that last read it)."""
        if self._inflight[slot]:
            self.model._ctx().wait(self._EV_DONE + slot)
        self.model.power_device_batch(self.pairs, self._views[slot][0::2], self._views[slot][1::2])

    def _as_dict(self, slot):
        a = self._hosts[slot].array
        return {p: (a[2 * i], a[2 * i + 1]) for i, p in enumerate(self.pairs)}

    def fetch(self, slot=0):
        """{(name, name2): (P_1h, P_2h)} as views of the pinned host block; blocks until the copy lands."""
        ctx = self.model._ctx()
        ctx.copy_to_pinned(self._hosts[slot], self._devs[slot])
        ctx.sync()
        self._inflight[slot] = False
        return self._as_dict(slot)

    def fetch_async(self, slot=0):
        """Start the copy of block `slot` on the copy lane, behind the launches enqueued so far; returns at once."""
        ctx = self.model._ctx()
        ctx.record(self._EV_READY + slot)
        ctx.lane(self._COPY_LANE)
        ctx.wait(self._EV_READY + slot)
        ctx.copy_to_pinned(self._hosts[slot], self._devs[slot])
        ctx.record(self._EV_DONE + slot)
        ctx.lane(0)
        self._inflight[slot] = True

    def wait(self, slot=0):
        """Views of block `slot` once its copy has landed (the device may already be busy with later passes)."""
        if self._inflight[slot]:
            self.model._ctx().event_synchronize(self._EV_DONE + slot)
        return self._as_dict(slot)


class _LazyArray:
    """Return value of add_nfw_profile's second element: behaves as the ndarray when used."""

    def __init__(self, mapping, name):
        self._m, self._n = mapping, name

    def __array__(self, dtype=None, copy=None):
        a = self._m[self._n]
        return a if dtype is None else a.astype(dtype)

    def __getattr__(self, k):
        return getattr(self._m[self._n], k)

    def __getitem__(self, idx):
        return self._m[self._n][idx]
