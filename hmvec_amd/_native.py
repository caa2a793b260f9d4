"""ctypes binding of libhmgrid.so (C ABI in include/hmgrid.h).

There is NO CPU fallback: if the shared library is missing or a call fails, this
module raises.  Build the library with ``python -c "import __graft_entry__ as g; g.build()"``
or ``make -C hmvec_amd/csrc``.
"""
import ctypes as C
import math
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HMG_LIB_PATH") or os.path.join(_HERE, "libhmgrid.so")   # override: tuning experiments only
ABI_VERSION = 9
COMM_ID_BYTES = 128

c_double_p = C.c_void_p  # device or host pointers travel as plain addresses


class NativeError(RuntimeError):
    pass


class HaloStageArgs(C.Structure):
    """hmg_halo_stage_args (include/hmgrid.h): the halo-stage half of hmg_sigma2_massfn_halo."""
    _fields_ = [("d_zs", C.c_void_p), ("d_delta", C.c_void_p), ("d_rho", C.c_void_p),
                ("duffy_A", C.c_double), ("duffy_alpha", C.c_double), ("duffy_beta", C.c_double),
                ("h", C.c_double), ("d_cs", C.c_void_p), ("d_rvir", C.c_void_p), ("d_rs", C.c_void_p),
                ("d_nfw_series", C.c_void_p), ("d_drho1", C.c_void_p), ("delta2", C.c_double),
                ("d_rho2", C.c_void_p), ("d_m2", C.c_void_p), ("d_r2", C.c_void_p)]


class MassFnParams(C.Structure):
    _fields_ = [("mode", C.c_int), ("deltac", C.c_double), ("st_A", C.c_double),
                ("st_a", C.c_double), ("st_p", C.c_double), ("rho_m0", C.c_double),
                ("lnm_uniform", C.c_int), ("lnm_step", C.c_double)]


class HodParams(C.Structure):
    _fields_ = [("sig_log_mstellar", C.c_double), ("alphasat", C.c_double), ("Bsat", C.c_double),
                ("betasat", C.c_double), ("Bcut", C.c_double), ("betacut", C.c_double),
                ("corr", C.c_int)]


class Tracer(C.Structure):
    _fields_ = [("kind", C.c_int), ("d_prof", C.c_void_p), ("d_cprof", C.c_void_p),
                ("d_Nc", C.c_void_p), ("d_Ns", C.c_void_p), ("d_NcNs", C.c_void_p),
                ("d_NsNsm1", C.c_void_p), ("d_ngal", C.c_void_p),
                ("d_bias_override", C.c_void_p),
                ("d_prof_nconst", C.c_void_p), ("d_prof_cconst", C.c_void_p),
                ("d_cprof_nconst", C.c_void_p), ("d_cprof_cconst", C.c_void_p)]


class MassFnPart(C.Structure):
    """hmg_massfn_part: second stage of the sigma^2 contraction + n(z,m), b(z,m) as a link of a per-z chain."""
    _fields_ = [("h_par", C.POINTER(MassFnParams)), ("d_ms", C.c_void_p), ("d_lnms", C.c_void_p),
                ("d_tinker_z", C.c_void_p), ("d_sigma2", C.c_void_p), ("d_nzm", C.c_void_p), ("d_bh", C.c_void_p)]


class HodPart(C.Structure):
    """hmg_hod_part: hmg_hod's arguments."""
    _fields_ = [("stage", C.c_int), ("h_par", C.POINTER(HodParams)), ("d_zs", C.c_void_p), ("d_ms", C.c_void_p),
                ("d_log10mstar_thresh", C.c_void_p), ("d_nzm", C.c_void_p), ("d_bh", C.c_void_p),
                ("d_wm", C.c_void_p), ("d_Nc", C.c_void_p), ("d_Ns", C.c_void_p), ("d_NsNsm1", C.c_void_p),
                ("d_NcNs", C.c_void_p), ("d_ngal", C.c_void_p), ("d_bg", C.c_void_p)]


class RowsPart(C.Structure):
    """hmg_rows_part: hmg_profile_rowparams's arguments."""
    _fields_ = [("kind", C.c_int), ("d_m200c", C.c_void_p), ("d_r200c", C.c_void_p), ("d_rvir", C.c_void_p),
                ("d_zs", C.c_void_p), ("d_rhocz", C.c_void_p), ("d_hz", C.c_void_p), ("fit", C.c_double * 9),
                ("gamma", C.c_double), ("alpha_const", C.c_double), ("amp_prefactor", C.c_double),
                ("post_prefactor", C.c_double), ("d_amp", C.c_void_p), ("d_xc", C.c_void_p),
                ("d_alpha", C.c_void_p), ("d_expo", C.c_void_p), ("d_cmax", C.c_void_p),
                ("d_rscale", C.c_void_p), ("d_post", C.c_void_p),
                ("d_ks", C.c_void_p), ("d_kts", C.c_void_p), ("nk", C.c_int), ("fft_m", C.c_int), ("d_rowsc", C.c_void_p)]


class NfwPart(C.Structure):
    """hmg_nfw_part: hmg_nfw_analytic's arguments."""
    _fields_ = [("d_cs", C.c_void_p), ("d_rs", C.c_void_p), ("d_zs", C.c_void_p), ("d_ks", C.c_void_p),
                ("d_nfw_series", C.c_void_p), ("d_uk", C.c_void_p)]


class ProfileFftPart(C.Structure):
    """hmg_profile_fft_part: hmg_profile_fft's arguments after nk."""
    _fields_ = [("nxs", C.c_int), ("fft_step", C.c_double), ("d_xs", C.c_void_p), ("d_kts", C.c_void_p),
                ("d_amp", C.c_void_p), ("d_xc", C.c_void_p), ("d_alpha", C.c_void_p), ("d_expo", C.c_void_p),
                ("amp_const", C.c_double), ("xc_const", C.c_double), ("alpha_const", C.c_double),
                ("expo_const", C.c_double), ("gamma", C.c_double), ("d_cmax", C.c_void_p), ("d_rss", C.c_void_p),
                ("d_zs", C.c_void_p), ("d_ks", C.c_void_p), ("do_mass_norm", C.c_int), ("d_post", C.c_void_p),
                ("d_out", C.c_void_p), ("d_nconst", C.c_void_p), ("d_cconst", C.c_void_p), ("d_logxs", C.c_void_p),
                ("d_rowsc", C.c_void_p)]


class PowerBatchDesc(C.Structure):
    """hmg_power_batch_desc: hmg_power_batch's arguments after nk."""
    _fields_ = [("ntr", C.c_int), ("h_tr", C.POINTER(Tracer)), ("npairs", C.c_int),
                ("h_pair_a", C.POINTER(C.c_int)), ("h_pair_b", C.POINTER(C.c_int)),
                ("d_nzm", C.c_void_p), ("d_bh", C.c_void_p), ("d_ms", C.c_void_p), ("d_wm", C.c_void_p),
                ("d_ks", C.c_void_p), ("d_Pzk", C.c_void_p), ("rho_m0", C.c_double), ("kstar", C.c_double),
                ("h_P1h", C.POINTER(C.c_void_p)), ("h_P2h", C.POINTER(C.c_void_p))]


PB_PREPARED = 1
HOD_ALL, HOD_OCCUPATIONS, HOD_SUMS = 0, 1, 2
MF_SHETH_TORMEN, MF_TINKER10 = 0, 1
PROF_BATTAGLIA_GAS, PROF_BATTAGLIA_PRES = 1, 2
TRACER_MATTER, TRACER_HOD, TRACER_PRESSURE = 0, 1, 2
KERNEL_POWER, KERNEL_NFW, KERNEL_PROFILE_FFT = 0, 1, 2
EVENT_SLOTS = 4096
NFW_SERIES_STRIDE = 36     # HMG_NFW_SERIES_STRIDE
ROWSC_STRIDE = 8            # HMG_ROWSC_STRIDE

_I, _D, _P, _Z = C.c_int, C.c_double, C.c_void_p, C.c_size_t
# name -> argtypes (restype is int for all but hmg_last_error); mirrors include/hmgrid.h
SIGNATURES = {
    "hmg_abi_version": [],
    "hmg_ctx_create": [_I, C.POINTER(_P)],
    "hmg_ctx_destroy": [_P],
    "hmg_malloc": [_P, _Z, C.POINTER(_P)],
    "hmg_free": [_P, _P],
    "hmg_memcpy_h2d": [_P, _P, _P, _Z],
    "hmg_memcpy_d2h": [_P, _P, _P, _Z],
    "hmg_memcpy_d2d": [_P, _P, _P, _Z],
    "hmg_sync": [_P],
    "hmg_host_alloc": [_P, _Z, C.POINTER(_P)],
    "hmg_host_free": [_P, _P],
    "hmg_memcpy_d2h_async": [_P, _P, _P, _Z],
    "hmg_memcpy_h2d_async": [_P, _P, _P, _Z],
    "hmg_event_synchronize": [_P, _I],
    "hmg_graph_begin": [_P],
    "hmg_graph_end": [_P, C.POINTER(_I)],
    "hmg_graph_abort": [_P],
    "hmg_graph_kernel_nodes": [_P, _I, C.POINTER(_I)],
    "hmg_graph_launch": [_P, _I],
    "hmg_graph_destroy": [_P, _I],
    "hmg_lane_set": [_P, _I],
    "hmg_profile_support_epoch": [_P, C.c_longlong],
    "hmg_event_wait": [_P, _I],
    "hmg_event_record": [_P, _I],
    "hmg_elapsed_ms": [_P, _I, _I, C.POINTER(_D)],
    "hmg_bracket_next": [_P, _I, _I, _I],
    "hmg_sigma2": [_P, _I, _I, _I, _P, _P, _P, _P, _D, _P],
    "hmg_sigma2_layout_size": [_I, _I, C.POINTER(_Z)],
    "hmg_sigma2_prepare": [_P, _I, _I, _P, _P],
    "hmg_sigma2_prepared": [_P, _I, _I, _I, _P, _P, _P, _P, _D, _P],
    "hmg_sigma2_massfn": [_P, _I, _I, _I, _P, _P, _P, _P, _D, C.POINTER(MassFnParams), _P, _P, _P, _P, _P, _P],
    "hmg_sigma2_massfn_halo": [_P, _I, _I, _I, _P, _P, _P, _P, _D, C.POINTER(MassFnParams), _P, _P, _P, _P, _P, _P,
                               C.POINTER(HaloStageArgs)],
    "hmg_halo_stage": [_P, _I, _I, _P, _P, _P, _P, _D, _D, _D, _D, _P, _P, _P, _P, _P, _D, _P, _P, _P],
    "hmg_massfn": [_P, _I, _I, C.POINTER(MassFnParams), _P, _P, _P, _P, _P, _P],
    "hmg_halo_structure": [_P, _I, _I, _P, _P, _P, _P, _D, _D, _D, _D, _P, _P, _P],
    "hmg_mdelta_convert": [_P, _I, _I, _P, _P, _P, _D, _P, _P, _P],
    "hmg_nfw_analytic": [_P, _I, _I, _I, _P, _P, _P, _P, _P, _P],
    "hmg_profile_rowparams": [_P, _I, _I, _I, _P, _P, _P, _P, _P, _P, C.POINTER(_D * 9), _D, _D, _D,
                              _D, _P, _P, _P, _P, _P, _P, _P],
    "hmg_profile_rows_from_mvir": [_P, _I, _I, _I, _P, _P, _P, _P, _P, _D, _P, _P, C.POINTER(_D * 9), _D, _D,
                                   _D, _D, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "hmg_profile_fft": [_P, _I, _I, _I, _I, _D, _P, _P, _P, _P, _P, _P, _D, _D, _D, _D, _D,
                        _P, _P, _P, _P, _I, _P, _P, _P, _P, _P],
    "hmg_profile_fft_logx": [_P, _I, _P, _P],
    "hmg_hod": [_P, _I, _I, C.POINTER(HodParams), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "hmg_power": [_P, _I, _I, _I, C.POINTER(Tracer), C.POINTER(Tracer), _P, _P, _P, _P, _P, _P,
                  _D, _D, _P, _P],
    "hmg_power_batch": [_P, _I, _I, _I, _I, C.POINTER(Tracer), _I, C.POINTER(_I), C.POINTER(_I),
                        _P, _P, _P, _P, _P, _P, _D, _D, C.POINTER(_P), C.POINTER(_P)],
    "hmg_power_2halo_terms": [_P, _I, _I, _I, C.POINTER(Tracer), C.POINTER(Tracer), _P, _P, _P, _P, _P, _D,
                              _P, _P, _P],
    "hmg_sigma2_halo_front": [_P, _I, _I, _I, _P, _P, _P, _P, _D, _P, C.POINTER(HaloStageArgs), C.POINTER(HodPart),
                              C.POINTER(RowsPart)],
    "hmg_group_rows": [_P, _I, _I, _I, _I, C.POINTER(MassFnPart), C.POINTER(HodPart), C.POINTER(RowsPart),
                       C.POINTER(NfwPart)],
    "hmg_group_profile": [_P, _I, _I, _I, C.POINTER(ProfileFftPart), C.POINTER(HodPart), C.POINTER(PowerBatchDesc)],
    "hmg_group_tensors": [_P, _I, _I, _I, _I, C.POINTER(MassFnPart), C.POINTER(HodPart), C.POINTER(PowerBatchDesc),
                          C.POINTER(NfwPart), C.POINTER(ProfileFftPart)],
    "hmg_power_batch_run": [_P, _I, _I, _I, C.POINTER(PowerBatchDesc), _I],
    "hmg_add": [_P, _Z, _P, _P, _P],
    "hmg_limber": [_P, _I, _P, _I, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P],
    "hmg_fn2d": [_P, _I, _I, _I, _I, C.POINTER(_P), C.POINTER(_I), C.POINTER(_I), C.POINTER(_D), _I, _P],
    "hmg_mstellar_halo": [_P, _I, _I, _P, _P, _P],
    "hmg_trapz_rows": [_P, _I, _I, _P, _P, _P],
    "hmg_sine_transform": [_P, _I, _I, _P, _P, _P],
    "hmg_profile_fft_table": [_P, _I, _I, _I, _I, _D, _P, _P, _P, _I, _P, _P, _P, _P, _I, _P],
    "hmg_comm_unique_id": [C.c_char * COMM_ID_BYTES],
    "hmg_comm_init": [_P, C.c_char * COMM_ID_BYTES, _I, _I],
    "hmg_comm_allgather": [_P, _P, _P, _Z],
    "hmg_comm_allgather_multi": [_P, _I, C.POINTER(_P), C.POINTER(_P), _Z],
    "hmg_comm_gather_async": [_P, _I, C.POINTER(_P), C.POINTER(_P), _Z, _I, _I, _I],
    "hmg_comm_allgatherv_multi": [_P, _I, C.POINTER(_P), C.POINTER(_P), C.POINTER(_Z)],
    "hmg_comm_gatherv_async": [_P, _I, C.POINTER(_P), C.POINTER(_P), C.POINTER(_Z), _I, _I, _I],
    "hmg_comm_info": [_P, C.POINTER(_I), C.POINTER(_I)],
    "hmg_comm_barrier": [_P],
    "hmg_comm_destroy": [_P],
}

_lib = None


def kernel_source_sha16():
    """First 16 hex digits of the SHA-256 over the kernel sources and their build recipe: what a stored
    profile (profiles/rNN/*.json) records and bench.py compares, so that counters measured on one build are
    never quoted for another."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(_HERE, "csrc")
    names = ["hmgrid.hip", "longgrid.hip", "longgrid.hpp", "rowdev.hpp", "sici.hpp", "ldsfft.hpp", "fastmath.hpp", "Makefile"]
    names += sorted(os.path.join("kernels", n) for n in os.listdir(os.path.join(csrc, "kernels")) if n.endswith(".hpp"))
    for name in names:          # (runtime.hip / comm.hip / hmctx.hpp hold no device code: not part of the kernel identity)
        with open(os.path.join(csrc, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def load():
    """Load libhmgrid.so once; raise loudly if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is not built and there is no CPU "
            "fallback. Run `make -C hmvec_amd/csrc` (needs hipcc, --offload-arch=gfx950).")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI drifted
        fn.argtypes = argtypes
        fn.restype = C.c_int
    lib.hmg_last_error.argtypes = []
    lib.hmg_last_error.restype = C.c_char_p
    if lib.hmg_abi_version() != ABI_VERSION:
        raise ImportError(f"libhmgrid ABI {lib.hmg_abi_version()} != binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise NativeError(load().hmg_last_error().decode("utf-8", "replace"))


class DeviceArray:
    """A C-contiguous fp64 array living in HBM, owned by a Context."""

    __slots__ = ("ctx", "ptr", "shape", "_owner", "__weakref__")

    def __init__(self, ctx, ptr, shape, owner=True):
        self.ctx, self.ptr, self.shape, self._owner = ctx, ptr, tuple(int(s) for s in shape), owner

    @property
    def size(self):
        n = 1
        for s in self.shape:
            n *= s
        return n

    @property
    def nbytes(self):
        return self.size * 8

    def numpy(self):
        self.ctx.flush()
        out = np.empty(self.shape, dtype=np.float64)
        check(self.ctx.lib.hmg_memcpy_d2h(self.ctx.handle, out.ctypes.data, self.ptr, self.nbytes))
        return out

    def view(self, offset_elems, shape):
        """Non-owning window into this buffer (e.g. a z-slab)."""
        v = DeviceArray(self.ctx, self.ptr + 8 * int(offset_elems), shape, owner=False)
        v._owner = self  # keep parent alive
        return v

    def free(self):
        """Hand the block back to the context's free list (no device synchronisation).  The free list's
        invariant - whatever still touches a freed block was enqueued earlier on the same stream - only holds
        for work that HAS been enqueued.  Deferred stages hold raw addresses, so their owner must keep every
        array they refer to alive until the queue is issued: HaloModel keeps them in its buffer pool and input
        cache and issues the queue before it releases or replaces an entry of either (HaloModel._buf,
        HaloModel._release_inputs).  Flushing here instead would issue the queue whenever any unrelated temporary
        goes out of scope and split the grouped launches."""
        if self._owner is True and self.ptr and self.ctx is not None and self.ctx.handle:
            self.ctx.lib.hmg_free(self.ctx.handle, self.ptr)
        self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class PinnedArray:
    """Page-locked host block viewed as a numpy array: the destination of asynchronous D2H copies
    (hmg_memcpy_d2h_async).  Owns the block; numpy views into it keep this object alive through
    ``.base``."""

    def __init__(self, ctx, shape):
        self.ctx = ctx
        shape = (shape,) if np.isscalar(shape) else tuple(int(s) for s in shape)
        n = int(np.prod(shape)) if shape else 1
        p = C.c_void_p()
        check(ctx.lib.hmg_host_alloc(ctx.handle, n * 8, C.byref(p)))
        self.ptr = p.value
        buf = (C.c_double * n).from_address(self.ptr)
        buf._hmg_owner = self                     # the ctypes buffer is numpy's base object
        self.array = np.frombuffer(buf, dtype=np.float64).reshape(shape)

    def __del__(self):
        try:
            if self.ptr and self.ctx.handle:
                self.ctx.lib.hmg_host_free(self.ctx.handle, self.ptr)
            self.ptr = 0
        except Exception:
            pass


class Context:
    """One per GPU / process: stream, scratch, rocFFT plans, RCCL communicator."""

    def __init__(self, device=0):
        self.lib = load()
        h = C.c_void_p()
        check(self.lib.hmg_ctx_create(int(device), C.byref(h)))
        self.handle = h.value
        self.device = int(device)
        self.capture_serial = 0       # changes at every capture begin/end: events recorded before are off limits
        self._deferred = []           # objects with stages queued for a grouped launch (HaloModel), in order
        self._trace = None            # list of (name, args) while Context.trace records a launch-only sequence
        self.shared = {}              # id(read-only host array) -> (array, device copies ...): inputs several models share

    # deferred stages: a HaloModel queues the launch-only stages of a pass so that independent ones can
    # share a launch (hmg_group_*); ANY other native call of this context issues them first, so program
    # order is what every consumer sees
    def defer(self, owner):
        if owner not in self._deferred:
            self._deferred.append(owner)

    def flush(self):
        while self._deferred:
            self._deferred.pop(0)._flush()

    def call_now(self, name, *args):
        """A native call that does not flush the deferred stages (used while they are being issued).  Every
        stream-affecting native call of this class goes through here, so that a recorded call list
        (Context.trace) keeps lane switches, event records/waits and copies exactly as a capture does."""
        if self._trace is not None:
            self._trace.append((name, args))
        check(getattr(self.lib, name)(self.handle, *args))

    # call lists: the native calls of a launch-only sequence (a pass), recorded once and re-issued without the
    # Python facade in between - what a step that cannot be a HIP graph (event records with timestamps) uses so
    # that the host stays ahead of a 0.1 ms pass.  Same restrictions as a capture: same buffers, no allocation.
    def trace(self, fn):
        self.flush()
        self._trace = []
        try:
            fn()
            self.flush()
            return self._trace
        finally:
            self._trace = None

    def run_trace(self, calls):
        self.flush()
        lib, h = self.lib, self.handle
        for name, args in calls:
            check(getattr(lib, name)(h, *args))

    def close(self):
        self.shared = {}
        if getattr(self, "handle", None):
            self.lib.hmg_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # memory
    def empty(self, shape):
        shape = (int(shape),) if isinstance(shape, (int, np.integer)) else tuple(int(v) for v in shape)
        n = math.prod(shape)                     # (np.prod costs 3 us per call; a model makes ~60 of these)
        p = C.c_void_p()
        check(self.lib.hmg_malloc(self.handle, n * 8, C.byref(p)))
        return DeviceArray(self, p.value, shape)

    def upload(self, arr):
        a = np.ascontiguousarray(arr, dtype=np.float64)
        d = self.empty(a.shape)          # (a recycled block: its last owner released it, and owners of queued
                                         # stages issue the queue before they release anything - DeviceArray.free)
        check(self.lib.hmg_memcpy_h2d(self.handle, d.ptr, a.ctypes.data, a.nbytes))
        return d

    def write(self, dst, arr):
        """Overwrite DeviceArray ``dst`` with a host array (stream-ordered behind everything queued so far)."""
        self.flush()
        a = np.ascontiguousarray(arr, dtype=np.float64)
        if self._trace is not None:
            raise NativeError("Context.write inside Context.trace: a recorded call list cannot own host data")
        check(self.lib.hmg_memcpy_h2d(self.handle, dst.ptr, a.ctypes.data, a.nbytes))

    def copy(self, src):
        self.flush()
        d = self.empty(src.shape)
        self.call_now("hmg_memcpy_d2d", d.ptr, src.ptr, src.nbytes)
        return d

    def sync(self):
        self.flush()
        check(self.lib.hmg_sync(self.handle))

    def record(self, slot):
        self.flush()
        self.call_now("hmg_event_record", slot)

    def lane(self, i):
        """Route subsequent launches to lane i (0 = main stream)."""
        self.flush()
        self.call_now("hmg_lane_set", i)

    def wait(self, slot):
        """Current lane waits for the event last recorded in `slot`."""
        self.flush()
        self.call_now("hmg_event_wait", slot)

    def elapsed_ms(self, s0, s1):
        ms = C.c_double()
        check(self.lib.hmg_elapsed_ms(self.handle, s0, s1, C.byref(ms)))
        return ms.value

    _NO_FLUSH = frozenset(["hmg_bracket_next", "hmg_profile_support_epoch"])      # calls that enqueue nothing and read nothing

    def call(self, name, *args):
        if name not in self._NO_FLUSH:
            self.flush()
        if self._trace is not None:
            self._trace.append((name, args))
        check(getattr(self.lib, name)(self.handle, *args))

    # captured steps
    def capture(self, fn):
        """Run ``fn()`` (launch-only: no allocation, upload, download or synchronisation) inside a
        HIP-graph capture and return the graph id for ``replay``."""
        self.flush()                      # stages queued before the capture are not part of it
        check(self.lib.hmg_graph_begin(self.handle))
        self.capture_serial += 1
        try:
            fn()
            self.flush()                  # ... and those queued inside it are
        except BaseException:
            for o in self._deferred:       # whatever was queued inside the failed capture is dropped with it
                o._stages = []
            self._deferred.clear()
            self.lib.hmg_graph_abort(self.handle)
            self.capture_serial += 1
            raise
        gid = C.c_int()
        try:
            check(self.lib.hmg_graph_end(self.handle, C.byref(gid)))
        finally:
            self.capture_serial += 1
        return gid.value

    def graph_kernel_nodes(self, gid):
        """Kernel launches the captured step `gid` holds."""
        n = C.c_int()
        check(self.lib.hmg_graph_kernel_nodes(self.handle, gid, C.byref(n)))
        return n.value

    def replay(self, gid):
        self.flush()
        self.call_now("hmg_graph_launch", gid)

    def copy_to_pinned(self, pinned, src):
        """Asynchronous D2H of DeviceArray ``src`` into PinnedArray ``pinned`` on the current lane."""
        self.flush()
        self.call_now("hmg_memcpy_d2h_async", pinned.ptr, src.ptr, src.nbytes)


    def copy_from_pinned(self, dst, pinned):
        """Asynchronous H2D of PinnedArray ``pinned`` into DeviceArray ``dst`` on the current lane."""
        self.flush()
        self.call_now("hmg_memcpy_h2d_async", dst.ptr, pinned.ptr, dst.nbytes)

    def event_synchronize(self, slot):
        """Block the host until the event last recorded in ``slot`` has happened (other work keeps running)."""
        check(self.lib.hmg_event_synchronize(self.handle, slot))


_default_ctx = {}


def default_context(device=0):
    """Process-wide shared Context per device (streams, scratch, FFT tables are reused by
    every HaloModel that is not given its own)."""
    c = _default_ctx.get(device)
    if c is None or not c.handle:
        c = _default_ctx[device] = Context(device)
    return c


def ptr(x):
    """Device pointer of a DeviceArray or NULL."""
    return None if x is None else x.ptr
