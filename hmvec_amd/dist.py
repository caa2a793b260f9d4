"""z-slab sharding of the hot path across the GPUs of one node.

Every stage of the path is independent along z (the mass axis is the reduction axis and
must not be split), so rank r owns a contiguous slab of redshifts, builds everything for
its slab with no communication, and the per-rank (nz_local, nk) spectra are joined with a
single RCCL all-gather group over xGMI.  Because slabs are contiguous in z and spectra are
[z][k] row-major, the all-gather lands every slab directly in its final position.

The communicator interface is two methods (``allgather_rows``, ``barrier``): ``RcclComm``
is the product implementation (RCCL inside libhmgrid); tests drive the same sharding logic
with a torch.distributed/gloo communicator on CPU.
"""
import ctypes as C
import os
import time

import numpy as np

from . import _native as nat


def slab_bounds(nz, world, rank):
    """Contiguous z-slab [lo, hi) of `rank`; requires nz % world == 0 so that a plain
    all-gather (equal counts) reassembles the full grid."""
    if nz % world != 0:
        raise ValueError(f"nz={nz} must be divisible by the number of ranks ({world})")
    per = nz // world
    return rank * per, (rank + 1) * per


def rendezvous_path(tag, world):
    """One file per launch: the tag (MASTER_PORT + run id from the launcher), the world size
    and the launcher's PID, which all ranks of one torch.distributed.run share as parent."""
    name = f"hmg_rdzv_{tag}_w{world}_pp{os.getppid()}"
    return os.path.join(os.environ.get("HMG_RDZV_DIR", "/tmp"), name)


def exchange_unique_id(ctx, rank, world, tag):
    """Rank 0 creates the RCCL unique id and publishes it through a file; the other ranks
    of the node poll for it.  (One node only: SURVEY §8e; the id is 128 opaque bytes.)
    Files older than 10 minutes are ignored as leftovers of a crashed launch."""
    path = rendezvous_path(tag, world)
    buf = C.create_string_buffer(nat.COMM_ID_BYTES)
    if rank == 0:
        nat.check(ctx.lib.hmg_comm_unique_id(buf))
        tmp = path + f".tmp{os.getpid()}"
        with open(tmp, "wb") as f:
            f.write(buf.raw)
        os.replace(tmp, path)
        return buf
    deadline = time.time() + float(os.environ.get("HMG_RDZV_TIMEOUT", "600"))   # first library page-in on a fresh box can take minutes
    while time.time() < deadline:
        try:
            fresh = time.time() - os.path.getmtime(path) < 600.0
            with open(path, "rb") as f:
                raw = f.read()
            if fresh and len(raw) == nat.COMM_ID_BYTES:
                buf.raw = raw
                return buf
        except FileNotFoundError:
            pass
        time.sleep(0.01)
    raise TimeoutError(f"no RCCL unique id at {path}")


class RcclComm:
    """RCCL communicator owned by the native context."""

    def __init__(self, ctx, rank, world, tag, force_init=False):
        self.ctx, self.rank, self.world = ctx, rank, world
        self._path = None
        self._inited = False
        if world > 1 or force_init:      # force_init: exercise RCCL with a 1-rank communicator (tests)
            self._inited = True
            uid = exchange_unique_id(ctx, rank, world, tag)
            ctx.call("hmg_comm_init", uid, rank, world)
            self._path = rendezvous_path(tag, world)

    def allgather_rows(self, sends, recvs):
        """sends[i]: DeviceArray (nz_local, nk) -> recvs[i]: DeviceArray (nz, nk); one group launch."""
        n = len(sends)
        sp = (C.c_void_p * n)(*[s.ptr for s in sends])
        rp = (C.c_void_p * n)(*[r.ptr for r in recvs])
        self.ctx.call("hmg_comm_allgather_multi", n, sp, rp, sends[0].size)

    def allgather_host(self, values):
        """Small host-side all-gather of a float vector (timings); blocks."""
        v = np.ascontiguousarray(values, dtype=np.float64)
        d_s = self.ctx.upload(v)
        d_r = self.ctx.empty((self.world, v.size))
        self.ctx.call("hmg_comm_allgather", d_s.ptr, d_r.ptr, v.size)
        return d_r.numpy()

    def barrier(self):
        self.ctx.call("hmg_comm_barrier")

    def close(self):
        if self._inited:
            self.ctx.call("hmg_comm_destroy")
            if self.rank == 0 and self._path:
                try:
                    os.remove(self._path)
                except OSError:
                    pass


class ShardedSpectra:
    """Runs a list of (name, name2) spectra on this rank's z-slab model and gathers the
    full-z (nz, nk) results on every rank.

    `model` is a HaloModel built on the slab redshifts; `comm` provides allgather_rows.
    """

    _EV_SPECTRA, _EV_GATHERED = 8, 9      # event slots (HaloModel uses 0-3, bench.py 16 and up)
    _COMM_LANE = 3

    def __init__(self, model, comm, nz_total, pairs, force_gather=False):
        """force_gather: take the gather path (separate `full` buffers, communication lane) even with a
        single rank - used to exercise it on one GPU."""
        self.model, self.comm, self.pairs = model, comm, list(pairs)
        self._gather = comm.world > 1 or force_gather
        ctx = model._ctx()
        nzl, nk = model.zs.size, model.ks.size
        if nzl * comm.world != nz_total:
            raise ValueError("slab size x ranks != nz_total")
        self.local = [ctx.empty((nzl, nk)) for _ in range(2 * len(self.pairs))]
        if self._gather:
            self.full = [ctx.empty((nz_total, nk)) for _ in range(2 * len(self.pairs))]
        else:
            self.full = self.local

    def run(self, bracket=None, batched=True):
        """Launch all spectra + the gather; asynchronous (no host sync).  `bracket` = (s0, s1)
        event slots around the mass-integral kernel (batched mode) for bench.py."""
        m = self.model
        ctx = m._ctx()
        gather = self._gather
        if gather:
            # the previous call's gather (on its own lane) must have finished reading `local`
            ctx.wait(self._EV_GATHERED)
        if batched:
            if bracket is not None:
                ctx.call("hmg_bracket_next", nat.KERNEL_POWER, bracket[0], bracket[1])
            m.power_device_batch(self.pairs, self.local[0::2], self.local[1::2])
        else:
            for i, (a, b) in enumerate(self.pairs):
                m.power_device(a, b, out1=self.local[2 * i], out2=self.local[2 * i + 1])
        if gather:
            # The collective runs on the communication lane behind an event, so that the next pass of
            # the path (which does not touch `full`, and `local` only after the wait above) overlaps it:
            # over xGMI the all-gather of a 0.2 ms slab step would otherwise be a visible fraction of it.
            ctx.record(self._EV_SPECTRA)
            ctx.lane(self._COMM_LANE)
            ctx.wait(self._EV_SPECTRA)
            self.comm.allgather_rows(self.local, self.full)
            ctx.record(self._EV_GATHERED)
            ctx.lane(0)

    def results(self):
        """{(a,b): (P1h, P2h)} as numpy (nz_total, nk) arrays; blocks."""
        return {p: (self.full[2 * i].numpy(), self.full[2 * i + 1].numpy())
                for i, p in enumerate(self.pairs)}


def mthresh_from_ngal_global(zs_full, ks, ms, ngal_full, **model_kwargs):
    """Stellar-mass thresholds for a target galaxy density on the FULL redshift grid.

    The reference's bisection (hmvec/utils.py:9-42, called at hmvec/hmvec.py:426-433) stops only
    when EVERY redshift has converged, so its result depends on the whole z vector: a slab model
    that bisects only its own redshifts gets slightly different thresholds (dP_gg ~ 5e-5,
    SURVEY 8e).  Every rank therefore runs the (tiny: (nz, nm) per iteration) bisection for the
    full grid redundantly and then passes ``mthresh=result[lo:hi]`` to ``add_hod`` of its slab.
    Returns mthresh with shape (nz_full,).
    """
    from .halomodel import HaloModel
    full = HaloModel(np.asarray(zs_full, dtype=np.float64), ks, ms=ms, skip_nfw=True, **model_kwargs)
    pparams = {k: full.p[k] for k in HaloModel._HOD_PARAMS}
    log10mthresh = full._bisect_mthresh(np.asarray(ngal_full, dtype=np.float64), pparams)
    return 10 ** (log10mthresh * pparams["hod_A_log10mthresh"])
