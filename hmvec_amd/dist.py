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
    """Contiguous z-slab [lo, hi) of `rank`.  nz need not be a multiple of the number of ranks (the README grid
    has nz = 20, /root/reference/README.rst:55): the first nz % world ranks take one redshift more, and the gather
    moves per-rank counts (slab_counts).  Every rank must own at least one redshift."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside 0..{world - 1}")
    if nz < world:
        raise ValueError(f"nz={nz} redshifts cannot be split over {world} ranks (a rank would own none)")
    base, rem = divmod(nz, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def slab_counts(nz, world):
    """Redshifts per rank, in rank order (sums to nz)."""
    return [hi - lo for lo, hi in (slab_bounds(nz, world, r) for r in range(world))]


def rendezvous_path(tag, world):
    """One file per launch: the tag (MASTER_PORT + run id from the launcher), the world size,
    the launcher's PID (all ranks of one launch share it as parent) and, under an elastic
    launcher, the restart count - so a restarted group never reads its predecessor's id."""
    restart = os.environ.get("TORCHELASTIC_RESTART_COUNT", "0")
    named = _launch_name()
    who = "named" if named else f"pp{os.getppid()}"       # (the file NAME only has to be common to the ranks; the
    name = f"hmg_rdzv_{tag}_w{world}_{who}_r{restart}"     # nonce inside it is what identifies the launch)
    return os.path.join(os.environ.get("HMG_RDZV_DIR", "/tmp"), name)


def _parent_start_ticks():
    """Start time of the parent process in clock ticks since boot (field 22 of /proc/<pid>/stat): together
    with its PID it names ONE launcher for the lifetime of the machine, whatever the wall clock does."""
    try:
        with open(f"/proc/{os.getppid()}/stat", "rb") as f:
            return f.read().rsplit(b")", 1)[1].split()[19].decode()
    except (OSError, IndexError):
        return "0"


def _launch_name():
    """The name the launcher gave this launch, if it gave one that identifies it: HMG_LAUNCH_TAG, or torchrun's
    TORCHELASTIC_RUN_ID unless that is its default ("none" - the same for every launch that does not pass --rdzv-id,
    so a crashed earlier launch on the same port would share it)."""
    tag = os.environ.get("HMG_LAUNCH_TAG")
    if tag:
        return tag
    run_id = os.environ.get("TORCHELASTIC_RUN_ID", "")
    return run_id if run_id not in ("", "none") else None


def _launch_salt():
    """Something only THIS launch has, for launches that carry a name: the name alone (a fixed --rdzv-id in a job
    script, a fixed HMG_LAUNCH_TAG) is shared with a crashed earlier launch of the same job, whose rendezvous file
    would then carry the same nonce.  HMG_LAUNCH_NONCE (bench.py's launcher exports a fresh one per launch), else the
    per-launch directory torchrun creates for its error files (TORCHELASTIC_ERROR_FILE =
    <log dir>/<run id>_<random>/attempt_<n>/<rank>/error.json: two levels up is common to the ranks of an attempt
    and random per launch).  Empty when the launcher offers neither: see _stale()."""
    salt = os.environ.get("HMG_LAUNCH_NONCE")
    if salt:
        return salt
    ef = os.environ.get("TORCHELASTIC_ERROR_FILE")
    if ef:
        return os.path.dirname(os.path.dirname(ef))
    return ""


def _parent_start_wall():
    """Wall-clock start of the parent process (boot time + start ticks), or None."""
    try:
        with open("/proc/stat", "rb") as f:
            btime = next(int(l.split()[1]) for l in f if l.startswith(b"btime"))
        return btime + int(_parent_start_ticks()) / os.sysconf("SC_CLK_TCK")
    except (OSError, StopIteration, ValueError):
        return None


def _stale(path):
    """A NAMED launch without a per-launch salt cannot tell its own file from the leftover of a crashed launch of the
    same name by content.  Its readers then ignore a file written before their parent process started (two seconds of
    slack for clock granularity): a leftover predates the whole launch, while rank 0 of this launch writes after the
    launcher - every rank's parent or an ancestor of it - has started.  (A per-rank wrapper that starts AFTER rank 0
    has already published would wait out its time limit: launchers that wrap ranks should export HMG_LAUNCH_NONCE.)"""
    if not _launch_name() or _launch_salt():
        return False
    t0 = _parent_start_wall()
    try:
        return t0 is not None and os.stat(path).st_mtime < t0 - 2.0
    except OSError:
        return False


def launch_identity(tag, world):
    """Text that every rank of THIS launch knows and no earlier launch could have written.  A launcher that
    names the launch says so: HMG_LAUNCH_TAG, or torchrun's TORCHELASTIC_RUN_ID (with the restart count) - these
    hold whatever sits between the launcher and the ranks (a per-rank wrapper script gives every rank a different
    parent) - together with the launch's salt (_launch_salt), since a name can be the same from launch to launch.
    Only when neither is set does the identity fall back on the parent process: its PID and start time."""
    restart = os.environ.get("TORCHELASTIC_RESTART_COUNT", "0")
    named = _launch_name()
    who = f"id:{named}+{_launch_salt()}" if named else f"pp{os.getppid()}@{_parent_start_ticks()}"
    return f"{tag}|w{world}|{who}|r{restart}"


def launch_nonce(tag, world):
    """The launch identity as stored in the rendezvous file in front of the id and compared by the readers - a
    leftover of a crashed earlier launch that happens to have the same name (same port, same run id, a recycled
    PID) is rejected by content, not by how old it looks.  SHA-256 of the text (hex), so that a long tag cannot
    push part of the identity out of the fixed-size field."""
    import hashlib
    return hashlib.sha256(launch_identity(tag, world).encode()).hexdigest().encode().ljust(NONCE_BYTES, b"\0")


NONCE_BYTES = 96


def exchange_unique_id(ctx, rank, world, tag):
    """Rank 0 creates the RCCL unique id and publishes it through a file; the other ranks
    of the node poll for it.  (One node only: SURVEY 8e; the id is 128 opaque bytes.)
    Rank 0 removes any leftover of an earlier launch first and creates the file exclusively
    with mode 0600; the file carries the launch nonce in front of the id and the other ranks accept
    only a file with THEIR nonce, so the id of a crashed earlier launch is never picked up."""
    path = rendezvous_path(tag, world)
    nonce = launch_nonce(tag, world)
    buf = C.create_string_buffer(nat.COMM_ID_BYTES)
    if rank == 0:
        nat.check(ctx.lib.hmg_comm_unique_id(buf))
        tmp = path + f".tmp{os.getpid()}"
        for stale in (path, tmp):
            try:
                os.unlink(stale)
            except FileNotFoundError:
                pass
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
        with os.fdopen(fd, "wb") as f:
            f.write(nonce + buf.raw)
        os.replace(tmp, path)
        return buf
    # (the first library page-in on a fresh box can take minutes; a launcher that supervises its ranks sets
    # a shorter limit: bench.py)
    deadline = time.time() + float(os.environ.get("HMG_RDZV_TIMEOUT", "600"))
    seen = None
    while time.time() < deadline:
        try:
            with open(path, "rb") as f:
                raw = f.read()
            if len(raw) == NONCE_BYTES + nat.COMM_ID_BYTES:
                if raw[:NONCE_BYTES] == nonce and not _stale(path):
                    buf.raw = raw[NONCE_BYTES:]
                    return buf
                seen = raw[:NONCE_BYTES]
        except FileNotFoundError:
            pass
        time.sleep(0.01)
    found = "no file" if seen is None else f"a file with nonce {seen.rstrip(bytes(1)).decode(errors='replace')[:16]}..."
    raise TimeoutError(f"no RCCL unique id of this launch at {path}: expected nonce "
                       f"{nonce.rstrip(bytes(1)).decode()[:16]}... (identity '{launch_identity(tag, world)}'), found {found}")


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

    def _counts(self, counts):
        """Per-rank element counts as a size_t array, or None when all ranks send the same amount."""
        if counts is None or len(set(counts)) <= 1:
            return None
        if len(counts) != self.world:
            raise ValueError("one count per rank")
        return (C.c_size_t * self.world)(*[int(c) for c in counts])

    def allgather_rows(self, sends, recvs, counts=None):
        """sends[i]: DeviceArray (nz_local, nk) -> recvs[i]: DeviceArray (nz, nk); one group launch.
        counts: elements per rank when the slabs are unequal (same list on every rank)."""
        n = len(sends)
        sp = (C.c_void_p * n)(*[s.ptr for s in sends])
        rp = (C.c_void_p * n)(*[r.ptr for r in recvs])
        cv = self._counts(counts)
        if cv is None:
            self.ctx.call("hmg_comm_allgather_multi", n, sp, rp, sends[0].size)
        else:
            self.ctx.call("hmg_comm_allgatherv_multi", n, sp, rp, cv)

    def gather_rows_async(self, sends, recvs, ready_slot, done_slot, comm_lane, counts=None):
        """allgather_rows on the communication lane, ordered by events (one native call)."""
        n = len(sends)
        sp = (C.c_void_p * n)(*[s.ptr for s in sends])
        rp = (C.c_void_p * n)(*[r.ptr for r in recvs])
        cv = self._counts(counts)
        if cv is None:
            self.ctx.call("hmg_comm_gather_async", n, sp, rp, sends[0].size, ready_slot, done_slot, comm_lane)
        else:
            self.ctx.call("hmg_comm_gatherv_async", n, sp, rp, cv, ready_slot, done_slot, comm_lane)

    def allgather_host(self, values):
        """Small host-side all-gather of a float vector (timings); blocks."""
        v = np.ascontiguousarray(values, dtype=np.float64)
        d_s = self.ctx.upload(v)
        d_r = self.ctx.empty((self.world, v.size))
        self.ctx.call("hmg_comm_allgather", d_s.ptr, d_r.ptr, v.size)
        return d_r.numpy()

    def barrier(self):
        self.ctx.call("hmg_comm_barrier")

    def info(self):
        """(rank, nranks) as RCCL itself reports them; (None, None) when no communicator was created
        (a single rank needs none)."""
        if not self._inited:
            return None, None
        r, n = C.c_int(), C.c_int()
        self.ctx.call("hmg_comm_info", C.byref(r), C.byref(n))
        return r.value, n.value

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
        lo, hi = slab_bounds(nz_total, comm.world, comm.rank)
        if nzl != hi - lo:
            raise ValueError(f"rank {comm.rank} of {comm.world} owns {hi - lo} of {nz_total} redshifts, its model has {nzl}")
        zc = slab_counts(nz_total, comm.world)
        self.counts = [c * nk for c in zc] if len(set(zc)) > 1 else None     # None: equal slabs, plain all-gather
        self.local = [ctx.empty((nzl, nk)) for _ in range(2 * len(self.pairs))]
        if self._gather:
            self.full = [ctx.empty((nz_total, nk)) for _ in range(2 * len(self.pairs))]
        else:
            self.full = self.local

    def launch_spectra(self, bracket=None, batched=True):
        """The mass integrals of all pairs into the local (slab) buffers; launch-only, capturable."""
        m = self.model
        ctx = m._ctx()
        if batched:
            if bracket is not None:
                ctx.call("hmg_bracket_next", nat.KERNEL_POWER, bracket[0], bracket[1])
            m.power_device_batch(self.pairs, self.local[0::2], self.local[1::2])
        else:
            for i, (a, b) in enumerate(self.pairs):
                m.power_device(a, b, out1=self.local[2 * i], out2=self.local[2 * i + 1])

    def wait_gathered(self):
        """Order the next overwrite of the local buffers behind the previous pass's gather."""
        if self._gather:
            self.model._ctx().wait(self._EV_GATHERED)

    def gather(self):
        """The collective of one pass, on the communication lane behind an event: the next pass of the
        path (which does not touch `full`, and `local` only after wait_gathered) overlaps it - over
        xGMI the all-gather of a 0.1 ms slab step would otherwise be a visible fraction of it."""
        if not self._gather:
            return
        kw = {} if self.counts is None else {"counts": self.counts}
        if hasattr(self.comm, "gather_rows_async"):
            self.comm.gather_rows_async(self.local, self.full, self._EV_SPECTRA, self._EV_GATHERED, self._COMM_LANE, **kw)
            return
        ctx = self.model._ctx()            # communicators without the fused entry point (tests/helpers)
        ctx.record(self._EV_SPECTRA)
        ctx.lane(self._COMM_LANE)
        ctx.wait(self._EV_SPECTRA)
        self.comm.allgather_rows(self.local, self.full, **kw)
        ctx.record(self._EV_GATHERED)
        ctx.lane(0)

    def run(self, bracket=None, batched=True):
        """Launch all spectra + the gather; asynchronous (no host sync).  `bracket` = (s0, s1)
        event slots around the mass-integral kernel (batched mode) for bench.py."""
        self.wait_gathered()
        self.launch_spectra(bracket, batched)
        self.gather()

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
