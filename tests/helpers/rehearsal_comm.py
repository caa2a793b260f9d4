"""File transport that REHEARSES the N > 1 flow where RCCL cannot run (test scaffolding, not product code).

RCCL refuses two ranks on one device, so on a one-GPU box a two-rank run stops at ncclCommInitRank.  With this
communicator the ranks exchange through files instead (device -> host -> file -> host -> device, blocking), which
exercises everything else of the multi-rank path - slab bounds, the gather landing every slab in place, barriers,
max-over-ranks timing.  Same interface as hmvec_amd.dist.RcclComm minus ``gather_rows_async`` (ShardedSpectra then
takes its event-ordered ``allgather_rows`` branch).  Its timings mean nothing."""
import os
import time

import numpy as np

from hmvec_amd import _native as nat


class HostRehearsalComm:
    """See the module docstring."""

    def __init__(self, ctx, rank, world, tag, directory=None):
        self.ctx, self.rank, self.world = ctx, rank, world
        self._dir = directory or os.environ.get("HMG_REHEARSAL_DIR", "/dev/shm")
        self._base = os.path.join(self._dir, f"hmg_reh_{tag}_w{world}_pp{os.getppid()}")
        self._seq = 0

    # -- pure-host core (CPU-testable): every rank contributes one array, every rank gets all of them
    def exchange(self, arr):
        self._seq += 1
        mine = f"{self._base}_s{self._seq}_r{self.rank}.npy"
        tmp = mine + ".tmp"
        with open(tmp, "wb") as f:
            np.save(f, np.ascontiguousarray(arr, dtype=np.float64))
        os.replace(tmp, mine)
        deadline = time.time() + float(os.environ.get("HMG_RDZV_TIMEOUT", "600"))
        parts = []
        for r in range(self.world):
            path = f"{self._base}_s{self._seq}_r{r}.npy"
            while not os.path.exists(path):
                if time.time() > deadline:
                    raise TimeoutError(f"rank {r} never wrote {path}")
                time.sleep(0.0005)
            parts.append(np.load(path))
        # Having read round s, every rank had written its file of round s, i.e. had finished reading round
        # s-1 (a rank writes round s only after that) - so my file of round s-1 has no reader left.  (Not
        # round s itself: a slower rank may still be reading it.)
        old = f"{self._base}_s{self._seq - 1}_r{self.rank}.npy"
        if os.path.exists(old):
            os.remove(old)
        return parts

    def allgather_rows(self, sends, recvs, counts=None):
        self.ctx.sync()                               # (blocking by design: the lanes are drained first)
        flat = np.concatenate([s.numpy().reshape(-1) for s in sends])
        parts = self.exchange(flat)
        for i, rcv in enumerate(recvs):
            # rank r's part holds len(sends) arrays of its own slab length (unequal slabs: counts[r])
            full = np.concatenate([p[i * (p.size // len(sends)):(i + 1) * (p.size // len(sends))] for p in parts])
            assert counts is None or [p.size // len(sends) for p in parts] == list(counts)
            nat.check(self.ctx.lib.hmg_memcpy_h2d(self.ctx.handle, rcv.ptr, full.ctypes.data, full.nbytes))

    def allgather_host(self, values):
        return np.stack(self.exchange(np.ascontiguousarray(values, dtype=np.float64)))

    def barrier(self):
        self.exchange(np.zeros(1))

    def info(self):
        return self.rank, self.world

    def close(self):
        # The file of the last round must outlive this rank: a slower rank may not have read it yet.  It is a
        # few bytes; the test's tmp_path goes away with the test.
        self.barrier()

    @staticmethod
    def cleanup(tag, world, parent_pid, directory=None):
        import glob
        d = directory or os.environ.get("HMG_REHEARSAL_DIR", "/dev/shm")
        for f in glob.glob(os.path.join(d, f"hmg_reh_{tag}_w{world}_pp{parent_pid}_s*_r*.npy*")):
            try:
                os.remove(f)
            except OSError:
                pass
