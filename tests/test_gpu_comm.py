"""RCCL plumbing inside libhmgrid on one GPU: a 1-rank communicator exercises the same entry
points the multi-GPU path uses (unique id -> file rendezvous -> ncclCommInitRank -> grouped
all-gather -> all-reduce barrier).  The >1-rank logic is rehearsed on CPU in test_shard_gloo.py."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_single_rank_communicator_roundtrip(tmp_path, monkeypatch):
    from hmvec_amd import _native as nat
    from hmvec_amd.dist import RcclComm
    monkeypatch.setenv("HMG_RDZV_DIR", str(tmp_path))
    ctx = nat.Context(0)
    comm = RcclComm(ctx, 0, 1, f"test_{os.getpid()}", force_init=True)
    rng = np.random.default_rng(1)
    a = [rng.standard_normal((3, 17)) for _ in range(4)]
    sends = [ctx.upload(x) for x in a]
    recvs = [ctx.empty((3, 17)) for _ in a]
    comm.allgather_rows(sends, recvs)
    comm.barrier()
    for x, r in zip(a, recvs):
        assert np.array_equal(r.numpy(), x)
    got = comm.allgather_host([1.5, -2.0])
    assert got.shape == (1, 2) and np.array_equal(got[0], [1.5, -2.0])
    comm.close()
    assert not any(f.startswith("hmg_rdzv_") for f in os.listdir(tmp_path))   # rendezvous file removed
    ctx.close()
