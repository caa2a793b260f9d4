"""World-size-2 CPU rehearsal of the multi-GPU path (prompt §5): contiguous z-slabs, no
data-path collective on the way in, one all-gather at the end.  The product's sharding code
runs unchanged with gloo standing in for RCCL and the CPU oracle for the HIP engine."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO
from hmvec_amd.dist import slab_bounds, slab_counts


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_slab_bounds():
    assert [slab_bounds(32, 8, r) for r in (0, 3, 7)] == [(0, 4), (12, 16), (28, 32)]
    assert slab_bounds(20, 1, 0) == (0, 20)
    # nz need not divide: the README grid (nz = 20) on 8 ranks - four slabs of 3, four of 2, contiguous, in order
    b = [slab_bounds(20, 8, r) for r in range(8)]
    assert b == [(0, 3), (3, 6), (6, 9), (9, 12), (12, 14), (14, 16), (16, 18), (18, 20)]
    assert slab_counts(20, 8) == [3, 3, 3, 3, 2, 2, 2, 2] and slab_counts(5, 2) == [3, 2]
    for nz, world in ((7, 3), (33, 8), (8, 8), (9, 8)):
        bb = [slab_bounds(nz, world, r) for r in range(world)]
        assert bb[0][0] == 0 and bb[-1][1] == nz and all(x[1] == y[0] for x, y in zip(bb, bb[1:]))
        assert max(h - l for l, h in bb) - min(h - l for l, h in bb) <= 1
    with pytest.raises(ValueError):
        slab_bounds(3, 8, 0)           # a rank would own no redshift


@pytest.mark.parametrize("nz", [6, 5])
def test_two_rank_zslab_gather_matches_full_grid(tmp_path, nz):
    """nz = 6: equal slabs (one all-gather); nz = 5: slabs of 3 and 2 redshifts (per-rank counts)."""
    port = free_port()
    worker = os.path.join(REPO, "tests", "helpers", "shard_worker.py")
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, worker, str(tmp_path), str(nz)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=240)
        assert p.returncode == 0, out.decode()[-2000:]
    full = np.load(tmp_path / "full.npz")
    for rank in range(2):
        got = np.load(tmp_path / f"rank{rank}.npz")
        assert set(got.files) == set(full.files)
        for k in full.files:
            # every stage is independent along z: slab results are bit-identical;
            # the secant mass conversion is allowed its documented <=7e-15
            assert got[k].shape == full[k].shape
            assert np.allclose(got[k], full[k], rtol=1e-12, atol=0), k


def test_unique_id_file_rendezvous(tmp_path, monkeypatch):
    """The 128-byte RCCL id travels from rank 0 to the other ranks through a file named after
    the launch (MASTER_PORT, run id, world size, launcher pid) that carries the launch nonce in front of the id;
    a file with another nonce is ignored, whatever its age."""
    import ctypes
    import threading
    import time
    from hmvec_amd import _native as nat
    from hmvec_amd import dist

    monkeypatch.setenv("HMG_RDZV_DIR", str(tmp_path))

    class StubLib:
        @staticmethod
        def hmg_comm_unique_id(buf):
            ctypes.memmove(buf, bytes(range(128)), 128)
            return 0

    class StubCtx:
        lib = StubLib()

    tag = "29500_none"
    path = dist.rendezvous_path(tag, 2)
    # a leftover of a crashed earlier launch with the SAME name (same port, same run id, a recycled launcher
    # PID) must not be picked up, however fresh it looks: it carries another launch's nonce
    other = bytearray(dist.launch_nonce(tag, 2))
    other[-1] ^= 0x55
    with open(path, "wb") as f:
        f.write(bytes(other) + b"\xff" * nat.COMM_ID_BYTES)
    got = {}

    def reader():
        got["id"] = dist.exchange_unique_id(StubCtx(), 1, 2, tag).raw

    t = threading.Thread(target=reader)
    t.start()
    time.sleep(0.2)
    assert "id" not in got                       # still waiting: the stale file was rejected
    uid = dist.exchange_unique_id(StubCtx(), 0, 2, tag)
    t.join(timeout=10)
    assert got["id"] == uid.raw == bytes(range(128))


def test_launch_identity_survives_wrappers_and_long_tags(monkeypatch):
    """the nonce must not depend on the parent process when the launcher names the launch (a per-rank
    wrapper script gives every rank another parent), and a long tag must not push part of it out of the field."""
    from hmvec_amd import dist
    monkeypatch.delenv("HMG_LAUNCH_TAG", raising=False)
    monkeypatch.delenv("TORCHELASTIC_RUN_ID", raising=False)
    by_parent = dist.launch_nonce("29500_x", 2)
    assert f"pp{os.getppid()}" in dist.launch_identity("29500_x", 2)
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "run-" + "x" * 300)
    monkeypatch.setenv("TORCHELASTIC_RESTART_COUNT", "0")
    a = dist.launch_nonce("29500_x", 2)
    assert "pp" not in dist.launch_identity("29500_x", 2).split("|")[2] and a != by_parent
    assert len(a) == dist.NONCE_BYTES and a.rstrip(b"\0").isalnum()
    monkeypatch.setenv("TORCHELASTIC_RESTART_COUNT", "1")          # a restarted group: another identity
    assert dist.launch_nonce("29500_x", 2) != a
    monkeypatch.setenv("TORCHELASTIC_RESTART_COUNT", "0")
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "run-" + "x" * 299 + "y")   # differs only beyond byte 96 of the text
    assert dist.launch_nonce("29500_x", 2) != a
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "none")              # torchrun's default names nothing: parent identity
    assert f"pp{os.getppid()}" in dist.launch_identity("29500_x", 2)
    monkeypatch.setenv("HMG_LAUNCH_TAG", "mine")                    # the explicit tag wins
    assert "id:mine" in dist.launch_identity("29500_x", 2)


def test_a_named_launch_rejects_the_leftover_of_an_earlier_launch_with_the_same_name(tmp_path, monkeypatch):
    """a launch that carries a name (HMG_LAUNCH_TAG, a fixed --rdzv-id) shares that name - file path and,
    without more, nonce - with a crashed earlier launch of the same job.  (a) With a per-launch salt (HMG_LAUNCH_NONCE
    from bench.py's launcher, or torchrun's per-launch error-file directory) the nonces differ: the leftover is
    rejected by content.  (b) Without one the readers ignore a file written before their parent process started."""
    import ctypes
    import threading
    import time
    from hmvec_amd import _native as nat
    from hmvec_amd import dist

    monkeypatch.setenv("HMG_RDZV_DIR", str(tmp_path))
    monkeypatch.setenv("HMG_LAUNCH_TAG", "nightly-job")
    monkeypatch.delenv("TORCHELASTIC_ERROR_FILE", raising=False)

    class StubLib:
        @staticmethod
        def hmg_comm_unique_id(buf):
            ctypes.memmove(buf, bytes(range(128)), 128)
            return 0

    class StubCtx:
        lib = StubLib()

    tag = "29500_fixed"
    path = dist.rendezvous_path(tag, 2)

    def leftover(nonce):
        with open(path, "wb") as f:
            f.write(nonce + b"\xee" * nat.COMM_ID_BYTES)

    def read_with(expect_wait):
        got = {}
        t = threading.Thread(target=lambda: got.update(id=dist.exchange_unique_id(StubCtx(), 1, 2, tag).raw))
        t.start()
        time.sleep(0.2)
        assert ("id" not in got) == expect_wait
        uid = dist.exchange_unique_id(StubCtx(), 0, 2, tag)
        t.join(timeout=10)
        assert got["id"] == uid.raw == bytes(range(128))

    # (a) the earlier launch had another salt
    monkeypatch.setenv("HMG_LAUNCH_NONCE", "launch-1")
    old = dist.launch_nonce(tag, 2)
    monkeypatch.setenv("HMG_LAUNCH_NONCE", "launch-2")
    assert dist.launch_nonce(tag, 2) != old
    leftover(old)
    read_with(expect_wait=True)
    # torchrun's per-launch directory plays the same part
    monkeypatch.delenv("HMG_LAUNCH_NONCE")
    monkeypatch.setenv("TORCHELASTIC_ERROR_FILE", "/tmp/torchelastic_ab12/job_x1/attempt_0/1/error.json")
    a = dist.launch_nonce(tag, 2)
    monkeypatch.setenv("TORCHELASTIC_ERROR_FILE", "/tmp/torchelastic_ab12/job_x1/attempt_0/0/error.json")
    assert dist.launch_nonce(tag, 2) == a                       # common to the ranks of an attempt
    monkeypatch.setenv("TORCHELASTIC_ERROR_FILE", "/tmp/torchelastic_zz99/job_q7/attempt_0/0/error.json")
    assert dist.launch_nonce(tag, 2) != a                       # another launch
    # (b) no salt at all: same name, same nonce - the leftover is older than this process's parent
    monkeypatch.delenv("TORCHELASTIC_ERROR_FILE")
    leftover(dist.launch_nonce(tag, 2))
    os.utime(path, (time.time() - 7 * 86400,) * 2)
    assert dist._stale(path)
    read_with(expect_wait=True)
