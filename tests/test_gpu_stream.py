"""Result hand-over of a stream of passes (the reference returns host arrays, hmvec/hmvec.py:500-572): a
double-buffered SpectraBlock copies pass i to the host on the copy lane while pass i+1 computes.  Every pass's
host copy must hold exactly that pass's spectra - nothing torn, nothing overwritten early, nothing stale."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PAIRS = [("nfw", "nfw"), ("electron", "electron"), ("g", "g"), ("g", "electron")]


def test_streamed_result_blocks_hold_their_own_pass():
    import hmvec_amd as hm
    from hmvec_amd import _native as nat
    ctx = nat.Context(0)
    zs = np.linspace(0.1, 2.5, 6)
    ms = np.geomspace(2e10, 1e17, 96)
    ks = np.geomspace(1e-4, 100, 1024)
    h = hm.HaloModel(zs, ks, ms=ms, accuracy="low", engine="analytic", ctx=ctx)
    h.add_battaglia_profile("electron", family="AGN", xmax=20, nxs=1000)
    thresholds = [10.2, 10.5, 10.8, 11.1, 11.4, 10.35, 10.65]
    # reference: one pass at a time, synchronous fetch
    one = h.spectra_block(PAIRS)
    want = []
    for t in thresholds:
        h.add_hod("g", mthresh=10 ** t + zs * 0.0, ignore_existing=True)
        one.compute()
        want.append({p: (a.copy(), b.copy()) for p, (a, b) in one.fetch().items()})
    # streamed: pass i into block i % 2, copy on the copy lane, host reads pass i-1 while pass i is enqueued
    blk = h.spectra_block(PAIRS, nbuf=2)
    got = [None] * len(thresholds)
    for i, t in enumerate(thresholds):
        h.add_hod("g", mthresh=10 ** t + zs * 0.0, ignore_existing=True)
        blk.compute(i % 2)
        blk.fetch_async(i % 2)
        if i >= 1:
            got[i - 1] = {p: (a.copy(), b.copy()) for p, (a, b) in blk.wait((i - 1) % 2).items()}
    got[-1] = {p: (a.copy(), b.copy()) for p, (a, b) in blk.wait((len(thresholds) - 1) % 2).items()}
    ctx.sync()
    for i in range(len(thresholds)):
        for p in PAIRS:
            assert np.array_equal(got[i][p][0], want[i][p][0]), (i, p)
            assert np.array_equal(got[i][p][1], want[i][p][1]), (i, p)
    # passes differ (the galaxy spectra move by tens of per cent between thresholds): a stale block would show
    assert not np.array_equal(want[0][("g", "g")][0], want[1][("g", "g")][0])


def test_pinned_uploads_round_trip():
    from hmvec_amd import _native as nat
    ctx = nat.Context(0)
    rng = np.random.default_rng(3)
    a = rng.standard_normal((32, 10000))
    pin = nat.PinnedArray(ctx, a.shape)
    pin.array[...] = a
    d = ctx.empty(a.shape)
    ctx.copy_from_pinned(d, pin)
    ctx.record(20)
    ctx.event_synchronize(20)
    assert np.array_equal(d.numpy(), a)
