"""Thin slabs: is a rank's slab of 4 redshifts faster as TWO independent captured passes of 2 redshifts each, replayed on
two contexts (two streams) so that one pass's launch tails overlap the other's ramps?  ms per whole slab, graph replay."""
import sys, time, numpy as np
sys.path.insert(0, ".")
import hmvec_amd as hm
from hmvec_amd import _native as nat
PAIRS = [("nfw", "nfw"), ("electron", "electron"), ("g", "g"), ("nfw", "electron"), ("g", "nfw"), ("g", "electron")]
ms = np.geomspace(2e10, 1e17, 512); ks = np.geomspace(1e-4, 100, 4096)
zall = np.linspace(0.01, 3.0, 32)[:4]

def make(ctx, zs):
    h = hm.HaloModel(zs, ks, ms=ms, accuracy="low", engine="analytic", ctx=ctx)
    thr = 10 ** 10.5 + zs * 0.0
    o1 = [ctx.empty((zs.size, ks.size)) for _ in PAIRS]; o2 = [ctx.empty((zs.size, ks.size)) for _ in PAIRS]
    def one():
        h.init_mass_function(ms); h.add_nfw_profile("nfw", ignore_existing=True)
        h.add_battaglia_profile("electron", family="AGN", xmax=20, nxs=5000, ignore_existing=True)
        h.add_hod("g", mthresh=thr, ignore_existing=True)
        h.power_device_batch(PAIRS, outs1=o1, outs2=o2)
    one(); one(); ctx.sync()
    return h, ctx.capture(one)

def timeit(items, n=400):
    for _ in range(100):
        for c, g in items: c.replay(g)
    for c, _ in items: c.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        for c, g in items: c.replay(g)
    for c, _ in items: c.sync()
    return (time.perf_counter() - t0) / n * 1e3

c0 = nat.Context(0); h0, g0 = make(c0, zall)
print(f"one pass of 4 redshifts:            {timeit([(c0, g0)]):.4f} ms per slab")
ca, cb = nat.Context(0), nat.Context(0)
ha, ga = make(ca, zall[:2]); hb, gb = make(cb, zall[2:])
print(f"two passes of 2 redshifts, 2 streams: {timeit([(ca, ga), (cb, gb)]):.4f} ms per slab")
print(f"one pass of 2 redshifts alone:        {timeit([(ca, ga)]):.4f} ms")
