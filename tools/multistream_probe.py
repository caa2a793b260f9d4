"""Probe: z-slabs of ONE grid on concurrent HIP streams of one GPU (no cross-stream dependencies): does the
HBM-bound mass-integral kernel of one slab overlap the VALU-bound profile kernels of another?
Usage: python tools/multistream_probe.py [nslabs ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hmvec_amd as hm
from hmvec_amd import _native as nat

PAIRS = [("nfw", "nfw"), ("electron", "electron"), ("g", "g"), ("nfw", "electron"), ("g", "nfw"), ("g", "electron")]
zs = np.linspace(0.01, 3.0, 32); ms = np.geomspace(2e10, 1e17, 512); ks = np.geomspace(1e-4, 100, 4096)


def build(ctx, z):
    h = hm.HaloModel(z, ks, ms=ms, accuracy="low", engine="analytic", ctx=ctx)
    h.add_battaglia_profile("electron", family="AGN", xmax=20, nxs=5000)
    mthr = 10 ** 10.5 + z * 0.0
    h.add_hod("g", mthresh=mthr)
    blk = h.spectra_block(PAIRS)

    def step():
        h.init_mass_function(ms)
        h.add_nfw_profile("nfw", ignore_existing=True)
        h.add_battaglia_profile("electron", family="AGN", xmax=20, nxs=5000, ignore_existing=True)
        h.add_hod("g", mthresh=mthr, ignore_existing=True)
        blk.compute()
    step(); step()
    ctx.sync()
    return h, blk, ctx.capture(step)


for ns in [int(a) for a in sys.argv[1:]] or [1, 2, 4]:
    per = zs.size // ns
    ctxs = [nat.Context(0) for _ in range(ns)]
    models = [build(c, zs[i * per:(i + 1) * per]) for i, c in enumerate(ctxs)]
    for _ in range(80):
        for c, (_, _, g) in zip(ctxs, models):
            c.replay(g)
    for c in ctxs:
        c.sync()
    K = 100
    t0 = time.perf_counter()
    for _ in range(K):
        for c, (_, _, g) in zip(ctxs, models):
            c.replay(g)
    for c in ctxs:
        c.sync()
    dt = (time.perf_counter() - t0) / K
    print(f"slabs on {ns} stream(s): {dt*1e3:.4f} ms per full-grid step")
    del models
    for c in ctxs:
        c.close()
