#!/usr/bin/env python3
"""Sweep-throughput experiment: in a STREAM of passes the HBM-bound mass integrals of pass
i and the VALU-bound profile kernels of pass i+1 use complementary resources.  Two models on two contexts of the
same device (each context has its own streams and scratch arenas, so the passes share nothing but the GPU)
alternate; each pass issues front / rows group / profile group on its lane 0 and its mass integrals on lane 2
behind an event, and a model's next pass waits for its previous mass integrals.  Reports the time per pass of
the alternating stream next to the back-to-back time per pass of one model.
Usage: python tools/stream_passes.py [--nz 32] [--passes 200]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hmvec_amd as hm
from hmvec_amd import _native as nat

ap = argparse.ArgumentParser()
ap.add_argument("--nz", type=int, default=32)
ap.add_argument("--passes", type=int, default=200)
ap.add_argument("--power-lane", type=int, default=2)
args = ap.parse_args()
PAIRS = [("nfw", "nfw"), ("electron", "electron"), ("g", "g"), ("nfw", "electron"), ("g", "nfw"), ("g", "electron")]
zs = np.linspace(0.01, 3.0, args.nz); ms = np.geomspace(2e10, 1e17, 512); ks = np.geomspace(1e-4, 100, 4096)
thr = 10 ** 10.5 + zs * 0.0
EV_PROFILES, EV_POWER = 20, 21


def make(ctx):
    h = hm.HaloModel(zs, ks, ms=ms, accuracy="low", engine="analytic", ctx=ctx)
    h.add_battaglia_profile("electron", family="AGN", xmax=20, nxs=5000)
    h.add_hod("g", mthresh=thr)
    blk = h.spectra_block(PAIRS)
    blk.compute()
    ctx.sync()
    return h, blk


def one_pass(ctx, h, blk, overlap):
    if overlap:
        ctx.wait(EV_POWER)                     # this model's previous mass integrals still read its tensors
    h.init_mass_function(ms)
    h.add_nfw_profile("nfw", ignore_existing=True)
    h.add_battaglia_profile("electron", family="AGN", xmax=20, nxs=5000, ignore_existing=True)
    h.add_hod("g", mthresh=thr, ignore_existing=True)
    if not overlap:
        blk.compute()
        return
    # issue everything but the mass integrals on lane 0 (a read of nothing: the queue is flushed by the record)
    h._flush()
    ctx.record(EV_PROFILES)
    ctx.lane(args.power_lane)
    ctx.wait(EV_PROFILES)
    blk.compute()
    ctx.record(EV_POWER)
    ctx.lane(0)


ctxs = [nat.Context(0), nat.Context(0)]
models = [make(c) for c in ctxs]
ref = {p: (a.copy(), b.copy()) for p, (a, b) in models[0][1].fetch().items()}


def run(overlap, n):
    for c in ctxs:
        c.sync()
    t0 = time.perf_counter()
    for i in range(n):
        j = i % 2 if overlap else 0
        one_pass(ctxs[j], *models[j], overlap)
    for c in ctxs:
        c.sync()
    return (time.perf_counter() - t0) / n * 1e3


for _ in range(2):
    run(False, 30); run(True, 30)
serial = min(run(False, args.passes) for _ in range(3))
streamed = min(run(True, args.passes) for _ in range(3))
for j in (0, 1):
    got = models[j][1].fetch()
    for p in PAIRS:
        assert np.array_equal(got[p][0], ref[p][0]) and np.array_equal(got[p][1], ref[p][1]), (j, p)
print(f"nz={args.nz}: back-to-back {serial:.4f} ms/pass, alternating stream {streamed:.4f} ms/pass "
      f"({(1 - streamed / serial) * 100:+.1f} % time), results identical")
