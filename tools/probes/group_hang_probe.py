"""Step-by-step probe of the grouped launches on a grid where a test run stalled (run under `timeout`)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import hmvec_amd as hm
from hmvec_amd import _native as nat

nz, nm, nk = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
mode = sys.argv[4]
zs = np.linspace(0.01, 2.0, nz); ms = np.geomspace(2e10, 1e17, nm); ks = np.geomspace(1e-4, 100, nk)
t0 = time.time()
def say(*a):
    print(f"[{time.time()-t0:6.2f}]", *a, flush=True)
h = hm.HaloModel(zs, ks, ms=ms, accuracy="low", engine="analytic")
say("ctor queued", [s[0] for s in h._stages])
ctx = h._ctx()
if mode == "plain":
    ctx.sync(); say("front+rows done")
    h.add_hod("g", mthresh=10 ** 10.5 + zs * 0.0); say("hod queued")
    say("ngal", h.hods["g"]["ngal"][:3])
elif mode == "occ":
    h.add_hod("g", mthresh=10 ** 10.5 + zs * 0.0); say("hod queued", [s[0] for s in h._stages])
    ctx.sync(); say("front(occ)+rows(sums) done")
    say("ngal", h.hods["g"]["ngal"][:3])
elif mode == "bisect":
    h.add_hod("g", ngal=1e-4 + zs * 0.0); say("bisected")
    say("ngal", h.hods["g"]["ngal"][:3])
say("ok")
