"""Which arrays differ between the grouped and the one-launch-per-stage path when the HOD rides with the front?"""
import os, sys
import numpy as np
sys.path.insert(0, ".")
import hmvec_amd as hm

nz, nm, nk = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
zs = np.linspace(0.01, 2.0, nz); ms = np.geomspace(2e10, 1e17, nm); ks = np.geomspace(1e-4, 100, nk)
thr = 10 ** 10.5 + zs * 0.0
os.environ["HMG_NO_GROUPS"] = "1"
e = hm.HaloModel(zs, ks, ms=ms, accuracy="low", engine="analytic"); e.add_hod("g", mthresh=thr)
E = {k: e.hods["g"][k] for k in ("Nc", "Ns", "NsNsm1", "NcNs", "ngal", "bg")}; E["nzm"] = e.nzm; E["bh"] = e.bh; E["sigma2"] = e.sigma2
os.environ["HMG_NO_GROUPS"] = "0"
g = hm.HaloModel(zs, ks, ms=ms, accuracy="low", engine="analytic"); g.add_hod("g", mthresh=thr)
print("queued", [s[0] for s in g._stages])
G = {k: g.hods["g"][k] for k in ("Nc", "Ns", "NsNsm1", "NcNs", "ngal", "bg")}; G["nzm"] = g.nzm; G["bh"] = g.bh; G["sigma2"] = g.sigma2
for k in E:
    bad = np.argwhere(~(E[k] == G[k]))
    print(k, "equal" if bad.size == 0 else f"{len(bad)} differ, first {bad[:3].tolist()} e={E[k][tuple(bad[0])]} g={G[k][tuple(bad[0])]}")
