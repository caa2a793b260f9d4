"""Print the measured GPU-vs-reference differences for every golden case (evidence for the
parity table in DESIGN.md).  Runs on the GPU box; reads only tests/golden."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
from conftest import load_golden
from test_gpu_parity import build_gpu, add_all

def rel(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))

for case in ("case_a", "case_b", "case_c"):
    g = load_golden(case)
    h = build_gpu(g); add_all(h, g)
    row = {"sigma2": rel(h.sigma2, g["sigma2"]), "nzm": rel(h.nzm, g["nzm"]), "bh": rel(h.bh, g["bh"]),
           "uk_nfw(abs)": float(np.max(np.abs(h.uk_profiles["nfw"] - g["uk_nfw"]))),
           "uk_e(abs)": float(np.max(np.abs(h.uk_profiles["electron"] - g["uk_electron"]))),
           "m200c": rel(h._m200c()[0].numpy(), g["m200c"])}
    for k in ("Nc", "Ns", "ngal", "bg"):
        row["hod_" + k] = rel(np.asarray(h.hods["g"][k]) + 1e-300, g["hod_" + k] + 1e-300)
    names = ["nfw", "electron", "g"] + (["y"] if g["meta"]["pres"] else [])
    worst = 0.0
    for i, a in enumerate(names):
        for b in names[i:]:
            for lab, P in (("P1h", h.get_power_1halo(a, b)), ("P2h", h.get_power_2halo(a, b))):
                R = g[f"{lab}_{a}_{b}"]
                tol = 1e-8 * np.abs(R) + 1e-12 * np.max(np.abs(R), axis=-1, keepdims=True)
                worst = max(worst, float(np.max(np.abs(P - R) / tol)))
    row["P worst |d|/tol"] = worst
    print(case, {k: float(f"{v:.2e}") for k, v in row.items()})
