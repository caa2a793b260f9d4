"""add_battaglia_pres_profile(xmax=2, nxs=30000) (the reference's tSZ notebook) on the Config-3 grid: narrow-band route
against rocFFT."""
import sys, numpy as np
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import shape_sweep as ss
zs = np.linspace(0.01, 3.0, 32); ms = np.geomspace(2e10, 1e17, 512); ks = np.geomspace(1e-4, 100, 4096)
names = ["nfw", "electron", "g", "y"]
ten = [(a, b) for i, a in enumerate(names) for b in names[i:]]
ss.run_case("pressure nxs=30000 xmax=2: narrow-band route ('profile stage' = the gas profile; see pass)", zs, ms, ks, (5000, 20), ten, pressure=(30000, 2))
ss.run_case("pressure nxs=30000 xmax=2: rocFFT route (HMG_BAND_FFT=0)", zs, ms, ks, (5000, 20), ten, pressure=(30000, 2), env={"HMG_BAND_FFT": "0"}, reps=5)
ss.run_case("pressure nxs=5000 xmax=5 (for scale)", zs, ms, ks, (5000, 20), ten, pressure=(5000, 5))
