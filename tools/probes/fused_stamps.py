"""Phase durations inside the one-row profile kernel (nxs = 5000, Config 3) from in-kernel clock stamps, taken in a
steady stream of passes (library built with -DHMG_FR_STAMP: make -C hmvec_amd/csrc OUT=../libhmgrid_fstamp.so EXTRA=-DHMG_FR_STAMP)."""
import ctypes as C, sys, numpy as np
sys.path.insert(0, ".")
import hmvec_amd as hm
from hmvec_amd import _native as nat
zs = np.linspace(0.01, 3.0, 32); ms = np.geomspace(2e10, 1e17, 512); ks = np.geomspace(1e-4, 100, 4096)
h = hm.HaloModel(zs, ks, ms=ms, accuracy="low", engine="analytic")
thr = 10 ** 10.5 + zs * 0.0
six = [("nfw", "nfw"), ("electron", "electron"), ("g", "g"), ("nfw", "electron"), ("g", "nfw"), ("g", "electron")]
for _ in range(200):                   # a stream of whole passes: the clock and caches of the steady state
    h.init_mass_function(ms); h.add_nfw_profile("nfw", ignore_existing=True)
    h.add_battaglia_profile("electron", family="AGN", xmax=20, nxs=5000, ignore_existing=True)
    h.add_hod("g", mthresh=thr, ignore_existing=True)
    h.power_device_batch(six)
h._ctx().sync()
lib = nat.load()
n = 4096 * 16
buf = (C.c_longlong * n)()
lib.hmg_debug_stamps_fused.argtypes = [C.POINTER(C.c_longlong), C.c_int]
assert lib.hmg_debug_stamps_fused(buf, n) == 0
S = np.frombuffer(buf, dtype=np.int64).reshape(4096, 16)
S = S[S[:, 0] != 0]
wall = (S[:, 15] - S[:, 14]) * 10.0
cyc = S[:, 7] - S[:, 0]
print(f"{S.shape[0]} sampled rows; shader clock {np.median(cyc / wall):.3f} GHz; kernel spans {(S[:, 15].max() - S[:, 14].min()) * 1e-2:.1f} us")
def stat(x): return f"median {np.median(x):8.0f}  mean {np.mean(x):8.0f}"
names = ["scalars + integrand + norm", "(pruned first pass) second pass", "pass Ns=20", "pass Ns=100", "pass Ns=500 (pruned to the band)", "unpack", "interpolation + stores"]
print("whole row                          ", stat(cyc))
for i, nme in enumerate(names):
    print(f"{nme:35s}", stat(S[:, i + 1] - S[:, i]))
print("inside the first phase (thread 0 = wavefront 0):")
print("   row parameters + pruning decisions ", stat(S[:, 8] - S[:, 0]))
print("   integrand of its sample pair       ", stat(S[:, 9] - S[:, 8]))
print("   wavefront sum + barrier            ", stat(S[:, 10] - S[:, 9]))
print("   partial sums -> scale (wavefront 0)", stat(S[:, 1] - S[:, 10]))
