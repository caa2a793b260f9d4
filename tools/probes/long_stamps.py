"""Phase durations inside the long-grid row kernel from in-kernel clock stamps (library built with -DHMG_LG_STAMP:
tools/long_build.sh stamp "-DHMG_LG_STAMP"; HMG_LIB_PATH=hmvec_amd/libhmgrid_stamp.so python tools/probes/long_stamps.py [gas|nfw]).
Cycles of the shader clock, per sampled row (every 29th workgroup), chirp rows and decomposition rows apart."""
import ctypes as C, sys, numpy as np
sys.path.insert(0, ".")
import hmvec_amd as hm
from hmvec_amd import _native as nat
which = sys.argv[1] if len(sys.argv) > 1 else "gas"
zs = np.linspace(0.01, 3.0, 32); ms = np.geomspace(2e10, 1e17, 512); ks = np.geomspace(1e-4, 100, 4096)
h = hm.HaloModel(zs, ks, ms=ms, accuracy="low", engine="analytic")
for _ in range(3):
    if which == "gas":
        h.add_battaglia_profile("electron", family="AGN", xmax=50, nxs=30000, ignore_existing=True)
    else:
        h.add_nfw_profile("nfwnum", numeric=True, ignore_existing=True)
    h._ctx().sync()
lib = nat.load()
n = 4096 * 64
buf = (C.c_longlong * n)()
lib.hmg_debug_stamps.argtypes = [C.POINTER(C.c_longlong), C.c_int]
assert lib.hmg_debug_stamps(buf, n) == 0
S = np.frombuffer(buf, dtype=np.int64).reshape(4096, 64)
S = S[S[:, 0] != 0]
jn = S[:, 62]
chirp = S[:, 2] != 0
print(f"{which}: {S.shape[0]} sampled rows, {chirp.sum()} chirp rows, {(~chirp).sum()} decomposition rows (cycles of s_memtime)")
def stat(x): return f"median {np.median(x):9.0f}  mean {np.mean(x):9.0f}"
wall = (S[:, 61] - S[:, 60]) * 10.0           # ns (wall_clock64: 100 MHz)
cyc = S[:, 7] - S[:, 0]
ok = wall > 2000
print(f"shader clock during the rows: {np.median(cyc[ok] / wall[ok]):.3f} GHz (s_memtime ticks per ns of wall_clock64, median over rows); "
      f"first row starts {(S[:, 60].min()) * 1e-5:.3f} ms, last row ends {(S[:, 61].max() - S[:, 60].min()) * 1e-5:.3f} ms after it")
tot = S[:, 7] - S[:, 0]
print("whole row            all   ", stat(tot))
for name, sel in (("chirp", chirp), ("decomposition", ~chirp)):
    if sel.sum() == 0: continue
    T = S[sel]
    print(f"-- {name} rows: jn median {np.median(T[:, 62]):.0f}")
    print("   whole row                 ", stat(T[:, 7] - T[:, 0]))
    print("   scalars+integrand+norm    ", stat(T[:, 1] - T[:, 0]))
    if name == "chirp":
        print("   first pass -> barrier     ", stat(T[:, 2] - T[:, 1]))
        print("   forward transform passes  ", stat(T[:, 3] - T[:, 2]))
        print("   window product + pass 0   ", stat(T[:, 4] - T[:, 3]))
        print("   second transform passes   ", stat(T[:, 5] - T[:, 4]))
        print("   unpack -> barrier         ", stat(T[:, 6] - T[:, 5]))
    else:
        g0 = T[:, 8:8 + 3 * 9].reshape(T.shape[0], 9, 3)
        ng = (g0[:, :, 0] != 0).sum(axis=1)
        print("   groups per row            ", stat(ng))
        first = np.where(g0[:, 1:, 0] != 0, g0[:, 1:, 0] - g0[:, :-1, 2], 0)
        print("   first pass+barrier (g>=1) ", stat(first[first != 0]))
        pp = g0[:, :, 1] - g0[:, :, 0]; print("   passes 1..4 of a group    ", stat(pp[g0[:, :, 0] != 0]))
        up = g0[:, :, 2] - g0[:, :, 1]; print("   unpack + barrier          ", stat(up[g0[:, :, 0] != 0]))
        last = np.array([g0[i, ng[i] - 1, 2] for i in range(T.shape[0])])
        print("   group loop total          ", stat(last - T[:, 1]))
        print("   fence + barrier           ", stat(T[:, 6] - last))
    print("   interpolation + stores    ", stat(T[:, 7] - T[:, 6]))
