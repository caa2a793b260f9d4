"""One rank of the two-process z-slab run (tests/test_gpu_comm.py): builds the model of its slab, runs the
sharded spectra through the named transport and saves what the gather left in its full-grid buffers.
argv: rank world tag directory out.npz [transport [nz]]
transport "files" (default): every rank on device 0, exchange through tests/helpers/rehearsal_comm.py (a one-GPU
box); "rccl": rank r on device r, the product's RcclComm (needs as many devices as ranks)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hmvec_amd as hm                                   # noqa: E402
from hmvec_amd import _native as nat                     # noqa: E402
from hmvec_amd.dist import RcclComm, ShardedSpectra, slab_bounds   # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from rehearsal_comm import HostRehearsalComm             # noqa: E402

rank, world, tag, directory, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
transport = sys.argv[6] if len(sys.argv) > 6 else "files"
zs = np.linspace(0.1, 2.6, int(sys.argv[7]) if len(sys.argv) > 7 else 8)
ms = np.geomspace(2e10, 1e16, 96)
ks = np.geomspace(1e-3, 50, 384)
PAIRS = [("nfw", "nfw"), ("electron", "electron"), ("g", "g"), ("g", "electron"), ("nfw", "electron")]
lo, hi = slab_bounds(zs.size, world, rank)
if transport == "rccl":
    os.environ["HMG_RDZV_DIR"] = directory
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ctx = nat.Context(rank)
    comm = RcclComm(ctx, rank, world, tag)
else:
    ctx = nat.Context(0)                                 # every rank on device 0: a one-GPU box
    comm = HostRehearsalComm(ctx, rank, world, tag, directory=directory)
h = hm.HaloModel(zs[lo:hi], ks, ms=ms, accuracy="low", engine="analytic", ctx=ctx)
h.add_battaglia_profile("electron", family="AGN", xmax=20, nxs=1000)
h.add_hod("g", mthresh=10 ** (10.3 + 0.1 * zs[lo:hi]))
spec = ShardedSpectra(h, comm, zs.size, PAIRS)
for _ in range(3):                                       # several passes: the ordering events of the gather path
    spec.run()
comm.barrier()
res = spec.results()
np.savez(out, **{f"{a}|{b}|{i}": res[(a, b)][i] for (a, b) in PAIRS for i in (0, 1)})
comm.close()
ctx.close()
