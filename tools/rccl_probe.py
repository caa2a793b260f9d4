"""Time the RCCL bring-up of a 1-rank communicator (diagnostic)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t0 = time.time()
from hmvec_amd import _native as nat
from hmvec_amd.dist import RcclComm
ctx = nat.Context(0)
t1 = time.time()
comm = RcclComm(ctx, 0, 1, f"probe_{os.getpid()}", force_init=True)
t2 = time.time()
comm.barrier()
t3 = time.time()
comm.close()
print(f"ctx {t1-t0:.2f}s  comm_init {t2-t1:.2f}s  barrier {t3-t2:.3f}s")
