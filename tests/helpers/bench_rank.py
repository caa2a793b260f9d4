"""One rank of bench.py's N > 1 path on a one-GPU box (tests/test_gpu_comm.py): bench.main() unchanged, with the product's
RcclComm replaced by the file transport of rehearsal_comm.py (RCCL refuses two ranks on one device) and every rank on
device 0.  RANK / WORLD_SIZE / HMG_LAUNCH_TAG / HMG_REHEARSAL_DIR come from the test; argv = bench.py's flags."""
import os
import sys

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
sys.path.insert(0, here)
import hmvec_amd.dist as dist                            # noqa: E402
from rehearsal_comm import HostRehearsalComm             # noqa: E402

dist.RcclComm = HostRehearsalComm
os.environ["LOCAL_RANK"] = "0"
import bench                                             # noqa: E402

sys.argv = ["bench.py"] + sys.argv[1:]
bench.main()
