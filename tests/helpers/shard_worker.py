"""Worker for tests/test_shard_gloo.py: one rank of a world_size-2 CPU rehearsal of the z-slab
path.  The product's sharding logic (hmvec_amd.dist.slab_bounds / ShardedSpectra) runs
unchanged; the GPU engine is replaced by the CPU oracle and RCCL by torch.distributed/gloo."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

from hmvec_amd.dist import ShardedSpectra, slab_bounds  # noqa: E402
from hmvec_amd.params import battaglia_defaults, default_params  # noqa: E402
from oracle import hmref  # noqa: E402


class HostArray:
    """numpy-backed stand-in for _native.DeviceArray."""

    def __init__(self, shape):
        self.a = np.zeros(shape)
        self.shape, self.size, self.ptr = self.a.shape, self.a.size, None

    def numpy(self):
        return self.a.copy()


class HostCtx:
    def empty(self, shape):
        return HostArray(shape)

    # stream/event plumbing of the device context: nothing to order on the host
    def wait(self, slot):
        pass

    def record(self, slot):
        pass

    def lane(self, i):
        pass


class OracleEngine:
    """Looks like the slice of HaloModel that ShardedSpectra touches."""

    def __init__(self, ref):
        self.ref, self.zs, self.ks = ref, ref.zs, ref.ks
        self._c = HostCtx()

    def _ctx(self):
        return self._c

    def power_device_batch(self, pairs, outs1, outs2):
        for (a, b), o1, o2 in zip(pairs, outs1, outs2):
            o1.a[...] = self.ref.get_power_1halo(a, b)
            o2.a[...] = self.ref.get_power_2halo(a, b)
        return outs1, outs2


class GlooComm:
    def __init__(self):
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def allgather_rows(self, sends, recvs, counts=None):
        for s, r in zip(sends, recvs):
            if counts is None:
                out = torch.empty(r.shape, dtype=torch.float64)
                dist.all_gather_into_tensor(out, torch.from_numpy(s.a))
                r.a[...] = out.numpy()
                continue
            # unequal slabs: one broadcast per rank into its prefix-sum offset - what hmg_comm_allgatherv_multi
            # does with ncclBroadcast
            flat, off = r.a.reshape(-1), 0
            for root, cnt in enumerate(counts):
                t = torch.from_numpy(s.a.reshape(-1).copy()) if root == self.rank else torch.empty(cnt, dtype=torch.float64)
                assert t.numel() == cnt
                dist.broadcast(t, src=root)
                flat[off:off + cnt] = t.numpy()
                off += cnt

    def barrier(self):
        dist.barrier()


def build_ref(zs, ms, ks, ngal_full=None):
    import hmvec_amd as hm
    p = dict(default_params)
    p["sigma2_numks"] = 2000
    cos = hm.Cosmology(p, accuracy="low", engine="analytic")
    ksig = np.geomspace(p["sigma2_kmin"], p["sigma2_kmax"], p["sigma2_numks"])
    ci = hmref.CosmoInputs(h=cos.h, omm0=cos.omm0, ombh2=p["ombh2"],
                           rho_crit_0=float(cos.rho_critical_z(0.0)), rho_crit_zs=cos.rho_critical_z(zs),
                           Pzk=cos.P_lin_approx(ks, zs), sPzk=cos.P_lin_approx(ksig, zs), ks_sigma2=ksig,
                           h_of_z_zs=cos.h_of_z(zs))
    o = hmref.RefHaloModel(ci, zs, ks, ms, p)
    o.add_battaglia_profile("electron", "AGN", p["battaglia_gas_gamma"], battaglia_defaults["AGN"], 300, 20)
    o.add_hod("g", mthresh=10 ** 10.5 + zs * 0.0)
    return o


def main():
    out_dir = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    zs = np.linspace(0.05, 2.5, int(sys.argv[2]) if len(sys.argv) > 2 else 6)
    ms = np.geomspace(1e11, 1e16, 20)
    ks = np.geomspace(1e-3, 10, 24)
    pairs = [("nfw", "nfw"), ("g", "electron"), ("g", "g")]
    lo, hi = slab_bounds(zs.size, world, rank)
    eng = OracleEngine(build_ref(zs[lo:hi], ms, ks))
    spec = ShardedSpectra(eng, GlooComm(), zs.size, pairs)
    spec.run()
    res = spec.results()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"),
             **{f"{a}_{b}_{i}": arr for (a, b), pr in res.items() for i, arr in enumerate(pr)})
    if rank == 0:
        full = build_ref(zs, ms, ks)
        np.savez(os.path.join(out_dir, "full.npz"),
                 **{f"{a}_{b}_{i}": arr for a, b in pairs
                    for i, arr in enumerate((full.get_power_1halo(a, b), full.get_power_2halo(a, b)))})
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
