"""Resource table of the kernels a pass launches: registers, scratch reservation, occupancy and
- the part the compiler's summary does not tell - how many scratch INSTRUCTIONS and SGPR-spill lane moves the ISA
holds and where.  Reads hmvec_amd/csrc/hmgrid.resources.txt and hmgrid.s (`make -C hmvec_amd/csrc asm`).
Usage: python tools/kernel_resources.py > profiles/rNN/kernel_resources.txt"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(HERE, "hmvec_amd", "csrc")
sys.path.insert(0, HERE)
KERNELS = [   # (label, regex on the mangled name)
    ("front_group_kernel<2>  (nz > 16)", r"front_group_kernelILi2E"),
    ("front_group_kernel<1>  (nz <= 16)", r"front_group_kernelILi1E"),
    ("tensor_group_kernel<2,3,2500>  (nxs = 5000: chain | profile rows | NFW rows, the launch of a pass)", r"tensor_group_kernelILi2ELi3ELi2500E"),
    ("rows_group_kernel  (HMG_NO_TENSOR_GROUP=1, and passes without a profile transform)", r"17rows_group_kernelE"),
    ("profile_group_kernel<2,3,2500>  (nxs = 5000; a second profile of a pass, HMG_NO_TENSOR_GROUP=1)", r"profile_group_kernelILi2ELi3ELi2500E"),
    ("power_batch_kernel<2,3,2,false,593>  (Config 3, full grid)", r"power_batch_kernelILi2ELi3ELi2ELb0ELj593E"),
    ("power_batch_kernel<2,3,1,true,593>  (thin slab)", r"power_batch_kernelILi2ELi3ELi1ELb1ELj593E"),
    ("profile_pruned_kernel<512,1000>  (nxs = 30000)", r"profile_pruned_kernelILi512ELi1000E"),
    ("profile_pruned_kernel<512,1250>  (nxs = 40000)", r"profile_pruned_kernelILi512ELi1250E"),
    ("profile_band_kernel<512,1000,1>  (tSZ: nxs = 30000, xmax = 2, <= 255 modes)", r"profile_band_kernelILi512ELi1000ELi1E"),
    ("profile_band_kernel<512,1000,2>  (the same, up to 499 modes)", r"profile_band_kernelILi512ELi1000ELi2E"),
    ("profile_group_kernel<1,2,1500>  (nxs = 3000, compile-time plan, round 5)", r"profile_group_kernelILi1ELi2ELi1500E"),
    ("profile_group_kernel<3,3,3000>  (nxs = 6000 as one row, round 5)", r"profile_group_kernelILi3ELi3ELi3000E"),
    ("profile_group_kernel<2,3,0>  (run-time plan)", r"profile_group_kernelILi2ELi3ELi0E"),
    ("profile_table_kernel<512,2,3,2500>  (user callable, nxs = 5000, round 5)", r"profile_table_kernelILi512ELi2ELi3ELi2500E"),
    ("profile_table_kernel<512,2,4,0>  (user callable, run-time plan)", r"profile_table_kernelILi512ELi2ELi4ELi0E"),
    ("stand-alone: profile_fused_kernel<512,2,3,2500>", r"profile_fused_kernelILi512ELi2ELi3ELi2500E"),
    ("stand-alone: nfw_kernel", r"10nfw_kernelE"),
]


def main():
    if "--no-build" not in sys.argv:
        subprocess.run(["make", "-j2", "-B", "-C", CSRC, "asm"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    res = open(os.path.join(CSRC, "hmgrid.resources.txt")).read() + open(os.path.join(CSRC, "longgrid.resources.txt")).read()
    asm = open(os.path.join(CSRC, "hmgrid.s")).read() + open(os.path.join(CSRC, "longgrid.s")).read()
    from hmvec_amd._native import kernel_source_sha16
    print(f"# kernel sources {kernel_source_sha16()}  (hipcc -O3 --offload-arch=gfx950 -mllvm -disable-machine-licm; make asm)")
    print("# columns: SGPRs VGPRs | reserved scratch B/lane | occupancy waves/SIMD | SGPR spills, VGPR spills (compiler summary) |")
    print("#          scratch_* instructions in the ISA | v_writelane/v_readlane (SGPR spills live in VGPR lanes, not in memory)")
    for label, pat in KERNELS:
        m = re.search(r"Function Name: (_ZN3hmg\d*" + pat + r"\S*)(.*?)LDS Size", res, re.S)
        if not m:
            print(f"{label}: not found")
            continue
        name, block = m.group(1), m.group(2)
        g = lambda key: re.search(key + r": (\d+)", block).group(1)      # noqa: E731
        body = asm[asm.index("\n" + name + ":"):]
        body = body[:body.index(".Lfunc_end")]
        lines = body.splitlines()
        scr = [i for i, l in enumerate(lines) if re.match(r"\s*scratch_", l)]
        wl = sum(1 for l in lines if re.match(r"\s*v_writelane", l))
        rl = sum(1 for l in lines if re.match(r"\s*v_readlane", l))
        scratch, occ = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]")
        where = (", at lines " + ",".join(map(str, scr))) if scr else ""
        print(f"{label}\n    SGPRs {g('TotalSGPRs')}  VGPRs {g('VGPRs')} | scratch {scratch} B | occupancy {occ} | spills: SGPR "
              f"{g('SGPRs Spill')}, VGPR {g('VGPRs Spill')} | scratch instructions {len(scr)} (of {len(lines)} lines{where}) | "
              f"lane moves {wl}/{rl}")


if __name__ == "__main__":
    main()
