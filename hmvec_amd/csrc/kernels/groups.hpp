// grouped launches: independent stages of a pass as block ranges of one grid.
// Part of the ONE translation unit hmgrid.hip (included there in this order; not a stand-alone header).
#pragma once

namespace hmg {

// ---------------------------------------------------------------- grouped launches
// Stages of a pass that do not depend on each other share ONE launch as disjoint block ranges of one grid:
// a kernel boundary costs ~2 us and on a thin z-slab (the rank of an 8-GPU job holds 4 redshifts) every
// per-(z,m) launch is pure latency - sigma^2 17 us, HOD 9, coefficient rows 6, row parameters 5 against
// 90 us for the three chip-filling kernels.  Streams do not help on this runtime (a cross-stream
// dependency costs more than it hides); block ranges do: the short, latency-bound
// workgroups come first in the grid, are dispatched first and finish while the long ones still fill the chip.
//   front   (64 threads):  halo stage points | sigma^2 contraction blocks           - both need inputs only
//   rows    (256 threads): per-z chain | Battaglia row parameters | analytic NFW rows - need the front
//   profile (512 threads): per-z chain | fused radial-profile rows                   - needs the rows group
// The per-z CHAIN is what the mass integrals wait for besides the tensors: second stage of sigma^2 + n, b
// -> HOD -> coefficient rows of the batched mass integrals, one workgroup per redshift, each link optional.
// Every role runs the device function of its stand-alone kernel, so grouped and separate launches give
// the same bits (tests/test_gpu_groups.py).
struct RowsArgs {
    int n;            // nz*nm, 0: no such role in this launch
    int kind, nm;
    const double *m200, *r200, *rvir, *zs, *rhoc, *hz;
    RowFit F;
    double gamma, alpha_const, pref, post_pref;
    RowOut O;
};
struct SigmaFrontArgs {
    int nz, nzp, nm, nq, gx, nseg;
    const double *PT, *kq, *wq, *R;
    double tswitch;
    double* partial;
};
template <int ZB>
__global__ __launch_bounds__(64, HMG_SIG_OCC) void front_group_kernel(SigmaFrontArgs G, HaloStageArgs H, int nhalo,
                                                                     HodRowArgs O, int nocc, RowsArgs Rw) {
    int b = blockIdx.x;
    if (b < nocc) {               // HOD occupations: the longest dependent chain of the launch, so first in the grid
        const int idx = b * 64 + threadIdx.x;
        if (idx < G.nz * O.nm) hod_occ_point(O, idx / O.nm, idx - (idx / O.nm) * O.nm);
        return;
    }
    b -= nocc;
    if (b < nhalo) {
        const int idx = b * 64 + threadIdx.x;
        if (idx < H.nz * H.nm) {
            // the Battaglia row parameters need only what this thread has just computed (M_200c, R_200c, r_vir):
            // the same thread goes on to them, from the same values the stand-alone launch would load
            double rv, m2, r2;
            halo_stage_point(H, idx, &rv, &m2, &r2);
            if (Rw.n) {
                const int z = idx / Rw.nm;
                rowparams_body(Rw.kind, idx, m2, r2, rv, 1.0 + Rw.zs[z], Rw.rhoc[z], Rw.hz ? Rw.hz[z] : 1.0, Rw.F,
                               Rw.gamma, Rw.alpha_const, Rw.pref, Rw.post_pref, Rw.O);
            }
        }
        return;
    }
    b -= nhalo;
    const int r = b / G.gx, bx = b - r * G.gx;
    const int bz = r / G.nseg, seg = r - bz * G.nseg;
    sigma2_mfma_block<ZB>(bx, seg, bz, G.nz, G.nzp, G.nm, G.nq, G.PT, G.kq, G.wq, G.R, G.tswitch, G.partial);
}

struct ChainArgs {
    int has_hod, has_prep;
    HodRowArgs H;
    PrepArgs PA;
    int has_mf, mf_pad;        // first link sigma^2 -> n, b (all masses of the redshift, massfn_row)
    SigmaMassFnArgs S;
};
// doubles of LDS a chain workgroup of nt threads needs (the links use it one after the other)
static inline size_t chain_lds_doubles(int nm, bool has_mf = false, int nt = 512) {
    const size_t hod = 2 * (size_t)((nm + 63) / 64), mf = has_mf ? (size_t)nt : 0;
    return hod > mf ? hod : mf;
}
// The chain is kept LIGHT on purpose - the n_gal, b_g sums of an HOD and the compact coefficient rows: loads, a
// few divisions, wavefront sums - so that it fits the register budget of the launch it rides in without a
// spill.  Everything heavy of the HOD (its occupation numbers: SHMR inversion, erf, powers) is in the front
// launch or, when there is no front to ride with, in hmg_hod's own kernel.
#define HMG_KERNARG __attribute__((address_space(4)))
template <class T>
__device__ __forceinline__ const T HMG_KERNARG* uniform_kernarg(const T HMG_KERNARG* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const T HMG_KERNARG*)(((unsigned long long)hi << 32) | lo);
}
template <class T>
__device__ __forceinline__ T kernarg_load(const T HMG_KERNARG* p) {      // a by-value copy out of the segment,
    static_assert(sizeof(T) % 8 == 0, "pad the argument block to 8 bytes");   // word by word through the constant
    union { T v; unsigned long long w[sizeof(T) / 8]; } u;                   // address space (-> scalar loads)
    const unsigned long long HMG_KERNARG* q = (const unsigned long long HMG_KERNARG*)p;
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 8; ++i) u.w[i] = q[i];
    return u.v;
}
// ROLES OF A GROUPED LAUNCH (chain_row, rows_block, massfn_block, nfw_rows, profile_fused_row, the front's roles) ARE
// __forceinline__ INTO THEIR __global__ KERNEL AND TAKE KERNEL PARAMETERS BY VALUE.  Round 3 tried a role as a
// `noinline` function with its arguments behind a pointer: hipcc 7.2 lost the thread index on one path of it and part
// of a workgroup skipped a barrier (a launch that never finished); read through __builtin_amdgcn_kernarg_segment_ptr()
// INSIDE a called function the argument block sits at address 0 (a memory fault).  DESIGN.md section 3, "What stalled
// and what aborted in round 3".
// MF: the launch is compiled with the sigma^2 -> n, b link (its erfc / exp / pow need ~30 registers more than the
// other links: only the tensor group carries it)
template <int NT, bool MF = false>
__device__ __forceinline__ void chain_row(const ChainArgs& C, int z, double* lds) {
    if (MF && C.has_mf) {
        massfn_row<NT>(C.S, z, lds);
        __threadfence_block();                 // n, b of this redshift are read back below by other threads of the workgroup
        __syncthreads();
    }
    if (C.has_hod) {
        hod_sums_row(C.H, z, NT, lds);
        __syncthreads();
    }
    if (C.has_prep)
        for (int blk = threadIdx.x >> 6; blk < C.PA.nblk; blk += NT / 64) batch_prep_tile_compact(C.PA, z, blk);
}

__device__ __forceinline__ void rows_block(const RowsArgs& Rw, int b) {
    const int idx = b * 256 + threadIdx.x;
    if (idx < Rw.n) {
        const int z = idx / Rw.nm;
        rowparams_body(Rw.kind, idx, Rw.m200[idx], Rw.r200[idx], Rw.rvir[idx], 1.0 + Rw.zs[z], Rw.rhoc[z],
                       Rw.hz ? Rw.hz[z] : 1.0, Rw.F, Rw.gamma, Rw.alpha_const, Rw.pref, Rw.post_pref, Rw.O);
    }
}
__device__ __forceinline__ void massfn_block(const SigmaMassFnArgs& S, int b, int nt) {
    __shared__ double red[4 * 64];
    __shared__ double sig[64];
    sigma2_massfn_tile(S, b / nt, b - (b / nt) * nt, red, sig);
}
struct RowsGroupArgs {
    ChainArgs C;
    RowsArgs Rw;
    SigmaMassFnArgs S;
    int nchain, nrowblk, nmfblk, mf_ntile;
};
// (the NFW role's pointers are kernel parameters of their own: see nfw_rows)
#ifndef HMG_ROWS_OCC
#define HMG_ROWS_OCC 7
#endif
__global__ __launch_bounds__(256, HMG_ROWS_OCC) void rows_group_kernel(RowsGroupArgs G, const SiciTable* __restrict__ T,
                                                                      const double* __restrict__ acoef, int ktile, int nm,
                                                                      int nk, const double* __restrict__ cs,
                                                                      const double* __restrict__ rss,
                                                                      const double* __restrict__ zs,
                                                                      const double* __restrict__ ks,
                                                                      double* __restrict__ uk) {
    extern __shared__ double lds[];
    int b = blockIdx.x;
    if (b < G.nchain) {
        chain_row<256>(G.C, b, lds);
        return;
    }
    b -= G.nchain;
    if (b < G.nmfblk) {
        massfn_block(G.S, b, G.mf_ntile);
        return;
    }
    b -= G.nmfblk;
    if (b < G.nrowblk) {
        rows_block(G.Rw, b);
        return;
    }
    nfw_rows(T, acoef, ktile, nm, nk, cs, rss, zs, ks, uk, b - G.nrowblk, 256, threadIdx.x);
}

template <int MAXB, int MAXP, int SPECM>
__global__ __launch_bounds__(512, (fused_occ<MAXB, SPECM>())) void profile_group_kernel(ChainArgs C, FusedArgs A,
                                                                                              int nchain) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int b = blockIdx.x;
    if (b < nchain) {
        chain_row<512>(C, b, smem);
        return;
    }
    profile_fused_row<512, MAXB, MAXP, SPECM, false, true>(A, row_order(b - nchain, A.nm), smem);     // (A.rowsc is required)
}

// tensor group (512 threads): per-z chain INCLUDING sigma^2 -> n, b | fused radial-profile rows | analytic NFW rows -
// everything between the front and the mass integrals as one launch (the chain's first link used to be mass tiles of the
// rows group, a launch of its own in front of this one: ~11 us of a 0.1 ms thin-slab step).  The two kinds of rows are
// both bound by fp64 issue and take the sum of their times, one kernel boundary less.  Profile rows first: measured
// against alternating rows and NFW rows first on MI355X (0.308 / 0.324 / 0.321 ms at nz = 32).
template <int MAXB, int MAXP, int SPECM>
__global__ __launch_bounds__(512, (fused_occ<MAXB, SPECM>())) void tensor_group_kernel(
    ChainArgs C, FusedArgs A, int nchain, int nprof, const SiciTable* __restrict__ T, const double* __restrict__ acoef,
    int ktile, int nm, int nk, const double* __restrict__ cs, const double* __restrict__ rss,
    const double* __restrict__ zs, const double* __restrict__ ks, double* __restrict__ uk) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    int b = blockIdx.x;
    if (b < nchain) {
        chain_row<512, true>(C, b, smem);
        return;
    }
    b -= nchain;
    if (b < nprof) profile_fused_row<512, MAXB, MAXP, SPECM, false, true>(A, row_order(b, A.nm), smem);
    else nfw_rows(T, acoef, ktile, nm, nk, cs, rss, zs, ks, uk, b - nprof, 512, threadIdx.x);
}

}  // namespace hmg
