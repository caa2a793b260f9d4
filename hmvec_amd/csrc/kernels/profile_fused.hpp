// K45 - one radial-profile row per workgroup: integrand, in-LDS packed-real FFT, k-interpolation (hmvec/fft.py:35-115).
// Part of the ONE translation unit hmgrid.hip (included there in this order; not a stand-alone header).
#pragma once

namespace hmg {

// ---------------------------------------------------------------- K45: fused radial-profile transform
// One workgroup per (z,m) row does the whole of generic_profile_fft (hmvec/fft.py:56-115)
// without touching HBM in between: integrand + mass norm -> packed-real FFT in LDS
// (ldsfft.hpp) -> Im F_j -> u_j -> linear interpolation onto the target k grid.  The only
// HBM traffic is the (nk) output row plus per-row scalars; the rocFFT path it replaces moves
// 2*8*nxs + 2*16*(nxs/2+1) bytes per row through the memory system (3.2 GB at Config 3).
// Used when nxs is even, nxs/2 factors into 5/4/3/2 and fits LDS; otherwise hmg_profile_fft
// falls back to the chunked rocFFT path.
// (UnpackTw - the per-mode constants of the unpack step, one 32-byte load - lives in ldsfft.hpp)
// (FusedArgs - the description of a launch of radial-profile rows - lives in rowdev.hpp)

// ln x_n of the radial grid: the same for all (z,m) rows, so with many rows one small launch replaces a
// quarter of the integrand's transcendentals (same log_fast as the in-kernel path: identical bits).
__global__ void logx_kernel(int n, const double* __restrict__ xs, double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = log_fast(xs[i]);
}

// (gnfw_rho_fast: rowdev.hpp)

template <int NT, int R, int MAXB, bool SMALL = false, int NIN = R, int SRC_SHIFT = 0>
__device__ __forceinline__ void fused_pass(cplx* buf, const cplx* __restrict__ twM, int M, int Ns, int twstep,
                                           unsigned magic, int keep) {
    // keep >= 0 (last pass only, Ns == M/R): butterfly j writes Z[j + t*Ns]; only Z[0..keep] and
    // Z[M-keep..M-1] will be read, i.e. butterflies j <= keep (t = 0) and j >= Ns - keep (t = R-1).
    cplx v[MAXB][R];
    const int nb = M / R;
#pragma unroll
    for (int b = 0; b < MAXB; ++b) {
        const int j = threadIdx.x + b * NT;
        if (j < nb && (keep < 0 || j <= keep || j >= nb - keep)) pass_load<R, SMALL, NIN, SRC_SHIFT>(buf, twM, M, Ns, twstep, magic, j, v[b]);
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < MAXB; ++b) {
        const int j = threadIdx.x + b * NT;
        if (j < nb && (keep < 0 || j <= keep || j >= nb - keep)) pass_store<R, SMALL, NIN>(buf, Ns, magic, j, v[b]);
    }
    __syncthreads();
}

// The passes of a length-M transform whose plan the compiler knows (ldsfft.hpp: SubPass<M, PS>): strides, twiddle
// steps and index multipliers are immediates, every pass gets the butterflies-per-thread count it needs, and the
// pass loop with its radix dispatch is gone - what the run-time plan pays in registers (116 B/lane of scratch in
// round 3's profile_group_kernel<*,*,0>) and scalar instructions.
template <int NT, int M, int PS>
__device__ __forceinline__ void fused_passes_ct(cplx* buf, const cplx* __restrict__ twM, bool pruned, int jn) {
    if constexpr (PS < SubPass<M, 0>::P.npass) {
        using S = SubPass<M, PS>;
        constexpr int MAXB = (S::nb + NT - 1) / NT;
        // the last pass only has to produce Z[0..jn] and Z[M-jn..M-1]
        const int keep = (S::last && 2 * jn + 2 < S::nb) ? jn : -1;
        if (!(PS == 0 && pruned)) fused_pass<NT, S::R, MAXB, S::SMALL>(buf, twM + S::twoff, M, S::Ns, 1, S::mg, keep);
        fused_passes_ct<NT, M, PS + 1>(buf, twM, pruned, jn);
    }
}
template <int SPECM> constexpr int fused_first_radix() {
    if constexpr (SPECM == 0) return 0; else return SubPass<SPECM, 0>::R;
}

#ifndef HMG_FUSED_OCC
#define HMG_FUSED_OCC 8
#endif
// waves per SIMD a fused-row launch is compiled for: 8 with a compile-time plan (<= 64 VGPRs, no spill).  The
// run-time plan needs ~91 registers for its pass loop and dispatch chain; measured on the Config-3 grid at
// nxs = 3000 / 2000 (tools/shape_sweep.py): 8 waves/SIMD (64 VGPRs, 28-34 spilled, 116 B/lane) 0.294 / 0.277 ms,
// 6 (80 VGPRs, 12 spilled, 52 B/lane) 0.263 / 0.243 ms, 5 (91 VGPRs, nothing spilled) 0.320 / 0.285 ms: 6 it is.
// The lengths people use have compile-time plans (nxs = 1000, 2000, 4000, 5000: 0.129, 0.155, 0.193 ms for the
// first three against 0.190, 0.238, 0.280 with this run-time plan), so this path serves the odd ones.
#ifndef HMG_RT_OCC
#define HMG_RT_OCC 6
#endif
template <int MAXB, int SPECM> constexpr int fused_occ() { return MAXB > 2 ? 4 : (SPECM ? HMG_FUSED_OCC : HMG_RT_OCC); }
// SPECM != 0: the plan is known at compile time (SPECM = 2500, passes 4,5,5,5,5: nxs = 5000, the default
// length of the Battaglia profiles) - strides, twiddle steps and the j/Ns multipliers become immediates and
// the pass loop with its dispatch chain unrolls.
// TAB: the profile is read from a table (a user's callable evaluated on the x grid: hmvec/fft.py:56-94) instead of
// evaluated from the family; everything behind the integrand is the same code.
// RSC: the output-side scalars of the row come from the record the rows stage left (A.rowsc, hmg_rows_part ABI 8) - the
// grouped launch of the facade; without it one wavefront of the workgroup works them out (stand-alone launches).
template <int NT, int MAXB, int MAXP, int SPECM, bool TAB = false, bool RSC = false>
__device__ __forceinline__ void profile_fused_row(const FusedArgs& A, int row, double* smem) {
    // dynamic LDS only (base stays 16 B aligned for the 128-bit complex accesses):
    // [0, 2M) doubles = packed row as cplx, later u[0..M-1]; then 16 doubles of reduction
    // scratch, the broadcast mass norm and the left-fill counter.
    cplx* buf = reinterpret_cast<cplx*>(smem);
    const int M = SPECM ? SPECM : A.plan.M, nxs = SPECM ? 2 * SPECM : A.nxs;
    double* red = smem + 2 * (size_t)M;
    int* s_cnt = reinterpret_cast<int*>(red + 17);
    const double Aamp = A.amp ? A.amp[row] : A.amp_c;
    const double XC = A.xc ? A.xc[row] : A.xc_c;
    const double AL = A.alpha ? A.alpha[row] : A.alpha_c;
    const double EX = A.expo ? A.expo[row] : A.expo_c;
    const double cm = A.cmax[row];
    // ln(x/xc) = ln x - ln xc: ln x is row-independent (xc == 1 for the gas and NFW members: no logarithm)
    const double ln_xc = (TAB || (A.xc == nullptr && A.xc_c == 1.0)) ? 0.0 : log_fast(XC);
    const double* __restrict__ tab = TAB ? A.rho_tab + (A.rho_shared ? (size_t)0 : (size_t)row * (size_t)(SPECM ? 2 * SPECM : A.nxs)) : nullptr;
    // Output side of the row: the FFT modes sit on the uniform grid kout_j = j k_lo,
    // k_lo = kt_1 / (r_s (1+z)).  Targets below k_lo take np.interp's left fill u_1, targets above
    // kout_M are zero, and only the modes j <= jn = floor(max(ks)/k_lo) + 2 can be reached at all:
    // low-mass rows (large k_lo) need a few dozen of the M modes, so the unpack and the last FFT
    // pass are cut down to those.  max(ks) is only known without a search when ks is ascending,
    // which is the caller's promise that comes with the hint arrays (include/hmgrid.h).
    // These row scalars are the same for all 512 threads and cost a few divisions: one wavefront works them out
    // while the others start on the integrand, and they travel through LDS behind the barrier that is there
    // anyway (red[17..23]: length of the left-fill prefix, jn, 1/(r_s(1+z)), k_lo, k_hi, 1/k_lo, 1/kt_1).
    const int z = row / A.nm;
    int* s_jn = reinterpret_cast<int*>(red + 18);
    // The LAST wavefront works them out (in the truncated Battaglia rows it holds no non-zero sample, so it is the
    // one with nothing to do in phase A); every lane computes the same values and lane 0 stores them.  With the
    // hint arrays (ks ascending) the same wavefront also locates the end of the left-fill prefix - the first target
    // wavenumber that is not below k_lo - by a 64-way search: each lane tests the last wavenumber of its segment,
    // the number of lanes that see it below k_lo is the number of segments that lie in the prefix entirely, and the
    // next segment holds the boundary (two dependent loads for nk <= 4096).  Phase D then fills [0, nleft)
    // without loading or testing a wavenumber.
    // (with A.rowsc - the grouped passes of the facade - the launch that computed the rows' length scales left these
    // numbers per row: they arrive by scalar loads and no wavefront of this workgroup divides or searches)
    const double* __restrict__ rsc = RSC ? A.rowsc + (size_t)row * HMG_ROWSC_STRIDE : nullptr;
    if (!RSC && threadIdx.x >= NT - 64) {
        const int lane = threadIdx.x & 63;
        const double isc0 = 1.0 / (A.rss[row] * (1.0 + A.zs[z]));      // kout_j = kts[j] * isc
        const double klo0 = A.kts[1] * isc0;
        const double idk0 = 1.0 / klo0;
        int jn0 = M, nleft = 0;
        if (A.nconst) {
            const double tmax = A.ks[A.nk - 1] * idk0;
            if (tmax < (double)(M - 4)) jn0 = (int)tmax + 3;           // one spare mode for the rounding of tmax
            int base = 0, end = A.nk;
            for (;;) {
                const int stp = (end - base + 63) >> 6;
                const int first = base + lane * stp;
                bool below = false;
                if (first < end) {
                    const int last = first + stp - 1;
                    below = A.ks[last < end ? last : end - 1] < klo0;
                }
                base += __popcll(__ballot(below)) * stp;
                if (base >= end) { base = end; break; }
                if (stp == 1) break;
                end = base + stp < end ? base + stp : end;
            }
            nleft = base;
        }
        if (lane == 0) {
            *s_cnt = nleft;
            *s_jn = jn0;
            red[19] = isc0; red[20] = klo0; red[21] = A.kts[M] * isc0; red[22] = idk0;
            red[23] = 1.0 / A.kts[1];
        }
    }
    // ---- phase A: y_n = x_n rho(x_n) theta(x_n <= cmax) packed as (y_2p, y_2p+1); mass norm
    // Pruned first pass: the integrand is zero beyond the truncation radius (85 % of a Battaglia
    // row at xmax = 20).  When every packed sample p >= M/R0 is zero, the first radix-R0 pass
    // sees (v0, 0, ..., 0) in every butterfly, whose DFT is v0 in all R0 outputs - exactly, in
    // floating point - so phase A writes each sample straight into its R0 output slots and the
    // pass (an LDS round trip, two barriers, the zero fill of the rest of the row) is skipped.
    const int R0 = SPECM ? fused_first_radix<SPECM>() : A.plan.radix[0];
    const int stride0 = M / R0;
    const bool pruned = (SPECM || A.plan.npass > 1) && A.xs[2 * stride0] > cm;   // xs is increasing
    // compile-time plan: when the row is zero from sample 375 on, the pass behind the pruned one reads samples
    // 0..374 only (3-of-5 butterflies, below) and the rest of the row need not even be cleared
    const bool lead3 = SPECM == 2500 && pruned && A.xs[2 * 375] > cm;
    const int pend = lead3 ? 375 : (pruned ? stride0 : M);
    double acc = 0.0;
    for (int p = threadIdx.x; p < pend; p += NT) {
        const int j = 2 * p;
        const double2 xv = *reinterpret_cast<const double2*>(A.xs + j);
        double r0 = 0.0, r1 = 0.0;
        if constexpr (TAB) {
            if (!(fabs(xv.x) > cm)) r0 = tab[j];
            if (!(fabs(xv.y) > cm)) r1 = tab[j + 1];
        } else {
            if (!(fabs(xv.x) > cm)) r0 = gnfw_rho_fast((A.logx ? A.logx[j] : log_fast(xv.x)) - ln_xc, Aamp, AL, EX, A.gamma);
            if (!(fabs(xv.y) > cm)) r1 = gnfw_rho_fast((A.logx ? A.logx[j + 1] : log_fast(xv.y)) - ln_xc, Aamp, AL, EX, A.gamma);
        }
        const cplx y = cplx{xv.x * r0, xv.y * r1};
        if (pruned && SPECM != 2500) {
            // the R0 copies go out in an order rotated by lane/4: with t the same in every lane, lanes l and l+4
            // (64 B apart) hit the same LDS banks and every one of these 16-B stores takes two passes
            const int rot = (threadIdx.x >> 2);
            for (int t = 0; t < R0; ++t) buf[R0 * p + (t + rot) % R0] = y;
        } else {
            // (the hand-sequenced 2500 plan replicates nothing: its second pass reads slot i as sample i >> 2)
            buf[p] = y;
        }
        if (A.do_norm && (r0 != 0.0 || r1 != 0.0)) {
            const double xl = (j > 0) ? A.xs[j - 1] : xv.x, xr = (j + 2 < nxs) ? A.xs[j + 2] : xv.y;
            acc += 0.5 * (xv.y - xl) * (r0 * (xv.x * xv.x)) + 0.5 * (xr - xv.x) * (r1 * (xv.y * xv.y));
        }
    }
    // mass norm: wavefront sums (DPP), one LDS exchange, and EVERY thread adds the eight partials itself in
    // wave order - no second reduction stage.  The barrier also publishes buf and the row scalars the last
    // wavefront wrote (red[0..7] are written nowhere else, so nothing has to be waited for before).
    {
        const double ws = wave_sum(acc);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ws;
        __syncthreads();
    }
    // The first wavefront adds the eight partials in wave order and forms the one number the rest of the row needs
    // from the norm: the scale of the unpack step, u_j = Im F_j * (-step / (mnorm kt_1)) / j.  It travels through
    // red[24] behind the barriers of the FFT passes (a division and seven additions that 448 threads used to repeat).
    if (threadIdx.x < 64) {
        double tot = red[0];
#pragma unroll
        for (int w = 1; w < NT / 64; ++w) tot += red[w];
        const double mnorm = A.do_norm ? tot : 1.0;
        if (threadIdx.x == 0) red[24] = -A.step / mnorm * (RSC ? rsc[4] : red[23]);
    }
    const int jn = RSC ? __double2hiint(rsc[5]) : __builtin_amdgcn_readfirstlane(*s_jn);
    // ---- phase B: in-place Stockham FFT of length M
    // (Tried and dropped, MI355X: fetching all R operands before the twiddle products and requesting the
    // next pass's twiddle between the two halves of a pass.  Both lengthen live ranges under the 64-VGPR
    // cap of 8 waves/SIMD: 0.277 -> 0.315 ms.)
    if constexpr (SPECM == 2500) {
        // butterfly indices stay below 1024: the 24-bit index arithmetic of ldsfft.hpp (div_ns)
        constexpr unsigned mg4 = small_magic(4), mg20 = small_magic(20), mg100 = small_magic(100), mg500 = small_magic(500);
        // (A.twM is the per-pass twiddle table, ldsfft.hpp: the slices of the passes start at 0, 1, 5, 25, 125 and a
        // butterfly reads element k = j mod Ns of its pass's slice - consecutive lanes, consecutive elements)
        if (!pruned) fused_pass<NT, 4, MAXB, true>(buf, A.twM, 2500, 1, 1, 0u, -1);
        // behind the pruned first pass slot i of the row holds sample i/4: a butterfly of this pass reads slots
        // j + 500 t, and those with t >= 3 are zero when the row is zero from sample 1500/4 on (cmax < 3 at xmax = 20)
        if (pruned) {
            if (lead3) fused_pass<NT, 5, 1, true, 3, 2>(buf, A.twM + 1, 2500, 4, 1, mg4, -1);
            else fused_pass<NT, 5, 1, true, 5, 2>(buf, A.twM + 1, 2500, 4, 1, mg4, -1);
        } else fused_pass<NT, 5, 1, true>(buf, A.twM + 1, 2500, 4, 1, mg4, -1);
        fused_pass<NT, 5, 1, true>(buf, A.twM + 5, 2500, 20, 1, mg20, -1);
        fused_pass<NT, 5, 1, true>(buf, A.twM + 25, 2500, 100, 1, mg100, -1);
        fused_pass<NT, 5, 1, true>(buf, A.twM + 125, 2500, 500, 1, mg500, 2 * jn + 2 < 500 ? jn : -1);
    } else if constexpr (SPECM != 0) {
        fused_passes_ct<NT, SPECM, 0>(buf, A.twM, pruned, jn);
    } else
    for (int ps = pruned ? 1 : 0; ps < A.plan.npass; ++ps) {
        const int R = A.plan.radix[ps], Ns = A.plan.ns[ps], tws = 1;      // (per-pass twiddle table: element k of the slice)
        const unsigned mg = A.plan.magic[ps];
        const cplx* __restrict__ twp = A.twM + A.plan.twoff[ps];
        // a pass whose butterflies fit one per thread uses the MAXB = 1 body (fewer live registers)
        const bool one = (M / R) <= NT;
        // the last pass only has to produce Z[0..jn] and Z[M-jn..M-1]
        const int keep = (ps == A.plan.npass - 1 && 2 * jn + 2 < M / R) ? jn : -1;
        if (R == 5) { if (one) fused_pass<NT, 5, 1>(buf, twp, M, Ns, tws, mg, keep); else fused_pass<NT, 5, MAXB>(buf, twp, M, Ns, tws, mg, keep); }
        else if (R == 4) { if (one) fused_pass<NT, 4, 1>(buf, twp, M, Ns, tws, mg, keep); else fused_pass<NT, 4, MAXB>(buf, twp, M, Ns, tws, mg, keep); }
        else if (R == 3) { if (one) fused_pass<NT, 3, 1>(buf, twp, M, Ns, tws, mg, keep); else fused_pass<NT, 3, MAXB>(buf, twp, M, Ns, tws, mg, keep); }
        else { if (one) fused_pass<NT, 2, 1>(buf, twp, M, Ns, tws, mg, keep); else fused_pass<NT, 2, MAXB>(buf, twp, M, Ns, tws, mg, keep); }
    }
    // ---- phase C: Im F_j -> u_j = -Im F_j * step / kt_j / mnorm for the reachable modes
    // j = 1..jn, into smem[0..jn-1]
    const double sc = red[24];                    // u_j = Im F_j * this / j   (kt_j = j kt_1)
    double ua[MAXP], ub[MAXP];
    const int half = M / 2;
#pragma unroll
    for (int b = 0; b < MAXP; ++b) {
        const int j = 1 + threadIdx.x + b * NT;
        const bool hi = (M - j <= jn);                 // the mirrored mode M-j is reachable too
        if (j <= half && (j <= jn || hi)) {
            const cplx zj = buf[j], zmj = buf[M - j];
            const UnpackTw w = A.twN[j];
            double fa, fb;
            unpack_imag_pair(zj, zmj, w.co, w.si, fa, fb);
            ua[b] = fa * sc * w.rj;
            ub[b] = hi ? fb * sc * w.rmj : 0.0;
        }
    }
    __syncthreads();
    double* u = smem;
#pragma unroll
    for (int b = 0; b < MAXP; ++b) {
        const int j = 1 + threadIdx.x + b * NT;
        const bool hi = (M - j <= jn);
        if (j <= half && (j <= jn || hi)) {
            u[j - 1] = ua[b];
            if (hi && M - j >= 1) u[M - j - 1] = ub[b];
        }
    }
    if (threadIdx.x == 0) u[M - 1] = 0.0;  // Nyquist mode: Im F_M == 0
    __syncthreads();
    // ---- phase D: np.interp(ks, kout, u, left=u_1, right=0) on the uniform source grid:
    // bracket j = floor(k/k_lo), weight k/k_lo - j (one FMA), two LDS reads.  The left fill is a
    // plain splat (63 % of the Battaglia tensor at Config 3).
    const double k_lo = RSC ? rsc[1] : red[20], k_hi = RSC ? rsc[2] : red[21], inv_dk = RSC ? rsc[3] : red[22];
    const double pf = A.post ? A.post[row] : 1.0;
    const double u1 = u[0];
    double* __restrict__ dst = A.out + (size_t)row * A.nk;
    // with the hint arrays the left fill [0, nleft) is a plain fill in 16-byte stores (no wavenumber is loaded or
    // tested) and the interpolation starts at the 64-aligned index below nleft, so that its stores stay on whole
    // 512-byte wavefront segments; without them (ks in any order) every target is tested
    const int nleft = RSC ? __double2loint(rsc[5]) : (A.nconst ? __builtin_amdgcn_readfirstlane(*s_cnt) : 0);
    if (nleft > 0) {
        typedef double v2d __attribute__((ext_vector_type(2)));
        const double c = u1 * pf;
        const int head = (int)((reinterpret_cast<uintptr_t>(dst) >> 3) & 1);     // row start not 16-B aligned
        const int npair = (nleft - head) >> 1;
        v2d* __restrict__ d2 = reinterpret_cast<v2d*>(dst + head);
        const v2d cc = {c, c};
        for (int q = threadIdx.x; q < npair; q += NT) __builtin_nontemporal_store(cc, &d2[q]);
        if (threadIdx.x == 0) {
            if (head) __builtin_nontemporal_store(c, &dst[0]);
            if ((nleft - head) & 1) __builtin_nontemporal_store(c, &dst[nleft - 1]);
        }
    }
    auto interp = [&](double k) {
        int j = (int)(k * inv_dk);
        j = j < 1 ? 1 : (j > M - 1 ? M - 1 : j);
        const double fr = fma(k, inv_dk, -(double)j);
        const double y0 = u[j - 1], y1 = u[j];
        return fma(y1 - y0, fr, y0);
    };
    if (A.nconst) {
        // behind the prefix every target is at or above k_lo (ks ascending)
        for (int i = (nleft & ~63) + threadIdx.x; i < A.nk; i += NT) {
            if (i < nleft) continue;
            // (requesting the next trip's wavenumber one trip ahead was measured: +-0, the other wavefronts of
            // the workgroup already cover the load)
            const double k = A.ks[i];
            const double val = k > k_hi ? 0.0 : interp(k);
            __builtin_nontemporal_store(val * pf, &dst[i]);
        }
    } else {
        for (int i = threadIdx.x; i < A.nk; i += NT) {
            const double k = A.ks[i];
            const double val = k < k_lo ? u1 : (k > k_hi ? 0.0 : interp(k));
            __builtin_nontemporal_store(val * pf, &dst[i]);
        }
    }
    if (A.nconst && threadIdx.x == 0) {
        A.nconst[row] = nleft;
        A.cconst[row] = u1 * pf;
    }
}
template <int NT, int MAXB, int MAXP, int SPECM>
__global__ __launch_bounds__(NT, (fused_occ<MAXB, SPECM>())) void profile_fused_kernel(FusedArgs A) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    profile_fused_row<NT, MAXB, MAXP, SPECM>(A, blockIdx.x, smem);
}
template <int NT, int MAXB, int MAXP, int SPECM>
__global__ __launch_bounds__(NT, (fused_occ<MAXB, SPECM>())) void profile_table_kernel(FusedArgs A) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    profile_fused_row<NT, MAXB, MAXP, SPECM, true>(A, blockIdx.x, smem);
}

// (K45p, the long radial grids with short support - profile_pruned_kernel and the chirp route: longgrid.hip, a
// translation unit of its own.  In this one the mere presence of its instantiations changed the address arithmetic
// hipcc emits for profile_group_kernel<2,3,2500> - 605 instead of 593 VALU instructions per wavefront.)

}  // namespace hmg
