// Long radial grids with short support: the row kernels behind hmg_profile_fft for nxs = 30000 / 40000 - the grids the
// reference's own callers use (examples/lensing_baryons.py:27, bin/tests.py:308, hmvec/params.py:59-60).
// A translation unit of its own (see the note in hmgrid.hip where this code used to be).  Design: DESIGN.md section 3.
#include <hip/hip_runtime.h>

#include "longgrid.hpp"

namespace hmg {

// ---------------------------------------------------------------- K45p: long radial grids with short support
// nxs = 30000 / 40000 - what the reference's own callers pass (examples/lensing_baryons.py:27 and bin/tests.py:308:
// add_battaglia_profile(xmax=50, nxs=30000); hmvec/params.py:59-60: numeric NFW, nxs = 40000, xmax = 200) - do not
// fit LDS as one packed row (M = nxs/2 complex = 240-320 KB), but the profile is cut at cmax << xmax: only the first
// P0 = ceil(#{x_n <= cmax} / 2) packed samples are non-zero (820 of 15000 for the gas profile at xmax = 50).  With
// LP >= P0, M = R LP, the transform is R transforms of length LP of the row times W_M^{rp} (ldsfft.hpp, "pruned
// decomposition"): one workgroup per (z,m) row keeps the P0 samples in LDS, transforms the residues r and R - r
// side by side with the compile-time plan of length LP, unpacks the pair on the spot into u_j (a per-row scratch
// line in HBM/L2: up to M modes do not fit LDS either) and interpolates as the fused kernel does.  Nothing of
// length nxs is ever written: the rocFFT route this replaces moves 2*8*nxs + 2*16*(nxs/2+1) bytes per row.

// The passes 1 .. npass-1 of the sub-transforms (pass 0 runs from registers, profile_pruned_row).
// (Measured and dropped, MI355X: a thread works on the same butterflies in every group of residues, so its twiddle
// per pass can be fetched once per row and held in registers - 16 more VGPRs at LP = 1000 spill inside the group loop
// under the 64- and the 80-register caps alike: 1.16 -> 2.38 / 2.04 ms.)
template <int NT, int LP, int PS, int NBUF = 2>
__device__ __forceinline__ void pruned_passes(cplx* buf, const cplx* twL, int nbuf, int keep) {
    if constexpr (PS < SubPass<LP, 0>::P.npass) {
        using S = SubPass<LP, PS>;
        constexpr int MAXB = (NBUF * S::nb + NT - 1) / NT;
        cplx v[MAXB][S::R];
#pragma unroll
        for (int b = 0; b < MAXB; ++b) {
            const int jj = threadIdx.x + b * NT;
            if (sub_pass_active<LP, PS>(jj, nbuf, keep)) sub_pass_load<LP, PS>(buf, twL, jj, v[b]);
        }
        __syncthreads();
#pragma unroll
        for (int b = 0; b < MAXB; ++b) {
            const int jj = threadIdx.x + b * NT;
            if (sub_pass_active<LP, PS>(jj, nbuf, keep)) sub_pass_store<LP, PS>(buf, jj, v[b]);
        }
        __syncthreads();
        pruned_passes<NT, LP, PS + 1, NBUF>(buf, twL, nbuf, keep);
    }
}
// lengths whose chirp route is compiled in: the thread that owns the samples j, j + LP/2 of the decomposition's
// radix-2 first pass owns exactly the two non-zero inputs of butterfly j of the radix-4 first pass at Lc = 2 LP
template <int LP> constexpr bool chirp_ok() {
    if constexpr (LP == 1000 || LP == 1250) return SubPass<LP, 0>::R == 2 && SubPass<2 * LP, 0>::R == 4;
    else return false;
}

#ifndef HMG_PRUNED_ULDS
#define HMG_PRUNED_ULDS 1
#endif
// LDS of a row workgroup: [0, 2 LP) cplx transform buffers | 32 doubles of scalars | the modes of a chirp row
// (rows with more modes than fit take the decomposition)
template <int LP> constexpr int pruned_uls_doubles() { return (HMG_PRUNED_ULDS && chirp_ok<LP>()) ? (LP / 2 + LP / 8 + 8) / 2 * 2 : 0; }
template <int LP> constexpr size_t pruned_lds_bytes() {
    return (size_t)2 * LP * 16 + (32 + pruned_uls_doubles<LP>()) * sizeof(double);
}
template <int NT, int LP>
__device__ __forceinline__ void profile_pruned_row(const PrunedArgs& G, int row, double* smem) {
    const FusedArgs& A = G.F;
    // dynamic LDS: [0, 2 LP) cplx = the two transform buffers, then 32 doubles of scalars laid out as in
    // profile_fused_row.  The packed samples of the row stay in REGISTERS: thread j < LP/R0 owns the R0 inputs
    // j + t LP/R0 of butterfly j of the first pass (radix R0, sub-transform size 1: no pass twiddles), so the
    // multiplication by W_M^{rp} and the first pass of every residue's transform need no LDS read at all.
    cplx* buf = reinterpret_cast<cplx*>(smem);
    double* red = smem + 4 * (size_t)LP;
    int* s_cnt = reinterpret_cast<int*>(red + 17);
    int* s_jn = reinterpret_cast<int*>(red + 18);
    // behind the scalars: the modes u_j of a row that took the chirp route (jn <= Jw: a few hundred) - they never
    // leave the CU; only the rows of the decomposition use the scratch line in HBM/L2
    double* uls = red + 32;
    const cplx* __restrict__ twl = G.twL;
    using S0 = SubPass<LP, 0>;
    constexpr int R0 = S0::R, nb0 = S0::nb, MAXB0 = (nb0 + NT - 1) / NT;
    static_assert(S0::Ns == 1, "first pass");
    const int M = G.M, R = G.R, nxs = 2 * M;
    const double Aamp = A.amp ? A.amp[row] : A.amp_c;
    const double XC = A.xc ? A.xc[row] : A.xc_c;
    const double AL = A.alpha ? A.alpha[row] : A.alpha_c;
    const double EX = A.expo ? A.expo[row] : A.expo_c;
    const double cm = A.cmax[row];
    const double ln_xc = (A.xc == nullptr && A.xc_c == 1.0) ? 0.0 : log_fast(XC);
    const int z = row / A.nm;
    double* __restrict__ dst = A.out + (size_t)row * A.nk;
    // The plan was sized from a bound on the support (profile_support); a row that exceeds it cannot be transformed
    // here: it is filled with NaN and the context's fault word is raised, which the next synchronising call reports.
    if (!(A.xs[2 * LP] > cm)) {                    // (2 LP < nxs: R >= 2; xs increasing)
        for (int i = threadIdx.x; i < A.nk; i += NT) dst[i] = __builtin_nan("");
        if (threadIdx.x == 0) {
            __hip_atomic_store(G.fault, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (A.nconst) { A.nconst[row] = 0; A.cconst[row] = __builtin_nan(""); }
        }
        return;
    }
    // row scalars and the end of the left-fill prefix: the last wavefront, as in profile_fused_row
    if (threadIdx.x >= NT - 64) {
        const int lane = threadIdx.x & 63;
        const double isc0 = 1.0 / (A.rss[row] * (1.0 + A.zs[z]));
        const double klo0 = A.kts[1] * isc0;
        const double idk0 = 1.0 / klo0;
        int jn0 = M, nleft = 0;
        if (A.nconst) {
            const double tmax = A.ks[A.nk - 1] * idk0;
            if (tmax < (double)(M - 4)) jn0 = (int)tmax + 3;
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
    // ---- phase A: the LP packed samples that can be non-zero (into registers), and the mass norm
    const double* __restrict__ tab = A.rho_tab ? A.rho_tab + (A.rho_shared ? (size_t)0 : (size_t)row * (size_t)nxs) : nullptr;
    cplx zp[MAXB0][R0];
    double acc = 0.0;
#pragma unroll
    for (int b = 0; b < MAXB0; ++b) {
#pragma unroll
        for (int t = 0; t < R0; ++t) {
            const int jb = threadIdx.x + b * NT;
            zp[b][t] = cplx{0.0, 0.0};
            if (jb < nb0) {
                const int j = 2 * (jb + t * nb0);
                const double2 xv = *reinterpret_cast<const double2*>(A.xs + j);
                double r0 = 0.0, r1 = 0.0;
                if (tab) {       // a user's profile from a table (hmg_profile_fft_table): wave-uniform
                    if (!(fabs(xv.x) > cm)) r0 = tab[j];
                    if (!(fabs(xv.y) > cm)) r1 = tab[j + 1];
                } else {
                    if (!(fabs(xv.x) > cm)) r0 = gnfw_rho_fast((A.logx ? A.logx[j] : log_fast(xv.x)) - ln_xc, Aamp, AL, EX, A.gamma);
                    if (!(fabs(xv.y) > cm)) r1 = gnfw_rho_fast((A.logx ? A.logx[j + 1] : log_fast(xv.y)) - ln_xc, Aamp, AL, EX, A.gamma);
                }
                zp[b][t] = cplx{xv.x * r0, xv.y * r1};
                if (A.do_norm && (r0 != 0.0 || r1 != 0.0)) {
                    const double xl = (j > 0) ? A.xs[j - 1] : xv.x, xr = (j + 2 < nxs) ? A.xs[j + 2] : xv.y;
                    acc += 0.5 * (xv.y - xl) * (r0 * (xv.x * xv.x)) + 0.5 * (xr - xv.x) * (r1 * (xv.y * xv.y));
                }
            }
        }
    }
    {
        const double ws = wave_sum(acc);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ws;
        __syncthreads();
    }
    if (threadIdx.x < 64) {
        double tot = red[0];
#pragma unroll
        for (int w = 1; w < NT / 64; ++w) tot += red[w];
        const double mnorm = A.do_norm ? tot : 1.0;
        if (threadIdx.x == 0) red[24] = -A.step / mnorm * red[23];
    }
    const int jn = __builtin_amdgcn_readfirstlane(*s_jn);
    // ---- phase B + C: per group of residues {g, R - g}: first pass from registers, the other passes in LDS,
    // unpack into the scratch line
    double* u = G.u + (size_t)(row - G.row0) * M;
    constexpr int nb_last = SubPass<LP, S0::P.npass - 1>::nb;
    const int keep = pruned_keep(R, M, nb_last, jn);
    bool chirped = false;
    if constexpr (chirp_ok<LP>()) {
        // ---- rows that need few modes: the chirp route - two transforms of length 2 LP instead of R of length LP
        // (the window was built for supports up to p0 packed samples: a row beyond it takes the decomposition)
        if (G.Bw != nullptr && jn <= G.Jw && (!HMG_PRUNED_ULDS || jn <= pruned_uls_doubles<LP>()) &&
            A.xs[2 * G.p0 < nxs ? 2 * G.p0 : nxs - 1] > cm) {
            chirped = true;
            constexpr int LC = 2 * LP;
            using C0 = SubPass<LC, 0>;
            static_assert(C0::nb == nb0 && MAXB0 == (C0::nb + NT - 1) / NT, "sample ownership");
            constexpr int nb_last_c = SubPass<LC, C0::P.npass - 1>::nb;
#pragma unroll
            for (int b = 0; b < MAXB0; ++b) {
                const int jb = threadIdx.x + b * NT;
                if (jb < nb0) {
                    cplx v[4];
                    chirp_first_pass(zp[b][0], zp[b][1], G.chP[jb], G.chP[jb + nb0], v);
#pragma unroll
                    for (int t = 0; t < 4; ++t) buf[4 * jb + t] = v[t];
                }
            }
            __syncthreads();
            pruned_passes<NT, LC, 1, 1>(buf, G.twC, 1, -1);
            {   // product with the window's transform, fused into the first pass of the second transform
                cplx v[MAXB0][4];
#pragma unroll
                for (int b = 0; b < MAXB0; ++b) {
                    const int jb = threadIdx.x + b * NT;
                    if (jb < nb0) {
#pragma unroll
                        for (int t = 0; t < 4; ++t) v[b][t] = cmul(buf[jb + t * nb0], G.Bw[jb + t * nb0]);
                    }
                }
                __syncthreads();
#pragma unroll
                for (int b = 0; b < MAXB0; ++b) {
                    const int jb = threadIdx.x + b * NT;
                    if (jb < nb0) {
                        dft_small<4>(v[b]);
#pragma unroll
                        for (int t = 0; t < 4; ++t) buf[4 * jb + t] = v[b][t];
                    }
                }
                __syncthreads();
            }
            pruned_passes<NT, LC, 1, 1>(buf, G.twC, 1, (2 * jn + 2 < nb_last_c) ? jn : -1);
            const double sc = red[24];
            double* __restrict__ ud = HMG_PRUNED_ULDS ? uls : u;
            for (int j = 1 + (int)threadIdx.x; j <= jn; j += NT) {
                const UnpackTw w = A.twN[j];
                ud[j - 1] = chirp_unpack(buf, LC, j, G.chJ[j], w) * sc * w.rj;
            }
        }
    }
    for (int g = 0; g <= R / 2 && !chirped; ++g) {
        if (!pruned_group_needed(R, M, g, jn)) break;          // (groups are needed in ascending order of g)
        const int s1 = pruned_group_partner(R, g), nbuf = s1 < 0 ? 1 : 2;
#pragma unroll
        for (int b = 0; b < MAXB0; ++b) {
            const int jb = threadIdx.x + b * NT;
            if (jb < nb0) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (h < nbuf) {
                        const int sres = h ? s1 : g;
                        cplx v[R0];
                        const cplx* __restrict__ twr = G.twR + (size_t)sres * LP;      // W_M^(sres p): consecutive p
#pragma unroll
                        for (int t = 0; t < R0; ++t) v[t] = cmul(zp[b][t], twr[jb + t * nb0]);
                        dft_small<R0>(v);
#pragma unroll
                        for (int t = 0; t < R0; ++t) buf[h * LP + jb * R0 + t] = v[t];      // Ns = 1: q = j, k = 0
                    }
                }
            }
        }
        __syncthreads();                                       // (also publishes red[24] before the first unpack)
        pruned_passes<NT, LP, 1>(buf, twl, nbuf, keep);
        const double sc = red[24];
        pruned_unpack(buf, LP, R, M, g, 0, nbuf == 2 ? 1 : 0, jn, G.twNr, sc, u, (int)threadIdx.x, NT);
        if (nbuf == 2) pruned_unpack(buf, LP, R, M, s1, 1, 0, jn, G.twNr, sc, u, (int)threadIdx.x, NT);
        __syncthreads();                                       // the next group overwrites the buffers
    }
    // A chirp row's modes sit by mode number - u_j at [j-1] - in LDS (or, without HMG_PRUNED_ULDS, in the scratch line);
    // the decomposition's in the scratch line by residue.  Mode M (Nyquist, Im F_M == 0) has no slot there: the
    // interpolation substitutes the zero.
    if (HMG_PRUNED_ULDS && chirped) u = uls;                   // (jn <= Jw < M - 4: mode M is never read)
    else __threadfence_block();
    __syncthreads();                                           // u is read by other threads below
    // ---- phase D: as profile_fused_row, the modes read from the scratch line
    const double k_lo = red[20], k_hi = red[21], inv_dk = red[22];
    const double pf = A.post ? A.post[row] : 1.0;
    const double u1 = chirped ? u[0] : u[R > 1 ? LP : 1];        // mode 1: residue 1, quotient 0
    const int nleft = A.nconst ? __builtin_amdgcn_readfirstlane(*s_cnt) : 0;
    if (nleft > 0) {
        typedef double v2d __attribute__((ext_vector_type(2)));
        const double c = u1 * pf;
        const int head = (int)((reinterpret_cast<uintptr_t>(dst) >> 3) & 1);
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
        double y0, y1;
        if (chirped) {
            y0 = u[j - 1]; y1 = u[j];
        } else {                                   // modes j and j + 1 of the residue-major line
            const int q = (int)fast_div((unsigned)j, G.rmagic), sr = j - q * R;
            const int i0 = sr * LP + q;
            const int i1 = sr + 1 < R ? i0 + LP : q + 1;
            y0 = u[i0];
            y1 = j + 1 < M ? u[i1] : 0.0;
        }
        return fma(y1 - y0, fr, y0);
    };
    if (A.nconst) {
        for (int i = (nleft & ~63) + threadIdx.x; i < A.nk; i += NT) {
            if (i < nleft) continue;
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
#ifndef HMG_PRUNED_OCC
#define HMG_PRUNED_OCC 0
#endif
// waves per SIMD the LDS footprint (two buffers of LP complex numbers, the scalars, the chirp rows' modes) allows:
// NT/64 per workgroup
template <int NT, int LP> constexpr int pruned_occ() {
    constexpr int wgs = (160 * 1024) / (int)pruned_lds_bytes<LP>();
    constexpr int w = wgs * (NT / 64) / 4;
    return HMG_PRUNED_OCC ? HMG_PRUNED_OCC : (w >= 8 ? 8 : (w < 1 ? 1 : w));
}
template <int NT, int LP>
__global__ __launch_bounds__(NT, (pruned_occ<NT, LP>())) void profile_pruned_kernel(PrunedArgs G) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    profile_pruned_row<NT, LP>(G, G.row0 + blockIdx.x, smem);
}

// ---- Rows whose support does not prune but which need few modes: the narrow-band route (ldsfft.hpp).  The tSZ notebook
// of the reference (examples/tSZ example.ipynb: add_battaglia_pres_profile("y", family="pres", xmax=2, nxs=30000)) is
// the case: with xmax = 2 the pressure profile fills half the grid and more - nothing to prune on the input side, the
// pruned decomposition does not apply - while r_s = R_200c and the coarse mode spacing 2 pi / 2 put every needed mode
// below j ~ 250 of 15000.  One workgroup per (z,m) row; per residue p1 < D of the sample index: the decimated row's
// first pass straight from the integrand (the thread that owns the inputs of butterfly j evaluates them), the other
// passes of the length-LB plan in LDS with the last one pruned to the band, and one multiply-add per needed mode into
// a register accumulator with the running twiddle (W_M^j)^{p1}.  Every sample is evaluated exactly once; the modes
// never leave the CU (u_j sits in LDS for the interpolation, as in the one-row kernel).  G.R is D here.
// The decimated rows read x_n, ln x_n and the trapezoid weights of the mass norm at a stride of D sample pairs: as they
// lie in memory that is one 16-byte piece per 64-byte line and lane.  One small launch per profile call lays the three
// out the way the row kernel walks them - pair (p1, p2) at [p1 LB + p2], p = p1 + D p2 - so that a wavefront's loads are
// contiguous again (G.u holds the tables: 3 nxs doubles).  w_n = (x_{n+1} - x_{n-1})/2 with the one-sided ends of np.trapz.
#ifndef HMG_BAND_NBUF
#define HMG_BAND_NBUF 2
#endif
__global__ void band_tables_kernel(int nxs, int D, int LB, const double* __restrict__ xs, const double* __restrict__ logx,
                                   double2* __restrict__ xT, double2* __restrict__ lT, double2* __restrict__ wT) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;       // q = p1 LB + p2
    if (q >= nxs / 2) return;
    const int p1 = q / LB, p2 = q - p1 * LB;
    const int j = 2 * (p1 + D * p2);
    const double x0 = xs[j], x1 = xs[j + 1];
    const double xl = j > 0 ? xs[j - 1] : x0, xr = j + 2 < nxs ? xs[j + 2] : x1;
    xT[q] = make_double2(x0, x1);
    lT[q] = logx ? make_double2(logx[j], logx[j + 1]) : make_double2(log_fast(x0), log_fast(x1));
    wT[q] = make_double2(0.5 * (x1 - xl), 0.5 * (xr - x0));
}

template <int NT, int LB, int MAXA>
__device__ __forceinline__ void profile_band_row(const PrunedArgs& G, int row, double* smem) {
    const FusedArgs& A = G.F;
    // dynamic LDS: [0, NBUF LB) cplx = the transform buffers (later the band's modes), then LB/2 doubles of u_j, then
    // 32 doubles of scalars laid out as in profile_fused_row.  Two residues p1, p1 + 1 of the sample index are
    // transformed side by side (HMG_BAND_NBUF = 2): every thread has a butterfly in every pass and a row passes
    // half as many barriers as with one buffer.
    cplx* buf = reinterpret_cast<cplx*>(smem);
    double* us = smem + 2 * (size_t)LB * HMG_BAND_NBUF;
    double* red = us + LB / 2 + (LB / 2 & 1);
    int* s_cnt = reinterpret_cast<int*>(red + 17);
    int* s_jn = reinterpret_cast<int*>(red + 18);
    using S0 = SubPass<LB, 0>;
    constexpr int R0 = S0::R, nb0 = S0::nb, MAXB0 = (nb0 + NT - 1) / NT;
    // MAXA: accumulator slots per thread; 2 jn + 1 <= MAXA NT of them are in use (the launch picks 1 when the bound on
    // the needed modes allows: 24 registers fewer)
    constexpr int nb_last = SubPass<LB, S0::P.npass - 1>::nb;
    static_assert(S0::Ns == 1, "first pass");
    const int M = G.M, D = G.R;
    const double Aamp = A.amp ? A.amp[row] : A.amp_c;
    const double XC = A.xc ? A.xc[row] : A.xc_c;
    const double AL = A.alpha ? A.alpha[row] : A.alpha_c;
    const double EX = A.expo ? A.expo[row] : A.expo_c;
    const double cm = A.cmax[row];
    const double ln_xc = (A.xc == nullptr && A.xc_c == 1.0) ? 0.0 : log_fast(XC);
    // alpha == 1 for every row (the pressure profile this route was built for): t^alpha = x / xc, no exponential
    const bool alpha1 = A.alpha == nullptr && A.alpha_c == 1.0;
    const double inv_xc = fm_rcp(XC);
    const int z = row / A.nm;
    double* __restrict__ dst = A.out + (size_t)row * A.nk;
    // row scalars and the end of the left-fill prefix: the last wavefront, as in profile_fused_row (hints are a
    // precondition of this route: without them every mode is needed and the launch would not have come here)
    if (threadIdx.x >= NT - 64) {
        const int lane = threadIdx.x & 63;
        const double isc0 = 1.0 / (A.rss[row] * (1.0 + A.zs[z]));
        const double klo0 = A.kts[1] * isc0;
        const double idk0 = 1.0 / klo0;
        int jn0 = M, nleft = 0;
        if (A.nconst) {
            const double tmax = A.ks[A.nk - 1] * idk0;
            if (tmax < (double)(M - 4)) jn0 = (int)tmax + 3;
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
    __syncthreads();
    const int jn = __builtin_amdgcn_readfirstlane(*s_jn);
    // The launch was sized from a bound on the needed modes (profile_support); a row beyond it cannot be done here: NaN
    // and the context's fault word, as for the support bound of the pruned route.
    if (2 * jn + 2 > LB || 2 * jn + 1 > MAXA * NT) {
        for (int i = threadIdx.x; i < A.nk; i += NT) dst[i] = __builtin_nan("");
        if (threadIdx.x == 0) {
            __hip_atomic_store(G.fault, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (A.nconst) { A.nconst[row] = 0; A.cconst[row] = __builtin_nan(""); }
        }
        return;
    }
    const int nslot = 2 * jn + 1;
    // The mode twiddles W_M^(p1 j) of a round come from the residue-major table (ldsfft.hpp: residue_tw_table(M, LB),
    // [p1 LB + |j|], conjugated for j < 0): consecutive slots read consecutive elements, every value is exact, and a
    // thread holds nothing but its accumulators (running products W^j (W^j)^p1 cost two more complex registers per slot
    // and a complex product per slot and residue - with them the two-slot build spilled 15 registers in this loop).
    cplx acc[MAXA];
#pragma unroll
    for (int a = 0; a < MAXA; ++a) acc[a] = cplx{0.0, 0.0};
    const int keep = (2 * jn + 2 < nb_last) ? jn : -1;
    const double2* __restrict__ xT = reinterpret_cast<const double2*>(G.u);
    const double2* __restrict__ lT = xT + M;
    const double2* __restrict__ wT = lT + M;
    double nrm = 0.0;
    for (int p1 = 0; p1 < D; p1 += HMG_BAND_NBUF) {
        const int nbuf = (HMG_BAND_NBUF == 2 && p1 + 1 < D) ? 2 : 1;
        // first pass (radix R0, sub-transform size 1) of the decimated rows z[p1 + h + D p2], straight from the integrand
#pragma unroll
        for (int h = 0; h < HMG_BAND_NBUF; ++h) {
            if (h < nbuf) {
#pragma unroll
                for (int b = 0; b < MAXB0; ++b) {
                    const int jb = threadIdx.x + b * NT;
                    if (jb < nb0) {
                        cplx v[R0];
#pragma unroll
                        for (int t = 0; t < R0; ++t) {
                            const int q = (p1 + h) * LB + jb + t * nb0;             // pair p = p1 + h + D p2, p2 = jb + t nb0
                            const double2 xv = xT[q], lv = lT[q], wv = wT[q];
                            double r0 = 0.0, r1 = 0.0;
                            if (alpha1) {
                                if (!(fabs(xv.x) > cm)) r0 = gnfw_rho_alpha1(lv.x - ln_xc, xv.x * inv_xc, Aamp, EX, A.gamma);
                                if (!(fabs(xv.y) > cm)) r1 = gnfw_rho_alpha1(lv.y - ln_xc, xv.y * inv_xc, Aamp, EX, A.gamma);
                            } else {
                                if (!(fabs(xv.x) > cm)) r0 = gnfw_rho_fast(lv.x - ln_xc, Aamp, AL, EX, A.gamma);
                                if (!(fabs(xv.y) > cm)) r1 = gnfw_rho_fast(lv.y - ln_xc, Aamp, AL, EX, A.gamma);
                            }
                            v[t] = cplx{xv.x * r0, xv.y * r1};
                            if (A.do_norm && (r0 != 0.0 || r1 != 0.0))
                                nrm += wv.x * (r0 * (xv.x * xv.x)) + wv.y * (r1 * (xv.y * xv.y));
                        }
                        dft_small<R0>(v);
#pragma unroll
                        for (int t = 0; t < R0; ++t) buf[h * LB + jb * R0 + t] = v[t];
                    }
                }
            }
        }
        __syncthreads();
        pruned_passes<NT, LB, 1, HMG_BAND_NBUF>(buf, G.twL, nbuf, keep);
        const cplx* __restrict__ twm = G.twR + (size_t)p1 * LB;
#pragma unroll
        for (int a = 0; a < MAXA; ++a) {
            const int t = threadIdx.x + a * NT;
            if (t < nslot) {
                const int j = band_mode(t, jn), idx = band_index(j, LB), ja = j < 0 ? -j : j;
                cplx w = twm[ja];                                                  // W_M^(p1 j): residue p1 ...
                if (j < 0) w.y = -w.y;
                acc[a] = cadd(acc[a], cmul(buf[idx], w));
                if (nbuf == 2) {
                    w = twm[LB + ja];                                              // ... then p1 + 1: the same order of sums
                    if (j < 0) w.y = -w.y;
                    acc[a] = cadd(acc[a], cmul(buf[LB + idx], w));
                }
            }
        }
        __syncthreads();                                       // the next residues' first pass overwrites the buffers
    }
    // mass norm (the order of the partial sums differs from the one-row kernel: per thread over its samples of all
    // residues, then wavefronts in order) and the scale of the unpack step
    {
        const double ws = wave_sum(nrm);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ws;
    }
    // the band's modes to LDS: slot t = mode t - jn
#pragma unroll
    for (int a = 0; a < MAXA; ++a) {
        const int t = threadIdx.x + a * NT;
        if (t < nslot) buf[t] = acc[a];
    }
    __syncthreads();
    double tot = red[0];
#pragma unroll
    for (int w = 1; w < NT / 64; ++w) tot += red[w];
    const double sc = -A.step / (A.do_norm ? tot : 1.0) * red[23];
    for (int j = 1 + (int)threadIdx.x; j <= jn; j += NT) {
        const UnpackTw w = A.twN[j];
        double fa, fb;
        unpack_imag_pair(buf[jn + j], buf[jn - j], w.co, w.si, fa, fb);
        us[j - 1] = fa * sc * w.rj;
    }
    __syncthreads();
    // ---- interpolation: as profile_fused_row, the modes read from LDS (targets beyond mode jn - 2 do not exist:
    // jn = floor(max(ks)/k_lo) + 3)
    const double k_hi = red[21], inv_dk = red[22];
    const double pf = A.post ? A.post[row] : 1.0;
    const double u1 = us[0];
    const int nleft = __builtin_amdgcn_readfirstlane(*s_cnt);
    if (nleft > 0) {
        typedef double v2d __attribute__((ext_vector_type(2)));
        const double c = u1 * pf;
        const int head = (int)((reinterpret_cast<uintptr_t>(dst) >> 3) & 1);
        const int npair = (nleft - head) >> 1;
        v2d* __restrict__ d2 = reinterpret_cast<v2d*>(dst + head);
        const v2d cc = {c, c};
        for (int q = threadIdx.x; q < npair; q += NT) __builtin_nontemporal_store(cc, &d2[q]);
        if (threadIdx.x == 0) {
            if (head) __builtin_nontemporal_store(c, &dst[0]);
            if ((nleft - head) & 1) __builtin_nontemporal_store(c, &dst[nleft - 1]);
        }
    }
    for (int i = (nleft & ~63) + threadIdx.x; i < A.nk; i += NT) {
        if (i < nleft) continue;
        const double k = A.ks[i];
        double val = 0.0;
        if (!(k > k_hi)) {
            int j = (int)(k * inv_dk);
            j = j < 1 ? 1 : (j > jn - 1 ? jn - 1 : j);
            const double fr = fma(k, inv_dk, -(double)j);
            const double y0 = us[j - 1], y1 = us[j];
            val = fma(y1 - y0, fr, y0);
        }
        __builtin_nontemporal_store(val * pf, &dst[i]);
    }
    if (threadIdx.x == 0) {
        A.nconst[row] = nleft;
        A.cconst[row] = u1 * pf;
    }
}
#ifndef HMG_BAND_OCC
#define HMG_BAND_OCC 6
#endif
template <int NT, int LB, int MAXA>
__global__ __launch_bounds__(NT, HMG_BAND_OCC) void profile_band_kernel(PrunedArgs G) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    profile_band_row<NT, LB, MAXA>(G, G.row0 + blockIdx.x, smem);
}

// Upper bound of the support of a launch's rows: max over rows of the number of PACKED samples that can be non-zero,
// ceil(#{n : x_n <= cmax[row]} / 2) (xs increasing; the mask of hmvec/fft.py:81 is strict, |x| > cmax).
// out[1]: the largest needed mode of the launch, jn = floor(max(ks)/k_lo) + 3 exactly as the row kernels form it (ks
// ascending: the promise that comes with the hint arrays; without it out[1] stays 0 and means "every mode").
__global__ void profile_support_kernel(int rows, int nxs, const double* __restrict__ xs, const double* __restrict__ cmax,
                                       const double* __restrict__ rss, const double* __restrict__ zs, int nm,
                                       const double* __restrict__ kts, const double* __restrict__ ks, int nk,
                                       int* __restrict__ out) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    int p0 = 0, jn = 0;
    if (row < rows && rss != nullptr) {
        const double isc0 = 1.0 / (rss[row] * (1.0 + zs[row / nm]));
        const double klo0 = kts[1] * isc0;
        const double idk0 = 1.0 / klo0;
        const double tmax = ks[nk - 1] * idk0;
        jn = tmax < (double)(nxs / 2 - 4) ? (int)tmax + 3 : nxs / 2;
    }
    for (int off = 32; off; off >>= 1) jn = max(jn, __shfl_xor(jn, off));
    if ((threadIdx.x & 63) == 0 && jn > 0) atomicMax(out + 1, jn);
    if (row < rows) {
        const double cm = cmax[row];
        int lo = 0, hi = nxs;                       // first n with xs[n] > cm
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (xs[mid] > cm) hi = mid; else lo = mid + 1;
        }
        p0 = (lo + 1) >> 1;
        if (!(cm == cm)) p0 = nxs;                  // NaN cmax: nothing is masked (|x| > NaN is false)
    }
    for (int off = 32; off; off >>= 1) p0 = max(p0, __shfl_xor(p0, off));
    if ((threadIdx.x & 63) == 0 && p0 > 0) atomicMax(out, p0);
}


template <int LP>
static int launch_pruned_lp(hipStream_t stream, PrunedArgs G, int rows, size_t rows_per_launch) {
    const size_t lds = pruned_lds_bytes<LP>();
    if (lds > 48 * 1024) {
        const hipError_t e = hipFuncSetAttribute((const void*)profile_pruned_kernel<LONG_NT, LP>,
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    for (int r0 = 0; r0 < rows; r0 += (int)rows_per_launch) {
        const int nr = rows - r0 < (int)rows_per_launch ? rows - r0 : (int)rows_per_launch;
        G.row0 = r0;
        hipLaunchKernelGGL((profile_pruned_kernel<LONG_NT, LP>), dim3(nr), dim3(LONG_NT), lds, stream, G);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
    }
    return (int)hipSuccess;
}

bool pruned_lp_compiled(int LP) {
    return LP == 1000 || LP == 1024 || LP == 1250 || LP == 1500 || LP == 2000 || LP == 2048 || LP == 2500;
}

int launch_pruned(hipStream_t stream, int LP, PrunedArgs G, int rows, size_t rows_per_launch) {
    switch (LP) {
        case 1000: return launch_pruned_lp<1000>(stream, G, rows, rows_per_launch);
        case 1024: return launch_pruned_lp<1024>(stream, G, rows, rows_per_launch);
        case 1250: return launch_pruned_lp<1250>(stream, G, rows, rows_per_launch);
        case 1500: return launch_pruned_lp<1500>(stream, G, rows, rows_per_launch);
        case 2000: return launch_pruned_lp<2000>(stream, G, rows, rows_per_launch);
        case 2048: return launch_pruned_lp<2048>(stream, G, rows, rows_per_launch);
        case 2500: return launch_pruned_lp<2500>(stream, G, rows, rows_per_launch);
    }
    return (int)hipErrorInvalidValue;
}

int launch_profile_support(hipStream_t stream, int rows, int nxs, const double* xs, const double* cmax, const double* rss,
                           const double* zs, int nm, const double* kts, const double* ks, int nk, int* d_out) {
    hipLaunchKernelGGL(profile_support_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, stream, rows, nxs, xs,
                       cmax, rss, zs, nm, kts, ks, nk, d_out);
    return (int)hipGetLastError();
}

template <int LB>
static int launch_band_lb(hipStream_t stream, PrunedArgs G, int rows, int jnmax) {
    const int M = G.M;
    double2* tb = reinterpret_cast<double2*>(G.u);
    hipLaunchKernelGGL(band_tables_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, stream, 2 * M, G.R, LB, G.F.xs,
                       G.F.logx, tb, tb + M, tb + 2 * (size_t)M);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    const size_t lds = (size_t)HMG_BAND_NBUF * LB * 16 + (size_t)(LB / 2 + 2) * 8 + 32 * sizeof(double);
    if (2 * jnmax + 1 <= FUSED_NT)
        hipLaunchKernelGGL((profile_band_kernel<FUSED_NT, LB, 1>), dim3(rows), dim3(FUSED_NT), lds, stream, G);
    else
        hipLaunchKernelGGL((profile_band_kernel<FUSED_NT, LB, (LB + FUSED_NT - 1) / FUSED_NT>), dim3(rows), dim3(FUSED_NT), lds,
                           stream, G);
    return (int)hipGetLastError();
}
bool band_lb_compiled(int LB) { return LB == 1000 || LB == 1024 || LB == 1250; }
int launch_band(hipStream_t stream, int LB, PrunedArgs G, int rows, int jnmax) {
    switch (LB) {
        case 1000: return launch_band_lb<1000>(stream, G, rows, jnmax);
        case 1024: return launch_band_lb<1024>(stream, G, rows, jnmax);
        case 1250: return launch_band_lb<1250>(stream, G, rows, jnmax);
    }
    return (int)hipErrorInvalidValue;
}

}  // namespace hmg

