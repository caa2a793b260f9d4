// K3 - analytic NFW u(k|m,z) through Si/Ci and its small-argument series (hmvec/hmvec.py:346-353).
// Part of the ONE translation unit hmgrid.hip (included there in this order; not a stand-alone header).
#pragma once

namespace hmg {

// ---------------------------------------------------------------- K3: analytic NFW (A6)
// fp64-VALU bound (two Si/Ci rational evaluations + two sincos per 8 bytes written), so the
// kernel is organised to minimise instructions, not bytes: one block per (z,m) row so the
// row constants (c, r_s, 1/m_c: a log and two divisions) are computed once per thread
// instead of once per point; sin(c x) comes from the angle-difference identity on the two
// sincos the Si/Ci asymptotics need anyway; k is the fast axis -> coalesced 8 B stores.
// Small-argument series of the NFW transform, one coefficient row per (z,m):
//   u(k) = (1/m_c) int_0^c [sin(x t)/(x t)] t/(1+t)^2 dt = sum_n a_n x^(2n),
//   a_n = (-1)^n J_(2n+1)(c) / ((2n+1)! m_c),   J_p(c) = int_0^c t^p/(1+t)^2 dt,
//   J_0 = c/(1+c), J_1 = m_c,  J_p = c^(p-1)/(p-1) - 2 J_(p-1) - J_(p-2).
// With NFW_NS = 16 terms the series is exact to 2 ulp for (1+c) x <= 4 and c >= 0.5 (checked
// against 50-digit arithmetic for c in [0.5, 60]); it replaces two Si/Ci rational evaluations
// and two sincos by 16 FMAs on about 2/3 of a typical grid, and it does not suffer the
// cancellation of the closed form at small x.  a[row][0] = 0 flags "do not use" (c < 0.5).
constexpr int NFW_NS = 16;     // terms used for (1+c) x <= 4
constexpr int NFW_NS2 = 32;    // terms used for 4 < (1+c) x <= NFW_X2 (same coefficient row, first 16 shared)
constexpr double NFW_X2 = 10.0;
constexpr int NFW_ROW = HMG_NFW_SERIES_STRIDE;   // doubles per (z,m) row: 32 series coefficients + row constants
constexpr int NFW_NT1 = 5;          // terms for (1+c) x <= NFW_XS1
constexpr double NFW_XS1 = 0.1;
constexpr int NFW_NT2 = 8;          // terms for (1+c) x <= NFW_XS2
constexpr double NFW_XS2 = 0.8;
// With 32 terms the series stays within 3e-15 (absolute, against 50-digit arithmetic, c in [0.5, 100])
// up to (1+c) x = 10: the band 4 < (1+c) x <= 10 - where x itself is still on the small-argument
// branch of Si/Ci, the most expensive case of the closed form - costs 32 FMAs instead.
__device__ __forceinline__ void nfw_series_row(double c, double* __restrict__ a) {
    constexpr double INVFACT[NFW_NS2] = {1.0, 0.16666666666666666, 0.008333333333333333, 0.0001984126984126984, 2.7557319223985893e-06, 2.505210838544172e-08, 1.6059043836821613e-10, 7.647163731819816e-13, 2.8114572543455206e-15, 8.22063524662433e-18, 1.9572941063391263e-20, 3.8681701706306835e-23, 6.446950284384474e-26, 9.183689863795546e-29, 1.1309962886447718e-31, 1.2161250415535181e-34, 1.151633562077195e-37, 9.67759295863189e-41, 7.265460179153071e-44, 4.902469756513544e-47, 2.9893108271424046e-50, 1.6552108677421951e-53, 8.359650847182804e-57, 3.866628513960594e-60, 1.643974708316579e-63, 6.446959640457174e-67, 2.3392451525606576e-70, 7.876246304918039e-74, 2.4674957095607893e-77, 7.210682961895936e-81, 1.9701319568021682e-84, 5.043860616493007e-88};
    const double opc = 1.0 + c;
    const double mc = log(opc) - c / opc;
    const double inv_mc = 1.0 / mc;
    double jm2 = c / opc, jm1 = mc, cp = c;   // J_0, J_1, c^(p-1) for p = 2
    a[NFW_NS2 + 0] = log(opc);              // row constants of the closed forms, computed once per row
    a[NFW_NS2 + 1] = inv_mc;                //   here instead of once per thread of the row's workgroup
    a[NFW_NS2 + 2] = 1.0 / (opc * opc);
    a[NFW_NS2 + 3] = 0.0;
    a[0] = (c >= 0.5) ? 1.0 : 0.0;
    // unrolled: 1/(p-1) becomes a compile-time factor (a division here is ~15 dependent instructions in a
    // 62-step chain); coefficients stay within 5e-15 of 80-digit arithmetic for c in [0.5, 100]
#pragma unroll
    for (int p = 2; p < 2 * NFW_NS2; ++p) {
        const double jp = cp * (1.0 / (double)(p - 1)) - 2.0 * jm1 - jm2;
        cp *= c;
        if (p & 1) {
            const int n = (p - 1) >> 1;
            a[n] = ((n & 1) ? -jp : jp) * INVFACT[n] * inv_mc;
        }
        jm2 = jm1;
        jm1 = jp;
    }
}
__global__ void nfw_series_kernel(int rows, const double* __restrict__ cs, double* __restrict__ acoef) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    nfw_series_row(cs[row], acoef + (size_t)row * NFW_ROW);
}

// Order in which the row workgroups of a launch take the masses of a redshift: heaviest first.  Workgroups are
// dispatched in index order and the rows of the heavy end of the mass grid are the expensive ones in both tensor
// producers (NFW: most of their wavenumbers are on the Si/Ci branch; Battaglia: many FFT modes are reachable), so
// ascending order leaves the most expensive rows for the tail of the launch.  Measured on MI355X (rows group +
// profile group): 58.9 -> 55.3 us on a 4-redshift slab, 97.9 -> 96.1 at nz = 8, +-0 at nz = 32; folding the mass
// axis (light half ascending, heavy half descending) and mass-major order over all redshifts were no better.
__device__ __forceinline__ int row_order(int r, int nm) {
    const int z = r / nm, i = r - z * nm;
    return z * nm + (nm - 1 - i);
}

// ktile = k values per workgroup (a multiple of the block size)
// 46 VGPRs, no scratch (with machine LICM off: see the Makefile); the bound only keeps it under 64.
#ifndef HMG_NFW_OCC
#define HMG_NFW_OCC 8
#endif
struct NfwArgs {
    const SiciTable* T;
    const double* acoef;
    int ktile, nm, nk;
    const double *cs, *rss, *zs, *ks;
    double* uk;
};
// blk: index of the (row, k tile) this workgroup owns; nthr: threads that share it (the workgroup size).
// The pointers must reach this function as __restrict__ KERNEL PARAMETERS (not as fields of a by-value
// struct): only then can hipcc prove that the stores to uk do not clobber the row constants, series
// coefficients and Si/Ci tables and fetch those with scalar loads - as struct fields they became 249 vector
// loads per thread and the kernel ran 2.6x slower (0.15 -> 0.40 ms at Config 3).
__device__ __forceinline__ void nfw_rows(const SiciTable* __restrict__ T, const double* __restrict__ acoef, int ktile,
                                         int nm, int nk, const double* __restrict__ cs,
                                         const double* __restrict__ rss, const double* __restrict__ zs,
                                         const double* __restrict__ ks, double* __restrict__ uk, int blk, int nthr,
                                         int tid) {
    // one (z,m) row per workgroup, the whole k axis in one tile: measured against two rows per workgroup
    // (+4 %), half tiles (+40 %) and 128 threads per row (+-0): a workgroup's fixed cost is the latency of
    // its scalar loads (row constants, series coefficients), not instructions
    const int ktiles = (nk + ktile - 1) / ktile;
    const int brow = blk / ktiles;
    const int row = row_order(brow, nm);  // z*nm + m
    const int k_lo = (blk - brow * ktiles) * ktile;
    const int k_hi = min(nk, k_lo + ktile);
    const int z = row / nm;
    const double c = cs[row];
    const double rs = rss[row];
    const double z1 = 1.0 + zs[z];
    const double opc = 1.0 + c;
    // small-argument series coefficients and closed-form constants of this row: wave-uniform -> SGPRs
    const double* __restrict__ a = acoef + (size_t)row * NFW_ROW;
    const double ln_opc = a[NFW_NS2 + 0], inv_mc = a[NFW_NS2 + 1], inv_opc2 = a[NFW_NS2 + 2];
    const bool use_series = (a[0] != 0.0);
    double* __restrict__ dst = uk + (size_t)row * nk;
    for (int k = k_lo + tid; k < k_hi; k += nthr) {
        const double x = ks[k] * rs * z1;
        const double xc = opc * x;
        if (use_series && xc <= 4.0) {
            // The series alternates and its n-th term is below (xc)^(2n) / (2n (2n+1)! m_c): 5 terms are
            // exact to 1e-18 for (1+c) x <= 0.1, 8 terms to 2e-17 for <= 0.8 - about 60 % of a typical
            // grid (k starts four decades below the halo scale) takes one of the two short forms.
            const double z = x * x;
            double u;
            if (xc <= NFW_XS1) {
                u = fma_svs(a[NFW_NT1 - 1], z, a[NFW_NT1 - 2]);
#pragma unroll
                for (int n = NFW_NT1 - 3; n >= 0; --n) u = fma_vvs(u, z, a[n]);
            } else if (xc <= NFW_XS2) {
                u = fma_svs(a[NFW_NT2 - 1], z, a[NFW_NT2 - 2]);
#pragma unroll
                for (int n = NFW_NT2 - 3; n >= 0; --n) u = fma_vvs(u, z, a[n]);
            } else {
                u = fma_svs(a[NFW_NS - 1], z, a[NFW_NS - 2]);
#pragma unroll
                for (int n = NFW_NS - 3; n >= 0; --n) u = fma_vvs(u, z, a[n]);
            }
            __builtin_nontemporal_store(u, &dst[k]);
            continue;
        }
        if (use_series && xc <= NFW_X2) {
            const double z = x * x;
            double u = fma_svs(a[NFW_NS2 - 1], z, a[NFW_NS2 - 2]);
#pragma unroll
            for (int n = NFW_NS2 - 3; n >= 0; --n) u = fma_vvs(u, z, a[n]);
            __builtin_nontemporal_store(u, &dst[k]);
            continue;
        }
        if (x > 4.0 && xc < 1.0e9) {
            // Both arguments on the auxiliary-function branch, Si = pi/2 - f cos - g sin,
            // Ci = f sin - g cos.  Substituting into the NFW formula the terms in f(x) cancel and
            // the rest collapses, exactly, to
            //     u m_c = g(x) + f(xc) sin(c x) - g(xc) cos(c x) - sin(c x)/xc :
            // one sincos (of c x, the argument the reference itself uses for sin(c x)) instead of
            // two, three rationals instead of four, and none of the pi/2-sized cancellations.
            const double zx = rcp_fast(x * x), zc = zx * inv_opc2;
            double f1, g1, f2, g2, sd, cd;
            sici_aux<false>(T, x, zx, f1, g1);
            sici_aux<true>(T, xc, zc, f2, g2);
            sincos_fast(c * x, sd, cd);
            __builtin_nontemporal_store((g1 + (f2 - xc * zc) * sd - g2 * cd) * inv_mc, &dst[k]);
            continue;
        }
        if (x <= 4.0 && xc > 8.0) {
            // Mixed band (3.5 % of a typical grid, beyond the reach of the series): x on the rational
            // branch of Si/Ci, (1+c)x on the auxiliary-function branch.  Substituting
            // Si(xc) = pi/2 - f cos xc - g sin xc, Ci(xc) = f sin xc - g cos xc and xc - x = c x,
            //     u m_c = (pi/2) sin x + f(xc) sin(cx) - g(xc) cos(cx) - sin(cx)/xc - sin x Si(x) - cos x Ci(x):
            // sincos of x and c x (the reference's own arguments) instead of x and xc, two rational
            // pairs instead of four, one short logarithm.
            double s1, c1, sd, cd, f2, g2;
            sincos_fast(x, s1, c1);
            sincos_fast(c * x, sd, cd);
            const double x2 = x * x;
            const double zc = rcp_fast(x2) * inv_opc2;                   // 1/xc^2
            const double sden = horner_s<6>(x2, T->SD), cden = horner_s<6>(x2, T->CD);
            const double r = rcp_fast(sden * cden);
            const double si = x * horner_s<6>(x2, T->SN) * (cden * r);
            const double ci = (EULER_GAMMA + log_fast(x)) + x2 * horner_s<6>(x2, T->CN) * (sden * r);
            sici_aux<true>(T, xc, zc, f2, g2);
            __builtin_nontemporal_store((HALF_PI * s1 + (f2 - xc * zc) * sd - g2 * cd - s1 * si - c1 * ci) * inv_mc,
                                        &dst[k]);
            continue;
        }
        // everything else (rows with c < 0.5, arguments beyond 1e9): the closed form as the reference writes it
        double s1, c1, s2, c2;
        if (xc < 1.0e9) {
            sincos_fast(x, s1, c1);
            sincos_fast(xc, s2, c2);
        } else {  // outside the Cody-Waite range: library reduction
            sincos(x, &s1, &c1);
            sincos(xc, &s2, &c2);
        }
        const double zx = rcp_fast(x * x);   // 1/x^2
        const double zc = zx * inv_opc2;     // 1/xc^2
        double si1, ci1, si2, ci2;
        bool sm1, sm2;
        sici_fast(T, x, s1, c1, zx, si1, ci1, sm1);
        sici_fast(T, xc, s2, c2, zc, si2, ci2, sm2);
        // Ci((1+c)x) - Ci(x): x <= xc, so the cases are (small,small), (small,large), (large,large)
        double dci = ci2 - ci1;
        if (sm1) dci += sm2 ? ln_opc : -(EULER_GAMMA + log(x));
        const double scx = s2 * c1 - c2 * s1;  // sin(c x) = sin((1+c)x - x)
        // sin(cx)/((1+c)x) = scx * xc / xc^2
        __builtin_nontemporal_store((s1 * (si2 - si1) - scx * (xc * zc) + c1 * dci) * inv_mc, &dst[k]);
    }
}
__global__ __launch_bounds__(256, HMG_NFW_OCC) void nfw_kernel(const SiciTable* __restrict__ T,
                                                  const double* __restrict__ acoef, int ktile, int nm, int nk,
                                                  const double* __restrict__ cs, const double* __restrict__ rss,
                                                  const double* __restrict__ zs, const double* __restrict__ ks,
                                                  double* __restrict__ uk) {
    nfw_rows(T, acoef, ktile, nm, nk, cs, rss, zs, ks, uk, blockIdx.x, blockDim.x, threadIdx.x);
}

}  // namespace hmg
