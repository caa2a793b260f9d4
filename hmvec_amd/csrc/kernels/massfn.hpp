// K2 - Sheth-Tormen / Tinker mass function and halo bias from sigma^2 (hmvec/hmvec.py:133-185, tinker.py:26-67).
// Part of the ONE translation unit hmgrid.hip (included there in this order; not a stand-alone header).
#pragma once

namespace hmg {

// ---------------------------------------------------------------- K2: mass function (A3/A4)
struct MassFnDev {
    int mode;
    double deltac, A, a, p, rho_m0;
    int uniform;
    double step;
};

__device__ __forceinline__ double tinker10_bias(double nu) {
    const double dc = 1.686;
    const double y = log10(200.0);
    const double ey = exp(-pow(4.0 / y, 4.0));
    const double A = 1.0 + 0.24 * y * ey;
    const double a = 0.44 * y - 0.88;
    const double C = 0.019 + 0.107 * y + 0.19 * ey;
    const double nua = pow(nu, a);
    return 1.0 - A * nua / (nua + pow(dc, a)) + 0.183 * pow(nu, 1.5) + C * pow(nu, 2.4);
}

// n(z,m) and b(z,m) of one grid point; S(i) returns sigma2[z][i] (from global memory or from LDS)
template <class SigmaAt>
__device__ __forceinline__ void massfn_point(const MassFnDev& P, int z, int m, int nm, SigmaAt S,
                                             const double* __restrict__ ms, const double* __restrict__ lnm,
                                             const double* __restrict__ tz, double& n_out, double& b_out) {
    const double sig2 = S(m);
    const double dc = P.deltac;
    double f, b;
    if (P.mode == HMG_MF_SHETH_TORMEN) {
        const double sig = sqrt(sig2);
        f = P.A * sqrt(2.0 * P.a / M_PI) * (1.0 + pow(sig2 / P.a / (dc * dc), P.p)) * (dc / sig) *
            exp(-P.a * (dc * dc) / 2.0 / sig2);
        const double t = P.a * (dc * dc) / sig2;
        b = 1.0 + (1.0 / dc) * (t - 1.0) + (2.0 * P.p / dc) / (1.0 + pow(t, P.p));
    } else {
        const double nu = dc / sqrt(sig2);
        const double al = tz[z * 5 + 0], be = tz[z * 5 + 1], ph = tz[z * 5 + 2], et = tz[z * 5 + 3],
                     ga = tz[z * 5 + 4];
        const double fnu = al * ((1.0 + pow(be * nu, -2.0 * ph)) * pow(nu, 2.0 * et) *
                                 exp(-ga * (nu * nu) / 2.0));
        f = nu * fnu;
        b = tinker10_bias(nu);
    }
    // d ln(1/sigma) / d ln m with numpy.gradient's stencils (second order interior,
    // one-sided first order at the ends; uniform-grid shortcut when numpy would take it)
    auto L = [&](int i) { return -0.5 * log(S(i)); };
    double g;
    if (nm == 1) {
        g = 0.0;
    } else if (m == 0) {
        g = (L(1) - L(0)) / (P.uniform ? P.step : (lnm[1] - lnm[0]));
    } else if (m == nm - 1) {
        g = (L(nm - 1) - L(nm - 2)) / (P.uniform ? P.step : (lnm[nm - 1] - lnm[nm - 2]));
    } else if (P.uniform) {
        g = (L(m + 1) - L(m - 1)) / (2.0 * P.step);
    } else {
        const double d1 = lnm[m] - lnm[m - 1], d2 = lnm[m + 1] - lnm[m];
        const double ca = -d2 / (d1 * (d1 + d2)), cb = (d2 - d1) / (d1 * d2), cc = d1 / (d2 * (d1 + d2));
        g = ca * L(m - 1) + cb * L(m) + cc * L(m + 1);
    }
    const double mm = ms[m];
    n_out = P.rho_m0 * f * g / (mm * mm);
    b_out = b;
}

__global__ void massfn_kernel(int nz, int nm, MassFnDev P, const double* __restrict__ s2,
                              const double* __restrict__ ms, const double* __restrict__ lnm,
                              const double* __restrict__ tz, double* __restrict__ nzm,
                              double* __restrict__ bh) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nz * nm) return;
    const int z = idx / nm, m = idx - z * nm;
    const double* row = s2 + (size_t)z * nm;
    double n, b;
    massfn_point(P, z, m, nm, [&](int i) { return row[i]; }, ms, lnm, tz, n, b);
    nzm[idx] = n;
    bh[idx] = b;
}

// Second stage of sigma^2 (the ordered sum over the k' segments, exactly sigma2_combine_kernel's) and
// the mass function in ONE launch: a workgroup owns 64 consecutive masses of one redshift, sums the
// partials of those and of the two neighbours the gradient stencil reaches, keeps the 66 values in
// LDS, writes sigma2 and evaluates n(z,m), b(z,m) from LDS.  512 threads: wavefronts 0-3 take the four
// interleaved part groups of the 64 masses, two lanes of wavefronts 4-7 those of the two neighbours (so
// that no lane walks the parts twice).
struct SigmaMassFnArgs {
    int nz, nm, parts;
    MassFnDev P;
    const double *partial /*[parts][nz*nm]*/, *ms, *lnm, *tz;
    double *s2, *nzm, *bh;
};
__device__ __forceinline__ void sigma2_massfn_block(const SigmaMassFnArgs& A, int z, int m0, double (*red)[66],
                                                    double* sig) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nm = A.nm, n = A.nz * nm;
    // slot j <-> mass m0 - 1 + j (clamped to the row)
    const int j = w < 4 ? lane + 1 : (lane == 0 ? 0 : 65);
    if (w < 4 || lane < 2) {
        const int m = min(max(m0 - 1 + j, 0), nm - 1);
        red[w & 3][j] = sigma2_segment_sum(n, A.parts, A.partial, (size_t)z * nm + m, w & 3);
    }
    __syncthreads();
    if (threadIdx.x < 66) {
        const int jj = threadIdx.x;
        const double v = ((red[0][jj] + red[1][jj]) + red[2][jj]) + red[3][jj];
        sig[jj] = v;
        const int m = m0 - 1 + jj;
        if (jj >= 1 && jj <= 64 && m < nm) A.s2[(size_t)z * nm + m] = v;
    }
    __syncthreads();
    const int m = m0 + threadIdx.x;
    if (threadIdx.x < 64 && m < nm) {
        double nn, bb;
        massfn_point(A.P, z, m, nm, [&](int i) { return sig[i - m0 + 1]; }, A.ms, A.lnm, A.tz, nn, bb);
        A.nzm[(size_t)z * nm + m] = nn;
        A.bh[(size_t)z * nm + m] = bb;
    }
}
__global__ __launch_bounds__(512) void sigma2_massfn_kernel(SigmaMassFnArgs A) {
    __shared__ double red[4][66];
    __shared__ double sig[66];
    sigma2_massfn_block(A, blockIdx.y, blockIdx.x * 64, red, sig);
}
// The same stage for a 256-thread workgroup (the form a grouped launch uses beside the NFW rows): a tile is
// 62 masses plus its two stencil neighbours = 64 slots, one per lane, so that the four wavefronts take the
// four interleaved part groups of all 64 slots and nobody walks the parts twice.  Every sigma2[z][m] is summed
// exactly as above (group g = parts g, g+4, ... in order, then ((g0 + g1) + g2) + g3), so the results are the
// same bit for bit.  red: 4 x 64 doubles of LDS, sig: 64.
constexpr int MF_TILE = 62;
__device__ __forceinline__ void sigma2_massfn_tile(const SigmaMassFnArgs& A, int z, int tile, double* red, double* sig) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nm = A.nm, n = A.nz * nm, m0 = tile * MF_TILE;
    if (w < 4) {   // slot `lane` <-> mass m0 - 1 + lane (clamped to the row)
        const int m = min(max(m0 - 1 + lane, 0), nm - 1);
        red[w * 64 + lane] = sigma2_segment_sum(n, A.parts, A.partial, (size_t)z * nm + m, w);
    }
    __syncthreads();
    if (w == 0) {
        const double v = ((red[lane] + red[64 + lane]) + red[128 + lane]) + red[192 + lane];
        sig[lane] = v;
        const int m = m0 - 1 + lane;
        if (lane >= 1 && lane <= MF_TILE && m < nm) A.s2[(size_t)z * nm + m] = v;
    }
    __syncthreads();
    const int m = m0 + (int)threadIdx.x;
    if (threadIdx.x < MF_TILE && m < nm) {
        double nn, bb;
        massfn_point(A.P, z, m, nm, [&](int i) { return sig[i - m0 + 1]; }, A.ms, A.lnm, A.tz, nn, bb);
        A.nzm[(size_t)z * nm + m] = nn;
        A.bh[(size_t)z * nm + m] = bb;
    }
}

// The same stage for ALL masses of one redshift by ONE workgroup of NT threads: the form the per-z chain uses when
// sigma^2 -> n, b is its first link (tensor group: the chain then needs nothing from another workgroup of its launch, and
// a serial walk over the 62-mass tiles above would be the longest thing in it).  One mass per thread: the thread sums
// the four interleaved part groups of its mass itself (same groups, same order, ((g0 + g1) + g2) + g3: the bits of
// sigma2_combine_kernel), the values go to LDS, every thread evaluates n, b of its mass from there.  Mass grids longer
// than NT in chunks of NT - 2 masses with their two stencil neighbours.  sig: NT doubles of LDS.
template <int NT>
__device__ __forceinline__ void massfn_row(const SigmaMassFnArgs& A, int z, double* sig) {
    const int nm = A.nm, n = A.nz * nm, t = (int)threadIdx.x;
    const bool one = nm <= NT;
    const int stride = one ? NT : NT - 2;
    for (int first = 0; first < nm; first += stride) {
        const int base = one ? 0 : first - 1;                 // mass of slot 0
        const int m = base + t;                               // this thread's slot (clamped to the row for the sum)
        const size_t i = (size_t)z * nm + min(max(m, 0), nm - 1);
        const double g0 = sigma2_segment_sum(n, A.parts, A.partial, i, 0), g1 = sigma2_segment_sum(n, A.parts, A.partial, i, 1);
        const double g2 = sigma2_segment_sum(n, A.parts, A.partial, i, 2), g3 = sigma2_segment_sum(n, A.parts, A.partial, i, 3);
        const double v = ((g0 + g1) + g2) + g3;
        sig[t] = v;
        const bool mine = m < nm && (one || (t >= 1 && t <= NT - 2));
        if (mine) A.s2[i] = v;
        __syncthreads();
        if (mine) {
            double nn, bb;
            massfn_point(A.P, z, m, nm, [&](int k) { return sig[k - base]; }, A.ms, A.lnm, A.tz, nn, bb);
            A.nzm[i] = nn;
            A.bh[i] = bb;
        }
        __syncthreads();                                      // the next chunk overwrites sig
    }
}

}  // namespace hmg
