// N1 - Limber projection of P(z,k) (hmvec/cosmology.py:867-904).
// Part of the ONE translation unit hmgrid.hip (included there in this order; not a stand-alone header).
#pragma once

namespace hmg {

// ---------------------------------------------------------------- N1: Limber integral
// C_ell = int dz pref(z) P(z, k=(ell+1/2)/chi(z)) with P bilinear in (z,k) on the model grid,
// clamped to the grid box (hmvec/cosmology.py:867-904; the degree-1 fitpack spline the
// reference evaluates clamps its arguments).  One thread per multipole, loop over the nz_w
// window redshifts; wz = trapezoid weights over those redshifts (or {1} for a delta window).
// One thread per (multipole, window redshift) term - the bracket searches and the four spectrum loads of the nells x ngz
// terms are independent, and a thread per multipole walking its redshifts one after the other (rounds 1-4) spent 40 us
// in 32 x 12 dependent loads for 2000 multipoles - and one lane per multipole adds the terms up in ascending g, 32 at a
// time through LDS: the same order of sums as the sequential loop, so the same bits.
constexpr int LIMBER_G = 32, LIMBER_E = 8;       // window redshifts per round x multipoles per workgroup (256 threads)
__global__ __launch_bounds__(LIMBER_G * LIMBER_E) void limber_kernel(
    int nells, const double* __restrict__ ells, int nz, int nk, const double* __restrict__ zs,
    const double* __restrict__ ks, const double* __restrict__ P, const double* __restrict__ P2, int ngz,
    const double* __restrict__ gzs, const double* __restrict__ pref, const double* __restrict__ chis,
    const double* __restrict__ wz, double* __restrict__ out) {
#pragma clang fp contract(off)
    __shared__ double term[LIMBER_E][LIMBER_G];
    const int el = threadIdx.x / LIMBER_G, gs = threadIdx.x - el * LIMBER_G;
    const int e = blockIdx.x * LIMBER_E + el;
    const double ell = e < nells ? ells[e] : 0.0;
    double acc = 0.0;
    for (int g0 = 0; g0 < ngz; g0 += LIMBER_G) {
        const int g = g0 + gs;
        double t = 0.0;
        if (e < nells && g < ngz) {
            double k = (ell + 0.5) / chis[g];
            k = fmin(fmax(k, ks[0]), ks[nk - 1]);
            int lo = 0, hi = nk - 1;            // largest i with ks[i] <= k, capped at nk-2
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (ks[mid] <= k) lo = mid; else hi = mid;
            }
            const int i = lo;
            const double tx = (k - ks[i]) / (ks[i + 1] - ks[i]);
            double val;
            // P2 (optional) is added on the fly: C_ell of P_1h + P_2h without materialising the sum
            auto at = [&](size_t o) { return P2 ? P[o] + P2[o] : P[o]; };
            if (nz == 1) {
                val = (1.0 - tx) * at(i) + tx * at(i + 1);
            } else {
                double z = fmin(fmax(gzs[g], zs[0]), zs[nz - 1]);
                int jl = 0, jh = nz - 1;
                while (jh - jl > 1) {
                    const int mid = (jl + jh) >> 1;
                    if (zs[mid] <= z) jl = mid; else jh = mid;
                }
                const int j = jl;
                const double ty = (z - zs[j]) / (zs[j + 1] - zs[j]);
                const size_t r0 = (size_t)j * nk + i, r1 = r0 + nk;
                val = (1.0 - tx) * (1.0 - ty) * at(r0) + tx * (1.0 - ty) * at(r0 + 1) +
                      (1.0 - tx) * ty * at(r1) + tx * ty * at(r1 + 1);
            }
            t = wz[g] * (val * pref[g]);
        }
        term[el][gs] = t;
        __syncthreads();
        if (gs == 0) {
            const int n = ngz - g0 < LIMBER_G ? ngz - g0 : LIMBER_G;
            for (int q = 0; q < n; ++q) acc += term[el][q];
        }
        __syncthreads();
    }
    if (gs == 0 && e < nells) out[e] = acc;
}

}  // namespace hmg
