// A8/X1 - Battaglia row parameters, the halo stage and the constructor stage (hmvec/hmvec.py:800-860,906-927).
// Part of the ONE translation unit hmgrid.hip (included there in this order; not a stand-alone header).
#pragma once

namespace hmg {

// ---------------------------------------------------------------- A8/X1: row parameters
struct RowFit { double f[9]; };
struct RowOut {
    double *amp, *xc, *alpha, *expo, *cmax, *rscale, *post;
    // optional row scalars of the profile transform that will read these rows (hmg_rows_part, ABI 8)
    double* rowsc;
    const double *ks, *kts;
    int nk, M;
};
// The output-side scalars of one profile row (hmvec/fft.py:96-107), the SAME expressions profile_fused_row evaluates
// when it has to work them out itself: isc = 1/(r (1+z)) (kout_j = kt_j isc), k_lo = kt_1 isc, k_hi = kt_M isc, 1/k_lo,
// 1/kt_1, jn = modes the target grid can reach, nleft = targets below k_lo (ks ascending: a bisection here, a 64-way
// search there - the same count).
__device__ __forceinline__ void rowscal_store(const RowOut& O, int idx, double rscale, double z1) {
    if (!O.rowsc) return;
    const double isc0 = 1.0 / (rscale * z1);
    const double kt1 = O.kts[1];
    const double klo0 = kt1 * isc0;
    const double idk0 = 1.0 / klo0;
    int jn0 = O.M;
    const double tmax = O.ks[O.nk - 1] * idk0;
    if (tmax < (double)(O.M - 4)) jn0 = (int)tmax + 3;
    int lo = 0, hi = O.nk;                     // ks[i] < k_lo for i < lo, ks[i] >= k_lo for i >= hi
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (O.ks[mid] < klo0) lo = mid + 1; else hi = mid;
    }
    double* __restrict__ r = O.rowsc + (size_t)idx * HMG_ROWSC_STRIDE;
    r[0] = isc0; r[1] = klo0; r[2] = O.kts[O.M] * isc0; r[3] = idk0; r[4] = 1.0 / kt1;
    r[5] = __hiloint2double(jn0, lo);
    r[6] = 0.0; r[7] = 0.0;
}
__device__ __forceinline__ void rowparams_body(int kind, int idx, double M, double R, double rv, double z1,
                                               double rhoc, double hz, const RowFit& F, double gamma,
                                               double alpha_const, double pref, double post_pref,
                                               const RowOut& O) {
    // A0 (M/1e14)^am (1+z)^az for the three fits: the two logarithms are shared and each power
    // product is one exp2 (|exponent| < 10, so the result is within a few ulp of pow*pow)
    const double lm = log2(M / 1.0e14), lz = log2(z1);
    const double X0 = F.f[0] * exp2(F.f[1] * lm + F.f[2] * lz);
    const double X1 = F.f[3] * exp2(F.f[4] * lm + F.f[5] * lz);
    const double X2 = F.f[6] * exp2(F.f[7] * lm + F.f[8] * lz);
    if (kind == HMG_PROF_BATTAGLIA_GAS) {
        // (Ob/Om) rho_c rho0 x^g (1+x^alpha)^(-(beta+g)/alpha),  x = r/(R200c/2)
        O.amp[idx] = pref * rhoc * X0;
        O.xc[idx] = 1.0;
        O.alpha[idx] = X1;
        O.expo[idx] = (X2 + gamma) / X1;
        const double rg = R / 2.0;
        O.rscale[idx] = rg;
        O.cmax[idx] = rv / rg;
        if (O.post) O.post[idx] = 1.0;
        rowscal_store(O, idx, rg, z1);
    } else {
        // eFrac (Ob/Om) 200 M G rho_c / (2 R200) P0 (x/xc)^g (1+(x/xc)^alpha)^(-beta),  x = r/R200c
        O.amp[idx] = pref * M * rhoc / (2.0 * R) * X0;
        O.xc[idx] = X1;
        O.alpha[idx] = alpha_const;
        O.expo[idx] = X2;
        O.rscale[idx] = R;
        O.cmax[idx] = rv / R;
        if (O.post) O.post[idx] = post_pref * ((R * R * R) * ((z1 * z1) / hz));
        rowscal_store(O, idx, R, z1);
    }
}

__global__ void rowparams_kernel(int kind, int nz, int nm, const double* __restrict__ m200,
                                 const double* __restrict__ r200, const double* __restrict__ rvir,
                                 const double* __restrict__ zs, const double* __restrict__ rhoc,
                                 const double* __restrict__ hz, RowFit F, double gamma,
                                 double alpha_const, double pref, double post_pref, RowOut O) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nz * nm) return;
    const int z = idx / nm;
    rowparams_body(kind, idx, m200[idx], r200[idx], rvir[idx], 1.0 + zs[z], rhoc[z], hz ? hz[z] : 1.0, F,
                   gamma, alpha_const, pref, post_pref, O);
}

// mass conversion + row parameters in one launch (one kernel boundary fewer per profile)
__global__ void rows_from_mvir_kernel(int kind, int nz, int nm, const double* __restrict__ ms,
                                      const double* __restrict__ cs, const double* __restrict__ rvir,
                                      const double* __restrict__ zs, const double* __restrict__ d1,
                                      double delta2, const double* __restrict__ rhoc,
                                      const double* __restrict__ hz, RowFit F, double gamma,
                                      double alpha_const, double pref, double post_pref,
                                      double* __restrict__ m2, double* __restrict__ r2, RowOut O) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nz * nm) return;
    const int z = idx / nm, m = idx - z * nm;
    const double M2 = mdelta_solve(ms[m], cs[idx], d1[z] / (delta2 * rhoc[z]));
    const double R2 = cbrt(3.0 * M2 / 4.0 / M_PI / delta2 / rhoc[z]);
    m2[idx] = M2;
    r2[idx] = R2;
    rowparams_body(kind, idx, M2, R2, rvir[idx], 1.0 + zs[z], rhoc[z], hz ? hz[z] : 1.0, F, gamma,
                   alpha_const, pref, post_pref, O);
}

// c, rvir, rs + the NFW series row + the mass conversion of one (z,m) per thread: the three
// per-(z,m) launches that precede the profile kernels of a pass, in one (hmg_halo_stage).
__device__ __forceinline__ void nfw_series_row(double c, double* __restrict__ a);
struct HaloStageArgs {
    int nz, nm;
    const double *ms, *zs, *delta, *rho;
    double A, alpha, beta, h;
    double *cs, *rv, *rs, *series /*[nz*nm][NFW_ROW] or null*/;
    const double* d1;
    double delta2;
    const double* rho2;
    double *m2, *r2 /* both or neither */;
};
// (rv_out, m2_out, r2_out: the values just stored, for a caller that goes on to the Battaglia row parameters)
__device__ __forceinline__ void halo_stage_point(const HaloStageArgs& H, int idx, double* rv_out = nullptr,
                                                 double* m2_out = nullptr, double* r2_out = nullptr) {
    const int z = idx / H.nm, m = idx - z * H.nm;
    const double mm = H.ms[m];
    const double c = H.A * pow(H.h * mm / 2.0e12, H.alpha) * pow(1.0 + H.zs[z], H.beta);
    const double r = pow(3.0 * mm / 4.0 / M_PI / H.delta[z] / H.rho[z], 1.0 / 3.0);
    H.cs[idx] = c;
    H.rv[idx] = r;
    H.rs[idx] = r / c;
    if (rv_out) *rv_out = r;
    if (H.m2) {
        const double M2 = mdelta_solve(mm, c, H.d1[z] / (H.delta2 * H.rho2[z]));
        const double R2 = cbrt(3.0 * M2 / 4.0 / M_PI / H.delta2 / H.rho2[z]);
        H.m2[idx] = M2;
        H.r2[idx] = R2;
        if (m2_out) { *m2_out = M2; *r2_out = R2; }
    }
    if (H.series) nfw_series_row(c, H.series + (size_t)idx * NFW_ROW);
}
__global__ void halo_stage_kernel(HaloStageArgs H) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < H.nz * H.nm) halo_stage_point(H, idx);
}

// Everything the constructor computes per (z,m), in ONE launch behind the sigma^2 contraction: plane 0 of
// the grid is sigma2_massfn_kernel's work (needs the contraction's partial sums), plane 1 the halo stage
// (needs only m and z).  The two do not depend on each other, so the halo stage's workgroups fill the
// compute units the 8 x nz mass-function workgroups leave idle instead of waiting for a launch of their own.
// grid (ceil(nm/64), nz, 2), 512 threads; plane 1 uses the first wavefront of each workgroup.
__global__ __launch_bounds__(512) void ctor_stage_kernel(SigmaMassFnArgs A, HaloStageArgs H) {
    __shared__ double red[4][66];
    __shared__ double sig[66];
    if (blockIdx.z == 0) {
        sigma2_massfn_block(A, blockIdx.y, blockIdx.x * 64, red, sig);
    } else if (threadIdx.x < 64) {
        const int m = blockIdx.x * 64 + threadIdx.x;
        if (m < H.nm) halo_stage_point(H, blockIdx.y * H.nm + m);
    }
}

}  // namespace hmg
