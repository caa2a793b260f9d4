// Device pieces shared by the translation units that hold radial-profile row kernels (hmgrid.hip: the one-row-in-LDS
// kernels; longgrid.hip: the long-grid kernels): wavefront sums, the profile family, the description of a launch.
#pragma once
#include <hip/hip_runtime.h>

#include "fastmath.hpp"
#include "ldsfft.hpp"

#ifndef HMG_FUSED_NT
#define HMG_FUSED_NT 512
#endif

namespace hmg {

constexpr int FUSED_NT = HMG_FUSED_NT;   // threads per row workgroup of the fused profile kernels

constexpr int WAVE = 64;

// Sum over the 64 lanes of a wavefront, returned in EVERY lane.  Cross-lane moves are DPP modifiers
// (register-to-register, a few cycles) instead of __shfl (ds_bpermute: an LDS-pipeline round trip per
// step, twelve dependent ones per double).  Fixed combination tree: pairs, quads, half rows, rows of 16
// (quad_perm / row_half_mirror / row_mirror leave every lane of a row with the row's sum), then
// row 0 -> 1 and 2 -> 3 (row_bcast:15), rows 0+1 -> 2,3 (row_bcast:31); lane 63 holds the total.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_move<0xB1, 0xf>(v);      // quad_perm [1,0,3,2]
    v += dpp_move<0x4E, 0xf>(v);      // quad_perm [2,3,0,1]
    v += dpp_move<0x141, 0xf>(v);     // row_half_mirror
    v += dpp_move<0x140, 0xf>(v);     // row_mirror
    v += dpp_move<0x142, 0xa>(v);     // row_bcast:15 into rows 1 and 3 (other rows receive 0)
    v += dpp_move<0x143, 0xc>(v);     // row_bcast:31 into rows 2 and 3
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63),
                            __builtin_amdgcn_readlane(__double2loint(v), 63));
}

// One launch of radial-profile rows (hmvec/fft.py:56-115: integrand, transform, interpolation): what every row kernel
// is given.
struct FusedArgs {
    FftPlanDev plan;
    int nxs, nm, nk, do_norm;
    const double* xs;
    const cplx* twM;     // per-pass twiddle table of the plan (ldsfft.hpp: pass_tw_table)
    const UnpackTw* twN; // (cos, sin)(2 pi j / nxs), 1/j, 1/(M-j) for j <= M/2
    const double* kts;
    const double *amp, *xc, *alpha, *expo;
    double amp_c, xc_c, alpha_c, expo_c, gamma, step;
    const double *cmax, *rss, *zs, *ks, *post;
    double* out;
    int* nconst;               // optional constant-prefix hint per row
    double* cconst;
    const double* logx;        // ln xs[n], shared by every row (nullptr: evaluated per sample)
    // TAB builds of the row kernels (generic_profile_fft with a user's callable, hmvec/fft.py:56-94): the profile comes
    // from a table instead of the family - rho_tab[n] shared by every row, or rho_tab[row nxs + n]
    const double* rho_tab;
    int rho_shared;
    // optional: the output-side scalars of every row ([rows][HMG_ROWSC_STRIDE], include/hmgrid.h: hmg_rows_part), left by
    // the launch that computed the rows' length scales; nullptr: one wavefront of the row's workgroup works them out
    const double* rowsc;
};

// amp * t^gamma * (1 + t^alpha)^(-expo), t = x/xc, through exp/log (one log shared by the two
// powers of t) with the short fp64 log/exp/log1p of fastmath.hpp (< 2 ulp each, host-tested):
// ~100 VALU ops per sample instead of ~190 with the device library's and ~700 with three pow().
__device__ __forceinline__ double gnfw_rho_fast(double lt /* ln(x/xc) */, double A, double AL, double EX,
                                                double gamma) {
    // (|exponents| stay far below 1e9: no clamp; the logarithm's absolute error is what the outer exp sees)
    const double ta = exp_fast<false>(fmin(AL * lt, 700.0));
    return A * exp_fast<false>(gamma * lt - EX * log1p_abs(ta));
}

// The member with alpha == 1 (the Battaglia pressure profile, hmvec/hmvec.py:906-927 with battaglia_pres_alpha = 1):
// t^alpha is t = x / xc itself - one exponential per sample instead of two.
__device__ __forceinline__ double gnfw_rho_alpha1(double lt /* ln(x/xc) */, double t /* x/xc */, double A, double EX,
                                                  double gamma) {
    return A * exp_fast<false>(gamma * lt - EX * log1p_abs(t));
}

}  // namespace hmg
