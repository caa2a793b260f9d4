// K7 - Behroozi SHMR, HOD occupation numbers, n_gal and b_g (hmvec/hmvec.py:357-466,634-731).
// Part of the ONE translation unit hmgrid.hip (included there in this order; not a stand-alone header).
#pragma once

namespace hmg {

// ---------------------------------------------------------------- K7: HOD (H1-H3)
// 10^y and x^p through exp2/log2 (one transcendental each instead of the ~6x longer generic
// pow); relative error <= ~|y| * 8e-16, far inside the 1e-9 gate on the HOD arrays.
__device__ __forceinline__ double pow10_fast(double y) { return exp2(y * 3.32192809488736234787); }
__device__ __forceinline__ double powr_fast(double x, double p) { return exp2(p * log2(x)); }

struct ShmrSet {
    double Ms0, Msa, M1, M1a, b0, ba, g0, ga, d0, da;
};
__device__ __forceinline__ ShmrSet shmr_for(double z) {
    // Behroozi+10 table 2, split at z = 0.8 (hmvec/hmvec.py:668-691)
    if (z <= 0.8) return {10.72, 0.55, 12.35, 0.28, 0.44, 0.18, 1.56, 2.51, 0.57, 0.17};
    return {11.09, 0.56, 12.27, -0.84, 0.65, 0.31, 1.12, -0.53, 0.56, -0.12};
}
__device__ __forceinline__ double shmr_log10mh(double lms, double a, const ShmrSet& s) {
    const double am1 = a - 1.0;
    const double lM1 = s.M1 + s.M1a * am1;
    const double lMs0 = s.Ms0 + s.Msa * am1;
    const double beta = s.b0 + s.ba * am1;
    const double gamma = s.g0 + s.ga * am1;
    const double delta = s.d0 + s.da * am1;
    const double d = lms - lMs0;
    return -0.5 + lM1 + beta * d + pow10_fast(delta * d) / (1.0 + pow10_fast(-gamma * d));
}

constexpr int SHMR_N = 4000;
// log10 M* grid of the reference's inverse table, np.linspace(-18,18,4000) (hmvec.py:640)
__device__ __forceinline__ double shmr_grid(int j) {
    const double gstep = 36.0 / (double)(SHMR_N - 1);
    return j == SHMR_N - 1 ? 18.0 : (double)j * gstep + (-18.0);
}
// np.interp(lmh, mh, grid) with numpy's clamped ends and exact-knot rule (hmvec.py:645)
__device__ __forceinline__ double shmr_inverse(const double* mh /* LDS, SHMR_N */, double lmh) {
    if (lmh < mh[0]) return shmr_grid(0);
    if (lmh >= mh[SHMR_N - 1]) return shmr_grid(SHMR_N - 1);
    int lo = 0, hi = SHMR_N - 1;          // mh[lo] <= lmh < mh[lo+1]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (mh[mid] <= lmh) lo = mid; else hi = mid;
    }
    if (mh[lo] == lmh) return shmr_grid(lo);
    const double slope = (shmr_grid(lo + 1) - shmr_grid(lo)) / (mh[lo + 1] - mh[lo]);
    return slope * (lmh - mh[lo]) + shmr_grid(lo);
}
struct HodDev {
    double sig, alphasat, Bsat, betasat, Bcut, betacut;
    int corr;
};

// The same inversion without the table: the bracket search evaluates the table entries it visits on the
// fly (12 of the 4000 per mass).  Entry j is shmr_log10mh(shmr_grid(j)) in both forms, so the bracket, the
// knots and the interpolated value are the same numbers - but nothing has to be built first and no LDS is
// held, which is what lets the HOD of a redshift run as one link of a per-z chain inside a grouped launch
// beside workgroups of another kind (LDS is allocated per launch, for every workgroup alike).
__device__ __forceinline__ double shmr_inverse_direct(double lmh, double a, const ShmrSet& S) {
    const double m0 = shmr_log10mh(shmr_grid(0), a, S);
    if (lmh < m0) return shmr_grid(0);
    const double mN = shmr_log10mh(shmr_grid(SHMR_N - 1), a, S);
    if (lmh >= mN) return shmr_grid(SHMR_N - 1);
    int lo = 0, hi = SHMR_N - 1;          // mh[lo] <= lmh < mh[hi]
    double mlo = m0, mhi = mN;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        const double mm = shmr_log10mh(shmr_grid(mid), a, S);
        if (mm <= lmh) { lo = mid; mlo = mm; } else { hi = mid; mhi = mm; }
    }
    if (mlo == lmh) return shmr_grid(lo);
    const double slope = (shmr_grid(lo + 1) - shmr_grid(lo)) / (mhi - mlo);
    return slope * (lmh - mlo) + shmr_grid(lo);
}

struct HodRowArgs {
    int nm;
    HodDev P;
    const double *zs, *ms, *lthr, *nzm, *bh, *wm;
    double *Nc, *Ns, *NsNsm1, *NcNs, *ngal, *bg;
};
constexpr int HOD_MAX_TILES = 1024;      // 64-mass tiles per redshift (nm <= 65536)

// The HOD has two halves.  The occupation numbers <Nc>, <Ns>, <Ns(Ns-1)>, <NcNs> of a (z,m) point depend on
// INPUTS only (z, m, the stellar-mass threshold, the HOD parameters) - and carry all the cost: the SHMR
// inversion, an erf, two powers, an exp.  n_gal and b_g are sums over m of those times n(z,m), b(z,m).  A
// grouped pass therefore evaluates the occupations in its FRONT launch beside the sigma^2 contraction (no
// register cap there, one thread per point) and leaves only the sums to the per-z chain.
__device__ __forceinline__ void hod_occ_point(const HodRowArgs& A, int z, int m) {
#pragma clang fp contract(off)
    const HodDev& P = A.P;
    const double zz = A.zs[z], a = 1.0 / (1.0 + zz);
    const ShmrSet S = shmr_for(zz);
    const double thr = A.lthr[z];
    const double mthr_halo = shmr_log10mh(thr, a, S);
    const double Msat = 1.0e12 * P.Bsat * pow10_fast((mthr_halo - 12.0) * P.betasat);
    const double Mcut = 1.0e12 * P.Bcut * pow10_fast((mthr_halo - 12.0) * P.betacut);
    const double denom = sqrt(2.0) * P.sig;
    const double lmh = log10(A.ms[m]);
    const double lmstar = shmr_inverse_direct(lmh, a, S);
    const double nc = 0.5 * (1.0 - erf((thr - lmstar) / denom));
    const double mass = pow10_fast(lmh);
    const double ns = nc * powr_fast(mass / Msat, P.alphasat) * exp(-Mcut / mass);
    double nn, cn;
    if (P.corr == 0) {
        nn = (fabs(nc) <= 1.0e-8) ? 0.0 : (ns * ns) / nc;   // np.isclose(Nc, 0)
        cn = ns;
    } else {
        nn = ns * ns;
        cn = ns * nc;
    }
    const size_t idx = (size_t)z * A.nm + m;
    A.Nc[idx] = nc; A.Ns[idx] = ns; A.NsNsm1[idx] = nn; A.NcNs[idx] = cn;
}

// n_gal(z), b_g(z) of one redshift by one workgroup of nthr threads, from the stored occupations.  The order
// is fixed by nm alone: wavefront sums over the 64-mass tiles, then the tiles in order - whatever the
// workgroup size.  part: 2 * ceil(nm/64) doubles of LDS.
__device__ __forceinline__ void hod_sums_row(const HodRowArgs& A, int z, int nthr, double* part) {
#pragma clang fp contract(off)
    const int nm = A.nm, lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = nthr >> 6;
    const int ntile = (nm + 63) / 64;
    for (int tile = w; tile < ntile; tile += nw) {
        const int m = tile * 64 + lane;
        double t = 0.0, tb = 0.0;
        if (m < nm) {
            const size_t idx = (size_t)z * nm + m;
            t = A.wm[m] * (A.nzm[idx] * (A.Nc[idx] + A.Ns[idx]));
            tb = t * A.bh[idx];
        }
        const double tn = wave_sum(t), tbs = wave_sum(tb);
        if (lane == 0) { part[2 * tile] = tn; part[2 * tile + 1] = tbs; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double sn = 0.0, sb = 0.0;
        for (int tile = 0; tile < ntile; ++tile) { sn += part[2 * tile]; sb += part[2 * tile + 1]; }
        A.ngal[z] = sn;
        A.bg[z] = sb / sn;
    }
}

// One block per z: both halves.
__global__ __launch_bounds__(1024) void hod_kernel(HodRowArgs A) {
    __shared__ double part[2 * HOD_MAX_TILES];
    for (int m = threadIdx.x; m < A.nm; m += blockDim.x) hod_occ_point(A, blockIdx.x, m);
    __syncthreads();
    hod_sums_row(A, blockIdx.x, blockDim.x, part);
}

}  // namespace hmg
