// K4/K5 - integrand and scale+interpolation kernels of the rocFFT route (hmvec/fft.py:56-115).
// Part of the ONE translation unit hmgrid.hip (included there in this order; not a stand-alone header).
#pragma once

namespace hmg {

// ---------------------------------------------------------------- K4: profile integrand (F1)
// One block per (z,m) row of the current chunk.  Writes the R2C input x*rho*theta and
// reduces mnorm = trapz(theta rho x^2, x) in the same pass.  Samples beyond the
// truncation radius are exact zeros and skip the pow evaluations (85 % of a Battaglia
// row at xmax=20); the row is written with 16 B stores.  trapz on the x grid: the weight
// of sample j is (x[j+1]-x[j-1])/2, one-sided at the two ends.
__device__ __forceinline__ double gnfw_rho(double x, double A, double XC, double AL, double EX,
                                           double gamma) {
    const double t = x / XC;
    return A * pow(t, gamma) * pow(1.0 + pow(t, AL), -EX);
}

__global__ __launch_bounds__(256) void integrand_kernel(
    int nxs, int row0, const double* __restrict__ xs, const double* __restrict__ amp,
    const double* __restrict__ xcs, const double* __restrict__ alphas,
    const double* __restrict__ expos, double amp_c, double xc_c, double alpha_c, double expo_c,
    double gamma, const double* __restrict__ cmax, int do_norm, int allow_vec, double* __restrict__ fin,
    double* __restrict__ mnorm) {
    __shared__ double lds[16];
    const int lrow = blockIdx.x, row = row0 + lrow;
    const double A = amp ? amp[row] : amp_c;
    const double XC = xcs ? xcs[row] : xc_c;
    const double AL = alphas ? alphas[row] : alpha_c;
    const double EX = expos ? expos[row] : expo_c;
    const double cm = cmax[row];
    double* dst = fin + (size_t)lrow * nxs;
    double acc = 0.0;
    const bool vec = allow_vec && ((nxs & 1) == 0);  // rows stay 16 B aligned when nxs is even
    const int npair = vec ? nxs / 2 : 0;
    for (int p = threadIdx.x; p < npair; p += blockDim.x) {
        const int j = 2 * p;
        const double2 xv = *reinterpret_cast<const double2*>(xs + j);
        double r0 = 0.0, r1 = 0.0;
        if (!(fabs(xv.x) > cm)) r0 = gnfw_rho(xv.x, A, XC, AL, EX, gamma);
        if (!(fabs(xv.y) > cm)) r1 = gnfw_rho(xv.y, A, XC, AL, EX, gamma);
        *reinterpret_cast<double2*>(dst + j) = make_double2(xv.x * r0, xv.y * r1);
        if (do_norm && (r0 != 0.0 || r1 != 0.0)) {
            const double xl = (j > 0) ? xs[j - 1] : xv.x, xr = (j + 2 < nxs) ? xs[j + 2] : xv.y;
            acc += 0.5 * (xv.y - xl) * (r0 * (xv.x * xv.x)) + 0.5 * (xr - xv.x) * (r1 * (xv.y * xv.y));
        }
    }
    if (!vec) {
        for (int j = threadIdx.x; j < nxs; j += blockDim.x) {
            const double x = xs[j];
            double rho = 0.0;
            if (!(fabs(x) > cm)) rho = gnfw_rho(x, A, XC, AL, EX, gamma);
            dst[j] = x * rho;
            if (do_norm) {
                const double xl = (j > 0) ? xs[j - 1] : x, xr = (j + 1 < nxs) ? xs[j + 1] : x;
                acc += 0.5 * (xr - xl) * (rho * (x * x));
            }
        }
    }
    if (do_norm) {
        const double tot = block_sum(acc, lds);
        if (threadIdx.x == 0) mnorm[lrow] = tot;
    } else if (threadIdx.x == 0) {
        mnorm[lrow] = 1.0;
    }
}

// ---------------------------------------------------------------- K5: fused scale + interp (F1 tail, F3)
// One block per (z,m) row.  The nh = nxs/2 positive-frequency modes of the row,
//     u_j = -Im(F_j) * step / kt_j / mnorm,
// are staged once in LDS (20 KB at nxs=5000); threads then walk the target k grid.  The
// source grid is uniform in k, so the bracket comes from one multiply + a +-1 fix-up
// against kout_j = kt_j / rss / (1+z) evaluated exactly as the reference does — this is
// the reference's Python double loop of np.interp (hmvec/fft.py:97-115).
template <bool STAGE>
__global__ __launch_bounds__(256) void interp_kernel(int nm, int nk, int nh, int row0, double step,
                                                     const double2* __restrict__ F /*[rows][nh+1]*/,
                                                     const double* __restrict__ kts,
                                                     const double* __restrict__ mnorm,
                                                     const double* __restrict__ rss,
                                                     const double* __restrict__ zs,
                                                     const double* __restrict__ ks,
                                                     const double* __restrict__ post,
                                                     double* __restrict__ out,
                                                     int* __restrict__ nconst,
                                                     double* __restrict__ cconst) {
#pragma clang fp contract(off)
    extern __shared__ double u[];  // u[j-1] for j = 1..nh
    const int lrow = blockIdx.x, row = row0 + lrow;
    const int z = row / nm;
    const double mn = mnorm[lrow];
    const double2* Frow = F + (size_t)lrow * (nh + 1);
    auto mode = [&](int j) {  // u_j, j in 1..nh
        const double ukt = -Frow[j].y * step;
        return ukt / kts[j] / mn;
    };
    if (STAGE) {
        for (int j = 1 + threadIdx.x; j <= nh; j += blockDim.x) u[j - 1] = mode(j);
        __syncthreads();
    }
    auto U = [&](int j) { return STAGE ? u[j - 1] : mode(j); };
    const double rs = rss[row], z1 = 1.0 + zs[z];
    const double pf = post ? post[row] : 1.0;
    auto kout = [&](int j) { return kts[j] / rs / z1; };  // j in 1..nh
    const double k_lo = kout(1), k_hi = kout(nh);
    const double inv_dk = 1.0 / k_lo;  // kts[j] = j*kts[1] up to rounding
    double* dst = out + (size_t)row * nk;
    if (nconst && threadIdx.x == 0) {
        int lo = 0, hi = nk;              // first i with !(ks[i] < k_lo); ks ascending
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (ks[mid] < k_lo) lo = mid + 1; else hi = mid;
        }
        nconst[row] = lo;
        const double v1 = U(1);
        cconst[row] = post ? v1 * pf : v1;
    }
    for (int i = threadIdx.x; i < nk; i += blockDim.x) {
        const double k = ks[i];
        double val;
        if (k < k_lo) {
            val = U(1);  // left = first positive-k mode
        } else if (k > k_hi) {
            val = 0.0;   // right = 0
        } else if (k == k_hi) {
            val = U(nh);
        } else {
            int j = (int)(k * inv_dk);
            j = j < 1 ? 1 : (j > nh - 1 ? nh - 1 : j);
            while (j > 1 && kout(j) > k) --j;
            while (j < nh - 1 && kout(j + 1) <= k) ++j;
            const double x0 = kout(j), x1 = kout(j + 1);
            const double y0 = U(j), y1 = U(j + 1);
            if (x0 == k) {
                val = y0;
            } else {
                const double slope = (y1 - y0) / (x1 - x0);
                val = slope * (k - x0) + y0;
            }
        }
        dst[i] = post ? val * pf : val;
    }
}

}  // namespace hmg
