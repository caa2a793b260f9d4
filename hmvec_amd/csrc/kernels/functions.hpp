// device mirrors of the reference's free functions (hmg_fn2d and friends).
// Part of the ONE translation unit hmgrid.hip (included there in this order; not a stand-alone header).
#pragma once

namespace hmg {

// ---------------------------------------------------------------- function mirrors (hmg_fn2d & co)
// The reference's free functions on the path, evaluated over a broadcast (rows, cols) grid.
// These mirror numpy's expressions operation by operation (generic pow/exp/log10, no fused
// multiply-add) - they are the API-parity entry points, not the fused hot kernels above.
struct FnArgs {
    int op, rows, cols;
    const double* in[HMG_FN_MAXIN];
    int sr[HMG_FN_MAXIN], sc[HMG_FN_MAXIN];
    double par[HMG_FN_MAXPAR];
    double* out;
};

__device__ __forceinline__ double batt_fit(double m, double z, const double* f) {
    return f[0] * pow(m / 1.0e14, f[1]) * pow(1.0 + z, f[2]);
}

__global__ __launch_bounds__(256) void fn2d_kernel(FnArgs A) {
#pragma clang fp contract(off)
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)A.rows * A.cols) return;
    const int r = (int)(idx / A.cols), c = (int)(idx - (size_t)r * A.cols);
    auto X = [&](int i) { return A.in[i][(size_t)r * A.sr[i] + (size_t)c * A.sc[i]]; };
    const double* par = A.par;
    double y = 0.0;
    switch (A.op) {
    case HMG_FN_TINKER_BIAS: {
        const double nu = X(0), dc = 1.686, yy = log10(par[0]);
        const double ey = exp(-pow(4.0 / yy, 4.0));
        const double Ay = 1.0 + 0.24 * yy * ey, ay = 0.44 * yy - 0.88, Cy = 0.019 + 0.107 * yy + 0.19 * ey;
        const double nua = pow(nu, ay);
        y = 1.0 - Ay * (nua / (nua + pow(dc, ay))) + 0.183 * pow(nu, 1.5) + Cy * pow(nu, 2.4);
        break;
    }
    case HMG_FN_TINKER_FNU:
    case HMG_FN_TINKER_FSIGMA: {
        const bool from_sigma2 = (A.op == HMG_FN_TINKER_FSIGMA);
        const double nu = from_sigma2 ? par[3] / sqrt(X(0)) : X(0), zin = X(1);
        // zs*heaviside(3-zs,0) + 3*heaviside(zs-3,0): z<3 -> z, z==3 -> 0, z>3 -> 3 (tinker.py:53)
        const double z = zin < 3.0 ? zin : (zin > 3.0 ? 3.0 : 0.0);
        const double beta = 0.589 * pow(1.0 + z, 0.20), phi = -0.729 * pow(1.0 + z, -0.08);
        const double eta = -0.243 * pow(1.0 + z, 0.27), gamma = 0.864 * pow(1.0 + z, -0.01);
        const double un = (1.0 + pow(beta * nu, -2.0 * phi)) * pow(nu, 2.0 * eta) * exp(-gamma * (nu * nu) / 2.0);
        double alpha = par[1];
        if (par[0] != 0.0) {   // interp1d(izs, ialphas) - linear; out-of-range z is rejected on the host
            const double* tz = A.in[2];
            const double* ta = A.in[3];
            const int nt = (int)par[2];
            int lo = 0, hi = nt - 1;
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (tz[mid] < z) lo = mid; else hi = mid;     // searchsorted(side='left') bracket
            }
            const double slope = (ta[hi] - ta[lo]) / (tz[hi] - tz[lo]);
            alpha = slope * (z - tz[lo]) + ta[lo];
        }
        y = alpha * un;
        if (from_sigma2) y = nu * y;     // the mass function's f is nu * f_nu (hmvec.py:145)
        break;
    }
    case HMG_FN_ST_FSIGMA: {
        const double s2 = X(0), sig = sqrt(s2), sA = par[0], sa = par[1], sp = par[2], dc = par[3];
        y = sA * sqrt(2.0 * sa / M_PI) * (1.0 + pow(s2 / sa / (dc * dc), sp)) * (dc / sig) *
            exp(-sa * (dc * dc) / 2.0 / s2);
        break;
    }
    case HMG_FN_MHALO_STELLAR: {
        const double z = X(0), lms = X(1), a = 1.0 / (1.0 + z), am1 = a - 1.0;
        const ShmrSet s = shmr_for(z);
        const double d = lms - (s.Ms0 + s.Msa * am1);
        y = -0.5 + (s.M1 + s.M1a * am1) + (s.b0 + s.ba * am1) * d +
            pow(10.0, (s.d0 + s.da * am1) * d) / (1.0 + pow(10.0, -(s.g0 + s.ga * am1) * d));
        break;
    }
    case HMG_FN_MHALO_STELLAR_CORE: {
        const double lms = X(0), am1 = X(1) - 1.0;
        const double d = lms - (par[0] + par[1] * am1);
        y = -0.5 + (par[2] + par[3] * am1) + (par[4] + par[5] * am1) * d +
            pow(10.0, (par[8] + par[9] * am1) * d) / (1.0 + pow(10.0, -(par[6] + par[7] * am1) * d));
        break;
    }
    case HMG_FN_HOD_NC:
        y = 0.5 * (1.0 - erf((X(1) - X(0)) / (sqrt(2.0) * par[0])));
        break;
    case HMG_FN_HOD_NS: {
        const double mass = pow(10.0, X(1));
        y = X(0) * pow(mass / X(2), par[0]) * exp(-X(3) / mass);
        break;
    }
    case HMG_FN_HOD_MFUNC:
        y = 1.0e12 * par[0] * pow(10.0, (X(0) - 12.0) * par[1]);
        break;
    case HMG_FN_HOD_NSNSM1: {
        const double nc = X(0), ns = X(1);
        if (par[0] == 0.0) y = (fabs(nc) <= 1.0e-8) ? 0.0 : (ns * ns) / nc;   // np.isclose(Nc, 0)
        else y = ns * ns;
        break;
    }
    case HMG_FN_HOD_NCNS:
        y = par[0] == 0.0 ? X(1) : X(1) * X(0);
        break;
    case HMG_FN_FCON: {
        const double cc = X(0);
        y = log(1.0 + cc) - cc / (1.0 + cc);
        break;
    }
    case HMG_FN_RHO_NFW: {
        const double x = X(0) / X(2), op = 1.0 + x;
        y = X(1) / x / (op * op);
        break;
    }
    case HMG_FN_R_FROM_M:
        y = pow(3.0 * X(0) / 4.0 / M_PI / X(2) / X(1), 1.0 / 3.0);
        break;
    case HMG_FN_DUFFY:
        y = par[0] * pow(par[3] * X(0) / 2.0e12, par[1]) * pow(1.0 + X(1), par[2]);
        break;
    case HMG_FN_BATT_FIT:
        y = batt_fit(X(0), X(1), par);
        break;
    case HMG_FN_RHO_GAS_X:
    case HMG_FN_RHO_GAS_R: {
        const double m = X(1), z = X(2), rhoc = X(3);
        double x = X(0);
        if (A.op == HMG_FN_RHO_GAS_R) x = 2.0 * x / pow(3.0 * m / 4.0 / M_PI / 200.0 / rhoc, 1.0 / 3.0);
        const double omb = par[0], omm = par[1], gamma = par[2];
        const double rho0 = batt_fit(m, z, par + 3), alpha = batt_fit(m, z, par + 6), beta = batt_fit(m, z, par + 9);
        y = (omb / omm) * rhoc * rho0 * pow(x, gamma) * pow(1.0 + pow(x, alpha), -(beta + gamma) / alpha);
        break;
    }
    case HMG_FN_PE_X:
    case HMG_FN_PE_R: {
        double x = X(0), m, R200, z, rhoc;
        if (A.op == HMG_FN_PE_X) {
            m = X(1); R200 = X(2); z = X(3); rhoc = X(4);
        } else {
            m = X(1); z = X(2); rhoc = X(3);
            R200 = pow(3.0 * m / 4.0 / M_PI / 200.0 / rhoc, 1.0 / 3.0);
            x = x / R200;
        }
        const double omb = par[0], omm = par[1], alpha = par[2], gamma = par[3], G = par[13];
        const double P0 = batt_fit(m, z, par + 4), xc = batt_fit(m, z, par + 7), beta = batt_fit(m, z, par + 10);
        const double eFrac = 2.0 * (0.76 + 1.0) / (5.0 * 0.76 + 3.0);
        const double t = x / xc;
        y = eFrac * (omb / omm) * 200.0 * m * G * rhoc / (2.0 * R200) * P0 * pow(t, gamma) *
            pow(1.0 + pow(t, alpha), -beta);
        break;
    }
    case HMG_FN_NGAL_INTEGRAND:
        y = X(0) * (X(1) + X(2));
        break;
    case HMG_FN_A2Z:
        y = 1.0 / X(0) - 1.0;
        break;
    case HMG_FN_MDELTA:
        y = mdelta_solve(X(0), X(1), X(2) / X(3));
        break;
    case HMG_FN_BG_INTEGRAND:
        y = X(0) * (X(1) + X(2)) * X(3);
        break;
    case HMG_FN_WKR: {
        const double kR = X(0) * X(1);
        if (kR < par[0]) {
            const double xx = kR * kR;
            y = 1.0 - 0.1 * xx + 0.00357142857143 * xx * xx;
        } else {
            y = 3.0 * (sin(kR) - kR * cos(kR)) / (kR * kR * kR);
        }
        break;
    }
    case HMG_FN_LINCOMB3:
        y = par[0] * X(0) + par[1] * X(1) + par[2] * X(2);
        break;
    case HMG_FN_BRUTE_INTEGRAND: {
        const double r = X(0), k = X(2);
        y = 4.0 * M_PI * r * sin(r * k) * X(1) / k;
        break;
    }
    }
    A.out[idx] = y;
}

// Mstellar_halo: one block per z, table in LDS, exactly the inversion hod_kernel uses.
__global__ __launch_bounds__(1024) void mstellar_halo_kernel(int nm, const double* __restrict__ zs,
                                                            const double* __restrict__ lmh,
                                                            double* __restrict__ out) {
#pragma clang fp contract(off)
    __shared__ double mh[SHMR_N];
    const int z = blockIdx.x;
    const double zz = zs[z], a = 1.0 / (1.0 + zz);
    const ShmrSet S = shmr_for(zz);
    for (int j = threadIdx.x; j < SHMR_N; j += blockDim.x) mh[j] = shmr_log10mh(shmr_grid(j), a, S);
    __syncthreads();
    for (int m = threadIdx.x; m < nm; m += blockDim.x) out[(size_t)z * nm + m] = shmr_inverse(mh, lmh[m]);
}

// out = a + b (get_power = P_1h + P_2h on the device: one array crosses PCIe instead of two)
__global__ void add2_kernel(size_t n, const double* __restrict__ a, const double* __restrict__ b,
                            double* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] + b[i];
}

// np.trapz(y, x, axis=-1): sum_i (x[i+1]-x[i]) * (y[i+1]+y[i]) / 2, one block per row.
__global__ __launch_bounds__(256) void trapz_rows_kernel(int cols, const double* __restrict__ y,
                                                         const double* __restrict__ x,
                                                         double* __restrict__ out) {
#pragma clang fp contract(off)
    __shared__ double lds[16];
    const double* row = y + (size_t)blockIdx.x * cols;
    double acc = 0.0;
    for (int i = threadIdx.x; i + 1 < cols; i += blockDim.x) acc += (x[i + 1] - x[i]) * (row[i + 1] + row[i]) / 2.0;
    const double tot = block_sum(acc, lds);
    if (threadIdx.x == 0) out[blockIdx.x] = tot;
}

// fft_integral pieces: integrand x*y, and uk = -Im(F) * step
__global__ void xy_kernel(int rows, int n, const double* __restrict__ x, const double* __restrict__ y,
                          double* __restrict__ out) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)rows * n) return;
    out[idx] = x[idx % n] * y[idx];
}
__global__ void neg_imag_kernel(size_t count, double step, const double2* __restrict__ F, double* __restrict__ out) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= count) return;
    out[idx] = -F[idx].y * step;
}

// Tabulated-integrand twin of integrand_kernel (generic_profile_fft with an arbitrary rhofunc_x):
// theta(|x| <= cmax) * rho, the R2C input x*rho*theta and the trapz mass norm of the row.
__global__ __launch_bounds__(256) void table_integrand_kernel(int nxs, int row0, const double* __restrict__ xs,
                                                              const double* __restrict__ rho, int rho_shared,
                                                              const double* __restrict__ cmax, int do_norm,
                                                              double* __restrict__ fin, double* __restrict__ mnorm) {
#pragma clang fp contract(off)
    __shared__ double lds[16];
    const int lrow = blockIdx.x, row = row0 + lrow;
    const double* src = rho + (rho_shared ? 0 : (size_t)row * nxs);
    const double cm = cmax[row];
    double* dst = fin + (size_t)lrow * nxs;
    double acc = 0.0;
    for (int j = threadIdx.x; j < nxs; j += blockDim.x) {
        const double x = xs[j];
        const double rv = (fabs(x) > cm) ? 0.0 : src[j];
        dst[j] = x * rv;
        if (do_norm) {
            const double xl = (j > 0) ? xs[j - 1] : x, xr = (j + 1 < nxs) ? xs[j + 1] : x;
            acc += 0.5 * (xr - xl) * (rv * (x * x));
        }
    }
    if (do_norm) {
        const double tot = block_sum(acc, lds);
        if (threadIdx.x == 0) mnorm[lrow] = tot;
    } else if (threadIdx.x == 0) {
        mnorm[lrow] = 1.0;
    }
}

}  // namespace hmg
