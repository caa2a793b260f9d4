// A5/A7 - concentration, virial radius, M_vir -> M_200c (hmvec/hmvec.py:68-73,111-115,748-798).
// Part of the ONE translation unit hmgrid.hip (included there in this order; not a stand-alone header).
#pragma once

namespace hmg {

// ---------------------------------------------------------------- A5: c(m,z), rvir(m,z)
__global__ void halo_structure_kernel(int nz, int nm, const double* __restrict__ ms,
                                      const double* __restrict__ zs,
                                      const double* __restrict__ delta,
                                      const double* __restrict__ rho, double A, double alpha,
                                      double beta, double h, double* __restrict__ cs,
                                      double* __restrict__ rv, double* __restrict__ rs) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nz * nm) return;
    const int z = idx / nm, m = idx - z * nm;
    const double mm = ms[m];
    const double c = A * pow(h * mm / 2.0e12, alpha) * pow(1.0 + zs[z], beta);
    const double r = pow(3.0 * mm / 4.0 / M_PI / delta[z] / rho[z], 1.0 / 3.0);
    cs[idx] = c;
    rv[idx] = r;
    rs[idx] = r / c;
}

// ---------------------------------------------------------------- A7: mass conversion
__device__ __forceinline__ double fcon(double c) { return log(1.0 + c) - c / (1.0 + c); }

// The reference solves M1 F(c1) = M2 F(c2), F = 1/mu(c), mu(c) = ln(1+c) - c/(1+c), for ln M2 with
// c2 = c1 ((M2/M1) ratio)^(1/3) (scipy.optimize.newton without fprime: a vectorised secant from
// ln M1 to |dl| < 1.5e-8 with a global stop test).  Eliminating M2 = M1 (c2/c1)^3 / ratio leaves
// one equation in the new concentration alone,
//     h(c) = c^3/mu(c) - K = 0,   K = ratio c1^3 / mu(c1),
// solved here by Newton with h' = 3c^2/mu - c^4/((1+c)^2 mu^2): one logarithm per iteration, 4-5
// iterations from c = c1 ratio^(1/3) (the secant's starting point M2 = M1) to rounding.  Same
// root, so same M2 (to ~1e-15 instead of the secant's 1e-8).
__device__ __forceinline__ double mdelta_solve(double M1, double c1, double ratio) {
    const double K = ratio * (c1 * c1 * c1) / fcon(c1);
    double c = c1 * cbrt(ratio);
    for (int it = 0; it < 16; ++it) {
        const double ip = rcp_fast(1.0 + c);
        const double q = c * ip;                    // c/(1+c)
        const double mu = log1p(c) - q;
        const double c2 = c * c;
        // dc = h/h' with numerator and denominator multiplied by mu^2
        const double dc = (c2 * c - K * mu) * mu * rcp_fast(c2 * (3.0 * mu - q * q));
        c -= dc;
        if (fabs(dc) <= 4.0e-15 * c) break;   // quadratic: the step just taken leaves an error ~dc^2/c
    }
    const double s = c / c1;
    return M1 * (s * s * s) / ratio;
}

__global__ void mdelta_kernel(int nz, int nm, const double* __restrict__ ms,
                              const double* __restrict__ cs, const double* __restrict__ d1,
                              double delta2, const double* __restrict__ rho2,
                              double* __restrict__ m2, double* __restrict__ r2) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nz * nm) return;
    const int z = idx / nm, m = idx - z * nm;
    const double M2 = mdelta_solve(ms[m], cs[idx], d1[z] / (delta2 * rho2[z]));
    m2[idx] = M2;
    r2[idx] = pow(3.0 * M2 / 4.0 / M_PI / delta2 / rho2[z], 1.0 / 3.0);
}

}  // namespace hmg
