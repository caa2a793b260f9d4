// Short fp64 log / exp / log1p for the radial-profile integrand (gfx950) - host-testable.
//
// The generalised-NFW integrand of the Battaglia profiles, amp * t^gamma * (1 + t^alpha)^(-e)
// (hmvec/hmvec.py:844-860,906-927), costs four transcendentals per sample.  The device library's
// log/exp/log1p are correctly rounded to < 1 ulp through double-double tails and special-case
// handling that this path does not need (arguments are finite, positive, normal): the versions
// here are the classical fdlibm reductions with FMA, about half the instructions, and stay within
// 2 ulp (checked against long double on the host: tests/test_fastmath_cpu.py).  A 2-ulp
// error on rho is 4e-16 relative on the integrand, far inside the 1e-12 absolute gate on u(k).
//
// The functions are plain C++ (frexp/ldexp/rint/fma) so that the same code runs on the host for
// the accuracy test; on the device the four calls map to single instructions.
#pragma once

#include <cmath>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define HMG_FM_HD __host__ __device__ __forceinline__
#else
#define HMG_FM_HD inline
#endif

namespace hmg {

// 1/x to ~1 ulp for finite, non-zero, normal x: estimate + two Newton steps on the device.
HMG_FM_HD double fm_rcp(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(x);
    double e = fma(-x, r, 1.0);
    r = fma(r, e, r);
    e = fma(-x, r, 1.0);
    return fma(r, e, r);
#else
    return 1.0 / x;
#endif
}

// One Horner step p*x + c with a compile-time coefficient c.  gfx950's VOP3 cannot encode a 64-bit literal, and left to
// itself hipcc prefers the two-address v_fmac_f64 with the coefficient copied into the destination first (two v_mov_b32
// per step: ~10 % of the profile row kernel's VALU instructions were such copies).  The VOP3 form reads the coefficient
// from an SGPR pair (two s_mov_b32 on the scalar unit, one constant-bus read): the same fused multiply-add, same bits.
HMG_FM_HD double fm_hstep(double p, double x, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(p), "v"(x), "s"(c));
    return r;
#else
    return fma(p, x, c);
#endif
}

// ln x for finite x > 0 (normal).  fdlibm e_log.c reduction: x = 2^k m, m in [sqrt(1/2), sqrt 2),
// f = m - 1, s = f/(2+f), ln m = f - f^2/2 + s (f^2/2 + R(s^2)), R the degree-7 minimax in s^2.
HMG_FM_HD double log_fast(double x) {
    int k;
    double m = frexp(x, &k);            // m in [0.5, 1)
    if (m < 0.70710678118654752440) {
        m *= 2.0;
        k -= 1;
    }
    const double f = m - 1.0;
    const double s = f * fm_rcp(2.0 + f);
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * fm_hstep(fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), w, 3.999999999940941908e-01);
    const double t2 = z * fm_hstep(fm_hstep(fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), w,
                                            2.857142874366239149e-01), w, 6.666666666666735130e-01);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)k;
    return fma(dk, 6.93147180369123816490e-01, -((hfsq - fma(s, hfsq + R, dk * 1.90821492927058770002e-10)) - f));
}

// e^y for |y| < 700 (outside, ldexp saturates to 0 / inf as exp would).  k = rint(y/ln 2),
// r = y - k ln 2 in two FMA steps, degree-13 Taylor polynomial on |r| <= ln2/2 (remainder 4e-18).
// CLAMP = false: the caller guarantees |y| < 1e9 (the exponent then fits an int without the two extra
// instructions; results are the same).
template <bool CLAMP = true>
HMG_FM_HD double exp_fast(double y) {
    const double kd = rint(y * 1.44269504088896338700);
    double r = fma(-kd, 6.93147180559945286227e-01, y);
    r = fma(-kd, 2.31904681384629955842e-17, r);
    double p = fma(r, 1.6059043836821613e-10, 2.08767569878681e-09);     // 1/13!, 1/12!
    p = fm_hstep(p, r, 2.505210838544172e-08);                             // 1/11!
    p = fm_hstep(p, r, 2.755731922398589e-07);                             // 1/10!
    p = fm_hstep(p, r, 2.7557319223985893e-06);                            // 1/9!
    p = fm_hstep(p, r, 2.48015873015873e-05);                              // 1/8!
    p = fm_hstep(p, r, 1.984126984126984e-04);                             // 1/7!
    p = fm_hstep(p, r, 1.3888888888888889e-03);                            // 1/6!
    p = fm_hstep(p, r, 8.333333333333333e-03);                             // 1/5!
    p = fm_hstep(p, r, 4.1666666666666664e-02);                            // 1/4!
    p = fm_hstep(p, r, 1.6666666666666666e-01);                            // 1/3!
    p = fma(p, r, 0.5);                                                    // (0.5 and 1.0 are inline constants)
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    if constexpr (CLAMP) return ldexp(p, (int)fmin(fmax(kd, -2000.0), 2000.0));
    else return ldexp(p, (int)kd);
}

// ln(1 + a) for finite a >= 0: ln w with w = fl(1 + a), plus the first-order correction for the
// rounding of w (a - (w - 1) is exact), which carries the whole result when a < 2^-53.
HMG_FM_HD double log1p_fast(double a) {
    const double w = 1.0 + a;
    const double c = a - (w - 1.0);
    return log_fast(w) + c * fm_rcp(w);
}

// ln(1 + a) for finite a >= 0 to 1.2e-16 ABSOLUTE (plus log_fast's 2 ulp): ln of the rounded sum, without the
// correction term - for an exponent of an exp, where only the absolute error counts (the integrand's
// (1 + t^alpha)^(-e) = exp(-e ln(1 + t^alpha)) takes a relative error e * 1.2e-16 from it).
HMG_FM_HD double log1p_abs(double a) { return log_fast(1.0 + a); }

}  // namespace hmg
