// Device sine/cosine integrals Si(x), Ci(x) in fp64 for gfx950.
//
// The reference calls scipy.special.sici (hmvec/hmvec.py:350-351), a third-party
// routine not in the reference tree: Cephes Math Library 2.1 `sici.c` (S. L. Moshier,
// 1984-1989) as vendored by scipy 1.15.3 (scipy/special/xsf/cephes/sici.h).  This is a
// restatement of that published algorithm for x > 0 with the published minimax
// coefficients: rational approximations in x^2 for x <= 4, and auxiliary functions
// f(x), g(x) (rationals in 1/x^2, split at x = 8) for x > 4 with
//     Si = pi/2 - f cos x - g sin x,   Ci = f sin x - g cos x.
// sx, cx = sin(x), cos(x) are only read on the x > 4 branch, so the caller shares them with
// its own trigonometry.
#pragma once
#include <hip/hip_runtime.h>

namespace hmg {

constexpr double EULER_GAMMA = 0.577215664901532860606512090082402431;
constexpr double HALF_PI = 1.57079632679489661923;

// Implementation notes:
//  * the Horner chains contract to FMAs and the two quotients of each branch share one
//    reciprocal (differs from sici_pos by a few ulp, |delta| < 1e-15 on Si/Ci, far inside
//    the 1e-12 absolute gate on u(k));
//  * the 88 coefficients are read from a table in global memory at wave-uniform addresses,
//    so they arrive in SGPRs through the scalar cache and feed the FMAs directly.  As
//    compile-time constants they become 64-bit literals, which gfx950's VOP3 cannot encode:
//    the compiler then spends one v_mov_b64 per coefficient per evaluation (+35 % VALU).
struct SiciTable {
    double SN[6], SD[6], CN[6], CD[6], FN4[7], FD4[7], GN4[8], GD4[7], FN8[9], FD8[8], GN8[9], GD8[9];
    double AF[8], AG[9];   // asymptotic series of x f(x) and x^2 g(x) in 1/x^2, highest power first (x >= SICI_ASYM_X)
};

// a*x + c with the coefficient c read straight from an SGPR pair.  Left to itself hipcc
// copies every scalar-loaded coefficient into VGPRs (2 v_mov_b32) so that it can use the
// 2-address v_fmac_f64; the VOP3 form takes the SGPR directly (one constant-bus read).
__device__ __forceinline__ double fma_vvs(double a, double x, double c) {
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(x), "s"(c));
    return r;
}
// c0*x + c1 with both coefficients scalar: only one may ride the constant bus.
__device__ __forceinline__ double fma_svs(double c0, double x, double c1) {
    double r;
    asm("v_mul_f64 %0, %1, %2\n\tv_add_f64 %0, %0, %3" : "=&v"(r) : "s"(c0), "v"(x), "s"(c1));
    return r;
}

template <int N>
__device__ __forceinline__ double horner_s(double x, const double* __restrict__ c) {
    double a = fma_svs(c[0], x, c[1]);
#pragma unroll
    for (int i = 2; i < N; ++i) a = fma_vvs(a, x, c[i]);
    return a;
}
template <int N>
__device__ __forceinline__ double horner1_s(double x, const double* __restrict__ c) {
    double a = x + c[0];
#pragma unroll
    for (int i = 1; i < N; ++i) a = fma_vvs(a, x, c[i]);
    return a;
}

// 1/x to ~1 ulp: hardware estimate + two Newton steps (5 VALU ops instead of the ~10 of an
// IEEE-exact division).  x must be finite, non-zero and normal.
__device__ __forceinline__ double rcp_fast(double x) {
    double r = __builtin_amdgcn_rcp(x);
    double e = fma(-x, r, 1.0);
    r = fma(r, e, r);
    e = fma(-x, r, 1.0);
    return fma(r, e, r);
}

// sin and cos of 0 <= x < 2^30 by a three-term Cody-Waite reduction (exact products under
// FMA) and the fdlibm kernel polynomials on [-pi/4, pi/4]; |error| < 1 ulp of the result plus
// 2e-16 absolute from the reduction.  About 30 VALU ops for both values.
__device__ __forceinline__ void sincos_fast(double x, double& s, double& c) {
    const double kd = rint(x * 0.63661977236758134308);
    double r = fma(-kd, 1.5707963267948966, x);
    r = fma(-kd, 6.123233995736766e-17, r);
    r = fma(-kd, -1.4973849048591698e-33, r);
    const int q = (int)kd;
    const double z = r * r;
    double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma(ps, z, 2.75573137070700676789e-06);
    ps = fma(ps, z, -1.98412698298579493134e-04);
    ps = fma(ps, z, 8.33333333332248946124e-03);
    ps = fma(ps, z, -1.66666666666666324348e-01);
    const double sr = fma(r * z, ps, r);
    double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma(pc, z, -2.75573143513906633035e-07);
    pc = fma(pc, z, 2.48015872894767294178e-05);
    pc = fma(pc, z, -1.38888888888741095749e-03);
    pc = fma(pc, z, 4.16666666666666019037e-02);
    const double cr = fma(z * z, pc, fma(z, -0.5, 1.0));
    const double sa = (q & 1) ? cr : sr;
    const double ca = (q & 1) ? sr : cr;
    s = (q & 2) ? -sa : sa;
    c = ((q + 1) & 2) ? -ca : ca;
}

// +-(sin x - y cos x) for 0 <= x < 2^30 with the SIGN LEFT OPEN: the numerator of the top-hat window (y = x), which
// the sigma^2 contraction squares.  The same reduction and kernel polynomials as sincos_fast; the quadrant then only
// decides WHICH of the two kernel values plays the sine - q even: sr - y cr, q odd: cr + y sr - and the two sign flips
// of sincos_fast drop out (exact negations: the magnitude has the bits of sn - y cs formed from sincos_fast's results).
__device__ __forceinline__ double sin_minus_ycos_nosign(double x, double y) {
    const double kd = rint(x * 0.63661977236758134308);
    double r = fma(-kd, 1.5707963267948966, x);
    r = fma(-kd, 6.123233995736766e-17, r);
    r = fma(-kd, -1.4973849048591698e-33, r);
    const int q = (int)kd;
    const double z = r * r;
    // (Horner steps with the coefficient in an SGPR pair - fma_vvs: as literals of a plain fma() each costs two v_mov_b32
    // into the destination of a v_fmac_f64, 20 of the ~75 VALU instructions of a window)
    double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma_vvs(ps, z, 2.75573137070700676789e-06);
    ps = fma_vvs(ps, z, -1.98412698298579493134e-04);
    ps = fma_vvs(ps, z, 8.33333333332248946124e-03);
    ps = fma_vvs(ps, z, -1.66666666666666324348e-01);
    const double sr = fma(r * z, ps, r);
    double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma_vvs(pc, z, -2.75573143513906633035e-07);
    pc = fma_vvs(pc, z, 2.48015872894767294178e-05);
    pc = fma_vvs(pc, z, -1.38888888888741095749e-03);
    pc = fma_vvs(pc, z, 4.16666666666666019037e-02);
    const double cr = fma(z * z, pc, fma(z, -0.5, 1.0));
    const double sa = (q & 1) ? cr : sr;          // |sin x|-side value
    const double cb = (q & 1) ? -sr : cr;         // cos x up to the common sign
    return sa - y * cb;
}

// Si(x) and Ci(x) for x > 0 given sx = sin x, cx = cos x and z = 1/x^2 (read for x > 4 only).
// For x <= 4 the returned "ci" is only the rational part c(x) of
//     Ci(x) = gamma + ln x + c(x),
// and `small` is set: the caller adds gamma + ln x itself.  The NFW formula needs
// Ci((1+c)x) - Ci(x) only, where the logarithms of two small arguments collapse to the row
// constant ln(1+c) - no per-point log at all on ~2/3 of a typical grid, and no cancellation.
__device__ __forceinline__ void sici_fast(const SiciTable* __restrict__ T, double x, double sx,
                                          double cx, double z, double& si, double& ci, bool& small) {
    small = (x <= 4.0);
    if (small) {
        const double x2 = x * x;
        const double sd = horner_s<6>(x2, T->SD), cd = horner_s<6>(x2, T->CD);
        const double r = rcp_fast(sd * cd);
        si = x * horner_s<6>(x2, T->SN) * (cd * r);
        ci = x2 * horner_s<6>(x2, T->CN) * (sd * r);
        return;
    }
    double fn, fd, gn, gd;
    if (x < 8.0) {
        fn = horner_s<7>(z, T->FN4); fd = x * horner1_s<7>(z, T->FD4);
        gn = z * horner_s<8>(z, T->GN4); gd = horner1_s<7>(z, T->GD4);
    } else {
        fn = horner_s<9>(z, T->FN8); fd = x * horner1_s<8>(z, T->FD8);
        gn = z * horner_s<9>(z, T->GN8); gd = horner1_s<9>(z, T->GD8);
    }
    const double r = rcp_fast(fd * gd);
    const double f = fn * (gd * r), g = gn * (fd * r);
    si = HALF_PI - f * cx - g * sx;
    ci = f * sx - g * cx;
}

// Auxiliary functions f(x), g(x) of Si/Ci for x > 4 (z = 1/x^2): the same Cephes rationals as in
// sici_fast, returned on their own.  WANT_F = false evaluates g only.
// For x >= 64 the divergent asymptotic series
//     x f(x) ~ sum_n (-1)^n (2n)! / x^(2n),     x^2 g(x) ~ sum_n (-1)^n (2n+1)! / x^(2n)
// are at full double precision after 8 and 9 terms (first omitted terms 16!/64^16 = 2.6e-16 and
// 19!/64^18 = 3.7e-16 of the sums; checked against 40-digit arithmetic on [64, 256]: 2.5e-16 and 4.2e-16):
// 17 FMAs and no division instead of the 35 FMAs and one reciprocal of the two rationals.
constexpr double SICI_ASYM_X = 64.0;
template <bool WANT_F>
__device__ __forceinline__ void sici_aux(const SiciTable* __restrict__ T, double x, double z, double& f, double& g) {
    if (x >= SICI_ASYM_X) {
        g = z * horner_s<9>(z, T->AG);
        f = WANT_F ? (x * z) * horner_s<8>(z, T->AF) : 0.0;
        return;
    }
    double fn = 0.0, fd = 1.0, gn, gd;
    if (x < 8.0) {
        if (WANT_F) { fn = horner_s<7>(z, T->FN4); fd = x * horner1_s<7>(z, T->FD4); }
        gn = z * horner_s<8>(z, T->GN4); gd = horner1_s<7>(z, T->GD4);
    } else {
        if (WANT_F) { fn = horner_s<9>(z, T->FN8); fd = x * horner1_s<8>(z, T->FD8); }
        gn = z * horner_s<9>(z, T->GN8); gd = horner1_s<9>(z, T->GD8);
    }
    if (WANT_F) {
        const double r = rcp_fast(fd * gd);
        f = fn * (gd * r);
        g = gn * (fd * r);
    } else {
        f = 0.0;
        g = gn * rcp_fast(gd);
    }
}

// Host copy of the table (uploaded once per context).
inline SiciTable sici_table_host() {
    SiciTable t;
    auto cp = [](double* d, const double* s, int n) { for (int i = 0; i < n; ++i) d[i] = s[i]; };
    static const double hSN[6] = {-8.39167827910303881427E-11, 4.62591714427012837309E-8, -9.75759303843632795789E-6, 9.76945438170435310816E-4, -4.13470316229406538752E-2, 1.00000000000000000302E0};
    static const double hSD[6] = {2.03269266195951942049E-12, 1.27997891179943299903E-9, 4.41827842801218905784E-7, 9.96412122043875552487E-5, 1.42085239326149893930E-2, 9.99999999999999996984E-1};
    static const double hCN[6] = {2.02524002389102268789E-11, -1.35249504915790756375E-8, 3.59325051419993077021E-6, -4.74007206873407909465E-4, 2.89159652607555242092E-2, -1.00000000000000000080E0};
    static const double hCD[6] = {4.07746040061880559506E-12, 3.06780997581887812692E-9, 1.23210355685883423679E-6, 3.17442024775032769882E-4, 5.10028056236446052392E-2, 4.00000000000000000080E0};
    static const double hFN4[7] = {4.23612862892216586994E0, 5.45937717161812843388E0, 1.62083287701538329132E0, 1.67006611831323023771E-1, 6.81020132472518137426E-3, 1.08936580650328664411E-4, 5.48900223421373614008E-7};
    static const double hFD4[7] = {8.16496634205391016773E0, 7.30828822505564552187E0, 1.86792257950184183883E0, 1.78792052963149907262E-1, 7.01710668322789753610E-3, 1.10034357153915731354E-4, 5.48900252756255700982E-7};
    static const double hFN8[9] = {4.55880873470465315206E-1, 7.13715274100146711374E-1, 1.60300158222319456320E-1, 1.16064229408124407915E-2, 3.49556442447859055605E-4, 4.86215430826454749482E-6, 3.20092790091004902806E-8, 9.41779576128512936592E-11, 9.70507110881952024631E-14};
    static const double hFD8[8] = {9.17463611873684053703E-1, 1.78685545332074536321E-1, 1.22253594771971293032E-2, 3.58696481881851580297E-4, 4.92435064317881464393E-6, 3.21956939101046018377E-8, 9.43720590350276732376E-11, 9.70507110881952025725E-14};
    static const double hGN4[8] = {8.71001698973114191777E-2, 6.11379109952219284151E-1, 3.97180296392337498885E-1, 7.48527737628469092119E-2, 5.38868681462177273157E-3, 1.61999794598934024525E-4, 1.97963874140963632189E-6, 7.82579040744090311069E-9};
    static const double hGD4[7] = {1.64402202413355338886E0, 6.66296701268987968381E-1, 9.88771761277688796203E-2, 6.22396345441768420760E-3, 1.73221081474177119497E-4, 2.02659182086343991969E-6, 7.82579218933534490868E-9};
    static const double hGN8[9] = {6.97359953443276214934E-1, 3.30410979305632063225E-1, 3.84878767649974295920E-2, 1.71718239052347903558E-3, 3.48941165502279436777E-5, 3.47131167084116673800E-7, 1.70404452782044526189E-9, 3.85945925430276600453E-12, 3.14040098946363334640E-15};
    static const double hGD8[9] = {1.68548898811011640017E0, 4.87852258695304967486E-1, 4.67913194259625806320E-2, 1.90284426674399523638E-3, 3.68475504442561108162E-5, 3.57043223443740838771E-7, 1.72693748966316146736E-9, 3.87830166023954706752E-12, 3.14040098946363335242E-15};
    static const double hAF[8] = {-87178291200.0, 479001600.0, -3628800.0, 40320.0, -720.0, 24.0, -2.0, 1.0};   // (-1)^n (2n)!, n = 7..0
    static const double hAG[9] = {355687428096000.0, -1307674368000.0, 6227020800.0, -39916800.0, 362880.0, -5040.0, 120.0, -6.0, 1.0};   // (-1)^n (2n+1)!, n = 8..0
    cp(t.AF, hAF, 8); cp(t.AG, hAG, 9);
    cp(t.SN, hSN, 6); cp(t.SD, hSD, 6); cp(t.CN, hCN, 6); cp(t.CD, hCD, 6);
    cp(t.FN4, hFN4, 7); cp(t.FD4, hFD4, 7); cp(t.GN4, hGN4, 8); cp(t.GD4, hGD4, 7);
    cp(t.FN8, hFN8, 9); cp(t.FD8, hFD8, 8); cp(t.GN8, hGN8, 9); cp(t.GD8, hGD8, 9);
    return t;
}

}  // namespace hmg
