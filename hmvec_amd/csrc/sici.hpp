// Device sine/cosine integrals Si(x), Ci(x) in fp64 for gfx950.
//
// The reference calls scipy.special.sici (hmvec/hmvec.py:350-351), a third-party
// routine not in the reference tree: Cephes Math Library 2.1 `sici.c` (S. L. Moshier,
// 1984-1989) as vendored by scipy 1.15.3 (scipy/special/xsf/cephes/sici.h).  This is a
// restatement of that published algorithm for x > 0 with the published minimax
// coefficients: rational approximations in x^2 for x <= 4, and auxiliary functions
// f(x), g(x) (rationals in 1/x^2, split at x = 8) for x > 4 with
//     Si = pi/2 - f cos x - g sin x,   Ci = f sin x - g cos x.
// FP contraction is disabled so the Horner chains round exactly like the scalar CPU
// build scipy ships (no FMA), which keeps the GPU/CPU difference at the ulp level even
// where the NFW formula cancels.
#pragma once
#include <hip/hip_runtime.h>

namespace hmg {

__device__ static const double SN[6] = {
    -8.39167827910303881427E-11, 4.62591714427012837309E-8, -9.75759303843632795789E-6,
    9.76945438170435310816E-4, -4.13470316229406538752E-2, 1.00000000000000000302E0};
__device__ static const double SD[6] = {
    2.03269266195951942049E-12, 1.27997891179943299903E-9, 4.41827842801218905784E-7,
    9.96412122043875552487E-5, 1.42085239326149893930E-2, 9.99999999999999996984E-1};
__device__ static const double CN[6] = {
    2.02524002389102268789E-11, -1.35249504915790756375E-8, 3.59325051419993077021E-6,
    -4.74007206873407909465E-4, 2.89159652607555242092E-2, -1.00000000000000000080E0};
__device__ static const double CD[6] = {
    4.07746040061880559506E-12, 3.06780997581887812692E-9, 1.23210355685883423679E-6,
    3.17442024775032769882E-4, 5.10028056236446052392E-2, 4.00000000000000000080E0};
__device__ static const double FN4[7] = {
    4.23612862892216586994E0, 5.45937717161812843388E0, 1.62083287701538329132E0,
    1.67006611831323023771E-1, 6.81020132472518137426E-3, 1.08936580650328664411E-4,
    5.48900223421373614008E-7};
__device__ static const double FD4[7] = {
    8.16496634205391016773E0, 7.30828822505564552187E0, 1.86792257950184183883E0,
    1.78792052963149907262E-1, 7.01710668322789753610E-3, 1.10034357153915731354E-4,
    5.48900252756255700982E-7};
__device__ static const double FN8[9] = {
    4.55880873470465315206E-1, 7.13715274100146711374E-1, 1.60300158222319456320E-1,
    1.16064229408124407915E-2, 3.49556442447859055605E-4, 4.86215430826454749482E-6,
    3.20092790091004902806E-8, 9.41779576128512936592E-11, 9.70507110881952024631E-14};
__device__ static const double FD8[8] = {
    9.17463611873684053703E-1, 1.78685545332074536321E-1, 1.22253594771971293032E-2,
    3.58696481881851580297E-4, 4.92435064317881464393E-6, 3.21956939101046018377E-8,
    9.43720590350276732376E-11, 9.70507110881952025725E-14};
__device__ static const double GN4[8] = {
    8.71001698973114191777E-2, 6.11379109952219284151E-1, 3.97180296392337498885E-1,
    7.48527737628469092119E-2, 5.38868681462177273157E-3, 1.61999794598934024525E-4,
    1.97963874140963632189E-6, 7.82579040744090311069E-9};
__device__ static const double GD4[7] = {
    1.64402202413355338886E0, 6.66296701268987968381E-1, 9.88771761277688796203E-2,
    6.22396345441768420760E-3, 1.73221081474177119497E-4, 2.02659182086343991969E-6,
    7.82579218933534490868E-9};
__device__ static const double GN8[9] = {
    6.97359953443276214934E-1, 3.30410979305632063225E-1, 3.84878767649974295920E-2,
    1.71718239052347903558E-3, 3.48941165502279436777E-5, 3.47131167084116673800E-7,
    1.70404452782044526189E-9, 3.85945925430276600453E-12, 3.14040098946363334640E-15};
__device__ static const double GD8[9] = {
    1.68548898811011640017E0, 4.87852258695304967486E-1, 4.67913194259625806320E-2,
    1.90284426674399523638E-3, 3.68475504442561108162E-5, 3.57043223443740838771E-7,
    1.72693748966316146736E-9, 3.87830166023954706752E-12, 3.14040098946363335242E-15};

// Horner, leading coefficient first, degree = n-1.
template <int N>
__device__ __forceinline__ double horner(double x, const double (&c)[N]) {
#pragma clang fp contract(off)
    double a = c[0];
#pragma unroll
    for (int i = 1; i < N; ++i) a = a * x + c[i];
    return a;
}
// Same with an implicit leading coefficient of 1 (degree = N).
template <int N>
__device__ __forceinline__ double horner1(double x, const double (&c)[N]) {
#pragma clang fp contract(off)
    double a = x + c[0];
#pragma unroll
    for (int i = 1; i < N; ++i) a = a * x + c[i];
    return a;
}

constexpr double EULER_GAMMA = 0.577215664901532860606512090082402431;
constexpr double HALF_PI = 1.57079632679489661923;

// x > 0 finite.  sx, cx = sin(x), cos(x) are only read on the x > 4 branch, so the
// caller can share them with its own trigonometry.
__device__ __forceinline__ void sici_pos(double x, double sx, double cx, double& si, double& ci) {
#pragma clang fp contract(off)
    if (x <= 4.0) {
        const double z = x * x;
        si = x * horner(z, SN) / horner(z, SD);
        const double c = z * horner(z, CN) / horner(z, CD);
        ci = EULER_GAMMA + log(x) + c;
        return;
    }
    const double z = 1.0 / (x * x);
    double f, g;
    if (x < 8.0) {
        f = horner(z, FN4) / (x * horner1(z, FD4));
        g = z * horner(z, GN4) / horner1(z, GD4);
    } else {
        f = horner(z, FN8) / (x * horner1(z, FD8));
        g = z * horner(z, GN8) / horner1(z, GD8);
    }
    si = HALF_PI - f * cx - g * sx;
    ci = f * sx - g * cx;
}

}  // namespace hmg
