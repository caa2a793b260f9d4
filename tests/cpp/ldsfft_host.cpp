// Host harness for hmvec_amd/csrc/ldsfft.hpp: runs the workgroup FFT "thread by thread" on the
// CPU exactly as the fused GPU kernel sequences it (all loads of a pass, barrier, all stores).
// Built by tests/test_ldsfft_cpu.py with g++; not part of the product.
#include <cmath>
#include <vector>
#include "../../hmvec_amd/csrc/ldsfft.hpp"

using namespace hmg;

template <int R>
static void run_pass(std::vector<cplx>& buf, const std::vector<cplx>& tw, const FftPlanDev& plan, int p, int nthreads) {
    const int M = plan.M, Ns = plan.ns[p], nb = M / R;
    std::vector<cplx> regs((size_t)nb * R);
    for (int tid = 0; tid < nthreads; ++tid)             // load half
        for (int j = tid; j < nb; j += nthreads)
            pass_load<R>(buf.data(), tw.data(), M, Ns, plan.twstep[p], plan.magic[p], j, &regs[(size_t)j * R]);
    for (int tid = 0; tid < nthreads; ++tid)             // (barrier) store half
        for (int j = tid; j < nb; j += nthreads) pass_store<R>(buf.data(), Ns, plan.magic[p], j, &regs[(size_t)j * R]);
}

extern "C" int ldsfft_rfft_imag(const double* y, int n, int nthreads, double* imF /* n/2+1 */) {
    if (n % 2) return 1;
    const int M = n / 2;
    FftPlanDev plan;
    if (!fft_make_plan(M, &plan)) return 2;
    std::vector<cplx> buf(M), tw(M);
    const long double twopi = 6.283185307179586476925286766559L;
    for (int t = 0; t < M; ++t) tw[t] = {(double)cosl(twopi * t / M), (double)-sinl(twopi * t / M)};
    for (int m = 0; m < M; ++m) buf[m] = {y[2 * m], y[2 * m + 1]};
    for (int p = 0; p < plan.npass; ++p) {
        switch (plan.radix[p]) {
            case 2: run_pass<2>(buf, tw, plan, p, nthreads); break;
            case 3: run_pass<3>(buf, tw, plan, p, nthreads); break;
            case 4: run_pass<4>(buf, tw, plan, p, nthreads); break;
            case 5: run_pass<5>(buf, tw, plan, p, nthreads); break;
            default: return 3;
        }
    }
    imF[0] = 0.0;
    imF[M] = 0.0;
    for (int j = 1; j <= M / 2; ++j) {
        const double th = (double)(twopi * j / n);
        double a, b;
        unpack_imag_pair(buf[j], buf[M - j], cos(th), sin(th), a, b);
        imF[j] = a;
        imF[M - j] = b;
    }
    return 0;
}
