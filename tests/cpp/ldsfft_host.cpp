// Host harness for hmvec_amd/csrc/ldsfft.hpp: runs the workgroup FFT "thread by thread" on the
// CPU exactly as the fused GPU kernel sequences it (all loads of a pass, barrier, all stores).
// Built by tests/test_ldsfft_cpu.py with g++; not part of the product.
#include <algorithm>
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

// The compile-time plan of the fused profile kernel for nxs = 5000 (M = 2500, passes 4,5,5,5,5), sequenced as the
// kernel sequences it for a row that is zero from packed sample `nz_from` on:
//   nz_from <= 375: pruned first pass, samples 0..374 only, 3-of-5 butterflies reading the compact source;
//   nz_from <= 625: pruned first pass, full butterflies reading the compact source;
//   otherwise     : all five passes.
// Everything runs the 24-bit index arithmetic (SMALL).  Returns the same Im F_j as ldsfft_rfft_imag.
template <int R, int NIN, int SRC>
static void run_spec_pass(std::vector<cplx>& buf, const std::vector<cplx>& tw, int Ns, int twstep, int nthreads) {
    const int M = 2500, nb = M / R;
    const unsigned mg = small_magic((unsigned)Ns);
    std::vector<cplx> regs((size_t)nb * R);
    for (int tid = 0; tid < nthreads; ++tid)
        for (int j = tid; j < nb; j += nthreads)
            pass_load<R, true, NIN, SRC>(buf.data(), tw.data(), M, Ns, twstep, mg, j, &regs[(size_t)j * R]);
    for (int tid = 0; tid < nthreads; ++tid)
        for (int j = tid; j < nb; j += nthreads) pass_store<R, true, NIN>(buf.data(), Ns, mg, j, &regs[(size_t)j * R]);
}
extern "C" int ldsfft_rfft_imag_spec2500(const double* y, int nz_from, int nthreads, double* imF /* 2501 */) {
    const int M = 2500, n = 5000;
    std::vector<cplx> tw(M);
    const long double twopi = 6.283185307179586476925286766559L;
    for (int t = 0; t < M; ++t) tw[t] = {(double)cosl(twopi * t / M), (double)-sinl(twopi * t / M)};
    // stale values where the kernel leaves LDS untouched: a wrong read shows up in the result
    std::vector<cplx> buf(M, cplx{1.0e30, -1.0e30});
    const bool pruned = nz_from <= 625, lead3 = nz_from <= 375;
    const int pend = lead3 ? 375 : (pruned ? 625 : M);
    for (int m = 0; m < pend; ++m) buf[m] = {y[2 * m], y[2 * m + 1]};
    if (!pruned) run_spec_pass<4, 4, 0>(buf, tw, 1, 625, nthreads);
    if (lead3) run_spec_pass<5, 3, 2>(buf, tw, 4, 125, nthreads);
    else if (pruned) run_spec_pass<5, 5, 2>(buf, tw, 4, 125, nthreads);
    else run_spec_pass<5, 5, 0>(buf, tw, 4, 125, nthreads);
    run_spec_pass<5, 5, 0>(buf, tw, 20, 25, nthreads);
    run_spec_pass<5, 5, 0>(buf, tw, 100, 5, nthreads);
    run_spec_pass<5, 5, 0>(buf, tw, 500, 1, nthreads);
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

// The pruned decomposition of the long-grid route (ldsfft.hpp, "Long radial grids with short support"), sequenced as
// profile_pruned_kernel sequences it: per group of residues {g, R-g} the twiddled copies of the row, the passes of
// the compile-time plan of length LP over both buffers (all loads, barrier, all stores), the unpack step.
// y must be zero from real sample 2 LP on.  Writes Im F_j = u_j * j for the needed modes, NaN elsewhere.
template <int LP, int PS>
static void run_sub_passes(std::vector<cplx>& buf, const std::vector<cplx>& twL, int nbuf, int keep, int nthreads) {
    if constexpr (PS < SubPass<LP, 0>::P.npass) {
        using S = SubPass<LP, PS>;
        const int maxb = (2 * S::nb + nthreads - 1) / nthreads;
        std::vector<cplx> regs((size_t)nthreads * maxb * S::R);
        for (int tid = 0; tid < nthreads; ++tid)
            for (int b = 0; b < maxb; ++b) {
                const int jj = tid + b * nthreads;
                if (sub_pass_active<LP, PS>(jj, nbuf, keep))
                    sub_pass_load<LP, PS>(buf.data(), twL.data(), jj, &regs[((size_t)tid * maxb + b) * S::R]);
            }
        for (int tid = 0; tid < nthreads; ++tid)
            for (int b = 0; b < maxb; ++b) {
                const int jj = tid + b * nthreads;
                if (sub_pass_active<LP, PS>(jj, nbuf, keep))
                    sub_pass_store<LP, PS>(buf.data(), jj, &regs[((size_t)tid * maxb + b) * S::R]);
            }
        run_sub_passes<LP, PS + 1>(buf, twL, nbuf, keep, nthreads);
    }
}
template <int LP>
static int pruned_rfft_imag(const double* y, int n, int nthreads, int jn, double* imF) {
    const int M = n / 2;
    if (n % 2 || M % LP || M / LP < 2) return 2;
    const int R = M / LP;
    for (int i = 2 * LP; i < n; ++i)
        if (y[i] != 0.0) return 4;
    const long double twopi = 6.283185307179586476925286766559L;
    (void)twopi;
    std::vector<cplx> src(LP), buf(2 * (size_t)LP, cplx{1.0e30, -1.0e30});
    // the tables as the library lays them out: twiddles per pass, residue twiddles and unpack constants by residue
    const std::vector<cplx> twL = pass_tw_table(SubPass<LP, 0>::P), twR = residue_tw_table(M, LP);
    const std::vector<UnpackTw> twNr = residue_unpack_table(M, LP);
    const unsigned rmagic = (unsigned)(4294967296ull / (unsigned)R) + 1u;
    for (int p = 0; p < LP; ++p) src[p] = {y[2 * p], y[2 * p + 1]};
    std::vector<double> u(M, NAN);
    constexpr int nb_last = SubPass<LP, SubPass<LP, 0>::P.npass - 1>::nb;
    for (int g = 0; g <= R / 2; ++g) {
        if (!pruned_group_needed(R, M, g, jn)) continue;
        const int s1 = pruned_group_partner(R, g), nbuf = s1 < 0 ? 1 : 2;
        // first pass (radix R0, sub-transform size 1) straight from the samples a thread owns: j + t LP/R0
        {
            using S0 = SubPass<LP, 0>;
            for (int tid = 0; tid < nthreads; ++tid)
                for (int jb = tid; jb < S0::nb; jb += nthreads)
                    for (int hb = 0; hb < nbuf; ++hb) {
                        const int sres = hb ? s1 : g;
                        cplx v[S0::R];
                        for (int t = 0; t < S0::R; ++t) v[t] = cmul(src[jb + t * S0::nb], twR[(size_t)sres * LP + jb + t * S0::nb]);
                        dft_small<S0::R>(v);
                        for (int t = 0; t < S0::R; ++t) buf[hb * LP + jb * S0::R + t] = v[t];
                    }
        }
        run_sub_passes<LP, 1>(buf, twL, nbuf, pruned_keep(R, M, nb_last, jn), nthreads);
        for (int tid = 0; tid < nthreads; ++tid) {
            pruned_unpack(buf.data(), LP, R, M, g, 0, nbuf == 2 ? 1 : 0, jn, twNr.data(), 1.0, u.data(), tid, nthreads);
            if (nbuf == 2) pruned_unpack(buf.data(), LP, R, M, s1, 1, 0, jn, twNr.data(), 1.0, u.data(), tid, nthreads);
        }
    }
    imF[0] = 0.0;
    imF[M] = 0.0;
    for (int j = 1; j < M; ++j) imF[j] = u[pruned_u_index(R, LP, rmagic, j)] * j;      // the line is laid out by residue
    return 0;
}
extern "C" int ldsfft_pruned_rfft_imag(const double* y, int n, int LP, int nthreads, int jn, double* imF /* n/2+1 */) {
    switch (LP) {
        case 1000: return pruned_rfft_imag<1000>(y, n, nthreads, jn, imF);
        case 1024: return pruned_rfft_imag<1024>(y, n, nthreads, jn, imF);
        case 1250: return pruned_rfft_imag<1250>(y, n, nthreads, jn, imF);
        case 1500: return pruned_rfft_imag<1500>(y, n, nthreads, jn, imF);
        case 2000: return pruned_rfft_imag<2000>(y, n, nthreads, jn, imF);
        case 2048: return pruned_rfft_imag<2048>(y, n, nthreads, jn, imF);
        case 2500: return pruned_rfft_imag<2500>(y, n, nthreads, jn, imF);
        default: return 3;
    }
}

// The chirp route of the long-grid kernel (ldsfft.hpp, "Rows that need FEW modes"), sequenced as profile_pruned_row
// sequences it for a row with jn <= Jw: first pass of the chirped row from the samples a thread owns, the other
// passes of the length-Lc plan, the product with the tabulated chirp-window transform fused into the first pass of
// the second transform, its other passes (last one pruned), unpack.  Lc = 2 LP.
template <int LC, int PS>
static void run_single_passes(std::vector<cplx>& buf, const std::vector<cplx>& tw, int keep, int nthreads) {
    if constexpr (PS < SubPass<LC, 0>::P.npass) {
        using S = SubPass<LC, PS>;
        const int maxb = (S::nb + nthreads - 1) / nthreads;
        std::vector<cplx> regs((size_t)nthreads * maxb * S::R);
        for (int tid = 0; tid < nthreads; ++tid)
            for (int b = 0; b < maxb; ++b) {
                const int jj = tid + b * nthreads;
                if (sub_pass_active<LC, PS>(jj, 1, keep)) sub_pass_load<LC, PS>(buf.data(), tw.data(), jj, &regs[((size_t)tid * maxb + b) * S::R]);
            }
        for (int tid = 0; tid < nthreads; ++tid)
            for (int b = 0; b < maxb; ++b) {
                const int jj = tid + b * nthreads;
                if (sub_pass_active<LC, PS>(jj, 1, keep)) sub_pass_store<LC, PS>(buf.data(), jj, &regs[((size_t)tid * maxb + b) * S::R]);
            }
        run_single_passes<LC, PS + 1>(buf, tw, keep, nthreads);
    }
}
template <int LP>
static int chirp_rfft_imag(const double* y, int n, int p0, int nwin, int nthreads, int jn, double* imF) {
    constexpr int LC = 2 * LP;
    using C0 = SubPass<LC, 0>;
    static_assert(C0::R == 4 && C0::Ns == 1, "first pass of the chirp transforms is radix 4");
    const int M = n / 2;
    if (n % 2 || p0 > LP) return 2;
    for (int i = 2 * p0; i < n; ++i)
        if (y[i] != 0.0) return 4;
    const ChirpTables T = chirp_make_tables(M, LC, p0, nwin);
    if (jn > T.Jw + T.nwin * T.Kp) return 5;
    const long double twopi = 6.283185307179586476925286766559L;
    std::vector<cplx> buf(LC, cplx{1.0e30, -1.0e30});
    const std::vector<cplx> tw = pass_tw_table(C0::P);
    constexpr int nb0 = C0::nb;                          // = LP / 2
    constexpr int nb_last = SubPass<LC, C0::P.npass - 1>::nb;
    for (int jb = 0; jb < nb0; ++jb) {
        cplx v[4];
        const cplx z0 = {y[2 * jb], y[2 * jb + 1]}, z1 = {y[2 * (jb + nb0)], y[2 * (jb + nb0) + 1]};
        chirp_first_pass(z0, z1, T.chP[jb], T.chP[jb + nb0], v);
        for (int t = 0; t < 4; ++t) buf[4 * jb + t] = v[t];
    }
    run_single_passes<LC, 1>(buf, tw, -1, nthreads);
    const std::vector<cplx> A = buf;                     // the forward transform: every window multiplies it again
    auto second = [&](const cplx* Bw, int keep) {
        std::vector<cplx> regs((size_t)nb0 * 4);
        for (int jb = 0; jb < nb0; ++jb)
            for (int t = 0; t < 4; ++t) regs[4 * jb + t] = cmul(A[jb + t * nb0], Bw[jb + t * nb0]);
        for (int jb = 0; jb < nb0; ++jb) {
            dft_small<4>(&regs[4 * jb]);
            for (int t = 0; t < 4; ++t) buf[4 * jb + t] = regs[4 * jb + t];
        }
        run_single_passes<LC, 1>(buf, tw, keep, nthreads);
    };
    auto unpack = [&](int j, cplx zj, cplx zmj) {
        const long double th = twopi * j / n;
        double fa, fb;
        unpack_imag_pair(zj, zmj, (double)cosl(th), (double)sinl(th), fa, fb);
        imF[j] = fa;
    };
    for (int j = 0; j <= M; ++j) imF[j] = NAN;
    const int jc = jn < T.Jw ? jn : T.Jw;
    second(T.Bw.data(), (2 * jc + 2 < nb_last) ? jc : -1);
    for (int j = 1; j <= jc; ++j) {
        const UnpackTw w{0.0, 0.0, 1.0 / j, 1.0 / (M - j)};
        (void)w;
        unpack(j, cmul(T.chJ[j], buf[LC - j]), cmul(T.chJ[j], buf[j]));
    }
    std::vector<cplx> stash(T.Kp);
    for (int w = 1; w <= T.nwin && T.Jw + 1 + (w - 1) * T.Kp <= jn; ++w) {
        const int j0 = T.Jw + 1 + (w - 1) * T.Kp, j1 = std::min(jn, j0 + T.Kp - 1);
        second(T.Bw.data() + (size_t)(2 * w - 1) * LC, -1);
        for (int j = j0; j <= j1; ++j) stash[j - j0] = chirp_plus(buf.data(), T.Kp, j0, j, T.chJ[j]);
        second(T.Bw.data() + (size_t)(2 * w) * LC, -1);
        for (int j = j0; j <= j1; ++j) unpack(j, stash[j - j0], chirp_minus(buf.data(), j0, j, T.chJ[j]));
    }
    imF[0] = 0.0;
    return 0;
}
extern "C" int ldsfft_chirp_rfft_imag(const double* y, int n, int LP, int p0, int nwin, int nthreads, int jn, double* imF) {
    switch (LP) {
        case 1000: return chirp_rfft_imag<1000>(y, n, p0, nwin, nthreads, jn, imF);
        case 1250: return chirp_rfft_imag<1250>(y, n, p0, nwin, nthreads, jn, imF);
        default: return 3;
    }
}
extern "C" int ldsfft_chirp_window(int n, int LP, int p0, int nwin) {
    const ChirpTables T = chirp_make_tables(n / 2, 2 * LP, p0, nwin);
    return T.Jw + T.nwin * T.Kp;
}

// The narrow-band route (ldsfft.hpp, "Rows whose support does NOT prune"), sequenced as profile_band_kernel sequences
// it: per residue p1 of the sample index the first pass from the samples a thread owns (p2 = j + t LB/R0), the other
// passes of the length-LB plan (last one pruned to the band), the accumulation with the running twiddle; unpack.
template <int LB>
static int band_rfft_imag(const double* y, int n, int nthreads, int jn, double* imF) {
    using S0 = SubPass<LB, 0>;
    const int M = n / 2;
    if (n % 2 || M % LB) return 2;
    const int D = M / LB;
    if (2 * jn + 2 > LB) return 5;
    const long double twopi = 6.283185307179586476925286766559L;
    std::vector<cplx> twB(M), buf(LB, cplx{1.0e30, -1.0e30});
    const std::vector<cplx> twL = pass_tw_table(S0::P);
    for (int t = 0; t < M; ++t) twB[t] = {(double)cosl(twopi * t / M), (double)-sinl(twopi * t / M)};
    const int nacc = 2 * jn + 1;
    std::vector<cplx> acc(nacc, cplx{0.0, 0.0});
    const std::vector<cplx> twR = residue_tw_table(M, LB);       // W_M^(p1 j) at [p1 LB + j], as the kernel reads it
    (void)twB;
    constexpr int nb_last = SubPass<LB, S0::P.npass - 1>::nb;
    const int keep = (2 * jn + 2 < nb_last) ? jn : -1;
    for (int p1 = 0; p1 < D; ++p1) {
        for (int jb = 0; jb < S0::nb; ++jb) {
            cplx v[S0::R];
            for (int t = 0; t < S0::R; ++t) {
                const int p = p1 + D * (jb + t * S0::nb);
                v[t] = cplx{y[2 * p], y[2 * p + 1]};
            }
            dft_small<S0::R>(v);
            for (int t = 0; t < S0::R; ++t) buf[jb * S0::R + t] = v[t];
        }
        run_single_passes<LB, 1>(buf, twL, keep, nthreads);
        for (int t = 0; t < nacc; ++t) {
            const int j = band_mode(t, jn), ja = j < 0 ? -j : j;
            const cplx yv = buf[band_index(j, LB)];
            cplx w = twR[(size_t)p1 * LB + ja];
            if (j < 0) w.y = -w.y;
            acc[t] = cadd(acc[t], cmul(yv, w));
        }
    }
    for (int j = 0; j <= M; ++j) imF[j] = NAN;
    imF[0] = 0.0;
    for (int j = 1; j <= jn; ++j) {
        const long double th = twopi * j / n;
        double fa, fb;
        unpack_imag_pair(acc[jn + j], acc[jn - j], (double)cosl(th), (double)sinl(th), fa, fb);
        imF[j] = fa;
    }
    return 0;
}
extern "C" int ldsfft_band_rfft_imag(const double* y, int n, int LB, int nthreads, int jn, double* imF) {
    switch (LB) {
        case 1000: return band_rfft_imag<1000>(y, n, nthreads, jn, imF);
        case 1024: return band_rfft_imag<1024>(y, n, nthreads, jn, imF);
        case 1250: return band_rfft_imag<1250>(y, n, nthreads, jn, imF);
        default: return 3;
    }
}
