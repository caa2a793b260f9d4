// libhmgrid — MI355X (gfx950 / CDNA4) kernels + C ABI for the halo-model grid hot path.
// Boundary and reference citations: include/hmgrid.h.  Design notes: DESIGN.md.
//
// Everything here is fp64 and either HBM-bandwidth or fp64-VALU bound; the one dense contraction
// of the path (sigma^2: a (z x k') . (k' x m) product) runs on the fp64 matrix cores
// (v_mfma_f64_16x16x4_f64).  Layout is [z][m][k] with k fastest: a wavefront (64 lanes) always
// walks consecutive k, so every tensor access is a fully coalesced 512 B (or 1 KiB with double2)
// wave transaction, and per-(z,m) scalars are wave-uniform.
//
// This translation unit holds every kernel of the headline path (kernels/*.hpp, included below in dependency order) and
// the C-ABI entry points that validate arguments and launch them.  Contexts, memory, events, lanes and captured steps:
// runtime.hip; RCCL: comm.hip; the long-radial-grid row kernels: longgrid.hip (a unit of its own: in one unit their
// instantiations changed the code hipcc emits for the hot profile kernel).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "hmctx.hpp"
#include "fastmath.hpp"
#include "rowdev.hpp"
#include "longgrid.hpp"

#include "kernels/common.hpp"
#include "kernels/sigma2.hpp"
#include "kernels/massfn.hpp"
#include "kernels/halo.hpp"
#include "kernels/nfw.hpp"
#include "kernels/rows.hpp"
#include "kernels/profile_chain.hpp"
#include "kernels/profile_fused.hpp"
#include "kernels/hod.hpp"
#include "kernels/power.hpp"
#include "kernels/power_batch.hpp"
#include "kernels/groups.hpp"
#include "kernels/limber.hpp"
#include "kernels/functions.hpp"

// ------------------------------------------------------------------------------------------
// C ABI: launch entry points (definitions inherit C linkage from the declarations in hmgrid.h)
// ------------------------------------------------------------------------------------------
using namespace hmg;

static inline int sigma2_ztile(int nz) { return nz > 16 ? 32 : 16; }
static inline int sigma2_nzp(int nz) { const int t = sigma2_ztile(nz); return (nz + t - 1) / t * t; }

int hmg_sigma2_layout_size(int nz, int nq, size_t* doubles) {
    REQUIRE(doubles && nz > 0 && nq > 0, "bad argument");
    *doubles = (size_t)nq * sigma2_nzp(nz);
    return 0;
}
int hmg_sigma2_prepare(hmg_ctx* c, int nz, int nq, const double* sP, double* PT) {
    REQUIRE(c && sP && PT, "NULL argument");
    REQUIRE(nz > 0 && nq > 0, "empty grid");
    const int nzp = sigma2_nzp(nz);
    hipLaunchKernelGGL(transpose_pad_kernel, grid1d((size_t)nzp * nq, 256), dim3(256), 0, c->stream, nz, nzp,
                       nq, sP, PT);
    HIP_TRY(hipGetLastError());
    return 0;
}
int hmg_sigma2_prepared(hmg_ctx* c, int nz, int nm, int nq, const double* PT, const double* kq,
                        const double* wq, const double* R, double tswitch, double* out) {
    REQUIRE(c && PT && kq && wq && R && out, "NULL argument");
    REQUIRE(nz > 0 && nm > 0 && nq > 0, "empty grid");
    const int nseg = (nq + SIG_SEG_LEN - 1) / SIG_SEG_LEN;
    const int ztile = sigma2_ztile(nz), nzp = sigma2_nzp(nz);
    if (ensure_scratch(c, 4, (size_t)nseg * nz * nm * 8)) return 1;
    double* partial = (double*)c->scratch[4];
    dim3 grid((nm + 15) / 16, nseg, nzp / ztile);
    REQUIRE(grid.y <= 65535 && grid.z <= 65535, "grid too large");
    if (ztile == 32)
        hipLaunchKernelGGL(sigma2_mfma_kernel<2>, grid, dim3(64), 0, c->stream, nz, nzp, nm, nq,
                           PT, kq, wq, R, tswitch, partial);
    else
        hipLaunchKernelGGL(sigma2_mfma_kernel<1>, grid, dim3(64), 0, c->stream, nz, nzp, nm, nq,
                           PT, kq, wq, R, tswitch, partial);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(sigma2_combine_kernel, grid1d((size_t)nz * nm, 64), dim3(256), 0, c->stream,
                       nz * nm, nseg, (const double*)partial, out);
    HIP_TRY(hipGetLastError());
    return 0;
}
// first stage of the contraction for hmg_sigma2_massfn / hmg_sigma2_massfn_halo: launches the matrix-core
// kernel, returns the partial sums' buffer and the number of k' segments
static int sigma2_partials(hmg_ctx* c, int nz, int nm, int nq, const double* PT, const double* kq, const double* wq,
                           const double* R, double tswitch, const double** partial_out, int* nseg_out) {
    const int nseg = (nq + SIG_SEG_LEN - 1) / SIG_SEG_LEN;
    const int ztile = sigma2_ztile(nz), nzp = sigma2_nzp(nz);
    if (ensure_scratch(c, 4, (size_t)nseg * nz * nm * 8)) return 1;
    double* partial = (double*)c->scratch[4];
    dim3 grid((nm + 15) / 16, nseg, nzp / ztile);
    REQUIRE(grid.y <= 65535 && grid.z <= 65535, "grid too large");
    if (ztile == 32)
        hipLaunchKernelGGL(sigma2_mfma_kernel<2>, grid, dim3(64), 0, c->stream, nz, nzp, nm, nq,
                           PT, kq, wq, R, tswitch, partial);
    else
        hipLaunchKernelGGL(sigma2_mfma_kernel<1>, grid, dim3(64), 0, c->stream, nz, nzp, nm, nq,
                           PT, kq, wq, R, tswitch, partial);
    HIP_TRY(hipGetLastError());
    c->sig_nz = nz; c->sig_nm = nm; c->sig_nq = nq;
    *partial_out = partial;
    *nseg_out = nseg;
    return 0;
}
static int sigma2_massfn_check(hmg_ctx* c, int nz, int nm, int nq, const double* PT, const double* kq,
                               const double* wq, const double* R, const hmg_massfn_params* p, const double* ms,
                               const double* lnms, const double* tz, double* sigma2, double* nzm, double* bh) {
    REQUIRE(c && PT && kq && wq && R && p && ms && lnms && sigma2 && nzm && bh, "NULL argument");
    REQUIRE(nz > 0 && nm > 0 && nq > 0, "empty grid");
    REQUIRE(p->mode == HMG_MF_SHETH_TORMEN || p->mode == HMG_MF_TINKER10, "unknown mass function");
    REQUIRE(p->mode != HMG_MF_TINKER10 || tz, "Tinker mode needs d_tinker_z");
    REQUIRE(nz <= 65535, "nz too large");
    return 0;
}
int hmg_sigma2_massfn(hmg_ctx* c, int nz, int nm, int nq, const double* PT, const double* kq, const double* wq,
                      const double* R, double tswitch, const hmg_massfn_params* p, const double* ms,
                      const double* lnms, const double* tz, double* sigma2, double* nzm, double* bh) {
    if (sigma2_massfn_check(c, nz, nm, nq, PT, kq, wq, R, p, ms, lnms, tz, sigma2, nzm, bh)) return 1;
    const double* partial;
    int nseg;
    if (sigma2_partials(c, nz, nm, nq, PT, kq, wq, R, tswitch, &partial, &nseg)) return 1;
    MassFnDev P{p->mode, p->deltac, p->st_A, p->st_a, p->st_p, p->rho_m0, p->lnm_uniform, p->lnm_step};
    SigmaMassFnArgs A{nz, nm, nseg, P, partial, ms, lnms, tz, sigma2, nzm, bh};
    hipLaunchKernelGGL(sigma2_massfn_kernel, dim3((nm + 63) / 64, nz), dim3(512), 0, c->stream, A);
    HIP_TRY(hipGetLastError());
    return 0;
}
static int halo_stage_check(hmg_ctx* c, int nz, int nm, const double* ms, const hmg_halo_stage_args* h,
                            HaloStageArgs* H) {
    REQUIRE(c && ms && h && h->d_zs && h->d_delta && h->d_rho && h->d_cs && h->d_rvir && h->d_rs, "NULL argument");
    REQUIRE(nz > 0 && nm > 0, "empty grid");
    REQUIRE((h->d_m2 == nullptr) == (h->d_r2 == nullptr), "pass both d_m2 and d_r2 or neither");
    REQUIRE(!h->d_m2 || (h->d_drho1 && h->d_rho2), "the mass conversion needs d_drho1 and d_rho2");
    *H = HaloStageArgs{nz, nm, ms, h->d_zs, h->d_delta, h->d_rho, h->duffy_A, h->duffy_alpha, h->duffy_beta, h->h,
                       h->d_cs, h->d_rvir, h->d_rs, h->d_nfw_series, h->d_drho1, h->delta2, h->d_rho2, h->d_m2,
                       h->d_r2};
    return 0;
}
int hmg_sigma2_massfn_halo(hmg_ctx* c, int nz, int nm, int nq, const double* PT, const double* kq,
                           const double* wq, const double* R, double tswitch, const hmg_massfn_params* p,
                           const double* ms, const double* lnms, const double* tz, double* sigma2, double* nzm,
                           double* bh, const hmg_halo_stage_args* h) {
    if (sigma2_massfn_check(c, nz, nm, nq, PT, kq, wq, R, p, ms, lnms, tz, sigma2, nzm, bh)) return 1;
    HaloStageArgs H;
    if (halo_stage_check(c, nz, nm, ms, h, &H)) return 1;
    const double* partial;
    int nseg;
    if (sigma2_partials(c, nz, nm, nq, PT, kq, wq, R, tswitch, &partial, &nseg)) return 1;
    MassFnDev P{p->mode, p->deltac, p->st_A, p->st_a, p->st_p, p->rho_m0, p->lnm_uniform, p->lnm_step};
    SigmaMassFnArgs A{nz, nm, nseg, P, partial, ms, lnms, tz, sigma2, nzm, bh};
    hipLaunchKernelGGL(ctor_stage_kernel, dim3((nm + 63) / 64, nz, 2), dim3(512), 0, c->stream, A, H);
    HIP_TRY(hipGetLastError());
    return 0;
}

int hmg_sigma2(hmg_ctx* c, int nz, int nm, int nq, const double* sP, const double* kq,
               const double* wq, const double* R, double tswitch, double* out) {
    REQUIRE(c && sP && kq && wq && R && out, "NULL argument");
    REQUIRE(nz > 0 && nm > 0 && nq > 0, "empty grid");
    if (ensure_scratch(c, 6, (size_t)nq * sigma2_nzp(nz) * 8)) return 1;
    double* PT = (double*)c->scratch[6];
    if (hmg_sigma2_prepare(c, nz, nq, sP, PT)) return 1;
    return hmg_sigma2_prepared(c, nz, nm, nq, PT, kq, wq, R, tswitch, out);
}

int hmg_massfn(hmg_ctx* c, int nz, int nm, const hmg_massfn_params* p, const double* s2,
               const double* ms, const double* lnms, const double* tz, double* nzm, double* bh) {
    REQUIRE(c && p && s2 && ms && lnms && nzm && bh, "NULL argument");
    REQUIRE(nz > 0 && nm > 0, "empty grid");
    REQUIRE(p->mode == HMG_MF_SHETH_TORMEN || p->mode == HMG_MF_TINKER10, "unknown mass function");
    REQUIRE(p->mode != HMG_MF_TINKER10 || tz, "Tinker mode needs d_tinker_z");
    MassFnDev P{p->mode, p->deltac, p->st_A, p->st_a, p->st_p, p->rho_m0, p->lnm_uniform, p->lnm_step};
    hipLaunchKernelGGL(massfn_kernel, grid1d((size_t)nz * nm, 256), dim3(256), 0, c->stream, nz, nm,
                       P, s2, ms, lnms, tz, nzm, bh);
    HIP_TRY(hipGetLastError());
    return 0;
}

int hmg_halo_structure(hmg_ctx* c, int nz, int nm, const double* ms, const double* zs,
                       const double* delta, const double* rho, double A, double alpha, double beta,
                       double h, double* cs, double* rv, double* rs) {
    REQUIRE(c && ms && zs && delta && rho && cs && rv && rs, "NULL argument");
    REQUIRE(nz > 0 && nm > 0, "empty grid");
    hipLaunchKernelGGL(halo_structure_kernel, grid1d((size_t)nz * nm, 256), dim3(256), 0, c->stream,
                       nz, nm, ms, zs, delta, rho, A, alpha, beta, h, cs, rv, rs);
    HIP_TRY(hipGetLastError());
    return 0;
}

int hmg_halo_stage(hmg_ctx* c, int nz, int nm, const double* ms, const double* zs, const double* delta,
                   const double* rho, double A, double alpha, double beta, double h, double* cs, double* rv,
                   double* rs, double* nfw_series, const double* drho1, double delta2, const double* rho2,
                   double* m2, double* r2) {
    const hmg_halo_stage_args a{zs, delta, rho, A, alpha, beta, h, cs, rv, rs, nfw_series, drho1, delta2, rho2, m2, r2};
    HaloStageArgs H;
    if (halo_stage_check(c, nz, nm, ms, &a, &H)) return 1;
    hipLaunchKernelGGL(halo_stage_kernel, grid1d((size_t)nz * nm, 64), dim3(64), 0, c->stream, H);
    HIP_TRY(hipGetLastError());
    return 0;
}

int hmg_mdelta_convert(hmg_ctx* c, int nz, int nm, const double* ms, const double* cs,
                       const double* d1, double delta2, const double* rho2, double* m2, double* r2) {
    REQUIRE(c && ms && cs && d1 && rho2 && m2 && r2, "NULL argument");
    REQUIRE(nz > 0 && nm > 0, "empty grid");
    hipLaunchKernelGGL(mdelta_kernel, grid1d((size_t)nz * nm, 128), dim3(128), 0, c->stream, nz, nm,
                       ms, cs, d1, delta2, rho2, m2, r2);
    HIP_TRY(hipGetLastError());
    return 0;
}

int hmg_nfw_analytic(hmg_ctx* c, int nz, int nm, int nk, const double* cs, const double* rs,
                     const double* zs, const double* ks, const double* series, double* uk) {
    REQUIRE(c && cs && rs && zs && ks && uk, "NULL argument");
    REQUIRE(nz > 0 && nm > 0 && nk > 0, "empty grid");
    // 16 k per thread amortise the per-row prologue (a log, two divisions, the scalar loads of the
    // series row); smaller tiles were measured slower at every grid size once the series made the
    // per-point cost small
    int threads = 256, ktile = 4096;
    if (const char* e = getenv("HMG_NFW_THREADS")) threads = atoi(e);
    if (const char* e = getenv("HMG_NFW_KTILE")) ktile = atoi(e);
    REQUIRE(threads >= 64 && threads <= 256 && threads % 64 == 0, "HMG_NFW_THREADS must be 64/128/192/256");
    REQUIRE(ktile >= threads, "HMG_NFW_KTILE must not be smaller than the block size");
    const size_t blocks = (size_t)nz * nm * ((nk + ktile - 1) / ktile);
    REQUIRE(blocks <= 2147483647u, "grid too large");
    const double* acoef = series;
    if (!acoef) {      // the caller did not bring the series rows (hmg_halo_stage): build them here
        if (ensure_scratch(c, 5, (size_t)nz * nm * NFW_ROW * 8)) return 1;
        acoef = (const double*)c->scratch[5];
        hipLaunchKernelGGL(nfw_series_kernel, grid1d((size_t)nz * nm, 128), dim3(128), 0, c->stream, nz * nm, cs,
                           (double*)c->scratch[5]);
        HIP_TRY(hipGetLastError());
    }
    int stop = -1;
    if (bracket_open(c, HMG_KERNEL_NFW, &stop)) return 1;
    hipLaunchKernelGGL(nfw_kernel, dim3((unsigned)blocks), dim3(threads), 0, c->stream, c->d_sici,
                       acoef, ktile, nm, nk, cs, rs, zs, ks, uk);
    HIP_TRY(hipGetLastError());
    if (bracket_close(c, stop)) return 1;
    return 0;
}

int hmg_profile_rowparams(hmg_ctx* c, int kind, int nz, int nm, const double* m200, const double* r200,
                          const double* rvir, const double* zs, const double* rhoc, const double* hz,
                          const double f[9], double gamma, double alpha_const, double pref,
                          double post_pref, double* amp, double* xc, double* alpha, double* expo,
                          double* cmax, double* rscale, double* post) {
    REQUIRE(c && m200 && r200 && rvir && zs && rhoc && f && amp && xc && alpha && expo && cmax && rscale,
            "NULL argument");
    REQUIRE(kind == HMG_PROF_BATTAGLIA_GAS || kind == HMG_PROF_BATTAGLIA_PRES, "unknown profile kind");
    REQUIRE(kind != HMG_PROF_BATTAGLIA_PRES || (hz && post), "pressure needs d_hz and d_post");
    REQUIRE(nz > 0 && nm > 0, "empty grid");
    RowFit F;
    for (int i = 0; i < 9; ++i) F.f[i] = f[i];
    RowOut O{amp, xc, alpha, expo, cmax, rscale, post};
    hipLaunchKernelGGL(rowparams_kernel, grid1d((size_t)nz * nm, 128), dim3(128), 0, c->stream, kind,
                       nz, nm, m200, r200, rvir, zs, rhoc, hz, F, gamma, alpha_const, pref, post_pref, O);
    HIP_TRY(hipGetLastError());
    return 0;
}

int hmg_profile_rows_from_mvir(hmg_ctx* c, int kind, int nz, int nm, const double* ms, const double* cs,
                               const double* rvir, const double* zs, const double* drho1, double delta2,
                               const double* rhoc, const double* hz, const double f[9], double gamma,
                               double alpha_const, double pref, double post_pref, double* m200,
                               double* r200, double* amp, double* xc, double* alpha, double* expo,
                               double* cmax, double* rscale, double* post) {
    REQUIRE(c && ms && cs && rvir && zs && drho1 && rhoc && f && m200 && r200 && amp && xc && alpha && expo &&
                cmax && rscale, "NULL argument");
    REQUIRE(kind == HMG_PROF_BATTAGLIA_GAS || kind == HMG_PROF_BATTAGLIA_PRES, "unknown profile kind");
    REQUIRE(kind != HMG_PROF_BATTAGLIA_PRES || (hz && post), "pressure needs d_hz and d_post");
    REQUIRE(nz > 0 && nm > 0, "empty grid");
    RowFit F;
    for (int i = 0; i < 9; ++i) F.f[i] = f[i];
    RowOut O{amp, xc, alpha, expo, cmax, rscale, post};
    hipLaunchKernelGGL(rows_from_mvir_kernel, grid1d((size_t)nz * nm, 128), dim3(128), 0, c->stream, kind,
                       nz, nm, ms, cs, rvir, zs, drho1, delta2, rhoc, hz, F, gamma, alpha_const, pref,
                       post_pref, m200, r200, O);
    HIP_TRY(hipGetLastError());
    return 0;
}

static int get_plan(hmg_ctx* c, int nxs, int batch, FftPlan** out) {
    auto key = std::make_pair(nxs, batch);
    auto it = c->plans.find(key);
    if (it != c->plans.end()) { *out = &it->second; return 0; }
    // (plan creation compiles kernels at run time and allocates the work buffer: nothing a captured step may contain)
    REQUIRE(!c->capturing, "a rocFFT plan cannot be created inside a captured step: run the step once eagerly first");
    FftPlan P;
    size_t len = (size_t)nxs;
    FFT_TRY(rocfft_plan_create(&P.plan, rocfft_placement_notinplace, rocfft_transform_type_real_forward,
                               rocfft_precision_double, 1, &len, (size_t)batch, nullptr));
    FFT_TRY(rocfft_plan_get_work_buffer_size(P.plan, &P.work_bytes));
    FFT_TRY(rocfft_execution_info_create(&P.info));
    if (P.work_bytes) {
        HIP_TRY(hipMalloc(&P.work, P.work_bytes));
        FFT_TRY(rocfft_execution_info_set_work_buffer(P.info, P.work, P.work_bytes));
    }
    FFT_TRY(rocfft_execution_info_set_stream(P.info, c->stream));
    auto res = c->plans.emplace(key, P);
    *out = &res.first->second;
    return 0;
}

// (FUSED_NT, the threads per row workgroup of the fused profile kernels: rowdev.hpp)

// Workgroup-FFT tables for a given nxs; returns nullptr (no error) when the fused kernel
// cannot take this length.
static int get_fused_plan(hmg_ctx* c, int nxs, FusedPlan** out) {
    *out = nullptr;
    auto it = c->fused.find(nxs);
    if (it != c->fused.end()) {
        if (it->second.twM) *out = &it->second;
        return 0;
    }
    FusedPlan P;
    const int M = nxs / 2;
    bool ok = (nxs % 2 == 0) && M >= 4 && fft_make_plan(M, &P.plan) && M <= c->fused_max_m;
    if (ok) {
        int maxb = 0;
        for (int i = 0; i < P.plan.npass; ++i) {
            const int nb = M / P.plan.radix[i];
            maxb = std::max(maxb, (nb + FUSED_NT - 1) / FUSED_NT);
        }
        P.maxb = maxb;
        P.maxp = (M / 2 + FUSED_NT - 1) / FUSED_NT;
        // (M = 5000, nxs = 10000, has five butterflies per thread in its radix-2 pass and is left to the long-grid
        // route: measured on the Config-3 grid, one 80-KB row in LDS with a compile-time plan 0.498 ms, long-grid route
        // 0.408 ms, rocFFT 3.26 ms - tools/probes/nxs10000_routes.py)
        ok = maxb <= 4 && P.maxp <= 8;
    }
    if (!ok) {
        c->fused[nxs] = FusedPlan();  // remember the rejection
        return 0;
    }
    // twiddles per pass (ldsfft.hpp: pass_tw_table): element k of a pass's slice is W_M^(k twstep), so that
    // consecutive butterflies read consecutive elements instead of gathering at a stride of twstep from one table
    const std::vector<cplx> twM = pass_tw_table(P.plan);
    std::vector<UnpackTw> twN(M / 2 + 1);
    const long double twopi = 6.283185307179586476925286766559L;
    for (int j = 0; j <= M / 2; ++j)
        twN[j] = UnpackTw{(double)cosl(twopi * j / nxs), (double)sinl(twopi * j / nxs), j ? 1.0 / j : 0.0, 1.0 / (M - j)};
    HIP_TRY(hipMalloc((void**)&P.twM, twM.size() * sizeof(cplx)));
    HIP_TRY(hipMalloc((void**)&P.twN, twN.size() * sizeof(UnpackTw)));
    HIP_TRY(hipMemcpy(P.twM, twM.data(), twM.size() * sizeof(cplx), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(P.twN, twN.data(), twN.size() * sizeof(UnpackTw), hipMemcpyHostToDevice));
    auto res = c->fused.emplace(nxs, P);
    *out = &res.first->second;
    return 0;
}

template <int MAXB, int MAXP, int SPECM = 0>
static int launch_fused(hmg_ctx* c, const FusedArgs& A, int rows) {
    const size_t lds = (size_t)A.plan.M * 16 + 32 * sizeof(double);
    if (lds > 48 * 1024)
        HIP_TRY(hipFuncSetAttribute((const void*)profile_fused_kernel<FUSED_NT, MAXB, MAXP, SPECM>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // (Tried and dropped: fewer workgroups that loop over rows, to take the ~4 us of workgroup launch and
    // first-load latency per row off the path.  The loop-carried state spills under the 64-VGPR cap:
    // 0.21 -> 0.56-0.61 ms, with 1023, 2047 or one workgroup per row alike.)
    hipLaunchKernelGGL((profile_fused_kernel<FUSED_NT, MAXB, MAXP, SPECM>), dim3(rows), dim3(FUSED_NT), lds,
                       c->stream, A);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <int MAXB, int MAXP, int SPECM = 0>
static int launch_table(hmg_ctx* c, const FusedArgs& A, int rows) {
    const size_t lds = (size_t)A.plan.M * 16 + 32 * sizeof(double);
    if (lds > 48 * 1024)
        HIP_TRY(hipFuncSetAttribute((const void*)profile_table_kernel<FUSED_NT, MAXB, MAXP, SPECM>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((profile_table_kernel<FUSED_NT, MAXB, MAXP, SPECM>), dim3(rows), dim3(FUSED_NT), lds, c->stream, A);
    HIP_TRY(hipGetLastError());
    return 0;
}

// profile_group_kernel = the fused row kernel with `nchain` per-z chain workgroups in front of the rows;
// with N: tensor_group_kernel, the analytic NFW rows behind them in the same grid
template <int MAXB, int MAXP, int SPECM = 0>
static int launch_fused_group(hmg_ctx* c, const FusedArgs& A, int rows, const ChainArgs& C, int nchain, size_t chain_lds,
                              const NfwArgs* N = nullptr, size_t nfw_blocks = 0) {
    size_t lds = (size_t)A.plan.M * 16 + 32 * sizeof(double);
    if (chain_lds > lds) lds = chain_lds;
    if (N || C.has_mf) {      // (only the tensor kernel carries the chain's sigma^2 -> n, b link)
        if constexpr (SPECM != 0) {
            const NfwArgs none{};
            if (!N) { N = &none; nfw_blocks = 0; }
            REQUIRE((size_t)rows + nchain + nfw_blocks <= 2147483647u, "bad grid");
            if (lds > 48 * 1024)
                HIP_TRY(hipFuncSetAttribute((const void*)tensor_group_kernel<MAXB, MAXP, SPECM>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL((tensor_group_kernel<MAXB, MAXP, SPECM>), dim3((unsigned)(rows + nchain + nfw_blocks)),
                               dim3(FUSED_NT), lds, c->stream, C, A, nchain, rows, N->T, N->acoef, N->ktile, N->nm, N->nk, N->cs,
                               N->rss, N->zs, N->ks, N->uk);
            HIP_TRY(hipGetLastError());
            return 0;
        } else {
            REQUIRE(false, "internal: the tensor group is compiled for compile-time plans only");
        }
    }
    if (lds > 48 * 1024)
        HIP_TRY(hipFuncSetAttribute((const void*)profile_group_kernel<MAXB, MAXP, SPECM>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((profile_group_kernel<MAXB, MAXP, SPECM>), dim3(rows + nchain), dim3(FUSED_NT), lds, c->stream, C,
                       A, nchain);
    HIP_TRY(hipGetLastError());
    return 0;
}


// ---- pruned long-grid route (profile_pruned_kernel) -------------------------------------------------------
// Lengths of the sub-transforms that are compiled in.  A launch takes the smallest one that divides M = nxs/2
// and covers the support bound of its rows.
static const int PRUNED_LP[] = {1000, 1024, 1250, 1500, 2000, 2048, 2500};

template <class T>
static int upload_table(const std::vector<T>& h, T** d) {
    HIP_TRY(hipMalloc((void**)d, h.size() * sizeof(T)));
    HIP_TRY(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

// LP == 0: the tables by mode (chirp and narrow-band routes); LP > 0: the decomposition's tables by residue
static int get_pruned_plan(hmg_ctx* c, int nxs, int LP, PrunedPlan** out) {
    const auto key = std::make_pair(nxs, LP);
    auto it = c->pruned.find(key);
    if (it != c->pruned.end()) { *out = &it->second; return 0; }
    REQUIRE(!c->capturing, "long-grid tables cannot be built inside a captured step: run the step once eagerly first");
    const int M = nxs / 2;
    PrunedPlan P;
    if (LP == 0) {
        std::vector<cplx> twB(M);
        std::vector<UnpackTw> twN(M / 2 + 1);
        const long double twopi = 6.283185307179586476925286766559L;
        for (int t = 0; t < M; ++t) twB[t] = cplx{(double)cosl(twopi * t / M), (double)-sinl(twopi * t / M)};
        for (int j = 0; j <= M / 2; ++j)
            twN[j] = UnpackTw{(double)cosl(twopi * j / nxs), (double)sinl(twopi * j / nxs), j ? 1.0 / j : 0.0, 1.0 / (M - j)};
        if (upload_table(twB, &P.twB) || upload_table(twN, &P.twN)) return 1;
    } else {
        if (upload_table(residue_tw_table(M, LP), &P.twR) || upload_table(residue_unpack_table(M, LP), &P.twNr)) return 1;
    }
    auto res = c->pruned.emplace(key, P);
    *out = &res.first->second;
    return 0;
}

// Per-pass twiddle table of the length-L plan the long-grid kernels are compiled for (independent of the lengths the
// one-row kernel takes: HMG_FUSED_MAX_M does not reach here).
static int get_pass_table(hmg_ctx* c, int L, const cplx** out) {
    auto it = c->pass_tw.find(L);
    if (it != c->pass_tw.end()) { *out = it->second; return 0; }
    REQUIRE(!c->capturing, "long-grid tables cannot be built inside a captured step: run the step once eagerly first");
    FftPlanDev plan;
    REQUIRE(fft_make_plan(L, &plan), "no radix-2/3/4/5 plan for a compiled sub-transform length");
    cplx* d = nullptr;
    if (upload_table(pass_tw_table(plan), &d)) return 1;
    c->pass_tw[L] = d;
    *out = d;
    return 0;
}

static int get_chirp_plan(hmg_ctx* c, int nxs, int LP, int p0, ChirpPlan** out) {
    const auto key = std::make_tuple(nxs, LP, p0);
    auto it = c->chirp.find(key);
    if (it != c->chirp.end()) { *out = &it->second; return 0; }
    REQUIRE(!c->capturing, "chirp tables cannot be built inside a captured step: run the step once eagerly first");
    const ChirpTables T = chirp_make_tables(nxs / 2, 2 * LP, p0);
    ChirpPlan P;
    P.Jw = T.Jw;
    HIP_TRY(hipMalloc((void**)&P.chP, T.chP.size() * sizeof(cplx)));
    HIP_TRY(hipMalloc((void**)&P.chJ, T.chJ.size() * sizeof(cplx)));
    HIP_TRY(hipMalloc((void**)&P.Bw, T.Bw.size() * sizeof(cplx)));
    HIP_TRY(hipMemcpy(P.chP, T.chP.data(), T.chP.size() * sizeof(cplx), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(P.chJ, T.chJ.data(), T.chJ.size() * sizeof(cplx), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(P.Bw, T.Bw.data(), T.Bw.size() * sizeof(cplx), hipMemcpyHostToDevice));
    auto res = c->chirp.emplace(key, P);
    *out = &res.first->second;
    return 0;
}

// Support bound of a launch's rows in packed samples, and the bound on the modes they need.  A call measures them (one
// small kernel, a 4-byte copy, a stream synchronisation) unless a bound measured for the same arrays is on file AND the
// contents of those arrays are known not to have changed: inside a captured step (what a replay computes is what was
// captured), or when the caller has tagged the contents (hmg_profile_support_epoch != 0: the facade tags them per
// model and mass grid) - eager calls of a model then cost no host synchronisation after the first.  Every row
// re-checks itself against the bound its launch was sized for (fault word).  *known = 0: nothing on file inside a capture.
static int profile_support(hmg_ctx* c, int rows, const FusedArgs& A, int* p0max, int* jnmax, int* known) {
    const SupportKey key{A.cmax, A.xs, A.nconst ? A.rss : nullptr, A.nconst ? A.ks : nullptr, rows, A.nxs, A.nk, c->support_epoch};
    *known = 1;
    if (c->capturing || c->support_epoch != 0) {
        auto it = c->support.find(key);
        if (it != c->support.end()) {
            *p0max = it->second.first;
            *jnmax = it->second.second;
            return 0;
        }
        if (c->capturing) { *known = 0; return 0; }
    }
    if (ensure_scratch(c, 2, 64)) return 1;
    int* d_p0 = (int*)c->scratch[2];
    HIP_TRY(hipMemsetAsync(d_p0, 0, 2 * sizeof(int), c->stream));
    // (the needed modes are only bounded when the target grid is ascending - the promise of the hint arrays)
    HIP_TRY((hipError_t)launch_profile_support(c->stream, rows, A.nxs, A.xs, A.cmax, A.nconst ? A.rss : nullptr, A.zs, A.nm,
                                               A.kts, A.ks, A.nk, d_p0));
    int h[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(h, d_p0, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (!A.nconst) h[1] = A.nxs / 2;
    if (c->support.size() >= 256) c->support.clear();     // (bounded; a dropped entry costs one more measurement)
    c->support[key] = std::make_pair(h[0], h[1]);
    *p0max = h[0];
    *jnmax = h[1];
    return 0;
}

// Returns 0 and *taken = 1 when the pruned route ran, *taken = 0 when the launch is not one it can take.
// can_fall_back: the caller has another route for this length (one row in LDS), so an unknown bound inside a captured
// step is no error.
static int profile_fft_pruned(hmg_ctx* c, const FusedArgs& A0, int rows, bool can_fall_back, int* taken) {
    *taken = 0;
    const int nxs = A0.nxs, M = nxs / 2;
    if ((nxs & 1) || M < 2 * PRUNED_LP[0]) return 0;
    bool any = false;
    for (int lp : PRUNED_LP) any = any || (M % lp == 0 && M / lp >= 2);
    if (!any) return 0;
    int p0max = 0, jnmax = 0, known = 1;
    if (profile_support(c, rows, A0, &p0max, &jnmax, &known)) return 1;
    if (!known) {
        REQUIRE(can_fall_back, "profile support bound unknown inside a captured step: run the step once eagerly first");
        return 0;
    }
    int LP = 0;
    const int lp_min = c->pruned_lp_min > p0max ? c->pruned_lp_min : p0max;     // (HMG_PRUNED_LP_MIN: tuning / tests)
    for (int lp : PRUNED_LP)
        if (M % lp == 0 && M / lp >= 2 && lp >= lp_min) { LP = lp; break; }
    if (!LP)
        for (int lp : PRUNED_LP)
            if (M % lp == 0 && M / lp >= 2 && lp >= p0max) { LP = lp; break; }
    if (!LP) {
        // The support does not prune (the tSZ notebook's pressure profile at xmax = 2).  If every row needs few modes
        // the narrow-band route takes the launch: D = M / LB transforms of length LB >= 2 jn + 2 of the decimated
        // rows (longgrid.hip); otherwise rocFFT.
        if (!c->use_band_fft || !A0.nconst) return 0;
        int LB = 0;
        for (int lb : {1000, 1024, 1250})
            if (M % lb == 0 && band_lb_compiled(lb) && 2 * jnmax + 2 <= lb) { LB = lb; break; }
        if (!LB) return 0;
        PrunedPlan *PP = nullptr, *PR = nullptr;
        const cplx* twL = nullptr;
        if (get_pruned_plan(c, nxs, 0, &PP)) return 1;
        if (get_pruned_plan(c, nxs, LB, &PR)) return 1;      // its twR: the mode twiddles W_M^(p1 j) by residue p1
        if (get_pass_table(c, LB, &twL)) return 1;
        PrunedArgs G{};
        G.F = A0;
        G.F.twN = PP->twN;
        if (ensure_scratch(c, 0, (size_t)3 * nxs * 8)) return 1;       // x, ln x, trapezoid weights in the kernel's walk order
        G.M = M; G.R = M / LB; G.twB = PP->twB; G.twL = twL; G.u = (double*)c->scratch[0]; G.fault = c->d_fault; G.row0 = 0;
        G.twR = PR->twR;
        int stop = -1;
        if (bracket_open(c, HMG_KERNEL_PROFILE_FFT, &stop)) return 1;
        HIP_TRY((hipError_t)launch_band(c->stream, LB, G, rows, jnmax));
        *taken = 1;
        return bracket_close(c, stop);
    }
    PrunedPlan *PP = nullptr, *PR = nullptr;
    const cplx* twL = nullptr;
    if (get_pruned_plan(c, nxs, 0, &PP)) return 1;
    if (get_pruned_plan(c, nxs, LP, &PR)) return 1;
    if (get_pass_table(c, LP, &twL)) return 1;
    // scratch line of M doubles per row, at most 8 GiB per launch
    size_t rpl = ((size_t)8 << 30) / ((size_t)M * 8);
    if (rpl < 1) rpl = 1;
    if (rpl > (size_t)rows) rpl = rows;
    if (ensure_scratch(c, 0, rpl * (size_t)M * 8)) return 1;
    PrunedArgs G;
    G.F = A0;
    G.F.twN = PP->twN;
    G.M = M; G.R = M / LP; G.twB = PP->twB; G.twL = twL; G.u = (double*)c->scratch[0]; G.fault = c->d_fault; G.row0 = 0;
    G.twR = PR->twR; G.twNr = PR->twNr; G.rmagic = (unsigned)(4294967296ull / (unsigned)G.R) + 1u;
    G.chP = G.chJ = G.Bw = G.twC = nullptr;
    G.Jw = 0; G.p0 = 0;
    if (c->use_chirp && (LP == 1000 || LP == 1250) && p0max >= 1) {
        // the window is built for the support bound rounded up to 16 samples (fewer distinct tables; Jw shrinks by <= 8)
        int p0 = (p0max + 15) / 16 * 16;
        if (p0 > LP) p0 = LP;
        ChirpPlan* CP = nullptr;
        const cplx* twC = nullptr;
        if (get_chirp_plan(c, nxs, LP, p0, &CP)) return 1;
        if (get_pass_table(c, 2 * LP, &twC)) return 1;
        G.chP = CP->chP; G.chJ = CP->chJ; G.Bw = CP->Bw; G.twC = twC; G.Jw = CP->Jw; G.p0 = p0;
    }
    int stop = -1;
    if (bracket_open(c, HMG_KERNEL_PROFILE_FFT, &stop)) return 1;
    HIP_TRY((hipError_t)launch_pruned(c->stream, LP, G, rows, rpl));
    *taken = 1;
    return bracket_close(c, stop);
}

// One hmg_profile_fft; with a chain (nchain > 0) and a length the in-LDS transform takes, chain and rows share
// the launch, otherwise *chain_done stays 0 and the caller issues the chain on its own.
// rho_tab != nullptr: the profile comes from a table (hmg_profile_fft_table); *taken = 0 when no in-LDS route takes the
// launch (the caller then runs its rocFFT chain); the family parameters of p are not read.
// The length M of a one-row transform whose plan is compiled in (strides, twiddle steps, index multipliers are immediates):
// nxs = 5000 (the headline length) and 1000, 2000, 3000, 4000, 6000; 0: the run-time plan.
static int fused_ct_plan(const hmg_ctx* c, const FftPlanDev& pl) {
    if (FUSED_NT != 512 || c->fused_generic) return 0;       // (fused_generic: testing, force the run-time plan)
    if (pl.M == 2500 && pl.npass == 5 && pl.radix[0] == 4 && pl.radix[1] == 5 && pl.radix[2] == 5 && pl.radix[3] == 5 &&
        pl.radix[4] == 5)
        return 2500;
    if (pl.M == 500 || pl.M == 1000 || pl.M == 1500 || pl.M == 2000 || pl.M == 3000) return pl.M;
    return 0;
}

// N != nullptr: analytic NFW rows that may ride in the same launch when chain and rows share one (*nfw_done = 1 then).
static int profile_fft_impl(hmg_ctx* c, int nz, int nm, int nk, const hmg_profile_fft_part& p, const ChainArgs* C,
                            int nchain, size_t chain_lds, int* chain_done, const double* rho_tab = nullptr,
                            int rho_shared = 0, int* taken = nullptr, const NfwArgs* N = nullptr, size_t nfw_blocks = 0,
                            int* nfw_done = nullptr) {
    const int nxs = p.nxs;
    const double step = p.fft_step;
    const double *xs = p.d_xs, *kts = p.d_kts, *amp = p.d_amp, *xcs = p.d_xc, *alpha = p.d_alpha, *expo = p.d_expo;
    const double amp_c = p.amp_const, xc_c = p.xc_const, alpha_c = p.alpha_const, expo_c = p.expo_const, gamma = p.gamma;
    const double *cmax = p.d_cmax, *rss = p.d_rss, *zs = p.d_zs, *ks = p.d_ks, *post = p.d_post, *logxs = p.d_logxs;
    const int do_mass_norm = p.do_mass_norm;
    double* out = p.d_out;
    int* nconst = p.d_nconst;
    double* cconst = p.d_cconst;
    if (chain_done) *chain_done = 0;
    if (nfw_done) *nfw_done = 0;
    REQUIRE(c && xs && kts && cmax && rss && zs && ks && out, "NULL argument");
    REQUIRE((nconst == nullptr) == (cconst == nullptr), "pass both hint arrays or neither");
    REQUIRE(nz > 0 && nm > 0 && nk > 0, "empty grid");
    REQUIRE(nxs >= 4, "nxs too small");
    const int nh = nxs / 2;  // rfft output length is nh+1
    const int rows = nz * nm;
    REQUIRE(step > 0.0, "step must be positive");
    const bool xs_aligned = ((uintptr_t)xs % 16) == 0;   // the row kernels read x in 16-B pairs
    if (c->use_fused_fft && xs_aligned) {
        FusedPlan* FP = nullptr;
        if (get_fused_plan(c, nxs, &FP)) return 1;
        // Rows longer than M = 2500 that would still fit LDS as one row (run-time plan, one or two workgroups per CU) are
        // faster on the long-grid route when it applies - Config-3 grid, profile stage, nxs = 6000 / 8000 / 12000: 0.465 /
        // 0.551 / 0.976 ms as one row against 0.342 / 0.400 / 0.438 ms (tools/probes/mid_length_routes.py)
        const bool prefer_long = FP && FP->plan.M > c->fused_prefer_m;
        if ((!FP || prefer_long) && c->use_pruned_fft) {
            // a grid too long for one LDS row: the pruned decomposition, if the support of the rows is short enough
            FusedArgs A{};
            A.nxs = nxs; A.nm = nm; A.nk = nk; A.do_norm = do_mass_norm;
            A.xs = xs; A.twM = nullptr; A.twN = nullptr; A.kts = kts;
            A.amp = amp; A.xc = xcs; A.alpha = alpha; A.expo = expo;
            A.amp_c = amp_c; A.xc_c = xc_c; A.alpha_c = alpha_c; A.expo_c = expo_c; A.gamma = gamma;
            A.step = step; A.cmax = cmax; A.rss = rss; A.zs = zs; A.ks = ks; A.post = post; A.out = out;
            A.nconst = nconst; A.cconst = cconst;
            A.logx = logxs;
            A.rho_tab = rho_tab; A.rho_shared = rho_shared;
            int took = 0;
            if (profile_fft_pruned(c, A, rows, FP != nullptr, &took)) return 1;
            if (took) {
                if (taken) *taken = 1;
                return 0;
            }
        }
        if (FP) {
            FusedArgs A;
            A.plan = FP->plan; A.nxs = nxs; A.nm = nm; A.nk = nk; A.do_norm = do_mass_norm;
            A.xs = xs; A.twM = FP->twM; A.twN = FP->twN; A.kts = kts;
            A.amp = amp; A.xc = xcs; A.alpha = alpha; A.expo = expo;
            A.amp_c = amp_c; A.xc_c = xc_c; A.alpha_c = alpha_c; A.expo_c = expo_c; A.gamma = gamma;
            A.step = step; A.cmax = cmax; A.rss = rss; A.zs = zs; A.ks = ks; A.post = post; A.out = out;
            A.nconst = nconst; A.cconst = cconst;
            A.logx = logxs;
            A.rho_tab = rho_tab; A.rho_shared = rho_shared;
            // the row scalars a rows part left for exactly this transform (they carry the left-fill count: hints required)
            A.rowsc = (p.d_rowsc && nconst) ? p.d_rowsc : nullptr;
            int stop = -1;
            if (bracket_open(c, HMG_KERNEL_PROFILE_FFT, &stop)) return 1;
            if (rho_tab) {      // table builds: the nxs = 5000 plan and two run-time-plan shapes cover every one-row length
                int rc;
                if (FUSED_NT == 512 && FP->plan.M == 2500 && !c->fused_generic) rc = launch_table<2, 3, 2500>(c, A, rows);
                else if (FP->maxb <= 2 && FP->maxp <= 4) rc = launch_table<2, 4>(c, A, rows);
                else rc = launch_table<4, 8>(c, A, rows);
                if (rc) return 1;
                if (taken) *taken = 1;
                return bracket_close(c, stop);
            }
            if (!logxs && rows >= 8192) {   // no prepared table: its own launch pays from ~8000 rows (MI355X: -1 % at 16384 rows, +2 % at 4096)
                if (ensure_scratch(c, 2, (size_t)nxs * 8)) return 1;
                hipLaunchKernelGGL(logx_kernel, grid1d((size_t)nxs, 256), dim3(256), 0, c->stream, nxs, xs,
                                   (double*)c->scratch[2]);
                HIP_TRY(hipGetLastError());
                A.logx = (const double*)c->scratch[2];
            }
            int rc;
            const int mb = FP->maxb, mp = FP->maxp;
            const FftPlanDev& pl = FP->plan;
            const bool spec2500 = fused_ct_plan(c, pl) == 2500;
            // the group kernels read the rows' output-side scalars from the record of the rows stage; a caller without
            // one (no hint arrays: ks not ascending) gets the stand-alone row kernel and its chain as a launch of its own
            const bool grouped = C && nchain > 0 && FUSED_NT == 512 && A.rowsc != nullptr;
            // lengths with a compile-time plan (fused_passes_ct): nxs = 1000, 2000, 3000, 4000, 6000 (the last one only
            // when its rows' support does not let the long-grid route take it)
            const int ctM = spec2500 ? 0 : fused_ct_plan(c, pl);
            if (grouped) {
                if (spec2500) rc = launch_fused_group<2, 3, 2500>(c, A, rows, *C, nchain, chain_lds, N, nfw_blocks);
                else if (ctM == 500) rc = launch_fused_group<1, 1, 500>(c, A, rows, *C, nchain, chain_lds, N, nfw_blocks);
                else if (ctM == 1000) rc = launch_fused_group<1, 1, 1000>(c, A, rows, *C, nchain, chain_lds, N, nfw_blocks);
                else if (ctM == 1500) rc = launch_fused_group<1, 2, 1500>(c, A, rows, *C, nchain, chain_lds, N, nfw_blocks);
                else if (ctM == 2000) rc = launch_fused_group<2, 2, 2000>(c, A, rows, *C, nchain, chain_lds, N, nfw_blocks);
                else if (ctM == 3000) rc = launch_fused_group<3, 3, 3000>(c, A, rows, *C, nchain, chain_lds, N, nfw_blocks);
                else if (mb <= 1 && mp <= 2) rc = launch_fused_group<1, 2>(c, A, rows, *C, nchain, chain_lds, N, nfw_blocks);
                else if (mb <= 2 && mp <= 3) rc = launch_fused_group<2, 3>(c, A, rows, *C, nchain, chain_lds, N, nfw_blocks);
                else if (mb <= 2 && mp <= 4) rc = launch_fused_group<2, 4>(c, A, rows, *C, nchain, chain_lds, N, nfw_blocks);
                else rc = launch_fused_group<4, 8>(c, A, rows, *C, nchain, chain_lds, N, nfw_blocks);
                if (!rc && chain_done) *chain_done = 1;
                if (!rc && N && nfw_done) *nfw_done = 1;
            }
            else if (spec2500) rc = launch_fused<2, 3, 2500>(c, A, rows);                 // nxs = 5000, compile-time plan
            else if (ctM == 500) rc = launch_fused<1, 1, 500>(c, A, rows);
            else if (ctM == 1000) rc = launch_fused<1, 1, 1000>(c, A, rows);
            else if (ctM == 1500) rc = launch_fused<1, 2, 1500>(c, A, rows);
            else if (ctM == 2000) rc = launch_fused<2, 2, 2000>(c, A, rows);
            else if (ctM == 3000) rc = launch_fused<3, 3, 3000>(c, A, rows);
            else if (mb <= 1 && mp <= 2) rc = launch_fused<1, 2>(c, A, rows);
            else if (mb <= 2 && mp <= 3) rc = launch_fused<2, 3>(c, A, rows);
            else if (mb <= 2 && mp <= 4) rc = launch_fused<2, 4>(c, A, rows);
            else rc = launch_fused<4, 8>(c, A, rows);
            if (rc) return 1;
            return bracket_close(c, stop);
        }
    }
    if (rho_tab) {              // no in-LDS route for this length: the caller's table -> rocFFT chain
        if (taken) *taken = 0;
        return 0;
    }
    // ---- rocFFT path.  Chunk the batch so integrand + spectrum of a chunk stay inside the 256 MiB Infinity Cache:
    // the R2C input written by K4 and the spectrum read by K5 then never round-trip through HBM.
    const size_t per_row = (size_t)nxs * 8 + (size_t)(nh + 1) * 16;
    size_t budget = c->fft_chunk_bytes ? c->fft_chunk_bytes : ((size_t)160 << 20);
    int chunk = (int)(budget / per_row);
    if (chunk < 1) chunk = 1;
    if (chunk > rows) chunk = rows;
    if (ensure_scratch(c, 0, (size_t)chunk * nxs * 8)) return 1;
    if (ensure_scratch(c, 1, (size_t)chunk * (nh + 1) * 16)) return 1;
    if (ensure_scratch(c, 2, (size_t)chunk * 8)) return 1;
    double* fin = (double*)c->scratch[0];
    double2* fout = (double2*)c->scratch[1];
    double* mnorm = (double*)c->scratch[2];
    const bool stage = (size_t)nh * sizeof(double) <= 64 * 1024;
    const size_t lds = stage ? (size_t)nh * sizeof(double) : 0;
    int stop = -1;
    if (bracket_open(c, HMG_KERNEL_PROFILE_FFT, &stop)) return 1;
    for (int r0 = 0; r0 < rows; r0 += chunk) {
        const int nr = rows - r0 < chunk ? rows - r0 : chunk;
        hipLaunchKernelGGL(integrand_kernel, dim3(nr), dim3(256), 0, c->stream, nxs, r0, xs, amp, xcs,
                           alpha, expo, amp_c, xc_c, alpha_c, expo_c, gamma, cmax, do_mass_norm, (int)xs_aligned, fin, mnorm);
        HIP_TRY(hipGetLastError());
        FftPlan* P = nullptr;
        if (get_plan(c, nxs, nr, &P)) return 1;
        void* ib[1] = {fin};
        void* ob[1] = {fout};
        FFT_TRY(rocfft_execution_info_set_stream(P->info, c->stream));
        FFT_TRY(rocfft_execute(P->plan, ib, ob, P->info));
        if (stage)
            hipLaunchKernelGGL(interp_kernel<true>, dim3(nr), dim3(256), lds, c->stream, nm, nk, nh, r0, step,
                               (const double2*)fout, kts, mnorm, rss, zs, ks, post, out, nconst, cconst);
        else
            hipLaunchKernelGGL(interp_kernel<false>, dim3(nr), dim3(256), 0, c->stream, nm, nk, nh, r0, step,
                               (const double2*)fout, kts, mnorm, rss, zs, ks, post, out, nconst, cconst);
        HIP_TRY(hipGetLastError());
    }
    return bracket_close(c, stop);
}

int hmg_profile_fft(hmg_ctx* c, int nz, int nm, int nk, int nxs, double step, const double* xs, const double* kts,
                    const double* amp, const double* xcs, const double* alpha, const double* expo,
                    double amp_c, double xc_c, double alpha_c, double expo_c, double gamma,
                    const double* cmax, const double* rss, const double* zs, const double* ks,
                    int do_mass_norm, const double* post, double* out, int* nconst, double* cconst,
                    const double* logxs) {
    const hmg_profile_fft_part p{nxs, step, xs, kts, amp, xcs, alpha, expo, amp_c, xc_c, alpha_c, expo_c, gamma,
                                 cmax, rss, zs, ks, do_mass_norm, post, out, nconst, cconst, logxs};
    return profile_fft_impl(c, nz, nm, nk, p, nullptr, 0, 0, nullptr);
}

int hmg_profile_support_epoch(hmg_ctx* c, long long epoch) {
    REQUIRE(c, "NULL ctx");
    c->support_epoch = epoch;
    return 0;
}

int hmg_profile_fft_logx(hmg_ctx* c, int nxs, const double* xs, double* logxs) {
    REQUIRE(c && xs && logxs && nxs > 0, "bad argument");
    hipLaunchKernelGGL(logx_kernel, grid1d((size_t)nxs, 256), dim3(256), 0, c->stream, nxs, xs, logxs);
    HIP_TRY(hipGetLastError());
    return 0;
}

int hmg_hod(hmg_ctx* c, int nz, int nm, const hmg_hod_params* p, const double* zs, const double* ms,
            const double* lthr, const double* nzm, const double* bh, const double* wm, double* Nc,
            double* Ns, double* NsNsm1, double* NcNs, double* ngal, double* bg) {
    REQUIRE(c && p && zs && ms && lthr && nzm && bh && wm && Nc && Ns && NsNsm1 && NcNs && ngal && bg,
            "NULL argument");
    REQUIRE(nz > 0 && nm > 0, "empty grid");
    REQUIRE(p->corr == 0 || p->corr == 1, "corr must be 0 (max) or 1 (min)");
    HodDev P{p->sig_log_mstellar, p->alphasat, p->Bsat, p->betasat, p->Bcut, p->betacut, p->corr};
    int hod_threads = 1024;
    if (const char* e = getenv("HMG_HOD_THREADS")) hod_threads = atoi(e);
    REQUIRE(hod_threads >= 64 && hod_threads <= 1024 && hod_threads % 64 == 0, "HMG_HOD_THREADS must be a multiple of 64 up to 1024");
    REQUIRE((nm + 63) / 64 <= HOD_MAX_TILES, "nm too large for the HOD reduction (65536)");
    if (!getenv("HMG_HOD_THREADS")) hod_threads = std::min(1024, std::max(64, (nm + 63) / 64 * 64));
    const HodRowArgs A{nm, P, zs, ms, lthr, nzm, bh, wm, Nc, Ns, NsNsm1, NcNs, ngal, bg};
    hipLaunchKernelGGL(hod_kernel, dim3(nz), dim3(hod_threads), 0, c->stream, A);
    HIP_TRY(hipGetLastError());
    return 0;
}

static int tensor_slot(std::vector<const double*>& list, const double* p) {
    if (!p) return -1;
    for (size_t i = 0; i < list.size(); ++i)
        if (list[i] == p) return (int)i;
    list.push_back(p);
    return (int)list.size() - 1;
}

static int fill_tracer(const hmg_tracer* t, std::vector<const double*>& tens, TracerDev* out) {
    REQUIRE(t->kind == HMG_TRACER_MATTER || t->kind == HMG_TRACER_HOD || t->kind == HMG_TRACER_PRESSURE,
            "unknown tracer kind");
    REQUIRE(t->d_prof, "tracer has no profile tensor");
    out->kind = t->kind;
    out->t_prof = tensor_slot(tens, t->d_prof);
    out->t_cprof = (t->kind == HMG_TRACER_HOD) ? tensor_slot(tens, t->d_cprof) : -1;
    out->Nc = t->d_Nc; out->Ns = t->d_Ns; out->NcNs = t->d_NcNs; out->NsNsm1 = t->d_NsNsm1;
    out->ngal = t->d_ngal; out->bias_override = t->d_bias_override;
    if (t->kind == HMG_TRACER_HOD)
        REQUIRE(t->d_Nc && t->d_Ns && t->d_NcNs && t->d_NsNsm1 && t->d_ngal, "HOD tracer needs Nc,Ns,NcNs,NsNsm1,ngal");
    return 0;
}

template <int NT, int V>
static int launch_power(hmg_ctx* c, const PowerArgs& A, int nz, int ms_split) {
    const int per_block = 64 * V;
    dim3 grid((A.nk + per_block - 1) / per_block, nz);
    const size_t lds = (size_t)ms_split * 3 * V * 64 * sizeof(double);
    if (lds > 48 * 1024)
        HIP_TRY(hipFuncSetAttribute((const void*)power_kernel<NT, V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int stop = -1;
    if (bracket_open(c, HMG_KERNEL_POWER, &stop)) return 1;
    hipLaunchKernelGGL((power_kernel<NT, V>), grid, dim3(64 * ms_split), lds, c->stream, A);
    HIP_TRY(hipGetLastError());
    return bracket_close(c, stop);
}

static int power_impl(hmg_ctx* c, int nz, int nm, int nk, const hmg_tracer* ta, const hmg_tracer* tb,
                      const double* nzm, const double* bh, const double* ms, const double* wm, const double* ks,
                      const double* Pzk, double rho_m0, double kstar, double* P1h, double* P2h,
                      double* I1, double* I2, double* Cout) {
    REQUIRE(c && ta && tb && nzm && bh && ms && wm && ks, "NULL argument");
    REQUIRE(P1h || P2h || (I1 && I2 && Cout), "no output requested");
    REQUIRE(!P2h || Pzk, "P2h needs Pzk");
    REQUIRE(nz > 0 && nm > 0 && nk > 0, "empty grid");
    REQUIRE(nz <= 65535, "nz too large");
    std::vector<const double*> tens;
    PowerPrep Q;
    if (fill_tracer(ta, tens, &Q.a)) return 1;
    if (fill_tracer(tb, tens, &Q.b)) return 1;
    Q.nt = (int)tens.size();
    Q.rho_m0 = rho_m0;
    REQUIRE(Q.nt >= 1 && Q.nt <= PW_MAXT, "bad tensor count");
    const int nc1 = 1 + Q.nt;
    if (ensure_scratch(c, 3, (size_t)nz * nm * PW_NF * nc1 * 8 + (size_t)nz * 4 * 8 + 64)) return 1;
    double* coef = (double*)c->scratch[3];
    double* side = coef + (size_t)nz * nm * PW_NF * nc1;
    hipLaunchKernelGGL(power_prep_kernel, dim3(nz), dim3(256), 0, c->stream, nm, Q, nzm, bh, ms, wm, coef, side);
    HIP_TRY(hipGetLastError());
    PowerArgs A;
    for (int i = 0; i < PW_MAXT; ++i) A.tens[i] = i < Q.nt ? tens[i] : nullptr;
    A.coef = coef; A.side = side; A.ks = ks; A.Pzk = Pzk; A.P1h = P1h; A.P2h = P2h;
    A.I1 = I1; A.I2 = I2; A.Cout = Cout;
    A.kstar = kstar; A.nm = nm; A.nk = nk;
    bool vec2 = (nk % 2 == 0);
    for (int i = 0; i < Q.nt; ++i) vec2 = vec2 && (((uintptr_t)tens[i]) % 16 == 0);
    // enough waves to cover the chip: MS mass slices per block
    const int V = vec2 ? 2 : 1;
    const long blocks = (long)((nk + 64 * V - 1) / (64 * V)) * nz;
    // the number of mass slices fixes the summation order over m: keep it a function of nm only,
    // so that a z-slab run (multi-GPU) reproduces the full-grid numbers bit for bit
    (void)blocks;
    int ms_split = 8;
    while (ms_split > 1 && ms_split > nm) ms_split >>= 1;
#define PW_CASE(NT_)                                                      \
    case NT_:                                                             \
        return vec2 ? launch_power<NT_, 2>(c, A, nz, ms_split) : launch_power<NT_, 1>(c, A, nz, ms_split);
    switch (Q.nt) {
        PW_CASE(1)
        PW_CASE(2)
        PW_CASE(3)
        PW_CASE(4)
    }
#undef PW_CASE
    return fail("hmg_power", "unreachable", __FILE__, __LINE__);
}

int hmg_power(hmg_ctx* c, int nz, int nm, int nk, const hmg_tracer* ta, const hmg_tracer* tb,
              const double* nzm, const double* bh, const double* ms, const double* wm, const double* ks,
              const double* Pzk, double rho_m0, double kstar, double* P1h, double* P2h) {
    return power_impl(c, nz, nm, nk, ta, tb, nzm, bh, ms, wm, ks, Pzk, rho_m0, kstar, P1h, P2h, nullptr, nullptr, nullptr);
}

int hmg_power_2halo_terms(hmg_ctx* c, int nz, int nm, int nk, const hmg_tracer* ta, const hmg_tracer* tb,
                          const double* nzm, const double* bh, const double* ms, const double* wm,
                          const double* ks, double rho_m0, double* I1, double* I2, double* C12) {
    REQUIRE(I1 && I2 && C12, "NULL output");
    return power_impl(c, nz, nm, nk, ta, tb, nzm, bh, ms, wm, ks, nullptr, rho_m0, 1.0, nullptr, nullptr, I1, I2, C12);
}

template <int NT, int NTR, int V, bool W16, unsigned CODE = 0>
static int launch_power_batch(hmg_ctx* c, const BatchArgs& A, int nz) {
    const int per_block = 64 * V;
    dim3 grid((A.nk + per_block - 1) / per_block, nz);
    constexpr int NACC = NTR + NTR * (NTR + 1) / 2;
    const size_t lds = (size_t)8 * NACC * V * 64 * sizeof(double);
    if (lds > 48 * 1024)
        HIP_TRY(hipFuncSetAttribute((const void*)power_batch_kernel<NT, NTR, V, W16, CODE>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int stop = -1;
    if (bracket_open(c, HMG_KERNEL_POWER, &stop)) return 1;
    hipLaunchKernelGGL((power_batch_kernel<NT, NTR, V, W16, CODE>), grid, dim3(W16 ? 1024 : 512), lds, c->stream, A);
    HIP_TRY(hipGetLastError());
    return bracket_close(c, stop);
}

// Everything hmg_power_batch decides on the host: canonical tracer order, structure code, the argument blocks
// of the two launches and the launch shape.  Deterministic in its inputs, so that a preparation issued from a
// grouped launch and the main launch issued later agree on every pointer.
struct PbPlan {
    BatchPrep Q;
    PrepArgs PA;
    BatchArgs A;
    int ntr = 0, nt = 0, thin = 0;
    unsigned code = 0;
    bool vec2 = false;
};
static int pb_plan(hmg_ctx* c, int nz, int nm, int nk, const hmg_power_batch_desc* d, PbPlan* P) {
    REQUIRE(c && d && P, "NULL argument");
    const int ntr = d->ntr, npairs = d->npairs;
    const hmg_tracer* tr_in = d->h_tr;
    const int *pair_a = d->h_pair_a, *pair_b = d->h_pair_b;
    double* const* P1h = d->h_P1h;
    double* const* P2h = d->h_P2h;
    REQUIRE(tr_in && pair_a && pair_b && d->d_nzm && d->d_bh && d->d_ms && d->d_wm && d->d_ks, "NULL argument");
    REQUIRE(nz > 0 && nm > 0 && nk > 0, "empty grid");
    REQUIRE(nz <= 65535, "nz too large");
    REQUIRE(ntr >= 1 && ntr <= PB_MAXTR, "1..4 tracers per batch");
    REQUIRE(npairs >= 1, "no pairs");
    // Canonical tracer order: matter / pressure tracers first - those whose tensor an HOD of the batch uses for its
    // satellites before the others -, HOD tracers after them, each group in the caller's order.  A pair's sums do not depend on the tracers' order in the batch (the forms of a tracer involve only
    // its own tensors; the two 2-halo brackets are multiplied before P_lin), so this changes no result - it
    // only makes batches of the same structure look the same to the dispatch below.
    hmg_tracer tr[PB_MAXTR];
    int where[PB_MAXTR];
    {
        auto group = [&](int t) {
            if (tr_in[t].kind == HMG_TRACER_HOD) return 2;
            for (int u = 0; u < ntr; ++u)
                if (tr_in[u].kind == HMG_TRACER_HOD && tr_in[u].d_prof == tr_in[t].d_prof) return 0;
            return 1;
        };
        int n = 0;
        for (int pass = 0; pass < 3; ++pass)
            for (int t = 0; t < ntr; ++t)
                if (group(t) == pass) { tr[n] = tr_in[t]; where[t] = n++; }
    }
    std::vector<const double*> tens;
    BatchPrep& Q = P->Q;
    for (int t = 0; t < ntr; ++t) {
        REQUIRE(!tr[t].d_bias_override, "bias overrides are not supported in the batched kernel");
        if (fill_tracer(&tr[t], tens, &Q.tr[t])) return 1;
    }
    Q.ntr = ntr;
    Q.nt = (int)tens.size();
    Q.rho_m0 = d->rho_m0;
    REQUIRE(Q.nt >= 1 && Q.nt <= PW_MAXT, "more than 4 distinct profile tensors in one batch");
    // structure code of the batch (0 if a tracer has no compact form: an HOD with a central profile)
    unsigned code = 0;
    for (int t = 0; t < ntr; ++t) {
        const bool hod = Q.tr[t].kind == HMG_TRACER_HOD;
        if (hod && Q.tr[t].t_cprof >= 0) { code = 0; break; }
        code |= (hod ? PB_HOD(Q.tr[t].t_prof) : PB_LIN(Q.tr[t].t_prof)) << (4 * t);
    }
    if (getenv("HMG_PB_GENERIC")) code = 0;            // (testing: the generic forms for every batch)
    BatchArgs& A = P->A;
    for (int p = 0; p < PB_MAXPAIR; ++p) A.P1h[p] = A.P2h[p] = nullptr;
    bool any2 = false;
    for (int i = 0; i < npairs; ++i) {
        REQUIRE(pair_a[i] >= 0 && pair_a[i] < ntr && pair_b[i] >= 0 && pair_b[i] < ntr, "pair index out of range");
        int a = where[pair_a[i]], b = where[pair_b[i]];
        if (a > b) { int t = a; a = b; b = t; }
        const int p = a * ntr - a * (a - 1) / 2 + (b - a);
        REQUIRE((P1h && P1h[i]) || (P2h && P2h[i]), "pair without output");
        REQUIRE(!A.P1h[p] && !A.P2h[p], "the same unordered pair was requested twice");
        if (P1h && P1h[i]) A.P1h[p] = P1h[i];
        if (P2h && P2h[i]) { A.P2h[p] = P2h[i]; any2 = true; }
    }
    REQUIRE(!any2 || d->d_Pzk, "P2h needs Pzk");
    // the structures the kernel is compiled for; anything else runs the generic forms
    static const struct { int nt, ntr; unsigned code; } spec_list[] = {
#define PB_SPEC(NT_, NTR_, ...) {NT_, NTR_, pb_code(__VA_ARGS__)},
        PB_SPEC_LIST
#undef PB_SPEC
    };
    bool compiled = false;
    for (const auto& e : spec_list) compiled = compiled || (e.nt == Q.nt && e.ntr == ntr && e.code == code);
    if (!compiled) code = 0;
    Q.code = code;
    const int nc1 = 1 + Q.nt;
    const int stride = pb_stride(code, ntr, nc1);
    const int nblk = (nm + 63) / 64;
    REQUIRE(nblk <= 65535, "nm too large");
    if (ensure_scratch(c, 3, (size_t)nz * nm * stride * 8 + (size_t)nz * nblk * ntr * 2 * 8 + 64)) return 1;
    double* coef = (double*)c->scratch[3];
    double* sidep = coef + (size_t)nz * nm * stride;
    P->PA = PrepArgs{nm, nblk, Q, d->d_nzm, d->d_bh, d->d_ms, d->d_wm, coef, sidep};
    for (int i = 0; i < PW_MAXT; ++i) {
        A.tens[i] = i < Q.nt ? tens[i] : nullptr;
        A.nconst[i] = nullptr;
        A.cconst[i] = nullptr;
    }
    for (int t = 0; t < ntr; ++t) {   // constant-prefix hints travel with the tracer that names the tensor
        const int sp = Q.tr[t].t_prof, sc = Q.tr[t].t_cprof;
        REQUIRE((tr[t].d_prof_nconst == nullptr) == (tr[t].d_prof_cconst == nullptr) &&
                (tr[t].d_cprof_nconst == nullptr) == (tr[t].d_cprof_cconst == nullptr), "hint arrays come in pairs");
        if (sp >= 0 && tr[t].d_prof_nconst && !A.nconst[sp]) { A.nconst[sp] = tr[t].d_prof_nconst; A.cconst[sp] = tr[t].d_prof_cconst; }
        if (sc >= 0 && tr[t].d_cprof_nconst && !A.nconst[sc]) { A.nconst[sc] = tr[t].d_cprof_nconst; A.cconst[sc] = tr[t].d_cprof_cconst; }
    }
    for (int t = 0; t < PB_MAXTR; ++t) {
        A.ngal[t] = (t < ntr && tr[t].kind == HMG_TRACER_HOD) ? tr[t].d_ngal : nullptr;
        A.bias_const[t] = (t < ntr && tr[t].kind == HMG_TRACER_MATTER) ? 1.0 : 0.0;
    }
    A.nblk = nblk;
    A.coef = coef; A.sidep = sidep; A.ks = d->d_ks; A.Pzk = d->d_Pzk; A.kstar = d->kstar; A.nm = nm; A.nk = nk;
    bool vec2 = (nk % 2 == 0);
    for (int i = 0; i < Q.nt; ++i) vec2 = vec2 && (((uintptr_t)tens[i]) % 16 == 0);
    // thin z-slabs: narrower k tiles so that every CU still gets a workgroup, and sixteen wavefronts per
    // workgroup (one per virtual mass slice) so that each CU keeps twice the loads in flight.  The
    // summation order is the same in both shapes (see power_batch_kernel).
    // Shapes by the number of 128-k tiles the launch offers the chip (measured, MI355X, nm = 512, nk = 4096):
    // fewer than one per CU (nz = 4): 64-k tiles, 16 wavefronts (0.035 ms; 128-k tiles 0.039);
    // one to two per CU (nz = 8): 128-k tiles, 16 wavefronts (0.046 ms; 8 wavefronts 0.055);
    // more: 128-k tiles, 8 wavefronts, two workgroups per CU (16 wavefronts: +2 %).
    const long tiles128 = (long)((nk + 127) / 128) * nz;
    int thin = tiles128 < c->num_cu ? 1 : (tiles128 < 2L * c->num_cu && vec2 ? 3 : 0);
    if (const char* e = getenv("HMG_PB_THIN")) thin = atoi(e);      // tuning/testing: force a shape (2: V=1, 8 wavefronts)
    if (thin == 3 && !vec2) thin = 1;
    if (thin == 1 || thin == 2) vec2 = false;
    P->ntr = ntr; P->nt = Q.nt; P->thin = thin; P->code = code; P->vec2 = vec2;
    return 0;
}

static int pb_launch_main(hmg_ctx* c, const PbPlan& P, int nz) {
    const BatchArgs& A = P.A;
    const int thin = P.thin, ntr = P.ntr;
    const bool vec2 = P.vec2;
    const unsigned code = P.code;
#define PB_SHAPES(NT_, NTR_, CODE_)                                                        \
    if (thin == 3) return launch_power_batch<NT_, NTR_, 2, true, CODE_>(c, A, nz);         \
    return thin == 1 ? launch_power_batch<NT_, NTR_, 1, true, CODE_>(c, A, nz)             \
                     : (vec2 ? launch_power_batch<NT_, NTR_, 2, false, CODE_>(c, A, nz)    \
                             : launch_power_batch<NT_, NTR_, 1, false, CODE_>(c, A, nz));
    if (code) {
#define PB_SPEC(NT_, NTR_, ...)                                                            \
        if (P.nt == NT_ && ntr == NTR_ && code == pb_code(__VA_ARGS__)) { PB_SHAPES(NT_, NTR_, pb_code(__VA_ARGS__)) }
        PB_SPEC_LIST
#undef PB_SPEC
    }
#define PB_V(NT_, NTR_) PB_SHAPES(NT_, NTR_, 0u)
#define PB_NTR(NT_)                         \
    switch (ntr) {                          \
        case 1: PB_V(NT_, 1)                \
        case 2: PB_V(NT_, 2)                \
        case 3: PB_V(NT_, 3)                \
        case 4: PB_V(NT_, 4)                \
    }                                       \
    break;
    switch (P.nt) {
        case 1: PB_NTR(1)
        case 2: PB_NTR(2)
        case 3: PB_NTR(3)
        case 4: PB_NTR(4)
    }
#undef PB_NTR
#undef PB_V
#undef PB_SHAPES
    return fail("hmg_power_batch", "unreachable", __FILE__, __LINE__);
}

int hmg_power_batch_run(hmg_ctx* c, int nz, int nm, int nk, const hmg_power_batch_desc* d, int flags) {
    PbPlan P;
    if (pb_plan(c, nz, nm, nk, d, &P)) return 1;
    if (!(flags & HMG_PB_PREPARED)) {
        hipLaunchKernelGGL(power_batch_prep_kernel, dim3(nz, P.PA.nblk), dim3(64), 0, c->stream, P.PA);
        HIP_TRY(hipGetLastError());
    }
    return pb_launch_main(c, P, nz);
}

int hmg_power_batch(hmg_ctx* c, int nz, int nm, int nk, int ntr, const hmg_tracer* tr_in, int npairs,
                    const int* pair_a, const int* pair_b, const double* nzm, const double* bh,
                    const double* ms, const double* wm, const double* ks, const double* Pzk,
                    double rho_m0, double kstar, double* const* P1h, double* const* P2h) {
    const hmg_power_batch_desc d{ntr, tr_in, npairs, pair_a, pair_b, nzm, bh, ms, wm, ks, Pzk, rho_m0, kstar, P1h, P2h};
    return hmg_power_batch_run(c, nz, nm, nk, &d, 0);
}

// ---- grouped launches ---------------------------------------------------------------------------
static int rows_setup(int nz, int nm, const hmg_rows_part* rows, RowsArgs* out) {
    RowsArgs Rw{};
    REQUIRE(rows->d_m200c && rows->d_r200c && rows->d_rvir && rows->d_zs && rows->d_rhocz && rows->d_amp && rows->d_xc &&
                rows->d_alpha && rows->d_expo && rows->d_cmax && rows->d_rscale, "NULL argument in the rows part");
    REQUIRE(rows->kind == HMG_PROF_BATTAGLIA_GAS || rows->kind == HMG_PROF_BATTAGLIA_PRES, "unknown profile kind");
    REQUIRE(rows->kind != HMG_PROF_BATTAGLIA_PRES || (rows->d_hz && rows->d_post), "pressure needs d_hz and d_post");
    Rw.n = nz * nm; Rw.kind = rows->kind; Rw.nm = nm;
    Rw.m200 = rows->d_m200c; Rw.r200 = rows->d_r200c; Rw.rvir = rows->d_rvir; Rw.zs = rows->d_zs;
    Rw.rhoc = rows->d_rhocz; Rw.hz = rows->d_hz;
    for (int i = 0; i < 9; ++i) Rw.F.f[i] = rows->fit[i];
    Rw.gamma = rows->gamma; Rw.alpha_const = rows->alpha_const; Rw.pref = rows->amp_prefactor;
    Rw.post_pref = rows->post_prefactor;
    Rw.O = RowOut{rows->d_amp, rows->d_xc, rows->d_alpha, rows->d_expo, rows->d_cmax, rows->d_rscale, rows->d_post};
    if (rows->d_rowsc) {
        REQUIRE(rows->d_ks && rows->d_kts && rows->nk > 0 && rows->fft_m >= 2, "row scalars need d_ks, d_kts, nk and fft_m");
        Rw.O.rowsc = rows->d_rowsc; Rw.O.ks = rows->d_ks; Rw.O.kts = rows->d_kts; Rw.O.nk = rows->nk; Rw.O.M = rows->fft_m;
    }
    *out = Rw;
    return 0;
}

static int hod_args(int nm, const hmg_hod_part* hod, HodRowArgs* A) {
    const hmg_hod_params* p = hod->h_par;
    REQUIRE(p && hod->d_zs && hod->d_ms && hod->d_log10mstar_thresh && hod->d_Nc && hod->d_Ns && hod->d_NsNsm1 &&
                hod->d_NcNs, "NULL argument in the HOD part");
    REQUIRE(hod->stage == HMG_HOD_OCCUPATIONS || (hod->d_nzm && hod->d_bh && hod->d_wm && hod->d_ngal && hod->d_bg),
            "NULL argument in the HOD part");
    REQUIRE(p->corr == 0 || p->corr == 1, "corr must be 0 (max) or 1 (min)");
    REQUIRE((nm + 63) / 64 <= HOD_MAX_TILES, "nm too large for the HOD reduction (65536)");
    const HodDev P{p->sig_log_mstellar, p->alphasat, p->Bsat, p->betasat, p->Bcut, p->betacut, p->corr};
    *A = HodRowArgs{nm, P, hod->d_zs, hod->d_ms, hod->d_log10mstar_thresh, hod->d_nzm, hod->d_bh, hod->d_wm,
                    hod->d_Nc, hod->d_Ns, hod->d_NsNsm1, hod->d_NcNs, hod->d_ngal, hod->d_bg};
    return 0;
}

int hmg_sigma2_halo_front(hmg_ctx* c, int nz, int nm, int nq, const double* PT, const double* kq, const double* wq,
                          const double* R, double tswitch, const double* ms, const hmg_halo_stage_args* h,
                          const hmg_hod_part* hod, const hmg_rows_part* rows) {
    REQUIRE(c && PT && kq && wq && R, "NULL argument");
    REQUIRE(nz > 0 && nm > 0 && nq > 0, "empty grid");
    HaloStageArgs H;
    if (halo_stage_check(c, nz, nm, ms, h, &H)) return 1;
    HodRowArgs O{};
    int nocc = 0;
    if (hod) {
        REQUIRE(hod->stage == HMG_HOD_OCCUPATIONS, "only the occupations of an HOD need inputs only");
        if (hod_args(nm, hod, &O)) return 1;
        nocc = (nz * nm + 63) / 64;
    }
    RowsArgs Rw{};
    if (rows) {
        if (rows_setup(nz, nm, rows, &Rw)) return 1;
        REQUIRE(h->d_m2 && rows->d_m200c == h->d_m2 && rows->d_r200c == h->d_r2 && rows->d_rvir == h->d_rvir,
                "row parameters in the front launch take M_200c, R_200c, r_vir from the halo stage of the same call");
    }
    const int nseg = (nq + SIG_SEG_LEN - 1) / SIG_SEG_LEN;
    const int ztile = sigma2_ztile(nz), nzp = sigma2_nzp(nz);
    if (ensure_scratch(c, 4, (size_t)nseg * nz * nm * 8)) return 1;
    const int gx = (nm + 15) / 16;
    const size_t nsig = (size_t)gx * nseg * (nzp / ztile);
    const int nhalo = (nz * nm + 63) / 64;
    REQUIRE(nsig + nhalo + nocc <= 2147483647u, "grid too large");
    const SigmaFrontArgs G{nz, nzp, nm, nq, gx, nseg, PT, kq, wq, R, tswitch, (double*)c->scratch[4]};
    const dim3 grid((unsigned)(nsig + nhalo + nocc));
    if (ztile == 32)
        hipLaunchKernelGGL(front_group_kernel<2>, grid, dim3(64), 0, c->stream, G, H, nhalo, O, nocc, Rw);
    else
        hipLaunchKernelGGL(front_group_kernel<1>, grid, dim3(64), 0, c->stream, G, H, nhalo, O, nocc, Rw);
    HIP_TRY(hipGetLastError());
    c->sig_nz = nz; c->sig_nm = nm; c->sig_nq = nq;
    return 0;
}

// the optional links of a per-z chain from their parts; *n = 1 if there is any
static int chain_setup(int nm, const hmg_hod_part* hod, const PbPlan* prep, ChainArgs* C, int* n) {
    C->has_hod = C->has_prep = C->has_mf = C->mf_pad = 0;
    if (hod) {
        REQUIRE(hod->stage == HMG_HOD_SUMS, "a chain takes the n_gal, b_g sums of an HOD (its occupations ride with the front)");
        if (hod_args(nm, hod, &C->H)) return 1;
        C->has_hod = 1;
    }
    if (prep && prep->code) {           // (generic coefficient rows: their own launch, see hmg_group_profile)
        C->PA = prep->PA;
        C->has_prep = 1;
    }
    *n = (C->has_hod || C->has_prep) ? 1 : 0;
    return 0;
}
static int massfn_setup(hmg_ctx* c, int nz, int nm, int nq, const hmg_massfn_part* mf, SigmaMassFnArgs* S) {
    const hmg_massfn_params* p = mf->h_par;
    REQUIRE(p && mf->d_ms && mf->d_lnms && mf->d_sigma2 && mf->d_nzm && mf->d_bh, "NULL argument in the massfn part");
    REQUIRE(p->mode == HMG_MF_SHETH_TORMEN || p->mode == HMG_MF_TINKER10, "unknown mass function");
    REQUIRE(p->mode != HMG_MF_TINKER10 || mf->d_tinker_z, "Tinker mode needs d_tinker_z");
    REQUIRE(c->sig_nz == nz && c->sig_nm == nm && c->sig_nq == nq && c->scratch[4],
            "no sigma^2 partial sums of this shape in the context: call hmg_sigma2_halo_front first");
    const int nseg = (nq + SIG_SEG_LEN - 1) / SIG_SEG_LEN;
    const MassFnDev P{p->mode, p->deltac, p->st_A, p->st_a, p->st_p, p->rho_m0, p->lnm_uniform, p->lnm_step};
    *S = SigmaMassFnArgs{nz, nm, nseg, P, (const double*)c->scratch[4], mf->d_ms, mf->d_lnms, mf->d_tinker_z,
                         mf->d_sigma2, mf->d_nzm, mf->d_bh};
    return 0;
}

static int launch_rows_group(hmg_ctx* c, int nz, int nm, const ChainArgs& C, int nchain, const SigmaMassFnArgs* S,
                             const RowsArgs& Rw, const NfwArgs* N, size_t nfw_blocks) {
    const int nrowblk = Rw.n ? (Rw.n + 255) / 256 : 0;
    const int mf_ntile = (nm + MF_TILE - 1) / MF_TILE;
    const int nmfblk = S ? nz * mf_ntile : 0;
    const size_t blocks = (size_t)nchain + nmfblk + nrowblk + nfw_blocks;
    REQUIRE(blocks > 0 && blocks <= 2147483647u, "bad grid");
    const size_t lds = nchain ? chain_lds_doubles(nm) * 8 : 0;
    RowsGroupArgs G{};
    G.C = C; G.Rw = Rw; G.nchain = nchain; G.nrowblk = nrowblk; G.nmfblk = nmfblk; G.mf_ntile = mf_ntile;
    if (S) G.S = *S;
    const NfwArgs n = N ? *N : NfwArgs{};
    hipLaunchKernelGGL(rows_group_kernel, dim3((unsigned)blocks), dim3(256), lds, c->stream, G, n.T, n.acoef, n.ktile,
                       n.nm, n.nk, n.cs, n.rss, n.zs, n.ks, n.uk);
    HIP_TRY(hipGetLastError());
    return 0;
}

int hmg_group_rows(hmg_ctx* c, int nz, int nm, int nk, int nq, const hmg_massfn_part* mf, const hmg_hod_part* hod,
                   const hmg_rows_part* rows, const hmg_nfw_part* nfw) {
    REQUIRE(c, "NULL ctx");
    REQUIRE(nz > 0 && nm > 0, "empty grid");
    REQUIRE(mf || hod || rows || nfw, "empty group");
    REQUIRE(!(mf && hod), "an HOD needs the n, b of the massfn part: it cannot share its launch");
    REQUIRE((nm + 63) / 64 <= HOD_MAX_TILES, "nm too large");
    ChainArgs C;
    int one = 0;
    if (chain_setup(nm, hod, nullptr, &C, &one)) return 1;
    REQUIRE(one || mf || rows || nfw, "empty group");
    SigmaMassFnArgs S;
    if (mf && massfn_setup(c, nz, nm, nq, mf, &S)) return 1;
    RowsArgs Rw{};
    if (rows && rows_setup(nz, nm, rows, &Rw)) return 1;
    NfwArgs N{};
    size_t nfw_blocks = 0;
    int stop = -1;
    if (nfw) {
        REQUIRE(nk > 0, "empty grid");
        REQUIRE(nfw->d_cs && nfw->d_rs && nfw->d_zs && nfw->d_ks && nfw->d_nfw_series && nfw->d_uk, "NULL argument in the NFW part");
        const int ktile = 4096;
        nfw_blocks = (size_t)nz * nm * ((nk + ktile - 1) / ktile);
        N = NfwArgs{c->d_sici, nfw->d_nfw_series, ktile, nm, nk, nfw->d_cs, nfw->d_rs, nfw->d_zs, nfw->d_ks, nfw->d_uk};
        if (bracket_open(c, HMG_KERNEL_NFW, &stop)) return 1;
    }
    if (launch_rows_group(c, nz, nm, C, one ? nz : 0, mf ? &S : nullptr, Rw, nfw ? &N : nullptr, nfw_blocks)) return 1;
    return bracket_close(c, stop);
}

// The launches of a profile group once its chain is set up: the rows of the transform with the chain (and, in a tensor group,
// the NFW rows N) in their launch when the transform's route shares one, the chain as a launch of its own otherwise, the
// generic coefficient rows behind them.  *nfw_done = 1 if the NFW rows rode along.
static int profile_group_launches(hmg_ctx* c, int nz, int nm, int nk, const hmg_profile_fft_part* fft, const ChainArgs& C, int one,
                                  const hmg_power_batch_desc* prep, const PbPlan& P, const NfwArgs* N = nullptr,
                                  size_t nfw_blocks = 0, int* nfw_done = nullptr) {
    int chain_done = 0;
    if (fft && profile_fft_impl(c, nz, nm, nk, *fft, &C, one ? nz : 0, chain_lds_doubles(nm, C.has_mf != 0, 512) * 8, &chain_done,
                                nullptr, 0, nullptr, N, nfw_blocks, nfw_done))
        return 1;
    if (one && !chain_done) {      // no rows to share a launch with (or a length the in-LDS transform does not take)
        REQUIRE(!C.has_mf, "internal: the transform did not take the route the tensor group was set up for");
        const RowsArgs none{};
        if (launch_rows_group(c, nz, nm, C, nz, nullptr, none, nullptr, 0)) return 1;
    }
    if (prep && !P.code) {         // generic coefficient rows (register-hungry): a launch of their own
        hipLaunchKernelGGL(power_batch_prep_kernel, dim3(nz, P.PA.nblk), dim3(64), 0, c->stream, P.PA);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

int hmg_group_profile(hmg_ctx* c, int nz, int nm, int nk, const hmg_profile_fft_part* fft, const hmg_hod_part* hod,
                      const hmg_power_batch_desc* prep) {
    REQUIRE(c, "NULL ctx");
    REQUIRE(nz > 0 && nm > 0 && nk > 0, "empty grid");
    REQUIRE(fft || hod || prep, "empty group");
    REQUIRE((nm + 63) / 64 <= HOD_MAX_TILES, "nm too large");
    PbPlan P;
    if (prep && pb_plan(c, nz, nm, nk, prep, &P)) return 1;
    ChainArgs C;
    int one = 0;
    if (chain_setup(nm, hod, prep ? &P : nullptr, &C, &one)) return 1;
    return profile_group_launches(c, nz, nm, nk, fft, C, one, prep, P);
}

int hmg_group_tensors(hmg_ctx* c, int nz, int nm, int nk, int nq, const hmg_massfn_part* mf, const hmg_hod_part* hod,
                      const hmg_power_batch_desc* prep, const hmg_nfw_part* nfw, const hmg_profile_fft_part* fft) {
    REQUIRE(c, "NULL ctx");
    REQUIRE(nz > 0 && nm > 0 && nk > 0, "empty grid");
    REQUIRE(fft, "the tensor group is built around the rows of a profile transform: use hmg_group_rows without one");
    REQUIRE((nm + 63) / 64 <= HOD_MAX_TILES, "nm too large");
    PbPlan P;
    if (prep && pb_plan(c, nz, nm, nk, prep, &P)) return 1;
    ChainArgs C;
    int one = 0;
    if (chain_setup(nm, hod, prep ? &P : nullptr, &C, &one)) return 1;
    // Will the transform share its launch with a chain?  (One row in LDS with the row scalars of the rows stage and a chain
    // to ride with: the decision of profile_fft_impl for such a length.)  If not - long grids, the rocFFT route, no hint
    // arrays, nothing for a chain to do - the two groups run one after the other, as the two calls would.
    bool merge = c->use_tensor_group && c->use_fused_fft && fft->d_xs && ((uintptr_t)fft->d_xs % 16) == 0 && fft->d_rowsc &&
                 fft->d_nconst && FUSED_NT == 512 && (one || mf);
    if (merge) {
        FusedPlan* FP = nullptr;
        if (get_fused_plan(c, fft->nxs, &FP)) return 1;
        // ... and with a compile-time plan: the run-time-plan row kernel needs 80 registers (6 wavefronts per SIMD), and
        // NFW rows sharing that allocation lose more than the kernel boundary costs (MI355X, Config-3 grid, nxs = 3000
        // forced onto the run-time plan: 0.560 against 0.530 ms per step)
        merge = FP && !(FP->plan.M > c->fused_prefer_m && c->use_pruned_fft) && fused_ct_plan(c, FP->plan) != 0;
    }
    if (!merge) {
        if ((mf || nfw) && hmg_group_rows(c, nz, nm, nk, nq, mf, nullptr, nullptr, nfw)) return 1;
        return profile_group_launches(c, nz, nm, nk, fft, C, one, prep, P);
    }
    if (mf) {           // sigma^2 -> n, b as the first link of the chain: every later link reads what its own workgroup wrote
        if (massfn_setup(c, nz, nm, nq, mf, &C.S)) return 1;
        C.has_mf = 1;
        one = 1;
    }
    NfwArgs N{};
    size_t nfw_blocks = 0;
    if (nfw) {
        REQUIRE(nfw->d_cs && nfw->d_rs && nfw->d_zs && nfw->d_ks && nfw->d_nfw_series && nfw->d_uk, "NULL argument in the NFW part");
        const int ktile = 4096;
        nfw_blocks = (size_t)nz * nm * ((nk + ktile - 1) / ktile);
        N = NfwArgs{c->d_sici, nfw->d_nfw_series, ktile, nm, nk, nfw->d_cs, nfw->d_rs, nfw->d_zs, nfw->d_ks, nfw->d_uk};
    }
    int nfw_done = 0;
    if (profile_group_launches(c, nz, nm, nk, fft, C, one, prep, P, nfw ? &N : nullptr, nfw_blocks, &nfw_done)) return 1;
    REQUIRE(!nfw || nfw_done, "internal: the transform did not take the route the tensor group was set up for");
    return 0;
}

int hmg_limber(hmg_ctx* c, int nells, const double* ells, int nz, int nk, const double* zs,
               const double* ks, const double* P, const double* P2, int ngz, const double* gzs,
               const double* pref, const double* chis, const double* wz, double* out) {
    REQUIRE(c && ells && zs && ks && P && gzs && pref && chis && wz && out, "NULL argument");
    REQUIRE(nells > 0 && nz >= 1 && nk >= 2 && ngz >= 1, "bad sizes");
    hipLaunchKernelGGL(limber_kernel, grid1d((size_t)nells, LIMBER_E), dim3(LIMBER_G * LIMBER_E), 0, c->stream, nells, ells, nz,
                       nk, zs, ks, P, P2, ngz, gzs, pref, chis, wz, out);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---- function mirrors ------------------------------------------------------------------------
int hmg_fn2d(hmg_ctx* c, int op, int rows, int cols, int nin, const double* const* in, const int* sr,
             const int* sc, const double* par, int npar, double* out) {
    static const int need_in[HMG_FN_COUNT] = {1, 4, 2, 2, 4, 1, 2, 2, 1, 3, 3, 2, 2, 4, 4, 5, 4, 3, 1, 4, 4, 1, 4, 2, 3, 2, 3};
    static const int need_par[HMG_FN_COUNT] = {1, 3, 0, 1, 1, 2, 1, 1, 0, 0, 0, 4, 3, 12, 12, 14, 14, 0, 0, 0, 0, 4, 4, 1, 3, 10, 0};
    REQUIRE(c && in && sr && sc && out, "NULL argument");
    REQUIRE(op >= 0 && op < HMG_FN_COUNT, "unknown function id");
    REQUIRE(rows > 0 && cols > 0, "empty grid");
    REQUIRE(nin == need_in[op] && nin <= HMG_FN_MAXIN, "wrong number of inputs for this function");
    REQUIRE(npar == need_par[op] && npar <= HMG_FN_MAXPAR && (npar == 0 || par), "wrong number of parameters for this function");
    FnArgs A;
    A.op = op; A.rows = rows; A.cols = cols; A.out = out;
    for (int i = 0; i < HMG_FN_MAXIN; ++i) { A.in[i] = nullptr; A.sr[i] = 0; A.sc[i] = 0; }
    for (int i = 0; i < nin; ++i) {
        REQUIRE(in[i], "NULL input");
        REQUIRE(sr[i] >= 0 && sc[i] >= 0, "negative stride");
        A.in[i] = in[i]; A.sr[i] = sr[i]; A.sc[i] = sc[i];
    }
    for (int i = 0; i < HMG_FN_MAXPAR; ++i) A.par[i] = i < npar ? par[i] : 0.0;
    if (op == HMG_FN_TINKER_FNU || op == HMG_FN_TINKER_FSIGMA) REQUIRE(par[0] == 0.0 || par[2] >= 2.0, "alpha table needs >= 2 rows");
    if (op == HMG_FN_HOD_NSNSM1 || op == HMG_FN_HOD_NCNS) REQUIRE(par[0] == 0.0 || par[0] == 1.0, "corr must be 0 (max) or 1 (min)");
    hipLaunchKernelGGL(fn2d_kernel, grid1d((size_t)rows * cols, 256), dim3(256), 0, c->stream, A);
    HIP_TRY(hipGetLastError());
    return 0;
}

int hmg_mstellar_halo(hmg_ctx* c, int nz, int nm, const double* zs, const double* lmh, double* out) {
    REQUIRE(c && zs && lmh && out, "NULL argument");
    REQUIRE(nz > 0 && nm > 0, "empty grid");
    hipLaunchKernelGGL(mstellar_halo_kernel, dim3(nz), dim3(1024), 0, c->stream, nm, zs, lmh, out);
    HIP_TRY(hipGetLastError());
    return 0;
}

int hmg_add(hmg_ctx* c, size_t n, const double* a, const double* b, double* out) {
    REQUIRE(c && a && b && out, "NULL argument");
    if (!n) return 0;
    hipLaunchKernelGGL(add2_kernel, grid1d(n, 256), dim3(256), 0, c->stream, n, a, b, out);
    HIP_TRY(hipGetLastError());
    return 0;
}

int hmg_trapz_rows(hmg_ctx* c, int rows, int cols, const double* y, const double* x, double* out) {
    REQUIRE(c && y && x && out, "NULL argument");
    REQUIRE(rows > 0 && cols > 0, "empty grid");
    hipLaunchKernelGGL(trapz_rows_kernel, dim3(rows), dim3(256), 0, c->stream, cols, y, x, out);
    HIP_TRY(hipGetLastError());
    return 0;
}

int hmg_sine_transform(hmg_ctx* c, int rows, int n, const double* x, const double* y, double* uk) {
    REQUIRE(c && x && y && uk, "NULL argument");
    REQUIRE(rows > 0 && n >= 2, "bad sizes");
    const int nh1 = n / 2 + 1;
    // chunk the batch like the profile path so the work buffers stay bounded
    int chunk = (int)(((size_t)160 << 20) / ((size_t)n * 8 + (size_t)nh1 * 16));
    if (chunk < 1) chunk = 1;
    if (chunk > rows) chunk = rows;
    if (ensure_scratch(c, 0, (size_t)chunk * n * 8)) return 1;
    if (ensure_scratch(c, 1, (size_t)chunk * nh1 * 16)) return 1;
    double* fin = (double*)c->scratch[0];
    double2* fout = (double2*)c->scratch[1];
    // step = (x[-1]-x[0])/N needs the end points of the device grid
    double ends[2];
    HIP_TRY(hipMemcpyAsync(&ends[0], x, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(&ends[1], x + (n - 1), 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const double step = (ends[1] - ends[0]) / (double)n;
    for (int r0 = 0; r0 < rows; r0 += chunk) {
        const int nr = rows - r0 < chunk ? rows - r0 : chunk;
        hipLaunchKernelGGL(xy_kernel, grid1d((size_t)nr * n, 256), dim3(256), 0, c->stream, nr, n, x,
                           y + (size_t)r0 * n, fin);
        HIP_TRY(hipGetLastError());
        FftPlan* P = nullptr;
        if (get_plan(c, n, nr, &P)) return 1;
        void* ib[1] = {fin};
        void* ob[1] = {fout};
        FFT_TRY(rocfft_execution_info_set_stream(P->info, c->stream));
        FFT_TRY(rocfft_execute(P->plan, ib, ob, P->info));
        hipLaunchKernelGGL(neg_imag_kernel, grid1d((size_t)nr * nh1, 256), dim3(256), 0, c->stream,
                           (size_t)nr * nh1, step, (const double2*)fout, uk + (size_t)r0 * nh1);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

int hmg_profile_fft_table(hmg_ctx* c, int nz, int nm, int nk, int nxs, double step, const double* xs,
                          const double* kts, const double* rho, int rho_rows, const double* cmax,
                          const double* rss, const double* zs, const double* ks, int do_mass_norm,
                          double* out) {
    REQUIRE(c && xs && kts && rho && cmax && rss && zs && ks && out, "NULL argument");
    REQUIRE(nz > 0 && nm > 0 && nk > 0, "empty grid");
    REQUIRE(nxs >= 4, "nxs too small");
    REQUIRE(step > 0.0, "step must be positive");
    const int rows = nz * nm;
    REQUIRE(rho_rows == 1 || rho_rows == rows, "rho must have 1 or nz*nm rows");
    {   // the in-LDS routes of hmg_profile_fft with the table in the place of the family's integrand
        hmg_profile_fft_part p{};
        p.nxs = nxs; p.fft_step = step; p.d_xs = xs; p.d_kts = kts; p.d_cmax = cmax; p.d_rss = rss; p.d_zs = zs; p.d_ks = ks;
        p.do_mass_norm = do_mass_norm; p.d_out = out;
        p.amp_const = p.xc_const = p.alpha_const = p.expo_const = 1.0;
        int taken = 0;
        if (profile_fft_impl(c, nz, nm, nk, p, nullptr, 0, 0, nullptr, rho, rho_rows == 1, &taken)) return 1;
        if (taken) return 0;
    }
    const int nh = nxs / 2;
    const size_t per_row = (size_t)nxs * 8 + (size_t)(nh + 1) * 16;
    size_t budget = c->fft_chunk_bytes ? c->fft_chunk_bytes : ((size_t)160 << 20);
    int chunk = (int)(budget / per_row);
    if (chunk < 1) chunk = 1;
    if (chunk > rows) chunk = rows;
    if (ensure_scratch(c, 0, (size_t)chunk * nxs * 8)) return 1;
    if (ensure_scratch(c, 1, (size_t)chunk * (nh + 1) * 16)) return 1;
    if (ensure_scratch(c, 2, (size_t)chunk * 8)) return 1;
    double* fin = (double*)c->scratch[0];
    double2* fout = (double2*)c->scratch[1];
    double* mnorm = (double*)c->scratch[2];
    const bool stage = (size_t)nh * sizeof(double) <= 64 * 1024;
    const size_t lds = stage ? (size_t)nh * sizeof(double) : 0;
    for (int r0 = 0; r0 < rows; r0 += chunk) {
        const int nr = rows - r0 < chunk ? rows - r0 : chunk;
        hipLaunchKernelGGL(table_integrand_kernel, dim3(nr), dim3(256), 0, c->stream, nxs, r0, xs, rho,
                           (int)(rho_rows == 1), cmax, do_mass_norm, fin, mnorm);
        HIP_TRY(hipGetLastError());
        FftPlan* P = nullptr;
        if (get_plan(c, nxs, nr, &P)) return 1;
        void* ib[1] = {fin};
        void* ob[1] = {fout};
        FFT_TRY(rocfft_execution_info_set_stream(P->info, c->stream));
        FFT_TRY(rocfft_execute(P->plan, ib, ob, P->info));
        if (stage)
            hipLaunchKernelGGL(interp_kernel<true>, dim3(nr), dim3(256), lds, c->stream, nm, nk, nh, r0, step,
                               (const double2*)fout, kts, mnorm, rss, zs, ks, (const double*)nullptr, out,
                               (int*)nullptr, (double*)nullptr);
        else
            hipLaunchKernelGGL(interp_kernel<false>, dim3(nr), dim3(256), 0, c->stream, nm, nk, nh, r0, step,
                               (const double2*)fout, kts, mnorm, rss, zs, ks, (const double*)nullptr, out,
                               (int*)nullptr, (double*)nullptr);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}
