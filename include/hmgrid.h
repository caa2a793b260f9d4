/*
 * hmgrid.h — C ABI of libhmgrid.so, the MI355X (gfx950) halo-model grid engine.
 *
 * The reference (simonsobs/hmvec) is pure Python and has NO FFI/plugin boundary:
 * its drop-in boundary is the Python class hmvec.HaloModel (hmvec/hmvec.py:75-572).
 * This header defines the C boundary a maintainer would bind with ctypes underneath
 * that class (see INTEGRATION.md for the stub).  Each entry point cites the reference
 * lines it replaces.
 *
 * Conventions
 *  - All numeric data is IEEE fp64.  Grids are C-contiguous [z][m][k], k fastest
 *    (hmvec/hmvec.py:24-31).
 *  - Pointers named d_* are DEVICE pointers obtained from hmg_malloc; pointers
 *    named h_* are host pointers.  No framework types cross this boundary.
 *  - Every function returns 0 on success, non-zero on failure; the message for the
 *    calling thread's last failure is returned by hmg_last_error().
 *  - All work is enqueued on the context's stream; only hmg_memcpy_h2d, hmg_memcpy_d2h, hmg_sync,
 *    hmg_elapsed_ms, hmg_graph_destroy, hmg_comm_barrier and hmg_comm_destroy block the host.
 *  - A context is bound to one GPU and is not thread-safe; use one per GPU/process.
 */
#ifndef HMGRID_H
#define HMGRID_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HMG_ABI_VERSION 9

typedef struct hmg_ctx hmg_ctx;

/* ---- lifetime, memory, timing -------------------------------------------------- */
int         hmg_abi_version(void);
const char* hmg_last_error(void);
int hmg_ctx_create(int device, hmg_ctx** out);
int hmg_ctx_destroy(hmg_ctx* ctx);
/* Device blocks are recycled by size inside the context: hmg_free does not synchronise the device
 * (unless work was moved to another lane since the last synchronisation) and a stream of same-shaped
 * temporaries does not reach the driver's allocator.  Blocks are returned at hmg_ctx_destroy.      */
int hmg_malloc(hmg_ctx* ctx, size_t bytes, void** d_out);
int hmg_free(hmg_ctx* ctx, void* d_ptr);
int hmg_memcpy_h2d(hmg_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);
int hmg_memcpy_d2h(hmg_ctx* ctx, void* h_dst, const void* d_src, size_t bytes);   /* blocks */
int hmg_memcpy_d2d(hmg_ctx* ctx, void* d_dst, const void* d_src, size_t bytes);
int hmg_sync(hmg_ctx* ctx);                      /* waits for every lane */
/* Result hand-over without staging: a page-locked host block (allocate once, reuse) and an
 * asynchronous copy into it on the current lane; hmg_sync (or an event) completes it.  The
 * reference returns host arrays from get_power* (hmvec/hmvec.py:500-572); this is how a batch of
 * (nz,nk) spectra reaches the host in one DMA at link speed.                                      */
int hmg_host_alloc(hmg_ctx* ctx, size_t bytes, void** h_out);
int hmg_host_free(hmg_ctx* ctx, void* h_ptr);
int hmg_memcpy_d2h_async(hmg_ctx* ctx, void* h_pinned_dst, const void* d_src, size_t bytes);
/* The same towards the device (inputs of a sweep staged in a page-locked block), and the host-side wait for ONE
 * event - what a consumer of a streamed result block waits on while later passes keep running on the device
 * (hmg_sync would wait for those too).  The reference hands host arrays in and out (hmvec/hmvec.py:76-94,500-572). */
int hmg_memcpy_h2d_async(hmg_ctx* ctx, void* d_dst, const void* h_pinned_src, size_t bytes);
int hmg_event_synchronize(hmg_ctx* ctx, int slot);     /* blocks until the event last recorded in `slot` has happened */
/* Lanes: HMG_LANES HIP streams per context.  Every entry point enqueues on the CURRENT lane
 * (default 0).  Independent stages of the path (e.g. the NFW kernel and the profile-FFT kernel,
 * or the small per-(z,m) kernels) may be put on different lanes and ordered with events:
 * hmg_event_record(slot) on the producer lane, hmg_event_wait(slot) on the consumer lane.
 * hmg_memcpy_d2h, hmg_free and hmg_sync wait for all lanes.  Scratch inside the context is
 * shared: do not run the SAME entry point concurrently on two lanes.                        */
#define HMG_LANES 4
int hmg_lane_set(hmg_ctx* ctx, int lane);
int hmg_event_wait(hmg_ctx* ctx, int slot);      /* current lane waits for the event last recorded in slot */
/* HIP-event stopwatch on the context's stream: slots 0..HMG_EVENT_SLOTS-1. */
#define HMG_EVENT_SLOTS 4096
int hmg_event_record(hmg_ctx* ctx, int slot);
int hmg_elapsed_ms(hmg_ctx* ctx, int slot_start, int slot_stop, double* h_ms);     /* blocks */
/* One-shot: bracket the NEXT launch of the named kernel with event records at the two
 * slots (measurement hook for bench.py's roofline; -1,-1 clears).  For HMG_KERNEL_PROFILE_FFT
 * the bracket spans the integrand + rocFFT + interpolation launches of one hmg_profile_fft. */
#define HMG_KERNEL_POWER       0
#define HMG_KERNEL_NFW         1
#define HMG_KERNEL_PROFILE_FFT 2
#define HMG_KERNEL_COUNT       3
int hmg_bracket_next(hmg_ctx* ctx, int kernel_id, int slot_start, int slot_stop);
/* Captured steps.  Everything enqueued between hmg_graph_begin and hmg_graph_end - on lane 0 and on
 * every lane that joins through hmg_event_wait on an event recorded inside the capture - becomes one
 * HIP graph; hmg_graph_launch replays it with a single host call, independent branches running
 * concurrently.  No allocation, free or synchronisation may occur inside a capture (the calls that
 * would need one fail): run the same sequence once eagerly first.  A parameter sweep over a fixed
 * grid (the benchmark's step, an MCMC over HOD parameters held in device buffers) replays the graph. */
int hmg_graph_begin(hmg_ctx* ctx);
int hmg_graph_end(hmg_ctx* ctx, int* h_graph_id);
int hmg_graph_abort(hmg_ctx* ctx);                 /* leave capture mode after a failed call */
int hmg_graph_kernel_nodes(hmg_ctx* ctx, int graph_id, int* h_n);  /* kernel launches the captured step holds */
int hmg_graph_launch(hmg_ctx* ctx, int graph_id);  /* on the current lane */
int hmg_graph_destroy(hmg_ctx* ctx, int graph_id);

/* ---- A2: sigma^2(z, R(m)) ------------------------------------------------------
 * Replaces Cosmology.get_sigma2_R (hmvec/cosmology.py:245-269) + Wkr (:30-38).
 *   sigma2[z,m] = sum_j d_wq[j] * sPzk[z,j] * W(kq[j]*R[m])^2
 * d_wq carries the quadrature weight times k^2/(2 pi^2); the caller builds it from the
 * irregular-Simpson rule the reference uses (scipy.integrate.simpson, x given).
 * W is the Fourier top-hat with the Taylor branch for kR < taylor_switch.           */
int hmg_sigma2(hmg_ctx* ctx, int nz, int nm, int nq,
               const double* d_sPzk /*[nz][nq]*/, const double* d_kq /*[nq]*/,
               const double* d_wq /*[nq]*/, const double* d_R /*[nm]*/,
               double taylor_switch, double* d_sigma2 /*[nz][nm]*/);
/* The contraction reads P from a [k'][z] transposed, zero-padded copy (the MFMA A operand).  A caller
 * whose sPzk does not change between passes (the usual case: it is an INPUT of the path) lays it out
 * once - hmg_sigma2_layout_size doubles, hmg_sigma2_prepare - and calls hmg_sigma2_prepared, which
 * is hmg_sigma2 minus the transposition launch.  Same arithmetic, same result, bit for bit.        */
int hmg_sigma2_layout_size(int nz, int nq, size_t* h_doubles);
int hmg_sigma2_prepare(hmg_ctx* ctx, int nz, int nq, const double* d_sPzk, double* d_PT);
int hmg_sigma2_prepared(hmg_ctx* ctx, int nz, int nm, int nq, const double* d_PT,
                        const double* d_kq, const double* d_wq, const double* d_R,
                        double taylor_switch, double* d_sigma2);

/* ---- A3/A4: mass function n(z,m) and halo bias b(z,m) -----------------------------
 * Replaces get_fsigmaz/get_bh/get_nzm (hmvec/hmvec.py:133-161,178-185) and
 * tinker.f_nu/bias (hmvec/tinker.py:26-67).                                          */
#define HMG_MF_SHETH_TORMEN 0
#define HMG_MF_TINKER10     1
typedef struct {
    int    mode;               /* HMG_MF_* */
    double deltac, st_A, st_a, st_p;
    double rho_m0;             /* mean matter density today [Msun/Mpc^3] */
    int    lnm_uniform;        /* np.gradient takes its uniform-spacing branch */
    double lnm_step;           /* that spacing (only if lnm_uniform) */
} hmg_massfn_params;
int hmg_massfn(hmg_ctx* ctx, int nz, int nm, const hmg_massfn_params* h_par,
               const double* d_sigma2 /*[nz][nm]*/, const double* d_ms /*[nm]*/,
               const double* d_lnms /*[nm]*/,
               const double* d_tinker_z /*[nz][5] = alpha,beta,phi,eta,gamma at clamped z; NULL for ST*/,
               double* d_nzm /*[nz][nm]*/, double* d_bh /*[nz][nm]*/);

/* hmg_sigma2_prepared + hmg_massfn with the second stage of the contraction (the ordered sum over the
 * k' segments) folded into the mass-function launch: two launches instead of three, same sigma2 bits. */
int hmg_sigma2_massfn(hmg_ctx* ctx, int nz, int nm, int nq, const double* d_PT,
                      const double* d_kq, const double* d_wq, const double* d_R, double taylor_switch,
                      const hmg_massfn_params* h_par, const double* d_ms, const double* d_lnms,
                      const double* d_tinker_z, double* d_sigma2, double* d_nzm, double* d_bh);

/* ---- A5: concentration, virial radius, scale radius -----------------------------------
 * Replaces duffy_concentration / concentration / rvir (hmvec/hmvec.py:68-73,111-115,163-176).
 *   c = A (h m / 2e12)^alpha (1+z)^beta ;  rvir = (3 m / (4 pi delta[z] rho[z]))^(1/3) ;
 *   rs = rvir / c                                                                          */
int hmg_halo_structure(hmg_ctx* ctx, int nz, int nm, const double* d_ms, const double* d_zs,
                       const double* d_delta /*[nz]*/, const double* d_rho /*[nz]*/,
                       double duffy_A, double duffy_alpha, double duffy_beta, double h,
                       double* d_cs /*[nz][nm]*/, double* d_rvir /*[nz][nm]*/,
                       double* d_rs /*[nz][nm]*/);

/* hmg_halo_structure, the series rows of the analytic NFW kernel and hmg_mdelta_convert in ONE launch
 * (they are all one-thread-per-(z,m) stages that precede the profile kernels of a pass):
 *   d_nfw_series [nz][nm][HMG_NFW_SERIES_STRIDE] (or NULL): per-row coefficients of u_NFW's
 *     small-argument series and the row constants of its closed forms, to be handed to
 *     hmg_nfw_analytic, which otherwise computes them with a launch of its own;
 *   d_m2, d_r2 (both or NULL): the mass conversion of hmg_mdelta_convert with d_drho1, delta2, d_rho2.  */
#define HMG_NFW_SERIES_STRIDE 36
int hmg_halo_stage(hmg_ctx* ctx, int nz, int nm, const double* d_ms, const double* d_zs,
                   const double* d_delta /*[nz]*/, const double* d_rho /*[nz]*/,
                   double duffy_A, double duffy_alpha, double duffy_beta, double h,
                   double* d_cs, double* d_rvir, double* d_rs, double* d_nfw_series,
                   const double* d_drho1 /*[nz]*/, double delta2, const double* d_rho2 /*[nz]*/,
                   double* d_m2, double* d_r2);

/* Everything the constructor computes per (z,m) - hmg_sigma2_massfn and hmg_halo_stage (hmvec/hmvec.py:87-91,
 * 121-185 and :163-176, 225-227) - behind ONE launch after the sigma^2 contraction: the halo stage does not
 * depend on sigma^2, so its workgroups run beside the mass function's instead of waiting for a launch of
 * their own.  Same arithmetic, same results bit for bit as the two separate calls.                      */
typedef struct {
    const double *d_zs, *d_delta /*[nz]*/, *d_rho /*[nz]*/;
    double duffy_A, duffy_alpha, duffy_beta, h;
    double *d_cs, *d_rvir, *d_rs, *d_nfw_series /* or NULL */;
    const double* d_drho1 /*[nz]*/;
    double delta2;
    const double* d_rho2 /*[nz]*/;
    double *d_m2, *d_r2 /* both or NULL */;
} hmg_halo_stage_args;                 /* the arguments of hmg_halo_stage after d_ms, in its order */
int hmg_sigma2_massfn_halo(hmg_ctx* ctx, int nz, int nm, int nq, const double* d_PT,
                           const double* d_kq, const double* d_wq, const double* d_R, double taylor_switch,
                           const hmg_massfn_params* h_par, const double* d_ms, const double* d_lnms,
                           const double* d_tinker_z, double* d_sigma2, double* d_nzm, double* d_bh,
                           const hmg_halo_stage_args* h_halo);

/* ---- A7: mass-definition conversion -------------------------------------------------
 * Replaces mdelta_from_mdelta (hmvec/hmvec.py:748-798): the root in ln M2 of
 *   M1 F(c1) = M2 F(c2),  c2 = c1 ((M2/M1)(drho1/drho2))^(1/3),  F = 1/(ln(1+c)-c/(1+c)),
 * drho2[z] = delta2 * rho2[z].  Also returns r2 = (3 M2/(4 pi delta2 rho2))^(1/3)
 * (hmvec.py:225).                                                                       */
int hmg_mdelta_convert(hmg_ctx* ctx, int nz, int nm, const double* d_ms, const double* d_cs,
                       const double* d_drho1 /*[nz]*/, double delta2, const double* d_rho2 /*[nz]*/,
                       double* d_m2 /*[nz][nm]*/, double* d_r2 /*[nz][nm]*/);

/* ---- A6: analytic NFW u(k|m,z) --------------------------------------------------------
 * Replaces the Si/Ci branch of add_nfw_profile (hmvec/hmvec.py:346-353); Si/Ci follow
 * Cephes sici (scipy.special.sici).                                                     */
int hmg_nfw_analytic(hmg_ctx* ctx, int nz, int nm, int nk, const double* d_cs,
                     const double* d_rs, const double* d_zs, const double* d_ks,
                     const double* d_nfw_series /* from hmg_halo_stage for the SAME d_cs, or NULL */,
                     double* d_uk /*[nz][nm][nk]*/);

/* ---- A8/X1 row parameters of the generalised-NFW integrand ----------------------------
 * The three radial profiles of the path are one family
 *     f(x) = amp * (x/xc)^gamma * (1 + (x/xc)^alpha)^(-expo)
 *  NFW (hmvec.py:744): amp=1, xc=1, gamma=-1, alpha=1, expo=2
 *  Battaglia gas (hmvec.py:844-860): xc=1, alpha=alpha(m,z), expo=(beta+gamma)/alpha,
 *      amp=(Ob/Om) rho_c(z) rho0(m,z)
 *  Battaglia pressure (hmvec.py:906-927): xc=xc(m,z), alpha const, expo=beta(m,z),
 *      amp = eFrac (Ob/Om) 200 M G rho_c/(2 R200) P0(m,z)
 * with X(m,z) = A0 (m200c/1e14)^am (1+z)^az (hmvec.py:800-802).
 * fit[9] = {A0,am,az} x {rho0|P0, alpha|xc, beta}.  Outputs are [nz][nm].  The same launch
 * also emits the truncation radius cmax = rvir/rscale and rscale (hmvec.py:247-248,311-312)
 * and, for pressure, the y-conversion factor of hmvec.py:316.                              */
#define HMG_PROF_BATTAGLIA_GAS  1
#define HMG_PROF_BATTAGLIA_PRES 2
int hmg_profile_rowparams(hmg_ctx* ctx, int kind, int nz, int nm, const double* d_m200c,
                          const double* d_r200c, const double* d_rvir, const double* d_zs,
                          const double* d_rhocz /*[nz]*/, const double* d_hz /*[nz] H(z) in 1/Mpc, pressure only*/,
                          const double h_fit[9], double gamma, double alpha_const,
                          double amp_prefactor, double post_prefactor,
                          double* d_amp, double* d_xc, double* d_alpha, double* d_expo,
                          double* d_cmax /* rvir/rscale */, double* d_rscale /* R200c/2 (gas) or R200c (pressure) */,
                          double* d_post /* pressure: post_prefactor R200c^3 (1+z)^2/H(z); gas: may be NULL */);

/* hmg_mdelta_convert (with drho2 = delta2 * rhocz) + hmg_profile_rowparams in one launch; also
 * writes d_m200c / d_r200c.  Same results; one kernel boundary fewer per profile.           */
int hmg_profile_rows_from_mvir(hmg_ctx* ctx, int kind, int nz, int nm, const double* d_ms,
                               const double* d_cs, const double* d_rvir, const double* d_zs,
                               const double* d_drho1 /*[nz]*/, double delta2,
                               const double* d_rhocz /*[nz]*/, const double* d_hz /*[nz] or NULL*/,
                               const double h_fit[9], double gamma, double alpha_const,
                               double amp_prefactor, double post_prefactor,
                               double* d_m200c, double* d_r200c,
                               double* d_amp, double* d_xc, double* d_alpha, double* d_expo,
                               double* d_cmax, double* d_rscale, double* d_post);

/* ---- F1-F3: radial-profile sine transform + per-(z,m) k-interpolation ------------------
 * Replaces generic_profile_fft / fft_integral / _interp_loop (hmvec/fft.py:35-115),
 * bug-compatibly (step=(x[-1]-x[0])/N, 0-based DFT phase, left=u[first k>0], right=0,
 * strict x>cmax truncation).  Any of d_amp/d_xc/d_alpha/d_expo may be NULL meaning the
 * constant given in *_const.  d_xs [nxs] and d_kts [nxs/2+1] are the x grid
 * (linspace(0,xmax,nxs+1)[1:]) and the rfftfreq(nxs,fft_step)*2pi grid, built by the caller;
 * fft_step = (xs[-1]-xs[0])/nxs (hmvec/fft.py:45-47).  d_kts must be that uniform mode grid,
 * kts[j] = j * kts[1]: the in-LDS transform forms 1/kt_j as (1/j) / kt_1 and brackets the target
 * wavenumbers by division with the mode spacing.
 * out[z,m,k] *= d_post[z,m] if d_post != NULL (pressure prefactor, hmvec.py:316).
 *
 * Routes (chosen by the library from the radial grid and the rows' support; same results to <= 1e-12 in u):
 *   nxs = 1000, 2000, 4000, 5000      one (z,m) row per workgroup, packed-real FFT in LDS, compile-time plan;
 *   other even nxs <= 12288 (M = nxs/2 <= HMG_FUSED_MAX_M = 6144) whose half factors into 2, 3, 4, 5 with at most four
 *     butterflies per thread in every pass: the same with a run-time plan - for M > 2500 (HMG_FUSED_PREFER_M) only
 *     when the long-grid routes below do not apply (they are tried first there);
 *   longer grids - the ones the reference's own callers use: add_battaglia_profile(xmax=50, nxs=30000)
 *     (examples/lensing_baryons.py:27, bin/tests.py:308), numeric NFW nxs=40000 / xmax=200 (hmvec/params.py:59-60) -
 *     whose half is a multiple of a compiled sub-transform length LP >= the rows' support (profiles are cut at
 *     cmax << xmax): the long-grid kernels (R = nxs/2/LP pairs of length-LP transforms in LDS, or the chirp transform
 *     for rows that need few modes).  An EAGER call measures the support bound of its rows (one small kernel, a
 *     4-byte copy, one stream synchronisation) unless hmg_profile_support_epoch has tagged the arrays' contents;
 *     inside a captured step the bound on file for the same arrays is used (none on file: the one-row route if the
 *     length has one, else an error) and every row re-checks itself: a row beyond the bound is filled with NaN and
 *     the next synchronising call (hmg_sync, hmg_memcpy_d2h, hmg_event_synchronize) returns an error.  The tables of
 *     these routes do not depend on HMG_FUSED_MAX_M;
 *   long grids whose support does NOT prune but whose rows all need few modes (2 jn + 2 <= 1000..1250; the tSZ notebook's
 *     add_battaglia_pres_profile(xmax=2, nxs=30000)): the narrow-band kernel - nxs/2/LB transforms of length LB of the
 *     decimated row, one accumulator per needed mode; needs the hint arrays (ascending d_ks); the bound on the needed
 *     modes is measured and re-checked like the support bound;
 *   everything else (odd nxs, other prime factors, wide supports with many modes): integrand -> rocFFT R2C -> interpolation.
 * Environment switches for testing, read at hmg_ctx_create: HMG_FUSED_FFT=0 (rocFFT for everything), HMG_PRUNED_FFT=0,
 * HMG_CHIRP=0, HMG_BAND_FFT=0, HMG_PRUNED_LP_MIN, HMG_FUSED_MAX_M, HMG_FUSED_PREFER_M.                             */
int hmg_profile_fft(hmg_ctx* ctx, int nz, int nm, int nk, int nxs, double fft_step,
                    const double* d_xs, const double* d_kts,
                    const double* d_amp, const double* d_xc, const double* d_alpha,
                    const double* d_expo, double amp_const, double xc_const,
                    double alpha_const, double expo_const, double gamma,
                    const double* d_cmax /*[nz][nm]*/, const double* d_rss /*[nz][nm]*/,
                    const double* d_zs, const double* d_ks, int do_mass_norm,
                    const double* d_post /*[nz][nm] or NULL*/,
                    double* d_out /*[nz][nm][nk]*/,
                    int* d_nconst /*[nz][nm] or NULL*/, double* d_cconst /*[nz][nm] or NULL*/,
                    const double* d_logxs /*[nxs] from hmg_profile_fft_logx for the SAME d_xs, or NULL*/);
/* Tag of the CONTENTS of the d_cmax / d_rss / d_zs / d_ks arrays the following profile calls of this context will be
 * given (0, the default: no tag).  The long-grid routes size their launches from a bound on the rows' support and on
 * the modes they need; an untagged eager call measures that bound every time (a stream synchronisation).  With a
 * non-zero tag the bound measured by the first call with the same arrays, sizes and tag is reused - no host
 * synchronisation in later calls - on the caller's promise that equal tags mean equal contents.  A promise that
 * does not hold is caught, not computed through: every row re-checks itself, a row beyond the bound is filled with
 * NaN and the next synchronising call (hmg_sync, hmg_memcpy_d2h, hmg_event_synchronize) fails and drops the cached
 * bounds.  (The facade tags per model and mass grid: a model's concentrations and radii do not change.)          */
int hmg_profile_support_epoch(hmg_ctx* ctx, long long epoch);
/* ln xs[n]: the same for every (z,m) row, so a caller whose x grid does not change between passes
 * computes it once and hands it to hmg_profile_fft (one transcendental fewer per sample; without it
 * the kernel evaluates the logarithm itself, or builds the table per call when there are many rows). */
int hmg_profile_fft_logx(hmg_ctx* ctx, int nxs, const double* d_xs, double* d_logxs);
/* d_nconst / d_cconst (both or neither): constant-prefix hint of every output row for hmg_tracer -
 * the number of leading target wavenumbers below the row's first FFT mode and the value they all
 * receive.  With the hint arrays d_ks MUST be ascending (the caller's responsibility): the kernel finds the end of
 * that prefix by a search and fills it without loading or testing its wavenumbers; for a target grid in any
 * other order pass NULL for both.                                                                */

/* ---- H1-H3: HOD occupations ----------------------------------------------------------------
 * Replaces avg_Nc/avg_Ns/avg_NsNsm1/avg_NcNs, Mstellar_halo/Mhalo_stellar, get_ngal/get_bg
 * (hmvec/hmvec.py:634-731,462-466,936-957).  corr: 0 = "max", 1 = "min".                       */
typedef struct {
    double sig_log_mstellar, alphasat, Bsat, betasat, Bcut, betacut;
    int    corr;
} hmg_hod_params;
int hmg_hod(hmg_ctx* ctx, int nz, int nm, const hmg_hod_params* h_par, const double* d_zs,
            const double* d_ms, const double* d_log10mstar_thresh /*[nz]*/,
            const double* d_nzm, const double* d_bh, const double* d_wm /*[nm] trapz weights*/,
            double* d_Nc, double* d_Ns, double* d_NsNsm1, double* d_NcNs /*[nz][nm] each*/,
            double* d_ngal /*[nz]*/, double* d_bg /*[nz]*/);

/* ---- P1-P4: fused 1-halo + 2-halo mass integrals ---------------------------------------------
 * Replaces _get_matter/_get_hod/_get_hod_square/_get_pressure + get_power_1halo +
 * get_power_2halo (hmvec/hmvec.py:469-572) in ONE pass over the profile tensors:
 *   P1h[z,k] = trapz_m[n W_a W_b] (1-exp(-(k/kstar)^2))
 *   P2h[z,k] = Pzk (I_a + b_a - C_a)(I_b + b_b - C_b)
 * Either output pointer may be NULL.                                                          */
#define HMG_TRACER_MATTER   0
#define HMG_TRACER_HOD      1
#define HMG_TRACER_PRESSURE 2
typedef struct {
    int kind;                       /* HMG_TRACER_* */
    const double* d_prof;           /* matter: uk; pressure: pk; HOD: satellite uk   [nz][nm][nk] */
    const double* d_cprof;          /* HOD: central uk, or NULL meaning u_c == 1 */
    const double *d_Nc, *d_Ns, *d_NcNs, *d_NsNsm1, *d_ngal;   /* HOD only */
    const double* d_bias_override;  /* [nz] replaces b (b1_in/b2_in), or NULL */
    /* Optional constant-prefix hints for the two profile tensors (NULL = none): the first
     * nconst[z][m] entries of row (z,m) all equal cconst[z][m].  hmg_profile_fft can emit them
     * (np.interp's left fill, hmvec/fft.py:107: every target k below the first FFT mode gets the
     * same value - 63 % of a Battaglia tensor at Config 3).  hmg_power_batch then substitutes the
     * constant instead of reading those parts of the tensor; results are bit-identical.         */
    const int*    d_prof_nconst;   const double* d_prof_cconst;      /* [nz][nm] each */
    const int*    d_cprof_nconst;  const double* d_cprof_cconst;
} hmg_tracer;
int hmg_power(hmg_ctx* ctx, int nz, int nm, int nk, const hmg_tracer* h_a, const hmg_tracer* h_b,
              const double* d_nzm, const double* d_bh, const double* d_ms, const double* d_wm,
              const double* d_ks, const double* d_Pzk, double rho_m0, double kstar,
              double* d_P1h /*[nz][nk] or NULL*/, double* d_P2h /*[nz][nk] or NULL*/);

/* The terms get_power_2halo(verbose=True) prints (hmvec/hmvec.py:566-571): the two 2-halo
 * integrals I_a(z,k), I_b(z,k) [nz][nk] and their k -> 0 consistency limits d_C12[z] = {C_a, C_b}. */
int hmg_power_2halo_terms(hmg_ctx* ctx, int nz, int nm, int nk, const hmg_tracer* h_a, const hmg_tracer* h_b,
                          const double* d_nzm, const double* d_bh, const double* d_ms, const double* d_wm,
                          const double* d_ks, double rho_m0,
                          double* d_I1 /*[nz][nk]*/, double* d_I2 /*[nz][nk]*/, double* d_C12 /*[nz][2]*/);

/* Same integrals for SEVERAL spectra in one pass: ntr tracers (<= 4) over their distinct
 * profile tensors (<= 4), npairs (a,b) index pairs into h_tr.  Each distinct tensor is streamed
 * from HBM once for the whole batch instead of once per pair.  h_P1h[i] / h_P2h[i] are the
 * device outputs of pair i (either may be NULL).  The reference's first-name-only rule for
 * two DIFFERENT HOD (or two different pressure) names (hmvec.py:510-513) is not expressible
 * here: issue those pairs through hmg_power.  No bias overrides.                              */
int hmg_power_batch(hmg_ctx* ctx, int nz, int nm, int nk, int ntr, const hmg_tracer* h_tr,
                    int npairs, const int* h_pair_a, const int* h_pair_b,
                    const double* d_nzm, const double* d_bh, const double* d_ms, const double* d_wm,
                    const double* d_ks, const double* d_Pzk, double rho_m0, double kstar,
                    double* const* h_P1h, double* const* h_P2h);

/* ---- grouped launches: independent stages of a pass as block ranges of ONE grid ---------------------
 * The reference runs its stages one numpy expression after another (hmvec/hmvec.py:127-131, 188-250,
 * 318-355, 357-460, 504-572); which of them do NOT depend on each other is a property of the model:
 *     sigma^2 contraction, halo stage (c, r_vir, r_s, M_200c)          need inputs only
 *     n(z,m), b(z,m)        <- sigma^2           HOD  <- n, b
 *     NFW u(k), Battaglia row parameters <- halo stage    profile FFT rows <- row parameters
 *     coefficient rows of the mass integrals <- n, b, HOD;   mass integrals <- those + the tensors
 * A kernel boundary costs ~2 us and every per-(z,m) stage is latency-bound, so on a thin z-slab (a rank
 * of an 8-GPU job: 4 redshifts) the small stages take a third of the step when each is its own launch.
 * The three entry points below put independent stages into one launch each: short latency-bound
 * workgroups first in the grid, the chip-filling ones behind them.  Every part is optional (NULL); a
 * part has the arguments of its stand-alone entry point and produces the same bits.  Per-z CHAIN =
 * HOD (or only its n_gal, b_g sums) -> coefficient rows, each link optional, one workgroup per redshift. */
typedef struct {   /* second stage of the contraction + n, b: hmg_sigma2_massfn's arguments after taylor_switch */
    const hmg_massfn_params* h_par;
    const double *d_ms, *d_lnms, *d_tinker_z;
    double *d_sigma2, *d_nzm, *d_bh;
} hmg_massfn_part;
#define HMG_HOD_ALL         0   /* occupation numbers and the n_gal, b_g sums (= hmg_hod) */
#define HMG_HOD_OCCUPATIONS 1   /* <Nc>, <Ns>, <Ns(Ns-1)>, <NcNs> only: they depend on inputs alone (z, m, threshold) */
#define HMG_HOD_SUMS        2   /* n_gal, b_g from stored occupations and n, b */
typedef struct {   /* hmg_hod's arguments + which half */
    int stage;
    const hmg_hod_params* h_par;
    const double *d_zs, *d_ms, *d_log10mstar_thresh, *d_nzm, *d_bh, *d_wm;
    double *d_Nc, *d_Ns, *d_NsNsm1, *d_NcNs, *d_ngal, *d_bg;
} hmg_hod_part;
typedef struct {   /* hmg_profile_rowparams's arguments */
    int kind;
    const double *d_m200c, *d_r200c, *d_rvir, *d_zs, *d_rhocz, *d_hz;
    double fit[9], gamma, alpha_const, amp_prefactor, post_prefactor;
    double *d_amp, *d_xc, *d_alpha, *d_expo, *d_cmax, *d_rscale, *d_post;
    /* ABI 8, optional (d_rowsc NULL: none): the OUTPUT-side scalars of every profile row, worked out here once per row by
     * the thread that has just computed the row's length scale instead of by one wavefront of every row workgroup of
     * hmg_profile_fft (hmvec/fft.py:96-107: k_out = kt / (r_s (1+z)), np.interp's left fill below kt_1 / (r_s (1+z))).
     * d_rowsc [nz*nm][HMG_ROWSC_STRIDE]: 1/(rscale (1+z)), k_lo = kt_1 that, k_hi = kt_M that, 1/k_lo, 1/kt_1, and - packed
     * in one double, high word first - the number of FFT modes the target grid can reach and the length of the left-fill
     * prefix.  Needs d_ks ASCENDING ([nk]), d_kts ([fft_m + 1]), fft_m = nxs/2 of the transform that will read it.       */
    const double *d_ks, *d_kts;
    int nk, fft_m;
    double* d_rowsc;
} hmg_rows_part;
#define HMG_ROWSC_STRIDE 8
typedef struct {   /* hmg_nfw_analytic's arguments (d_nfw_series is required here) */
    const double *d_cs, *d_rs, *d_zs, *d_ks, *d_nfw_series;
    double* d_uk;
} hmg_nfw_part;
typedef struct {   /* hmg_profile_fft's arguments after nk */
    int nxs;
    double fft_step;
    const double *d_xs, *d_kts, *d_amp, *d_xc, *d_alpha, *d_expo;
    double amp_const, xc_const, alpha_const, expo_const, gamma;
    const double *d_cmax, *d_rss, *d_zs, *d_ks;
    int do_mass_norm;
    const double* d_post;
    double* d_out;
    int* d_nconst;
    double* d_cconst;
    const double* d_logxs;
    const double* d_rowsc;   /* ABI 8, optional: the row scalars an hmg_rows_part with the same d_ks, d_kts, nxs/2 left (needs the hint arrays) */
} hmg_profile_fft_part;
typedef struct {   /* hmg_power_batch's arguments after nk */
    int ntr;
    const hmg_tracer* h_tr;
    int npairs;
    const int *h_pair_a, *h_pair_b;
    const double *d_nzm, *d_bh, *d_ms, *d_wm, *d_ks, *d_Pzk;
    double rho_m0, kstar;
    double* const* h_P1h;
    double* const* h_P2h;
} hmg_power_batch_desc;
/* front: first stage of the sigma^2 contraction (hmg_sigma2_prepared's arguments; the partial sums stay
 * in the context until a massfn part consumes them) beside the halo stage and, optionally, the occupation
 * numbers of an HOD (stage HMG_HOD_OCCUPATIONS) and the row parameters of a Battaglia profile - everything
 * of a pass that needs inputs only.                                                                       */
int hmg_sigma2_halo_front(hmg_ctx* ctx, int nz, int nm, int nq, const double* d_PT, const double* d_kq,
                          const double* d_wq, const double* d_R, double taylor_switch,
                          const double* d_ms, const hmg_halo_stage_args* h_halo,
                          const hmg_hod_part* h_hod_occupations /* or NULL */,
                          const hmg_rows_part* h_rows /* or NULL: Battaglia row parameters from the M_200c, R_200c,
                                                         r_vir this call's halo stage computes (same arrays) */);
/* rows group: n(z,m), b(z,m) in 62-mass tiles | per-z chain (an HOD; not together with a massfn part, whose
 * n, b it would need) | Battaglia row parameters | analytic NFW rows.  A massfn part needs the partial sums
 * of the last hmg_sigma2_halo_front on this context with the same nz, nm, nq.                           */
int hmg_group_rows(hmg_ctx* ctx, int nz, int nm, int nk, int nq, const hmg_massfn_part* h_massfn,
                   const hmg_hod_part* h_hod, const hmg_rows_part* h_rows, const hmg_nfw_part* h_nfw);
/* profile group: per-z chain (an HOD or its sums -> the coefficient rows of the batched mass integrals
 * h_prep describes) | the rows of one hmg_profile_fft (lengths its in-LDS transform takes; otherwise the
 * parts are issued one after the other).  After a call with h_prep, hmg_power_batch_run(...,
 * HMG_PB_PREPARED) with the SAME description runs the mass integrals without their preparation launch. */
int hmg_group_profile(hmg_ctx* ctx, int nz, int nm, int nk, const hmg_profile_fft_part* h_fft,
                      const hmg_hod_part* h_hod, const hmg_power_batch_desc* h_prep);
/* tensor group: everything between the front and the mass integrals as ONE launch - per-z chain (sigma^2 second stage +
 * n, b of h_massfn -> the sums of h_hod -> the coefficient rows h_prep describes; each link optional) | the rows of
 * h_fft | the analytic NFW rows of h_nfw.  What hmg_group_rows(massfn, nfw) followed by hmg_group_profile(fft, hod,
 * prep) compute, bit for bit, with one kernel boundary less: the chain's first link is done per redshift by the
 * chain's own workgroup instead of by mass tiles of an earlier launch.  Lengths whose transform shares no launch
 * (long grids, rocFFT route, no row scalars) run as separate launches behind the same call.  h_fft is required;
 * a massfn part needs the partial sums of the last hmg_sigma2_halo_front, as in hmg_group_rows.  (ABI 9)         */
int hmg_group_tensors(hmg_ctx* ctx, int nz, int nm, int nk, int nq, const hmg_massfn_part* h_massfn /* or NULL */,
                      const hmg_hod_part* h_hod /* or NULL */, const hmg_power_batch_desc* h_prep /* or NULL */,
                      const hmg_nfw_part* h_nfw /* or NULL */, const hmg_profile_fft_part* h_fft);
#define HMG_PB_PREPARED 1
int hmg_power_batch_run(hmg_ctx* ctx, int nz, int nm, int nk, const hmg_power_batch_desc* h_desc, int flags);

/* d_out[i] = d_a[i] + d_b[i]: HaloModel.get_power = get_power_1halo + get_power_2halo (hmvec/hmvec.py:500-502)
 * summed on the device, so that one (nz,nk) array crosses PCIe instead of two.                          */
int hmg_add(hmg_ctx* ctx, size_t n, const double* d_a, const double* d_b, double* d_out);

/* ---- N1: Limber projection ------------------------------------------------------------------
 * Replaces limber_integral (hmvec/cosmology.py:867-904): for every multipole,
 *   C_ell = sum_g wz[g] * pref[g] * P(z = gzs[g], k = (ell + 1/2)/chis[g]),
 * P bilinear in (z,k) on the (nz,nk) grid and clamped to it; pref = H W1 W2 / chi^2 built by
 * the caller; wz = trapezoid weights over gzs (a single 1 for a delta-function window).
 * nz == 1 does 1-D interpolation in k.                                                        */
int hmg_limber(hmg_ctx* ctx, int nells, const double* d_ells, int nz, int nk, const double* d_zs,
               const double* d_ks, const double* d_Pzk,
               const double* d_Pzk2 /* optional second (nz,nk) array added to d_Pzk on the fly
                                       (P_1h + P_2h without materialising the sum), or NULL */,
               int ngz, const double* d_gzs,
               const double* d_pref, const double* d_chis, const double* d_wz, double* d_out);

/* ---- module-level helpers of the path -------
 * The reference exports its building blocks as free functions (hmvec/__init__.py:1 star-import);
 * its own tests call them (bin/tests.py:11,27,268-295).  hmg_fn2d evaluates one of them over a
 * (rows, cols) grid: element (r,c) of input i is d_in[i][r*h_sr[i] + c*h_sc[i]], so full arrays
 * (sr=cols, sc=1), per-row (1,0), per-column (0,1) and scalar (0,0) operands broadcast as numpy
 * would.  h_par are host scalars.  One fp64 output [rows][cols].                                 */
#define HMG_FN_TINKER_BIAS    0  /* in nu; par delta                              tinker.py:26-40 */
#define HMG_FN_TINKER_FNU     1  /* in nu, z, table_z[nt], table_alpha[nt] (strides ignored for
                                    the two tables); par norm_consistency, alpha, nt  tinker.py:43-67 */
#define HMG_FN_MHALO_STELLAR  2  /* in z, log10mstellar                        hmvec.py:648-695 */
#define HMG_FN_HOD_NC         3  /* in log10mstar(m), log10mstar_thresh; par sigma  hmvec.py:698-703 */
#define HMG_FN_HOD_NS         4  /* in Nc, log10mhalo, Msat, Mcut; par alphasat   hmvec.py:708-716 */
#define HMG_FN_HOD_MFUNC      5  /* in log10mthresh; par Bamp, Bind                 hmvec.py:706 */
#define HMG_FN_HOD_NSNSM1     6  /* in Nc, Ns; par corr (0 max, 1 min)            hmvec.py:719-725 */
#define HMG_FN_HOD_NCNS       7  /* in Nc, Ns; par corr                           hmvec.py:727-731 */
#define HMG_FN_FCON           8  /* in c                                            hmvec.py:737 */
#define HMG_FN_RHO_NFW        9  /* in r, rhoscale, rs                            hmvec.py:744-746 */
#define HMG_FN_R_FROM_M      10  /* in M, rho, delta                              hmvec.py:627-628 */
#define HMG_FN_DUFFY         11  /* in m, z; par A, alpha, beta, h                 hmvec.py:68-73 */
#define HMG_FN_BATT_FIT      12  /* in m200c, z; par A0, alpha_m, alpha_z         hmvec.py:800-802 */
#define HMG_FN_RHO_GAS_X     13  /* in x, m200c, z, rhocritz; par omb, omm, gamma, fit[9]  :844-860 */
#define HMG_FN_RHO_GAS_R     14  /* in r, m200c, z, rhocritz; same par                    :819-842 */
#define HMG_FN_PE_X          15  /* in x, m200c, R200c, z, rhocritz; par omb, omm, alpha, gamma,
                                    fit[9], G_newt                                hmvec.py:906-927 */
#define HMG_FN_PE_R          16  /* in r, m200c, z, rhocritz; same par            hmvec.py:881-904 */
#define HMG_FN_NGAL_INTEGRAND 17 /* in nzm, Nc, Ns -> nzm*(Nc+Ns)                   hmvec.py:956 */
#define HMG_FN_A2Z           18  /* in a                                            hmvec.py:933 */
#define HMG_FN_MDELTA        19  /* in M1, c1, delta_rho1, delta_rho2 -> M2        hmvec.py:748-798 */
#define HMG_FN_BG_INTEGRAND  20  /* in nzm, Nc, Ns, bh -> nzm*(Nc+Ns)*bh            hmvec.py:464-466 */
#define HMG_FN_ST_FSIGMA     21  /* in sigma2; par st_A, st_a, st_p, deltac  (get_fsigmaz, ST)  hmvec.py:136-141 */
#define HMG_FN_TINKER_FSIGMA 22  /* in sigma2, z, table_z, table_alpha; par as TINKER_FNU + deltac:
                                    nu f_nu(nu, z) with nu = deltac/sigma   (get_fsigmaz)      hmvec.py:142-145 */
#define HMG_FN_WKR           23  /* in k, R; par taylor_switch: Fourier top-hat W(kR)    cosmology.py:30-38 */
#define HMG_FN_LINCOMB3      24  /* in X0, X1, X2; par a, b, c -> a X0 + b X1 + c X2
                                    (total_matter_power_spectrum etc.)                 cosmology.py:599-658 */
#define HMG_FN_MHALO_STELLAR_CORE 25 /* in log10mstellar, a; par Mstar00, Mstara, M1, M1a, beta0, beta_a, gamma0,
                                       gamma_a, delta0, delta_a                         hmvec.py:648-657 */
#define HMG_FN_BRUTE_INTEGRAND 26 /* in r, rho, k -> 4 pi r sin(r k) rho / k  (uk_brute_force)           fft.py:22-33 */
#define HMG_FN_COUNT         27
#define HMG_FN_MAXIN   6
#define HMG_FN_MAXPAR 16
int hmg_fn2d(hmg_ctx* ctx, int op, int rows, int cols, int nin, const double* const* h_d_in,
             const int* h_sr, const int* h_sc, const double* h_par, int npar, double* d_out);

/* Mstellar_halo (hmvec/hmvec.py:634-646): per-redshift inverse of the SHMR through the
 * reference's 4000-point table linspace(-18,18,4000) and np.interp (clamped ends).
 * d_log10mhalo [nm] is shared by every z, as in the reference (it reads row 0).  -> [nz][nm]    */
int hmg_mstellar_halo(hmg_ctx* ctx, int nz, int nm, const double* d_zs, const double* d_log10mhalo,
                      double* d_out);

/* np.trapz(y, x, axis=-1) for y [rows][cols], x [cols] (ngal_from_mthresh hmvec.py:957, the
 * mass normalisation of uk_fft fft.py:15).  -> [rows]                                           */
int hmg_trapz_rows(hmg_ctx* ctx, int rows, int cols, const double* d_y, const double* d_x, double* d_out);

/* fft_integral (hmvec/fft.py:35-51): uk_j = -Im(rfft(x*y))_j * step, step = (x[n-1]-x[0])/n, for
 * every row of y [rows][n]; d_uk [rows][n/2+1].  The wavenumbers 2*pi*rfftfreq(n, step) are a
 * host one-liner and are left to the caller.                                                    */
int hmg_sine_transform(hmg_ctx* ctx, int rows, int n, const double* d_x, const double* d_y, double* d_uk);

/* generic_profile_fft (hmvec/fft.py:56-94) for a TABULATED integrand: d_rho is rho(x) on
 * xs = linspace(0,xmax,nxs+1)[1:], either one row shared by all (z,m) (rho_rows == 1) or one row
 * per (z,m) (rho_rows == nz*nm) - the two shapes the reference accepts (fft.py:75-78).  Truncation
 * |x| > cmax, trapz mass norm, sine transform, k scaling and interpolation as hmg_profile_fft.  */
int hmg_profile_fft_table(hmg_ctx* ctx, int nz, int nm, int nk, int nxs, double fft_step,
                          const double* d_xs, const double* d_kts, const double* d_rho, int rho_rows,
                          const double* d_cmax, const double* d_rss, const double* d_zs,
                          const double* d_ks, int do_mass_norm, double* d_out);

/* ---- z-slab gather over RCCL/xGMI -------------------------------------------------
 * One communicator per context.  The 128-byte id comes from hmg_comm_unique_id on rank 0
 * and is distributed by the caller (file, socket, MPI, ...).                                     */
#define HMG_COMM_ID_BYTES 128
int hmg_comm_unique_id(char h_id[HMG_COMM_ID_BYTES]);
int hmg_comm_init(hmg_ctx* ctx, const char h_id[HMG_COMM_ID_BYTES], int rank, int nranks);
int hmg_comm_allgather(hmg_ctx* ctx, const double* d_send, double* d_recv, size_t count_per_rank);
/* n gathers of count_per_rank doubles each, fused into one RCCL group launch. */
int hmg_comm_allgather_multi(hmg_ctx* ctx, int n, const double* const* h_d_send,
                             double* const* h_d_recv, size_t count_per_rank);
/* The gather of one pass off the compute stream: record ready_slot on the current lane, make
 * comm_lane wait for it, issue the grouped all-gather there and record done_slot.  A later
 * hmg_event_wait(done_slot) on the compute lane orders the next overwrite of the send buffers.   */
int hmg_comm_gather_async(hmg_ctx* ctx, int n, const double* const* h_d_send, double* const* h_d_recv,
                          size_t count_per_rank, int ready_slot, int done_slot, int comm_lane);
/* Slabs of unequal length (nz not a multiple of the number of ranks: SURVEY 8e partitions "contiguous z-slabs";
 * the README grid has nz = 20, /root/reference/README.rst:55): rank r contributes h_counts[r] doubles per array,
 * which land at the prefix-sum offset of every rank's receive buffer - one ncclBroadcast per (array, rank) in one
 * group launch.  h_counts has one entry per rank and must be the same on every rank; equal counts take the
 * all-gather of the entry points above. */
int hmg_comm_allgatherv_multi(hmg_ctx* ctx, int n, const double* const* h_d_send, double* const* h_d_recv,
                              const size_t* h_counts);
int hmg_comm_gatherv_async(hmg_ctx* ctx, int n, const double* const* h_d_send, double* const* h_d_recv,
                           const size_t* h_counts, int ready_slot, int done_slot, int comm_lane);
/* rank and size as RCCL reports them (ncclCommUserRank / ncclCommCount); 0 and 1 without a communicator */
int hmg_comm_info(hmg_ctx* ctx, int* h_rank, int* h_nranks);
int hmg_comm_barrier(hmg_ctx* ctx);     /* blocks */
int hmg_comm_destroy(hmg_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* HMGRID_H */
