// libhmgrid — MI355X (gfx950 / CDNA4) kernels + C ABI for the halo-model grid hot path.
// Boundary and reference citations: include/hmgrid.h.  Design notes: DESIGN.md.
//
// Everything here is fp64 and either HBM-bandwidth or fp64-VALU bound; the one dense contraction
// of the path (sigma^2: a (z x k') . (k' x m) product) runs on the fp64 matrix cores
// (v_mfma_f64_16x16x4_f64).  Layout is [z][m][k] with k fastest: a wavefront (64 lanes) always
// walks consecutive k, so every tensor access is a fully coalesced 512 B (or 1 KiB with double2)
// wave transaction, and per-(z,m) scalars are wave-uniform.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <rocfft/rocfft.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/hmgrid.h"
#include "fastmath.hpp"
#include "ldsfft.hpp"
#include "rowdev.hpp"
#include "longgrid.hpp"
#include "sici.hpp"

// ------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

static int fail(const char* what, const char* detail, const char* file, int line) {
    char buf[512];
    snprintf(buf, sizeof(buf), "%s: %s (%s:%d)", what, detail, file, line);
    g_last_error = buf;
    return 1;
}
// (a failed runtime call also leaves a sticky "last error" behind: clear it, or the next
// hipGetLastError() check after a perfectly good kernel launch would report it again)
#define HIP_TRY(expr)                                                                   \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess) {                                                         \
            (void)hipGetLastError();                                                    \
            return fail(#expr, hipGetErrorString(e_), __FILE__, __LINE__);              \
        }                                                                               \
    } while (0)
#define FFT_TRY(expr)                                                                   \
    do {                                                                                \
        rocfft_status s_ = (expr);                                                      \
        if (s_ != rocfft_status_success) {                                              \
            char m_[32];                                                                \
            snprintf(m_, sizeof(m_), "rocfft status %d", (int)s_);                      \
            return fail(#expr, m_, __FILE__, __LINE__);                                 \
        }                                                                               \
    } while (0)
#define NCCL_TRY(expr)                                                                  \
    do {                                                                                \
        ncclResult_t r_ = (expr);                                                       \
        if (r_ != ncclSuccess) return fail(#expr, ncclGetErrorString(r_), __FILE__, __LINE__); \
    } while (0)
#define REQUIRE(cond, msg)                                                              \
    do {                                                                                \
        if (!(cond)) return fail("invalid argument", msg, __FILE__, __LINE__);          \
    } while (0)

// ------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------
struct FftPlan {
    rocfft_plan plan = nullptr;
    rocfft_execution_info info = nullptr;
    void* work = nullptr;
    size_t work_bytes = 0;
};

namespace hmg { struct UnpackTw; }
struct FusedPlan {
    hmg::FftPlanDev plan;
    hmg::cplx* twM = nullptr;
    hmg::UnpackTw* twN = nullptr;
    int maxb = 0, maxp = 0;
};

struct PrunedPlan {                       // tables of the long-grid routes, per (nxs, LP); LP = 0: the mode-ordered pair
    hmg::cplx* twB = nullptr;             // LP == 0: exp(-2 pi i t / M), t < M = nxs/2 (narrow-band route)
    hmg::UnpackTw* twN = nullptr;         // LP == 0: unpack constants for j <= M/2 by mode (chirp and narrow-band routes)
    hmg::cplx* twR = nullptr;             // LP > 0: the residues' twiddles on the samples, by residue (ldsfft.hpp)
    hmg::UnpackTw* twNr = nullptr;        // LP > 0: unpack constants by residue
};
struct ChirpPlan {                        // tables of the chirp route (ldsfft.hpp: ChirpTables), per (nxs, LP, p0)
    hmg::cplx *chP = nullptr, *chJ = nullptr, *Bw = nullptr;
    int Jw = 0;
};
struct SupportKey {                       // what a measured bound on a launch's rows was measured for
    const void *cmax, *xs, *rss, *ks;
    int rows, nxs, nk;
    long long epoch;                      // the caller's tag of the arrays' CONTENTS (hmg_profile_support_epoch), 0 = none
    bool operator<(const SupportKey& o) const {
        return std::tie(cmax, xs, rss, ks, rows, nxs, nk, epoch) < std::tie(o.cmax, o.xs, o.rss, o.ks, o.rows, o.nxs, o.nk, o.epoch);
    }
};

struct hmg_ctx {
    int device = 0;
    hipStream_t stream = nullptr;             // stream of the current lane
    hipStream_t lanes[HMG_LANES] = {};        // lane 0 is the main stream
    int lane = 0;
    hipEvent_t ev[HMG_EVENT_SLOTS] = {};
    int bracket[HMG_KERNEL_COUNT][2];  // one-shot event brackets per kernel id, -1 = off
    // grow-only scratch arenas (device)
    void* scratch[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t scratch_bytes[7] = {0, 0, 0, 0, 0, 0, 0};
    std::map<std::pair<int, int>, FftPlan> plans;  // (nxs, batch) -> plan
    std::map<int, struct FusedPlan> fused;          // nxs -> workgroup-FFT tables
    size_t fft_chunk_bytes = 0;                    // 0 = default
    int use_fused_fft = 1;                         // HMG_FUSED_FFT=0 forces the rocFFT path
    int use_pruned_fft = 1;                        // HMG_PRUNED_FFT=0: long grids go to rocFFT as before round 4
    int fused_max_m = 6144;                        // HMG_FUSED_MAX_M: longest packed row the one-row-in-LDS kernel takes
    int fused_prefer_m = 2500;                     // HMG_FUSED_PREFER_M: above this the long-grid route is tried first
    int pruned_lp_min = 0;                         // HMG_PRUNED_LP_MIN: smallest sub-transform length to consider
    int use_chirp = 1;                             // HMG_CHIRP=0: every row of a long grid takes the decomposition
    int use_band_fft = 1;                          // HMG_BAND_FFT=0: supports that do not prune go to rocFFT
    int fused_generic = 0;                         // HMG_FUSED_GENERIC=1 (testing): the run-time plan for every one-row length
    int force_gatherv = 0;                         // HMG_FORCE_GATHERV=1 (testing): no all-gather shortcut for equal slab lengths
    std::map<std::tuple<int, int, int>, ChirpPlan> chirp;   // (nxs, LP, p0) -> tables
    std::map<std::pair<int, int>, PrunedPlan> pruned;   // (nxs, LP) -> tables of the long-grid routes
    std::map<int, hmg::cplx*> pass_tw;             // L -> per-pass twiddle table of the length-L plan (ldsfft.hpp)
    std::map<SupportKey, std::pair<int, int>> support;   // last measured bounds of a launch's rows: (support in packed samples, needed modes)
    // word a kernel raises when it cannot do what it was launched for: ONE word in page-locked host memory that the
    // device writes directly (a system-scope store), so that a host that has waited for the kernel - through whichever
    // stream, lane or event - reads it without a copy and without a question of which stream the copy belongs to
    int* h_fault = nullptr;
    int* d_fault = nullptr;                        // the device's address of the same word
    long long support_epoch = 0;                   // hmg_profile_support_epoch: tag of the contents of cmax / rss / ks arrays
    int sig_nz = 0, sig_nm = 0, sig_nq = 0;        // shape of the partial sums the last sigma^2 contraction left in scratch[4]
    ncclComm_t comm = nullptr;
    int comm_rank = 0, comm_size = 1;
    double* d_barrier = nullptr;
    hmg::SiciTable* d_sici = nullptr;  // Si/Ci coefficients, read through the scalar cache
    void* pinned[2] = {nullptr, nullptr};   // host bounce buffers for pageable <-> device copies
    // small host -> device copies: a ring of pinned slots, so that an upload is a memcpy + an asynchronous DMA
    // and the host does not wait for the stream (a model's constructor makes ~20 of these)
    static constexpr int UP_SLOTS = 32;
    static constexpr size_t UP_SLOT_BYTES = (size_t)256 << 10;
    char* up_ring = nullptr;
    hipEvent_t up_ev[UP_SLOTS] = {};
    int up_next = 0;
    hipEvent_t pin_ev[2] = {nullptr, nullptr};
    int num_cu = 256;
    // device blocks handed back by hmg_free, kept for reuse by size: dropping an array never
    // synchronises the device and a steady stream of same-shaped temporaries never reaches hipMalloc
    std::multimap<size_t, void*> free_blocks;
    std::map<void*, size_t> block_bytes;           // every live or cached block from hmg_malloc
    size_t cached_bytes = 0;
    bool lanes_dirty = false;                      // work was enqueued on a lane other than 0 since the last sync
    // captured steps (hmg_graph_*)
    bool capturing = false;
    std::vector<void*> freed_in_capture;           // hmg_free calls that arrived during a capture ...
    std::map<int, std::vector<void*>> graph_blocks; // ... stay with the graph that may use them until it is destroyed
    std::map<int, hipGraphExec_t> graphs;
    std::map<int, int> graph_kernels;              // kernel nodes per captured graph
    int next_graph_id = 1;
    hmg_ctx() { for (auto& b : bracket) b[0] = b[1] = -1; }
};
constexpr size_t FREE_CACHE_LIMIT = (size_t)4 << 30;   // bytes kept in the free list before real frees

static int rocfft_refcount = 0;

// events are created on first use (a context rarely needs more than a handful of the slots)
static int event_at(hmg_ctx* c, int slot, hipEvent_t* out) {
    if (!c->ev[slot]) HIP_TRY(hipEventCreate(&c->ev[slot]));
    *out = c->ev[slot];
    return 0;
}

// A kernel that finds it cannot do what it was launched for (a row whose support exceeds the plan the launch was
// sized for) raises the context's fault word instead of writing wrong numbers quietly; synchronising calls report it.
static int check_fault(hmg_ctx* c) {
    if (!*(volatile int*)c->h_fault) return 0;
    *(volatile int*)c->h_fault = 0;
    c->support.clear();
    return fail("device fault", "a profile row's support exceeded the bound its launch was sized for (the rows were "
                "filled with NaN); the cached bound is dropped - run the step eagerly again", __FILE__, __LINE__);
}
static int sync_all(hmg_ctx* c) {
    REQUIRE(!c->capturing, "this call synchronises the device and cannot be part of a captured step");
    for (auto& st : c->lanes) HIP_TRY(hipStreamSynchronize(st));
    c->lanes_dirty = false;
    return check_fault(c);
}

static int ensure_scratch(hmg_ctx* c, int slot, size_t bytes) {
    if (c->scratch_bytes[slot] >= bytes) return 0;
    REQUIRE(!c->capturing, "scratch must not grow inside a captured step: run the step once eagerly first");
    if (c->scratch[slot]) {
        if (sync_all(c)) return 1;
        HIP_TRY(hipFree(c->scratch[slot]));
        c->scratch[slot] = nullptr;
        c->scratch_bytes[slot] = 0;
    }
    size_t want = bytes + bytes / 8;
    HIP_TRY(hipMalloc(&c->scratch[slot], want));
    c->scratch_bytes[slot] = want;
    return 0;
}

// ------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------
namespace hmg {

// (WAVE, dpp_move, wave_sum: rowdev.hpp - shared with the long-grid kernels of longgrid.hip)

// Sum over a 1-D block (blockDim.x multiple of 64, <= 1024).  Result valid in thread 0.
__device__ __forceinline__ double block_sum(double v, double* lds /* >= 16 doubles */) {
    v = wave_sum(v);
    const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) lds[w] = v;
    __syncthreads();
    const int nw = (blockDim.x + WAVE - 1) >> 6;
    double r = 0.0;
    if (w == 0) {
        r = (lane < nw) ? lds[lane] : 0.0;
        r = wave_sum(r);
    }
    return r;
}

// Sum N per-thread values over a 1-D block with two barriers in total (instead of 2N):
// wave shuffles, one LDS exchange of N x (#waves) partials, fixed-order final sum.  The
// results are valid in threads 0..N-1 (thread i holds the total of v[i]).  lds >= N*16 doubles.
template <int N>
__device__ __forceinline__ double block_sum_multi(const double (&v)[N], double* lds) {
    const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x >> 6;
    const int nw = (blockDim.x + WAVE - 1) >> 6;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const double s = wave_sum(v[i]);
        if (lane == 0) lds[i * 16 + w] = s;
    }
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x < N)
        for (int k = 0; k < nw; ++k) r += lds[threadIdx.x * 16 + k];
    return r;
}

// ---------------------------------------------------------------- K1: sigma^2(z,m) (A2)
// sigma2[z,m] = sum_j P[z,j] * A[j,m],  A[j,m] = wq[j] W(kq[j] R[m])^2, is the one dense
// contraction of the path (nz x nm x 10^4): it runs on the fp64 matrix cores.  A wavefront
// owns a 16-mass tile, 16*ZB redshifts and one segment of the k' axis; per step of four k'
// every lane evaluates ONE window value - which is directly its element of the MFMA B operand
// (B[k = lane>>4][col = lane&15]) - loads its element(s) of P from a [k'][z] transposed,
// zero-padded copy (A[row = lane&15][k = lane>>4], 128 B coalesced per 16 lanes), and issues
// ZB v_mfma_f64_16x16x4_f64.  The window (one sincos) is therefore evaluated exactly once per
// (m, k') for up to 32 redshifts, and the nz-fold multiply-accumulate is off the vector ALU.
// The k' axis is cut into a number of segments that depends on nq only, so the summation
// order - and the result, bit for bit - is the same for a z-slab and for the full grid; the
// per-segment partial sums are combined in order by sigma2_combine_kernel.  Nothing of shape
// (nz,nm,nq) is materialised (the reference builds 1.3 GB temporaries here).
typedef double d4_t __attribute__((ext_vector_type(4)));
#ifndef HMG_SIG_SEG_LEN
#define HMG_SIG_SEG_LEN 80
#endif
constexpr int SIG_SEG_LEN = HMG_SIG_SEG_LEN;    // k' values per segment (multiple of 16)

// out[c][r] = in[r][c] for r < rows, zero for rows <= r < rows_pad
__global__ void transpose_pad_kernel(int rows, int rows_pad, int cols, const double* __restrict__ in,
                                     double* __restrict__ out /*[cols][rows_pad]*/) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)rows_pad * cols) return;
    const int c = (int)(i / rows_pad), r = (int)(i - (size_t)c * rows_pad);
    out[i] = r < rows ? in[(size_t)r * cols + c] : 0.0;
}

// (132 VGPRs, 3 waves/SIMD; forcing 4 spills and is no faster: 23.5 vs 24.0 us at Config 3)
#ifndef HMG_SIG_OCC
#define HMG_SIG_OCC 1
#endif
template <int ZB>
__device__ __forceinline__ void sigma2_mfma_block(int bx, int seg, int bz, int nz, int nzp, int nm, int nq,
                                                  const double* __restrict__ PT /*[nq][nzp]*/,
                                                  const double* __restrict__ kq,
                                                  const double* __restrict__ wq,
                                                  const double* __restrict__ R, double tswitch,
                                                  double* __restrict__ partial /*[seg][nz][nm]*/) {
    const int lane = threadIdx.x & 63, col = lane & 15, kk = lane >> 4;
    const int m = bx * 16 + col;
    const int z0 = bz * (16 * ZB);
    const double r = R[min(m, nm - 1)];
    d4_t acc[ZB];
#pragma unroll
    for (int b = 0; b < ZB; ++b) acc[b] = d4_t{0.0, 0.0, 0.0, 0.0};
    const int q_lo = seg * SIG_SEG_LEN, q_hi = min(nq, q_lo + SIG_SEG_LEN);
    constexpr int NT = SIG_SEG_LEN / 16;      // trips of four MFMA k-steps
    // Two-stage pipeline over the trips: the loads of trip t+1 (k', quadrature weight, the P rows; positions
    // past the end of the segment are clamped and given zero weight) are issued before trip t's window
    // values are evaluated, so a wavefront holds two trips of operands instead of the whole segment
    // (236 -> ~120 VGPRs: four wavefronts per SIMD instead of two, which is what feeds the VALU here).
    // The MFMA accumulation order over k' is unchanged.
    struct Trip { double kv[4], wv[4], pv[4][ZB]; };
    auto load_trip = [&](Trip& T, int t) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = q_lo + 16 * t + 4 * u + kk;
            const int qc = min(q, nq - 1);
            T.kv[u] = kq[qc];
            const double w = wq[qc];
            T.wv[u] = (q < q_hi) ? w : 0.0;
            const double* __restrict__ prow = PT + (size_t)qc * nzp + z0 + col;
#pragma unroll
            for (int b = 0; b < ZB; ++b) T.pv[u][b] = prow[16 * b];
        }
    };
    // window values (branch-free: Taylor and trigonometric forms both evaluated, selected by kR; the library
    // sincos is only called if some lane has kR >= 1e9) and the MFMA accumulation.  (Round 3: skipping the
    // trigonometric form on trips whose 64 values all lie below the Taylor switch - a quarter of the trips of a
    // default grid - was measured and is slower, 19.2 -> 20.4 us at nz = 4, 28.7 -> 29.6 at nz = 32: the branch
    // splits the four interleaved evaluations the scheduler overlaps.)
    auto consume = [&](const Trip& T) {
        double a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double kR = T.kv[u] * r;
            const double xx = kR * kR;
            const double wt = 1.0 - 0.1 * xx + 0.00357142857143 * xx * xx;
            double sn, cs;
            sincos_fast(fmin(kR, 1.0e9), sn, cs);
            if (__builtin_expect(__any(kR >= 1.0e9), 0)) {
                if (kR >= 1.0e9) sincos(kR, &sn, &cs);
            }
            const double wtr = 3.0 * (sn - kR * cs) * rcp_fast(fmax(xx * kR, 1.0e-300));
            const double w = (kR < tswitch) ? wt : wtr;
            a[u] = T.wv[u] * (w * w);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int b = 0; b < ZB; ++b)
                acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(T.pv[u][b], a[u], acc[b], 0, 0, 0);
    };
    Trip ta, tb;
    load_trip(ta, 0);
#pragma unroll
    for (int t = 0; t < NT; t += 2) {
        if (t + 1 < NT) load_trip(tb, t + 1);
        consume(ta);
        if (t + 2 < NT) load_trip(ta, t + 2);
        if (t + 1 < NT) consume(tb);
    }
    // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
    if (m < nm) {
#pragma unroll
        for (int b = 0; b < ZB; ++b)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int z = z0 + 16 * b + kk + 4 * rg;
                if (z < nz) partial[((size_t)seg * nz + z) * nm + m] = acc[b][rg];
            }
    }
}
template <int ZB>
__global__ __launch_bounds__(64, HMG_SIG_OCC) void sigma2_mfma_kernel(int nz, int nzp, int nm, int nq,
                                                         const double* __restrict__ PT, const double* __restrict__ kq,
                                                         const double* __restrict__ wq, const double* __restrict__ R,
                                                         double tswitch, double* __restrict__ partial) {
    sigma2_mfma_block<ZB>(blockIdx.x, blockIdx.y, blockIdx.z, nz, nzp, nm, nq, PT, kq, wq, R, tswitch, partial);
}

__device__ __forceinline__ double sigma2_segment_sum(int n, int parts, const double* __restrict__ partial, size_t i, int w) {
    // parts w, w+4, w+8, ... in order.  Sixteen loads are in flight per round (they do not depend on the
    // running sum); positions past the end contribute +0.0, which leaves the sum's bits alone - a scalar
    // tail loop here cost one memory round trip per leftover part (7 of them at nq = 10^4).
    double s = 0.0;
    for (int p = w; p < parts; p += 64) {
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int pp = p + 4 * u;
            v[u] = pp < parts ? partial[(size_t)pp * n + i] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) s += v[u];
    }
    return s;
}
// out[i] = sum_p partial[p][i] in a fixed order: 4 wavefronts per 64 outputs take interleaved
// segments, then add up through LDS (wave 0, in wave order).
__global__ __launch_bounds__(256) void sigma2_combine_kernel(int n, int parts,
                                                             const double* __restrict__ partial,
                                                             double* __restrict__ out) {
    __shared__ double red[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    const double s = i < n ? sigma2_segment_sum(n, parts, partial, (size_t)i, w) : 0.0;
    red[w][lane] = s;
    __syncthreads();
    if (w == 0 && i < n) out[i] = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
}

// ---------------------------------------------------------------- K2: mass function (A3/A4)
struct MassFnDev {
    int mode;
    double deltac, A, a, p, rho_m0;
    int uniform;
    double step;
};

__device__ __forceinline__ double tinker10_bias(double nu) {
    const double dc = 1.686;
    const double y = log10(200.0);
    const double ey = exp(-pow(4.0 / y, 4.0));
    const double A = 1.0 + 0.24 * y * ey;
    const double a = 0.44 * y - 0.88;
    const double C = 0.019 + 0.107 * y + 0.19 * ey;
    const double nua = pow(nu, a);
    return 1.0 - A * nua / (nua + pow(dc, a)) + 0.183 * pow(nu, 1.5) + C * pow(nu, 2.4);
}

// n(z,m) and b(z,m) of one grid point; S(i) returns sigma2[z][i] (from global memory or from LDS)
template <class SigmaAt>
__device__ __forceinline__ void massfn_point(const MassFnDev& P, int z, int m, int nm, SigmaAt S,
                                             const double* __restrict__ ms, const double* __restrict__ lnm,
                                             const double* __restrict__ tz, double& n_out, double& b_out) {
    const double sig2 = S(m);
    const double dc = P.deltac;
    double f, b;
    if (P.mode == HMG_MF_SHETH_TORMEN) {
        const double sig = sqrt(sig2);
        f = P.A * sqrt(2.0 * P.a / M_PI) * (1.0 + pow(sig2 / P.a / (dc * dc), P.p)) * (dc / sig) *
            exp(-P.a * (dc * dc) / 2.0 / sig2);
        const double t = P.a * (dc * dc) / sig2;
        b = 1.0 + (1.0 / dc) * (t - 1.0) + (2.0 * P.p / dc) / (1.0 + pow(t, P.p));
    } else {
        const double nu = dc / sqrt(sig2);
        const double al = tz[z * 5 + 0], be = tz[z * 5 + 1], ph = tz[z * 5 + 2], et = tz[z * 5 + 3],
                     ga = tz[z * 5 + 4];
        const double fnu = al * ((1.0 + pow(be * nu, -2.0 * ph)) * pow(nu, 2.0 * et) *
                                 exp(-ga * (nu * nu) / 2.0));
        f = nu * fnu;
        b = tinker10_bias(nu);
    }
    // d ln(1/sigma) / d ln m with numpy.gradient's stencils (second order interior,
    // one-sided first order at the ends; uniform-grid shortcut when numpy would take it)
    auto L = [&](int i) { return -0.5 * log(S(i)); };
    double g;
    if (nm == 1) {
        g = 0.0;
    } else if (m == 0) {
        g = (L(1) - L(0)) / (P.uniform ? P.step : (lnm[1] - lnm[0]));
    } else if (m == nm - 1) {
        g = (L(nm - 1) - L(nm - 2)) / (P.uniform ? P.step : (lnm[nm - 1] - lnm[nm - 2]));
    } else if (P.uniform) {
        g = (L(m + 1) - L(m - 1)) / (2.0 * P.step);
    } else {
        const double d1 = lnm[m] - lnm[m - 1], d2 = lnm[m + 1] - lnm[m];
        const double ca = -d2 / (d1 * (d1 + d2)), cb = (d2 - d1) / (d1 * d2), cc = d1 / (d2 * (d1 + d2));
        g = ca * L(m - 1) + cb * L(m) + cc * L(m + 1);
    }
    const double mm = ms[m];
    n_out = P.rho_m0 * f * g / (mm * mm);
    b_out = b;
}

__global__ void massfn_kernel(int nz, int nm, MassFnDev P, const double* __restrict__ s2,
                              const double* __restrict__ ms, const double* __restrict__ lnm,
                              const double* __restrict__ tz, double* __restrict__ nzm,
                              double* __restrict__ bh) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nz * nm) return;
    const int z = idx / nm, m = idx - z * nm;
    const double* row = s2 + (size_t)z * nm;
    double n, b;
    massfn_point(P, z, m, nm, [&](int i) { return row[i]; }, ms, lnm, tz, n, b);
    nzm[idx] = n;
    bh[idx] = b;
}

// Second stage of sigma^2 (the ordered sum over the k' segments, exactly sigma2_combine_kernel's) and
// the mass function in ONE launch: a workgroup owns 64 consecutive masses of one redshift, sums the
// partials of those and of the two neighbours the gradient stencil reaches, keeps the 66 values in
// LDS, writes sigma2 and evaluates n(z,m), b(z,m) from LDS.  512 threads: wavefronts 0-3 take the four
// interleaved part groups of the 64 masses, two lanes of wavefronts 4-7 those of the two neighbours (so
// that no lane walks the parts twice).
struct SigmaMassFnArgs {
    int nz, nm, parts;
    MassFnDev P;
    const double *partial /*[parts][nz*nm]*/, *ms, *lnm, *tz;
    double *s2, *nzm, *bh;
};
__device__ __forceinline__ void sigma2_massfn_block(const SigmaMassFnArgs& A, int z, int m0, double (*red)[66],
                                                    double* sig) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nm = A.nm, n = A.nz * nm;
    // slot j <-> mass m0 - 1 + j (clamped to the row)
    const int j = w < 4 ? lane + 1 : (lane == 0 ? 0 : 65);
    if (w < 4 || lane < 2) {
        const int m = min(max(m0 - 1 + j, 0), nm - 1);
        red[w & 3][j] = sigma2_segment_sum(n, A.parts, A.partial, (size_t)z * nm + m, w & 3);
    }
    __syncthreads();
    if (threadIdx.x < 66) {
        const int jj = threadIdx.x;
        const double v = ((red[0][jj] + red[1][jj]) + red[2][jj]) + red[3][jj];
        sig[jj] = v;
        const int m = m0 - 1 + jj;
        if (jj >= 1 && jj <= 64 && m < nm) A.s2[(size_t)z * nm + m] = v;
    }
    __syncthreads();
    const int m = m0 + threadIdx.x;
    if (threadIdx.x < 64 && m < nm) {
        double nn, bb;
        massfn_point(A.P, z, m, nm, [&](int i) { return sig[i - m0 + 1]; }, A.ms, A.lnm, A.tz, nn, bb);
        A.nzm[(size_t)z * nm + m] = nn;
        A.bh[(size_t)z * nm + m] = bb;
    }
}
__global__ __launch_bounds__(512) void sigma2_massfn_kernel(SigmaMassFnArgs A) {
    __shared__ double red[4][66];
    __shared__ double sig[66];
    sigma2_massfn_block(A, blockIdx.y, blockIdx.x * 64, red, sig);
}
// The same stage for a 256-thread workgroup (the form a grouped launch uses beside the NFW rows): a tile is
// 62 masses plus its two stencil neighbours = 64 slots, one per lane, so that the four wavefronts take the
// four interleaved part groups of all 64 slots and nobody walks the parts twice.  Every sigma2[z][m] is summed
// exactly as above (group g = parts g, g+4, ... in order, then ((g0 + g1) + g2) + g3), so the results are the
// same bit for bit.  red: 4 x 64 doubles of LDS, sig: 64.
constexpr int MF_TILE = 62;
__device__ __forceinline__ void sigma2_massfn_tile(const SigmaMassFnArgs& A, int z, int tile, double* red, double* sig) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nm = A.nm, n = A.nz * nm, m0 = tile * MF_TILE;
    if (w < 4) {   // slot `lane` <-> mass m0 - 1 + lane (clamped to the row)
        const int m = min(max(m0 - 1 + lane, 0), nm - 1);
        red[w * 64 + lane] = sigma2_segment_sum(n, A.parts, A.partial, (size_t)z * nm + m, w);
    }
    __syncthreads();
    if (w == 0) {
        const double v = ((red[lane] + red[64 + lane]) + red[128 + lane]) + red[192 + lane];
        sig[lane] = v;
        const int m = m0 - 1 + lane;
        if (lane >= 1 && lane <= MF_TILE && m < nm) A.s2[(size_t)z * nm + m] = v;
    }
    __syncthreads();
    const int m = m0 + (int)threadIdx.x;
    if (threadIdx.x < MF_TILE && m < nm) {
        double nn, bb;
        massfn_point(A.P, z, m, nm, [&](int i) { return sig[i - m0 + 1]; }, A.ms, A.lnm, A.tz, nn, bb);
        A.nzm[(size_t)z * nm + m] = nn;
        A.bh[(size_t)z * nm + m] = bb;
    }
}

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

// ---------------------------------------------------------------- K3: analytic NFW (A6)
// fp64-VALU bound (two Si/Ci rational evaluations + two sincos per 8 bytes written), so the
// kernel is organised to minimise instructions, not bytes: one block per (z,m) row so the
// row constants (c, r_s, 1/m_c: a log and two divisions) are computed once per thread
// instead of once per point; sin(c x) comes from the angle-difference identity on the two
// sincos the Si/Ci asymptotics need anyway; k is the fast axis -> coalesced 8 B stores.
// Small-argument series of the NFW transform, one coefficient row per (z,m):
//   u(k) = (1/m_c) int_0^c [sin(x t)/(x t)] t/(1+t)^2 dt = sum_n a_n x^(2n),
//   a_n = (-1)^n J_(2n+1)(c) / ((2n+1)! m_c),   J_p(c) = int_0^c t^p/(1+t)^2 dt,
//   J_0 = c/(1+c), J_1 = m_c,  J_p = c^(p-1)/(p-1) - 2 J_(p-1) - J_(p-2).
// With NFW_NS = 16 terms the series is exact to 2 ulp for (1+c) x <= 4 and c >= 0.5 (checked
// against 50-digit arithmetic for c in [0.5, 60]); it replaces two Si/Ci rational evaluations
// and two sincos by 16 FMAs on about 2/3 of a typical grid, and it does not suffer the
// cancellation of the closed form at small x.  a[row][0] = 0 flags "do not use" (c < 0.5).
constexpr int NFW_NS = 16;     // terms used for (1+c) x <= 4
constexpr int NFW_NS2 = 32;    // terms used for 4 < (1+c) x <= NFW_X2 (same coefficient row, first 16 shared)
constexpr double NFW_X2 = 10.0;
constexpr int NFW_ROW = HMG_NFW_SERIES_STRIDE;   // doubles per (z,m) row: 32 series coefficients + row constants
constexpr int NFW_NT1 = 5;          // terms for (1+c) x <= NFW_XS1
constexpr double NFW_XS1 = 0.1;
constexpr int NFW_NT2 = 8;          // terms for (1+c) x <= NFW_XS2
constexpr double NFW_XS2 = 0.8;
// With 32 terms the series stays within 3e-15 (absolute, against 50-digit arithmetic, c in [0.5, 100])
// up to (1+c) x = 10: the band 4 < (1+c) x <= 10 - where x itself is still on the small-argument
// branch of Si/Ci, the most expensive case of the closed form - costs 32 FMAs instead.
__device__ __forceinline__ void nfw_series_row(double c, double* __restrict__ a) {
    constexpr double INVFACT[NFW_NS2] = {1.0, 0.16666666666666666, 0.008333333333333333, 0.0001984126984126984, 2.7557319223985893e-06, 2.505210838544172e-08, 1.6059043836821613e-10, 7.647163731819816e-13, 2.8114572543455206e-15, 8.22063524662433e-18, 1.9572941063391263e-20, 3.8681701706306835e-23, 6.446950284384474e-26, 9.183689863795546e-29, 1.1309962886447718e-31, 1.2161250415535181e-34, 1.151633562077195e-37, 9.67759295863189e-41, 7.265460179153071e-44, 4.902469756513544e-47, 2.9893108271424046e-50, 1.6552108677421951e-53, 8.359650847182804e-57, 3.866628513960594e-60, 1.643974708316579e-63, 6.446959640457174e-67, 2.3392451525606576e-70, 7.876246304918039e-74, 2.4674957095607893e-77, 7.210682961895936e-81, 1.9701319568021682e-84, 5.043860616493007e-88};
    const double opc = 1.0 + c;
    const double mc = log(opc) - c / opc;
    const double inv_mc = 1.0 / mc;
    double jm2 = c / opc, jm1 = mc, cp = c;   // J_0, J_1, c^(p-1) for p = 2
    a[NFW_NS2 + 0] = log(opc);              // row constants of the closed forms, computed once per row
    a[NFW_NS2 + 1] = inv_mc;                //   here instead of once per thread of the row's workgroup
    a[NFW_NS2 + 2] = 1.0 / (opc * opc);
    a[NFW_NS2 + 3] = 0.0;
    a[0] = (c >= 0.5) ? 1.0 : 0.0;
    // unrolled: 1/(p-1) becomes a compile-time factor (a division here is ~15 dependent instructions in a
    // 62-step chain); coefficients stay within 5e-15 of 80-digit arithmetic for c in [0.5, 100]
#pragma unroll
    for (int p = 2; p < 2 * NFW_NS2; ++p) {
        const double jp = cp * (1.0 / (double)(p - 1)) - 2.0 * jm1 - jm2;
        cp *= c;
        if (p & 1) {
            const int n = (p - 1) >> 1;
            a[n] = ((n & 1) ? -jp : jp) * INVFACT[n] * inv_mc;
        }
        jm2 = jm1;
        jm1 = jp;
    }
}
__global__ void nfw_series_kernel(int rows, const double* __restrict__ cs, double* __restrict__ acoef) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    nfw_series_row(cs[row], acoef + (size_t)row * NFW_ROW);
}

// Order in which the row workgroups of a launch take the masses of a redshift: heaviest first.  Workgroups are
// dispatched in index order and the rows of the heavy end of the mass grid are the expensive ones in both tensor
// producers (NFW: most of their wavenumbers are on the Si/Ci branch; Battaglia: many FFT modes are reachable), so
// ascending order leaves the most expensive rows for the tail of the launch.  Measured on MI355X (rows group +
// profile group): 58.9 -> 55.3 us on a 4-redshift slab, 97.9 -> 96.1 at nz = 8, +-0 at nz = 32; folding the mass
// axis (light half ascending, heavy half descending) and mass-major order over all redshifts were no better.
__device__ __forceinline__ int row_order(int r, int nm) {
    const int z = r / nm, i = r - z * nm;
    return z * nm + (nm - 1 - i);
}

// ktile = k values per workgroup (a multiple of the block size)
// 46 VGPRs, no scratch (with machine LICM off: see the Makefile); the bound only keeps it under 64.
#ifndef HMG_NFW_OCC
#define HMG_NFW_OCC 8
#endif
struct NfwArgs {
    const SiciTable* T;
    const double* acoef;
    int ktile, nm, nk;
    const double *cs, *rss, *zs, *ks;
    double* uk;
};
// blk: index of the (row, k tile) this workgroup owns; nthr: threads that share it (the workgroup size).
// The pointers must reach this function as __restrict__ KERNEL PARAMETERS (not as fields of a by-value
// struct): only then can hipcc prove that the stores to uk do not clobber the row constants, series
// coefficients and Si/Ci tables and fetch those with scalar loads - as struct fields they became 249 vector
// loads per thread and the kernel ran 2.6x slower (0.15 -> 0.40 ms at Config 3).
__device__ __forceinline__ void nfw_rows(const SiciTable* __restrict__ T, const double* __restrict__ acoef, int ktile,
                                         int nm, int nk, const double* __restrict__ cs,
                                         const double* __restrict__ rss, const double* __restrict__ zs,
                                         const double* __restrict__ ks, double* __restrict__ uk, int blk, int nthr,
                                         int tid) {
    // one (z,m) row per workgroup, the whole k axis in one tile: measured against two rows per workgroup
    // (+4 %), half tiles (+40 %) and 128 threads per row (+-0): a workgroup's fixed cost is the latency of
    // its scalar loads (row constants, series coefficients), not instructions
    const int ktiles = (nk + ktile - 1) / ktile;
    const int brow = blk / ktiles;
    const int row = row_order(brow, nm);  // z*nm + m
    const int k_lo = (blk - brow * ktiles) * ktile;
    const int k_hi = min(nk, k_lo + ktile);
    const int z = row / nm;
    const double c = cs[row];
    const double rs = rss[row];
    const double z1 = 1.0 + zs[z];
    const double opc = 1.0 + c;
    // small-argument series coefficients and closed-form constants of this row: wave-uniform -> SGPRs
    const double* __restrict__ a = acoef + (size_t)row * NFW_ROW;
    const double ln_opc = a[NFW_NS2 + 0], inv_mc = a[NFW_NS2 + 1], inv_opc2 = a[NFW_NS2 + 2];
    const bool use_series = (a[0] != 0.0);
    double* __restrict__ dst = uk + (size_t)row * nk;
    for (int k = k_lo + tid; k < k_hi; k += nthr) {
        const double x = ks[k] * rs * z1;
        const double xc = opc * x;
        if (use_series && xc <= 4.0) {
            // The series alternates and its n-th term is below (xc)^(2n) / (2n (2n+1)! m_c): 5 terms are
            // exact to 1e-18 for (1+c) x <= 0.1, 8 terms to 2e-17 for <= 0.8 - about 60 % of a typical
            // grid (k starts four decades below the halo scale) takes one of the two short forms.
            const double z = x * x;
            double u;
            if (xc <= NFW_XS1) {
                u = fma_svs(a[NFW_NT1 - 1], z, a[NFW_NT1 - 2]);
#pragma unroll
                for (int n = NFW_NT1 - 3; n >= 0; --n) u = fma_vvs(u, z, a[n]);
            } else if (xc <= NFW_XS2) {
                u = fma_svs(a[NFW_NT2 - 1], z, a[NFW_NT2 - 2]);
#pragma unroll
                for (int n = NFW_NT2 - 3; n >= 0; --n) u = fma_vvs(u, z, a[n]);
            } else {
                u = fma_svs(a[NFW_NS - 1], z, a[NFW_NS - 2]);
#pragma unroll
                for (int n = NFW_NS - 3; n >= 0; --n) u = fma_vvs(u, z, a[n]);
            }
            __builtin_nontemporal_store(u, &dst[k]);
            continue;
        }
        if (use_series && xc <= NFW_X2) {
            const double z = x * x;
            double u = fma_svs(a[NFW_NS2 - 1], z, a[NFW_NS2 - 2]);
#pragma unroll
            for (int n = NFW_NS2 - 3; n >= 0; --n) u = fma_vvs(u, z, a[n]);
            __builtin_nontemporal_store(u, &dst[k]);
            continue;
        }
        if (x > 4.0 && xc < 1.0e9) {
            // Both arguments on the auxiliary-function branch, Si = pi/2 - f cos - g sin,
            // Ci = f sin - g cos.  Substituting into the NFW formula the terms in f(x) cancel and
            // the rest collapses, exactly, to
            //     u m_c = g(x) + f(xc) sin(c x) - g(xc) cos(c x) - sin(c x)/xc :
            // one sincos (of c x, the argument the reference itself uses for sin(c x)) instead of
            // two, three rationals instead of four, and none of the pi/2-sized cancellations.
            const double zx = rcp_fast(x * x), zc = zx * inv_opc2;
            double f1, g1, f2, g2, sd, cd;
            sici_aux<false>(T, x, zx, f1, g1);
            sici_aux<true>(T, xc, zc, f2, g2);
            sincos_fast(c * x, sd, cd);
            __builtin_nontemporal_store((g1 + (f2 - xc * zc) * sd - g2 * cd) * inv_mc, &dst[k]);
            continue;
        }
        if (x <= 4.0 && xc > 8.0) {
            // Mixed band (3.5 % of a typical grid, beyond the reach of the series): x on the rational
            // branch of Si/Ci, (1+c)x on the auxiliary-function branch.  Substituting
            // Si(xc) = pi/2 - f cos xc - g sin xc, Ci(xc) = f sin xc - g cos xc and xc - x = c x,
            //     u m_c = (pi/2) sin x + f(xc) sin(cx) - g(xc) cos(cx) - sin(cx)/xc - sin x Si(x) - cos x Ci(x):
            // sincos of x and c x (the reference's own arguments) instead of x and xc, two rational
            // pairs instead of four, one short logarithm.
            double s1, c1, sd, cd, f2, g2;
            sincos_fast(x, s1, c1);
            sincos_fast(c * x, sd, cd);
            const double x2 = x * x;
            const double zc = rcp_fast(x2) * inv_opc2;                   // 1/xc^2
            const double sden = horner_s<6>(x2, T->SD), cden = horner_s<6>(x2, T->CD);
            const double r = rcp_fast(sden * cden);
            const double si = x * horner_s<6>(x2, T->SN) * (cden * r);
            const double ci = (EULER_GAMMA + log_fast(x)) + x2 * horner_s<6>(x2, T->CN) * (sden * r);
            sici_aux<true>(T, xc, zc, f2, g2);
            __builtin_nontemporal_store((HALF_PI * s1 + (f2 - xc * zc) * sd - g2 * cd - s1 * si - c1 * ci) * inv_mc,
                                        &dst[k]);
            continue;
        }
        // everything else (rows with c < 0.5, arguments beyond 1e9): the closed form as the reference writes it
        double s1, c1, s2, c2;
        if (xc < 1.0e9) {
            sincos_fast(x, s1, c1);
            sincos_fast(xc, s2, c2);
        } else {  // outside the Cody-Waite range: library reduction
            sincos(x, &s1, &c1);
            sincos(xc, &s2, &c2);
        }
        const double zx = rcp_fast(x * x);   // 1/x^2
        const double zc = zx * inv_opc2;     // 1/xc^2
        double si1, ci1, si2, ci2;
        bool sm1, sm2;
        sici_fast(T, x, s1, c1, zx, si1, ci1, sm1);
        sici_fast(T, xc, s2, c2, zc, si2, ci2, sm2);
        // Ci((1+c)x) - Ci(x): x <= xc, so the cases are (small,small), (small,large), (large,large)
        double dci = ci2 - ci1;
        if (sm1) dci += sm2 ? ln_opc : -(EULER_GAMMA + log(x));
        const double scx = s2 * c1 - c2 * s1;  // sin(c x) = sin((1+c)x - x)
        // sin(cx)/((1+c)x) = scx * xc / xc^2
        __builtin_nontemporal_store((s1 * (si2 - si1) - scx * (xc * zc) + c1 * dci) * inv_mc, &dst[k]);
    }
}
__global__ __launch_bounds__(256, HMG_NFW_OCC) void nfw_kernel(const SiciTable* __restrict__ T,
                                                  const double* __restrict__ acoef, int ktile, int nm, int nk,
                                                  const double* __restrict__ cs, const double* __restrict__ rss,
                                                  const double* __restrict__ zs, const double* __restrict__ ks,
                                                  double* __restrict__ uk) {
    nfw_rows(T, acoef, ktile, nm, nk, cs, rss, zs, ks, uk, blockIdx.x, blockDim.x, threadIdx.x);
}

// ---------------------------------------------------------------- A8/X1: row parameters
struct RowFit { double f[9]; };
struct RowOut {
    double *amp, *xc, *alpha, *expo, *cmax, *rscale, *post;
    // optional row scalars of the profile transform that will read these rows (hmg_rows_part, ABI 8)
    double* rowsc;
    const double *ks, *kts;
    int nk, M;
};
// The output-side scalars of one profile row (hmvec/fft.py:96-107), the SAME expressions profile_fused_row evaluates
// when it has to work them out itself: isc = 1/(r (1+z)) (kout_j = kt_j isc), k_lo = kt_1 isc, k_hi = kt_M isc, 1/k_lo,
// 1/kt_1, jn = modes the target grid can reach, nleft = targets below k_lo (ks ascending: a bisection here, a 64-way
// search there - the same count).
__device__ __forceinline__ void rowscal_store(const RowOut& O, int idx, double rscale, double z1) {
    if (!O.rowsc) return;
    const double isc0 = 1.0 / (rscale * z1);
    const double kt1 = O.kts[1];
    const double klo0 = kt1 * isc0;
    const double idk0 = 1.0 / klo0;
    int jn0 = O.M;
    const double tmax = O.ks[O.nk - 1] * idk0;
    if (tmax < (double)(O.M - 4)) jn0 = (int)tmax + 3;
    int lo = 0, hi = O.nk;                     // ks[i] < k_lo for i < lo, ks[i] >= k_lo for i >= hi
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (O.ks[mid] < klo0) lo = mid + 1; else hi = mid;
    }
    double* __restrict__ r = O.rowsc + (size_t)idx * HMG_ROWSC_STRIDE;
    r[0] = isc0; r[1] = klo0; r[2] = O.kts[O.M] * isc0; r[3] = idk0; r[4] = 1.0 / kt1;
    r[5] = __hiloint2double(jn0, lo);
    r[6] = 0.0; r[7] = 0.0;
}
__device__ __forceinline__ void rowparams_body(int kind, int idx, double M, double R, double rv, double z1,
                                               double rhoc, double hz, const RowFit& F, double gamma,
                                               double alpha_const, double pref, double post_pref,
                                               const RowOut& O) {
    // A0 (M/1e14)^am (1+z)^az for the three fits: the two logarithms are shared and each power
    // product is one exp2 (|exponent| < 10, so the result is within a few ulp of pow*pow)
    const double lm = log2(M / 1.0e14), lz = log2(z1);
    const double X0 = F.f[0] * exp2(F.f[1] * lm + F.f[2] * lz);
    const double X1 = F.f[3] * exp2(F.f[4] * lm + F.f[5] * lz);
    const double X2 = F.f[6] * exp2(F.f[7] * lm + F.f[8] * lz);
    if (kind == HMG_PROF_BATTAGLIA_GAS) {
        // (Ob/Om) rho_c rho0 x^g (1+x^alpha)^(-(beta+g)/alpha),  x = r/(R200c/2)
        O.amp[idx] = pref * rhoc * X0;
        O.xc[idx] = 1.0;
        O.alpha[idx] = X1;
        O.expo[idx] = (X2 + gamma) / X1;
        const double rg = R / 2.0;
        O.rscale[idx] = rg;
        O.cmax[idx] = rv / rg;
        if (O.post) O.post[idx] = 1.0;
        rowscal_store(O, idx, rg, z1);
    } else {
        // eFrac (Ob/Om) 200 M G rho_c / (2 R200) P0 (x/xc)^g (1+(x/xc)^alpha)^(-beta),  x = r/R200c
        O.amp[idx] = pref * M * rhoc / (2.0 * R) * X0;
        O.xc[idx] = X1;
        O.alpha[idx] = alpha_const;
        O.expo[idx] = X2;
        O.rscale[idx] = R;
        O.cmax[idx] = rv / R;
        if (O.post) O.post[idx] = post_pref * ((R * R * R) * ((z1 * z1) / hz));
        rowscal_store(O, idx, R, z1);
    }
}

__global__ void rowparams_kernel(int kind, int nz, int nm, const double* __restrict__ m200,
                                 const double* __restrict__ r200, const double* __restrict__ rvir,
                                 const double* __restrict__ zs, const double* __restrict__ rhoc,
                                 const double* __restrict__ hz, RowFit F, double gamma,
                                 double alpha_const, double pref, double post_pref, RowOut O) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nz * nm) return;
    const int z = idx / nm;
    rowparams_body(kind, idx, m200[idx], r200[idx], rvir[idx], 1.0 + zs[z], rhoc[z], hz ? hz[z] : 1.0, F,
                   gamma, alpha_const, pref, post_pref, O);
}

// mass conversion + row parameters in one launch (one kernel boundary fewer per profile)
__global__ void rows_from_mvir_kernel(int kind, int nz, int nm, const double* __restrict__ ms,
                                      const double* __restrict__ cs, const double* __restrict__ rvir,
                                      const double* __restrict__ zs, const double* __restrict__ d1,
                                      double delta2, const double* __restrict__ rhoc,
                                      const double* __restrict__ hz, RowFit F, double gamma,
                                      double alpha_const, double pref, double post_pref,
                                      double* __restrict__ m2, double* __restrict__ r2, RowOut O) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nz * nm) return;
    const int z = idx / nm, m = idx - z * nm;
    const double M2 = mdelta_solve(ms[m], cs[idx], d1[z] / (delta2 * rhoc[z]));
    const double R2 = cbrt(3.0 * M2 / 4.0 / M_PI / delta2 / rhoc[z]);
    m2[idx] = M2;
    r2[idx] = R2;
    rowparams_body(kind, idx, M2, R2, rvir[idx], 1.0 + zs[z], rhoc[z], hz ? hz[z] : 1.0, F, gamma,
                   alpha_const, pref, post_pref, O);
}

// c, rvir, rs + the NFW series row + the mass conversion of one (z,m) per thread: the three
// per-(z,m) launches that precede the profile kernels of a pass, in one (hmg_halo_stage).
__device__ __forceinline__ void nfw_series_row(double c, double* __restrict__ a);
struct HaloStageArgs {
    int nz, nm;
    const double *ms, *zs, *delta, *rho;
    double A, alpha, beta, h;
    double *cs, *rv, *rs, *series /*[nz*nm][NFW_ROW] or null*/;
    const double* d1;
    double delta2;
    const double* rho2;
    double *m2, *r2 /* both or neither */;
};
// (rv_out, m2_out, r2_out: the values just stored, for a caller that goes on to the Battaglia row parameters)
__device__ __forceinline__ void halo_stage_point(const HaloStageArgs& H, int idx, double* rv_out = nullptr,
                                                 double* m2_out = nullptr, double* r2_out = nullptr) {
    const int z = idx / H.nm, m = idx - z * H.nm;
    const double mm = H.ms[m];
    const double c = H.A * pow(H.h * mm / 2.0e12, H.alpha) * pow(1.0 + H.zs[z], H.beta);
    const double r = pow(3.0 * mm / 4.0 / M_PI / H.delta[z] / H.rho[z], 1.0 / 3.0);
    H.cs[idx] = c;
    H.rv[idx] = r;
    H.rs[idx] = r / c;
    if (rv_out) *rv_out = r;
    if (H.m2) {
        const double M2 = mdelta_solve(mm, c, H.d1[z] / (H.delta2 * H.rho2[z]));
        const double R2 = cbrt(3.0 * M2 / 4.0 / M_PI / H.delta2 / H.rho2[z]);
        H.m2[idx] = M2;
        H.r2[idx] = R2;
        if (m2_out) { *m2_out = M2; *r2_out = R2; }
    }
    if (H.series) nfw_series_row(c, H.series + (size_t)idx * NFW_ROW);
}
__global__ void halo_stage_kernel(HaloStageArgs H) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < H.nz * H.nm) halo_stage_point(H, idx);
}

// Everything the constructor computes per (z,m), in ONE launch behind the sigma^2 contraction: plane 0 of
// the grid is sigma2_massfn_kernel's work (needs the contraction's partial sums), plane 1 the halo stage
// (needs only m and z).  The two do not depend on each other, so the halo stage's workgroups fill the
// compute units the 8 x nz mass-function workgroups leave idle instead of waiting for a launch of their own.
// grid (ceil(nm/64), nz, 2), 512 threads; plane 1 uses the first wavefront of each workgroup.
__global__ __launch_bounds__(512) void ctor_stage_kernel(SigmaMassFnArgs A, HaloStageArgs H) {
    __shared__ double red[4][66];
    __shared__ double sig[66];
    if (blockIdx.z == 0) {
        sigma2_massfn_block(A, blockIdx.y, blockIdx.x * 64, red, sig);
    } else if (threadIdx.x < 64) {
        const int m = blockIdx.x * 64 + threadIdx.x;
        if (m < H.nm) halo_stage_point(H, blockIdx.y * H.nm + m);
    }
}

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

// ---------------------------------------------------------------- K45: fused radial-profile transform
// One workgroup per (z,m) row does the whole of generic_profile_fft (hmvec/fft.py:56-115)
// without touching HBM in between: integrand + mass norm -> packed-real FFT in LDS
// (ldsfft.hpp) -> Im F_j -> u_j -> linear interpolation onto the target k grid.  The only
// HBM traffic is the (nk) output row plus per-row scalars; the rocFFT path it replaces moves
// 2*8*nxs + 2*16*(nxs/2+1) bytes per row through the memory system (3.2 GB at Config 3).
// Used when nxs is even, nxs/2 factors into 5/4/3/2 and fits LDS; otherwise hmg_profile_fft
// falls back to the chunked rocFFT path.
// (UnpackTw - the per-mode constants of the unpack step, one 32-byte load - lives in ldsfft.hpp)
// (FusedArgs - the description of a launch of radial-profile rows - lives in rowdev.hpp)

// ln x_n of the radial grid: the same for all (z,m) rows, so with many rows one small launch replaces a
// quarter of the integrand's transcendentals (same log_fast as the in-kernel path: identical bits).
__global__ void logx_kernel(int n, const double* __restrict__ xs, double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = log_fast(xs[i]);
}

// (gnfw_rho_fast: rowdev.hpp)

template <int NT, int R, int MAXB, bool SMALL = false, int NIN = R, int SRC_SHIFT = 0>
__device__ __forceinline__ void fused_pass(cplx* buf, const cplx* __restrict__ twM, int M, int Ns, int twstep,
                                           unsigned magic, int keep) {
    // keep >= 0 (last pass only, Ns == M/R): butterfly j writes Z[j + t*Ns]; only Z[0..keep] and
    // Z[M-keep..M-1] will be read, i.e. butterflies j <= keep (t = 0) and j >= Ns - keep (t = R-1).
    cplx v[MAXB][R];
    const int nb = M / R;
#pragma unroll
    for (int b = 0; b < MAXB; ++b) {
        const int j = threadIdx.x + b * NT;
        if (j < nb && (keep < 0 || j <= keep || j >= nb - keep)) pass_load<R, SMALL, NIN, SRC_SHIFT>(buf, twM, M, Ns, twstep, magic, j, v[b]);
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < MAXB; ++b) {
        const int j = threadIdx.x + b * NT;
        if (j < nb && (keep < 0 || j <= keep || j >= nb - keep)) pass_store<R, SMALL, NIN>(buf, Ns, magic, j, v[b]);
    }
    __syncthreads();
}

// The passes of a length-M transform whose plan the compiler knows (ldsfft.hpp: SubPass<M, PS>): strides, twiddle
// steps and index multipliers are immediates, every pass gets the butterflies-per-thread count it needs, and the
// pass loop with its radix dispatch is gone - what the run-time plan pays in registers (116 B/lane of scratch in
// round 3's profile_group_kernel<*,*,0>) and scalar instructions.
template <int NT, int M, int PS>
__device__ __forceinline__ void fused_passes_ct(cplx* buf, const cplx* __restrict__ twM, bool pruned, int jn) {
    if constexpr (PS < SubPass<M, 0>::P.npass) {
        using S = SubPass<M, PS>;
        constexpr int MAXB = (S::nb + NT - 1) / NT;
        // the last pass only has to produce Z[0..jn] and Z[M-jn..M-1]
        const int keep = (S::last && 2 * jn + 2 < S::nb) ? jn : -1;
        if (!(PS == 0 && pruned)) fused_pass<NT, S::R, MAXB, S::SMALL>(buf, twM + S::twoff, M, S::Ns, 1, S::mg, keep);
        fused_passes_ct<NT, M, PS + 1>(buf, twM, pruned, jn);
    }
}
template <int SPECM> constexpr int fused_first_radix() {
    if constexpr (SPECM == 0) return 0; else return SubPass<SPECM, 0>::R;
}

#ifndef HMG_FUSED_OCC
#define HMG_FUSED_OCC 8
#endif
// waves per SIMD a fused-row launch is compiled for: 8 with a compile-time plan (<= 64 VGPRs, no spill).  The
// run-time plan needs ~91 registers for its pass loop and dispatch chain; measured on the Config-3 grid at
// nxs = 3000 / 2000 (tools/shape_sweep.py): 8 waves/SIMD (64 VGPRs, 28-34 spilled, 116 B/lane) 0.294 / 0.277 ms,
// 6 (80 VGPRs, 12 spilled, 52 B/lane) 0.263 / 0.243 ms, 5 (91 VGPRs, nothing spilled) 0.320 / 0.285 ms: 6 it is.
// The lengths people use have compile-time plans (nxs = 1000, 2000, 4000, 5000: 0.129, 0.155, 0.193 ms for the
// first three against 0.190, 0.238, 0.280 with this run-time plan), so this path serves the odd ones.
#ifndef HMG_RT_OCC
#define HMG_RT_OCC 6
#endif
template <int MAXB, int SPECM> constexpr int fused_occ() { return MAXB > 2 ? 4 : (SPECM ? HMG_FUSED_OCC : HMG_RT_OCC); }
// SPECM != 0: the plan is known at compile time (SPECM = 2500, passes 4,5,5,5,5: nxs = 5000, the default
// length of the Battaglia profiles) - strides, twiddle steps and the j/Ns multipliers become immediates and
// the pass loop with its dispatch chain unrolls.
// ABL: phase-ablation policy for timing experiments (tools/abl_run.sh, tools/abl_pmc.sh build the library with
// -DHMG_ABL=N): 0 = the product (the only value a normal build instantiates); 5 workgroup launch only, 6 + row
// scalars, 4 phase A without its transcendentals, 1 stop after phase A, 2 after phase B, 3 after phase C.
#ifndef HMG_ABL
#define HMG_ABL 0
#endif
// TAB: the profile is read from a table (a user's callable evaluated on the x grid: hmvec/fft.py:56-94) instead of
// evaluated from the family; everything behind the integrand is the same code.
// In-kernel time stamps (diagnostic builds only: -DHMG_FR_STAMP; tools/probes/fused_stamps.py)
#ifdef HMG_FR_STAMP
__device__ long long g_fstamps[4096 * 16];
#define FSTAMP(k) do { if (fslot >= 0 && threadIdx.x == 0) g_fstamps[fslot * 16 + (k)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define FSTAMP(k) do { } while (0)
#endif
template <int NT, int MAXB, int MAXP, int SPECM, int ABL = HMG_ABL, bool TAB = false>
__device__ __forceinline__ void profile_fused_row(const FusedArgs& A, int row, double* smem) {
#ifdef HMG_FR_STAMP
    const int fslot = (blockIdx.x % 29 == 0 && blockIdx.x / 29 < 4096) ? (int)(blockIdx.x / 29) : -1;
    if (fslot >= 0 && threadIdx.x < 16) g_fstamps[fslot * 16 + threadIdx.x] = 0;
    if (fslot >= 0 && threadIdx.x == 0) g_fstamps[fslot * 16 + 14] = (long long)wall_clock64();
#endif
    FSTAMP(0);
    // dynamic LDS only (base stays 16 B aligned for the 128-bit complex accesses):
    // [0, 2M) doubles = packed row as cplx, later u[0..M-1]; then 16 doubles of reduction
    // scratch, the broadcast mass norm and the left-fill counter.
    cplx* buf = reinterpret_cast<cplx*>(smem);
    const int M = SPECM ? SPECM : A.plan.M, nxs = SPECM ? 2 * SPECM : A.nxs;
    double* red = smem + 2 * (size_t)M;
    int* s_cnt = reinterpret_cast<int*>(red + 17);
    if constexpr (ABL == 5) {     // timing experiment: workgroup launch only
        if (threadIdx.x == 0) A.out[(size_t)row * A.nk] = 1.0;
        return;
    }
    const double Aamp = A.amp ? A.amp[row] : A.amp_c;
    const double XC = A.xc ? A.xc[row] : A.xc_c;
    const double AL = A.alpha ? A.alpha[row] : A.alpha_c;
    const double EX = A.expo ? A.expo[row] : A.expo_c;
    const double cm = A.cmax[row];
    // ln(x/xc) = ln x - ln xc: ln x is row-independent (xc == 1 for the gas and NFW members: no logarithm)
    const double ln_xc = (TAB || (A.xc == nullptr && A.xc_c == 1.0)) ? 0.0 : log_fast(XC);
    const double* __restrict__ tab = TAB ? A.rho_tab + (A.rho_shared ? (size_t)0 : (size_t)row * (size_t)(SPECM ? 2 * SPECM : A.nxs)) : nullptr;
    // Output side of the row: the FFT modes sit on the uniform grid kout_j = j k_lo,
    // k_lo = kt_1 / (r_s (1+z)).  Targets below k_lo take np.interp's left fill u_1, targets above
    // kout_M are zero, and only the modes j <= jn = floor(max(ks)/k_lo) + 2 can be reached at all:
    // low-mass rows (large k_lo) need a few dozen of the M modes, so the unpack and the last FFT
    // pass are cut down to those.  max(ks) is only known without a search when ks is ascending,
    // which is the caller's promise that comes with the hint arrays (include/hmgrid.h).
    // These row scalars are the same for all 512 threads and cost a few divisions: one wavefront works them out
    // while the others start on the integrand, and they travel through LDS behind the barrier that is there
    // anyway (red[17..23]: length of the left-fill prefix, jn, 1/(r_s(1+z)), k_lo, k_hi, 1/k_lo, 1/kt_1).
    const int z = row / A.nm;
    int* s_jn = reinterpret_cast<int*>(red + 18);
    // The LAST wavefront works them out (in the truncated Battaglia rows it holds no non-zero sample, so it is the
    // one with nothing to do in phase A); every lane computes the same values and lane 0 stores them.  With the
    // hint arrays (ks ascending) the same wavefront also locates the end of the left-fill prefix - the first target
    // wavenumber that is not below k_lo - by a 64-way search: each lane tests the last wavenumber of its segment,
    // the number of lanes that see it below k_lo is the number of segments that lie in the prefix entirely, and the
    // next segment holds the boundary (two dependent loads for nk <= 4096).  Phase D then fills [0, nleft)
    // without loading or testing a wavenumber.
    // (with A.rowsc - the grouped passes of the facade - the launch that computed the rows' length scales left these
    // numbers per row: they arrive by scalar loads and no wavefront of this workgroup divides or searches)
    const double* __restrict__ rsc = A.rowsc ? A.rowsc + (size_t)row * HMG_ROWSC_STRIDE : nullptr;
    if (!rsc && threadIdx.x >= NT - 64) {
        const int lane = threadIdx.x & 63;
        const double isc0 = 1.0 / (A.rss[row] * (1.0 + A.zs[z]));      // kout_j = kts[j] * isc
        const double klo0 = A.kts[1] * isc0;
        const double idk0 = 1.0 / klo0;
        int jn0 = M, nleft = 0;
        if (A.nconst) {
            const double tmax = A.ks[A.nk - 1] * idk0;
            if (tmax < (double)(M - 4)) jn0 = (int)tmax + 3;           // one spare mode for the rounding of tmax
            int base = 0, end = A.nk;
            for (;;) {
                const int stp = (end - base + 63) >> 6;
                const int first = base + lane * stp;
                bool below = false;
                if (first < end) {
                    const int last = first + stp - 1;
                    below = A.ks[last < end ? last : end - 1] < klo0;
                }
                base += __popcll(__ballot(below)) * stp;
                if (base >= end) { base = end; break; }
                if (stp == 1) break;
                end = base + stp < end ? base + stp : end;
            }
            nleft = base;
        }
        if (lane == 0) {
            *s_cnt = nleft;
            *s_jn = jn0;
            red[19] = isc0; red[20] = klo0; red[21] = A.kts[M] * isc0; red[22] = idk0;
            red[23] = 1.0 / A.kts[1];
        }
    }
    if constexpr (ABL == 6) {     // timing experiment: launch + row scalars, no integrand
        if (threadIdx.x == 0) A.out[(size_t)row * A.nk] = Aamp + XC + AL + EX + cm + ln_xc;
        return;
    }
    // ---- phase A: y_n = x_n rho(x_n) theta(x_n <= cmax) packed as (y_2p, y_2p+1); mass norm
    // Pruned first pass: the integrand is zero beyond the truncation radius (85 % of a Battaglia
    // row at xmax = 20).  When every packed sample p >= M/R0 is zero, the first radix-R0 pass
    // sees (v0, 0, ..., 0) in every butterfly, whose DFT is v0 in all R0 outputs - exactly, in
    // floating point - so phase A writes each sample straight into its R0 output slots and the
    // pass (an LDS round trip, two barriers, the zero fill of the rest of the row) is skipped.
    const int R0 = SPECM ? fused_first_radix<SPECM>() : A.plan.radix[0];
    const int stride0 = M / R0;
    const bool pruned = (SPECM || A.plan.npass > 1) && A.xs[2 * stride0] > cm;   // xs is increasing
    // compile-time plan: when the row is zero from sample 375 on, the pass behind the pruned one reads samples
    // 0..374 only (3-of-5 butterflies, below) and the rest of the row need not even be cleared
    const bool lead3 = SPECM == 2500 && pruned && A.xs[2 * 375] > cm;
    const int pend = lead3 ? 375 : (pruned ? stride0 : M);
#ifdef HMG_FR_STAMP
    if (fslot >= 0 && threadIdx.x == 0) g_fstamps[fslot * 16 + 8] = (long long)__builtin_readcyclecounter() + (pend == 12345 ? 1 : 0);
#endif
    double acc = 0.0;
    for (int p = threadIdx.x; p < pend; p += NT) {
        const int j = 2 * p;
        const double2 xv = *reinterpret_cast<const double2*>(A.xs + j);
        double r0 = 0.0, r1 = 0.0;
        if constexpr (ABL == 4) {     // timing experiment: phase A without its transcendentals
            if (!(fabs(xv.x) > cm)) r0 = Aamp * xv.x + AL;
            if (!(fabs(xv.y) > cm)) r1 = Aamp * xv.y + EX;
        } else if constexpr (TAB) {
            if (!(fabs(xv.x) > cm)) r0 = tab[j];
            if (!(fabs(xv.y) > cm)) r1 = tab[j + 1];
        } else {
            if (!(fabs(xv.x) > cm)) r0 = gnfw_rho_fast((A.logx ? A.logx[j] : log_fast(xv.x)) - ln_xc, Aamp, AL, EX, A.gamma);
            if (!(fabs(xv.y) > cm)) r1 = gnfw_rho_fast((A.logx ? A.logx[j + 1] : log_fast(xv.y)) - ln_xc, Aamp, AL, EX, A.gamma);
        }
        const cplx y = cplx{xv.x * r0, xv.y * r1};
        if (pruned && SPECM != 2500) {
            // the R0 copies go out in an order rotated by lane/4: with t the same in every lane, lanes l and l+4
            // (64 B apart) hit the same LDS banks and every one of these 16-B stores takes two passes
            const int rot = (threadIdx.x >> 2);
            for (int t = 0; t < R0; ++t) buf[R0 * p + (t + rot) % R0] = y;
        } else {
            // (the hand-sequenced 2500 plan replicates nothing: its second pass reads slot i as sample i >> 2)
            buf[p] = y;
        }
        if (A.do_norm && (r0 != 0.0 || r1 != 0.0)) {
            const double xl = (j > 0) ? A.xs[j - 1] : xv.x, xr = (j + 2 < nxs) ? A.xs[j + 2] : xv.y;
            acc += 0.5 * (xv.y - xl) * (r0 * (xv.x * xv.x)) + 0.5 * (xr - xv.x) * (r1 * (xv.y * xv.y));
        }
    }
    // mass norm: wavefront sums (DPP), one LDS exchange, and EVERY thread adds the eight partials itself in
    // wave order - no second reduction stage.  The barrier also publishes buf and the row scalars the last
    // wavefront wrote (red[0..7] are written nowhere else, so nothing has to be waited for before).
    {
#ifdef HMG_FR_STAMP
        if (fslot >= 0 && threadIdx.x == 0) g_fstamps[fslot * 16 + 9] = (long long)__builtin_readcyclecounter() + (acc == 1.2345e300 ? 1 : 0);
#endif
        const double ws = wave_sum(acc);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ws;
        __syncthreads();
#ifdef HMG_FR_STAMP
        if (fslot >= 0 && threadIdx.x == 0) g_fstamps[fslot * 16 + 10] = (long long)__builtin_readcyclecounter();
#endif
    }
    // The first wavefront adds the eight partials in wave order and forms the one number the rest of the row needs
    // from the norm: the scale of the unpack step, u_j = Im F_j * (-step / (mnorm kt_1)) / j.  It travels through
    // red[24] behind the barriers of the FFT passes (a division and seven additions that 448 threads used to repeat).
    if (threadIdx.x < 64) {
        double tot = red[0];
#pragma unroll
        for (int w = 1; w < NT / 64; ++w) tot += red[w];
        const double mnorm = A.do_norm ? tot : 1.0;
        if (threadIdx.x == 0) red[24] = -A.step / mnorm * (rsc ? rsc[4] : red[23]);
    }
    const int jn = rsc ? __double2hiint(rsc[5]) : __builtin_amdgcn_readfirstlane(*s_jn);
    FSTAMP(1);
#ifdef HMG_FR_STAMP
    if (fslot >= 0 && threadIdx.x == 0) g_fstamps[fslot * 16 + 13] = jn;
#endif
    if constexpr (ABL == 1 || ABL == 4) {     // timing experiments only: stop after phase A
        if (threadIdx.x < 8) A.out[(size_t)row * A.nk + threadIdx.x] = buf[threadIdx.x].x + red[0];
        return;
    }
    // ---- phase B: in-place Stockham FFT of length M
    // (Tried and dropped, MI355X: fetching all R operands before the twiddle products and requesting the
    // next pass's twiddle between the two halves of a pass.  Both lengthen live ranges under the 64-VGPR
    // cap of 8 waves/SIMD: 0.277 -> 0.315 ms.)
    if constexpr (SPECM == 2500) {
        // butterfly indices stay below 1024: the 24-bit index arithmetic of ldsfft.hpp (div_ns)
        constexpr unsigned mg4 = small_magic(4), mg20 = small_magic(20), mg100 = small_magic(100), mg500 = small_magic(500);
        // (A.twM is the per-pass twiddle table, ldsfft.hpp: the slices of the passes start at 0, 1, 5, 25, 125 and a
        // butterfly reads element k = j mod Ns of its pass's slice - consecutive lanes, consecutive elements)
        if (!pruned) fused_pass<NT, 4, MAXB, true>(buf, A.twM, 2500, 1, 1, 0u, -1);
        // behind the pruned first pass slot i of the row holds sample i/4: a butterfly of this pass reads slots
        // j + 500 t, and those with t >= 3 are zero when the row is zero from sample 1500/4 on (cmax < 3 at xmax = 20)
        if (pruned) {
            if (lead3) fused_pass<NT, 5, 1, true, 3, 2>(buf, A.twM + 1, 2500, 4, 1, mg4, -1);
            else fused_pass<NT, 5, 1, true, 5, 2>(buf, A.twM + 1, 2500, 4, 1, mg4, -1);
        } else fused_pass<NT, 5, 1, true>(buf, A.twM + 1, 2500, 4, 1, mg4, -1);
        FSTAMP(2);
        fused_pass<NT, 5, 1, true>(buf, A.twM + 5, 2500, 20, 1, mg20, -1);
        FSTAMP(3);
        fused_pass<NT, 5, 1, true>(buf, A.twM + 25, 2500, 100, 1, mg100, -1);
        FSTAMP(4);
        fused_pass<NT, 5, 1, true>(buf, A.twM + 125, 2500, 500, 1, mg500, 2 * jn + 2 < 500 ? jn : -1);
        FSTAMP(5);
    } else if constexpr (SPECM != 0) {
        fused_passes_ct<NT, SPECM, 0>(buf, A.twM, pruned, jn);
    } else
    for (int ps = pruned ? 1 : 0; ps < A.plan.npass; ++ps) {
        const int R = A.plan.radix[ps], Ns = A.plan.ns[ps], tws = 1;      // (per-pass twiddle table: element k of the slice)
        const unsigned mg = A.plan.magic[ps];
        const cplx* __restrict__ twp = A.twM + A.plan.twoff[ps];
        // a pass whose butterflies fit one per thread uses the MAXB = 1 body (fewer live registers)
        const bool one = (M / R) <= NT;
        // the last pass only has to produce Z[0..jn] and Z[M-jn..M-1]
        const int keep = (ps == A.plan.npass - 1 && 2 * jn + 2 < M / R) ? jn : -1;
        if (R == 5) { if (one) fused_pass<NT, 5, 1>(buf, twp, M, Ns, tws, mg, keep); else fused_pass<NT, 5, MAXB>(buf, twp, M, Ns, tws, mg, keep); }
        else if (R == 4) { if (one) fused_pass<NT, 4, 1>(buf, twp, M, Ns, tws, mg, keep); else fused_pass<NT, 4, MAXB>(buf, twp, M, Ns, tws, mg, keep); }
        else if (R == 3) { if (one) fused_pass<NT, 3, 1>(buf, twp, M, Ns, tws, mg, keep); else fused_pass<NT, 3, MAXB>(buf, twp, M, Ns, tws, mg, keep); }
        else { if (one) fused_pass<NT, 2, 1>(buf, twp, M, Ns, tws, mg, keep); else fused_pass<NT, 2, MAXB>(buf, twp, M, Ns, tws, mg, keep); }
    }
    if constexpr (ABL == 2) {     // stop after phase B
        if (threadIdx.x < 8) A.out[(size_t)row * A.nk + threadIdx.x] = buf[threadIdx.x].x + red[24];
        return;
    }
    // ---- phase C: Im F_j -> u_j = -Im F_j * step / kt_j / mnorm for the reachable modes
    // j = 1..jn, into smem[0..jn-1]
    const double sc = red[24];                    // u_j = Im F_j * this / j   (kt_j = j kt_1)
    double ua[MAXP], ub[MAXP];
    const int half = M / 2;
#pragma unroll
    for (int b = 0; b < MAXP; ++b) {
        const int j = 1 + threadIdx.x + b * NT;
        const bool hi = (M - j <= jn);                 // the mirrored mode M-j is reachable too
        if (j <= half && (j <= jn || hi)) {
            const cplx zj = buf[j], zmj = buf[M - j];
            const UnpackTw w = A.twN[j];
            double fa, fb;
            unpack_imag_pair(zj, zmj, w.co, w.si, fa, fb);
            ua[b] = fa * sc * w.rj;
            ub[b] = hi ? fb * sc * w.rmj : 0.0;
        }
    }
    __syncthreads();
    double* u = smem;
#pragma unroll
    for (int b = 0; b < MAXP; ++b) {
        const int j = 1 + threadIdx.x + b * NT;
        const bool hi = (M - j <= jn);
        if (j <= half && (j <= jn || hi)) {
            u[j - 1] = ua[b];
            if (hi && M - j >= 1) u[M - j - 1] = ub[b];
        }
    }
    if (threadIdx.x == 0) u[M - 1] = 0.0;  // Nyquist mode: Im F_M == 0
    __syncthreads();
    FSTAMP(6);
    if constexpr (ABL == 3) {     // stop after phase C
        if (threadIdx.x < 8) A.out[(size_t)row * A.nk + threadIdx.x] = u[threadIdx.x];
        return;
    }
    // ---- phase D: np.interp(ks, kout, u, left=u_1, right=0) on the uniform source grid:
    // bracket j = floor(k/k_lo), weight k/k_lo - j (one FMA), two LDS reads.  The left fill is a
    // plain splat (63 % of the Battaglia tensor at Config 3).
    const double k_lo = rsc ? rsc[1] : red[20], k_hi = rsc ? rsc[2] : red[21], inv_dk = rsc ? rsc[3] : red[22];
    const double pf = A.post ? A.post[row] : 1.0;
    const double u1 = u[0];
    double* __restrict__ dst = A.out + (size_t)row * A.nk;
    // with the hint arrays the left fill [0, nleft) is a plain fill in 16-byte stores (no wavenumber is loaded or
    // tested) and the interpolation starts at the 64-aligned index below nleft, so that its stores stay on whole
    // 512-byte wavefront segments; without them (ks in any order) every target is tested
    const int nleft = rsc ? __double2loint(rsc[5]) : (A.nconst ? __builtin_amdgcn_readfirstlane(*s_cnt) : 0);
    if (nleft > 0) {
        typedef double v2d __attribute__((ext_vector_type(2)));
        const double c = u1 * pf;
        const int head = (int)((reinterpret_cast<uintptr_t>(dst) >> 3) & 1);     // row start not 16-B aligned
        const int npair = (nleft - head) >> 1;
        v2d* __restrict__ d2 = reinterpret_cast<v2d*>(dst + head);
        const v2d cc = {c, c};
        for (int q = threadIdx.x; q < npair; q += NT) __builtin_nontemporal_store(cc, &d2[q]);
        if (threadIdx.x == 0) {
            if (head) __builtin_nontemporal_store(c, &dst[0]);
            if ((nleft - head) & 1) __builtin_nontemporal_store(c, &dst[nleft - 1]);
        }
    }
    auto interp = [&](double k) {
        int j = (int)(k * inv_dk);
        j = j < 1 ? 1 : (j > M - 1 ? M - 1 : j);
        const double fr = fma(k, inv_dk, -(double)j);
        const double y0 = u[j - 1], y1 = u[j];
        return fma(y1 - y0, fr, y0);
    };
    if (A.nconst) {
        // behind the prefix every target is at or above k_lo (ks ascending)
        for (int i = (nleft & ~63) + threadIdx.x; i < A.nk; i += NT) {
            if (i < nleft) continue;
            // (requesting the next trip's wavenumber one trip ahead was measured: +-0, the other wavefronts of
            // the workgroup already cover the load)
            const double k = A.ks[i];
            const double val = k > k_hi ? 0.0 : interp(k);
            __builtin_nontemporal_store(val * pf, &dst[i]);
        }
    } else {
        for (int i = threadIdx.x; i < A.nk; i += NT) {
            const double k = A.ks[i];
            const double val = k < k_lo ? u1 : (k > k_hi ? 0.0 : interp(k));
            __builtin_nontemporal_store(val * pf, &dst[i]);
        }
    }
    if (A.nconst && threadIdx.x == 0) {
        A.nconst[row] = nleft;
        A.cconst[row] = u1 * pf;
    }
    FSTAMP(7);
#ifdef HMG_FR_STAMP
    if (fslot >= 0 && threadIdx.x == 0) g_fstamps[fslot * 16 + 15] = (long long)wall_clock64();
#endif
}
template <int NT, int MAXB, int MAXP, int SPECM>
__global__ __launch_bounds__(NT, (fused_occ<MAXB, SPECM>())) void profile_fused_kernel(FusedArgs A) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    profile_fused_row<NT, MAXB, MAXP, SPECM>(A, blockIdx.x, smem);
}
template <int NT, int MAXB, int MAXP, int SPECM>
__global__ __launch_bounds__(NT, (fused_occ<MAXB, SPECM>())) void profile_table_kernel(FusedArgs A) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    profile_fused_row<NT, MAXB, MAXP, SPECM, HMG_ABL, true>(A, blockIdx.x, smem);
}

// (K45p, the long radial grids with short support - profile_pruned_kernel and the chirp route: longgrid.hip, a
// translation unit of its own.  In this one the mere presence of its instantiations changed the address arithmetic
// hipcc emits for profile_group_kernel<2,3,2500> - 605 instead of 593 VALU instructions per wavefront.)

// ---------------------------------------------------------------- K7: HOD (H1-H3)
// 10^y and x^p through exp2/log2 (one transcendental each instead of the ~6x longer generic
// pow); relative error <= ~|y| * 8e-16, far inside the 1e-9 gate on the HOD arrays.
__device__ __forceinline__ double pow10_fast(double y) { return exp2(y * 3.32192809488736234787); }
__device__ __forceinline__ double powr_fast(double x, double p) { return exp2(p * log2(x)); }

struct ShmrSet {
    double Ms0, Msa, M1, M1a, b0, ba, g0, ga, d0, da;
};
__device__ __forceinline__ ShmrSet shmr_for(double z) {
    // Behroozi+10 table 2, split at z = 0.8 (hmvec/hmvec.py:668-691)
    if (z <= 0.8) return {10.72, 0.55, 12.35, 0.28, 0.44, 0.18, 1.56, 2.51, 0.57, 0.17};
    return {11.09, 0.56, 12.27, -0.84, 0.65, 0.31, 1.12, -0.53, 0.56, -0.12};
}
__device__ __forceinline__ double shmr_log10mh(double lms, double a, const ShmrSet& s) {
    const double am1 = a - 1.0;
    const double lM1 = s.M1 + s.M1a * am1;
    const double lMs0 = s.Ms0 + s.Msa * am1;
    const double beta = s.b0 + s.ba * am1;
    const double gamma = s.g0 + s.ga * am1;
    const double delta = s.d0 + s.da * am1;
    const double d = lms - lMs0;
    return -0.5 + lM1 + beta * d + pow10_fast(delta * d) / (1.0 + pow10_fast(-gamma * d));
}

constexpr int SHMR_N = 4000;
// log10 M* grid of the reference's inverse table, np.linspace(-18,18,4000) (hmvec.py:640)
__device__ __forceinline__ double shmr_grid(int j) {
    const double gstep = 36.0 / (double)(SHMR_N - 1);
    return j == SHMR_N - 1 ? 18.0 : (double)j * gstep + (-18.0);
}
// np.interp(lmh, mh, grid) with numpy's clamped ends and exact-knot rule (hmvec.py:645)
__device__ __forceinline__ double shmr_inverse(const double* mh /* LDS, SHMR_N */, double lmh) {
    if (lmh < mh[0]) return shmr_grid(0);
    if (lmh >= mh[SHMR_N - 1]) return shmr_grid(SHMR_N - 1);
    int lo = 0, hi = SHMR_N - 1;          // mh[lo] <= lmh < mh[lo+1]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (mh[mid] <= lmh) lo = mid; else hi = mid;
    }
    if (mh[lo] == lmh) return shmr_grid(lo);
    const double slope = (shmr_grid(lo + 1) - shmr_grid(lo)) / (mh[lo + 1] - mh[lo]);
    return slope * (lmh - mh[lo]) + shmr_grid(lo);
}
struct HodDev {
    double sig, alphasat, Bsat, betasat, Bcut, betacut;
    int corr;
};

// The same inversion without the table: the bracket search evaluates the table entries it visits on the
// fly (12 of the 4000 per mass).  Entry j is shmr_log10mh(shmr_grid(j)) in both forms, so the bracket, the
// knots and the interpolated value are the same numbers - but nothing has to be built first and no LDS is
// held, which is what lets the HOD of a redshift run as one link of a per-z chain inside a grouped launch
// beside workgroups of another kind (LDS is allocated per launch, for every workgroup alike).
__device__ __forceinline__ double shmr_inverse_direct(double lmh, double a, const ShmrSet& S) {
    const double m0 = shmr_log10mh(shmr_grid(0), a, S);
    if (lmh < m0) return shmr_grid(0);
    const double mN = shmr_log10mh(shmr_grid(SHMR_N - 1), a, S);
    if (lmh >= mN) return shmr_grid(SHMR_N - 1);
    int lo = 0, hi = SHMR_N - 1;          // mh[lo] <= lmh < mh[hi]
    double mlo = m0, mhi = mN;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        const double mm = shmr_log10mh(shmr_grid(mid), a, S);
        if (mm <= lmh) { lo = mid; mlo = mm; } else { hi = mid; mhi = mm; }
    }
    if (mlo == lmh) return shmr_grid(lo);
    const double slope = (shmr_grid(lo + 1) - shmr_grid(lo)) / (mhi - mlo);
    return slope * (lmh - mlo) + shmr_grid(lo);
}

struct HodRowArgs {
    int nm;
    HodDev P;
    const double *zs, *ms, *lthr, *nzm, *bh, *wm;
    double *Nc, *Ns, *NsNsm1, *NcNs, *ngal, *bg;
};
constexpr int HOD_MAX_TILES = 1024;      // 64-mass tiles per redshift (nm <= 65536)

// The HOD has two halves.  The occupation numbers <Nc>, <Ns>, <Ns(Ns-1)>, <NcNs> of a (z,m) point depend on
// INPUTS only (z, m, the stellar-mass threshold, the HOD parameters) - and carry all the cost: the SHMR
// inversion, an erf, two powers, an exp.  n_gal and b_g are sums over m of those times n(z,m), b(z,m).  A
// grouped pass therefore evaluates the occupations in its FRONT launch beside the sigma^2 contraction (no
// register cap there, one thread per point) and leaves only the sums to the per-z chain.
__device__ __forceinline__ void hod_occ_point(const HodRowArgs& A, int z, int m) {
#pragma clang fp contract(off)
    const HodDev& P = A.P;
    const double zz = A.zs[z], a = 1.0 / (1.0 + zz);
    const ShmrSet S = shmr_for(zz);
    const double thr = A.lthr[z];
    const double mthr_halo = shmr_log10mh(thr, a, S);
    const double Msat = 1.0e12 * P.Bsat * pow10_fast((mthr_halo - 12.0) * P.betasat);
    const double Mcut = 1.0e12 * P.Bcut * pow10_fast((mthr_halo - 12.0) * P.betacut);
    const double denom = sqrt(2.0) * P.sig;
    const double lmh = log10(A.ms[m]);
    const double lmstar = shmr_inverse_direct(lmh, a, S);
    const double nc = 0.5 * (1.0 - erf((thr - lmstar) / denom));
    const double mass = pow10_fast(lmh);
    const double ns = nc * powr_fast(mass / Msat, P.alphasat) * exp(-Mcut / mass);
    double nn, cn;
    if (P.corr == 0) {
        nn = (fabs(nc) <= 1.0e-8) ? 0.0 : (ns * ns) / nc;   // np.isclose(Nc, 0)
        cn = ns;
    } else {
        nn = ns * ns;
        cn = ns * nc;
    }
    const size_t idx = (size_t)z * A.nm + m;
    A.Nc[idx] = nc; A.Ns[idx] = ns; A.NsNsm1[idx] = nn; A.NcNs[idx] = cn;
}

// n_gal(z), b_g(z) of one redshift by one workgroup of nthr threads, from the stored occupations.  The order
// is fixed by nm alone: wavefront sums over the 64-mass tiles, then the tiles in order - whatever the
// workgroup size.  part: 2 * ceil(nm/64) doubles of LDS.
__device__ __forceinline__ void hod_sums_row(const HodRowArgs& A, int z, int nthr, double* part) {
#pragma clang fp contract(off)
    const int nm = A.nm, lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = nthr >> 6;
    const int ntile = (nm + 63) / 64;
    for (int tile = w; tile < ntile; tile += nw) {
        const int m = tile * 64 + lane;
        double t = 0.0, tb = 0.0;
        if (m < nm) {
            const size_t idx = (size_t)z * nm + m;
            t = A.wm[m] * (A.nzm[idx] * (A.Nc[idx] + A.Ns[idx]));
            tb = t * A.bh[idx];
        }
        const double tn = wave_sum(t), tbs = wave_sum(tb);
        if (lane == 0) { part[2 * tile] = tn; part[2 * tile + 1] = tbs; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double sn = 0.0, sb = 0.0;
        for (int tile = 0; tile < ntile; ++tile) { sn += part[2 * tile]; sb += part[2 * tile + 1]; }
        A.ngal[z] = sn;
        A.bg[z] = sb / sn;
    }
}

// One block per z: both halves.
__global__ __launch_bounds__(1024) void hod_kernel(HodRowArgs A) {
    __shared__ double part[2 * HOD_MAX_TILES];
    for (int m = threadIdx.x; m < A.nm; m += blockDim.x) hod_occ_point(A, blockIdx.x, m);
    __syncthreads();
    hod_sums_row(A, blockIdx.x, blockDim.x, part);
}

// ---------------------------------------------------------------- K6: fused mass integrals (P1-P4)
// Tracer weights are linear forms in at most NT distinct [z][m][k] tensors:
//     form_f(z,m,k) = c[f][0](z,m) + sum_t c[f][1+t](z,m) * T_t(z,m,k)
// f = 0,1: the two factors of the 1-halo integrand (trapz weight and n(z,m) folded into
// factor 0); f = 2,3: the two 2-halo integrands (weight, n and b_h folded in).
// power_prep_kernel builds the coefficient table + the k->0 consistency integrals and
// biases; power_kernel streams every distinct tensor exactly once.
constexpr int PW_NF = 4;
constexpr int PW_MAXT = 4;

struct TracerDev {
    int kind;
    int t_prof, t_cprof;  // slots in the distinct-tensor list, -1 = none
    const double *Nc, *Ns, *NcNs, *NsNsm1, *ngal, *bias_override;
};
struct PowerPrep {
    TracerDev a, b;
    int nt;
    double rho_m0;
};

// Linear form of one tracer's 2-halo weight (also its 1-halo factor in the generic case).
__device__ __forceinline__ void tracer_form(const TracerDev& T, size_t idx, int z, double mass,
                                            double rho_m0, double* c /*[1+PW_MAXT]*/,
                                            double& lowk) {
    for (int i = 0; i <= PW_MAXT; ++i) c[i] = 0.0;
    if (T.kind == HMG_TRACER_MATTER) {
        c[1 + T.t_prof] = mass / rho_m0;
        lowk = mass / rho_m0;
    } else if (T.kind == HMG_TRACER_PRESSURE) {
        c[1 + T.t_prof] = 1.0;
        lowk = 0.0;
    } else {
        const double ng = T.ngal[z], nc = T.Nc[idx], ns = T.Ns[idx];
        if (T.t_cprof >= 0) c[1 + T.t_cprof] += nc / ng; else c[0] += nc / ng;
        c[1 + T.t_prof] += ns / ng;
        lowk = (nc + ns) / ng;
    }
}

// grid nz blocks, 256 threads; coef layout [z][m][PW_NF][1+nt]; side[z][4] = {bA, CA, bB, CB}
__global__ __launch_bounds__(256) void power_prep_kernel(int nm, PowerPrep Q,
                                                         const double* __restrict__ nzm,
                                                         const double* __restrict__ bh,
                                                         const double* __restrict__ ms,
                                                         const double* __restrict__ wm,
                                                         double* __restrict__ coef,
                                                         double* __restrict__ side) {
    __shared__ double lds[16];
    const int z = blockIdx.x;
    const int nc1 = 1 + Q.nt;
    double accCA = 0.0, accCB = 0.0, accBA = 0.0, accBB = 0.0;
    for (int m = threadIdx.x; m < nm; m += blockDim.x) {
        const size_t idx = (size_t)z * nm + m;
        const double mass = ms[m];
        const double wn = wm[m] * nzm[idx];
        const double wnb = wn * bh[idx];
        double fa[1 + PW_MAXT], fb[1 + PW_MAXT], x1[1 + PW_MAXT], x2[1 + PW_MAXT];
        double lowA, lowB;
        tracer_form(Q.a, idx, z, mass, Q.rho_m0, fa, lowA);
        tracer_form(Q.b, idx, z, mass, Q.rho_m0, fb, lowB);
        if (Q.a.kind == HMG_TRACER_HOD && Q.b.kind == HMG_TRACER_HOD) {
            // (2 u_c u_s <NcNs> + <Ns(Ns-1)> u_s^2)/ngal^2 of the FIRST name (hmvec.py:510-511)
            for (int i = 0; i <= PW_MAXT; ++i) x1[i] = x2[i] = 0.0;
            const double ng = Q.a.ngal[z], ng2 = ng * ng;
            x1[1 + Q.a.t_prof] = 1.0;
            const double cc = 2.0 * Q.a.NcNs[idx] / ng2;
            if (Q.a.t_cprof >= 0) x2[1 + Q.a.t_cprof] += cc; else x2[0] += cc;
            x2[1 + Q.a.t_prof] += Q.a.NsNsm1[idx] / ng2;
        } else if (Q.a.kind == HMG_TRACER_PRESSURE && Q.b.kind == HMG_TRACER_PRESSURE) {
            // pk_a**2 — first name only (hmvec.py:512-513)
            for (int i = 0; i <= PW_MAXT; ++i) { x1[i] = fa[i]; x2[i] = fa[i]; }
        } else {
            for (int i = 0; i <= PW_MAXT; ++i) { x1[i] = fa[i]; x2[i] = fb[i]; }
        }
        double* c = coef + idx * (size_t)(PW_NF * nc1);
        for (int i = 0; i < nc1; ++i) {
            c[0 * nc1 + i] = wn * x1[i];
            c[1 * nc1 + i] = x2[i];
            c[2 * nc1 + i] = wnb * fa[i];
            c[3 * nc1 + i] = wnb * fb[i];
        }
        accCA += wnb * lowA;
        accCB += wnb * lowB;
        if (Q.a.kind == HMG_TRACER_HOD) accBA += wnb * (Q.a.Nc[idx] + Q.a.Ns[idx]);
        if (Q.b.kind == HMG_TRACER_HOD) accBB += wnb * (Q.b.Nc[idx] + Q.b.Ns[idx]);
    }
    const double CA = block_sum(accCA, lds), CB = block_sum(accCB, lds);
    const double BA = block_sum(accBA, lds), BB = block_sum(accBB, lds);
    if (threadIdx.x == 0) {
        auto bias = [&](const TracerDev& T, double hodsum) {
            if (T.bias_override) return T.bias_override[z];
            if (T.kind == HMG_TRACER_MATTER) return 1.0;
            if (T.kind == HMG_TRACER_PRESSURE) return 0.0;
            return hodsum / T.ngal[z];
        };
        side[z * 4 + 0] = bias(Q.a, BA);
        side[z * 4 + 1] = CA;
        side[z * 4 + 2] = bias(Q.b, BB);
        side[z * 4 + 3] = CB;
    }
}

struct PowerArgs {
    const double* tens[PW_MAXT];
    const double* coef;
    const double* side;
    const double* ks;
    const double* Pzk;
    double* P1h;
    double* P2h;
    double* I1;       // optional: the two 2-halo integrals I_a(z,k), I_b(z,k) and
    double* I2;
    double* Cout;     // [nz][2] their k -> 0 limits C_a, C_b (get_power_2halo(verbose=True))
    double kstar;
    int nm, nk;
};

template <int V> struct VecT;
template <> struct VecT<1> { using type = double; };
template <> struct VecT<2> { using type = double2; };


template <int V> __device__ __forceinline__ double vget(const typename VecT<V>::type& v, int i);
template <> __device__ __forceinline__ double vget<1>(const double& v, int) { return v; }
template <> __device__ __forceinline__ double vget<2>(const double2& v, int i) { return i ? v.y : v.x; }
template <int V> __device__ __forceinline__ typename VecT<V>::type vsplat(double x);
template <> __device__ __forceinline__ double vsplat<1>(double x) { return x; }
template <> __device__ __forceinline__ double2 vsplat<2>(double x) { return make_double2(x, x); }
// streamed-once tensor data: non-temporal load (does not displace the coefficient rows and hints in the caches)
template <int V> __device__ __forceinline__ typename VecT<V>::type vload_nt(const double* p);
template <> __device__ __forceinline__ double vload_nt<1>(const double* p) { return __builtin_nontemporal_load(p); }
template <> __device__ __forceinline__ double2 vload_nt<2>(const double* p) {
    typedef double d2v __attribute__((ext_vector_type(2)));
    const d2v v = __builtin_nontemporal_load(reinterpret_cast<const d2v*>(p));
    return make_double2(v.x, v.y);
}


// grid (ceil(nk/(64 V)), nz); block 64*MS threads: lane -> V consecutive k, wave -> an
// interleaved slice of the mass axis.  Each wave streams 512 B*V per tensor per mass bin
// (fully coalesced), partial sums over the MS slices are combined through LDS.
template <int NT, int V>
__global__ __launch_bounds__(1024) void power_kernel(PowerArgs A) {
    extern __shared__ double red[];  // [MS][3][V][64]
    using vec_t = typename VecT<V>::type;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int MS = blockDim.x >> 6;
    const int z = blockIdx.y;
    const int k0 = (blockIdx.x * 64 + lane) * V;
    const bool live = k0 < A.nk;  // nk % V == 0 is guaranteed by the launcher
    constexpr int NC1 = 1 + NT;
    double a1[V], aA[V], aB[V];
#pragma unroll
    for (int v = 0; v < V; ++v) a1[v] = aA[v] = aB[v] = 0.0;
    const size_t zrow = (size_t)z * A.nm;
#pragma unroll 4
    for (int m = wv; m < A.nm; m += MS) {
        const double* __restrict__ c = A.coef + (zrow + m) * (size_t)(PW_NF * NC1);
        vec_t t[NT];
        const size_t off = (zrow + m) * (size_t)A.nk + k0;
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            if (live) t[i] = vload_nt<V>(A.tens[i] + off);
            else t[i] = vec_t{};
        }
#pragma unroll
        for (int v = 0; v < V; ++v) {
            double f0 = c[0 * NC1], f1 = c[1 * NC1], f2 = c[2 * NC1], f3 = c[3 * NC1];
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const double tv = vget<V>(t[i], v);
                f0 += c[0 * NC1 + 1 + i] * tv;
                f1 += c[1 * NC1 + 1 + i] * tv;
                f2 += c[2 * NC1 + 1 + i] * tv;
                f3 += c[3 * NC1 + 1 + i] * tv;
            }
            a1[v] += f0 * f1;
            aA[v] += f2;
            aB[v] += f3;
        }
    }
    // combine the MS mass slices
#pragma unroll
    for (int v = 0; v < V; ++v) {
        red[((wv * 3 + 0) * V + v) * 64 + lane] = a1[v];
        red[((wv * 3 + 1) * V + v) * 64 + lane] = aA[v];
        red[((wv * 3 + 2) * V + v) * 64 + lane] = aB[v];
    }
    __syncthreads();
    if (wv == 0 && live) {
        const double bA = A.side[z * 4 + 0], CA = A.side[z * 4 + 1];
        const double bB = A.side[z * 4 + 2], CB = A.side[z * 4 + 3];
#pragma unroll
        for (int v = 0; v < V; ++v) {
            double s1 = 0.0, sA = 0.0, sB = 0.0;
            for (int w = 0; w < MS; ++w) {
                s1 += red[((w * 3 + 0) * V + v) * 64 + lane];
                sA += red[((w * 3 + 1) * V + v) * 64 + lane];
                sB += red[((w * 3 + 2) * V + v) * 64 + lane];
            }
            const int k = k0 + v;
            const size_t o = (size_t)z * A.nk + k;
            if (A.P1h) {
                const double q = A.ks[k] / A.kstar;
                A.P1h[o] = s1 * (1.0 - exp(-(q * q)));
            }
            if (A.P2h) A.P2h[o] = A.Pzk[o] * (sA + bA - CA) * (sB + bB - CB);
            if (A.I1) A.I1[o] = sA;
            if (A.I2) A.I2[o] = sB;
            if (A.Cout && k == 0) { A.Cout[z * 2] = CA; A.Cout[z * 2 + 1] = CB; }
        }
    }
}


// ---------------------------------------------------------------- K6b: all-pairs mass integrals
// When several spectra share profile tensors (Config 3: six spectra over TWO tensors, because
// the galaxy tracer's satellite profile is the NFW tensor), the per-pair kernel re-reads
// each tensor once per pair (sum d = 8 tensor passes).  This kernel takes NTR tracers over NT
// distinct tensors and accumulates, in ONE pass over the tensors, the NTR 2-halo integrals
// I_t and all NTR(NTR+1)/2 1-halo integrals; the 2-halo spectrum of any pair is assembled in
// the epilogue from (I_a, I_b).  Per-tracer forms: W (2-halo weight and cross 1-halo factor)
// and (A1, A2), the two factors of the tracer's 1-halo AUTO integrand (= W, W except for an
// HOD, whose auto term is (2 u_c u_s <NcNs> + <Ns(Ns-1)> u_s^2)/ngal^2).
constexpr int PB_MAXTR = 4;
constexpr int PB_MAXPAIR = PB_MAXTR * (PB_MAXTR + 1) / 2;

// Structure of a batch.  Most coefficients of the generic linear forms are structural zeros or ones:
//   matter / pressure on tensor s   ("LIN s"):  W = A1 = A2 = c t_s                          1 number per (z,m)
//   HOD, satellites on s, u_c == 1  ("HOD s"):  W = c0 + c1 t_s, A1 = t_s, A2 = a0 + a1 t_s  4 numbers
// For the batches the facade issues most (PB_SPEC_LIST) the kernel is compiled for that structure: a row of
// 2 + sum numbers instead of 2 + 3 NTR (1+NT), padded to whole 64-byte lines - Config 3: 8 doubles, one
// s_load_dwordx16, against 29 - and only the non-zero terms are evaluated, with the operations the generic
// forms apply to them (adding an exact zero or multiplying by an exact one changes no bit), so both paths
// give the same sums.  What the mass loop cannot afford is scalar-memory traffic per bin.
// A code packs 4 bits per tracer, tracer 0 lowest: bits 0-1 kind (0 none, 1 LIN, 2 HOD), bits 2-3 tensor slot.
constexpr unsigned PB_LIN(int s) { return 1u | ((unsigned)s << 2); }
constexpr unsigned PB_HOD(int s) { return 2u | ((unsigned)s << 2); }
constexpr unsigned pb_code(unsigned t0, unsigned t1 = 0, unsigned t2 = 0, unsigned t3 = 0) {
    return t0 | (t1 << 4) | (t2 << 8) | (t3 << 12);
}
constexpr int pb_kind(unsigned code, int r) { return (int)((code >> (4 * r)) & 3u); }
constexpr int pb_slot(unsigned code, int r) { return (int)((code >> (4 * r + 2)) & 3u); }
constexpr int pb_ncoef(unsigned code, int ntr) {     // numbers per compact row before padding
    int n = 2;
    for (int r = 0; r < ntr; ++r) n += pb_kind(code, r) == 2 ? 4 : 1;
    return n;
}
constexpr int pb_stride(unsigned code, int ntr, int nc1) {   // doubles per (z,m) coefficient row
    return code ? ((pb_ncoef(code, ntr) + 7) & ~7) : 2 + ntr * 3 * nc1;
}

// (distinct tensors, tracers, structure) the kernel is compiled for: what get_power*/spectra_block produce for
// matter and pressure profiles and an HOD whose satellites follow the first matter profile (the reference's
// README usage: 'nfw', a Battaglia gas profile, a pressure profile, an HOD)
#define PB_SPEC_LIST                                                  \
    PB_SPEC(1, 1, PB_LIN(0))                                          \
    PB_SPEC(1, 2, PB_LIN(0), PB_HOD(0))                               \
    PB_SPEC(2, 2, PB_LIN(0), PB_LIN(1))                               \
    PB_SPEC(2, 3, PB_LIN(0), PB_LIN(1), PB_HOD(0))                    \
    PB_SPEC(3, 3, PB_LIN(0), PB_LIN(1), PB_LIN(2))                    \
    PB_SPEC(3, 4, PB_LIN(0), PB_LIN(1), PB_LIN(2), PB_HOD(0))

struct BatchPrep {
    TracerDev tr[PB_MAXTR];
    int ntr, nt;
    double rho_m0;
    unsigned code;       // 0: generic rows; else the compact rows of that structure
};

// coef layout per (z,m): [wn, wnb, {W[1+nt], A1[1+nt], A2[1+nt]} x ntr].
// grid (nz, nblk) with 64-thread blocks, one (z,m) per thread; the k->0 consistency sums C_t
// and HOD bias numerators B_t are written as per-block partials sidep[z][blk][t][2] = {B, C}
// and summed in block order by the main kernel's epilogue (deterministic).
struct PrepArgs {
    int nm, nblk;
    BatchPrep Q;
    const double *nzm, *bh, *ms, *wm;
    double *coef, *sidep;
};
// one wavefront = the 64 masses of tile blk of redshift z
__device__ __forceinline__ void batch_prep_tile(const PrepArgs& PA, int z, int blk) {
    const int nm = PA.nm, nblk = PA.nblk;
    const BatchPrep& Q = PA.Q;
    const double* __restrict__ nzm = PA.nzm;
    const double* __restrict__ bh = PA.bh;
    const double* __restrict__ ms = PA.ms;
    const double* __restrict__ wm = PA.wm;
    double* __restrict__ coef = PA.coef;
    double* __restrict__ sidep = PA.sidep;
    const int m = blk * 64 + (threadIdx.x & 63);
    const int nc1 = 1 + Q.nt;
    const int stride = pb_stride(Q.code, Q.ntr, nc1);
    double accC[PB_MAXTR], accB[PB_MAXTR];
    for (int t = 0; t < PB_MAXTR; ++t) accC[t] = accB[t] = 0.0;
    if (m < nm) {
        const size_t idx = (size_t)z * nm + m;
        const double mass = ms[m];
        const double wn = wm[m] * nzm[idx];
        const double wnb = wn * bh[idx];
        double* c = coef + idx * (size_t)stride;
        c[0] = wn;
        c[1] = wnb;
        int pos = 2;
        for (int t = 0; t < Q.ntr; ++t) {
            const TracerDev& T = Q.tr[t];
            double w[1 + PW_MAXT], a1[1 + PW_MAXT], a2[1 + PW_MAXT], low;
            tracer_form(T, idx, z, mass, Q.rho_m0, w, low);
            for (int i = 0; i <= PW_MAXT; ++i) { a1[i] = w[i]; a2[i] = w[i]; }
            if (T.kind == HMG_TRACER_HOD) {
                for (int i = 0; i <= PW_MAXT; ++i) a1[i] = a2[i] = 0.0;
                const double ng = T.ngal[z], ng2 = ng * ng;
                a1[1 + T.t_prof] = 1.0;
                const double cc = 2.0 * T.NcNs[idx] / ng2;
                if (T.t_cprof >= 0) a2[1 + T.t_cprof] += cc; else a2[0] += cc;
                a2[1 + T.t_prof] += T.NsNsm1[idx] / ng2;
                accB[t] = wnb * (T.Nc[idx] + T.Ns[idx]);
            }
            if (Q.code == 0) {
                double* ct = c + 2 + t * 3 * nc1;
                for (int i = 0; i < nc1; ++i) {
                    ct[i] = w[i];
                    ct[nc1 + i] = a1[i];
                    ct[2 * nc1 + i] = a2[i];
                }
            } else {                  // compact row: only the numbers that are not structural zeros / ones
                const int sl = 1 + pb_slot(Q.code, t);
                if (pb_kind(Q.code, t) == 1) {
                    c[pos++] = w[sl];
                } else {
                    c[pos++] = w[0]; c[pos++] = w[sl]; c[pos++] = a2[0]; c[pos++] = a2[sl];
                }
            }
            accC[t] = wnb * low;
        }
        if (Q.code) for (; pos < stride; ++pos) c[pos] = 0.0;
    }
    for (int t = 0; t < Q.ntr; ++t) {
        const double C = wave_sum(accC[t]);
        const double B = wave_sum(accB[t]);
        if ((threadIdx.x & 63) == 0) {
            double* sp = sidep + ((size_t)(z * nblk + blk) * Q.ntr + t) * 2;
            sp[0] = B;
            sp[1] = C;
        }
    }
}
// The same rows for a batch with a structure code (every batch of PB_SPEC_LIST), written without the
// generic forms' dynamically indexed coefficient arrays: a handful of registers, so that it can run as a
// link of the per-z chain inside the profile group under that kernel's 64-register budget without spilling
// (a spill anywhere gives the whole launch a scratch allocation, which cost the fused profile rows 7 %).
// Same numbers as batch_prep_tile: the generic forms add these terms to exact zeros.
__device__ __forceinline__ void batch_prep_tile_compact(const PrepArgs& PA, int z, int blk) {
    const BatchPrep& Q = PA.Q;
    const int nm = PA.nm, lane = threadIdx.x & 63, m = blk * 64 + lane;
    int ncoef = 2;
#pragma unroll
    for (int t = 0; t < PB_MAXTR; ++t)
        if (t < Q.ntr) ncoef += Q.tr[t].kind == HMG_TRACER_HOD ? 4 : 1;
    const int stride = (ncoef + 7) & ~7;
    double accC[PB_MAXTR], accB[PB_MAXTR];
#pragma unroll
    for (int t = 0; t < PB_MAXTR; ++t) accC[t] = accB[t] = 0.0;
    if (m < nm) {
        const size_t idx = (size_t)z * nm + m;
        const double mass = PA.ms[m];
        const double wn = PA.wm[m] * PA.nzm[idx];
        const double wnb = wn * PA.bh[idx];
        double* __restrict__ c = PA.coef + idx * (size_t)stride;
        c[0] = wn;
        c[1] = wnb;
        int pos = 2;
#pragma unroll
        for (int t = 0; t < PB_MAXTR; ++t) {
            if (t >= Q.ntr) continue;
            const TracerDev& T = Q.tr[t];
            double low = 0.0;
            if (T.kind == HMG_TRACER_HOD) {
                const double ng = T.ngal[z], ng2 = ng * ng, nc = T.Nc[idx], ns = T.Ns[idx];
                c[pos] = nc / ng;
                c[pos + 1] = ns / ng;
                c[pos + 2] = 2.0 * T.NcNs[idx] / ng2;
                c[pos + 3] = T.NsNsm1[idx] / ng2;
                pos += 4;
                accB[t] = wnb * (nc + ns);
                low = (nc + ns) / ng;
            } else {
                low = T.kind == HMG_TRACER_MATTER ? mass / Q.rho_m0 : 0.0;
                c[pos++] = T.kind == HMG_TRACER_MATTER ? low : 1.0;
            }
            accC[t] = wnb * low;
        }
        for (; pos < stride; ++pos) c[pos] = 0.0;
    }
#pragma unroll
    for (int t = 0; t < PB_MAXTR; ++t) {
        if (t >= Q.ntr) continue;
        const double C = wave_sum(accC[t]);
        const double B = wave_sum(accB[t]);
        if (lane == 0) {
            double* sp = PA.sidep + ((size_t)(z * PA.nblk + blk) * Q.ntr + t) * 2;
            sp[0] = B;
            sp[1] = C;
        }
    }
}
__global__ __launch_bounds__(64) void power_batch_prep_kernel(PrepArgs PA) {
    if (PA.Q.code) batch_prep_tile_compact(PA, blockIdx.x, blockIdx.y);
    else batch_prep_tile(PA, blockIdx.x, blockIdx.y);
}

struct BatchArgs {
    const double* tens[PW_MAXT];
    const int* nconst[PW_MAXT];      // constant-prefix hint of tensor i ([nz][nm]) or nullptr
    const double* cconst[PW_MAXT];
    const double* coef;
    const double* sidep;             // [nz][nblk][NTR][2] partial {B, C}
    const double* ngal[PB_MAXTR];    // HOD tracers: ngal[z] (bias = B/ngal); else nullptr
    double bias_const[PB_MAXTR];     // matter 1, pressure 0
    int nblk;
    const double* ks;
    const double* Pzk;
    double* P1h[PB_MAXPAIR];  // canonical pair index of (a<=b): a*NTR - a(a-1)/2 + (b-a)
    double* P2h[PB_MAXPAIR];
    double kstar;
    int nm, nk;
};

// Summation order over the mass axis (fixed by nm alone, so that a z-slab run and the full grid agree
// bit for bit whatever launch shape each picks): PB_NV = 16 virtual slices, slice v = the bins
// m = v, v+16, v+32, ... summed in that order from zero; then the pair sums t_w = s_w + s_{w+8};
// then t_0 + t_1 + ... + t_7 in that order.  Two launch shapes realise it:
//   W16 = false: 8 wavefronts (512 threads), wavefront w walks slice w, parks the sums in its private
//                part of LDS (no barrier), walks slice w+8 and adds the parked sums at the end;
//   W16 = true : 16 wavefronts (1024 threads), one slice each - twice the loads in flight per CU,
//                which is what a thin z-slab (one workgroup per CU) needs.
constexpr int PB_NV = 16;

template <int NT, int NTR, int V, bool W16, unsigned CODE = 0>
__global__ __launch_bounds__(W16 ? 1024 : 512) void power_batch_kernel(BatchArgs A) {
    extern __shared__ double red[];  // [8][NACC*V][64]: parked sums / pair exchange, then the cross-wave reduction
    using vec_t = typename VecT<V>::type;
    constexpr int NC1 = 1 + NT;
    constexpr int STRIDE = pb_stride(CODE, NTR, NC1);
    constexpr int NPAIR = NTR * (NTR + 1) / 2;
    constexpr int NACC = NTR + NPAIR;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int z = blockIdx.y;
    const int k0 = (blockIdx.x * 64 + lane) * V;
    const bool live = k0 < A.nk;
    double acc[NACC][V];      // [0, NTR): 2-halo integrals I_t; [NTR, NACC): 1-halo integrals of the pairs
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int v = 0; v < V; ++v) acc[a][v] = 0.0;
    const size_t zrow = (size_t)z * A.nm;
    const int kend = min(A.nk, (int)(blockIdx.x + 1) * 64 * V);   // one past the last k of this tile
    const size_t kofs = live ? (size_t)k0 : 0;   // dead lanes re-read column 0 (their sums are never stored)
    // the bins of this wavefront, in order: position i -> mass bin (>= nm: no such bin)
    const int L = (A.nm + PB_NV - 1) / PB_NV;            // positions per slice
    const int NB = W16 ? L : 2 * L;
    auto bin = [&](int i) {
        if (W16) return wv + PB_NV * i;
        return i < L ? wv + PB_NV * i : wv + 8 + PB_NV * (i - L);
    };
    // constant-prefix hint of mass bin m: how many leading k of the row equal `val`
    struct Hint { int n[NT]; double val[NT]; };
    auto load_hint = [&](Hint& h, int m) {
        const size_t r = zrow + min(m, A.nm - 1);
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            h.n[i] = A.nconst[i] ? A.nconst[i][r] : -1;
            h.val[i] = A.nconst[i] ? A.cconst[i][r] : 0.0;
        }
    };
    // rows whose whole k tile lies in a tensor's constant prefix are not read at all
    auto fetch = [&](vec_t (&dst)[NT], int m, const Hint& h) {
        const size_t off = (zrow + min(m, A.nm - 1)) * (size_t)A.nk + kofs;
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            if (h.n[i] >= kend) dst[i] = vsplat<V>(h.val[i]);
            else dst[i] = vload_nt<V>(A.tens[i] + off);
        }
    };
    auto park = [&]() {      // end of the first slice (8-wavefront shape): sums to LDS, start again from zero
#pragma unroll
        for (int a = 0; a < NACC; ++a)
#pragma unroll
            for (int v = 0; v < V; ++v) {
                red[((wv * NACC + a) * V + v) * 64 + lane] = acc[a][v];
                acc[a][v] = 0.0;
            }
    };
    auto accumulate = [&](const vec_t (&t)[NT], int i) {
        if (!W16 && i == L) park();
        const int m = bin(i);
        if (m >= A.nm) return;
        const double* __restrict__ c = A.coef + (zrow + m) * (size_t)STRIDE;
        const double wn = c[0], wnb = c[1];
#pragma unroll
        for (int v = 0; v < V; ++v) {
            double W[NTR], A1[NTR], A2[NTR];
            if constexpr (CODE != 0) {
                // compiled for this batch's structure: only the non-zero terms of the forms
                int pos = 2;
#pragma unroll
                for (int r = 0; r < NTR; ++r) {
                    const double ts = vget<V>(t[pb_slot(CODE, r)], v);
                    if (pb_kind(CODE, r) == 1) {          // (constant after unrolling)
                        W[r] = c[pos] * ts;
                        A1[r] = W[r]; A2[r] = W[r];
                        pos += 1;
                    } else {
                        W[r] = fma(c[pos + 1], ts, c[pos]);
                        A1[r] = ts;
                        A2[r] = fma(c[pos + 3], ts, c[pos + 2]);
                        pos += 4;
                    }
                    acc[r][v] += wnb * W[r];
                }
            } else
#pragma unroll
            for (int r = 0; r < NTR; ++r) {
                const double* cr = c + 2 + r * 3 * NC1;
                double w = cr[0], a1 = cr[NC1], a2 = cr[2 * NC1];
#pragma unroll
                for (int i2 = 0; i2 < NT; ++i2) {
                    const double tv = vget<V>(t[i2], v);
                    w += cr[1 + i2] * tv;
                    a1 += cr[NC1 + 1 + i2] * tv;
                    a2 += cr[2 * NC1 + 1 + i2] * tv;
                }
                W[r] = w; A1[r] = a1; A2[r] = a2;
                acc[r][v] += wnb * w;
            }
            int p = NTR;
#pragma unroll
            for (int a = 0; a < NTR; ++a) {
                acc[p][v] += wn * (A1[a] * A2[a]);
                ++p;
#pragma unroll
                for (int b = a + 1; b < NTR; ++b) {
                    acc[p][v] += wn * (W[a] * W[b]);
                    ++p;
                }
            }
        }
    };
    // Two-stage software pipeline over this wavefront's bins: the loads of the next bin are in flight
    // while the current one is consumed, and the hints run one bin further ahead so that a fetch never
    // waits for its own decision.  The scheduling barriers keep hipcc from sinking the early loads back
    // down to their first use.
    int i = 0;
    // (Round 3, thin z-slabs: a four-stage version of this pipeline - three bins of tensor loads in flight - was
    // measured on the 16-wavefront shapes: 25.9 -> 26.0 us at nz = 4, 39.9 -> 40.7 at nz = 8.  What a wavefront
    // waits for there is the scalar load of the next bin's coefficient row, which cannot run ahead: two rows do
    // not fit the scalar register file.  Staging each wavefront's rows in LDS a chunk ahead - vector loads in
    // flight during the previous chunk, coefficients read by broadcast ds_read_b64 - was also built: bit-identical
    // and slower, 26.4 -> 39.8 us at nz = 4 and 41.4 -> 45.0 at nz = 8, since 29 LDS reads per bin and wavefront
    // occupy the LDS pipe for longer than the scalar round trip they replace.  Fetching the (8-double, structure-
    // compiled) coefficient row one bin ahead with the tensors: 26.8 -> 26.9 us at nz = 4, 134.6 -> 138.2 at nz = 32.
    // The thin launch moves its 97 MB at 3.7 TB/s with every CU holding ~32 KB of loads in flight, the same
    // per-CU amount all shapes of this kernel reach: it is the memory system's latency.)
    vec_t ta[NT], tb[NT];
    Hint ha, hb;
    load_hint(ha, bin(0));
    load_hint(hb, bin(1));
    fetch(ta, bin(0), ha);
#pragma unroll 1
    for (; i + 1 < NB; i += 2) {
        fetch(tb, bin(i + 1), hb);
        load_hint(ha, bin(i + 2));
        __builtin_amdgcn_sched_barrier(0);
        accumulate(ta, i);
        __builtin_amdgcn_sched_barrier(0);
        fetch(ta, bin(i + 2), ha);
        load_hint(hb, bin(i + 3));
        __builtin_amdgcn_sched_barrier(0);
        accumulate(tb, i + 1);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (i < NB) accumulate(ta, i);
    // pair sums t_w = s_w + s_{w+8}
    if (W16) {
        if (wv >= 8) {
#pragma unroll
            for (int a = 0; a < NACC; ++a)
#pragma unroll
                for (int v = 0; v < V; ++v) red[(((wv - 8) * NACC + a) * V + v) * 64 + lane] = acc[a][v];
        }
        __syncthreads();
    }
    if (wv < 8) {
#pragma unroll
        for (int a = 0; a < NACC; ++a)
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const double other = red[((wv * NACC + a) * V + v) * 64 + lane];
                acc[a][v] = W16 ? acc[a][v] + other : other + acc[a][v];     // s_w + s_{w+8}
            }
    }
    // ordered sum over the eight pair sums through LDS, one accumulator at a time (the parked values have
    // been consumed: the same memory serves as [8][V][64] exchange buffer)
    auto reduce = [&](double (&x)[V]) {
        __syncthreads();
        if (wv < 8) {
#pragma unroll
            for (int v = 0; v < V; ++v) red[(wv * V + v) * 64 + lane] = x[v];
        }
        __syncthreads();
        if (wv == 0) {
#pragma unroll
            for (int v = 0; v < V; ++v) {
                double sum = 0.0;
                for (int w = 0; w < 8; ++w) sum += red[(w * V + v) * 64 + lane];
                x[v] = sum;
            }
        }
    };
#pragma unroll
    for (int a = 0; a < NACC; ++a) reduce(acc[a]);
    if (wv == 0 && live) {
        double bmc[NTR];  // b_t - C_t
#pragma unroll
        for (int t = 0; t < NTR; ++t) {
            double B = 0.0, C = 0.0;
            for (int blk = 0; blk < A.nblk; ++blk) {
                const double* sp = A.sidep + ((size_t)(z * A.nblk + blk) * NTR + t) * 2;
                B += sp[0];
                C += sp[1];
            }
            const double bias = A.ngal[t] ? B / A.ngal[t][z] : A.bias_const[t];
            bmc[t] = bias - C;
        }
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const int k = k0 + v;
            const size_t o = (size_t)z * A.nk + k;
            const double q = A.ks[k] / A.kstar;
            const double damp = 1.0 - exp(-(q * q));
            const double plin = A.Pzk ? A.Pzk[o] : 0.0;
            int p = 0;
#pragma unroll
            for (int a = 0; a < NTR; ++a) {
#pragma unroll
                for (int b = a; b < NTR; ++b) {
                    if (A.P1h[p]) A.P1h[p][o] = acc[NTR + p][v] * damp;
                    // (the two brackets are multiplied first: commutative, so the result does not depend on
                    // which of the two tracers got the lower index in this batch)
                    if (A.P2h[p]) A.P2h[p][o] = plin * ((acc[a][v] + bmc[a]) * (acc[b][v] + bmc[b]));
                    ++p;
                }
            }
        }
    }
}

// ---------------------------------------------------------------- grouped launches
// Stages of a pass that do not depend on each other share ONE launch as disjoint block ranges of one grid:
// a kernel boundary costs ~2 us and on a thin z-slab (the rank of an 8-GPU job holds 4 redshifts) every
// per-(z,m) launch is pure latency - sigma^2 17 us, HOD 9, coefficient rows 6, row parameters 5 against
// 90 us for the three chip-filling kernels.  Streams do not help on this runtime (a cross-stream
// dependency costs more than it hides); block ranges do: the short, latency-bound
// workgroups come first in the grid, are dispatched first and finish while the long ones still fill the chip.
//   front   (64 threads):  halo stage points | sigma^2 contraction blocks           - both need inputs only
//   rows    (256 threads): per-z chain | Battaglia row parameters | analytic NFW rows - need the front
//   profile (512 threads): per-z chain | fused radial-profile rows                   - needs the rows group
// The per-z CHAIN is what the mass integrals wait for besides the tensors: second stage of sigma^2 + n, b
// -> HOD -> coefficient rows of the batched mass integrals, one workgroup per redshift, each link optional.
// Every role runs the device function of its stand-alone kernel, so grouped and separate launches give
// the same bits (tests/test_gpu_groups.py).
struct RowsArgs {
    int n;            // nz*nm, 0: no such role in this launch
    int kind, nm;
    const double *m200, *r200, *rvir, *zs, *rhoc, *hz;
    RowFit F;
    double gamma, alpha_const, pref, post_pref;
    RowOut O;
};
struct SigmaFrontArgs {
    int nz, nzp, nm, nq, gx, nseg;
    const double *PT, *kq, *wq, *R;
    double tswitch;
    double* partial;
};
template <int ZB>
__global__ __launch_bounds__(64, HMG_SIG_OCC) void front_group_kernel(SigmaFrontArgs G, HaloStageArgs H, int nhalo,
                                                                     HodRowArgs O, int nocc, RowsArgs Rw) {
    int b = blockIdx.x;
    if (b < nocc) {               // HOD occupations: the longest dependent chain of the launch, so first in the grid
        const int idx = b * 64 + threadIdx.x;
        if (idx < G.nz * O.nm) hod_occ_point(O, idx / O.nm, idx - (idx / O.nm) * O.nm);
        return;
    }
    b -= nocc;
    if (b < nhalo) {
        const int idx = b * 64 + threadIdx.x;
        if (idx < H.nz * H.nm) {
            // the Battaglia row parameters need only what this thread has just computed (M_200c, R_200c, r_vir):
            // the same thread goes on to them, from the same values the stand-alone launch would load
            double rv, m2, r2;
            halo_stage_point(H, idx, &rv, &m2, &r2);
            if (Rw.n) {
                const int z = idx / Rw.nm;
                rowparams_body(Rw.kind, idx, m2, r2, rv, 1.0 + Rw.zs[z], Rw.rhoc[z], Rw.hz ? Rw.hz[z] : 1.0, Rw.F,
                               Rw.gamma, Rw.alpha_const, Rw.pref, Rw.post_pref, Rw.O);
            }
        }
        return;
    }
    b -= nhalo;
    const int r = b / G.gx, bx = b - r * G.gx;
    const int bz = r / G.nseg, seg = r - bz * G.nseg;
    sigma2_mfma_block<ZB>(bx, seg, bz, G.nz, G.nzp, G.nm, G.nq, G.PT, G.kq, G.wq, G.R, G.tswitch, G.partial);
}

struct ChainArgs {
    int has_hod, has_prep;
    HodRowArgs H;
    PrepArgs PA;
};
// doubles of LDS a chain workgroup needs
static inline size_t chain_lds_doubles(int nm) { return 2 * (size_t)((nm + 63) / 64); }
// The chain is kept LIGHT on purpose - the n_gal, b_g sums of an HOD and the compact coefficient rows: loads, a
// few divisions, wavefront sums - so that it fits the register budget of the launch it rides in without a
// spill.  Everything heavy of the HOD (its occupation numbers: SHMR inversion, erf, powers) is in the front
// launch or, when there is no front to ride with, in hmg_hod's own kernel.
#define HMG_KERNARG __attribute__((address_space(4)))
template <class T>
__device__ __forceinline__ const T HMG_KERNARG* uniform_kernarg(const T HMG_KERNARG* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const T HMG_KERNARG*)(((unsigned long long)hi << 32) | lo);
}
template <class T>
__device__ __forceinline__ T kernarg_load(const T HMG_KERNARG* p) {      // a by-value copy out of the segment,
    static_assert(sizeof(T) % 8 == 0, "pad the argument block to 8 bytes");   // word by word through the constant
    union { T v; unsigned long long w[sizeof(T) / 8]; } u;                   // address space (-> scalar loads)
    const unsigned long long HMG_KERNARG* q = (const unsigned long long HMG_KERNARG*)p;
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 8; ++i) u.w[i] = q[i];
    return u.v;
}
// ROLES OF A GROUPED LAUNCH (chain_row, rows_block, massfn_block, nfw_rows, profile_fused_row, the front's roles) ARE
// __forceinline__ INTO THEIR __global__ KERNEL AND TAKE KERNEL PARAMETERS BY VALUE.  Round 3 tried a role as a
// `noinline` function with its arguments behind a pointer: hipcc 7.2 lost the thread index on one path of it and part
// of a workgroup skipped a barrier (a launch that never finished); read through __builtin_amdgcn_kernarg_segment_ptr()
// INSIDE a called function the argument block sits at address 0 (a memory fault).  DESIGN.md section 3, "What stalled
// and what aborted in round 3".
template <int NT>
__device__ __forceinline__ void chain_row(const ChainArgs& C, int z, double* lds) {
    if (C.has_hod) {
        hod_sums_row(C.H, z, NT, lds);
        __syncthreads();
    }
    if (C.has_prep)
        for (int blk = threadIdx.x >> 6; blk < C.PA.nblk; blk += NT / 64) batch_prep_tile_compact(C.PA, z, blk);
}

__device__ __forceinline__ void rows_block(const RowsArgs& Rw, int b) {
    const int idx = b * 256 + threadIdx.x;
    if (idx < Rw.n) {
        const int z = idx / Rw.nm;
        rowparams_body(Rw.kind, idx, Rw.m200[idx], Rw.r200[idx], Rw.rvir[idx], 1.0 + Rw.zs[z], Rw.rhoc[z],
                       Rw.hz ? Rw.hz[z] : 1.0, Rw.F, Rw.gamma, Rw.alpha_const, Rw.pref, Rw.post_pref, Rw.O);
    }
}
__device__ __forceinline__ void massfn_block(const SigmaMassFnArgs& S, int b, int nt) {
    __shared__ double red[4 * 64];
    __shared__ double sig[64];
    sigma2_massfn_tile(S, b / nt, b - (b / nt) * nt, red, sig);
}
struct RowsGroupArgs {
    ChainArgs C;
    RowsArgs Rw;
    SigmaMassFnArgs S;
    int nchain, nrowblk, nmfblk, mf_ntile;
};
// (the NFW role's pointers are kernel parameters of their own: see nfw_rows)
#ifndef HMG_ROWS_OCC
#define HMG_ROWS_OCC 7
#endif
__global__ __launch_bounds__(256, HMG_ROWS_OCC) void rows_group_kernel(RowsGroupArgs G, const SiciTable* __restrict__ T,
                                                                      const double* __restrict__ acoef, int ktile, int nm,
                                                                      int nk, const double* __restrict__ cs,
                                                                      const double* __restrict__ rss,
                                                                      const double* __restrict__ zs,
                                                                      const double* __restrict__ ks,
                                                                      double* __restrict__ uk) {
    extern __shared__ double lds[];
    int b = blockIdx.x;
    if (b < G.nchain) {
        chain_row<256>(G.C, b, lds);
        return;
    }
    b -= G.nchain;
    if (b < G.nmfblk) {
        massfn_block(G.S, b, G.mf_ntile);
        return;
    }
    b -= G.nmfblk;
    if (b < G.nrowblk) {
        rows_block(G.Rw, b);
        return;
    }
    nfw_rows(T, acoef, ktile, nm, nk, cs, rss, zs, ks, uk, b - G.nrowblk, 256, threadIdx.x);
}

template <int MAXB, int MAXP, int SPECM>
__global__ __launch_bounds__(512, (fused_occ<MAXB, SPECM>())) void profile_group_kernel(ChainArgs C, FusedArgs A,
                                                                                              int nchain) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int b = blockIdx.x;
    if (b < nchain) {
        chain_row<512>(C, b, smem);
        return;
    }
    profile_fused_row<512, MAXB, MAXP, SPECM>(A, row_order(b - nchain, A.nm), smem);
}

// ---------------------------------------------------------------- N1: Limber integral
// C_ell = int dz pref(z) P(z, k=(ell+1/2)/chi(z)) with P bilinear in (z,k) on the model grid,
// clamped to the grid box (hmvec/cosmology.py:867-904; the degree-1 fitpack spline the
// reference evaluates clamps its arguments).  One thread per multipole, loop over the nz_w
// window redshifts; wz = trapezoid weights over those redshifts (or {1} for a delta window).
// One thread per (multipole, window redshift) term - the bracket searches and the four spectrum loads of the nells x ngz
// terms are independent, and a thread per multipole walking its redshifts one after the other (rounds 1-4) spent 40 us
// in 32 x 12 dependent loads for 2000 multipoles - and one lane per multipole adds the terms up in ascending g, 32 at a
// time through LDS: the same order of sums as the sequential loop, so the same bits.
constexpr int LIMBER_G = 32, LIMBER_E = 8;       // window redshifts per round x multipoles per workgroup (256 threads)
__global__ __launch_bounds__(LIMBER_G * LIMBER_E) void limber_kernel(
    int nells, const double* __restrict__ ells, int nz, int nk, const double* __restrict__ zs,
    const double* __restrict__ ks, const double* __restrict__ P, const double* __restrict__ P2, int ngz,
    const double* __restrict__ gzs, const double* __restrict__ pref, const double* __restrict__ chis,
    const double* __restrict__ wz, double* __restrict__ out) {
#pragma clang fp contract(off)
    __shared__ double term[LIMBER_E][LIMBER_G];
    const int el = threadIdx.x / LIMBER_G, gs = threadIdx.x - el * LIMBER_G;
    const int e = blockIdx.x * LIMBER_E + el;
    const double ell = e < nells ? ells[e] : 0.0;
    double acc = 0.0;
    for (int g0 = 0; g0 < ngz; g0 += LIMBER_G) {
        const int g = g0 + gs;
        double t = 0.0;
        if (e < nells && g < ngz) {
            double k = (ell + 0.5) / chis[g];
            k = fmin(fmax(k, ks[0]), ks[nk - 1]);
            int lo = 0, hi = nk - 1;            // largest i with ks[i] <= k, capped at nk-2
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (ks[mid] <= k) lo = mid; else hi = mid;
            }
            const int i = lo;
            const double tx = (k - ks[i]) / (ks[i + 1] - ks[i]);
            double val;
            // P2 (optional) is added on the fly: C_ell of P_1h + P_2h without materialising the sum
            auto at = [&](size_t o) { return P2 ? P[o] + P2[o] : P[o]; };
            if (nz == 1) {
                val = (1.0 - tx) * at(i) + tx * at(i + 1);
            } else {
                double z = fmin(fmax(gzs[g], zs[0]), zs[nz - 1]);
                int jl = 0, jh = nz - 1;
                while (jh - jl > 1) {
                    const int mid = (jl + jh) >> 1;
                    if (zs[mid] <= z) jl = mid; else jh = mid;
                }
                const int j = jl;
                const double ty = (z - zs[j]) / (zs[j + 1] - zs[j]);
                const size_t r0 = (size_t)j * nk + i, r1 = r0 + nk;
                val = (1.0 - tx) * (1.0 - ty) * at(r0) + tx * (1.0 - ty) * at(r0 + 1) +
                      (1.0 - tx) * ty * at(r1) + tx * ty * at(r1 + 1);
            }
            t = wz[g] * (val * pref[g]);
        }
        term[el][gs] = t;
        __syncthreads();
        if (gs == 0) {
            const int n = ngz - g0 < LIMBER_G ? ngz - g0 : LIMBER_G;
            for (int q = 0; q < n; ++q) acc += term[el][q];
        }
        __syncthreads();
    }
    if (gs == 0 && e < nells) out[e] = acc;
}

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

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
using namespace hmg;

static inline dim3 grid1d(size_t n, int block) { return dim3((unsigned)((n + block - 1) / block)); }

// (definitions below inherit C linkage from the declarations in hmgrid.h)

#ifdef HMG_FR_STAMP
extern "C" int hmg_debug_stamps_fused(long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(hmg::g_fstamps), (size_t)n * sizeof(long long));
}
#endif
int hmg_abi_version(void) { return HMG_ABI_VERSION; }
const char* hmg_last_error(void) { return g_last_error.c_str(); }

static int ctx_init(hmg_ctx* c, int device) {
    c->device = device;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    c->num_cu = prop.multiProcessorCount;
    {
        // lane 0 carries the short dependent kernels (mass function, HOD, spectra): give it the
        // highest priority so its workgroups are picked first whenever a slot frees up while a
        // long kernel of another lane is draining
        int lo = 0, hi = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));   // hi = numerically smallest = highest
        for (int i = 0; i < HMG_LANES; ++i)
            HIP_TRY(hipStreamCreateWithPriority(&c->lanes[i], hipStreamNonBlocking, i == 0 ? hi : lo));
    }
    c->stream = c->lanes[0];
    {
        const hmg::SiciTable t = hmg::sici_table_host();
        HIP_TRY(hipMalloc((void**)&c->d_sici, sizeof(t)));
        HIP_TRY(hipMemcpy(c->d_sici, &t, sizeof(t), hipMemcpyHostToDevice));
    }
    if (const char* s = getenv("HMG_FUSED_FFT")) c->use_fused_fft = atoi(s);
    if (const char* s = getenv("HMG_PRUNED_FFT")) c->use_pruned_fft = atoi(s);
    if (const char* s = getenv("HMG_FUSED_MAX_M")) c->fused_max_m = atoi(s);
    if (const char* s = getenv("HMG_FUSED_PREFER_M")) c->fused_prefer_m = atoi(s);
    if (const char* s = getenv("HMG_PRUNED_LP_MIN")) c->pruned_lp_min = atoi(s);
    if (const char* s = getenv("HMG_CHIRP")) c->use_chirp = atoi(s);
    if (const char* s = getenv("HMG_BAND_FFT")) c->use_band_fft = atoi(s);
    if (getenv("HMG_FUSED_GENERIC")) c->fused_generic = 1;
    if (const char* s = getenv("HMG_FORCE_GATHERV")) c->force_gatherv = atoi(s);
    HIP_TRY(hipHostMalloc((void**)&c->h_fault, 64, hipHostMallocMapped | hipHostMallocCoherent));
    *c->h_fault = 0;
    HIP_TRY(hipHostGetDevicePointer((void**)&c->d_fault, c->h_fault, 0));
    if (const char* s = getenv("HMG_FFT_CHUNK_MB")) c->fft_chunk_bytes = (size_t)atol(s) << 20;
    return 0;
}

int hmg_ctx_create(int device, hmg_ctx** out) {
    REQUIRE(out != nullptr, "out is NULL");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    REQUIRE(ndev > 0, "no HIP device visible");
    REQUIRE(device >= 0 && device < ndev, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    if (rocfft_refcount == 0) FFT_TRY(rocfft_setup());
    ++rocfft_refcount;
    hmg_ctx* c = new hmg_ctx();
    if (ctx_init(c, device)) {          // a failed set-up must not leak the half-built context
        const std::string keep = g_last_error;
        hmg_ctx_destroy(c);
        g_last_error = keep;
        return 1;
    }
    *out = c;
    return 0;
}

int hmg_ctx_destroy(hmg_ctx* c) {
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    for (auto& st : c->lanes) if (st) (void)hipStreamSynchronize(st);
    if (c->comm) { ncclCommDestroy(c->comm); c->comm = nullptr; }
    for (auto& kv : c->graphs) (void)hipGraphExecDestroy(kv.second);
    for (auto& kv : c->plans) {
        if (kv.second.info) rocfft_execution_info_destroy(kv.second.info);
        if (kv.second.plan) rocfft_plan_destroy(kv.second.plan);
        if (kv.second.work) (void)hipFree(kv.second.work);
    }
    for (auto& kv : c->fused) {
        if (kv.second.twM) (void)hipFree(kv.second.twM);
        if (kv.second.twN) (void)hipFree(kv.second.twN);
    }
    for (auto& kv : c->chirp) {
        if (kv.second.chP) (void)hipFree(kv.second.chP);
        if (kv.second.chJ) (void)hipFree(kv.second.chJ);
        if (kv.second.Bw) (void)hipFree(kv.second.Bw);
    }
    for (auto& kv : c->pruned) {
        if (kv.second.twB) (void)hipFree(kv.second.twB);
        if (kv.second.twN) (void)hipFree(kv.second.twN);
        if (kv.second.twR) (void)hipFree(kv.second.twR);
        if (kv.second.twNr) (void)hipFree(kv.second.twNr);
    }
    for (auto& kv : c->pass_tw) (void)hipFree(kv.second);
    if (c->h_fault) (void)hipHostFree(c->h_fault);
    for (auto& s : c->scratch) if (s) (void)hipFree(s);
    for (auto& kv : c->free_blocks) (void)hipFree(kv.second);
    for (auto& kv : c->graph_blocks)
        for (void* p : kv.second) (void)hipFree(p);
    for (void* p : c->freed_in_capture) (void)hipFree(p);
    if (c->d_barrier) (void)hipFree(c->d_barrier);
    if (c->d_sici) (void)hipFree(c->d_sici);
    if (c->up_ring) {
        (void)hipHostFree(c->up_ring);
        for (auto& e : c->up_ev) if (e) (void)hipEventDestroy(e);
    }
    for (int i = 0; i < 2; ++i) {
        if (c->pinned[i]) (void)hipHostFree(c->pinned[i]);
        if (c->pin_ev[i]) (void)hipEventDestroy(c->pin_ev[i]);
    }
    for (auto& e : c->ev) if (e) (void)hipEventDestroy(e);
    for (auto& st : c->lanes) if (st) (void)hipStreamDestroy(st);
    if (--rocfft_refcount == 0) rocfft_cleanup();
    delete c;
    return 0;
}

// Device blocks are recycled by size.  Every launch of the library is stream-ordered on lane 0
// unless the caller moved work to another lane (hmg_lane_set), so a block handed back by the host
// may be reused by later lane-0 work without a device synchronisation: whatever still reads or
// writes it was enqueued earlier on the same stream.  If other lanes have been used since the last
// synchronisation, hmg_free synchronises first, as it always used to.
int hmg_malloc(hmg_ctx* c, size_t bytes, void** d_out) {
    REQUIRE(c && d_out, "NULL argument");
    if (!bytes) bytes = 8;
    // No allocation at all inside a captured step, not even out of the free list: the address would be baked
    // into the graph, the block would go back to the list when its owner dies and be handed to somebody else,
    // and every later replay would write into memory it no longer owns.
    REQUIRE(!c->capturing, "device allocation inside a captured step: run the step once eagerly first");
    auto it = c->free_blocks.find(bytes);
    if (it != c->free_blocks.end()) {
        *d_out = it->second;
        c->cached_bytes -= bytes;
        c->free_blocks.erase(it);
        return 0;
    }
    HIP_TRY(hipSetDevice(c->device));
    hipError_t e = hipMalloc(d_out, bytes);
    if (e != hipSuccess && !c->free_blocks.empty()) {      // give the cache back and retry once
        if (sync_all(c)) return 1;
        for (auto& kv : c->free_blocks) { (void)hipFree(kv.second); c->block_bytes.erase(kv.second); }
        c->free_blocks.clear();
        c->cached_bytes = 0;
        e = hipMalloc(d_out, bytes);
    }
    HIP_TRY(e);
    c->block_bytes[*d_out] = bytes;
    return 0;
}
int hmg_free(hmg_ctx* c, void* p) {
    REQUIRE(c, "NULL ctx");
    if (!p) return 0;
    auto it = c->block_bytes.find(p);
    REQUIRE(it != c->block_bytes.end(), "pointer was not allocated by hmg_malloc of this context");
    if (c->capturing) {      // deferred, not dropped: the block joins the free list when the capture ends
        c->freed_in_capture.push_back(p);
        return 0;
    }
    if (c->lanes_dirty && sync_all(c)) return 1;
    const size_t bytes = it->second;
    if (c->cached_bytes + bytes <= FREE_CACHE_LIMIT) {
        c->free_blocks.emplace(bytes, p);
        c->cached_bytes += bytes;
        return 0;
    }
    if (sync_all(c)) return 1;
    HIP_TRY(hipFree(p));
    c->block_bytes.erase(it);
    return 0;
}
int hmg_host_alloc(hmg_ctx* c, size_t bytes, void** h_out) {
    REQUIRE(c && h_out, "NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipHostMalloc(h_out, bytes ? bytes : 8, hipHostMallocDefault));
    return 0;
}
int hmg_host_free(hmg_ctx* c, void* h) {
    REQUIRE(c, "NULL ctx");
    if (h) HIP_TRY(hipHostFree(h));
    return 0;
}
int hmg_memcpy_d2h_async(hmg_ctx* c, void* h_pinned, const void* d, size_t bytes) {
    REQUIRE(c && h_pinned && d, "NULL argument");
    HIP_TRY(hipMemcpyAsync(h_pinned, d, bytes, hipMemcpyDeviceToHost, c->stream));
    return 0;
}
int hmg_memcpy_h2d_async(hmg_ctx* c, void* d, const void* h_pinned, size_t bytes) {
    REQUIRE(c && h_pinned && d, "NULL argument");
    HIP_TRY(hipMemcpyAsync(d, h_pinned, bytes, hipMemcpyHostToDevice, c->stream));
    return 0;
}
int hmg_event_synchronize(hmg_ctx* c, int slot) {
    REQUIRE(c && slot >= 0 && slot < HMG_EVENT_SLOTS, "bad event slot");
    REQUIRE(!c->capturing, "hmg_event_synchronize inside a captured step");
    REQUIRE(c->ev[slot] != nullptr, "event slot was never recorded");
    HIP_TRY(hipEventSynchronize(c->ev[slot]));
    return check_fault(c);      // (the streamed hand-over of results waits here and nowhere else)
}
constexpr size_t PIN_CHUNK = (size_t)8 << 20;   // 8 MiB per bounce buffer

static int ensure_pinned(hmg_ctx* c) {
    for (int i = 0; i < 2; ++i) {
        if (!c->pinned[i]) HIP_TRY(hipHostMalloc(&c->pinned[i], PIN_CHUNK, hipHostMallocDefault));
        if (!c->pin_ev[i]) HIP_TRY(hipEventCreateWithFlags(&c->pin_ev[i], hipEventDisableTiming));
    }
    return 0;
}

// Pageable host memory moves at ~3 GB/s through the runtime's own staging; bouncing through two
// pinned 8 MiB buffers (DMA of chunk i+1 overlapped with the host memcpy of chunk i) is 5-8x faster.
int hmg_memcpy_h2d(hmg_ctx* c, void* d, const void* h, size_t bytes) {
    REQUIRE(c && d && h, "NULL argument");
    if (bytes <= hmg_ctx::UP_SLOT_BYTES && !c->capturing && !c->lanes_dirty && c->stream == c->lanes[0]) {
        // the caller's array is copied into a pinned slot now, the DMA out of the slot is stream-ordered: no
        // host wait (a slot is reused only after its own DMA has finished).  Only while everything runs on
        // lane 0: with other lanes in play the consumer may sit on another stream, and the synchronous path
        // below is what orders it
        if (!c->up_ring) {
            HIP_TRY(hipHostMalloc((void**)&c->up_ring, hmg_ctx::UP_SLOTS * hmg_ctx::UP_SLOT_BYTES, hipHostMallocDefault));
            for (int i = 0; i < hmg_ctx::UP_SLOTS; ++i)
                HIP_TRY(hipEventCreateWithFlags(&c->up_ev[i], hipEventDisableTiming));
        }
        const int slot = c->up_next;
        c->up_next = (c->up_next + 1) % hmg_ctx::UP_SLOTS;
        HIP_TRY(hipEventSynchronize(c->up_ev[slot]));
        char* stage = c->up_ring + (size_t)slot * hmg_ctx::UP_SLOT_BYTES;
        memcpy(stage, h, bytes);
        HIP_TRY(hipMemcpyAsync(d, stage, bytes, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipEventRecord(c->up_ev[slot], c->stream));
        return 0;
    }
    if (bytes < (256u << 10)) {
        HIP_TRY(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return 0;
    }
    if (ensure_pinned(c)) return 1;
    size_t done = 0;
    int b = 0;
    while (done < bytes) {
        const size_t n = bytes - done < PIN_CHUNK ? bytes - done : PIN_CHUNK;
        HIP_TRY(hipEventSynchronize(c->pin_ev[b]));          // previous DMA out of this buffer finished
        memcpy(c->pinned[b], (const char*)h + done, n);
        HIP_TRY(hipMemcpyAsync((char*)d + done, c->pinned[b], n, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipEventRecord(c->pin_ev[b], c->stream));
        done += n;
        b ^= 1;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}
int hmg_memcpy_d2h(hmg_ctx* c, void* h, const void* d, size_t bytes) {
    REQUIRE(c && d && h, "NULL argument");
    REQUIRE(!c->capturing, "hmg_memcpy_d2h inside a captured step");
    // the copy runs on the current lane, behind everything enqueued there; only when other lanes
    // have been used can the producer sit elsewhere
    if (c->lanes_dirty && sync_all(c)) return 1;
    if (bytes < (256u << 10)) {
        HIP_TRY(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return check_fault(c);
    }
    if (ensure_pinned(c)) return 1;
    // software pipeline: DMA chunk i+1 into the other buffer while chunk i is copied out
    size_t issued = 0, copied = 0;
    size_t len[2] = {0, 0};
    int bi = 0, bo = 0;
    while (copied < bytes) {
        while (issued < bytes && issued - copied < 2 * PIN_CHUNK && len[bi] == 0) {
            const size_t n = bytes - issued < PIN_CHUNK ? bytes - issued : PIN_CHUNK;
            HIP_TRY(hipMemcpyAsync(c->pinned[bi], (const char*)d + issued, n, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipEventRecord(c->pin_ev[bi], c->stream));
            len[bi] = n;
            issued += n;
            bi ^= 1;
        }
        HIP_TRY(hipEventSynchronize(c->pin_ev[bo]));
        memcpy((char*)h + copied, c->pinned[bo], len[bo]);
        copied += len[bo];
        len[bo] = 0;
        bo ^= 1;
    }
    return check_fault(c);
}
int hmg_memcpy_d2d(hmg_ctx* c, void* dst, const void* src, size_t bytes) {
    REQUIRE(c && dst && src, "NULL argument");
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->stream));
    return 0;
}
int hmg_sync(hmg_ctx* c) {
    REQUIRE(c, "NULL ctx");
    return sync_all(c);
}
int hmg_lane_set(hmg_ctx* c, int lane) {
    REQUIRE(c && lane >= 0 && lane < HMG_LANES, "bad lane");
    if (lane != 0) c->lanes_dirty = true;
    c->lane = lane;
    c->stream = c->lanes[lane];
    return 0;
}
int hmg_event_wait(hmg_ctx* c, int slot) {
    REQUIRE(c && slot >= 0 && slot < HMG_EVENT_SLOTS, "bad event slot");
    if (!c->ev[slot]) return 0;   // never recorded: nothing to wait for
    HIP_TRY(hipStreamWaitEvent(c->stream, c->ev[slot], 0));
    return 0;
}
int hmg_event_record(hmg_ctx* c, int slot) {
    REQUIRE(c && slot >= 0 && slot < HMG_EVENT_SLOTS, "bad event slot");
    hipEvent_t e;
    if (event_at(c, slot, &e)) return 1;
    HIP_TRY(hipEventRecord(e, c->stream));
    return 0;
}
int hmg_elapsed_ms(hmg_ctx* c, int s0, int s1, double* ms) {
    REQUIRE(c && ms && s0 >= 0 && s0 < HMG_EVENT_SLOTS && s1 >= 0 && s1 < HMG_EVENT_SLOTS, "bad event slot");
    REQUIRE(c->ev[s0] && c->ev[s1], "event slot was never recorded");
    HIP_TRY(hipEventSynchronize(c->ev[s1]));
    float f = 0.f;
    HIP_TRY(hipEventElapsedTime(&f, c->ev[s0], c->ev[s1]));
    *ms = (double)f;
    return 0;
}

// ---- captured steps ----------------------------------------------------------------------------
// Everything enqueued between hmg_graph_begin and hmg_graph_end (on lane 0 and on any lane that joins
// through hmg_event_wait on an event recorded inside the capture) becomes one HIP graph: a pass of
// the path is then ONE host call instead of ~15 launches, and independent branches (the sigma^2 ->
// n(z,m) -> HOD chain beside the two profile kernels) run concurrently.  Nothing that allocates,
// frees or synchronises may happen in between: run the same sequence once eagerly first, so that
// scratch arenas, FFT tables and output buffers exist.
int hmg_graph_begin(hmg_ctx* c) {
    REQUIRE(c, "NULL ctx");
    REQUIRE(!c->capturing, "already capturing");
    REQUIRE(c->lane == 0, "start a capture on lane 0");
    HIP_TRY(hipStreamBeginCapture(c->lanes[0], hipStreamCaptureModeRelaxed));
    c->capturing = true;
    return 0;
}
static void release_deferred_frees(hmg_ctx* c) {
    std::vector<void*> v;
    v.swap(c->freed_in_capture);
    for (void* p : v) (void)hmg_free(c, p);
}
int hmg_graph_end(hmg_ctx* c, int* id) {
    REQUIRE(c && id, "NULL argument");
    REQUIRE(c->capturing, "no capture in progress");
    c->capturing = false;
    c->lane = 0;
    c->stream = c->lanes[0];
    hipGraph_t g = nullptr;
    HIP_TRY(hipStreamEndCapture(c->lanes[0], &g));
    // how many kernel launches the captured step holds (bench.py reports it as launches_per_step)
    int nkern = 0;
    {
        size_t nn = 0;
        if (hipGraphGetNodes(g, nullptr, &nn) == hipSuccess && nn) {
            std::vector<hipGraphNode_t> nodes(nn);
            if (hipGraphGetNodes(g, nodes.data(), &nn) == hipSuccess)
                for (size_t i = 0; i < nn; ++i) {
                    hipGraphNodeType t;
                    if (hipGraphNodeGetType(nodes[i], &t) == hipSuccess && t == hipGraphNodeTypeKernel) ++nkern;
                }
        }
    }
    hipGraphExec_t ge = nullptr;
    hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) release_deferred_frees(c);       // no graph: nothing can refer to them any more
    HIP_TRY(e);
    *id = c->next_graph_id++;
    c->graphs[*id] = ge;
    c->graph_kernels[*id] = nkern;
    // A block handed back while the capture ran was allocated before it (allocation inside a capture is
    // refused) and may be an operand of a captured launch: it stays out of the free list as long as the
    // graph can be replayed.
    c->graph_blocks[*id].swap(c->freed_in_capture);
    return 0;
}
int hmg_graph_kernel_nodes(hmg_ctx* c, int id, int* n) {
    REQUIRE(c && n, "NULL argument");
    auto it = c->graph_kernels.find(id);
    REQUIRE(it != c->graph_kernels.end(), "unknown graph id");
    *n = it->second;
    return 0;
}
int hmg_graph_abort(hmg_ctx* c) {      // leave capture mode after a failed call inside a capture
    REQUIRE(c, "NULL ctx");
    if (!c->capturing) return 0;
    c->capturing = false;
    release_deferred_frees(c);
    c->lane = 0;
    c->stream = c->lanes[0];
    hipGraph_t g = nullptr;
    (void)hipStreamEndCapture(c->lanes[0], &g);
    if (g) (void)hipGraphDestroy(g);
    (void)hipGetLastError();
    return 0;
}
int hmg_graph_launch(hmg_ctx* c, int id) {
    REQUIRE(c, "NULL ctx");
    REQUIRE(!c->capturing, "cannot replay a graph inside a capture");
    auto it = c->graphs.find(id);
    REQUIRE(it != c->graphs.end(), "unknown graph id");
    HIP_TRY(hipGraphLaunch(it->second, c->stream));
    return 0;
}
int hmg_graph_destroy(hmg_ctx* c, int id) {
    REQUIRE(c, "NULL ctx");
    auto it = c->graphs.find(id);
    if (it == c->graphs.end()) return 0;
    if (sync_all(c)) return 1;
    HIP_TRY(hipGraphExecDestroy(it->second));
    c->graphs.erase(it);
    auto gb = c->graph_blocks.find(id);
    if (gb != c->graph_blocks.end()) {
        for (void* p : gb->second) (void)hmg_free(c, p);
        c->graph_blocks.erase(gb);
    }
    return 0;
}

int hmg_bracket_next(hmg_ctx* c, int kernel_id, int s0, int s1) {
    REQUIRE(c && kernel_id >= 0 && kernel_id < HMG_KERNEL_COUNT, "bad kernel id");
    REQUIRE(s0 >= -1 && s0 < HMG_EVENT_SLOTS && s1 >= -1 && s1 < HMG_EVENT_SLOTS, "bad event slot");
    c->bracket[kernel_id][0] = s0;
    c->bracket[kernel_id][1] = s1;
    return 0;
}

// RAII-free bracket: record start now, return the stop slot (or -1) and clear the one-shot.
static int bracket_open(hmg_ctx* c, int kid, int* stop_slot) {
    *stop_slot = -1;
    const int s0 = c->bracket[kid][0], s1 = c->bracket[kid][1];
    c->bracket[kid][0] = c->bracket[kid][1] = -1;
    if (s0 >= 0) {
        hipEvent_t e;
        if (event_at(c, s0, &e)) return 1;
        HIP_TRY(hipEventRecord(e, c->stream));
    }
    *stop_slot = s1;
    return 0;
}
static int bracket_close(hmg_ctx* c, int stop_slot) {
    if (stop_slot >= 0) {
        hipEvent_t e;
        if (event_at(c, stop_slot, &e)) return 1;
        HIP_TRY(hipEventRecord(e, c->stream));
    }
    return 0;
}

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

// profile_group_kernel = the fused row kernel with `nchain` per-z chain workgroups in front of the rows
template <int MAXB, int MAXP, int SPECM = 0>
static int launch_fused_group(hmg_ctx* c, const FusedArgs& A, int rows, const ChainArgs& C, int nchain, size_t chain_lds) {
    size_t lds = (size_t)A.plan.M * 16 + 32 * sizeof(double);
    if (chain_lds > lds) lds = chain_lds;
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
static int profile_fft_impl(hmg_ctx* c, int nz, int nm, int nk, const hmg_profile_fft_part& p, const ChainArgs* C,
                            int nchain, size_t chain_lds, int* chain_done, const double* rho_tab = nullptr,
                            int rho_shared = 0, int* taken = nullptr) {
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
            const bool spec2500 = FUSED_NT == 512 && pl.M == 2500 && pl.npass == 5 && pl.radix[0] == 4 && pl.radix[1] == 5 &&
                                  pl.radix[2] == 5 && pl.radix[3] == 5 && pl.radix[4] == 5 &&
                                  !c->fused_generic;      // (testing: force the run-time plan)
            const bool grouped = C && nchain > 0 && FUSED_NT == 512;
            // lengths with a compile-time plan (fused_passes_ct): nxs = 1000, 2000, 3000, 4000, 6000 (the last one only
            // when its rows' support does not let the long-grid route take it)
            const int ctM = (FUSED_NT == 512 && !c->fused_generic &&
                             (pl.M == 500 || pl.M == 1000 || pl.M == 1500 || pl.M == 2000 || pl.M == 3000)) ? pl.M : 0;
            if (grouped) {
                if (spec2500) rc = launch_fused_group<2, 3, 2500>(c, A, rows, *C, nchain, chain_lds);
                else if (ctM == 500) rc = launch_fused_group<1, 1, 500>(c, A, rows, *C, nchain, chain_lds);
                else if (ctM == 1000) rc = launch_fused_group<1, 1, 1000>(c, A, rows, *C, nchain, chain_lds);
                else if (ctM == 1500) rc = launch_fused_group<1, 2, 1500>(c, A, rows, *C, nchain, chain_lds);
                else if (ctM == 2000) rc = launch_fused_group<2, 2, 2000>(c, A, rows, *C, nchain, chain_lds);
                else if (ctM == 3000) rc = launch_fused_group<3, 3, 3000>(c, A, rows, *C, nchain, chain_lds);
                else if (mb <= 1 && mp <= 2) rc = launch_fused_group<1, 2>(c, A, rows, *C, nchain, chain_lds);
                else if (mb <= 2 && mp <= 3) rc = launch_fused_group<2, 3>(c, A, rows, *C, nchain, chain_lds);
                else if (mb <= 2 && mp <= 4) rc = launch_fused_group<2, 4>(c, A, rows, *C, nchain, chain_lds);
                else rc = launch_fused_group<4, 8>(c, A, rows, *C, nchain, chain_lds);
                if (!rc && chain_done) *chain_done = 1;
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
    C->has_hod = C->has_prep = 0;
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
    int chain_done = 0;
    if (fft && profile_fft_impl(c, nz, nm, nk, *fft, &C, one ? nz : 0, chain_lds_doubles(nm) * 8, &chain_done)) return 1;
    if (one && !chain_done) {      // no rows to share a launch with (or a length the in-LDS transform does not take)
        const RowsArgs none{};
        if (launch_rows_group(c, nz, nm, C, nz, nullptr, none, nullptr, 0)) return 1;
    }
    if (prep && !P.code) {         // generic coefficient rows (register-hungry): a launch of their own
        hipLaunchKernelGGL(power_batch_prep_kernel, dim3(nz, P.PA.nblk), dim3(64), 0, c->stream, P.PA);
        HIP_TRY(hipGetLastError());
    }
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

// ---- RCCL ------------------------------------------------------------------------------------
int hmg_comm_unique_id(char id[HMG_COMM_ID_BYTES]) {
    REQUIRE(id, "NULL id");
    static_assert(sizeof(ncclUniqueId) <= HMG_COMM_ID_BYTES, "id buffer too small");
    ncclUniqueId u;
    NCCL_TRY(ncclGetUniqueId(&u));
    memset(id, 0, HMG_COMM_ID_BYTES);
    memcpy(id, &u, sizeof(u));
    return 0;
}
int hmg_comm_init(hmg_ctx* c, const char id[HMG_COMM_ID_BYTES], int rank, int nranks) {
    REQUIRE(c && id, "NULL argument");
    REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "bad rank/nranks");
    REQUIRE(!c->comm, "communicator already initialised");
    HIP_TRY(hipSetDevice(c->device));
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    NCCL_TRY(ncclCommInitRank(&c->comm, nranks, u, rank));
    c->comm_rank = rank;
    c->comm_size = nranks;
    HIP_TRY(hipMalloc((void**)&c->d_barrier, 64));
    HIP_TRY(hipMemset(c->d_barrier, 0, 64));
    return 0;
}
int hmg_comm_allgather(hmg_ctx* c, const double* send, double* recv, size_t count) {
    REQUIRE(c && send && recv, "NULL argument");
    if (!c->comm) {  // single rank without a communicator: plain copy
        if (send != recv) HIP_TRY(hipMemcpyAsync(recv, send, count * 8, hipMemcpyDeviceToDevice, c->stream));
        return 0;
    }
    NCCL_TRY(ncclAllGather(send, recv, count, ncclDouble, c->comm, c->stream));
    return 0;
}
int hmg_comm_allgather_multi(hmg_ctx* c, int n, const double* const* send, double* const* recv,
                             size_t count) {
    REQUIRE(c && send && recv && n >= 0, "bad argument");
    if (!c->comm) {
        for (int i = 0; i < n; ++i)
            if (send[i] != recv[i])
                HIP_TRY(hipMemcpyAsync(recv[i], send[i], count * 8, hipMemcpyDeviceToDevice, c->stream));
        return 0;
    }
    NCCL_TRY(ncclGroupStart());
    for (int i = 0; i < n; ++i) {
        ncclResult_t r = ncclAllGather(send[i], recv[i], count, ncclDouble, c->comm, c->stream);
        if (r != ncclSuccess) {
            ncclGroupEnd();
            return fail("ncclAllGather", ncclGetErrorString(r), __FILE__, __LINE__);
        }
    }
    NCCL_TRY(ncclGroupEnd());
    return 0;
}
// Slabs of unequal length (nz not a multiple of the number of ranks, e.g. the README grid's nz = 20 on 8 GPUs):
// rank r contributes h_counts[r] doubles per array, landing at the prefix-sum offset - one ncclBroadcast per
// (array, rank), all in ONE group launch, so that every slab still arrives in its final position with no
// padding and no compaction pass.  Equal counts take the all-gather.
static int comm_gatherv_multi(hmg_ctx* c, int n, const double* const* send, double* const* recv, const size_t* counts) {
    const int nr = c->comm ? c->comm_size : 1, me = c->comm ? c->comm_rank : 0;
    bool equal = true;
    for (int r = 1; r < nr; ++r) equal = equal && counts[r] == counts[0];
    // (HMG_FORCE_GATHERV=1, testing: equal counts take the per-rank branch too, so that a one-rank communicator on a
    // one-GPU box runs the grouped broadcasts - root out of place - that only unequal slabs on several GPUs reach)
    if (equal && !(c->force_gatherv && c->comm)) return hmg_comm_allgather_multi(c, n, send, recv, counts[0]);
    NCCL_TRY(ncclGroupStart());
    for (int i = 0; i < n; ++i) {
        size_t off = 0;
        for (int r = 0; r < nr; ++r) {
            if (counts[r]) {
                ncclResult_t e = ncclBroadcast(r == me ? send[i] : recv[i] + off, recv[i] + off, counts[r], ncclDouble, r,
                                               c->comm, c->stream);
                if (e != ncclSuccess) {
                    ncclGroupEnd();
                    return fail("ncclBroadcast", ncclGetErrorString(e), __FILE__, __LINE__);
                }
            }
            off += counts[r];
        }
    }
    NCCL_TRY(ncclGroupEnd());
    return 0;
}
int hmg_comm_allgatherv_multi(hmg_ctx* c, int n, const double* const* send, double* const* recv, const size_t* counts) {
    REQUIRE(c && send && recv && counts && n >= 0, "bad argument");
    return comm_gatherv_multi(c, n, send, recv, counts);
}
// The z-slab gather of one pass, off the compute stream: an event marks "spectra ready" on the
// current lane, the communication lane waits for it, issues the grouped all-gather and records
// done_slot.  The next pass calls hmg_event_wait(done_slot) before it overwrites the local spectra,
// so the collective overlaps the next pass's first kernels instead of extending the step.
static int comm_gather_async(hmg_ctx* c, int n, const double* const* send, double* const* recv, size_t count,
                             const size_t* counts, int ready_slot, int done_slot, int comm_lane) {
    REQUIRE(c && send && recv && n >= 0, "bad argument");
    REQUIRE(!c->capturing, "the gather is issued outside captured steps");
    REQUIRE(ready_slot >= 0 && ready_slot < HMG_EVENT_SLOTS && done_slot >= 0 && done_slot < HMG_EVENT_SLOTS, "bad event slot");
    REQUIRE(comm_lane > 0 && comm_lane < HMG_LANES, "bad communication lane");
    hipEvent_t ready, done;
    if (event_at(c, ready_slot, &ready) || event_at(c, done_slot, &done)) return 1;
    HIP_TRY(hipEventRecord(ready, c->stream));
    hipStream_t keep = c->stream;
    c->stream = c->lanes[comm_lane];
    c->lanes_dirty = true;
    HIP_TRY(hipStreamWaitEvent(c->stream, ready, 0));
    const int rc = counts ? comm_gatherv_multi(c, n, send, recv, counts) : hmg_comm_allgather_multi(c, n, send, recv, count);
    if (!rc) {
        hipError_t e = hipEventRecord(done, c->stream);
        c->stream = keep;
        HIP_TRY(e);
    }
    c->stream = keep;
    return rc;
}
int hmg_comm_gather_async(hmg_ctx* c, int n, const double* const* send, double* const* recv, size_t count,
                          int ready_slot, int done_slot, int comm_lane) {
    return comm_gather_async(c, n, send, recv, count, nullptr, ready_slot, done_slot, comm_lane);
}
int hmg_comm_gatherv_async(hmg_ctx* c, int n, const double* const* send, double* const* recv, const size_t* counts,
                           int ready_slot, int done_slot, int comm_lane) {
    REQUIRE(counts, "NULL counts");
    return comm_gather_async(c, n, send, recv, 0, counts, ready_slot, done_slot, comm_lane);
}
int hmg_comm_info(hmg_ctx* c, int* rank, int* nranks) {
    REQUIRE(c && rank && nranks, "NULL argument");
    *rank = 0;
    *nranks = 1;
    if (c->comm) {      // ask RCCL, not our own bookkeeping: this is what the record of a run quotes
        NCCL_TRY(ncclCommCount(c->comm, nranks));
        NCCL_TRY(ncclCommUserRank(c->comm, rank));
    }
    return 0;
}
int hmg_comm_barrier(hmg_ctx* c) {
    REQUIRE(c, "NULL ctx");
    if (sync_all(c)) return 1;
    if (c->comm) NCCL_TRY(ncclAllReduce(c->d_barrier, c->d_barrier, 1, ncclDouble, ncclSum, c->comm, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}
int hmg_comm_destroy(hmg_ctx* c) {
    REQUIRE(c, "NULL ctx");
    if (c->comm) {
        if (sync_all(c)) return 1;      // collectives may be in flight on the communication lane
        NCCL_TRY(ncclCommDestroy(c->comm));
        c->comm = nullptr;
    }
    if (c->d_barrier) {
        HIP_TRY(hipFree(c->d_barrier));
        c->d_barrier = nullptr;
    }
    return 0;
}
