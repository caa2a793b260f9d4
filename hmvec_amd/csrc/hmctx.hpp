// Internal header of libhmgrid: the context, the error plumbing and the host helpers shared by its translation units -
// runtime.hip (contexts, memory, events, lanes, captured steps), comm.hip (RCCL) and hmgrid.hip (kernels + their launch
// entry points).  Not part of the C ABI (include/hmgrid.h).
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <rocfft/rocfft.h>

#include <cstdio>
#include <map>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/hmgrid.h"
#include "ldsfft.hpp"
#include "sici.hpp"

// ------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------
// (the message of the last failed call of this thread: hmg_last_error; both live in runtime.hip)
extern thread_local std::string g_last_error;
int fail(const char* what, const char* detail, const char* file, int line);
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
    int use_tensor_group = 1;                     // HMG_NO_TENSOR_GROUP=1: hmg_group_tensors as its two launches (testing, A/B)
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

// ---- host helpers defined in runtime.hip
int event_at(hmg_ctx* c, int slot, hipEvent_t* out);      // events are created on first use
int check_fault(hmg_ctx* c);                              // report and clear the context's device fault word
int sync_all(hmg_ctx* c);                                 // all lanes + fault check; refuses inside a capture
int ensure_scratch(hmg_ctx* c, int slot, size_t bytes);   // grow-only scratch arenas; refuses inside a capture
int bracket_open(hmg_ctx* c, int kid, int* stop_slot);    // one-shot event brackets around a kernel (hmg_bracket_next)
int bracket_close(hmg_ctx* c, int stop_slot);
static inline dim3 grid1d(size_t n, int block) { return dim3((unsigned)((n + block - 1) / block)); }
