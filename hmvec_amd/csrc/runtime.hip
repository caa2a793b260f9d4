// libhmgrid runtime: contexts, the device-block cache, pinned staging, events, lanes and captured steps (HIP graphs).
// Host code only - no kernel lives here; the kernels and their launch entry points are in hmgrid.hip, RCCL in comm.hip.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "hmctx.hpp"

using namespace hmg;

thread_local std::string g_last_error;

int fail(const char* what, const char* detail, const char* file, int line) {
    char buf[512];
    snprintf(buf, sizeof(buf), "%s: %s (%s:%d)", what, detail, file, line);
    g_last_error = buf;
    return 1;
}

static int rocfft_refcount = 0;

// events are created on first use (a context rarely needs more than a handful of the slots)
int event_at(hmg_ctx* c, int slot, hipEvent_t* out) {
    if (!c->ev[slot]) HIP_TRY(hipEventCreate(&c->ev[slot]));
    *out = c->ev[slot];
    return 0;
}

// A kernel that finds it cannot do what it was launched for (a row whose support exceeds the plan the launch was
// sized for) raises the context's fault word instead of writing wrong numbers quietly; synchronising calls report it.
int check_fault(hmg_ctx* c) {
    if (!*(volatile int*)c->h_fault) return 0;
    *(volatile int*)c->h_fault = 0;
    c->support.clear();
    return fail("device fault", "a profile row's support exceeded the bound its launch was sized for (the rows were "
                "filled with NaN); the cached bound is dropped - run the step eagerly again", __FILE__, __LINE__);
}
int sync_all(hmg_ctx* c) {
    REQUIRE(!c->capturing, "this call synchronises the device and cannot be part of a captured step");
    for (auto& st : c->lanes) HIP_TRY(hipStreamSynchronize(st));
    c->lanes_dirty = false;
    return check_fault(c);
}

int ensure_scratch(hmg_ctx* c, int slot, size_t bytes) {
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

// (definitions below inherit C linkage from the declarations in hmgrid.h)

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
    if (const char* s = getenv("HMG_NO_TENSOR_GROUP")) c->use_tensor_group = !atoi(s);
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
int bracket_open(hmg_ctx* c, int kid, int* stop_slot) {
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
int bracket_close(hmg_ctx* c, int stop_slot) {
    if (stop_slot >= 0) {
        hipEvent_t e;
        if (event_at(c, stop_slot, &e)) return 1;
        HIP_TRY(hipEventRecord(e, c->stream));
    }
    return 0;
}
