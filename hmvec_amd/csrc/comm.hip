// libhmgrid collectives: the z-slab gather of SURVEY 8(e) over RCCL (one process per GPU; backend = RCCL over xGMI).
// Host code only.
#include <cstring>

#include "hmctx.hpp"

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
