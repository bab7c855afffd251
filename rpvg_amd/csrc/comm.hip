// comm.hip — the communicator of a rank: RCCL over xGMI, one process per GPU.
//
// The reference has no exchange step (one process, OpenMP over clusters, src/main.cpp:829).  Clusters
// shard over GPUs without any data-path collective; the two places a sum over ranks is needed are
//   * one giant cluster whose rows are spread over the GPUs: the C partial column sums of every EM
//     iteration (rpvg_hip_em_dense_sharded, em_dense.hip), C <= 2048 doubles = 16 KB, latency bound;
//   * the TPM denominator, total_transcript_count = sum abundance / effective_length
//     (src/main.cpp:1029-1057): one double.
// Both are ncclAllReduce calls queued on the context's stream, so they are ordered with the kernels
// around them and cost no host synchronisation.  The one collective of the sharded-clusters case is the
// final gather of the per-path abundances (rpvg_hip_gather: ncclAllGather, once per run, latency bound).
// The ranks are one process per GPU (ids handed around by the harness) or the contexts of one process, one
// host thread per GPU (rpvg_hip_comm_init_all; rpvg_amd/host/device_group.hpp).
//
// RCCL is opened with dlopen at the first comm call: processes that never shard a cluster (and the
// CPU-side ABI tests) do not need the library, and a process that already has an RCCL mapped (the
// harness imports torch, which brings its own copy) gets that one instead of a second.

#include <dlfcn.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include <rccl/rccl.h>

#include "common.hpp"

namespace {

struct RcclApi {
    void * handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char * (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

RcclApi & rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char * names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char * name : names) {
            api.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (api.handle) break;
        }
        if (!api.handle) {
            const char * why = dlerror();  // (one call: it clears the error it returns)
            api.error = std::string("cannot open librccl: ") + (why ? why : "unknown");
            return;
        }
        auto sym = [&](const char * n) -> void * {
            void * p = dlsym(api.handle, n);
            if (!p && api.error.empty()) api.error = std::string("librccl lacks ") + n;
            return p;
        };
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
        api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
        api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
        api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
        api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return api;
}

#define RPVG_RCCL_CHECK(call)                                                                         \
    do {                                                                                              \
        const ncclResult_t rc_ = (call);                                                              \
        if (rc_ != ncclSuccess) {                                                                     \
            rpvg_hip_detail::setError("%s failed: %s", #call, rccl().GetErrorString(rc_));            \
            return RPVG_HIP_ERR_RUNTIME;                                                              \
        }                                                                                             \
    } while (0)

int requireRccl() {
    RcclApi & api = rccl();
    if (!api.error.empty() || !api.handle) {
        rpvg_hip_detail::setError("RCCL unavailable: %s", api.error.c_str());
        return RPVG_HIP_ERR_RUNTIME;
    }
    return RPVG_HIP_OK;
}

static_assert(sizeof(ncclUniqueId) == RPVG_HIP_COMM_ID_BYTES, "RPVG_HIP_COMM_ID_BYTES must equal sizeof(ncclUniqueId)");

}  // namespace

int rpvg_hip_ctx::allReduceSumF64(double * device_buf, uint64_t n) {
    if (!comm) {
        rpvg_hip_detail::setError("no communicator on this context (rpvg_hip_comm_init first)");
        return RPVG_HIP_ERR_INVALID;
    }
    RPVG_RCCL_CHECK(rccl().AllReduce(device_buf, device_buf, n, ncclDouble, ncclSum, static_cast<ncclComm_t>(comm), stream));
    return RPVG_HIP_OK;
}

extern "C" int rpvg_hip_comm_unique_id(uint8_t * id_out) {
    RPVG_REQUIRE(id_out, "rpvg_hip_comm_unique_id: NULL argument");
    if (const int rc = requireRccl()) return rc;
    ncclUniqueId id;
    RPVG_RCCL_CHECK(rccl().GetUniqueId(&id));
    std::memcpy(id_out, &id, sizeof(id));
    return RPVG_HIP_OK;
}

extern "C" int rpvg_hip_comm_init(rpvg_hip_ctx * ctx, const uint8_t * id, int world_size, int rank) {
    RPVG_REQUIRE(ctx && id, "rpvg_hip_comm_init: NULL argument");
    RPVG_REQUIRE(world_size >= 1 && rank >= 0 && rank < world_size, "rpvg_hip_comm_init: rank %d of %d", rank, world_size);
    RPVG_REQUIRE(!ctx->comm, "rpvg_hip_comm_init: the context already has a communicator");
    if (const int rc = requireRccl()) return rc;
    std::lock_guard<std::mutex> lock(ctx->mutex);
    RPVG_HIP_CHECK(hipSetDevice(ctx->device));
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    ncclComm_t comm = nullptr;
    RPVG_RCCL_CHECK(rccl().CommInitRank(&comm, world_size, uid, rank));
    ctx->comm = comm;
    ctx->comm_world = world_size;
    ctx->comm_rank = rank;
    return RPVG_HIP_OK;
}

extern "C" int rpvg_hip_comm_destroy(rpvg_hip_ctx * ctx) {
    RPVG_REQUIRE(ctx, "rpvg_hip_comm_destroy: NULL argument");
    if (!ctx->comm) return RPVG_HIP_OK;
    std::lock_guard<std::mutex> lock(ctx->mutex);
    RPVG_HIP_CHECK(hipSetDevice(ctx->device));
    RPVG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    RPVG_RCCL_CHECK(rccl().CommDestroy(static_cast<ncclComm_t>(ctx->comm)));
    ctx->comm = nullptr;
    ctx->comm_world = 1;
    ctx->comm_rank = 0;
    return RPVG_HIP_OK;
}

extern "C" int rpvg_hip_comm_allreduce_sum_f64(rpvg_hip_ctx * ctx, double * device_buf, uint64_t n) {
    RPVG_REQUIRE(ctx && device_buf, "rpvg_hip_comm_allreduce_sum_f64: NULL argument");
    std::lock_guard<std::mutex> lock(ctx->mutex);
    RPVG_HIP_CHECK(hipSetDevice(ctx->device));
    if (const int rc = ctx->allReduceSumF64(device_buf, n)) return rc;
    RPVG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return RPVG_HIP_OK;
}

extern "C" int rpvg_hip_comm_init_all(rpvg_hip_ctx * const * ctxs, int num_contexts) {
    RPVG_REQUIRE(ctxs && num_contexts >= 1, "rpvg_hip_comm_init_all: no contexts");
    for (int i = 0; i < num_contexts; ++i) {
        RPVG_REQUIRE(ctxs[i] && !ctxs[i]->comm, "rpvg_hip_comm_init_all: context %d is NULL or already has a communicator", i);
        for (int j = 0; j < i; ++j) {
            RPVG_REQUIRE(ctxs[i]->device != ctxs[j]->device, "rpvg_hip_comm_init_all: contexts %d and %d share GPU %d (one rank per GPU)", j, i,
                         ctxs[i]->device);
        }
    }
    if (const int rc = requireRccl()) return rc;
    ncclUniqueId uid;
    RPVG_RCCL_CHECK(rccl().GetUniqueId(&uid));
    std::vector<ncclComm_t> comms(num_contexts, nullptr);
    RPVG_RCCL_CHECK(rccl().GroupStart());
    for (int i = 0; i < num_contexts; ++i) {
        RPVG_HIP_CHECK(hipSetDevice(ctxs[i]->device));
        const ncclResult_t rc = rccl().CommInitRank(&comms[i], num_contexts, uid, i);
        if (rc != ncclSuccess) {
            (void) rccl().GroupEnd();
            rpvg_hip_detail::setError("ncclCommInitRank(rank %d of %d) failed: %s", i, num_contexts, rccl().GetErrorString(rc));
            return RPVG_HIP_ERR_RUNTIME;
        }
    }
    RPVG_RCCL_CHECK(rccl().GroupEnd());
    for (int i = 0; i < num_contexts; ++i) {
        ctxs[i]->comm = comms[i];
        ctxs[i]->comm_world = num_contexts;
        ctxs[i]->comm_rank = i;
    }
    return RPVG_HIP_OK;
}

extern "C" int rpvg_hip_gather(rpvg_hip_ctx * ctx, const double * local_values, uint64_t local_count, const uint64_t * counts, double * all_values) {
    RPVG_REQUIRE(ctx && counts && all_values && (local_values || local_count == 0), "rpvg_hip_gather: NULL argument");
    const int world = ctx->comm ? ctx->comm_world : 1, rank = ctx->comm ? ctx->comm_rank : 0;
    RPVG_REQUIRE(counts[rank] == local_count, "rpvg_hip_gather: counts[%d] = %llu but this rank brings %llu values", rank,
                 static_cast<unsigned long long>(counts[rank]), static_cast<unsigned long long>(local_count));
    uint64_t widest = 0;
    for (int r = 0; r < world; ++r) widest = std::max<uint64_t>(widest, counts[r]);
    if (widest == 0) return RPVG_HIP_OK;
    if (world == 1) {
        std::memcpy(all_values, local_values, local_count * sizeof(double));
        return RPVG_HIP_OK;
    }
    // ragged lengths: every rank sends `widest` values (its own, zero padded), the receiver drops the padding
    std::lock_guard<std::mutex> lock(ctx->mutex);
    RPVG_HIP_CHECK(hipSetDevice(ctx->device));
    rpvg_hip_detail::DeviceBuffer<double> send, recv;
    std::vector<double> padded(widest, 0.0);
    std::memcpy(padded.data(), local_values, local_count * sizeof(double));
    RPVG_HIP_CHECK(send.upload(padded.data(), widest, ctx->stream));
    RPVG_HIP_CHECK(recv.alloc(widest * world));
    RPVG_RCCL_CHECK(rccl().AllGather(send.ptr, recv.ptr, widest, ncclDouble, static_cast<ncclComm_t>(ctx->comm), ctx->stream));
    std::vector<double> gathered(widest * world);
    RPVG_HIP_CHECK(recv.download(gathered.data(), ctx->stream));
    RPVG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    uint64_t out = 0;
    for (int r = 0; r < world; ++r) {
        std::memcpy(all_values + out, gathered.data() + static_cast<uint64_t>(r) * widest, counts[r] * sizeof(double));
        out += counts[r];
    }
    return RPVG_HIP_OK;
}
