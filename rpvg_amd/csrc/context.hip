// Context, device memory helpers, cluster-batch upload and kernel timing of
// librpvg_hip.so.  gfx950 only; no CPU fallback anywhere in this library.

#include "common.hpp"

#include <hipcub/hipcub.hpp>

#include <sys/prctl.h>
#include <time.h>

#include <algorithm>
#include <thread>

#include <map>

namespace rpvg_hip_detail {

static thread_local char g_last_error[1024] = "";

void setError(const char * fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
}

namespace {

struct DevicePool {
    std::map<size_t, std::vector<void *>> free_blocks;  // size class -> cached blocks
    std::map<void *, size_t> live;                      // block -> size class
};

std::mutex g_pool_mutex;
std::map<int, DevicePool> g_pools;
std::map<void *, int> g_block_device;
std::map<int, int> g_device_contexts;

// power-of-two classes below 1 MiB, eight classes per octave above
size_t sizeClass(size_t bytes) {
    if (bytes <= 256) return 256;
    size_t pow2 = 256;
    while (pow2 < bytes) pow2 <<= 1;
    if (pow2 <= (1u << 20)) return pow2;
    const size_t step = pow2 >> 4;  // pow2/2 .. pow2 in 8 steps
    return ((bytes + step - 1) / step) * step;
}

}  // namespace

hipError_t poolAlloc(void ** ptr, size_t bytes) {
    int device = 0;
    hipError_t e = hipGetDevice(&device);
    if (e != hipSuccess) return e;
    const size_t cls = sizeClass(bytes);
    {
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        DevicePool & pool = g_pools[device];
        auto it = pool.free_blocks.find(cls);
        if (it != pool.free_blocks.end() && !it->second.empty()) {
            *ptr = it->second.back();
            it->second.pop_back();
            pool.live[*ptr] = cls;
            return hipSuccess;
        }
    }
    {
        static const bool trace = std::getenv("RPVG_AMD_TRACE") != nullptr;
        if (trace) std::fprintf(stderr, "[rpvg_hip trace] pool miss: device block of %zu bytes\n", cls);
    }
    e = hipMalloc(ptr, cls);
    if (e == hipErrorOutOfMemory) {
        // give cached blocks back to the driver and retry once
        (void) hipGetLastError();
        poolTrim(device);
        e = hipMalloc(ptr, cls);
    }
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    g_pools[device].live[*ptr] = cls;
    g_block_device[*ptr] = device;
    return hipSuccess;
}

void poolFree(void * ptr) {
    if (!ptr) return;
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    auto dev_it = g_block_device.find(ptr);
    if (dev_it == g_block_device.end()) {
        (void) hipFree(ptr);
        return;
    }
    DevicePool & pool = g_pools[dev_it->second];
    auto it = pool.live.find(ptr);
    if (it == pool.live.end()) return;
    pool.free_blocks[it->second].push_back(ptr);
    pool.live.erase(it);
}

void poolTrim(int device) {
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    DevicePool & pool = g_pools[device];
    for (auto & cls : pool.free_blocks) {
        for (void * p : cls.second) {
            (void) hipFree(p);
            g_block_device.erase(p);
        }
        cls.second.clear();
    }
}

namespace {
std::mutex g_pinned_mutex;
std::map<size_t, std::vector<void *>> g_pinned_free;  // size class -> cached blocks
std::map<void *, size_t> g_pinned_live;
}  // namespace

hipError_t pinnedAlloc(void ** ptr, size_t bytes) {
    const size_t cls = sizeClass(bytes);
    {
        std::lock_guard<std::mutex> lock(g_pinned_mutex);
        auto it = g_pinned_free.find(cls);
        if (it != g_pinned_free.end() && !it->second.empty()) {
            *ptr = it->second.back();
            it->second.pop_back();
            g_pinned_live[*ptr] = cls;
            return hipSuccess;
        }
    }
    {
        static const bool trace = std::getenv("RPVG_AMD_TRACE") != nullptr;
        if (trace) std::fprintf(stderr, "[rpvg_hip trace] pool miss: pinned block of %zu bytes\n", cls);
    }
    const hipError_t e = hipHostMalloc(ptr, cls, hipHostMallocDefault);
    if (e != hipSuccess) {
        (void) hipGetLastError();
        *ptr = nullptr;
        return e;
    }
    std::lock_guard<std::mutex> lock(g_pinned_mutex);
    g_pinned_live[*ptr] = cls;
    return hipSuccess;
}

size_t pinnedCapacity(const void * ptr) {
    std::lock_guard<std::mutex> lock(g_pinned_mutex);
    auto it = g_pinned_live.find(const_cast<void *>(ptr));
    return it == g_pinned_live.end() ? 0 : it->second;
}

void pinnedFree(void * ptr) {
    if (!ptr) return;
    std::lock_guard<std::mutex> lock(g_pinned_mutex);
    auto it = g_pinned_live.find(ptr);
    if (it == g_pinned_live.end()) return;
    g_pinned_free[it->second].push_back(ptr);
    g_pinned_live.erase(it);
}

void pinnedTrim() {
    std::lock_guard<std::mutex> lock(g_pinned_mutex);
    for (auto & cls : g_pinned_free) {
        for (void * p : cls.second) (void) hipHostFree(p);
        cls.second.clear();
    }
}

bool stagedUploads() {
    static const bool staged = std::getenv("RPVG_HIP_PAGEABLE_UPLOADS") == nullptr;
    return staged;
}

namespace {
std::mutex g_registered_mutex;
std::map<const char *, size_t> g_registered;  // start -> bytes of the ranges registered through the library
}  // namespace

bool hostIsPinned(const void * host, size_t bytes) {
    if (!host || bytes == 0) return false;
    std::lock_guard<std::mutex> lock(g_registered_mutex);
    if (g_registered.empty()) return false;
    const char * p = static_cast<const char *>(host);
    auto it = g_registered.upper_bound(p);
    if (it == g_registered.begin()) return false;
    --it;
    return p >= it->first && p + bytes <= it->first + it->second;
}

void copyToStaging(void * staging, const void * host, size_t bytes) {
    constexpr size_t kChunk = 8u << 20;
    if (bytes < 2 * kChunk) {
        std::memcpy(staging, host, bytes);
        return;
    }
    const size_t threads = std::min<size_t>(8, (bytes + kChunk - 1) / kChunk);
    const size_t share = ((bytes + threads - 1) / threads + 63) & ~size_t(63);
    std::vector<std::thread> workers;
    for (size_t t = 1; t < threads; ++t) {
        const size_t begin = t * share;
        if (begin >= bytes) break;
        workers.emplace_back([=] { std::memcpy(static_cast<char *>(staging) + begin, static_cast<const char *>(host) + begin, std::min(share, bytes - begin)); });
    }
    std::memcpy(staging, host, std::min(share, bytes));
    for (auto & w : workers) w.join();
}

namespace {
struct CopyLane {
    hipStream_t copy_stream;
    hipEvent_t copied;
};
std::mutex g_copy_mutex;
std::map<hipStream_t, CopyLane> g_copy_lanes;
}  // namespace

void registerCopyStream(hipStream_t stream, hipStream_t copy_stream, hipEvent_t copied) {
    std::lock_guard<std::mutex> lock(g_copy_mutex);
    g_copy_lanes[stream] = CopyLane{copy_stream, copied};
}

void forgetCopyStream(hipStream_t stream) {
    std::lock_guard<std::mutex> lock(g_copy_mutex);
    g_copy_lanes.erase(stream);
}

hipError_t stagedCopy(void * device_dst, const void * pinned_src, size_t bytes, hipStream_t stream) {
    // (measured on the configs[2] bench: 19.1-19.9 ms per batch with the copy stream against 14.4-15.8 ms with the
    // uploads on their own streams — the event waits between hardware queues cost more than the overlap brings; off
    // unless RPVG_HIP_COPY_STREAM=1)
    static const bool inline_uploads = RPVG_EXPERIMENT_ENV("RPVG_HIP_COPY_STREAM") == nullptr;
    CopyLane lane{nullptr, nullptr};
    if (!inline_uploads) {
        std::lock_guard<std::mutex> lock(g_copy_mutex);
        auto it = g_copy_lanes.find(stream);
        if (it != g_copy_lanes.end()) lane = it->second;
    }
    if (!lane.copy_stream) return hipMemcpyAsync(device_dst, pinned_src, bytes, hipMemcpyHostToDevice, stream);
    hipError_t e = hipMemcpyAsync(device_dst, pinned_src, bytes, hipMemcpyHostToDevice, lane.copy_stream);
    if (e == hipSuccess) e = hipEventRecord(lane.copied, lane.copy_stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(stream, lane.copied, 0);
    return e;
}

namespace {

int g_hardware_queues = 4;

__attribute__((constructor)) void askForHardwareQueues() {
    // (8: two host lanes of nine streams over one batch — 12 and 16 measured equal or slower there — and, since round 5, the batch
    // pipeline's four single-lane engines of four streams each: 6.0 ms per configs[2] batch on 8 queues, 6.1 on 6, 6.6 on 4, 7.0 on
    // 10, 5.9 on 12, 8.7 on 16; engines of nine streams each wanted 16 — rpvg_amd/host/batch_pipeline.hpp)
    (void) setenv("GPU_MAX_HW_QUEUES", "8", 0);  // keeps a value the user has set
    const char * env = std::getenv("GPU_MAX_HW_QUEUES");
    g_hardware_queues = env ? std::max(1, std::atoi(env)) : 4;
}

}  // namespace

int hardwareQueues() { return g_hardware_queues; }

}  // namespace rpvg_hip_detail

using namespace rpvg_hip_detail;

namespace rpvg_hip_detail {

namespace {

// how long a wait of the calling thread queries before it starts to nap (rpvg_hip_thread_wait_spin_us)
thread_local uint32_t t_spin_us = 20;

template <typename Query>
hipError_t pollUntilDone(Query query) {
    static const bool spin = std::getenv("RPVG_HIP_SPIN_WAITS") != nullptr;
    if (spin) return hipErrorNotSupported;  // (the caller falls back to the runtime's wait)
    thread_local bool slack_set = false;
    if (!slack_set) {
        (void) prctl(PR_SET_TIMERSLACK, 1000UL, 0UL, 0UL, 0UL);
        slack_set = true;
    }
    const auto begin = std::chrono::steady_clock::now();
    const auto spin_for = std::chrono::microseconds(t_spin_us);
    while (true) {
        const hipError_t e = query();
        if (e != hipErrorNotReady) return e;
        const auto waited = std::chrono::steady_clock::now() - begin;
        if (waited < spin_for) continue;
        // naps of a twentieth of the time waited so far, 30 to 200 us: a nap is ~15 us of CPU time (timer, two context switches), and
        // a wait of many milliseconds — a round of the device sampler, a large cluster's EM — took hundreds of them (8 ms of system
        // time per configs[4] call); the wait ends at most 5 % late
        const long nap_ns = std::min<long>(200000, std::max<long>(30000, std::chrono::duration_cast<std::chrono::nanoseconds>(waited).count() / 20));
        timespec nap{0, nap_ns};
        (void) nanosleep(&nap, nullptr);
    }
}

}  // namespace

namespace {
__global__ void zeroWordsKernel(uint32_t * __restrict__ words, const size_t n) {
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) words[i] = 0u;
}
}  // namespace

hipError_t zeroAsync(void * ptr, const size_t bytes, hipStream_t stream) {
    if (bytes == 0) return hipSuccess;
    if ((reinterpret_cast<uintptr_t>(ptr) & 3u) || (bytes & 3u)) return hipMemsetAsync(ptr, 0, bytes, stream);
    const size_t n = bytes / 4;
    const uint32_t blocks = static_cast<uint32_t>(std::min<size_t>((n + 255) / 256, 2048));
    zeroWordsKernel<<<dim3(blocks), dim3(256), 0, stream>>>(static_cast<uint32_t *>(ptr), n);
    return hipGetLastError();
}

hipError_t waitEvent(hipEvent_t event) {
    const hipError_t e = pollUntilDone([event]() { return hipEventQuery(event); });
    return e == hipErrorNotSupported ? hipEventSynchronize(event) : e;
}

hipError_t waitStream(hipStream_t stream) {
    const hipError_t e = pollUntilDone([stream]() { return hipStreamQuery(stream); });
    return e == hipErrorNotSupported ? hipStreamSynchronize(stream) : e;
}

}  // namespace rpvg_hip_detail

hipError_t rpvg_hip_ctx::forkAux() {
    hipError_t e = hipEventRecord(fork_event, stream);
    for (int i = 0; i < aux_count && e == hipSuccess; ++i) e = hipStreamWaitEvent(aux[i], fork_event, 0);
    return e;
}

hipError_t rpvg_hip_ctx::joinAux() {
    hipError_t e = hipSuccess;
    for (int i = 0; i < aux_count && e == hipSuccess; ++i) {
        e = hipEventRecord(join_event[i], aux[i]);
        if (e == hipSuccess) e = hipStreamWaitEvent(stream, join_event[i], 0);
    }
    return e;
}

// joinAux() with the submitting thread waiting for the side streams itself: `stream` is not left parked on their events (a
// stream whose next command waits for an event holds its hardware queue until it arrives, and the kernels of other
// streams on that queue — the other lane's — stand behind it)
hipError_t rpvg_hip_ctx::joinAuxOnHost() {
    hipError_t e = hipSuccess;
    for (int i = 0; i < aux_count && e == hipSuccess; ++i) e = hipEventRecord(join_event[i], aux[i]);
    for (int i = 0; i < aux_count && e == hipSuccess; ++i) e = rpvg_hip_detail::waitEvent(join_event[i]);
    for (int i = 0; i < aux_count && e == hipSuccess; ++i) e = hipStreamWaitEvent(stream, join_event[i], 0);  // (the ordering, for the record: they have arrived)
    return e;
}

namespace rpvg_hip_detail {
namespace {
// The clock of the timed intervals: one event per GPU that every span of every context on it is measured against
// (hipEventElapsedTime returns a float: the base is renewed by a statistics reset once it is older than a few
// seconds, so that the intervals of a measurement keep microsecond resolution).
struct DeviceClock {
    hipEvent_t base = nullptr;
    uint64_t id = 0;
    std::chrono::steady_clock::time_point taken;
};
std::mutex g_clock_mutex;
std::map<int, DeviceClock> g_clocks;

// Caller has set the device.  renew_if_old: take a new base when the present one is older than five seconds.
DeviceClock deviceClock(const int device, hipStream_t stream, const bool renew_if_old) {
    std::lock_guard<std::mutex> lock(g_clock_mutex);
    DeviceClock & clock = g_clocks[device];
    const auto now = std::chrono::steady_clock::now();
    if (!clock.base || (renew_if_old && now - clock.taken > std::chrono::seconds(5))) {
        hipEvent_t base = nullptr;
        // (the previous base is not destroyed: spans of other contexts may still be folded against it)
        if (hipEventCreate(&base) == hipSuccess && hipEventRecord(base, stream) == hipSuccess && hipEventSynchronize(base) == hipSuccess) {
            clock.base = base;
            clock.id++;
            clock.taken = now;
        }
    }
    return clock;
}
}  // namespace
}  // namespace rpvg_hip_detail

// The spans cost what they record with: two events each, two marker commands in the stream's queue — eighty per configs[2]
// batch with all of them, between kernels that depend on each other; with several batches in flight that was 0.6 of 4.8 ms per
// batch.  span_level (rpvg_hip_ctx): 2 every family (the contexts of rpvg_hip_create: one batch at a time, the statistics
// bench.py and the tools read), 1 the EM launches only (the contexts of rpvg_hip_create_with_streams and rpvg_hip_create_uploader:
// the batch pipeline — the two events around a batch's copies alone were 0.25 of 3.8 ms per upload), 0 none; RPVG_HIP_SPANS=n when a
// context is made overrides.
int rpvg_hip_ctx::spanBegin(int family, hipStream_t on, int sub) {
    if (span_level <= 0 || (span_level == 1 && family != FAM_EM_KERNEL)) return -1;
    // a caller that never reads the statistics: the spans that are done are folded now and then (nothing waits: a span still
    // running stays), once no span is open — handles are positions in the list
    if (open_spans == 0 && spans.size() >= 1024) foldFinishedSpans();
    TimedSpan s;
    s.family = family;
    s.sub = sub;
    if (hipEventCreate(&s.start) != hipSuccess) return -1;
    if (hipEventCreate(&s.stop) != hipSuccess) {
        (void) hipEventDestroy(s.start);
        return -1;
    }
    if (!on) on = stream;
    (void) hipEventRecord(s.start, on);
    spans.push_back(s);
    span_streams.push_back(on);
    ++open_spans;
    return static_cast<int>(spans.size()) - 1;
}

void rpvg_hip_ctx::spanEnd(int idx) {
    if (idx < 0) return;
    (void) hipEventRecord(spans[idx].stop, span_streams[idx]);
    if (open_spans > 0) --open_spans;
}

// what a finished span adds to the statistics (the caller destroys its events)
void rpvg_hip_ctx::accountSpan(const rpvg_hip_detail::TimedSpan & s, const void * clock_base, const uint64_t clock_id) {
    constexpr size_t kMaxIntervals = 1u << 20;
    float ms = 0;
    if (hipEventElapsedTime(&ms, s.start, s.stop) != hipSuccess) return;
    switch (s.family) {
        case FAM_EM_SPARSE: stats.em_sparse_ms += ms; break;
        case FAM_EM_DENSE: stats.em_dense_ms += ms; break;
        case FAM_LOGLIK: stats.loglik_ms += ms; break;
        case FAM_BUILD: stats.build_ms += ms; break;
        case FAM_H2D: stats.h2d_ms += ms; break;
        case FAM_COLLAPSE: stats.collapse_ms += ms; break;
        case FAM_GIBBS: stats.gibbs_ms += ms; break;
        case FAM_TILE: stats.search_tile_ms += ms; stats.search_tile_launches += 1; break;
        case FAM_EM_KERNEL:
            if (s.sub >= 0 && s.sub < RPVG_HIP_EM_KERNELS) stats.em_kernel[s.sub].ms += ms;
            break;
        default: break;
    }
    float at = 0;
    // (the per-kernel spans lie inside their call's FAM_EM_SPARSE span: not a second interval)
    // (... and the conditionals' FAM_LOGLIK spans inside their sampler's FAM_GIBBS span)
    hipEvent_t base = static_cast<hipEvent_t>(const_cast<void *>(clock_base));
    if (s.family != FAM_EM_KERNEL && s.family != FAM_GIBBS && base && intervals.size() < kMaxIntervals &&
        hipEventElapsedTime(&at, base, s.start) == hipSuccess) {
        intervals.push_back(TimedInterval{static_cast<double>(at), static_cast<double>(at) + ms, s.family, clock_id});
    }
}

void rpvg_hip_ctx::foldFinishedSpans() {
    const DeviceClock clock = deviceClock(device, stream, false);
    size_t kept = 0;
    for (size_t i = 0; i < spans.size(); ++i) {
        if (hipEventQuery(spans[i].stop) == hipSuccess) {
            accountSpan(spans[i], clock.base, clock.id);
            (void) hipEventDestroy(spans[i].start);
            (void) hipEventDestroy(spans[i].stop);
        } else {
            (void) hipGetLastError();
            spans[kept] = spans[i];
            span_streams[kept] = span_streams[i];
            ++kept;
        }
    }
    (void) hipGetLastError();
    spans.resize(kept);
    span_streams.resize(kept);
}

int rpvg_hip_ctx::foldSpans() {
    RPVG_HIP_CHECK(hipStreamSynchronize(stream));
    if (collapse_stream) RPVG_HIP_CHECK(hipStreamSynchronize(collapse_stream));
    const DeviceClock clock = deviceClock(device, stream, false);
    for (auto & s : spans) {
        (void) hipEventSynchronize(s.stop);  // spans on side streams
        accountSpan(s, clock.base, clock.id);
        (void) hipGetLastError();
        (void) hipEventDestroy(s.start);
        (void) hipEventDestroy(s.stop);
    }
    spans.clear();
    span_streams.clear();
    open_spans = 0;
    // union of the intervals on the present clock
    std::vector<std::pair<double, double>> iv;
    for (auto & t : intervals) {
        if (t.clock == clock.id) iv.emplace_back(t.start_ms, t.stop_ms);
    }
    std::sort(iv.begin(), iv.end());
    double busy = 0, cur_s = 0, cur_e = -1;
    for (auto & x : iv) {
        if (cur_e < cur_s || x.first > cur_e) {
            if (cur_e >= cur_s) busy += cur_e - cur_s;
            cur_s = x.first;
            cur_e = x.second;
        } else if (x.second > cur_e) {
            cur_e = x.second;
        }
    }
    if (cur_e >= cur_s) busy += cur_e - cur_s;
    stats.busy_ms = busy;
    return RPVG_HIP_OK;
}

// ---- kernels: expand (probability, path list) groups to entries -------------

// One thread per probability group: writes the group's probability next to
// each of its path indices (the path indices themselves are uploaded as is).
// (RowOff / GrpOff: the width the caller wrote the two long offset arrays in, include/rpvg_batch.h)
template <typename GrpOff>
__global__ void expandGroupsKernel(const uint64_t num_groups, const uint64_t num_entries, const GrpOff * __restrict__ grp_idx_off,
                                   const double * __restrict__ grp_prob, double * __restrict__ ent_prob) {
    const uint64_t g = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (g >= num_groups) return;
    const double p = grp_prob[g];
    // (clamped: the offsets are checked by validateRowsKernel, whose verdict the host reads after these kernels)
    for (uint64_t e = grp_idx_off[g]; e < min(static_cast<uint64_t>(grp_idx_off[g + 1]), num_entries); ++e) ent_prob[e] = p;
}

// One thread per row: entry range of the row and its count as double.
template <typename RowOff, typename GrpOff>
__global__ void rowMetaKernel(const uint64_t num_rows, const uint64_t num_groups, const RowOff * __restrict__ row_grp_off,
                              const GrpOff * __restrict__ grp_idx_off, const uint32_t * __restrict__ row_count_u32,
                              uint64_t * __restrict__ row_ent_off, double * __restrict__ row_count) {
    const uint64_t r = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (r > num_rows) return;
    row_ent_off[r] = grp_idx_off[min(static_cast<uint64_t>(row_grp_off[r]), num_groups)];
    if (r < num_rows) row_count[r] = static_cast<double>(row_count_u32[r]);
}

// The row invariants the estimators rely on (src/main.cpp:855-887,953-973; src/read_path_probabilities.cpp:91-105,184,
// 212-219), one thread per row: consistent offsets, noise probability in (0, 1], path indices inside the row's cluster.
// first_bad_row: the smallest row that breaks one (the host words the message: validateClusters).  On the device because
// the host pass over a batch's 280 MB cost more than their copy (8 threads: 5 ms, before the first byte moved).
template <typename RowOff, typename GrpOff>
__global__ void validateRowsKernel(const uint64_t num_rows, const uint64_t num_groups, const uint64_t num_entries, const uint32_t num_clusters,
                                   const uint64_t * __restrict__ cluster_row_off, const uint64_t * __restrict__ cluster_path_off,
                                   const RowOff * __restrict__ row_grp_off, const GrpOff * __restrict__ grp_idx_off,
                                   const double * __restrict__ row_noise, const uint32_t * __restrict__ path_idx,
                                   unsigned long long * __restrict__ first_bad_row) {
    const uint64_t r = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (r >= num_rows) return;
    bool good = true;
    const uint64_t g0 = row_grp_off[r], g1 = row_grp_off[r + 1];
    good = g0 <= g1 && g1 <= num_groups;
    const double nz = row_noise[r];
    good = good && nz > 0 && nz <= 1;
    if (good) {
        // the row's cluster: the last one that starts at or before it
        uint32_t lo = 0, hi = num_clusters;  // cluster_row_off[lo] <= r < cluster_row_off[hi]
        while (hi - lo > 1) {
            const uint32_t mid = lo + (hi - lo) / 2;
            if (cluster_row_off[mid] <= r) lo = mid;
            else hi = mid;
        }
        const uint64_t n_paths = cluster_path_off[lo + 1] - cluster_path_off[lo];
        uint64_t e = grp_idx_off[g0];
        good = e <= num_entries;
        for (uint64_t g = g0; good && g < g1; ++g) {
            const uint64_t e1 = grp_idx_off[g + 1];
            good = e <= e1 && e1 <= num_entries;
            for (; good && e < e1; ++e) good = path_idx[e] < n_paths;
        }
    }
    if (!good) atomicMin(first_bad_row, static_cast<unsigned long long>(r));
}

extern "C" {

int rpvg_hip_device_count(int * count) {
    RPVG_REQUIRE(count != nullptr, "rpvg_hip_device_count: count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        setError("hipGetDeviceCount failed: %s", hipGetErrorString(e));
        *count = 0;
        return RPVG_HIP_ERR_NO_DEVICE;
    }
    *count = n;
    return RPVG_HIP_OK;
}

namespace {
// The context's own stream carries the short kernels that stand between a host lane and its next stage (matrix build,
// row collapse, EM problem set); the long ones — the searches, the EM bins — run on the side streams.  With a higher
// priority the short ones of one lane get their workgroups in between those of the other lane's long kernels
// — the idea; measured 17.4-17.8 ms per configs[2] batch against 15.1-15.9 ms with all streams alike (same box, same
// call), so it is off unless RPVG_HIP_MAIN_PRIORITY=1.
hipError_t createMainStream(hipStream_t * stream, const bool highest_priority) {
    const char * env = RPVG_EXPERIMENT_ENV("RPVG_HIP_MAIN_PRIORITY");  // A/B knob: every context's stream at the highest priority (slower)
    if (!highest_priority && (!env || std::atoi(env) == 0)) return hipStreamCreateWithFlags(stream, hipStreamNonBlocking);
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess || least == greatest) {
        (void) hipGetLastError();
        return hipStreamCreateWithFlags(stream, hipStreamNonBlocking);
    }
    return hipStreamCreateWithPriority(stream, hipStreamNonBlocking, greatest);
}

// The runtime maps its streams onto a pool of GPU_MAX_HW_QUEUES hardware queues per priority (least used first), and the
// commands of streams that share a queue run one after the other.  With several batches in flight (the contexts of few side
// streams: rpvg_hip_create_with_streams) that is where a batch's time went: the main stream — the chain of dependent kernels
// from the matrices to the packed results — shared its queue with other contexts' side streams, whose EM kernels run for a
// millisecond each.  A stream with a CU mask gets a hardware queue of its own, outside the pools; the mask here names every
// CU.  5.7 -> 5.1-5.2 ms per configs[2] batch with four such contexts; the side or collapse streams as well: 14 and 7 ms —
// beyond some twenty hardware queues in the process the device time-slices them (GPU_MAX_HW_QUEUES=10: 8 ms).
// RPVG_HIP_POOLED_MAIN_QUEUE=1: the main stream from the pool, as in the contexts of rpvg_hip_create.
bool ownQueueForMainStream(const bool uploader, const int side_streams) {
    static const bool pooled = std::getenv("RPVG_HIP_POOLED_MAIN_QUEUE") != nullptr && std::atoi(std::getenv("RPVG_HIP_POOLED_MAIN_QUEUE")) != 0;
    return !uploader && side_streams < rpvg_hip_ctx::kAuxStreams && !pooled;
}

hipError_t createOwnQueueStream(hipStream_t * stream, const int num_cus) {
    uint32_t every_cu[32];
    for (uint32_t & word : every_cu) word = 0xffffffffu;
    const uint32_t words = static_cast<uint32_t>(std::min(32, std::max(1, (num_cus + 31) / 32)));  // (256 CUs: eight words; bits past the device's last are ignored)
    hipError_t e = hipExtStreamCreateWithCUMask(stream, words, every_cu);
    if (e != hipSuccess) {  // (a runtime or a partition mode that does not take the mask: a stream of the pool, as everywhere else)
        (void) hipGetLastError();
        e = hipStreamCreateWithFlags(stream, hipStreamNonBlocking);
    }
    return e;
}

int createContext(int device, bool uploader, int side_streams, rpvg_hip_ctx ** ctx_out);
}  // namespace

int rpvg_hip_create(int device, rpvg_hip_ctx ** ctx_out) { return createContext(device, false, rpvg_hip_ctx::kAuxStreams, ctx_out); }

int rpvg_hip_create_uploader(int device, rpvg_hip_ctx ** ctx_out) { return createContext(device, true, 1, ctx_out); }

int rpvg_hip_create_with_streams(int device, int side_streams, rpvg_hip_ctx ** ctx_out) {
    RPVG_REQUIRE(side_streams >= 1 && side_streams <= rpvg_hip_ctx::kAuxStreams, "rpvg_hip_create_with_streams: 1 to %d side streams", rpvg_hip_ctx::kAuxStreams);
    return createContext(device, false, side_streams, ctx_out);
}

namespace {
int createContext(int device, const bool uploader, const int side_streams, rpvg_hip_ctx ** ctx_out) {
    RPVG_REQUIRE(ctx_out != nullptr, "rpvg_hip_create: ctx_out is NULL");
    *ctx_out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        setError("rpvg_hip_create: no HIP device available (%s); this engine has no CPU fallback",
                 e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
        return RPVG_HIP_ERR_NO_DEVICE;
    }
    RPVG_REQUIRE(device >= 0 && device < n, "rpvg_hip_create: device %d out of range [0, %d)", device, n);
    rpvg_hip_ctx * ctx = new (std::nothrow) rpvg_hip_ctx();
    if (!ctx) {
        setError("rpvg_hip_create: out of host memory");
        return RPVG_HIP_ERR_ALLOC;
    }
    ctx->device = device;
    memset(&ctx->stats, 0, sizeof(ctx->stats));
    {
        const char * env = std::getenv("RPVG_HIP_SPANS");  // (read per context)
        ctx->span_level = env ? std::max(0, std::min(2, std::atoi(env))) : ((uploader || side_streams < rpvg_hip_ctx::kAuxStreams) ? 1 : 2);
    }
    if ((e = hipSetDevice(device)) != hipSuccess || (e = hipGetDeviceProperties(&ctx->props, device)) != hipSuccess ||
        (e = ownQueueForMainStream(uploader, side_streams) ? createOwnQueueStream(&ctx->stream, ctx->props.multiProcessorCount) : createMainStream(&ctx->stream, uploader)) != hipSuccess) {
        setError("rpvg_hip_create: %s", hipGetErrorString(e));
        delete ctx;
        return RPVG_HIP_ERR_RUNTIME;
    }
    // side_streams real side streams; the other entries of aux[] are aliases of them, so that every use site keeps its index
    // (launches that would have had streams of their own then follow one another on the stream they share)
    ctx->aux_count = side_streams;
    for (int i = 0; i < rpvg_hip_ctx::kAuxStreams && e == hipSuccess; ++i) {
        if (i < side_streams) {
            e = hipStreamCreateWithFlags(&ctx->aux[i], hipStreamNonBlocking);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->join_event[i], hipEventDisableTiming);
        } else {
            ctx->aux[i] = ctx->aux[i % side_streams];
        }
    }
    if (e == hipSuccess) e = createMainStream(&ctx->collapse_stream, RPVG_EXPERIMENT_ENV("RPVG_HIP_COLLAPSE_PRIORITY") == nullptr || std::atoi(RPVG_EXPERIMENT_ENV("RPVG_HIP_COLLAPSE_PRIORITY")) != 0);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->fork_event, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->search_done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->copied, hipEventDisableTiming);
    if (e == hipSuccess) {
        registerCopyStream(ctx->stream, ctx->copy_stream, ctx->copied);
        for (int i = 0; i < ctx->aux_count; ++i) registerCopyStream(ctx->aux[i], ctx->copy_stream, ctx->copied);
    }
    if (e != hipSuccess) {
        setError("rpvg_hip_create: %s", hipGetErrorString(e));
        delete ctx;
        return RPVG_HIP_ERR_RUNTIME;
    }
    if (strncmp(ctx->props.gcnArchName, "gfx950", 6) != 0) {
        setError("rpvg_hip_create: device %d is %s; this library is built for gfx950 only", device, ctx->props.gcnArchName);
        (void) hipStreamDestroy(ctx->stream);
        delete ctx;
        return RPVG_HIP_ERR_NO_DEVICE;
    }
    {
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        g_device_contexts[device]++;
    }
    *ctx_out = ctx;
    return RPVG_HIP_OK;
}
}  // namespace

void rpvg_hip_destroy(rpvg_hip_ctx * ctx) {
    if (!ctx) return;
    const bool trace = std::getenv("RPVG_AMD_TRACE_EXIT") != nullptr;
#define RPVG_EXIT_STEP(what) do { if (trace) std::fprintf(stderr, "[exit]   rpvg_hip_destroy: %s\n", what); } while (0)
    RPVG_EXIT_STEP("begin");
    (void) hipSetDevice(ctx->device);
    (void) ctx->foldSpans();
    RPVG_EXIT_STEP("spans folded");
    (void) rpvg_hip_comm_destroy(ctx);
    if (ctx->stream) forgetCopyStream(ctx->stream);
    for (int i = 0; i < ctx->aux_count; ++i) {
        if (ctx->aux[i]) forgetCopyStream(ctx->aux[i]);
    }
    if (ctx->copy_stream) {
        (void) hipStreamSynchronize(ctx->copy_stream);
        (void) hipStreamDestroy(ctx->copy_stream);
    }
    RPVG_EXIT_STEP("copy stream destroyed");
    if (ctx->copied) (void) hipEventDestroy(ctx->copied);
    // Every stream drained, then the pooled streams, and the main stream last: a main stream with a hardware queue of its own
    // (createOwnQueueStream) destroyed in front of the side streams left hipStreamDestroy of the first side stream hanging in one
    // process exit of twenty (tools/r06_exit_hang.sh: the reference-shaped factory binary, whose default engine goes at exit).
    if (ctx->stream) (void) hipStreamSynchronize(ctx->stream);
    for (int i = 0; i < ctx->aux_count; ++i) {
        if (ctx->aux[i]) (void) hipStreamSynchronize(ctx->aux[i]);
    }
    if (ctx->collapse_stream) (void) hipStreamSynchronize(ctx->collapse_stream);
    RPVG_EXIT_STEP("streams drained");
    for (int i = 0; i < ctx->aux_count; ++i) {  // (the entries behind are aliases)
        if (ctx->aux[i]) (void) hipStreamDestroy(ctx->aux[i]);
        RPVG_EXIT_STEP("a side stream destroyed");
        if (ctx->join_event[i]) (void) hipEventDestroy(ctx->join_event[i]);
    }
    if (ctx->collapse_stream) (void) hipStreamDestroy(ctx->collapse_stream);
    RPVG_EXIT_STEP("side and collapse streams destroyed");
    if (ctx->stream) (void) hipStreamDestroy(ctx->stream);
    RPVG_EXIT_STEP("main stream destroyed");
    if (ctx->fork_event) (void) hipEventDestroy(ctx->fork_event);
    for (hipStream_t grid_stream : ctx->grid_stream) {
        if (grid_stream) {
            (void) hipStreamSynchronize(grid_stream);
            (void) hipStreamDestroy(grid_stream);
        }
    }
    if (ctx->grid_ready) (void) hipEventDestroy(ctx->grid_ready);
    searchGateForget(ctx);
    if (ctx->search_done) (void) hipEventDestroy(ctx->search_done);
    bool last = false;
    {
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        last = (--g_device_contexts[ctx->device] <= 0);
    }
    RPVG_EXIT_STEP("events destroyed");
    if (last) {
        poolTrim(ctx->device);
        RPVG_EXIT_STEP("pool trimmed");
        bool any = false;
        {
            std::lock_guard<std::mutex> lock(g_pool_mutex);
            for (auto & kv : g_device_contexts) any = any || kv.second > 0;
        }
        if (!any) pinnedTrim();
        RPVG_EXIT_STEP("pinned blocks trimmed");
    }
    delete ctx;
    RPVG_EXIT_STEP("done");
#undef RPVG_EXIT_STEP
}

const char * rpvg_hip_last_error(void) { return g_last_error; }

int rpvg_hip_synchronize(rpvg_hip_ctx * ctx) {
    RPVG_REQUIRE(ctx != nullptr, "rpvg_hip_synchronize: ctx is NULL");
    std::lock_guard<std::mutex> lock(ctx->mutex);
    RPVG_HIP_CHECK(hipSetDevice(ctx->device));
    RPVG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (ctx->collapse_stream) RPVG_HIP_CHECK(hipStreamSynchronize(ctx->collapse_stream));
    return RPVG_HIP_OK;
}

int rpvg_hip_device_info(rpvg_hip_ctx * ctx, char * name, uint32_t name_cap, uint32_t * num_cus, uint64_t * mem_bytes) {
    RPVG_REQUIRE(ctx != nullptr, "rpvg_hip_device_info: ctx is NULL");
    if (name && name_cap) {
        snprintf(name, name_cap, "%s (%s)", ctx->props.name, ctx->props.gcnArchName);
    }
    if (num_cus) *num_cus = ctx->props.multiProcessorCount;
    if (mem_bytes) *mem_bytes = ctx->props.totalGlobalMem;
    return RPVG_HIP_OK;
}

int rpvg_hip_malloc(rpvg_hip_ctx * ctx, uint64_t bytes, void ** device_ptr_out) {
    RPVG_REQUIRE(ctx != nullptr && device_ptr_out != nullptr, "rpvg_hip_malloc: NULL argument");
    std::lock_guard<std::mutex> lock(ctx->mutex);
    RPVG_HIP_CHECK(hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(device_ptr_out, bytes);
    if (e != hipSuccess) {
        setError("rpvg_hip_malloc: %llu bytes: %s", static_cast<unsigned long long>(bytes), hipGetErrorString(e));
        return RPVG_HIP_ERR_ALLOC;
    }
    return RPVG_HIP_OK;
}

int rpvg_hip_free(rpvg_hip_ctx * ctx, void * device_ptr) {
    RPVG_REQUIRE(ctx != nullptr, "rpvg_hip_free: ctx is NULL");
    std::lock_guard<std::mutex> lock(ctx->mutex);
    RPVG_HIP_CHECK(hipSetDevice(ctx->device));
    RPVG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    RPVG_HIP_CHECK(hipFree(device_ptr));
    return RPVG_HIP_OK;
}

int rpvg_hip_memcpy_h2d(rpvg_hip_ctx * ctx, void * device_dst, const void * host_src, uint64_t bytes) {
    RPVG_REQUIRE(ctx != nullptr, "rpvg_hip_memcpy_h2d: ctx is NULL");
    std::lock_guard<std::mutex> lock(ctx->mutex);
    RPVG_HIP_CHECK(hipSetDevice(ctx->device));
    RPVG_HIP_CHECK(hipMemcpyAsync(device_dst, host_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    RPVG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return RPVG_HIP_OK;
}

int rpvg_hip_memcpy_d2h(rpvg_hip_ctx * ctx, void * host_dst, const void * device_src, uint64_t bytes) {
    RPVG_REQUIRE(ctx != nullptr, "rpvg_hip_memcpy_d2h: ctx is NULL");
    std::lock_guard<std::mutex> lock(ctx->mutex);
    RPVG_HIP_CHECK(hipSetDevice(ctx->device));
    RPVG_HIP_CHECK(hipMemcpyAsync(host_dst, device_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    RPVG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return RPVG_HIP_OK;
}

extern "C" int rpvg_hip_host_register(void * host, uint64_t bytes) {
    RPVG_REQUIRE(host && bytes > 0, "rpvg_hip_host_register: NULL or empty range");
    RPVG_HIP_CHECK(hipHostRegister(host, bytes, hipHostRegisterDefault));
    std::lock_guard<std::mutex> lock(g_registered_mutex);
    g_registered[static_cast<const char *>(host)] = bytes;
    return RPVG_HIP_OK;
}

extern "C" int rpvg_hip_host_unregister(void * host) {
    RPVG_REQUIRE(host, "rpvg_hip_host_unregister: NULL");
    {
        std::lock_guard<std::mutex> lock(g_registered_mutex);
        RPVG_REQUIRE(g_registered.erase(static_cast<const char *>(host)) == 1, "rpvg_hip_host_unregister: the range was not registered here");
    }
    RPVG_HIP_CHECK(hipHostUnregister(host));
    return RPVG_HIP_OK;
}

namespace {

// the two long offset arrays of a host batch, in whichever width the caller wrote them (include/rpvg_batch.h); of a batch that
// came with counts only, by adding them up (the wording of an error message: nothing else reads them here then)
inline bool countForm(const rpvg_cluster_batch * hb) { return hb->row_grp_count8 != nullptr && hb->grp_idx_count8 != nullptr; }
inline uint64_t rowGroupOffset(const rpvg_cluster_batch * hb, const uint64_t r) {
    if (hb->row_grp_off32) return hb->row_grp_off32[r];
    if (hb->row_grp_off) return hb->row_grp_off[r];
    uint64_t sum = 0;
    for (uint64_t i = 0; i < r; ++i) sum += hb->row_grp_count8[i];
    return sum;
}
inline uint64_t groupEntryOffset(const rpvg_cluster_batch * hb, const uint64_t g) {
    if (hb->grp_idx_off32) return hb->grp_idx_off32[g];
    if (hb->grp_idx_off) return hb->grp_idx_off[g];
    uint64_t sum = 0;
    for (uint64_t i = 0; i < g; ++i) sum += hb->grp_idx_count8[i];
    return sum;
}

// counts of one byte -> their running sums in 32 bits: offsets[i] = counts[0] + ... + counts[i - 1], i = 0 .. n
struct CountAt {
    const uint8_t * counts;
    uint64_t n;
    __host__ __device__ uint32_t operator()(const uint64_t i) const { return i < n ? counts[i] : 0u; }
};

hipError_t queueOffsetsFromCounts(hipStream_t stream, const uint8_t * counts, const uint64_t n, uint32_t * offsets, DeviceBuffer<unsigned char> & scratch) {
    hipcub::CountingInputIterator<uint64_t> index(0);
    hipcub::TransformInputIterator<uint32_t, CountAt, hipcub::CountingInputIterator<uint64_t> > values(index, CountAt{counts, n});
    size_t bytes = 0;
    hipError_t e = hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, values, offsets, static_cast<int>(n + 1), stream);
    if (e == hipSuccess) e = scratch.alloc(bytes);
    if (e == hipSuccess) e = hipcub::DeviceScan::ExclusiveSum(scratch.ptr, bytes, values, offsets, static_cast<int>(n + 1), stream);
    return e;
}

// the entry offset of every cluster's first row
__global__ void clusterEntryOffsetsKernel(const uint32_t num_clusters, const uint64_t * __restrict__ cluster_row_off, const uint64_t * __restrict__ row_ent_off,
                                          uint64_t * __restrict__ out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k <= num_clusters) out[k] = row_ent_off[cluster_row_off[k]];
}

// the narrow forms of a batch's copy back in 32 bits: path indices, read counts (a count of 255 stands for "listed": escapeRowCountsKernel), source ids
__global__ __launch_bounds__(256) void widenNarrowKernel(const uint16_t * __restrict__ path16, uint32_t * __restrict__ path32, const uint64_t num_entries,
                                                         const uint8_t * __restrict__ count8, uint32_t * __restrict__ count32, const uint64_t num_rows,
                                                         const uint16_t * __restrict__ source16, uint32_t * __restrict__ source32, const uint64_t num_sources,
                                                         const uint16_t * __restrict__ noise16, const double * __restrict__ noise_table, const uint32_t num_noise_values,
                                                         double * __restrict__ noise, const uint64_t num_noise_rows) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x, first = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    for (uint64_t i = first; i < num_noise_rows; i += stride) noise[i] = noise16[i] < num_noise_values ? noise_table[noise16[i]] : -1.0;
    for (uint64_t i = first; i < num_entries; i += stride) path32[i] = path16[i];
    for (uint64_t i = first; i < num_rows; i += stride) count32[i] = count8[i];
    for (uint64_t i = first; i < num_sources; i += stride) source32[i] = source16[i];
}

__global__ void escapeRowCountsKernel(const uint32_t * __restrict__ escape_row, const uint32_t * __restrict__ escape_count, const uint64_t num_escapes,
                                      const uint64_t num_rows, uint32_t * __restrict__ count32) {
    const uint64_t i = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (i < num_escapes && escape_row[i] < num_rows) count32[escape_row[i]] = escape_count[i];
}

__global__ void widenOffsetsKernel(const uint64_t n, const uint32_t * __restrict__ narrow, uint64_t * __restrict__ wide) {
    const uint64_t i = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (i < n) wide[i] = narrow[i];
}

// Words what is wrong with row r of cluster k (validateRowsKernel found it); true: nothing is.
constexpr size_t kProblemChars = 256;
bool validateRow(const rpvg_cluster_batch * hb, const uint32_t k, const uint64_t r, char * message) {
    message[0] = 0;
    const uint64_t n_paths = hb->cluster_path_off[k + 1] - hb->cluster_path_off[k];
    const double nz = hb->row_noise16 ? (hb->row_noise16[r] < hb->num_row_noise_values ? hb->row_noise_table[hb->row_noise16[r]] : -1.0) : hb->row_noise[r];
    if (!(nz > 0 && nz <= 1)) {
        std::snprintf(message, kProblemChars, "rpvg_hip_batch_upload: row %llu has noise probability %g outside (0, 1]",
                      static_cast<unsigned long long>(r), nz);
        return false;
    }
    for (uint64_t e = groupEntryOffset(hb, rowGroupOffset(hb, r)); e < groupEntryOffset(hb, rowGroupOffset(hb, r + 1)); ++e) {
        const uint32_t path = hb->path_idx16 ? hb->path_idx16[e] : hb->path_idx[e];
        if (!(path < n_paths)) {
            std::snprintf(message, kProblemChars, "rpvg_hip_batch_upload: row %llu refers to path %u of a cluster with %llu paths",
                          static_cast<unsigned long long>(r), path, static_cast<unsigned long long>(n_paths));
            return false;
        }
    }
    return true;
}

}  // namespace

namespace {
// ---- a batch from its callers' segments (include/rpvg_batch.h, rpvg_cluster_segment) ------------------------------------------
// What the kernel reads per cluster, in page-locked host memory like the segments themselves: the segment's arrays as pointers and
// where the cluster starts in every array of the device batch.
struct SegmentEntry {
    const uint32_t * row_count;
    const double * row_noise;
    const uint32_t * row_grp_off;
    const uint32_t * grp_idx_off;
    const double * grp_prob;
    const uint32_t * path_idx;
    const uint32_t * path_group_id;
    const uint32_t * path_source_off;
    const uint32_t * source_id;
    const uint32_t * col_count;  // the caller's haplotype columns (rpvg_cluster_segment::has_columns), else null
    const uint32_t * col_end;
    const uint32_t * col_path;
    uint32_t R, G, NNZ, P, S, C;  // (with columns S is their total list length L: the cluster's slots in the column arrays)
    uint64_t row_base, ent_base, path_base, src_base;
};

struct SegmentGatherArgs {
    const SegmentEntry * table;  // [K], host memory
    uint32_t num_clusters;
    uint32_t with_paths;
    uint64_t * cluster_row_off;
    uint64_t * cluster_path_off;
    uint64_t * cluster_src_off;
    double * row_count;
    double * row_noise;
    uint64_t * row_ent_off;
    uint32_t * ent_path;
    double * ent_prob;
    uint32_t * path_group_id;
    uint64_t * path_source_off;
    uint32_t * source_id;
    uint32_t with_columns;           // the segments carry their haplotype columns: src_col_* are written here
    uint32_t * src_col_count;
    uint32_t * src_col_end;
    uint32_t * src_col_path;
    unsigned long long * first_bad;  // smallest (cluster << 8 | kind) of an invalid segment; ~0: none
};

constexpr uint32_t kSegmentBadNoise = 1, kSegmentBadPath = 2, kSegmentBadOffsets = 3;

// Workgroups (k, y): cluster k's segment, slice y of its rows, groups, entries, paths and source ids.  Everything a thread reads
// from a segment is an index it has checked against the segment's own sizes first (the host has checked the arrays against the
// block): a caller's inconsistent offsets end in an error, not in a wild read.  Reads are 4 or 8 bytes per lane, a wave's next to
// each other: PCIe reads of 256-512 bytes.
__global__ __launch_bounds__(256) void segmentsGatherKernel(const SegmentGatherArgs a) {
    __shared__ SegmentEntry s_entry;
    const uint32_t k = blockIdx.x;
    if (threadIdx.x < sizeof(SegmentEntry) / 8) {
        reinterpret_cast<unsigned long long *>(&s_entry)[threadIdx.x] = reinterpret_cast<const unsigned long long *>(a.table + k)[threadIdx.x];
    }
    __syncthreads();
    const SegmentEntry & s = s_entry;
    const uint32_t t = blockIdx.y * blockDim.x + threadIdx.x, stride = gridDim.y * blockDim.x;
    uint32_t bad = 0;
    for (uint32_t r = t; r < s.R; r += stride) {
        const double noise = s.row_noise[r];
        const uint32_t g0 = s.row_grp_off[r], g1 = s.row_grp_off[r + 1];
        if (!(noise > 0 && noise <= 1)) bad = bad ? bad : kSegmentBadNoise;
        uint64_t first_entry = 0;
        if (g0 <= g1 && g1 <= s.G && (r > 0 || g0 == 0) && (r + 1 < s.R || g1 == s.G)) {
            first_entry = s.grp_idx_off[g0];
            if (first_entry > s.NNZ) bad = kSegmentBadOffsets;
        } else {
            bad = kSegmentBadOffsets;
        }
        a.row_count[s.row_base + r] = static_cast<double>(s.row_count[r]);
        a.row_noise[s.row_base + r] = noise;
        a.row_ent_off[s.row_base + r] = s.ent_base + first_entry;
    }
    for (uint32_t g = t; g < s.G; g += stride) {
        const uint32_t e0 = s.grp_idx_off[g], e1 = s.grp_idx_off[g + 1];
        if (e0 <= e1 && e1 <= s.NNZ && (g > 0 || e0 == 0) && (g + 1 < s.G || e1 == s.NNZ)) {
            const double prob = s.grp_prob[g];
            for (uint32_t e = e0; e < e1; ++e) a.ent_prob[s.ent_base + e] = prob;
        } else {
            bad = kSegmentBadOffsets;
        }
    }
    for (uint32_t e = t; e < s.NNZ; e += stride) {
        const uint32_t path = s.path_idx[e];
        if (!(path < s.P)) bad = bad ? bad : kSegmentBadPath;
        a.ent_path[s.ent_base + e] = path;
    }
    if (a.with_columns) {
        // the caller's columns into the cluster's slots (src_base: the lists' total lengths of the clusters before it)
        for (uint32_t p = t; p < s.P; p += stride) a.path_group_id[s.path_base + p] = s.path_group_id[p];
        for (uint32_t c = t; c < s.C; c += stride) {
            const uint32_t begin = c ? s.col_end[c - 1] : 0u, end = s.col_end[c];
            if (!(begin < end && end <= s.S && (c + 1 < s.C || end == s.S)) || s.col_count[c] == 0) bad = kSegmentBadOffsets;
            a.src_col_count[s.src_base + c] = s.col_count[c];
            a.src_col_end[s.src_base + c] = end;
        }
        for (uint32_t i = t; i < s.S; i += stride) {
            const uint32_t path = s.col_path[i];
            if (!(path < s.P)) bad = bad ? bad : kSegmentBadPath;
            a.src_col_path[s.src_base + i] = path;
        }
    } else if (a.with_paths) {
        for (uint32_t p = t; p < s.P; p += stride) {
            const uint32_t s0 = s.path_source_off[p], s1 = s.path_source_off[p + 1];
            if (!(s0 <= s1 && s1 <= s.S && (p > 0 || s0 == 0) && (p + 1 < s.P || s1 == s.S))) bad = kSegmentBadOffsets;
            a.path_group_id[s.path_base + p] = s.path_group_id[p];
            a.path_source_off[s.path_base + p] = s.src_base + s0;
        }
        for (uint32_t i = t; i < s.S; i += stride) a.source_id[s.src_base + i] = s.source_id[i];
    }
    if (t == 0) {
        a.cluster_row_off[k] = s.row_base;
        a.cluster_path_off[k] = s.path_base;
        a.row_ent_off[s.row_base + s.R] = s.ent_base + s.NNZ;  // (the next cluster's first row writes the same value)
        if (a.with_paths || a.with_columns) a.cluster_src_off[k] = s.src_base;
        if (a.with_paths && !a.with_columns) a.path_source_off[s.path_base + s.P] = s.src_base + s.S;
        if (k + 1 == a.num_clusters) {
            a.cluster_row_off[k + 1] = s.row_base + s.R;
            a.cluster_path_off[k + 1] = s.path_base + s.P;
            if (a.with_paths || a.with_columns) a.cluster_src_off[k + 1] = s.src_base + s.S;
        }
    }
    if (bad) atomicMin(a.first_bad, (static_cast<unsigned long long>(k) << 8) | bad);
}

}  // namespace

// The two halves of an upload.  Begin: offsets checked, copies queued on `ctx` (an uploader's stream, usually).  Finish: the
// kernels behind the copies — expansion of the (probability, path list) groups, row meta data, validation, read totals, the
// haplotype columns — on any context of the device, and the small results they bring back.
static int uploadBegin(rpvg_hip_ctx * ctx, const rpvg_cluster_batch * hb, rpvg_hip_batch ** batch_out) {
    RPVG_REQUIRE(ctx != nullptr && hb != nullptr && batch_out != nullptr, "rpvg_hip_batch_upload: NULL argument");
    *batch_out = nullptr;
    const uint32_t K = hb->num_clusters;
    RPVG_REQUIRE(hb->cluster_row_off && hb->cluster_path_off, "rpvg_hip_batch_upload: cluster offsets are NULL");
    const uint64_t R = hb->cluster_row_off[K];
    const uint64_t P = hb->cluster_path_off[K];
    const bool counts = R > 0 && countForm(hb);  // one byte per row and group instead of the offsets (include/rpvg_batch.h)
    RPVG_REQUIRE(R == 0 || ((hb->row_count || hb->row_count8) && (hb->row_noise || (hb->row_noise16 && hb->row_noise_table)) && (counts || ((hb->row_grp_off || hb->row_grp_off32) && (hb->grp_idx_off || hb->grp_idx_off32)))),
                 "rpvg_hip_batch_upload: row arrays are NULL");
    RPVG_REQUIRE(!counts || (hb->num_groups < 0xffffffffull && hb->num_entries < 0xffffffffull && hb->num_groups > 0),
                 "rpvg_hip_batch_upload: counts of one byte come with their totals (num_groups, num_entries: below 2^32 - 1)");
    const uint64_t G = counts ? hb->num_groups : (R ? rowGroupOffset(hb, R) : 0);
    const uint64_t NNZ = counts ? hb->num_entries : (G ? groupEntryOffset(hb, G) : 0);
    RPVG_REQUIRE(G == 0 || hb->grp_prob, "rpvg_hip_batch_upload: grp_prob is NULL");
    RPVG_REQUIRE(NNZ == 0 || hb->path_idx || hb->path_idx16, "rpvg_hip_batch_upload: path_idx is NULL");
    RPVG_REQUIRE(!hb->row_count8 || hb->num_row_count_escapes == 0 || (hb->row_count_escape_row && hb->row_count_escape_count),
                 "rpvg_hip_batch_upload: row_count8 comes with the list of the rows whose count does not fit a byte");

    std::unique_ptr<HostScope> scope(new HostScope("batch_upload: host checks + offsets"));
    // validation: the cluster offsets here (O(K)); the rows and entries on the device, behind their copy (validateRowsKernel)
    for (uint32_t k = 0; k < K; ++k) {
        RPVG_REQUIRE(hb->cluster_row_off[k] <= hb->cluster_row_off[k + 1] && hb->cluster_path_off[k] <= hb->cluster_path_off[k + 1],
                     "rpvg_hip_batch_upload: cluster %u has decreasing offsets", k);
        RPVG_REQUIRE(hb->cluster_path_off[k + 1] - hb->cluster_path_off[k] <= 0x7fffffffu, "rpvg_hip_batch_upload: cluster %u has too many paths", k);
    }
    RPVG_REQUIRE(hb->cluster_row_off[0] == 0, "rpvg_hip_batch_upload: the first cluster does not start at row 0");

    std::lock_guard<std::mutex> lock(ctx->mutex);
    RPVG_HIP_CHECK(hipSetDevice(ctx->device));

    rpvg_hip_batch * b = new (std::nothrow) rpvg_hip_batch();
    if (!b) {
        setError("rpvg_hip_batch_upload: out of host memory");
        return RPVG_HIP_ERR_ALLOC;
    }
    b->num_clusters = K;
    b->num_rows = R;
    b->num_entries = NNZ;
    b->num_paths = P;
    b->h_cluster_row_off.assign(hb->cluster_row_off, hb->cluster_row_off + K + 1);
    b->h_cluster_path_off.assign(hb->cluster_path_off, hb->cluster_path_off + K + 1);
    b->h_cluster_ent_off.resize(K + 1);
    for (uint32_t k = 0; k <= K && !counts; ++k) {  // (with the counts: from the device, behind their sums — uploadFinish)
        const uint64_t r = hb->cluster_row_off[k];
        b->h_cluster_ent_off[k] = R ? groupEntryOffset(hb, rowGroupOffset(hb, r)) : 0;
    }
    b->upload.reset(new rpvg_hip_batch::UploadInProgress());
    rpvg_hip_batch::UploadInProgress & up = *b->upload;
    up.num_groups = G;
    scope.reset(new HostScope("batch_upload: copies queued"));

    const int span = ctx->spanBegin(FAM_H2D);
    hipError_t e = hipSuccess;
    auto ok = [&](hipError_t r) { if (e == hipSuccess) e = r; return e == hipSuccess; };
    ok(b->cluster_row_off.upload(hb->cluster_row_off, K + 1, ctx->stream));
    ok(b->cluster_path_off.upload(hb->cluster_path_off, K + 1, ctx->stream));
    if (hb->row_noise16) {  // (looked up behind the copy: widenNarrowKernel; an index outside the table becomes a noise of -1 and fails the validation)
        ok(up.d_row_noise16.upload(hb->row_noise16, R, ctx->stream));
        ok(up.d_row_noise_table.upload(hb->row_noise_table, hb->num_row_noise_values, ctx->stream));
        ok(b->row_noise.alloc(R));
    } else {
        ok(b->row_noise.upload(hb->row_noise, R, ctx->stream));
    }
    if (hb->row_count8) {  // (widened behind the copy, the listed rows written over: widenNarrowKernel)
        ok(up.d_row_count8.upload(hb->row_count8, R, ctx->stream));
        ok(up.d_escape_row.upload(hb->row_count_escape_row, hb->num_row_count_escapes, ctx->stream));
        ok(up.d_escape_count.upload(hb->row_count_escape_count, hb->num_row_count_escapes, ctx->stream));
        ok(up.d_row_count_u32.alloc(R));
    } else {
        ok(up.d_row_count_u32.upload(hb->row_count, R, ctx->stream));
    }
    const uint64_t zero_off[1] = {0};
    // (the 32-bit forms travel as they are and are widened on the device, behind the copy)
    const bool narrow_offsets = R && G && hb->row_grp_off32 && hb->grp_idx_off32;
    if (counts) {
        ok(up.d_row_grp_count8.upload(hb->row_grp_count8, R, ctx->stream));
        ok(up.d_grp_idx_count8.upload(hb->grp_idx_count8, G, ctx->stream));
        ok(up.d_row_grp_off32.alloc(R + 1));
        ok(up.d_grp_idx_off32.alloc(G + 1));
    } else if (R && hb->row_grp_off32) {
        ok(up.d_row_grp_off32.upload(hb->row_grp_off32, R + 1, ctx->stream));
        if (!narrow_offsets) ok(up.d_row_grp_off.alloc(R + 1));
    } else {
        ok(up.d_row_grp_off.upload(R ? hb->row_grp_off : zero_off, R + 1, ctx->stream));
    }
    if (counts) {
        // (both arrays above)
    } else if (G && hb->grp_idx_off32) {
        ok(up.d_grp_idx_off32.upload(hb->grp_idx_off32, G + 1, ctx->stream));
        if (!narrow_offsets) ok(up.d_grp_idx_off.alloc(G + 1));
    } else {
        ok(up.d_grp_idx_off.upload(G ? hb->grp_idx_off : zero_off, G + 1, ctx->stream));
    }
    ok(up.d_grp_prob.upload(hb->grp_prob, G, ctx->stream));
    if (hb->path_idx16) {
        ok(up.d_path_idx16.upload(hb->path_idx16, NNZ, ctx->stream));
        ok(b->ent_path.alloc(NNZ));
    } else {
        ok(b->ent_path.upload(hb->path_idx, NNZ, ctx->stream));
    }
    ok(b->ent_prob.alloc(NNZ));
    ok(b->row_count.alloc(R));
    ok(b->row_ent_off.alloc(R + 1));
    // the path side, when the caller handed it in: PathInfo::group_id and source_ids (path_sources.hip)
    if (e == hipSuccess) ok(queuePathSourceCopies(ctx, b, hb, up.path_sources));
    ctx->spanEnd(span);
    ctx->stats.h2d_bytes += static_cast<double>((K + 1) * 16 + (hb->row_noise16 ? R * 2 + 8 * hb->num_row_noise_values : R * 8) + (hb->row_count8 ? R + 8 * hb->num_row_count_escapes : 4 * R) +
                                                (counts ? R : (R + 1) * (hb->row_grp_off32 ? 4 : 8)) +
                                                (counts ? G : (G + 1) * (hb->grp_idx_off32 ? 4 : 8)) + G * 8 + NNZ * (hb->path_idx16 ? 2 : 4));
    if (e != hipSuccess) {
        setError("rpvg_hip_batch_upload: %s", hipGetErrorString(e));
        (void) hipStreamSynchronize(ctx->stream);
        delete b;
        return (e == hipErrorOutOfMemory) ? RPVG_HIP_ERR_ALLOC : RPVG_HIP_ERR_RUNTIME;
    }
    *batch_out = b;
    return RPVG_HIP_OK;
}

// The second half of an upload in two steps.  Queue: the kernels behind the copies and the copies of their small results into a
// page-locked block, on stream `st` of `ctx` (the copies have been waited for: rpvg_hip_batch_upload_begin), an event behind them.
// Wait: for that event, from any thread, and the host's part (messages, sizes of the haplotype columns).  A pipeline's uploader
// queues them behind every batch's copies on its side stream and goes on copying; the estimator that takes the batch finds them
// done, or nearly (rpvg_amd/host/batch_pipeline.hpp).  (the caller holds no lock; `b` is deleted on failure)
static int uploadFinishQueue(rpvg_hip_ctx * ctx, rpvg_hip_batch * b, hipStream_t st) {
    const uint32_t K = b->num_clusters;
    const uint64_t R = b->num_rows, NNZ = b->num_entries;
    // (the context's lock for its main stream — and its statistics — only: on the side stream of an uploader's context a thread of
    // its own queues these while the uploader queues the next batch's copies)
    const bool own_stream = st == ctx->stream;
    std::unique_lock<std::mutex> lock(ctx->mutex, std::defer_lock);
    if (own_stream) lock.lock();
    hipError_t e = hipSetDevice(ctx->device);
    rpvg_hip_batch::UploadInProgress & up = *b->upload;
    const uint64_t G = up.num_groups;
    HostScope scope("batch_upload: kernels queued");
    // (on an uploader's side stream the span is opened and closed under a short hold of the context's lock, and only by a context
    // that times every kernel family — RPVG_HIP_SPANS=2, bench.py's instrumented pass: the thread that queues these stands behind
    // the uploader's copies for it)
    const bool side_span = !own_stream && ctx->span_level >= 2;
    int bspan = -1;
    if (own_stream) {
        bspan = ctx->spanBegin(FAM_BUILD, st);
    } else if (side_span) {
        std::lock_guard<std::mutex> span_lock(ctx->mutex);
        bspan = ctx->spanBegin(FAM_BUILD, st);
    }
    // both long offset arrays in 32 bits (what a caller that flattens rows for the GPU writes): the kernels read them as they are;
    // one of them only: that one is widened first
    const bool narrow = up.d_row_grp_off32.ptr && up.d_grp_idx_off32.ptr;
    if (e == hipSuccess && !narrow && up.d_row_grp_off32.ptr) {
        widenOffsetsKernel<<<dim3(static_cast<uint32_t>((R + 1 + 255) / 256)), dim3(256), 0, st>>>(R + 1, up.d_row_grp_off32.ptr, up.d_row_grp_off.ptr);
    }
    if (e == hipSuccess && !narrow && up.d_grp_idx_off32.ptr) {
        widenOffsetsKernel<<<dim3(static_cast<uint32_t>((G + 1 + 255) / 256)), dim3(256), 0, st>>>(G + 1, up.d_grp_idx_off32.ptr, up.d_grp_idx_off.ptr);
    }
    // the narrow forms of the copy (include/rpvg_batch.h): path indices, read counts and source ids back in 32 bits
    if (e == hipSuccess && (up.d_path_idx16.ptr || up.d_row_count8.ptr || up.path_sources.d_source_id16.ptr || up.d_row_noise16.ptr)) {
        const uint64_t S16 = up.path_sources.d_source_id16.ptr ? up.path_sources.num_sources_narrow : 0;
        const uint64_t most = std::max<uint64_t>(std::max<uint64_t>(up.d_path_idx16.ptr ? NNZ : 0, (up.d_row_count8.ptr || up.d_row_noise16.ptr) ? R : 0), S16);
        widenNarrowKernel<<<dim3(static_cast<uint32_t>(std::min<uint64_t>((most + 1023) / 1024 + 1, 4096))), dim3(256), 0, st>>>(
            up.d_path_idx16.ptr, b->ent_path.ptr, up.d_path_idx16.ptr ? NNZ : 0, up.d_row_count8.ptr, up.d_row_count_u32.ptr, up.d_row_count8.ptr ? R : 0,
            up.path_sources.d_source_id16.ptr, up.path_sources.d_source_id.ptr, S16,
            up.d_row_noise16.ptr, up.d_row_noise_table.ptr, static_cast<uint32_t>(up.d_row_noise_table.count), b->row_noise.ptr, up.d_row_noise16.ptr ? R : 0);
        if (up.d_escape_row.count) {
            escapeRowCountsKernel<<<dim3(static_cast<uint32_t>((up.d_escape_row.count + 255) / 256)), dim3(256), 0, st>>>(
                up.d_escape_row.ptr, up.d_escape_count.ptr, up.d_escape_row.count, R, up.d_row_count_u32.ptr);
        }
    }
    const uint32_t threads = 256;
    const dim3 group_grid(static_cast<uint32_t>((G + threads - 1) / threads)), meta_grid(static_cast<uint32_t>((R + 1 + threads - 1) / threads)),
        row_grid(static_cast<uint32_t>((R + threads - 1) / threads));
    if (e == hipSuccess) e = up.d_first_bad_row.alloc(1);
    if (e == hipSuccess) e = hipMemsetAsync(up.d_first_bad_row.ptr, 0xFF, sizeof(unsigned long long), st);
    up.counts = up.d_row_grp_count8.ptr != nullptr;
    const bool counts = up.counts;
    if (e == hipSuccess && counts) {  // the offsets the kernels below read: the counts' running sums
        e = queueOffsetsFromCounts(st, up.d_row_grp_count8.ptr, R, up.d_row_grp_off32.ptr, up.scan_scratch_rows);
        if (e == hipSuccess) e = queueOffsetsFromCounts(st, up.d_grp_idx_count8.ptr, G, up.d_grp_idx_off32.ptr, up.scan_scratch_groups);
    }
    if (e == hipSuccess && narrow) {
        if (G > 0) expandGroupsKernel<uint32_t><<<group_grid, dim3(threads), 0, st>>>(G, NNZ, up.d_grp_idx_off32.ptr, up.d_grp_prob.ptr, b->ent_prob.ptr);
        rowMetaKernel<uint32_t, uint32_t><<<meta_grid, dim3(threads), 0, st>>>(R, G, up.d_row_grp_off32.ptr, up.d_grp_idx_off32.ptr, up.d_row_count_u32.ptr,
                                                                                b->row_ent_off.ptr, b->row_count.ptr);
        if (R > 0) validateRowsKernel<uint32_t, uint32_t><<<row_grid, dim3(threads), 0, st>>>(
            R, G, NNZ, K, b->cluster_row_off.ptr, b->cluster_path_off.ptr, up.d_row_grp_off32.ptr, up.d_grp_idx_off32.ptr, b->row_noise.ptr, b->ent_path.ptr,
            up.d_first_bad_row.ptr);
    } else if (e == hipSuccess) {
        if (G > 0) expandGroupsKernel<uint64_t><<<group_grid, dim3(threads), 0, st>>>(G, NNZ, up.d_grp_idx_off.ptr, up.d_grp_prob.ptr, b->ent_prob.ptr);
        rowMetaKernel<uint64_t, uint64_t><<<meta_grid, dim3(threads), 0, st>>>(R, G, up.d_row_grp_off.ptr, up.d_grp_idx_off.ptr, up.d_row_count_u32.ptr,
                                                                                b->row_ent_off.ptr, b->row_count.ptr);
        if (R > 0) validateRowsKernel<uint64_t, uint64_t><<<row_grid, dim3(threads), 0, st>>>(
            R, G, NNZ, K, b->cluster_row_off.ptr, b->cluster_path_off.ptr, up.d_row_grp_off.ptr, up.d_grp_idx_off.ptr, b->row_noise.ptr, b->ent_path.ptr,
            up.d_first_bad_row.ptr);
    }
    // read counts per cluster (the host summed three million of them per batch with a team of its own)
    if (e == hipSuccess) e = up.d_cluster_total.alloc(K);
    if (e == hipSuccess) e = queueClusterTotals(st, K, b->cluster_row_off.ptr, up.d_row_count_u32.ptr, up.d_cluster_total.ptr);
    if (e == hipSuccess) e = queuePathSourceKernels(ctx, b, up.path_sources, st);
    if (e == hipSuccess && counts) {
        e = up.d_cluster_ent_off.alloc(K + 1);
        if (e == hipSuccess) clusterEntryOffsetsKernel<<<dim3((K + 1 + 255) / 256), dim3(256), 0, st>>>(K, b->cluster_row_off.ptr, b->row_ent_off.ptr, up.d_cluster_ent_off.ptr);
    }
    if (own_stream) {
        ctx->spanEnd(bspan);
        ctx->stats.build_launches += 5;
    } else if (side_span) {
        std::lock_guard<std::mutex> span_lock(ctx->mutex);
        ctx->spanEnd(bspan);
        ctx->stats.build_launches += 5;
    }
    if (e == hipSuccess) e = hipGetLastError();
    // the small results: [first bad row, 64 bits | the counts' two sums | - | read totals K doubles | entry offsets K + 1]
    const size_t result_bytes = 16 + 8 * static_cast<size_t>(K) + 8 * (static_cast<size_t>(K) + 1);
    if (e == hipSuccess) e = pinnedAlloc(&up.h_results, result_bytes);
    if (e == hipSuccess) {
        unsigned char * host = static_cast<unsigned char *>(up.h_results);
        memset(host, 0, 16);
        e = hipMemcpyAsync(host, up.d_first_bad_row.ptr, sizeof(unsigned long long), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess && counts) e = hipMemcpyAsync(host + 8, up.d_row_grp_off32.ptr + R, sizeof(uint32_t), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess && counts) e = hipMemcpyAsync(host + 12, up.d_grp_idx_off32.ptr + G, sizeof(uint32_t), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess && K > 0) e = hipMemcpyAsync(host + 16, up.d_cluster_total.ptr, K * sizeof(double), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess && counts) e = hipMemcpyAsync(host + 16 + 8 * static_cast<size_t>(K), up.d_cluster_ent_off.ptr, (K + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, st);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&up.finished, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(up.finished, st);
    if (e != hipSuccess) {
        setError("rpvg_hip_batch_upload: %s", hipGetErrorString(e));
        (void) hipStreamSynchronize(st);
        delete b;
        return RPVG_HIP_ERR_RUNTIME;
    }
    return RPVG_HIP_OK;
}

static int uploadFinishWait(rpvg_hip_batch * b, const rpvg_cluster_batch * hb) {
    const uint32_t K = b->num_clusters;
    const uint64_t NNZ = b->num_entries;
    rpvg_hip_batch::UploadInProgress & up = *b->upload;
    const uint64_t G = up.num_groups;
    HostScope scope("batch_upload: wait for the kernels");
    const hipError_t e = waitEvent(up.finished);
    if (e != hipSuccess) {
        setError("rpvg_hip_batch_upload: %s", hipGetErrorString(e));
        delete b;
        return RPVG_HIP_ERR_RUNTIME;
    }
    const unsigned char * host = static_cast<const unsigned char *>(up.h_results);
    unsigned long long first_bad_row = ~0ull;
    uint32_t count_totals[2] = {0, 0};  // what the counts add up to (against the totals the caller named)
    memcpy(&first_bad_row, host, sizeof(first_bad_row));
    memcpy(count_totals, host + 8, sizeof(count_totals));
    b->h_cluster_total.resize(K);
    if (K > 0) memcpy(b->h_cluster_total.data(), host + 16, K * sizeof(double));
    if (up.counts) memcpy(b->h_cluster_ent_off.data(), host + 16 + 8 * static_cast<size_t>(K), (K + 1) * sizeof(uint64_t));
    if (up.counts && (count_totals[0] != G || count_totals[1] != NNZ)) {
        delete b;
        setError("rpvg_hip_batch_upload: the counts of the rows' groups and of the groups' paths do not add up to num_groups = %llu and num_entries = %llu",
                 static_cast<unsigned long long>(G), static_cast<unsigned long long>(NNZ));
        return RPVG_HIP_ERR_INVALID;
    }
    if (first_bad_row != ~0ull) {  // the message: the host's reading of the offending row's cluster
        delete b;
        const uint32_t k = static_cast<uint32_t>(std::upper_bound(hb->cluster_row_off, hb->cluster_row_off + K + 1, first_bad_row) - hb->cluster_row_off) - 1;
        char message[kProblemChars];
        const uint64_t g0 = rowGroupOffset(hb, first_bad_row), g1 = rowGroupOffset(hb, first_bad_row + 1);
        bool offsets_ok = g0 <= g1 && g1 <= G;
        for (uint64_t g = g0; offsets_ok && g < g1; ++g) offsets_ok = groupEntryOffset(hb, g) <= groupEntryOffset(hb, g + 1) && groupEntryOffset(hb, g + 1) <= NNZ;
        if (!offsets_ok || validateRow(hb, k, first_bad_row, message)) {
            std::snprintf(message, kProblemChars, "rpvg_hip_batch_upload: row %llu has inconsistent group or entry offsets", first_bad_row);
        }
        setError("%s", message);
        return RPVG_HIP_ERR_INVALID;
    }
    const int sources_rc = finishPathSources(b, up.path_sources);
    if (sources_rc != RPVG_HIP_OK) {
        delete b;
        return sources_rc;
    }
    b->upload.reset();
    return RPVG_HIP_OK;
}

static int uploadFinish(rpvg_hip_ctx * ctx, rpvg_hip_batch * b, const rpvg_cluster_batch * hb) {
    const int rc = uploadFinishQueue(ctx, b, ctx->stream);
    return rc != RPVG_HIP_OK ? rc : uploadFinishWait(b, hb);
}

int rpvg_hip_batch_upload(rpvg_hip_ctx * ctx, const rpvg_cluster_batch * hb, rpvg_hip_batch ** batch_out) {
    rpvg_hip_batch * b = nullptr;
    int rc = uploadBegin(ctx, hb, &b);
    if (rc != RPVG_HIP_OK) return rc;
    rc = uploadFinish(ctx, b, hb);  // (same stream: behind the copies)
    if (rc != RPVG_HIP_OK) return rc;
    *batch_out = b;
    return RPVG_HIP_OK;
}

int rpvg_hip_batch_upload_begin(rpvg_hip_ctx * ctx, const rpvg_cluster_batch * hb, rpvg_hip_batch ** batch_out) {
    rpvg_hip_batch * b = nullptr;
    const int rc = uploadBegin(ctx, hb, &b);
    if (rc != RPVG_HIP_OK) return rc;
    // the copies are done: any context of the device may finish the batch.  (An event, waited for without the context's lock:
    // another thread may queue the kernels behind an earlier batch's copies on this context's side stream meanwhile.)
    hipError_t e = hipSuccess;
    hipEvent_t copied = nullptr;
    {
        std::lock_guard<std::mutex> lock(ctx->mutex);
        e = hipSetDevice(ctx->device);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&copied, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventRecord(copied, ctx->stream);
    }
    if (e == hipSuccess) e = waitEvent(copied);
    if (copied) (void) hipEventDestroy(copied);
    if (e != hipSuccess) {
        setError("rpvg_hip_batch_upload_begin: %s", hipGetErrorString(e));
        delete b;
        return RPVG_HIP_ERR_RUNTIME;
    }
    *batch_out = b;
    return RPVG_HIP_OK;
}

int rpvg_hip_batch_upload_finish_queue(rpvg_hip_ctx * ctx, rpvg_hip_batch * batch, const rpvg_cluster_batch * hb) {
    RPVG_REQUIRE(ctx != nullptr && batch != nullptr && hb != nullptr, "rpvg_hip_batch_upload_finish_queue: NULL argument");
    RPVG_REQUIRE(batch->upload != nullptr && batch->upload->finished == nullptr, "rpvg_hip_batch_upload_finish_queue: the batch is complete, or queued, already");
    RPVG_REQUIRE(hb->num_clusters == batch->num_clusters && hb->cluster_row_off && hb->cluster_row_off[hb->num_clusters] == batch->num_rows,
                 "rpvg_hip_batch_upload_finish_queue: not the host batch the upload began with");
    // (an uploader's context: its side stream — its main stream carries the next batch's copies)
    return uploadFinishQueue(ctx, batch, ctx->aux_count > 0 && ctx->aux[0] ? ctx->aux[0] : ctx->stream);
}

int rpvg_hip_batch_upload_finish_wait(rpvg_hip_batch * batch, const rpvg_cluster_batch * hb) {
    RPVG_REQUIRE(batch != nullptr && hb != nullptr, "rpvg_hip_batch_upload_finish_wait: NULL argument");
    RPVG_REQUIRE(batch->upload != nullptr && batch->upload->finished != nullptr, "rpvg_hip_batch_upload_finish_wait: nothing queued for this batch");
    return uploadFinishWait(batch, hb);
}

int rpvg_hip_batch_upload_finish(rpvg_hip_ctx * ctx, rpvg_hip_batch * batch, const rpvg_cluster_batch * hb) {
    RPVG_REQUIRE(ctx != nullptr && batch != nullptr && hb != nullptr, "rpvg_hip_batch_upload_finish: NULL argument");
    RPVG_REQUIRE(batch->upload != nullptr, "rpvg_hip_batch_upload_finish: the batch is complete already");
    RPVG_REQUIRE(hb->num_clusters == batch->num_clusters && hb->cluster_row_off && hb->cluster_row_off[hb->num_clusters] == batch->num_rows,
                 "rpvg_hip_batch_upload_finish: not the host batch the upload began with");
    return uploadFinish(ctx, batch, hb);
}

void rpvg_hip_thread_wait_spin_us(uint32_t microseconds) { t_spin_us = microseconds; }

int rpvg_hip_pinned_alloc(uint64_t bytes, void ** host_out) {
    RPVG_REQUIRE(host_out != nullptr, "rpvg_hip_pinned_alloc: host_out is NULL");
    *host_out = nullptr;
    const hipError_t e = pinnedAlloc(host_out, std::max<uint64_t>(bytes, 8));
    if (e != hipSuccess) {
        setError("rpvg_hip_pinned_alloc: %llu bytes: %s", static_cast<unsigned long long>(bytes), hipGetErrorString(e));
        return RPVG_HIP_ERR_ALLOC;
    }
    return RPVG_HIP_OK;
}

void rpvg_hip_pinned_free(void * host) { pinnedFree(host); }

int rpvg_hip_batch_upload_segments(rpvg_hip_ctx * ctx, const rpvg_cluster_segment * segments, uint32_t K, rpvg_hip_batch ** batch_out) {
    RPVG_REQUIRE(ctx != nullptr && batch_out != nullptr && (segments != nullptr || K == 0), "rpvg_hip_batch_upload_segments: NULL argument");
    *batch_out = nullptr;
    std::unique_ptr<HostScope> scope(new HostScope("batch_upload_segments: host checks + table"));
    const bool with_columns = K > 0 && segments[0].has_columns != 0;
    const bool with_paths = K > 0 && segments[0].has_paths != 0 && !with_columns;
    uint64_t R = 0, NNZ = 0, P = 0, S = 0, most_work = 0;
    for (uint32_t k = 0; k < K; ++k) {
        const rpvg_cluster_segment & g = segments[k];
        RPVG_REQUIRE(g.base != nullptr && pinnedCapacity(g.base) >= g.bytes, "rpvg_hip_batch_upload_segments: segment %u does not lie in a block of rpvg_hip_pinned_alloc", k);
        RPVG_REQUIRE((g.has_columns != 0) == with_columns, "rpvg_hip_batch_upload_segments: segment %u: all segments of a batch carry their haplotype columns, or none", k);
        RPVG_REQUIRE(with_columns || (g.has_paths != 0) == with_paths, "rpvg_hip_batch_upload_segments: segment %u: all segments of a batch carry their paths, or none", k);
        RPVG_REQUIRE(g.num_paths <= 0x7fffffffu, "rpvg_hip_batch_upload_segments: segment %u has too many paths", k);
        auto inside = [&](const uint64_t at, const uint64_t count, const uint64_t width) { return (at & 7) == 0 && at <= g.bytes && count * width <= g.bytes - at; };
        bool fits = inside(g.row_count_at, g.num_rows, 4) && inside(g.row_noise_at, g.num_rows, 8) && inside(g.row_grp_off_at, static_cast<uint64_t>(g.num_rows) + 1, 4) &&
                    inside(g.grp_idx_off_at, static_cast<uint64_t>(g.num_groups) + 1, 4) && inside(g.grp_prob_at, g.num_groups, 8) && inside(g.path_idx_at, g.num_entries, 4);
        if (with_paths) {
            fits = fits && inside(g.path_group_id_at, g.num_paths, 4) && inside(g.path_source_off_at, static_cast<uint64_t>(g.num_paths) + 1, 4) && inside(g.source_id_at, g.num_sources, 4);
        }
        if (with_columns) {
            fits = fits && inside(g.path_group_id_at, g.num_paths, 4) && inside(g.col_count_at, g.num_columns, 4) && inside(g.col_end_at, g.num_columns, 4) &&
                   inside(g.col_path_at, g.num_column_paths, 4) && g.num_columns <= g.num_column_paths && g.max_column_paths <= g.num_column_paths &&
                   (g.num_columns > 0) == (g.num_column_paths > 0);
        }
        RPVG_REQUIRE(fits, "rpvg_hip_batch_upload_segments: an array of segment %u is misaligned or outside its block", k);
        RPVG_REQUIRE(g.num_rows > 0 || (g.num_groups == 0 && g.num_entries == 0), "rpvg_hip_batch_upload_segments: segment %u has groups without rows", k);
        R += g.num_rows;
        NNZ += g.num_entries;
        P += g.num_paths;
        const uint32_t slots = with_columns ? g.num_column_paths : (with_paths ? g.num_sources : 0u);
        S += slots;
        most_work = std::max<uint64_t>(most_work, std::max<uint64_t>(std::max(g.num_rows, g.num_groups), std::max(g.num_entries, slots)));
    }
    RPVG_REQUIRE(NNZ < 0xffffffffull, "rpvg_hip_batch_upload_segments: a batch of 2^32 - 1 entries or more");

    std::unique_lock<std::mutex> lock(ctx->mutex);
    RPVG_HIP_CHECK(hipSetDevice(ctx->device));
    std::unique_ptr<rpvg_hip_batch> b(new (std::nothrow) rpvg_hip_batch());
    if (!b) {
        setError("rpvg_hip_batch_upload_segments: out of host memory");
        return RPVG_HIP_ERR_ALLOC;
    }
    b->num_clusters = K;
    b->num_rows = R;
    b->num_entries = NNZ;
    b->num_paths = P;
    b->h_cluster_row_off.assign(K + 1, 0);
    b->h_cluster_path_off.assign(K + 1, 0);
    b->h_cluster_ent_off.assign(K + 1, 0);
    b->h_cluster_total.resize(K);
    if (with_paths || with_columns) b->h_cluster_src_off.assign(K + 1, 0);
    if (with_columns) {
        b->h_src_num_cols.resize(K);
        b->h_src_col_paths.resize(K);
        b->h_src_max_col_paths.resize(K);
    }
    b->upload.reset(new rpvg_hip_batch::UploadInProgress());
    rpvg_hip_batch::UploadInProgress & up = *b->upload;
    hipStream_t st = ctx->stream;
    hipError_t e = hipSuccess;
    auto ok = [&](hipError_t r) { if (e == hipSuccess) e = r; return e == hipSuccess; };
    // the table, and behind it the two words that come back (first invalid segment)
    const size_t table_bytes = std::max<size_t>(K, 1) * sizeof(SegmentEntry);
    void * h_table = nullptr;
    if (pinnedAlloc(&h_table, table_bytes + 16) != hipSuccess) {
        setError("rpvg_hip_batch_upload_segments: out of page-locked host memory");
        return RPVG_HIP_ERR_ALLOC;
    }
    up.h_results = h_table;  // (freed with the upload)
    SegmentEntry * table = static_cast<SegmentEntry *>(h_table);
    for (uint32_t k = 0; k < K; ++k) {
        const rpvg_cluster_segment & g = segments[k];
        const unsigned char * base = static_cast<const unsigned char *>(g.base);
        SegmentEntry & t = table[k];
        t.row_count = reinterpret_cast<const uint32_t *>(base + g.row_count_at);
        t.row_noise = reinterpret_cast<const double *>(base + g.row_noise_at);
        t.row_grp_off = reinterpret_cast<const uint32_t *>(base + g.row_grp_off_at);
        t.grp_idx_off = reinterpret_cast<const uint32_t *>(base + g.grp_idx_off_at);
        t.grp_prob = reinterpret_cast<const double *>(base + g.grp_prob_at);
        t.path_idx = reinterpret_cast<const uint32_t *>(base + g.path_idx_at);
        t.path_group_id = (with_paths || with_columns) ? reinterpret_cast<const uint32_t *>(base + g.path_group_id_at) : nullptr;
        t.col_count = with_columns ? reinterpret_cast<const uint32_t *>(base + g.col_count_at) : nullptr;
        t.col_end = with_columns ? reinterpret_cast<const uint32_t *>(base + g.col_end_at) : nullptr;
        t.col_path = with_columns ? reinterpret_cast<const uint32_t *>(base + g.col_path_at) : nullptr;
        t.path_source_off = with_paths ? reinterpret_cast<const uint32_t *>(base + g.path_source_off_at) : nullptr;
        t.source_id = with_paths ? reinterpret_cast<const uint32_t *>(base + g.source_id_at) : nullptr;
        t.R = g.num_rows;
        t.G = g.num_groups;
        t.NNZ = g.num_entries;
        t.P = g.num_paths;
        t.S = with_columns ? g.num_column_paths : (with_paths ? g.num_sources : 0);
        t.C = with_columns ? g.num_columns : 0;
        t.row_base = b->h_cluster_row_off[k];
        t.ent_base = b->h_cluster_ent_off[k];
        t.path_base = b->h_cluster_path_off[k];
        t.src_base = (with_paths || with_columns) ? b->h_cluster_src_off[k] : 0;
        b->h_cluster_row_off[k + 1] = t.row_base + t.R;
        b->h_cluster_ent_off[k + 1] = t.ent_base + t.NNZ;
        b->h_cluster_path_off[k + 1] = t.path_base + t.P;
        if (with_paths || with_columns) b->h_cluster_src_off[k + 1] = t.src_base + t.S;
        if (with_columns) {
            b->h_src_num_cols[k] = g.num_columns;
            b->h_src_col_paths[k] = g.num_column_paths;
            b->h_src_max_col_paths[k] = g.max_column_paths;
        }
        b->h_cluster_total[k] = static_cast<double>(g.total_read_count);
    }
    unsigned long long * h_first_bad = reinterpret_cast<unsigned long long *>(static_cast<unsigned char *>(h_table) + table_bytes);
    *h_first_bad = ~0ull;

    scope.reset(new HostScope("batch_upload_segments: kernels queued"));
    ok(b->cluster_row_off.alloc(K + 1));
    ok(b->cluster_path_off.alloc(K + 1));
    ok(b->row_count.alloc(R));
    ok(b->row_noise.alloc(R));
    ok(b->row_ent_off.alloc(R + 1));
    ok(b->ent_path.alloc(NNZ));
    ok(b->ent_prob.alloc(NNZ));
    ok(up.d_first_bad_row.alloc(1));
    if (e == hipSuccess && with_paths) ok(reservePathSources(b.get(), K, P, S, up.path_sources));
    const bool sources = with_paths && up.path_sources.copied;
    const bool columns = with_columns && P > 0 && S > 0;
    if (e == hipSuccess && columns) {  // the columns come with the segments: their slots, nothing to form
        ok(b->path_group_id.alloc(P));
        ok(b->cluster_src_off.alloc(K + 1));
        ok(b->src_col_count.alloc(S));
        ok(b->src_col_end.alloc(S));
        ok(b->src_col_path.alloc(S));
    }
    if (e == hipSuccess) ok(hipMemsetAsync(up.d_first_bad_row.ptr, 0xFF, sizeof(unsigned long long), st));
    const int span = ctx->spanBegin(FAM_BUILD, st);
    if (e == hipSuccess && K > 0) {
        SegmentGatherArgs a;
        a.table = table;
        a.num_clusters = K;
        a.with_paths = sources ? 1 : 0;
        a.cluster_row_off = b->cluster_row_off.ptr;
        a.cluster_path_off = b->cluster_path_off.ptr;
        a.cluster_src_off = (sources || columns) ? b->cluster_src_off.ptr : nullptr;
        a.with_columns = columns ? 1 : 0;
        a.src_col_count = columns ? b->src_col_count.ptr : nullptr;
        a.src_col_end = columns ? b->src_col_end.ptr : nullptr;
        a.src_col_path = columns ? b->src_col_path.ptr : nullptr;
        a.row_count = b->row_count.ptr;
        a.row_noise = b->row_noise.ptr;
        a.row_ent_off = b->row_ent_off.ptr;
        a.ent_path = b->ent_path.ptr;
        a.ent_prob = b->ent_prob.ptr;
        a.path_group_id = (sources || columns) ? b->path_group_id.ptr : nullptr;
        a.path_source_off = sources ? up.path_sources.d_path_source_off.ptr : nullptr;
        a.source_id = sources ? up.path_sources.d_source_id.ptr : nullptr;
        a.first_bad = up.d_first_bad_row.ptr;
        // slices per cluster: four items of the largest cluster's longest array per thread, 64 at the most (a cluster of a million
        // rows: sixty passes of 16 384 threads)
        const uint32_t slices = static_cast<uint32_t>(std::min<uint64_t>(64, std::max<uint64_t>(1, (most_work + 1023) / 1024)));
        segmentsGatherKernel<<<dim3(K, slices), dim3(256), 0, st>>>(a);
        ok(hipGetLastError());
    }
    if (e == hipSuccess && sources) ok(queuePathSourceKernels(ctx, b.get(), up.path_sources, st));
    ctx->spanEnd(span);
    ctx->stats.build_launches += sources ? 3 : 1;
    ctx->stats.h2d_bytes += static_cast<double>(R * 16 + NNZ * 4 + P * 8 + S * 4);  // (what the kernel pulls: no copy command)
    if (e == hipSuccess) ok(hipMemcpyAsync(h_first_bad, up.d_first_bad_row.ptr, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    if (e == hipSuccess) ok(hipEventCreateWithFlags(&up.finished, hipEventDisableTiming));
    if (e == hipSuccess) ok(hipEventRecord(up.finished, st));
    if (e != hipSuccess) {
        setError("rpvg_hip_batch_upload_segments: %s", hipGetErrorString(e));
        (void) hipStreamSynchronize(st);
        return (e == hipErrorOutOfMemory) ? RPVG_HIP_ERR_ALLOC : RPVG_HIP_ERR_RUNTIME;
    }
    lock.unlock();
    scope.reset(new HostScope("batch_upload_segments: wait for the kernels"));
    e = waitEvent(up.finished);
    if (e != hipSuccess) {
        setError("rpvg_hip_batch_upload_segments: %s", hipGetErrorString(e));
        return RPVG_HIP_ERR_RUNTIME;
    }
    if (*h_first_bad != ~0ull) {
        const unsigned long long k = *h_first_bad >> 8, kind = *h_first_bad & 0xff;
        setError("rpvg_hip_batch_upload_segments: cluster %llu of the batch: %s", k,
                 kind == kSegmentBadNoise ? "a row has a noise probability outside (0, 1]"
                 : kind == kSegmentBadPath ? "a row or a haplotype column refers to a path outside its cluster"
                                           : "inconsistent row, group, entry or source offsets");
        return RPVG_HIP_ERR_INVALID;
    }
    const int sources_rc = finishPathSources(b.get(), up.path_sources);
    if (sources_rc != RPVG_HIP_OK) return sources_rc;
    if (columns) b->has_source_columns = true;
    b->upload.reset();
    *batch_out = b.release();
    return RPVG_HIP_OK;
}

void rpvg_hip_batch_free(rpvg_hip_ctx * ctx, rpvg_hip_batch * batch) {
    if (!batch) return;
    if (ctx) {
        std::lock_guard<std::mutex> lock(ctx->mutex);
        (void) hipSetDevice(ctx->device);
        (void) hipStreamSynchronize(ctx->stream);
        delete batch;
    } else {
        delete batch;
    }
}

int rpvg_hip_stats_get(rpvg_hip_ctx * ctx, rpvg_hip_kernel_stats * stats_out) {
    RPVG_REQUIRE(ctx != nullptr && stats_out != nullptr, "rpvg_hip_stats_get: NULL argument");
    std::lock_guard<std::mutex> lock(ctx->mutex);
    RPVG_HIP_CHECK(hipSetDevice(ctx->device));
    int rc = ctx->foldSpans();
    if (rc != RPVG_HIP_OK) return rc;
    *stats_out = ctx->stats;
    return RPVG_HIP_OK;
}

int rpvg_hip_stats_reset(rpvg_hip_ctx * ctx) {
    RPVG_REQUIRE(ctx != nullptr, "rpvg_hip_stats_reset: ctx is NULL");
    std::lock_guard<std::mutex> lock(ctx->mutex);
    RPVG_HIP_CHECK(hipSetDevice(ctx->device));
    int rc = ctx->foldSpans();
    if (rc != RPVG_HIP_OK) return rc;
    memset(&ctx->stats, 0, sizeof(ctx->stats));
    ctx->intervals.clear();
    (void) deviceClock(ctx->device, ctx->stream, true);
    return RPVG_HIP_OK;
}

int rpvg_hip_stats_intervals(rpvg_hip_ctx * ctx, uint64_t capacity, double * start_ms, double * stop_ms, int32_t * family,
                             uint64_t * count_out) {
    RPVG_REQUIRE(ctx != nullptr && count_out != nullptr, "rpvg_hip_stats_intervals: NULL argument");
    std::lock_guard<std::mutex> lock(ctx->mutex);
    RPVG_HIP_CHECK(hipSetDevice(ctx->device));
    int rc = ctx->foldSpans();
    if (rc != RPVG_HIP_OK) return rc;
    const DeviceClock clock = deviceClock(ctx->device, ctx->stream, false);
    uint64_t n = 0;
    for (auto & t : ctx->intervals) {
        if (t.clock != clock.id) continue;
        if (n < capacity) {
            if (start_ms) start_ms[n] = t.start_ms;
            if (stop_ms) stop_ms[n] = t.stop_ms;
            if (family) family[n] = t.family;
        }
        ++n;
    }
    *count_out = n;
    return RPVG_HIP_OK;
}

}  // extern "C"
