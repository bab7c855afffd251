// Internals shared by the translation units of librpvg_hip.so (gfx950 only).
#ifndef RPVG_HIP_COMMON_HPP
#define RPVG_HIP_COMMON_HPP

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <functional>
#include <vector>

#include "../../include/rpvg_hip.h"

namespace rpvg_hip_detail {

// ---- error reporting --------------------------------------------------------
void setError(const char * fmt, ...);

#define RPVG_HIP_CHECK(expr)                                                                                  \
    do {                                                                                                      \
        hipError_t err__ = (expr);                                                                            \
        if (err__ != hipSuccess) {                                                                            \
            rpvg_hip_detail::setError("%s failed at %s:%d: %s", #expr, __FILE__, __LINE__, hipGetErrorString(err__)); \
            /* the buffers of this call go back to the pool on return: nothing queued may still use them */   \
            (void) hipDeviceSynchronize();                                                                    \
            return RPVG_HIP_ERR_RUNTIME;                                                                      \
        }                                                                                                     \
    } while (0)

#define RPVG_REQUIRE(cond, ...)                       \
    do {                                              \
        if (!(cond)) {                                \
            rpvg_hip_detail::setError(__VA_ARGS__);   \
            return RPVG_HIP_ERR_INVALID;              \
        }                                             \
    } while (0)

// ---- A/B switches and timing experiments -----------------------------------------------------------
// Four rounds of measurements left some forty environment switches behind — alternative launch orders, stream layouts, grid
// sizes, kernels with parts of their work skipped for timing (docs/design/knobs.md).  Several change results or skip work; none
// belongs in a library somebody links.  The shipped build does not read them: the macro is a null pointer there and the
// names are not in the binary.  `make -C rpvg_amd/csrc clean all EXPERIMENTS=1` builds librpvg_hip.so with them
// (-DRPVG_HIP_EXPERIMENTS: in place, for tools/ — rebuild without the variable afterwards).
#ifdef RPVG_HIP_EXPERIMENTS
#define RPVG_EXPERIMENT_ENV(name) std::getenv(name)
#else
#define RPVG_EXPERIMENT_ENV(name) (static_cast<const char *>(nullptr))
#endif

// ---- hardware queues ---------------------------------------------------------------
// The HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (4 unless the environment says otherwise, read
// when the runtime starts); work of streams that share a queue runs in order.  Two host lanes with four busy streams
// each want eight — with four, the uploads and build kernels of the second lane queue behind the search kernels of
// the first (0.7 ms per batch).  The library asks for eight when it is loaded (before the first HIP call of a C++
// host; a Python harness must import rpvg_amd before it touches the GPU).  hardwareQueues() is what the environment
// said when the library was loaded, i.e. what the runtime got if it had not been started yet.  (12 and 16 queues
// measured slower than 8: 13.8-14.5 against 12.6 ms per bench batch — more of the lanes' kernels side by side.)
int hardwareQueues();

// ---- waiting without burning a core ---------------------------------------------------------------
// hipEventSynchronize / hipStreamSynchronize spin: a host lane that waits for its kernels is a core at 100 % (8.5 of the 9.3 ms
// of a configs[2] batch per lane, the bulk of the process's CPU time once findPathSourceGroups and the merge had left the
// host).  These poll instead — the event or stream is queried, the thread sleeps 30 us in between (timer slack of the
// thread lowered to 1 us) after a first 20 us of plain polling for the waits that are nearly over — which costs a wait at
// most one sleep of latency.  RPVG_HIP_SPIN_WAITS=1 restores the runtime's waits (A/B).
hipError_t waitEvent(hipEvent_t event);
hipError_t waitStream(hipStream_t stream);

// ---- zeroing device memory in one launch -----------------------------------------------------------
// hipMemsetAsync of a size or address that is not a multiple of its fill kernel's width becomes up to four fill kernels (head, body,
// tail), each a dispatch that waits its turn on a busy GPU (eleven per configs[2] batch, 22 us each next to other batches' kernels).
// One kernel of 4-byte stores instead; ptr and bytes multiples of four (anything else goes to hipMemsetAsync).
hipError_t zeroAsync(void * ptr, size_t bytes, hipStream_t stream);

// ---- caching device allocator -------------------------------------------------
// hipMalloc/hipFree cost tens of microseconds to milliseconds each (hipFree also
// synchronises the device); a step of the hot path needs ~60 scratch arrays whose
// sizes repeat from batch to batch, and the GPU has 288 GB.  Freed blocks are
// therefore kept in size-class free lists per device and handed out again;
// everything is returned to the driver when the last context of the device dies.
hipError_t poolAlloc(void ** ptr, size_t bytes);
void poolFree(void * ptr);
void poolTrim(int device);

// ---- pinned staging memory -----------------------------------------------------
// hipMemcpyAsync from pageable host memory is not asynchronous: the runtime stages it through its own pinned buffer
// and the calling thread waits for the copy — behind whatever the GPU is busy with (a second host lane was seen
// blocked for 1-2 ms per batch in the uploads of a few megabytes of column lists while the first lane's kernels ran).
// Uploads therefore go through pinned blocks of the library's own (cached by size class like the device blocks,
// hipHostMalloc costs milliseconds): memcpy into the block, hipMemcpyAsync from it, and the block lives as long as the
// device buffer it filled.  RPVG_HIP_PAGEABLE_UPLOADS=1 restores the direct copies (A/B).
hipError_t pinnedAlloc(void ** ptr, size_t bytes);
void pinnedFree(void * ptr);
size_t pinnedCapacity(const void * ptr);  // of a live block of pinnedAlloc that starts at ptr; 0: not one
void pinnedTrim();
bool stagedUploads();
// An upload does not depend on the kernels queued before it on its stream (it fills a fresh block with host data), but
// queued on that stream it would wait for them: the search's inputs sat behind the build of the matrices, ~0.5-0.9 ms
// per batch on the critical path.  Every context owns a copy stream; a staged upload "on" one of the context's streams
// runs on the copy stream and the stream waits for it (event), so it overlaps the kernels in front of it.
// (Blocks come from the pool only after the work that used them has been waited for — the rule the stream-unaware
// pool rests on anyway.)  Measured slower than uploads on their own streams (see stagedCopy): RPVG_HIP_COPY_STREAM=1
// turns it on.
// host memory the caller registered with rpvg_hip_host_register (page-locked: the copy engine reads it directly)
bool hostIsPinned(const void * host, size_t bytes);
// memcpy; large blocks are copied by several threads (one core moves ~10 GB/s, a batch of rows is ~230 MB)
void copyToStaging(void * staging, const void * host, size_t bytes);
void registerCopyStream(hipStream_t stream, hipStream_t copy_stream, hipEvent_t copied);
void forgetCopyStream(hipStream_t stream);
hipError_t stagedCopy(void * device_dst, const void * pinned_src, size_t bytes, hipStream_t stream);

// ---- device buffer (owning) -------------------------------------------------
template <typename T>
struct DeviceBuffer {
    T * ptr = nullptr;
    size_t count = 0;
    void * staging = nullptr;  // pinned block the last upload went through
    bool borrowed = false;     // ptr points into a block owned elsewhere
    DeviceBuffer() {}
    DeviceBuffer(const DeviceBuffer &) = delete;
    DeviceBuffer & operator=(const DeviceBuffer &) = delete;
    ~DeviceBuffer() { release(); }
    void release() {
        if (ptr && !borrowed) poolFree(ptr);
        if (staging) pinnedFree(staging);
        ptr = nullptr;
        staging = nullptr;
        count = 0;
        borrowed = false;
    }
    hipError_t alloc(size_t n) {
        release();
        count = n;
        if (n == 0) return hipSuccess;
        return poolAlloc(reinterpret_cast<void **>(&ptr), n * sizeof(T));
    }
    // part of a block somebody else owns (UploadPack): nothing to free
    void borrow(T * p, size_t n) {
        release();
        ptr = n ? p : nullptr;
        count = n;
        borrowed = true;
    }
    hipError_t upload(const T * host, size_t n, hipStream_t stream) {
        hipError_t e = alloc(n);
        if (e != hipSuccess || n == 0) return e;
        if (hostIsPinned(host, n * sizeof(T))) {  // the caller's own pinned memory (rpvg_hip_host_register): no staging
            return hipMemcpyAsync(ptr, host, n * sizeof(T), hipMemcpyHostToDevice, stream);
        }
        if (stagedUploads() && pinnedAlloc(&staging, n * sizeof(T)) == hipSuccess) {
            copyToStaging(staging, host, n * sizeof(T));
            return stagedCopy(ptr, staging, n * sizeof(T), stream);
        }
        staging = nullptr;
        return hipMemcpyAsync(ptr, host, n * sizeof(T), hipMemcpyHostToDevice, stream);
    }
    hipError_t download(T * host, hipStream_t stream) const {
        if (count == 0) return hipSuccess;
        return hipMemcpyAsync(host, ptr, count * sizeof(T), hipMemcpyDeviceToHost, stream);
    }
};

// ---- several small uploads as one --------------------------------------------
// A command queued on a stream costs the submitting thread 60-100 us when another host lane is submitting too: the ten
// little arrays a search uploads one by one were 1 ms of every lane's critical path (rocprofv3 kernel trace, round 2).
// The pack lays the arrays out in ONE device block (256-byte aligned pieces), fills one pinned staging block and queues
// one H2D copy; pieces that only need zeros sit behind the copied part and take one memset.  The device buffers handed
// to add()/addZero() become views of the block (DeviceBuffer::borrow): the pack has to outlive their use.
struct UploadPack {
    struct Piece {
        const void * host;
        size_t bytes, offset;
        std::function<void(unsigned char *)> bind;
    };
    std::vector<Piece> pieces, zeros;
    size_t copied_bytes = 0, zero_bytes = 0;
    DeviceBuffer<unsigned char> block;
    std::vector<unsigned char> pageable;  // RPVG_HIP_PAGEABLE_UPLOADS=1: assembled here instead of a pinned block
    static size_t aligned(size_t bytes) { return (bytes + 255) & ~size_t(255); }
    template <typename T>
    void add(DeviceBuffer<T> & target, const T * host, size_t n) {
        DeviceBuffer<T> * t = &target;
        pieces.push_back(Piece{host, n * sizeof(T), copied_bytes, [t, n](unsigned char * at) { t->borrow(reinterpret_cast<T *>(at), n); }});
        copied_bytes += aligned(n * sizeof(T));
    }
    template <typename T>
    void addZero(DeviceBuffer<T> & target, size_t n) {
        DeviceBuffer<T> * t = &target;
        zeros.push_back(Piece{nullptr, n * sizeof(T), zero_bytes, [t, n](unsigned char * at) { t->borrow(reinterpret_cast<T *>(at), n); }});
        zero_bytes += aligned(n * sizeof(T));
    }
    hipError_t commit(hipStream_t stream) {
        hipError_t e = block.alloc(copied_bytes + zero_bytes);
        if (e != hipSuccess || block.count == 0) {
            for (auto & p : pieces) p.bind(nullptr);
            for (auto & p : zeros) p.bind(nullptr);
            return e;
        }
        if (copied_bytes > 0) {
            unsigned char * host_block = nullptr;
            if (stagedUploads() && pinnedAlloc(&block.staging, copied_bytes) == hipSuccess) {
                host_block = static_cast<unsigned char *>(block.staging);
            } else {
                block.staging = nullptr;
                pageable.resize(copied_bytes);
                host_block = pageable.data();
            }
            for (auto & p : pieces) {
                if (p.bytes) copyToStaging(host_block + p.offset, p.host, p.bytes);
            }
            e = block.staging ? stagedCopy(block.ptr, host_block, copied_bytes, stream)
                              : hipMemcpyAsync(block.ptr, host_block, copied_bytes, hipMemcpyHostToDevice, stream);
            if (e != hipSuccess) return e;
        }
        if (zero_bytes > 0) {
            e = zeroAsync(block.ptr + copied_bytes, zero_bytes, stream);
            if (e != hipSuccess) return e;
        }
        for (auto & p : pieces) p.bind(block.ptr + p.offset);
        for (auto & p : zeros) p.bind(block.ptr + copied_bytes + p.offset);
        return hipSuccess;
    }
};

// The same for results: the device arrays are views of one block, fetch() queues ONE D2H copy into a pinned block
// (a copy into pageable memory is staged by the runtime and holds the calling thread), scatter() hands the pieces to
// the caller's arrays once the stream has been waited for.
struct DownloadPack {
    struct Piece {
        void * host;
        size_t bytes, offset;
        std::function<void(unsigned char *)> bind;
    };
    std::vector<Piece> pieces;
    size_t total = 0;
    DeviceBuffer<unsigned char> block;
    void * pinned = nullptr;
    std::vector<unsigned char> pageable;
    DownloadPack() {}
    DownloadPack(const DownloadPack &) = delete;
    DownloadPack & operator=(const DownloadPack &) = delete;
    ~DownloadPack() { if (pinned) pinnedFree(pinned); }
    template <typename T>
    void add(DeviceBuffer<T> & target, T * host, size_t n) {
        DeviceBuffer<T> * t = &target;
        pieces.push_back(Piece{host, n * sizeof(T), total, [t, n](unsigned char * at) { t->borrow(reinterpret_cast<T *>(at), n); }});
        total += UploadPack::aligned(n * sizeof(T));
    }
    hipError_t alloc() {
        hipError_t e = block.alloc(total);
        for (auto & p : pieces) p.bind(block.ptr ? block.ptr + p.offset : nullptr);
        return e;
    }
    hipError_t fetch(hipStream_t stream) {
        if (total == 0) return hipSuccess;
        if (!pinned && pinnedAlloc(&pinned, total) != hipSuccess) {
            pinned = nullptr;
            pageable.resize(total);
        }
        return hipMemcpyAsync(pinned ? pinned : pageable.data(), block.ptr, total, hipMemcpyDeviceToHost, stream);
    }
    void scatter() {
        const unsigned char * from = static_cast<const unsigned char *>(pinned ? pinned : pageable.data());
        for (auto & p : pieces) {
            if (p.bytes) copyToStaging(p.host, from + p.offset, p.bytes);
        }
    }
};

// ---- host-side wall-clock tracing (RPVG_AMD_TRACE=1) --------------------------
struct HostScope {
    const char * name;
    std::chrono::steady_clock::time_point t0;
    explicit HostScope(const char * n) : name(n), t0(std::chrono::steady_clock::now()) {}
    ~HostScope() {
        static const bool on = std::getenv("RPVG_AMD_TRACE") != nullptr;
        if (on) {
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            std::fprintf(stderr, "[rpvg_hip trace]   %-44s %9.3f ms\n", name, ms);
        }
    }
};

// ---- wave-level FP64 sum on the DPP network -------------------------------------
// __shfl_xor on a double lowers to two ds_bpermute_b32 per step (LDS crossbar, ~100 cycles of dependent
// latency each, 6 steps).  DPP moves stay in the SIMD: quad_perm x2, row_half_mirror, row_mirror give
// every lane the sum of its row of 16; the four row sums are then read with v_readlane and added.
// The result is uniform across the wave.  Summation order is fixed (deterministic).
// entry of the logarithm table (logPositive with a table, below)
struct LogTableEntry {
    float rc;
    float lo;
    double hi;
};
constexpr int kLogTableSize = 129;

#if defined(__HIP_DEVICE_COMPILE__)
template <int CTRL>
__device__ __forceinline__ double dppAddF64(const double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return v + __hiloint2double(hi2, lo2);
}

__device__ __forceinline__ double readLaneF64(const double v, const int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}

// sum over the lane's row of 16 lanes, in every lane of the row
__device__ __forceinline__ double rowSumF64(double v) {
    v = dppAddF64<0xB1>(v);   // quad_perm [1,0,3,2]
    v = dppAddF64<0x4E>(v);   // quad_perm [2,3,0,1]
    v = dppAddF64<0x141>(v);  // row_half_mirror
    return dppAddF64<0x140>(v);  // row_mirror
}

__device__ __forceinline__ double waveSumF64(double v) {
    v = dppAddF64<0xB1>(v);   // quad_perm [1,0,3,2]
    v = dppAddF64<0x4E>(v);   // quad_perm [2,3,0,1]
    v = dppAddF64<0x141>(v);  // row_half_mirror
    v = dppAddF64<0x140>(v);  // row_mirror
    return (readLaneF64(v, 0) + readLaneF64(v, 16)) + (readLaneF64(v, 32) + readLaneF64(v, 48));
}

// ---- natural log of a positive normal double -------------------------------------
// ROCm's device log() costs ~90 FP64 instructions per call (double-double arithmetic), and the
// log-likelihood kernels are bound by exactly that.  Their arguments are probabilities in
// [prob_precision, ~2): positive, finite, normal.  For that domain the classic reduction
// x = 2^k (1+f), sqrt(1/2) <= 1+f < sqrt(2), s = f/(2+f), log(1+f) = f - f^2/2 + s (f^2/2 + R(s^2))
// with a degree-7 minimax R (the algorithm and coefficients published with Sun's fdlibm log)
// needs ~35 instructions and stays below 1 ulp (checked against long double over the domain:
// 0.67 ulp max; glibc's log: 0.55).
__device__ __forceinline__ double logPositive(const double x) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    double m = __builtin_amdgcn_frexp_mant(x);  // [0.5, 1)
    int k = __builtin_amdgcn_frexp_exp(x);
    const bool low = m < 0.70710678118654752440;
    m = low ? m * 2.0 : m;
    k = low ? k - 1 : k;
    const double f = m - 1.0;
    // s = f / (2 + f) without the IEEE division sequence: hardware reciprocal (~24 bits) + two Newton steps.
    // s only multiplies the O(f^2) tail below, so its last-bit error reaches the result scaled by |f| / 2.
    const double d = 2.0 + f;
    double y = __builtin_amdgcn_rcp(d);
    y = fma(fma(-d, y, 1.0), y, y);
    y = fma(fma(-d, y, 1.0), y, y);
    const double s = f * y;
    const double z = s * s, w = z * z;
    const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = static_cast<double>(k);
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

// The same logarithm without the reciprocal: x = 2^k m with m in [0.5, 1); entry i = round((m - 0.5) * 256) of a
// 129-entry table (in the workgroup's LDS) holds rc ~ 1 / (0.5 + i/256) as a float and -log(rc) as hi + lo, so that
// log m = hi + lo + log1p(r) exactly for r = m * rc - 1, |r| <= 2^-8, and log1p is a degree-7 polynomial.
// ~21 instructions and one 16-byte LDS read; the absolute error stays below 2e-16 and the relative error below
// 1 ulp away from x ~ 1+ (k = 1, i = 0), where hi and k ln2 cancel and the float `lo` leaves ~1e-17 absolute.
static __device__ const LogTableEntry kLogTable[kLogTableSize] = {
#include "log_table.inc"
};

// Copies the table into the workgroup's LDS; every thread of the block calls it, a __syncthreads() follows.
__device__ __forceinline__ void loadLogTable(LogTableEntry * lds_table) {
    for (int i = threadIdx.x; i < kLogTableSize; i += blockDim.x) lds_table[i] = kLogTable[i];
}

__device__ __forceinline__ double logPositive(const double x, const LogTableEntry * lds_table) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    // x = 2^k m, m in [0.5, 1), through the bit pattern (positive normal x): integer instructions instead of
    // v_frexp_*_f64 and of the float -> int conversion of the table index
    const uint32_t hi_word = static_cast<uint32_t>(__double2hiint(x));
    const int k = static_cast<int>(hi_word >> 20) - 1022;
    const double m = __hiloint2double(static_cast<int>((hi_word & 0x000fffffu) | 0x3fe00000u), __double2loint(x));
    const uint32_t i = ((hi_word & 0x000fffffu) + 0x1000u) >> 13;  // round((m - 0.5) * 256): 0 .. 128
    const LogTableEntry e = lds_table[i];
    const double r = fma(m, static_cast<double>(e.rc), -1.0);
    double q = fma(r, 1.0 / 7.0, -1.0 / 6.0);
    q = fma(r, q, 0.2);
    q = fma(r, q, -0.25);
    q = fma(r, q, 1.0 / 3.0);
    q = fma(r, q, -0.5);
    const double p = fma(r * r, q, r);
    const double dk = static_cast<double>(k);
    return fma(dk, ln2_hi, e.hi) + (fma(dk, ln2_lo, static_cast<double>(e.lo)) + p);
}

#else
__device__ double waveSumF64(double v);  // host compilation pass: declarations only
__device__ double readLaneF64(double v, int lane);
__device__ double rowSumF64(double v);
__device__ double logPositive(double x);
__device__ double logPositive(double x, const LogTableEntry * lds_table);
__device__ void loadLogTable(LogTableEntry * lds_table);
#endif

// ---- sum of logarithms through a running product ------------------------------------
// sum_i c_i log(x_i) = log(prod_i x_i^c_i).  The rows of a group matrix are ordered by class (rpvg_hip_groups):
//   fast  read count 1                    : the factor joins an FP64 product — one multiplication instead of a logarithm;
//   mid   read counts 2 .. kMidMaxCount,
//         ascending                       : the factor is multiplied in c_i times (rows of a wave have nearly equal counts);
//   slow  everything else                 : one logarithm per row, times the count.
// Fast and mid rows have a noise probability of at least kProductMinNoise, a lower bound of every log argument of
// the row; arguments are sums of probabilities, far below 2^21.  A product that starts in [1, 2) therefore stays a
// normal number over kFoldFactors fast factors or one mid factor; fold() then moves its exponent to an integer sum.
// One logarithm at the end.  The rounding error of a product of n factors grows like n * 2^-53, the same order as
// that of a sum of n logarithms.
constexpr double kProductMinNoise = 9.313225746154785e-10;  // 2^-30
constexpr int kMidMaxCount = 8;
constexpr uint32_t kMidMinRows = 1;     // every matrix has the mid class: the tile kernel of the diploid search pays nothing for a class
                                        // boundary (round 1's sequential kernels did: 1 024), and half of the bench's rows sit in matrices below 1 024 rows
constexpr uint32_t kFoldFactors = 30;  // (30 + 3 remainder factors) * 30 bits < 1022
constexpr uint32_t kNumRowClasses = kMidMaxCount + 1;  // fast, counts 2 .. kMidMaxCount, slow

__host__ __device__ inline uint32_t rowClass(const double count, const double noise, const bool with_mid) {
    if (!(noise >= kProductMinNoise)) return kNumRowClasses - 1;
    if (count == 1.0) return 0;
    if (!with_mid) return kNumRowClasses - 1;
    for (int c = 2; c <= kMidMaxCount; ++c)
        if (count == static_cast<double>(c)) return static_cast<uint32_t>(c - 1);
    return kNumRowClasses - 1;
}

struct LogProduct {
    double p = 1.0;
    int e = 0;  // exponent folded out of p so far
    __device__ __forceinline__ void mul(const double x) { p *= x; }
    __device__ __forceinline__ void mulPow(const double x, const int c) {
        for (int k = 0; k < c; ++k) p *= x;
    }
    __device__ __forceinline__ void fold() {
        const uint32_t hi_word = static_cast<uint32_t>(__double2hiint(p));
        e += static_cast<int>(hi_word >> 20) - 1023;
        p = __hiloint2double(static_cast<int>((hi_word & 0x000fffffu) | 0x3ff00000u), __double2loint(p));
    }
    // both folded
    __device__ __forceinline__ void join(const LogProduct & other) {
        p *= other.p;
        e += other.e;
    }
    __device__ __forceinline__ double value(const LogTableEntry * lds_table) {
        const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
        fold();
        const double de = static_cast<double>(e);
        return fma(de, ln2_hi, fma(de, ln2_lo, logPositive(p, lds_table)));
    }
};

// sum over rows [begin, end) of a wave (lane handles begin + lane, + 64, ...) of count_i * log(x(i)); rows below
// fast_end are fast, rows below mid_end mid (classes above).
template <typename IndexT, typename XFn>
__device__ __forceinline__ double sumCountLogs(const LogTableEntry * lt, const double * __restrict__ cnt, XFn x, const IndexT begin,
                                               const IndexT fast_end, const IndexT mid_end, const IndexT end, const int lane) {
    constexpr uint32_t kSegment = 4 * 64 * kFoldFactors;  // four chains
    double acc = 0.0;
    if (begin < mid_end) {
        LogProduct p0, p1, p2, p3;
        for (IndexT seg = begin; seg < fast_end; seg += kSegment) {
            const IndexT seg_end = (fast_end - seg) < kSegment ? fast_end : seg + kSegment;
            IndexT i = seg + lane;
            for (; i + 192 < seg_end; i += 256) {
                p0.mul(x(i));
                p1.mul(x(i + 64));
                p2.mul(x(i + 128));
                p3.mul(x(i + 192));
            }
            for (; i < seg_end; i += 64) p0.mul(x(i));
            p0.fold();
            p1.fold();
            p2.fold();
            p3.fold();
        }
        for (IndexT i = (begin < fast_end ? fast_end : begin) + lane; i < mid_end; i += 64) {
            p1.mulPow(x(i), static_cast<int>(cnt[i]));
            p1.fold();
        }
        p0.join(p1);
        p2.join(p3);
        p0.join(p2);
        acc = p0.value(lt);
    }
    double acc0 = 0.0, acc1 = 0.0;
    IndexT i = (begin < mid_end ? mid_end : begin) + lane;
    for (; i + 64 < end; i += 128) {
        acc0 = fma(cnt[i], logPositive(x(i), lt), acc0);
        acc1 = fma(cnt[i + 64], logPositive(x(i + 64), lt), acc1);
    }
    for (; i < end; i += 64) acc0 = fma(cnt[i], logPositive(x(i), lt), acc0);
    return acc + (acc0 + acc1);
}

// The same for kOut sums at once that share the rows: x(i, xs) fills the kOut log arguments of row i.  The rows are
// dealt to a group of kGroup lanes (a wave, or 16 lanes of it): `lane` is the index within the group.
template <int kOut, int kGroup, typename IndexT, typename XFn>
__device__ __forceinline__ void sumCountLogsMulti(const LogTableEntry * lt, const double * __restrict__ cnt, XFn x, const IndexT begin,
                                                  const IndexT fast_end, const IndexT mid_end, const IndexT end, const int lane,
                                                  LogProduct (&pr)[kOut], double (&acc)[kOut]) {
    constexpr uint32_t kSegment = kGroup * kFoldFactors;
    for (IndexT seg = begin; seg < fast_end; seg += kSegment) {
        const IndexT seg_end = (fast_end - seg) < kSegment ? fast_end : seg + kSegment;
        for (IndexT i = seg + lane; i < seg_end; i += kGroup) {
            double xs[kOut];
            x(i, xs);
#pragma unroll
            for (int t = 0; t < kOut; ++t) pr[t].mul(xs[t]);
        }
#pragma unroll
        for (int t = 0; t < kOut; ++t) pr[t].fold();
    }
    for (IndexT i = (begin < fast_end ? fast_end : begin) + lane; i < mid_end; i += kGroup) {
        double xs[kOut];
        x(i, xs);
        const int c = static_cast<int>(cnt[i]);
        for (int k = 0; k < c; ++k) {
#pragma unroll
            for (int t = 0; t < kOut; ++t) pr[t].mul(xs[t]);
        }
#pragma unroll
        for (int t = 0; t < kOut; ++t) pr[t].fold();
    }
    for (IndexT i = (begin < mid_end ? mid_end : begin) + lane; i < end; i += kGroup) {
        double xs[kOut];
        x(i, xs);
        const double c = cnt[i];
#pragma unroll
        for (int t = 0; t < kOut; ++t) acc[t] = fma(c, logPositive(xs[t], lt), acc[t]);
    }
}

// ---- row collapse of the group matrices (row_collapse.hip) ---------------------------------------
// Every row of a normalised matrix carries a projection key: the sum of its values (noise last) with fixed weights
// in [0, 2).  Rows within prob_precision of each other in every column have keys within 2 (G + 1) prob_precision;
// rows of a normalised matrix sum to at most 1, so keys lie in [0, 4).  The sort key is
//   [ matrix : 20 | projection key, fixed point, 2.22 : 24 | largest value of the row, fixed point 0.27, low 20 bits ]
// sorted on the upper 44 bits only.  The largest values of close rows differ by less than prob_precision, i.e. by a
// step or two of the low field (modulo 2^20: neighbours in key order are not far apart to begin with), which
// therefore filters the neighbours of a row in key order at 7.5e-9 resolution without a memory access.
// A quantum of the projection of 2.4e-7 is a fraction of the window at the default precision (2 x 65 x 1e-8).
// Rows also carry the zero pattern of their first 64 columns (bit c: column c is not zero): rows that agree up to
// rounding on a stretch of columns have the same pattern there.
constexpr int kCollapseKeyFractionBits = 22;
constexpr int kCollapseProjectionBits = kCollapseKeyFractionBits + 2;
constexpr int kCollapseLargestBits = 20;
constexpr int kCollapseLargestFractionBits = 27;
constexpr int kCollapseMatrixShift = kCollapseProjectionBits + kCollapseLargestBits;
constexpr uint32_t kCollapseMaxMatrices = 1u << 20;

__host__ __device__ inline double collapseWeight(const uint32_t column) {
    return static_cast<double>(((column * 0x9E3779B1u) >> 8) & 0xFFFFu) * (1.0 / 32768.0);
}

__host__ __device__ inline uint64_t collapseSortKey(const uint32_t matrix, const double key, const double largest) {
    const double k = key < 0.0 ? 0.0 : (key > 3.999999 ? 3.999999 : key);
    const double l = largest < 0.0 ? 0.0 : (largest > 0.999999 ? 0.999999 : largest);
    return (static_cast<uint64_t>(matrix) << kCollapseMatrixShift) |
           (static_cast<uint64_t>(k * static_cast<double>(1ull << kCollapseKeyFractionBits)) << kCollapseLargestBits) |
           (static_cast<uint64_t>(l * static_cast<double>(1ull << kCollapseLargestFractionBits)) & ((1ull << kCollapseLargestBits) - 1));
}

// ---- products and sums that stay two roundings --------------------------------------------------------
// hipcc contracts a * b + c into a fused multiply-add by default (-ffp-contract=fast), and HIP's __dmul_rn / __dadd_rn are plain
// operators that take part in it.  Where a kernel restates host arithmetic addition for addition (the posterior-weighted merge,
// subset_em.hip: the host is compiled for baseline x86-64, which has no fused multiply-add), the operations go through these.
__device__ __forceinline__ double mulRounded(const double a, const double b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ double addRounded(const double a, const double b) {
#pragma clang fp contract(off)
    return a + b;
}

// ---- exclusive prefix sum over the threads of a workgroup ---------------------------------------------
// (scratch: BLOCK / 64 words of LDS; two barriers)
template <int BLOCK>
__device__ __forceinline__ uint32_t blockExclusiveSum(const uint32_t v, uint32_t & total, uint32_t * scratch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d, 64);
        if (lane >= d) incl += up;
    }
    __syncthreads();
    if (lane == 63) scratch[wave] = incl;
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w) {
        if (w < wave) before += scratch[w];
        all += scratch[w];
    }
    total = all;
    return before + incl - v;
}

// ---- kernel-family timing ---------------------------------------------------
enum KernelFamily { FAM_EM_SPARSE = 0, FAM_EM_DENSE, FAM_LOGLIK, FAM_BUILD, FAM_H2D, FAM_COLLAPSE, FAM_EM_KERNEL, FAM_GIBBS, FAM_TILE, FAM_COUNT };

struct TimedSpan {
    hipEvent_t start, stop;
    int family;
    int sub;  // FAM_EM_KERNEL: index of the kernel variant (rpvg_hip_em_kernel_name)
};

// A span folded into the statistics, on the clock of the device's base event (context.hip).
struct TimedInterval {
    double start_ms, stop_ms;
    int family;
    uint64_t clock;  // the base event it was measured against: intervals of different clocks do not compare
};

}  // namespace rpvg_hip_detail

namespace rpvg_hip_detail {
// what the calls of rpvg_hip_nested_subset_em on a context needed, per unit of their input (subsets per matrix, list entries per path,
// kept rows / entries per row / entry of the call's clusters): the largest of the recent calls, fading — the next call reserves by
// them whatever its size (the batches of PathEstimator::estimate()'s call combiner differ by orders of magnitude from one to the
// next); and the exact figures of a call that did not fit, for its second attempt (subset_em.hip)
struct SubsetEmHints {
    double subsets_per_matrix = 0, list_per_path = 0, rows_per_row = 0, entries_per_entry = 0;
    bool retry = false;
    unsigned long long retry_subsets = 0, retry_list_length = 0, retry_rows = 0, retry_entries = 0;
};
}  // namespace rpvg_hip_detail

struct rpvg_hip_ctx {
    static constexpr int kAuxStreams = 6;
    rpvg_hip_detail::SubsetEmHints subset_hints;
    bool search_gate_held = false;  // the search this context has queued takes part in "one search at a time" (bounded_search.hip)
    int device = 0;
    hipStream_t stream = nullptr;
    // Side streams for independent launches of one call (size bins of the batched kernels): their tails
    // overlap instead of adding up.  forkAux() makes them wait for the work queued on `stream` so far,
    // joinAux() makes `stream` wait for them.
    hipStream_t aux[kAuxStreams] = {};
    int aux_count = kAuxStreams;  // real side streams: aux[i] for i >= aux_count aliases aux[i % aux_count] (rpvg_hip_create_with_streams)
    // streams of the EM problems that run over the whole GPU (em_grid.hip: two at a time), made when a solve first has such problems
    hipStream_t grid_stream[2] = {nullptr, nullptr};
    hipEvent_t grid_ready = nullptr;
    hipStream_t collapse_stream = nullptr;  // row collapse of the matrices a build leaves behind (highest priority: short kernels next to a search)
    hipStream_t copy_stream = nullptr;  // staged uploads (stagedCopy)
    hipEvent_t copied = nullptr;
    hipEvent_t fork_event = nullptr;
    hipEvent_t search_done = nullptr;  // recorded behind the kernels of this context's last pair search (bounded_search.hip)
    hipEvent_t join_event[kAuxStreams] = {};
    hipError_t forkAux();
    hipError_t joinAux();
    hipError_t joinAuxOnHost();
    hipDeviceProp_t props;
    std::mutex mutex;  // serialises calls on this context
    // RCCL communicator of this rank (comm.hip); null until rpvg_hip_comm_init.  Collectives are
    // queued on `stream`, so they are ordered with the kernels around them without a host sync.
    void * comm = nullptr;
    int comm_world = 1, comm_rank = 0;
    // in-place sum over ranks of n doubles, stream-ordered (comm.hip); requires comm != null
    int allReduceSumF64(double * device_buf, uint64_t n);
    std::vector<rpvg_hip_detail::TimedSpan> spans;
    std::vector<hipStream_t> span_streams;  // the stream of each open span
    int span_level = 2;                     // which families spanBegin() times (context.hip)
    int open_spans = 0;                     // begun, not ended
    void accountSpan(const rpvg_hip_detail::TimedSpan & span, const void * clock_base, uint64_t clock_id);
    void foldFinishedSpans();
    std::vector<rpvg_hip_detail::TimedInterval> intervals;  // folded spans since the last reset
    rpvg_hip_kernel_stats stats;

    // Opens a timed span on `on` (the context's stream when null); returns its index (or -1 on failure).
    int spanBegin(int family, hipStream_t on = nullptr, int sub = -1);
    void spanEnd(int idx);
    // Folds all finished spans into stats (synchronises the stream).
    int foldSpans();
};

namespace rpvg_hip_detail {
// ---- the path side of a batch (path_sources.hip) -----------------------------------------------------
// What an upload keeps between its copies and the kernels behind them: the copied PathInfo::source_ids, the scratch of the
// kernel that forms the haplotype columns of every cluster, the sizes it brings back.
struct PathSourcesPending {
    DeviceBuffer<uint64_t> d_path_source_off;
    DeviceBuffer<uint32_t> d_source_id, d_sizes;
    DeviceBuffer<uint16_t> d_source_id16;  // rpvg_cluster_batch::source_id16: widened into d_source_id behind the copy
    DeviceBuffer<unsigned long long> d_arena;
    unsigned long long arena_words = 0, num_sources = 0, num_sources_narrow = 0;
    void * h_sizes = nullptr;  // pinned: [num_cols K | col_paths K | max_col_paths K | error, arena overflow]
    uint32_t K = 0;
    bool copied = false, queued = false;
    ~PathSourcesPending() { if (h_sizes) pinnedFree(h_sizes); }
};
}  // namespace rpvg_hip_detail

// Device-resident batch: the expanded CSR of every cluster.
struct rpvg_hip_batch {
    uint32_t num_clusters = 0;
    uint64_t num_rows = 0, num_entries = 0, num_paths = 0;
    // host copies of the small per-cluster offsets (needed to size problems)
    std::vector<uint64_t> h_cluster_row_off, h_cluster_path_off, h_cluster_ent_off;
    rpvg_hip_detail::DeviceBuffer<uint64_t> cluster_row_off;   // [K+1]
    rpvg_hip_detail::DeviceBuffer<uint64_t> cluster_path_off;  // [K+1]
    rpvg_hip_detail::DeviceBuffer<double> row_count;           // [R] read count as double
    rpvg_hip_detail::DeviceBuffer<double> row_noise;           // [R]
    rpvg_hip_detail::DeviceBuffer<uint64_t> row_ent_off;       // [R+1] entry range of each row
    rpvg_hip_detail::DeviceBuffer<uint32_t> ent_path;          // [NNZ] cluster-local path
    rpvg_hip_detail::DeviceBuffer<double> ent_prob;            // [NNZ]
    // read count of every cluster (exact: integers), summed on the device behind the copy of the rows
    std::vector<double> h_cluster_total;
    // ---- the path side of the batch (path_sources.hip), present when the host batch carried PathInfo::group_id and
    // PathInfo::source_ids: the haplotype columns of every cluster — findPathSourceGroups, src/path_abundance_estimator.cpp:493-546,
    // done once per batch on the device behind the copy — in a layout by bounds (a cluster has at most as many columns, and its
    // columns list at most as many paths, as it has (haplotype, path) incidences): cluster k owns the slots
    // [h_cluster_src_off[k], h_cluster_src_off[k + 1]) of the three arrays below; the first h_src_num_cols[k] of them are used.
    bool has_source_columns = false;
    rpvg_hip_detail::DeviceBuffer<uint32_t> path_group_id;     // [P]  PathInfo::group_id (the transcript of a path)
    rpvg_hip_detail::DeviceBuffer<uint64_t> cluster_src_off;   // [K+1]
    rpvg_hip_detail::DeviceBuffer<uint32_t> src_col_count;     // haplotypes that carry the column's path list (path_counts)
    rpvg_hip_detail::DeviceBuffer<uint32_t> src_col_end;       // end of the column's list in src_col_path, relative to the cluster's first slot
    rpvg_hip_detail::DeviceBuffer<uint32_t> src_col_path;      // the lists, ascending cluster-local paths, columns back to back
    std::vector<uint64_t> h_cluster_src_off;                   // [K+1]
    std::vector<uint32_t> h_src_num_cols, h_src_col_paths, h_src_max_col_paths;  // [K] columns, sum and maximum of their list lengths
    // An upload in two halves (rpvg_hip_batch_upload_begin / _finish): what the kernels of the second half read and free.
    struct UploadInProgress {
        rpvg_hip_detail::DeviceBuffer<uint32_t> d_row_count_u32, d_row_grp_off32, d_grp_idx_off32;
        rpvg_hip_detail::DeviceBuffer<uint64_t> d_row_grp_off, d_grp_idx_off;
        rpvg_hip_detail::DeviceBuffer<double> d_grp_prob;
        rpvg_hip_detail::DeviceBuffer<uint8_t> d_row_grp_count8, d_grp_idx_count8;  // the count form (include/rpvg_batch.h): summed up into the 32-bit offsets behind the copy
        // the narrow forms (rpvg_cluster_batch::path_idx16 / row_count8 + escapes): widened behind the copy (widenNarrowKernel)
        rpvg_hip_detail::DeviceBuffer<uint16_t> d_path_idx16, d_row_noise16;
        rpvg_hip_detail::DeviceBuffer<double> d_row_noise_table;
        rpvg_hip_detail::DeviceBuffer<uint8_t> d_row_count8;
        rpvg_hip_detail::DeviceBuffer<uint32_t> d_escape_row, d_escape_count;
        uint64_t num_groups = 0;
        rpvg_hip_detail::PathSourcesPending path_sources;
        // the second half queued (uploadFinishQueue): what its kernels write and what comes back, until the wait
        rpvg_hip_detail::DeviceBuffer<unsigned long long> d_first_bad_row;
        rpvg_hip_detail::DeviceBuffer<double> d_cluster_total;
        rpvg_hip_detail::DeviceBuffer<uint64_t> d_cluster_ent_off;
        rpvg_hip_detail::DeviceBuffer<unsigned char> scan_scratch_rows, scan_scratch_groups;
        void * h_results = nullptr;   // page-locked
        hipEvent_t finished = nullptr;
        bool counts = false;
        ~UploadInProgress() {
            if (finished) {
                (void) hipEventSynchronize(finished);  // (a batch freed between the two steps: its kernels use the buffers above)
                (void) hipEventDestroy(finished);
            }
            if (h_results) rpvg_hip_detail::pinnedFree(h_results);
        }
    };
    std::unique_ptr<UploadInProgress> upload;  // null: the batch is complete
};

// Device-resident group matrices (loglik.hip builds them).
// bounded_search.hip: the searches of different contexts of one device run one after the other on the device
void searchGateForget(const rpvg_hip_ctx * ctx);

struct rpvg_hip_groups {
    const rpvg_hip_batch * batch = nullptr;
    uint32_t num_matrices = 0;
    int32_t normalise = 0;
    std::vector<uint32_t> h_num_cols;
    std::vector<uint64_t> h_num_rows;
    rpvg_hip_detail::DeviceBuffer<double> values;         // all matrices back to back, each column-major
    rpvg_hip_detail::DeviceBuffer<double> rowmax;         // [sum R_m]
    rpvg_hip_detail::DeviceBuffer<uint64_t> mat_val_off;  // [M] offset of matrix m in values
    rpvg_hip_detail::DeviceBuffer<uint64_t> mat_row_off;  // [M] offset of matrix m in rowmax
    rpvg_hip_detail::DeviceBuffer<uint64_t> mat_row0;     // [M] first batch row of the matrix's cluster
    rpvg_hip_detail::DeviceBuffer<uint64_t> mat_rows;     // [M] R_m
    rpvg_hip_detail::DeviceBuffer<uint32_t> mat_cols;     // [M] G_m
    // Rows of a matrix are a permutation of its cluster's rows (every consumer sums over rows), ordered by class
    // (LogProduct, above).  Counts and noise in matrix order, indexed like rowmax.
    rpvg_hip_detail::DeviceBuffer<uint32_t> row_perm;     // [sum R_m] cluster-relative source row
    rpvg_hip_detail::DeviceBuffer<double> row_count;      // [sum R_m]
    rpvg_hip_detail::DeviceBuffer<double> row_noise;      // [sum R_m]
    rpvg_hip_detail::DeviceBuffer<uint32_t> mat_fast;     // [M] end of the fast rows
    rpvg_hip_detail::DeviceBuffer<uint32_t> mat_mid;      // [M] end of the mid rows
    // rpvg_hip_groups_build returns with its kernels still queued (no host sync: the caller's next call goes on the
    // same stream and prepares its own launches meanwhile): the temporaries of the build live until the matrices
    // are freed, and the validity flag the kernels set is read by the first consumer (buildError()).
    std::vector<std::shared_ptr<void> > build_temporaries;
    // the columns of the matrices as lists of cluster-local paths (the build's spec, kept on the device: subset_em.hip
    // expands the diplotypes the search retains into path subsets): columns of matrix m = [group_off[m], group_off[m + 1])
    const uint64_t * d_group_off = nullptr;       // [M+1]
    const uint64_t * d_group_path_off = nullptr;  // [G+1]
    const uint32_t * d_group_path = nullptr;
    const uint32_t * d_cluster = nullptr;         // [M] cluster of the batch
    // matrices built from the batch's own haplotype columns (rpvg_hip_groups_build_from_sources): the multiplicity of every
    // column (path_counts of calculatePathGroupPosteriorsBounded), laid out like the columns; null for a caller's spec
    const uint32_t * d_column_counts = nullptr;
    std::vector<uint32_t> h_cluster, h_max_col_paths, h_num_paths;  // per matrix: cluster, longest column list, paths of the cluster
    rpvg_hip_detail::DeviceBuffer<uint32_t> build_error_flag;
    // row collapse (row_collapse.hip): sort keys and row ids written by the build kernels; [0] matrices replayed,
    // [1] rows that took the values of a run head
    rpvg_hip_detail::DeviceBuffer<uint64_t> collapse_key;  // [sum R_m]
    rpvg_hip_detail::DeviceBuffer<uint32_t> collapse_row;  // [sum R_m]
    rpvg_hip_detail::DeviceBuffer<uint64_t> collapse_mask; // [sum R_m] zero pattern of the first 64 columns
    rpvg_hip_detail::DeviceBuffer<uint32_t> collapse_segment_off;  // [M + 1] mat_row_off as 32-bit segment offsets
    rpvg_hip_detail::DeviceBuffer<uint32_t> collapse_info;
    // The row collapse runs on the building context's collapse stream behind the build; whoever reads the matrices
    // makes its stream wait for it (waitCollapse): what a consumer queues before its kernels — uploads, allocations —
    // does not.
    hipEvent_t built = nullptr, collapse_done = nullptr;
    // The last stage of the collapse — the only one that writes the matrices: the rows of a run take the values of its head —
    // can be held back (matrices built from the batch's own columns, rpvg_hip_groups_build_from_sources): the diploid search then
    // reads the matrices as built WHILE the stages in front find the runs, and the stage that is left adjusts the search's sums
    // for the rows it rewrites (bounded_search.hip).  Every other reader gets it queued in front of its own kernels by
    // waitCollapse().  collapse_done: the stages that were queued.
    struct SearchSums {               // the per-chunk sums of the table path (pairTile2Kernel)
        double * part_pair = nullptr;         // [pair_part_off[m] + chunk * G * G + a * G + b], a <= b
        double * part_marginal = nullptr;     // [col_part_off[m] + chunk * G + a]
        const uint64_t * pair_part_off = nullptr;
        const uint64_t * col_part_off = nullptr;
        uint32_t chunk_rows = 0;
    };
    mutable std::function<hipError_t(hipStream_t, const SearchSums *)> held_back_runs;  // empty: nothing held back (any more)
    hipError_t waitCollapse(hipStream_t stream) const {
        hipError_t e = collapse_done ? hipStreamWaitEvent(stream, collapse_done, 0) : hipSuccess;
        if (e == hipSuccess && held_back_runs) {
            e = held_back_runs(stream, nullptr);
            held_back_runs = nullptr;
        }
        return e;
    }
    ~rpvg_hip_groups() {
        if (collapse_done) {
            (void) hipEventSynchronize(collapse_done);  // the collapse kernels use the buffers below
            (void) hipEventDestroy(collapse_done);
        }
        if (built) (void) hipEventDestroy(built);
    }
    rpvg_hip_detail::UploadPack uploads;  // the block behind the small arrays (views: mat_*, build temporaries, build_error_flag)
    mutable bool build_checked = false;
    // RPVG_HIP_OK, or the error of the build (after a sync of `stream`); consumers call it before trusting results
    int buildError(hipStream_t stream) const;
};

namespace rpvg_hip_detail {
// ---- EM solves on problem lists in device memory (em_sparse.hip) -------------------------------------
// The problems may come from the host (rpvg_hip_em_solve) or be written by kernels (subset_em.hip: the path subsets the
// diploid search retains): everything behind the list — compaction of the problems' rows, size bins, work queues, the EM
// kernels — is decided on the device.
struct EmProblemList {
    uint32_t P_bound = 0;                     // number of problems, or an upper bound of it when d_num_problems is set
    const uint32_t * d_num_problems = nullptr;
    const uint32_t * d_cluster = nullptr;     // [P] cluster of the batch
    const uint64_t * d_col_off = nullptr;     // [P+1]
    const uint32_t * d_col_path = nullptr;    // strictly ascending cluster-local paths of each problem
    const uint64_t * d_row_base = nullptr;    // [P] first compacted row / entry of the problem: a problem keeps at most the
    const uint64_t * d_ent_base = nullptr;    //     rows and entries of its cluster, the storage is laid out by that bound
    uint64_t rows_capacity = 0, entries_capacity = 0;
    // the rows of every problem's cluster in segments of emFillSegmentRows(): the work items of the compaction
    const uint64_t * d_seg_first = nullptr;   // [P+1] first item of each problem
    const uint32_t * d_item_problem = nullptr;  // [items]
    uint32_t items_bound = 0;                 // number of items, or an upper bound of it when d_num_items is set
    const uint32_t * d_num_items = nullptr;
    uint32_t max_cols = 0;                    // columns (paths + noise) of the widest problem, or a bound
    uint32_t max_cluster_paths = 0;           // paths of the widest cluster a problem sits on, or a bound
    unsigned long long wide_capacity = 0;     // doubles for the vectors of the problems too wide for LDS
    // rows + entries of the largest cluster a problem sits on (a bound of what a problem keeps): at or above the grid
    // threshold (emGridMinWork()) the solve looks for problems of the grid bin and solves them over the whole GPU (em_grid.hip)
    uint64_t max_cluster_work = 0;
};

struct EmOutputs {  // device arrays, [P] unless noted
    double * d_abundances;      // [col_off[P]] laid out like col_path
    double * d_noise_count;
    uint32_t * d_iterations;
    uint32_t * d_kept_rows;     // rows / entries of the problem that touch a selected path
    uint32_t * d_kept_entries;
    double * d_total;           // read count of the problem's cluster
};

// A problem of the grid bin whose dense row-major matrix (em_grid.hip, the dense route) is written by the compaction itself
// — straight from the cluster's rows, no CSR in between (em_sparse.hip, fillSegmentsKernel<true>): BASELINE.json configs[1].
struct EmFusedDense {
    uint32_t problem, pad;
    double * matrix;
    uint64_t ld;
};
constexpr int kEmMaxFusedDense = 4;
constexpr uint32_t kEmDenseMaxCols = 2048;   // em_dense.hip: a row in the registers of one workgroup
// the dense matrix is the smaller representation (8 B per cell against 12 B per entry + 20 B per row) and a row is narrow enough
__host__ __device__ inline bool emDenseRule(const uint32_t columns, const uint32_t rows, const uint32_t entries) {
    if (columns > kEmDenseMaxCols || columns < 2) return false;
    const uint64_t ld = (static_cast<uint64_t>(columns) + 1) & ~1ull;
    return 8ull * rows * ld <= 12ull * entries + 20ull * rows;
}

struct EmSolveWork {  // scratch of one solve: lives until its kernels are done
    DeviceBuffer<uint32_t> d_prow_off, d_pent_col, d_bucket, d_order, d_seg_rows, d_seg_entries;
    DeviceBuffer<double> d_prow_count, d_prow_noise, d_pent_val, d_zero, d_wide_vectors, d_seg_zero, d_seg_total;
    DeviceBuffer<unsigned long long> d_wide_off;
    DeviceBuffer<unsigned char> d_queues;
    unsigned char * zeroed_queues = nullptr;  // set by a caller that provides the (zeroed) work queues itself: emQueuesBytes()
    // row collapse of the problems (row_collapse.hip) and the second EM pass over the problems it merged rows in
    std::shared_ptr<void> collapse;
    DeviceBuffer<unsigned char> d_queues_merged;
    // dense matrices the compaction wrote itself (queueEmSolve)
    DeviceBuffer<double> fused_matrix[kEmMaxFusedDense];
    EmFusedDense fused[kEmMaxFusedDense];
    uint32_t num_fused = 0;
    hipEvent_t filled = nullptr, collapsed = nullptr, collapse_sorted = nullptr;
    ~EmSolveWork() {
        if (filled) (void) hipEventDestroy(filled);
        if (collapsed) (void) hipEventDestroy(collapsed);
        if (collapse_sorted) (void) hipEventDestroy(collapse_sorted);
    }
};
size_t emQueuesBytes();
uint32_t emFillSegmentRows();

// ---- EM problems too large for one workgroup (em_grid.hip, em_dense.hip) -----------------------------
// The stop rule of the EM on the device (src/path_abundance_estimator.cpp:67-95) for the solvers that run one iteration
// per round of launches: the launches of an iteration exit at once when `done` is set, so the host queues iterations in
// chunks without a synchronisation per iteration and the loop still stops at the reference's iteration.
struct EmGridControl {
    uint32_t done;
    uint32_t iterations;
    uint32_t conv_its;
    uint32_t viol;     // OR of the per-column convergence violations of the current iteration
    uint32_t error;    // row-sharded runs: some rank's column sums were not finite (the status word of the all-reduce)
    uint32_t arrived;  // workgroups of the update kernel that are through (the last one applies the stop rule)
};

// One problem of the grid bin as the device describes it to the host (emGridDescribeKernel, em_sparse.hip).
struct EmGridProblem {
    uint32_t problem, columns;  // index in the list; paths + noise
    uint32_t rows, entries;     // kept rows / entries (its compacted CSR)
    uint32_t merged, pad;       // the row collapse merged rows of it: the EM reads the merged read counts
    uint64_t row_base, ent_base, col_begin;
    double total_mass, zero_mass;
};

// device arrays of the solve the problems belong to
struct EmGridStorage {
    const uint32_t * prow_off;
    const double * prow_count;
    const double * merged_count;  // NULL: no collapse
    const double * prow_noise;
    const uint32_t * pent_col;
    const double * pent_val;
    double * abundances;
    double * noise_count;
    uint32_t * iterations;
    const EmFusedDense * fused = nullptr;  // problems whose dense matrix exists already
    uint32_t num_fused = 0;
};

// rows + entries from which a problem leaves the one-workgroup kernels (RPVG_HIP_EM_GRID_MIN_WORK; 0: never)
uint64_t emGridMinWork();
// whether a problem of the grid bin is solved on a dense row-major copy (em_dense.hip's streaming kernels) rather than
// on its CSR: the dense matrix is the smaller one (8 B per cell against 12 B per entry) and narrow enough for a
// register-resident row
bool emGridDenseRoute(uint32_t columns, uint32_t rows, uint32_t entries);
// Solves the described problems one after the other on `st`, every launch over the whole GPU; host-driven (waits for the
// stream).  Caller holds ctx->mutex and has set the device.
int runEmGridProblems(rpvg_hip_ctx * ctx, hipStream_t st, const EmGridProblem * problems, uint32_t count, const EmGridStorage & storage,
                      uint32_t max_em_its, double max_rel_em_conv);

// The dense streaming EM of em_dense.hip on a resident row-major matrix, up to the stop rule; the abundance vector stays
// on the device (d_a, C doubles).  zero_mass: read mass of the rows the matrix does not hold because they touch no
// selected path — the noise component takes it whole (em_sparse.hip).  Caller holds ctx->mutex and has set the device.
struct DenseEmRun {
    const double * matrix = nullptr;
    uint64_t num_rows = 0;
    uint32_t num_cols = 0;
    uint64_t ld = 0;
    const double * counts = nullptr;
    double total_count = 0, zero_mass = 0;
    uint32_t max_em_its = 0;
    double max_rel_em_conv = 0;
    bool sharded = false;
    DeviceBuffer<double> d_a;
    EmGridControl control = {};
};
int emDenseIterate(rpvg_hip_ctx * ctx, const char * who, DenseEmRun & run);

// collapse_precision > 0: readCollapseProbabilityMatrix on the rows of every problem (prob_precision of the reference)
int queueEmSolve(rpvg_hip_ctx * ctx, const rpvg_hip_batch * batch, const EmProblemList & list, uint32_t max_em_its,
                 double max_rel_em_conv, const EmOutputs & out, EmSolveWork & work, bool fill_only, double collapse_precision = 0.0);
void accountEmSolve(rpvg_hip_ctx * ctx, uint32_t P, const uint64_t * col_off, const uint32_t * kept_rows, const uint32_t * kept_entries,
                    const uint32_t * iterations);

// ---- the diploid search, queued (bounded_search.hip) -----------------------------------------------
struct PairSearchWork {  // what one search leaves on the device (and the host arrays its statistics need)
    uint32_t M = 0, num_big = 0, evals_word = 0, tail_words = 0;
    std::vector<uint32_t> order;
    std::vector<uint64_t> col_off, pair_cap_off;
    DeviceBuffer<uint32_t> d_order, d_col_count, d_col_order, d_out_first, d_out_second, d_item_matrix, d_item_col, d_item_chunk;
    DeviceBuffer<uint32_t> d_tail;  // [kept pairs per matrix: M words | evaluation counter: 2 words | validity flag of the build | -]
    DeviceBuffer<uint64_t> d_col_off, d_pair_cap_off, d_big_col_part_off, d_big_pair_part_off;
    DeviceBuffer<double> d_lf, d_marg, d_opt_raw, d_opt, d_out_value, d_part_marg, d_part_opt, d_part_pair, d_seq;
    UploadPack pack;
    // what the caller wants uploaded / zeroed with the search's own small arrays (one copy, one memset for everything)
    const uint64_t * extra_u64 = nullptr;
    size_t extra_u64_count = 0, extra_zero_bytes = 0;
    DeviceBuffer<uint64_t> d_extra_u64;
    DeviceBuffer<unsigned char> d_extra_zero;
};
int queuePairSearch(rpvg_hip_ctx * ctx, const rpvg_hip_groups * groups, const uint32_t * column_counts, double min_rel_likelihood,
                    PairSearchWork & w);
void leavePairSearch(rpvg_hip_ctx * ctx);  // the next search of another context may start behind what has been queued so far
void accountPairSearch(rpvg_hip_ctx * ctx, const rpvg_hip_groups * groups, const PairSearchWork & w, unsigned long long log_evals, uint64_t kept_pairs);

// ---- row collapse of the EM problems (row_collapse.hip) ----------------------------------------------
struct CsrCollapseInput {  // the compacted CSR of a solve's problems (em_sparse.hip), device pointers
    uint32_t num_problems_bound = 0;
    const uint32_t * num_problems_dev = nullptr;
    uint64_t rows_capacity = 0;
    const uint64_t * row_base = nullptr;
    const uint64_t * ent_base = nullptr;
    const uint32_t * kept_rows = nullptr;
    const uint64_t * col_off = nullptr;
    const uint32_t * prow_off = nullptr;
    const double * prow_count = nullptr;
    const double * prow_noise = nullptr;
    const uint32_t * pent_col = nullptr;
    const double * pent_val = nullptr;
    // the work items of the fill (EmProblemList): problem p has items seg_first[p] .. seg_first[p + 1], of segment_rows row slots each
    uint32_t num_items_bound = 0;
    const uint32_t * num_items_dev = nullptr;
    const uint64_t * seg_first = nullptr;
    const uint32_t * item_problem = nullptr;
    uint32_t segment_rows = 0;
    uint64_t max_rows_bound = 0;  // a bound of the rows of the largest problem (its cluster's): 0 = unknown
};
struct CsrCollapseWork {
    DeviceBuffer<double> merged_count;      // [rows] read counts after the merges, for the problems with merges
    DeviceBuffer<uint32_t> problem_merged;  // [P] flag per problem, [P]: their number
    DeviceBuffer<uint32_t> info;
    std::shared_ptr<void> temporaries;
};
// sorted: recorded behind the sort of the problems' rows (the first stages of the collapse: most of its time), if not null
hipError_t queueCsrCollapse(rpvg_hip_ctx * ctx, const CsrCollapseInput & in, double precision, CsrCollapseWork & work, hipStream_t stream,
                            hipEvent_t sorted = nullptr);

// The copies of PathInfo::group_id / source_ids of `hb` on the context's stream; then, on the stream of any context of the
// device, the kernel that forms the haplotype columns of every cluster and the copy of their sizes back to the host;
// finishPathSources() reads those once that stream has been waited for.  The caller holds ctx->mutex and has set the device.
// read count of every cluster (exact integer sums) from the 32-bit counts as uploaded
hipError_t queueClusterTotals(hipStream_t stream, uint32_t num_clusters, const uint64_t * d_cluster_row_off, const uint32_t * d_row_count_u32, double * d_totals);
hipError_t queuePathSourceCopies(rpvg_hip_ctx * ctx, rpvg_hip_batch * b, const rpvg_cluster_batch * hb, PathSourcesPending & pending);
hipError_t queuePathSourceKernels(rpvg_hip_ctx * ctx, rpvg_hip_batch * b, PathSourcesPending & pending, hipStream_t stream);
// what queuePathSourceCopies reserves for the kernels (column slots, scratch arena, sizes), for a batch whose path arrays some
// kernel writes on the device instead of a copy from the host (rpvg_hip_batch_upload_segments): b->path_group_id,
// pending.d_path_source_off / d_source_id and b->cluster_src_off are allocated, not filled; h_cluster_src_off is the caller's
hipError_t reservePathSources(rpvg_hip_batch * b, uint32_t num_clusters, uint64_t num_paths, uint64_t num_sources, PathSourcesPending & pending);
// RPVG_HIP_OK; RPVG_HIP_ERR_INVALID for inconsistent offsets.  A batch whose id ranges outgrow the scratch set aside for them
// simply has no source columns (has_source_columns stays false: the caller groups on the host).
int finishPathSources(rpvg_hip_batch * b, PathSourcesPending & pending);

// queues the replay of readCollapseProbabilityMatrix on the matrices of `groups` behind their build (row_collapse.hip)
// hold_back_runs: everything but the last stage (rpvg_hip_groups::held_back_runs receives that one)
hipError_t queueRowCollapse(rpvg_hip_ctx * ctx, rpvg_hip_groups * groups, uint64_t total_rows, double precision, hipStream_t stream, bool hold_back_runs = false);
}

#endif
