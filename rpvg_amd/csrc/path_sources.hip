// The path side of a cluster batch on the device (gfx950): the haplotype columns of every cluster.
//
// NestedPathAbundanceEstimator::findPathSourceGroups (src/path_abundance_estimator.cpp:493-546) turns the PathInfo::source_ids
// of a cluster's paths around — haplotype (source id) -> the list of the paths it carries — and makes one column of every
// distinct list; a column's multiplicity (path_counts of the posterior calculation) is the number of haplotypes that carry
// exactly that list.  The host did this per batch with a counting sort and a hash table per cluster: 8.7 ms of single-thread
// work per 200 k-path batch, the first thing every host lane did while the GPU waited (docs/design/host-orchestration.md).
// Here it is part of the batch's upload: the ids travel with the rows (one more copy), and one workgroup per cluster
//   1. finds the cluster's id range [min, max];
//   2. builds, per haplotype of the range, the SET of its paths as a bit vector over the cluster's paths (one 64-bit word per
//      64 paths; atomic OR per (path, id) incidence) — two haplotypes carry the same list iff their vectors are equal, exactly;
//   3. groups equal vectors through an open-addressing table keyed by a hash of the vector, with a word-by-word comparison
//      against the slot's first claimant (no false merges), counting the members of every group and keeping its smallest id;
//   4. numbers the groups by ascending smallest id (the order of the host classes: rpvg_amd/host/path_abundance_estimator.cpp
//      findPathSourceGroups; the reference's is that of its hash map) with a prefix sum over the id range;
//   5. writes, per column, its multiplicity and its path list (the set bits, ascending).
// Bit vectors and tables live in LDS for clusters of up to 1 024 ids and 3 072 words, otherwise in a scratch arena in device
// memory that workgroups carve up with an atomic cursor; a batch whose id ranges outgrow the arena (ids that are not small
// consecutive integers) is left to the host's grouping (rpvg_hip_batch::has_source_columns stays false).
//
// The columns are laid out by bounds (a cluster has at most as many columns, and lists at most as many paths, as it has
// incidences), so nothing is sized on the host in between; rpvg_hip_groups_build_from_sources (loglik.hip) gathers the
// columns of the clusters it is asked for into the compact arrays the build kernels take.

#include "common.hpp"

using namespace rpvg_hip_detail;

namespace {

constexpr uint32_t kEmptySlot = 0xffffffffu;
constexpr uint32_t kLdsWords = 3072;        // 64-bit words of bit vectors a workgroup keeps in LDS (24 KB; 56 KB with the tables)
constexpr uint32_t kLdsHaplotypes = 1024;   // id range up to which the tables stay in LDS
constexpr int kBlock = 256;
// Most clusters are small — tens of paths, tens of haplotypes: a wavefront each, 6 KB of LDS, many per CU (5 000 workgroups of
// 256 threads and 56 KB took 0.41 ms per configs[2] batch); what does not fit goes on a list for the launch of large workgroups.
constexpr uint32_t kSmallLdsWords = 256, kSmallLdsHaplotypes = 128;
constexpr int kSmallBlock = 64;

struct SourceArgs {
    uint32_t num_clusters;
    const uint64_t * cluster_path_off;   // [K+1]
    const uint64_t * path_source_off;    // [P+1]
    const uint32_t * source_id;          // [S]
    unsigned long long num_sources;
    unsigned long long * arena;          // scratch of the clusters that do not fit LDS
    unsigned long long arena_words;
    unsigned long long * arena_cursor;
    uint32_t * col_count;                // by bounds: cluster k from slot path_source_off[cluster_path_off[k]]
    uint32_t * col_end;
    uint32_t * col_path;
    uint32_t * num_cols;                 // [K]
    uint32_t * num_col_paths;            // [K]
    uint32_t * max_col_paths;            // [K]
    uint32_t * flags;                    // [0] inconsistent offsets, [1] clusters the arena had no room for
    uint32_t * big_list;                 // [K] clusters the small workgroups left to the large ones
    uint32_t * big_count;                // their number
};

__device__ __forceinline__ unsigned long long mixHash(unsigned long long h) {
    h ^= h >> 33;
    h *= 0xff51afd7ed558ccdull;
    h ^= h >> 33;
    return h;
}

// SMALL: the first launch, over all clusters — a cluster that does not fit its LDS is put on the list; otherwise the launch over the
// list (a workgroup whose index is beyond the list's length has nothing to do), with the arena for what does not fit LDS either.
template <int BLOCK, uint32_t LDS_WORDS, uint32_t LDS_HAPLOTYPES, bool SMALL>
__global__ __launch_bounds__(BLOCK) void sourceColumnsKernel(const SourceArgs a) {
    constexpr int kBlock = BLOCK;
    constexpr int kWaves = BLOCK / 64;
    constexpr uint32_t kLdsWords = LDS_WORDS, kLdsHaplotypes = LDS_HAPLOTYPES, kLdsTable = 2 * LDS_HAPLOTYPES;
    __shared__ unsigned long long s_bits[kLdsWords];
    __shared__ uint32_t s_owner[kLdsTable], s_min[kLdsTable], s_cnt[kLdsTable];
    __shared__ uint32_t s_group[kLdsHaplotypes], s_rep[kLdsHaplotypes];
    __shared__ uint32_t s_scan[kWaves], s_lo[kWaves], s_hi[kWaves];
    __shared__ unsigned long long s_base;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // (the large workgroups: a few of them walk the list — five thousand that look at its length and leave took 80 us to dispatch)
    for (uint32_t turn = blockIdx.x; SMALL ? turn == blockIdx.x : turn < *a.big_count; turn += gridDim.x) {
    __syncthreads();  // (the LDS of the cluster before)
    const uint32_t k = SMALL ? turn : a.big_list[turn];
    if (k >= a.num_clusters) return;
    const uint64_t p0 = a.cluster_path_off[k], p1 = a.cluster_path_off[k + 1];
    const uint32_t N = static_cast<uint32_t>(p1 - p0);
    const uint64_t i0 = a.path_source_off[p0], i1 = a.path_source_off[p1];
    auto leave = [&](const uint32_t cols, const uint32_t flag) {
        if (tid == 0) {
            a.num_cols[k] = cols;
            a.num_col_paths[k] = 0;
            a.max_col_paths[k] = 0;
            if (flag < 2) a.flags[flag] = 1;
        }
    };
    if (i1 < i0 || i1 > a.num_sources || i1 - i0 > 0xfffffffeull) {
        leave(0, 0);
        continue;
    }
    if (i1 == i0 || N == 0) {  // a cluster without haplotype ids: the estimator that needs columns says so
        leave(0, 2);
        continue;
    }
    // 1. the id range
    uint32_t lo = 0xffffffffu, hi = 0;
    for (uint64_t i = i0 + tid; i < i1; i += kBlock) {
        const uint32_t id = a.source_id[i];
        lo = min(lo, id);
        hi = max(hi, id);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        lo = min(lo, static_cast<uint32_t>(__shfl_xor(lo, d, 64)));
        hi = max(hi, static_cast<uint32_t>(__shfl_xor(hi, d, 64)));
    }
    if (lane == 0) {
        s_lo[wave] = lo;
        s_hi[wave] = hi;
    }
    __syncthreads();
    lo = s_lo[0];
    hi = s_hi[0];
#pragma unroll
    for (int w = 1; w < kWaves; ++w) {
        lo = min(lo, s_lo[w]);
        hi = max(hi, s_hi[w]);
    }
    const unsigned long long H = static_cast<unsigned long long>(hi - lo) + 1;
    const uint32_t W = (N + 63) / 64;
    const unsigned long long words = H * W;
    unsigned long long T = 16;
    while (T < 2 * H) T <<= 1;
    unsigned long long * bits = s_bits;
    uint32_t * owner = s_owner, * smin = s_min, * scnt = s_cnt, * group = s_group, * rep = s_rep;
    if (SMALL && !(words <= kLdsWords && H <= kLdsHaplotypes)) {  // (for a large workgroup)
        if (tid == 0) a.big_list[atomicAdd(a.big_count, 1u)] = k;
        continue;
    }
    if (!(words <= kLdsWords && H <= kLdsHaplotypes)) {
        const unsigned long long need = words + (3 * T + 2 * H + 1) / 2;
        if (tid == 0) s_base = (need <= a.arena_words) ? atomicAdd(a.arena_cursor, need) : a.arena_words;
        __syncthreads();
        const unsigned long long base = s_base;
        if (need > a.arena_words || base + need > a.arena_words) {
            leave(0, 1);
            continue;
        }
        bits = a.arena + base;
        owner = reinterpret_cast<uint32_t *>(bits + words);
        smin = owner + T;
        scnt = smin + T;
        group = scnt + T;
        rep = group + H;
    }
    for (unsigned long long w = tid; w < words; w += kBlock) bits[w] = 0ull;
    for (unsigned long long t = tid; t < T; t += kBlock) {
        owner[t] = kEmptySlot;
        smin[t] = 0xffffffffu;
        scnt[t] = 0;
    }
    __syncthreads();
    // 2. the path set of every haplotype: a wave per path, its lanes over the path's ids
    bool bad = false;
    for (uint32_t p = wave; p < N; p += kWaves) {
        const uint64_t b = a.path_source_off[p0 + p], e = a.path_source_off[p0 + p + 1];
        if (b > e || b < i0 || e > i1) {
            bad = true;
            continue;
        }
        for (uint64_t i = b + lane; i < e; i += 64) {
            const uint32_t id = a.source_id[i];
            atomicOr(&bits[static_cast<unsigned long long>(id - lo) * W + (p >> 6)], 1ull << (p & 63));
        }
    }
    if (bad && lane == 0) a.flags[0] = 1;
    __syncthreads();
    // 3. equal sets -> one group; its members counted, its smallest id kept
    for (unsigned long long h = tid; h < H; h += kBlock) {
        const unsigned long long * row = bits + h * W;
        unsigned long long hash = 1469598103934665603ull, any = 0;
        for (uint32_t w = 0; w < W; ++w) {
            const unsigned long long v = row[w];
            any |= v;
            hash = (hash ^ v) * 1099511628211ull;
        }
        uint32_t g = kEmptySlot;
        if (any) {
            unsigned long long slot = mixHash(hash) & (T - 1);
            while (true) {
                const uint32_t prev = atomicCAS(&owner[slot], kEmptySlot, static_cast<uint32_t>(h));
                if (prev == kEmptySlot) break;  // (the slot is this haplotype's)
                const unsigned long long * other = bits + static_cast<unsigned long long>(prev) * W;
                bool same = true;
                for (uint32_t w = 0; same && w < W; ++w) same = (row[w] == other[w]);
                if (same) break;
                slot = (slot + 1) & (T - 1);
            }
            g = static_cast<uint32_t>(slot);
            atomicMin(&smin[g], static_cast<uint32_t>(h));
            atomicAdd(&scnt[g], 1u);
        }
        group[h] = g;
    }
    __syncthreads();
    // 4. the groups in ascending order of their smallest id
    uint32_t G = 0;
    for (unsigned long long c0 = 0; c0 < H; c0 += kBlock) {
        const unsigned long long h = c0 + tid;
        const uint32_t g = h < H ? group[h] : kEmptySlot;
        const uint32_t first = (g != kEmptySlot && smin[g] == static_cast<uint32_t>(h)) ? 1u : 0u;
        uint32_t total;
        const uint32_t before = blockExclusiveSum<kBlock>(first, total, s_scan);
        if (first) rep[G + before] = static_cast<uint32_t>(h);
        G += total;
    }
    __syncthreads();
    // 5. multiplicities, list lengths, lists
    uint32_t run = 0, longest = 0;
    for (uint32_t c0 = 0; c0 < G; c0 += kBlock) {
        const uint32_t c = c0 + tid;
        uint32_t len = 0;
        if (c < G) {
            const unsigned long long * row = bits + static_cast<unsigned long long>(rep[c]) * W;
            for (uint32_t w = 0; w < W; ++w) len += __popcll(row[w]);
        }
        uint32_t total;
        const uint32_t before = blockExclusiveSum<kBlock>(len, total, s_scan);
        if (c < G) {
            a.col_end[i0 + c] = run + before + len;
            a.col_count[i0 + c] = scnt[group[rep[c]]];
        }
        longest = max(longest, len);
        run += total;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) longest = max(longest, static_cast<uint32_t>(__shfl_xor(longest, d, 64)));
    __syncthreads();  // (col_end of the whole cluster is written; s_hi is free again)
    if (lane == 0) s_hi[wave] = longest;
    __syncthreads();
    for (uint32_t c = wave; c < G; c += kWaves) {
        const unsigned long long * row = bits + static_cast<unsigned long long>(rep[c]) * W;
        uint32_t * out = a.col_path + i0 + (c ? a.col_end[i0 + c - 1] : 0u);
        uint32_t written = 0;
        for (uint32_t w = 0; w < W; ++w) {
            const unsigned long long v = row[w];
            if ((v >> lane) & 1ull) out[written + __popcll(v & ((1ull << lane) - 1ull))] = w * 64 + lane;
            written += __popcll(v);
        }
    }
    if (tid == 0) {
        a.num_cols[k] = G;
        a.num_col_paths[k] = run;
        uint32_t longest_of_all = s_hi[0];
        for (int w = 1; w < kWaves; ++w) longest_of_all = max(longest_of_all, s_hi[w]);
        a.max_col_paths[k] = longest_of_all;
    }
    }  // (the next cluster of this workgroup's)
}

// read count of every cluster: exact in 64-bit integers, one workgroup per cluster
__global__ __launch_bounds__(kBlock) void clusterTotalsKernel(const uint32_t num_clusters, const uint64_t * __restrict__ cluster_row_off,
                                                              const uint32_t * __restrict__ row_count, double * __restrict__ totals) {
    __shared__ unsigned long long s_sum[kBlock / 64];
    const uint32_t k = blockIdx.x;
    if (k >= num_clusters) return;
    unsigned long long sum = 0;
    for (uint64_t r = cluster_row_off[k] + threadIdx.x; r < cluster_row_off[k + 1]; r += kBlock) sum += row_count[r];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, 64);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) totals[k] = static_cast<double>(s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3]);
}

}  // namespace

namespace rpvg_hip_detail {

hipError_t queueClusterTotals(hipStream_t stream, const uint32_t num_clusters, const uint64_t * d_cluster_row_off, const uint32_t * d_row_count_u32,
                              double * d_totals) {
    if (num_clusters == 0) return hipSuccess;
    clusterTotalsKernel<<<dim3(num_clusters), dim3(kBlock), 0, stream>>>(num_clusters, d_cluster_row_off, d_row_count_u32, d_totals);
    return hipGetLastError();
}

// the column slots of every cluster (by bounds: as many as it has (haplotype, path) incidences), the scratch of the clusters whose
// id range or path count outgrows LDS (eight words per incidence, at least 128 MB) and the sizes that come back
static hipError_t reserveColumnSlots(rpvg_hip_batch * b, const uint32_t K, const uint64_t S, PathSourcesPending & pending) {
    hipError_t e = hipSuccess;
    auto ok = [&](hipError_t r) { if (e == hipSuccess) e = r; return e == hipSuccess; };
    ok(b->src_col_count.alloc(S));
    ok(b->src_col_end.alloc(S));
    ok(b->src_col_path.alloc(S));
    pending.arena_words = std::max<unsigned long long>(1ull << 24, 8ull * S);
    pending.num_sources = S;
    ok(pending.d_arena.alloc(pending.arena_words));
    ok(pending.d_sizes.alloc(4 * static_cast<size_t>(K) + 8));  // (layout: queuePathSourceKernels)
    if (e == hipSuccess && pinnedAlloc(&pending.h_sizes, (3 * static_cast<size_t>(K) + 4) * sizeof(uint32_t)) != hipSuccess) e = hipErrorOutOfMemory;
    return e;
}

hipError_t queuePathSourceCopies(rpvg_hip_ctx * ctx, rpvg_hip_batch * b, const rpvg_cluster_batch * hb, PathSourcesPending & pending) {
    const uint32_t K = hb->num_clusters;
    const uint64_t P = hb->cluster_path_off[K];
    pending.K = K;
    if (K == 0 || P == 0 || !hb->path_source_off || !hb->path_group_id) return hipSuccess;
    const uint64_t S = hb->path_source_off[P];
    if (S == 0 || !(hb->source_id || hb->source_id16) || S > 0xfffffff0ull) return hipSuccess;
    hipStream_t st = ctx->stream;
    hipError_t e = hipSuccess;
    auto ok = [&](hipError_t r) { if (e == hipSuccess) e = r; return e == hipSuccess; };
    b->h_cluster_src_off.resize(K + 1);
    for (uint32_t k = 0; k <= K; ++k) b->h_cluster_src_off[k] = hb->path_source_off[hb->cluster_path_off[k]];
    ok(b->path_group_id.upload(hb->path_group_id, P, st));
    ok(pending.d_path_source_off.upload(hb->path_source_off, P + 1, st));
    if (hb->source_id16) {  // (widened behind the copy: widenSourceIds)
        ok(pending.d_source_id16.upload(hb->source_id16, S, st));
        pending.num_sources_narrow = S;
        ok(pending.d_source_id.alloc(S));
    } else {
        ok(pending.d_source_id.upload(hb->source_id, S, st));
    }
    ok(b->cluster_src_off.upload(b->h_cluster_src_off.data(), K + 1, st));
    ctx->stats.h2d_bytes += static_cast<double>(P * 12 + S * (hb->source_id16 ? 2 : 4) + K * 8);
    ok(reserveColumnSlots(b, K, S, pending));
    pending.copied = (e == hipSuccess);
    return e;
}

hipError_t reservePathSources(rpvg_hip_batch * b, const uint32_t K, const uint64_t P, const uint64_t S, PathSourcesPending & pending) {
    pending.K = K;
    if (K == 0 || P == 0 || S == 0 || S > 0xfffffff0ull) return hipSuccess;
    hipError_t e = hipSuccess;
    auto ok = [&](hipError_t r) { if (e == hipSuccess) e = r; return e == hipSuccess; };
    ok(b->path_group_id.alloc(P));
    ok(pending.d_path_source_off.alloc(P + 1));
    ok(pending.d_source_id.alloc(S));
    ok(b->cluster_src_off.alloc(K + 1));
    ok(reserveColumnSlots(b, K, S, pending));
    pending.copied = (e == hipSuccess);
    return e;
}

hipError_t queuePathSourceKernels(rpvg_hip_ctx * ctx, rpvg_hip_batch * b, PathSourcesPending & pending, hipStream_t st) {
    (void) ctx;
    if (!pending.copied) return hipSuccess;
    const uint32_t K = pending.K;
    hipError_t e = hipSuccess;
    auto ok = [&](hipError_t r) { if (e == hipSuccess) e = r; return e == hipSuccess; };
    // d_sizes: [the arena's cursor, 64 bits | flags 2 | length of the large workgroups' list | - | sizes 3K | that list K]: the first six
    // words are zeroed by one memset; words 2 .. 6 + 3K come back to the host
    uint32_t * const words = pending.d_sizes.ptr;
    ok(zeroAsync(words, 6 * sizeof(uint32_t), st));
    SourceArgs a;
    a.num_clusters = K;
    a.cluster_path_off = b->cluster_path_off.ptr;
    a.path_source_off = pending.d_path_source_off.ptr;
    a.source_id = pending.d_source_id.ptr;
    a.num_sources = pending.num_sources;
    a.arena = pending.d_arena.ptr;
    a.arena_words = pending.arena_words;
    a.arena_cursor = reinterpret_cast<unsigned long long *>(words);
    a.col_count = b->src_col_count.ptr;
    a.col_end = b->src_col_end.ptr;
    a.col_path = b->src_col_path.ptr;
    a.flags = words + 2;
    a.big_count = words + 4;
    a.num_cols = words + 6;
    a.num_col_paths = words + 6 + K;
    a.max_col_paths = words + 6 + 2 * static_cast<size_t>(K);
    a.big_list = words + 6 + 3 * static_cast<size_t>(K);
    if (e == hipSuccess) {
        sourceColumnsKernel<kSmallBlock, kSmallLdsWords, kSmallLdsHaplotypes, true><<<dim3(K), dim3(kSmallBlock), 0, st>>>(a);
        sourceColumnsKernel<kBlock, kLdsWords, kLdsHaplotypes, false><<<dim3(std::min<uint32_t>(K, 128)), dim3(kBlock), 0, st>>>(a);
        ok(hipGetLastError());
    }
    ok(hipMemcpyAsync(pending.h_sizes, words + 2, (3 * static_cast<size_t>(K) + 4) * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    pending.queued = (e == hipSuccess);
    return e;
}

int finishPathSources(rpvg_hip_batch * b, PathSourcesPending & pending) {
    if (!pending.queued) return RPVG_HIP_OK;
    const uint32_t K = pending.K;
    const uint32_t * flags = static_cast<const uint32_t *>(pending.h_sizes);  // [flags 2 | length of the large list | - | sizes 3K]
    const uint32_t * sizes = flags + 4;
    if (flags[0]) {
        setError("rpvg_hip_batch_upload: path_source_off is not a non-decreasing sequence of offsets into source_id");
        return RPVG_HIP_ERR_INVALID;
    }
    if (flags[1]) {  // id ranges too wide for the scratch: the caller groups on the host
        b->src_col_count.release();
        b->src_col_end.release();
        b->src_col_path.release();
        return RPVG_HIP_OK;
    }
    b->h_src_num_cols.assign(sizes, sizes + K);
    b->h_src_col_paths.assign(sizes + K, sizes + 2 * static_cast<size_t>(K));
    b->h_src_max_col_paths.assign(sizes + 2 * static_cast<size_t>(K), sizes + 3 * static_cast<size_t>(K));
    b->has_source_columns = true;
    return RPVG_HIP_OK;
}

}  // namespace rpvg_hip_detail

extern "C" {

int rpvg_hip_batch_has_source_columns(const rpvg_hip_batch * batch) {
    return (batch && batch->has_source_columns) ? 1 : 0;
}

int rpvg_hip_batch_cluster_totals(const rpvg_hip_batch * batch, double * totals_out, uint32_t num_clusters) {
    RPVG_REQUIRE(batch && (totals_out || num_clusters == 0), "rpvg_hip_batch_cluster_totals: NULL argument");
    RPVG_REQUIRE(batch->h_cluster_total.size() == num_clusters, "rpvg_hip_batch_cluster_totals: the batch has %llu clusters with a total, not %u",
                 static_cast<unsigned long long>(batch->h_cluster_total.size()), num_clusters);
    std::copy(batch->h_cluster_total.begin(), batch->h_cluster_total.end(), totals_out);
    return RPVG_HIP_OK;
}

int rpvg_hip_batch_source_columns_sizes(const rpvg_hip_batch * batch, uint32_t cluster, uint32_t * num_columns_out, uint32_t * num_column_paths_out) {
    RPVG_REQUIRE(batch && num_columns_out && num_column_paths_out, "rpvg_hip_batch_source_columns_sizes: NULL argument");
    RPVG_REQUIRE(batch->has_source_columns, "rpvg_hip_batch_source_columns_sizes: the batch has no source columns");
    RPVG_REQUIRE(cluster < batch->num_clusters, "rpvg_hip_batch_source_columns_sizes: cluster %u of %u", cluster, batch->num_clusters);
    *num_columns_out = batch->h_src_num_cols[cluster];
    *num_column_paths_out = batch->h_src_col_paths[cluster];
    return RPVG_HIP_OK;
}

int rpvg_hip_batch_source_columns_get(rpvg_hip_ctx * ctx, const rpvg_hip_batch * batch, uint32_t cluster, uint32_t * column_counts_out,
                                      uint32_t * column_path_end_out, uint32_t * column_paths_out) {
    RPVG_REQUIRE(ctx && batch && column_counts_out && column_path_end_out && column_paths_out, "rpvg_hip_batch_source_columns_get: NULL argument");
    RPVG_REQUIRE(batch->has_source_columns, "rpvg_hip_batch_source_columns_get: the batch has no source columns");
    RPVG_REQUIRE(cluster < batch->num_clusters, "rpvg_hip_batch_source_columns_get: cluster %u of %u", cluster, batch->num_clusters);
    std::lock_guard<std::mutex> lock(ctx->mutex);
    RPVG_HIP_CHECK(hipSetDevice(ctx->device));
    const uint64_t slot0 = batch->h_cluster_src_off[cluster];
    const uint32_t G = batch->h_src_num_cols[cluster], L = batch->h_src_col_paths[cluster];
    if (G) {
        RPVG_HIP_CHECK(hipMemcpyAsync(column_counts_out, batch->src_col_count.ptr + slot0, G * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        RPVG_HIP_CHECK(hipMemcpyAsync(column_path_end_out, batch->src_col_end.ptr + slot0, G * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    }
    if (L) RPVG_HIP_CHECK(hipMemcpyAsync(column_paths_out, batch->src_col_path.ptr + slot0, L * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    RPVG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return RPVG_HIP_OK;
}

}  // extern "C"
