// path_clusters.hip — path clustering on the GPU.
//
// Takes over PathClusters::createPathClusters (src/path_clusters.cpp:163-207) together with the construction of its
// input (constructor :12-86: every alignment-path list connects all the paths it locates to its anchor path) and the
// node-sharing refinement (addNodeClusters / mergeClusters :88-262): all three say "paths that occur together in one
// id set belong to one cluster".  The reference materialises adjacency hash sets under mutexes and runs a BFS per
// component; here the sets drive a lock-free union-find (the smaller root id always wins, so a component's root is
// its smallest path id), followed by the reference's canonical numbering: clusters by ascending smallest path id,
// members ascending (src/path_clusters.cpp:172-204; test src/tests/path_clusters_test.cpp:82-87).

#include <hipcub/hipcub.hpp>

#include "common.hpp"

using namespace rpvg_hip_detail;

namespace {

__device__ __forceinline__ uint32_t findRoot(uint32_t * parent, uint32_t x) {
    uint32_t p = __hip_atomic_load(parent + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (p != x) {
        const uint32_t gp = __hip_atomic_load(parent + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // path halving with a plain store: any ancestor is a valid parent, so racing halvings cannot break the forest
        if (gp != p) __hip_atomic_store(parent + x, gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        x = p;
        p = gp;
    }
    return x;
}

__global__ void initParentKernel(const uint32_t n, uint32_t * parent) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) parent[i] = i;
}

// one thread per id set: joins every member with the first one (the reference's anchor, :31-47); sets are short
// (the paths one read aligns to), a few hundred members at most
__global__ void unionSetsKernel(const uint64_t num_sets, const uint64_t * __restrict__ set_off,
                                const uint32_t * __restrict__ set_path, uint32_t * parent) {
    const uint64_t s = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (s >= num_sets) return;
    const uint64_t first = set_off[s], last = set_off[s + 1];
    uint32_t anchor = set_path[first];
    for (uint64_t e = first + 1; e < last; ++e) {
        uint32_t a = anchor, b = set_path[e];
        while (true) {
            a = findRoot(parent, a);
            b = findRoot(parent, b);
            if (a == b) break;
            if (a > b) {
                const uint32_t t = a;
                a = b;
                b = t;
            }
            if (atomicCAS(parent + b, b, a) == b) break;  // b was still a root: hooked under the smaller root
        }
        anchor = a;  // the current root: shorter walks for the next member
    }
}

__global__ void flattenKernel(const uint32_t n, uint32_t * parent, uint32_t * is_root) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t r = i;
    while (parent[r] != r) r = parent[r];  // read-only: all unions are done
    is_root[i] = (r == i);
    parent[i] = r;  // benign race with the walkers above: they see the old parent or the root, both ancestors
}

__global__ void labelKernel(const uint32_t n, const uint32_t * __restrict__ root, const uint32_t * __restrict__ root_rank,
                            uint32_t * __restrict__ path_to_cluster, uint32_t * __restrict__ path_id, uint64_t * __restrict__ cluster_size) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t c = root_rank[root[i]];  // exclusive count of roots below the component's smallest path id
    path_to_cluster[i] = c;
    path_id[i] = i;
    atomicAdd(reinterpret_cast<unsigned long long *>(cluster_size + c), 1ull);
}

}  // namespace

extern "C" int rpvg_hip_path_clusters(rpvg_hip_ctx * ctx, uint32_t num_paths, uint64_t num_sets, const uint64_t * set_off,
                                      const uint32_t * set_path, uint32_t * path_to_cluster, uint32_t * num_clusters_out,
                                      uint64_t * cluster_off, uint32_t * cluster_paths) {
    RPVG_REQUIRE(ctx && path_to_cluster && num_clusters_out && cluster_off && cluster_paths, "rpvg_hip_path_clusters: NULL argument");
    RPVG_REQUIRE(num_sets == 0 || (set_off && set_path), "rpvg_hip_path_clusters: NULL set arrays");
    *num_clusters_out = 0;
    cluster_off[0] = 0;
    if (num_paths == 0) return RPVG_HIP_OK;
    const uint64_t num_members = num_sets ? set_off[num_sets] : 0;
    for (uint64_t s = 0; s < num_sets; ++s) {
        RPVG_REQUIRE(set_off[s] < set_off[s + 1], "rpvg_hip_path_clusters: set %llu is empty", static_cast<unsigned long long>(s));
    }
    for (uint64_t e = 0; e < num_members; ++e) {
        RPVG_REQUIRE(set_path[e] < num_paths, "rpvg_hip_path_clusters: path id %u of %u", set_path[e], num_paths);
    }

    std::lock_guard<std::mutex> lock(ctx->mutex);
    RPVG_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const uint32_t n = num_paths;
    const dim3 block(256), grid_n((n + 255) / 256);

    DeviceBuffer<uint64_t> d_set_off, d_cluster_size, d_cluster_off;
    DeviceBuffer<uint32_t> d_set_path, d_parent, d_is_root, d_root_rank, d_label, d_path_id, d_label_sorted, d_path_sorted;
    int span = ctx->spanBegin(FAM_H2D);
    if (num_sets) {
        RPVG_HIP_CHECK(d_set_off.upload(set_off, num_sets + 1, st));
        RPVG_HIP_CHECK(d_set_path.upload(set_path, num_members, st));
    }
    ctx->spanEnd(span);
    RPVG_HIP_CHECK(d_parent.alloc(n));
    RPVG_HIP_CHECK(d_is_root.alloc(n));
    RPVG_HIP_CHECK(d_root_rank.alloc(n));
    RPVG_HIP_CHECK(d_label.alloc(n));
    RPVG_HIP_CHECK(d_path_id.alloc(n));
    RPVG_HIP_CHECK(d_label_sorted.alloc(n));
    RPVG_HIP_CHECK(d_path_sorted.alloc(n));
    RPVG_HIP_CHECK(d_cluster_size.alloc(static_cast<size_t>(n) + 1));
    RPVG_HIP_CHECK(d_cluster_off.alloc(static_cast<size_t>(n) + 1));

    span = ctx->spanBegin(FAM_BUILD);
    initParentKernel<<<grid_n, block, 0, st>>>(n, d_parent.ptr);
    if (num_members) {
        unionSetsKernel<<<dim3(static_cast<uint32_t>((num_sets + 255) / 256)), block, 0, st>>>(num_sets, d_set_off.ptr, d_set_path.ptr,
                                                                                            d_parent.ptr);
    }
    flattenKernel<<<grid_n, block, 0, st>>>(n, d_parent.ptr, d_is_root.ptr);
    ctx->spanEnd(span);
    ctx->stats.build_launches += 3;
    RPVG_HIP_CHECK(hipGetLastError());
    {
        size_t bytes = 0;
        RPVG_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, d_is_root.ptr, d_root_rank.ptr, static_cast<int>(n), st));
        DeviceBuffer<uint8_t> tmp;
        RPVG_HIP_CHECK(tmp.alloc(bytes ? bytes : 1));
        RPVG_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(tmp.ptr, bytes, d_is_root.ptr, d_root_rank.ptr, static_cast<int>(n), st));
        RPVG_HIP_CHECK(hipStreamSynchronize(st));
    }
    RPVG_HIP_CHECK(hipMemsetAsync(d_cluster_size.ptr, 0, sizeof(uint64_t) * (static_cast<size_t>(n) + 1), st));
    labelKernel<<<grid_n, block, 0, st>>>(n, d_parent.ptr, d_root_rank.ptr, d_label.ptr, d_path_id.ptr, d_cluster_size.ptr);
    RPVG_HIP_CHECK(hipGetLastError());
    {
        size_t bytes = 0;
        RPVG_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, d_cluster_size.ptr, d_cluster_off.ptr, static_cast<int>(n + 1), st));
        DeviceBuffer<uint8_t> tmp;
        RPVG_HIP_CHECK(tmp.alloc(bytes ? bytes : 1));
        RPVG_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(tmp.ptr, bytes, d_cluster_size.ptr, d_cluster_off.ptr, static_cast<int>(n + 1), st));
        RPVG_HIP_CHECK(hipStreamSynchronize(st));
    }
    {
        // stable sort by cluster: members stay in ascending path id (:203)
        size_t bytes = 0;
        RPVG_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, d_label.ptr, d_label_sorted.ptr, d_path_id.ptr, d_path_sorted.ptr,
                                                          static_cast<int>(n), 0, 32, st));
        DeviceBuffer<uint8_t> tmp;
        RPVG_HIP_CHECK(tmp.alloc(bytes ? bytes : 1));
        RPVG_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(tmp.ptr, bytes, d_label.ptr, d_label_sorted.ptr, d_path_id.ptr, d_path_sorted.ptr,
                                                          static_cast<int>(n), 0, 32, st));
        RPVG_HIP_CHECK(hipStreamSynchronize(st));
    }
    uint32_t last_rank = 0, last_is_root = 0;
    RPVG_HIP_CHECK(hipMemcpyAsync(&last_rank, d_root_rank.ptr + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    RPVG_HIP_CHECK(hipMemcpyAsync(&last_is_root, d_is_root.ptr + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    RPVG_HIP_CHECK(hipMemcpyAsync(path_to_cluster, d_label.ptr, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, st));
    RPVG_HIP_CHECK(hipMemcpyAsync(cluster_paths, d_path_sorted.ptr, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, st));
    RPVG_HIP_CHECK(hipStreamSynchronize(st));
    const uint32_t num_clusters = last_rank + last_is_root;
    RPVG_HIP_CHECK(hipMemcpyAsync(cluster_off, d_cluster_off.ptr, sizeof(uint64_t) * (static_cast<size_t>(num_clusters) + 1), hipMemcpyDeviceToHost, st));
    RPVG_HIP_CHECK(hipStreamSynchronize(st));
    *num_clusters_out = num_clusters;
    return RPVG_HIP_OK;
}
