// Weighted minimum path cover on the GPU (gfx950): the greedy set cover of
// MinimumPathAbundanceEstimator (`-i strains`).
//
// Takes over, per cluster,
//   the cover matrix / path weights set-up      src/path_abundance_estimator.cpp:233-257
//   weightedMinimumPathCover                    src/path_abundance_estimator.cpp:297-340
// ONE workgroup per cluster runs all greedy rounds: per round every still-uncovered row adds its read
// count to the paths it contains (integers in FP64: exact in any order), the block picks the path with
// the largest covered-reads / weight (first index among equals, as the reference's ascending scan
// does), and the rows containing it become covered.  Integer/compare work on the sparse rows; the only
// floating point is the path weight  -sum_i count_i log(prob_ij)  and one division per path per round.

#include "common.hpp"

#include <algorithm>

using namespace rpvg_hip_detail;

namespace {

constexpr int kBlock = 256;

// Utils::doubleCompare(x, 1) (src/utils.hpp:87-93)
__device__ __forceinline__ bool isOne(const double x) {
    const double precision = 2.220446049250313e-16 * 100;
    return (x == 1.0) || (fabs(x - 1.0) < fabs(fmin(x, 1.0)) * precision);
}

__global__ __launch_bounds__(kBlock) void minPathCoverKernel(
    const uint32_t num_problems, const uint32_t * __restrict__ prob_cluster, const uint64_t * __restrict__ cluster_row_off,
    const uint64_t * __restrict__ cluster_path_off, const uint64_t * __restrict__ row_ent_off,
    const uint32_t * __restrict__ ent_path, const double * __restrict__ ent_prob, const double * __restrict__ row_count,
    const double * __restrict__ row_noise, const uint64_t * __restrict__ out_off, uint8_t * __restrict__ covered,
    uint32_t * __restrict__ cover_out, uint32_t * __restrict__ cover_size) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ double red_val[kBlock / 64];
    __shared__ uint32_t red_idx[kBlock / 64];
    __shared__ uint32_t best_shared;
    const uint32_t p = blockIdx.x;
    if (p >= num_problems) return;
    const uint32_t k = prob_cluster[p];
    const uint64_t r0 = cluster_row_off[k], r1 = cluster_row_off[k + 1];
    const uint32_t N = static_cast<uint32_t>(cluster_path_off[k + 1] - cluster_path_off[k]);
    double * weights = reinterpret_cast<double *>(smem_raw);  // [N]
    double * cov = weights + N;                                // [N]
    uint32_t * out = cover_out + out_off[p];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

    if (N == 1) {  // src/path_abundance_estimator.cpp:302-305
        if (threadIdx.x == 0) {
            out[0] = 0;
            cover_size[p] = 1;
        }
        return;
    }

    for (uint32_t j = threadIdx.x; j < N; j += kBlock) weights[j] = 0.0;
    __syncthreads();
    // path weights (:240-257); rows whose noise probability is 1 carry no reads for the cover
    for (uint64_t r = r0 + threadIdx.x; r < r1; r += kBlock) {
        const double c = isOne(row_noise[r]) ? 0.0 : row_count[r];
        covered[r] = (c > 0.0) ? 0 : 1;
        for (uint64_t e = row_ent_off[r]; e < row_ent_off[r + 1]; ++e) atomicAdd(&weights[ent_path[e]], log(ent_prob[e]) * c);
    }
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < N; j += kBlock) weights[j] *= -1.0;

    uint32_t n_cover = 0;
    while (true) {
        for (uint32_t j = threadIdx.x; j < N; j += kBlock) cov[j] = 0.0;
        __syncthreads();
        for (uint64_t r = r0 + threadIdx.x; r < r1; r += kBlock) {
            if (!covered[r]) {
                const double c = row_count[r];
                for (uint64_t e = row_ent_off[r]; e < row_ent_off[r + 1]; ++e) atomicAdd(&cov[ent_path[e]], c);
            }
        }
        __syncthreads();
        // first index with the largest positive covered / weight
        double best_val = 0.0;
        uint32_t best_idx = 0xFFFFFFFFu;
        for (uint32_t j = threadIdx.x; j < N; j += kBlock) {
            const double v = cov[j] / weights[j];
            if (v > best_val) {
                best_val = v;
                best_idx = j;
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const double ov = __shfl_xor(best_val, d, 64);
            const uint32_t oi = __shfl_xor(best_idx, d, 64);
            if (ov > best_val || (ov == best_val && oi < best_idx)) {
                best_val = ov;
                best_idx = oi;
            }
        }
        if (lane == 0) {
            red_val[wave] = best_val;
            red_idx[wave] = best_idx;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            double bv = red_val[0];
            uint32_t bi = red_idx[0];
            for (int w = 1; w < kBlock / 64; ++w) {
                if (red_val[w] > bv || (red_val[w] == bv && red_idx[w] < bi)) {
                    bv = red_val[w];
                    bi = red_idx[w];
                }
            }
            best_shared = (bv > 0.0) ? bi : 0xFFFFFFFFu;
            if (bv > 0.0) out[n_cover] = bi;
        }
        __syncthreads();
        const uint32_t best = best_shared;
        if (best == 0xFFFFFFFFu) break;  // nothing left to cover
        ++n_cover;
        for (uint64_t r = r0 + threadIdx.x; r < r1; r += kBlock) {
            if (!covered[r]) {
                bool hit = false;
                for (uint64_t e = row_ent_off[r]; e < row_ent_off[r + 1]; ++e) hit = hit || (ent_path[e] == best);
                if (hit) covered[r] = 1;
            }
        }
        __syncthreads();
    }
    // ascending order (:337); one thread, covers are short
    if (threadIdx.x == 0) {
        for (uint32_t i = 1; i < n_cover; ++i) {
            const uint32_t v = out[i];
            uint32_t j = i;
            while (j > 0 && out[j - 1] > v) {
                out[j] = out[j - 1];
                --j;
            }
            out[j] = v;
        }
        cover_size[p] = n_cover;
    }
}

}  // namespace

extern "C" int rpvg_hip_min_path_cover(rpvg_hip_ctx * ctx, const rpvg_hip_batch * batch, uint32_t num_clusters,
                                       const uint32_t * clusters, const uint64_t * cover_off, uint32_t * cover,
                                       uint32_t * cover_size) {
    RPVG_REQUIRE(ctx && batch, "rpvg_hip_min_path_cover: NULL argument");
    if (num_clusters == 0) return RPVG_HIP_OK;
    RPVG_REQUIRE(clusters && cover_off && cover && cover_size, "rpvg_hip_min_path_cover: NULL argument");
    uint32_t max_paths = 0;
    for (uint32_t i = 0; i < num_clusters; ++i) {
        const uint32_t k = clusters[i];
        RPVG_REQUIRE(k < batch->num_clusters, "rpvg_hip_min_path_cover: cluster %u of %u", k, batch->num_clusters);
        const uint64_t N = batch->h_cluster_path_off[k + 1] - batch->h_cluster_path_off[k];
        RPVG_REQUIRE(batch->h_cluster_row_off[k + 1] > batch->h_cluster_row_off[k] && N > 0, "rpvg_hip_min_path_cover: cluster %u is empty", k);
        RPVG_REQUIRE(cover_off[i + 1] - cover_off[i] >= N, "rpvg_hip_min_path_cover: output range of cluster %u is smaller than its %llu paths", k,
                     static_cast<unsigned long long>(N));
        max_paths = std::max<uint32_t>(max_paths, static_cast<uint32_t>(N));
    }
    const size_t lds = (static_cast<size_t>(max_paths) * 16 + 15) & ~static_cast<size_t>(15);
    RPVG_REQUIRE(lds <= 150 * 1024, "rpvg_hip_min_path_cover: a cluster with %u paths does not fit the LDS-resident weight vectors", max_paths);

    std::lock_guard<std::mutex> lock(ctx->mutex);
    RPVG_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    DeviceBuffer<uint32_t> d_clusters, d_cover, d_size;
    DeviceBuffer<uint64_t> d_off;
    DeviceBuffer<uint8_t> d_covered;
    RPVG_HIP_CHECK(d_clusters.upload(clusters, num_clusters, st));
    RPVG_HIP_CHECK(d_off.upload(cover_off, num_clusters + 1, st));
    RPVG_HIP_CHECK(d_cover.alloc(cover_off[num_clusters]));
    RPVG_HIP_CHECK(d_size.alloc(num_clusters));
    RPVG_HIP_CHECK(d_covered.alloc(batch->num_rows));
    if (lds > 64 * 1024) {
        RPVG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&minPathCoverKernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    }
    const int span = ctx->spanBegin(FAM_BUILD);
    minPathCoverKernel<<<dim3(num_clusters), dim3(kBlock), lds, st>>>(
        num_clusters, d_clusters.ptr, batch->cluster_row_off.ptr, batch->cluster_path_off.ptr, batch->row_ent_off.ptr,
        batch->ent_path.ptr, batch->ent_prob.ptr, batch->row_count.ptr, batch->row_noise.ptr, d_off.ptr, d_covered.ptr,
        d_cover.ptr, d_size.ptr);
    ctx->spanEnd(span);
    ctx->stats.build_launches += 1;
    RPVG_HIP_CHECK(hipGetLastError());
    RPVG_HIP_CHECK(d_cover.download(cover, st));
    RPVG_HIP_CHECK(d_size.download(cover_size, st));
    RPVG_HIP_CHECK(hipStreamSynchronize(st));
    return RPVG_HIP_OK;
}
