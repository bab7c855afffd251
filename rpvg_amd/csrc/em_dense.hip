// EM on one large dense cluster matrix streamed from HBM (gfx950).
//
// Takes over EMAbundanceEstimator (src/path_abundance_estimator.cpp:47-114)
// for a matrix that is far larger than the caches (1M x 2001 doubles = 16 GB):
// the reference makes ~5 sweeps over R x C per iteration (temp = P.*a, row
// sums, divide, GEMV); here one iteration is ONE read of the matrix.
//
// Layout (chosen by this engine, the reference is column-major): row-major,
// `ld` doubles per row (even), so that a wave reads a row with 16-byte loads,
// lane l owning columns {2l, 2l+1} + 128*m.  Per row the wave computes
//   s_i = sum_j P_ij a_j   (per-lane partial + 6-step wave shuffle reduction)
//   w_i = c_i / s_i
//   t_j += w_i P_ij        (per-lane register accumulators, no atomics)
// and per iteration three launches run on the stream:
//   emDenseAccumKernel<NCHUNK>  grid-wide streaming pass, one partial t[] per block
//   emDenseFinalizeKernel       a'_j = a_j * sum_blocks t_j / T, per-column
//                               convergence test, OR-ed into a device flag
//   emDenseControlKernel        the reference's stop rule (10 consecutive
//                               converged iterations) on the device; sets `done`
// Once `done` is set the remaining queued launches exit immediately, so the
// host can queue iterations in chunks without a sync per iteration and the
// loop still stops at exactly the reference's iteration.
//
// Deterministic: fixed row->wave assignment, fixed reduction orders.

#include "common.hpp"

#include <cmath>
#include <cstring>

#include <algorithm>
#include <cmath>
#include <cstdlib>

using namespace rpvg_hip_detail;

namespace {

constexpr double kMinEmAbundance = 1e-8;  // src/path_abundance_estimator.cpp:11
constexpr uint32_t kMinEmConvIts = 10;    // src/path_abundance_estimator.cpp:10
constexpr int kAccumBlock = 256;          // 4 waves

typedef EmGridControl DenseControl;  // common.hpp: the stop rule's words on the device

typedef double dvec2 __attribute__((ext_vector_type(2)));

// streaming 16-byte load (no reuse: keep it out of the way of the cached vectors)
__device__ __forceinline__ double2 loadStream(const double * p) {
    const dvec2 x = __builtin_nontemporal_load(reinterpret_cast<const dvec2 *>(p));
    return make_double2(x.x, x.y);
}

__device__ __forceinline__ double waveReduceSumD(double v) { return waveSumF64(v); }

// One streaming pass.  NCHUNK = ceil(C / 128): the wave holds a whole row in
// registers (2*NCHUNK doubles per lane).
template <int NCHUNK>
__global__ __launch_bounds__(kAccumBlock) void emDenseAccumKernel(
    const double * __restrict__ P, const uint64_t R, const uint32_t C, const uint64_t ld,
    const double * __restrict__ counts, const double * __restrict__ a_global, double * __restrict__ partials,
    const uint32_t partial_ld, const DenseControl * __restrict__ ctl) {
    if (ctl->done) return;
    __shared__ double t_lds[kAccumBlock / 64][NCHUNK * 128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t waves_total = gridDim.x * (kAccumBlock / 64);
    const uint32_t wave_global = blockIdx.x * (kAccumBlock / 64) + wave;

    double a[NCHUNK][2], t[NCHUNK][2];
#pragma unroll
    for (int m = 0; m < NCHUNK; ++m) {
        const uint32_t c0 = 2 * lane + 128 * m;
        a[m][0] = (c0 < C) ? a_global[c0] : 0.0;
        a[m][1] = (c0 + 1 < C) ? a_global[c0 + 1] : 0.0;
        t[m][0] = 0.0;
        t[m][1] = 0.0;
    }

    for (uint64_t r = wave_global; r < R; r += waves_total) {
        const double2 * row = reinterpret_cast<const double2 *>(P + r * ld);
        double2 v[NCHUNK];
#pragma unroll
        for (int m = 0; m < NCHUNK; ++m) {
            const uint32_t c0 = 2 * lane + 128 * m;
            if (c0 + 1 < ld) {
                v[m] = row[lane + 64 * m];
            } else {
                v[m].x = (c0 < ld) ? P[r * ld + c0] : 0.0;
                v[m].y = 0.0;
            }
            if (c0 >= C) v[m].x = 0.0;
            if (c0 + 1 >= C) v[m].y = 0.0;
        }
        const double cnt = counts[r];
        double s = 0.0;
#pragma unroll
        for (int m = 0; m < NCHUNK; ++m) {
            s = fma(v[m].x, a[m][0], s);
            s = fma(v[m].y, a[m][1], s);
        }
        s = waveReduceSumD(s);
        const double w = cnt / s;
#pragma unroll
        for (int m = 0; m < NCHUNK; ++m) {
            t[m][0] = fma(w, v[m].x, t[m][0]);
            t[m][1] = fma(w, v[m].y, t[m][1]);
        }
    }

    // combine the block's 4 waves in a fixed order, one partial vector per block
#pragma unroll
    for (int m = 0; m < NCHUNK; ++m) {
        t_lds[wave][2 * lane + 128 * m] = t[m][0];
        t_lds[wave][2 * lane + 128 * m + 1] = t[m][1];
    }
    __syncthreads();
    double * out = partials + static_cast<uint64_t>(blockIdx.x) * partial_ld;
    for (uint32_t j = threadIdx.x; j < C; j += kAccumBlock) {
        double acc = t_lds[0][j];
#pragma unroll
        for (int w = 1; w < kAccumBlock / 64; ++w) acc += t_lds[w][j];
        out[j] = acc;
    }
}


// ---- wide matrices: one row split across the 4 waves of a block ------------------
//
// For C > 256 a whole row per wave needs too many registers (223 VGPRs at C = 2001:
// two waves per SIMD, not enough loads in flight to cover HBM latency).  Here wave w of a
// block owns columns [512w, 512w + 512) (NCHUNK <= 4 chunks of 128 columns), the block
// walks its rows two at a time: every wave loads its quarter of both rows, reduces its
// partial s_i, the four partials meet in LDS behind ONE barrier, and every wave updates
// the t_j of its own columns — so the accumulators never need a cross-wave reduction.
template <int NCHUNK>
__global__ __launch_bounds__(256) void emDenseAccumWideKernel(
    const double * __restrict__ P, const uint64_t R, const uint32_t C, const uint64_t ld,
    const double * __restrict__ counts, const double * __restrict__ a_global, double * __restrict__ partials,
    const uint32_t partial_ld, const DenseControl * __restrict__ ctl) {
    if (ctl->done) return;
    constexpr int ROWS = 2;
    __shared__ double s_part[2][ROWS][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t col_base = wave * (NCHUNK * 128) + 2 * lane;

    double a[NCHUNK][2], t[NCHUNK][2];
#pragma unroll
    for (int m = 0; m < NCHUNK; ++m) {
        const uint32_t c0 = col_base + 128 * m;
        a[m][0] = (c0 < C) ? a_global[c0] : 0.0;
        a[m][1] = (c0 + 1 < C) ? a_global[c0 + 1] : 0.0;
        t[m][0] = 0.0;
        t[m][1] = 0.0;
    }

    const uint64_t row_stride = static_cast<uint64_t>(gridDim.x) * ROWS;
    int parity = 0;
    for (uint64_t r0 = static_cast<uint64_t>(blockIdx.x) * ROWS; r0 < R; r0 += row_stride, parity ^= 1) {
        double2 v[ROWS][NCHUNK];
#pragma unroll
        for (int q = 0; q < ROWS; ++q) {
            const uint64_t r = r0 + q;
            const double * row = P + (r < R ? r : r0) * ld;
#pragma unroll
            for (int m = 0; m < NCHUNK; ++m) {
                const uint32_t c0 = col_base + 128 * m;
                if (c0 + 1 < ld) {
                    v[q][m] = loadStream(row + c0);
                } else {
                    v[q][m].x = (c0 < ld) ? P[(r < R ? r : r0) * ld + c0] : 0.0;
                    v[q][m].y = 0.0;
                }
                if (c0 >= C || r >= R) v[q][m].x = 0.0;
                if (c0 + 1 >= C || r >= R) v[q][m].y = 0.0;
            }
        }
#pragma unroll
        for (int q = 0; q < ROWS; ++q) {
            double s = 0.0;
#pragma unroll
            for (int m = 0; m < NCHUNK; ++m) {
                s = fma(v[q][m].x, a[m][0], s);
                s = fma(v[q][m].y, a[m][1], s);
            }
            s = waveReduceSumD(s);
            if (lane == 0) s_part[parity][q][wave] = s;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < ROWS; ++q) {
            const uint64_t r = r0 + q;
            if (r < R) {
                const double s = ((s_part[parity][q][0] + s_part[parity][q][1]) + s_part[parity][q][2]) + s_part[parity][q][3];
                const double w = counts[r] / s;
#pragma unroll
                for (int m = 0; m < NCHUNK; ++m) {
                    t[m][0] = fma(w, v[q][m].x, t[m][0]);
                    t[m][1] = fma(w, v[q][m].y, t[m][1]);
                }
            }
        }
    }

    double * out = partials + static_cast<uint64_t>(blockIdx.x) * partial_ld;
#pragma unroll
    for (int m = 0; m < NCHUNK; ++m) {
        const uint32_t c0 = col_base + 128 * m;
        if (c0 + 1 < partial_ld) {
            *reinterpret_cast<double2 *>(out + c0) = make_double2(t[m][0], t[m][1]);
        } else if (c0 < partial_ld) {
            out[c0] = t[m][0];
        }
    }
}

// first finalize stage for many partial vectors: slice y of the partials -> one vector per slice
// status: row-sharded runs — the word behind the C column sums that travels through the all-reduce with them: a rank
// whose sums are not finite raises it, and every rank stops at the same iteration instead of one of them failing alone
// and the others waiting in the next collective.
__global__ void emDenseReducePartialsKernel(const uint32_t C, const uint32_t num_partials, const uint32_t partial_ld,
                                            const double * __restrict__ partials, double * __restrict__ reduced,
                                            const DenseControl * __restrict__ ctl, double * __restrict__ status = nullptr) {
    if (ctl->done) return;
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= C) return;
    const uint32_t slices = gridDim.y;
    const uint32_t b0 = static_cast<uint32_t>(static_cast<uint64_t>(num_partials) * blockIdx.y / slices);
    const uint32_t b1 = static_cast<uint32_t>(static_cast<uint64_t>(num_partials) * (blockIdx.y + 1) / slices);
    double acc = 0.0;
    for (uint32_t b = b0; b < b1; ++b) acc += partials[static_cast<uint64_t>(b) * partial_ld + j];
    reduced[static_cast<uint64_t>(blockIdx.y) * partial_ld + j] = acc;
    if (status && !isfinite(acc)) *status = 1.0;
}

__global__ void emDenseFinalizeKernel(const uint32_t C, const uint32_t num_partials, const uint32_t partial_ld,
                                      const double * __restrict__ partials, double * __restrict__ a_global,
                                      const double total_count, const double max_rel_em_conv, DenseControl * ctl,
                                      const double * __restrict__ status = nullptr, const double zero_mass = 0.0) {
    if (ctl->done) return;
    if (status && *status != 0.0) {  // (summed over the ranks: somebody's column sums were not finite)
        if (blockIdx.x == 0 && threadIdx.x == 0) ctl->error = 1;
        return;
    }
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    int viol = 0;
    if (j < C) {
        double tj = 0.0;
        for (uint32_t b = 0; b < num_partials; ++b) tj += partials[static_cast<uint64_t>(b) * partial_ld + j];
        const double aj = a_global[j];
        // (zero_mass: the rows without a selected path, which the matrix of an rpvg_hip_em_solve problem does not hold, put
        // their whole read count on the noise component — em_sparse.hip; 0 for a matrix that holds every row)
        const double an = (j + 1 == C ? aj * tj + zero_mass : aj * tj) / total_count;
        if (an >= kMinEmAbundance && fabs(an - aj) / an > max_rel_em_conv) viol = 1;
        a_global[j] = an;
    }
    if (__syncthreads_or(viol) && threadIdx.x == 0) atomicOr(&ctl->viol, 1u);
}

__global__ void emDenseControlKernel(DenseControl * ctl, const uint32_t max_em_its) {
    if (ctl->done) return;
    if (ctl->error) {
        ctl->done = 1;
        return;
    }
    ctl->iterations += 1;
    if (ctl->viol == 0) {
        ctl->conv_its += 1;
        if (ctl->conv_its == kMinEmConvIts) ctl->done = 1;
    } else {
        ctl->conv_its = 0;
    }
    ctl->viol = 0;
    if (ctl->iterations >= max_em_its) ctl->done = 1;
}

__global__ void fillConstantKernel(double * x, const uint32_t n, const double v) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = v;
}

template <int NCHUNK>
void launchAccum(uint32_t grid, hipStream_t st, const double * P, uint64_t R, uint32_t C, uint64_t ld,
                 const double * counts, const double * a, double * partials, uint32_t partial_ld,
                 const DenseControl * ctl) {
    emDenseAccumKernel<NCHUNK><<<dim3(grid), dim3(kAccumBlock), 0, st>>>(P, R, C, ld, counts, a, partials, partial_ld, ctl);
}

// ---- dense builder from the sparse rows of one cluster -----------------------

__global__ void denseFromClusterKernel(const uint64_t r0, const uint64_t num_rows, const uint32_t num_paths,
                                       const uint64_t * __restrict__ row_ent_off, const uint32_t * __restrict__ ent_path,
                                       const double * __restrict__ ent_prob, const double * __restrict__ row_count,
                                       const double * __restrict__ row_noise, double * __restrict__ P, const uint64_t ld,
                                       double * __restrict__ counts, double * __restrict__ total) {
    const uint64_t i = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    double c = 0.0;
    if (i < num_rows) {
        const uint64_t r = r0 + i;
        const uint64_t e0 = row_ent_off[r], e1 = row_ent_off[r + 1];
        double rowsum = 0.0;
        for (uint64_t e = e0; e < e1; ++e) rowsum += ent_prob[e];
        const double nz = row_noise[r];
        double * out = P + i * ld;
        for (uint64_t e = e0; e < e1; ++e) out[ent_path[e]] = (ent_prob[e] / rowsum) * (1 - nz);
        out[num_paths] = nz;
        c = row_count[r];
        counts[i] = c;
    }
    // read counts are integers: the sum is exact in any order
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
    if ((threadIdx.x & 63) == 0 && c != 0.0) atomicAdd(total, c);
}

// ---- synthetic dense cluster -------------------------------

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__device__ __forceinline__ double u01(uint64_t h) { return (h >> 11) * (1.0 / 9007199254740992.0); }

struct SynthTables {
    const double * theta_cdf;   // [N] inclusive CDF of the true-path distribution
    const double * inv_len;     // [N] 1 / effective length
    const double * score_prob;  // [21] exp(-score_log_base * d)
    const double * deficit_cdf; // [20] CDF of 1 + Poisson(3) capped at 20 (index d-1)
};

// one wave per row
__global__ __launch_bounds__(256) void synthDenseKernel(const uint64_t seed, const uint64_t row_begin, const uint64_t R, const uint32_t N,
                                                        const SynthTables tab, double * __restrict__ P, const uint64_t ld,
                                                        double * __restrict__ counts) {
    const int lane = threadIdx.x & 63;
    const uint64_t r = (blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x) >> 6;
    if (r >= R) return;
    const uint64_t row_key = mix64(seed ^ ((row_begin + r) * 0xD1B54A32D192ED03ull));
    // true path: inverse CDF by binary search (uniform across the wave)
    const double ut = u01(mix64(row_key ^ 0x1ull));
    uint32_t lo = 0, hi = N - 1;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (tab.theta_cdf[mid] < ut) lo = mid + 1; else hi = mid;
    }
    const uint32_t true_path = lo;
    // mapq in {60: 70 %, 30: 15 %, 10: 10 %, 3: 5 %} -> noise = max(1e-4, 10^(-mapq/10))
    const double um = u01(mix64(row_key ^ 0x2ull));
    const double noise = (um < 0.70) ? 1e-4 : (um < 0.85) ? 1e-3 : (um < 0.95) ? 0.1 : 0.50118723362727224;

    auto raw = [&](uint32_t j) -> double {
        uint32_t d = 0;
        if (j != true_path) {
            const double ud = u01(mix64(row_key ^ (0x100ull + j)));
            d = 1;
            while (d < 20 && tab.deficit_cdf[d - 1] < ud) ++d;
        }
        return tab.score_prob[d] * tab.inv_len[j];
    };
    double partial = 0.0;
    for (uint32_t j = lane; j < N; j += 64) partial += raw(j);
    const double rowsum = waveReduceSumD(partial);
    double * out = P + r * ld;
    for (uint32_t j = lane; j < N; j += 64) out[j] = (raw(j) / rowsum) * (1 - noise);
    for (uint64_t j = N + 1 + lane; j < ld; j += 64) out[j] = 0.0;
    if (lane == 0) {
        out[N] = noise;
        counts[r] = 1.0;
    }
}

// The same cluster as the rows of a cluster batch (rpvg_hip_batch: CSR of (path, probability) entries, every row holding
// all N paths in path order): what the estimator classes take.  One wave per row.
__global__ __launch_bounds__(256) void synthDenseBatchKernel(const uint64_t seed, const uint64_t row_begin, const uint64_t R, const uint32_t N,
                                                             const SynthTables tab, uint64_t * __restrict__ row_ent_off, uint32_t * __restrict__ ent_path,
                                                             double * __restrict__ ent_prob, double * __restrict__ row_count, double * __restrict__ row_noise) {
    const int lane = threadIdx.x & 63;
    const uint64_t r = (blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x) >> 6;
    if (r >= R) return;
    const uint64_t row_key = mix64(seed ^ ((row_begin + r) * 0xD1B54A32D192ED03ull));
    const double ut = u01(mix64(row_key ^ 0x1ull));
    uint32_t lo = 0, hi = N - 1;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (tab.theta_cdf[mid] < ut) lo = mid + 1; else hi = mid;
    }
    const uint32_t true_path = lo;
    const double um = u01(mix64(row_key ^ 0x2ull));
    const double noise = (um < 0.70) ? 1e-4 : (um < 0.85) ? 1e-3 : (um < 0.95) ? 0.1 : 0.50118723362727224;
    auto raw = [&](uint32_t j) -> double {
        uint32_t d = 0;
        if (j != true_path) {
            const double ud = u01(mix64(row_key ^ (0x100ull + j)));
            d = 1;
            while (d < 20 && tab.deficit_cdf[d - 1] < ud) ++d;
        }
        return tab.score_prob[d] * tab.inv_len[j];
    };
    double partial = 0.0;
    for (uint32_t j = lane; j < N; j += 64) partial += raw(j);
    const double rowsum = waveReduceSumD(partial);
    const uint64_t e0 = r * N;
    for (uint32_t j = lane; j < N; j += 64) {
        ent_path[e0 + j] = j;
        ent_prob[e0 + j] = (raw(j) / rowsum) * (1 - noise);
    }
    if (lane == 0) {
        row_ent_off[r] = e0;
        if (r + 1 == R) row_ent_off[R] = R * N;
        row_count[r] = 1.0;
        row_noise[r] = noise;
    }
}

}  // namespace

namespace {

// Shared body of rpvg_hip_em_dense and rpvg_hip_em_dense_sharded.  `sharded`: the rows held here are one
// rank's share of the cluster; the partial column sums are summed over the ranks of the context's
// communicator before the update.
int emDenseRun(rpvg_hip_ctx * ctx, const char * who, const bool sharded, const double * device_matrix, uint64_t num_rows,
               uint32_t num_cols, uint64_t ld, const double * device_counts, double total_count, uint32_t max_em_its,
               double max_rel_em_conv, double * abundances, double * noise_count, uint32_t * iterations) {
    // A row-sharded run is a collective: a rank that fails its own checks may not simply leave — its peers would wait for it
    // in the first all-reduce.  The ranks therefore exchange a status word first (the sum over ranks of "my checks failed"),
    // and all of them return an error if anybody's did.
    if (sharded && ctx && ctx->comm) {
        auto local_checks = [&]() -> int {
            RPVG_REQUIRE(abundances && noise_count && iterations && ((device_matrix && device_counts) || num_rows == 0), "%s: NULL argument", who);
            RPVG_REQUIRE(num_cols >= 2, "%s: need at least one path and the noise column", who);
            RPVG_REQUIRE(ld >= num_cols && (ld % 2) == 0, "%s: ld (%llu) must be even and >= num_cols (%u)", who, static_cast<unsigned long long>(ld), num_cols);
            RPVG_REQUIRE((reinterpret_cast<uintptr_t>(device_matrix) % 16) == 0, "%s: matrix must be 16-byte aligned", who);
            RPVG_REQUIRE(num_cols <= 2048, "%s: %u columns exceed the register-resident row limit (2048)", who, num_cols);
            RPVG_REQUIRE(total_count > 0 && max_em_its > 0, "%s: total_count and max_em_its must be positive", who);
            const char * inject = std::getenv("RPVG_HIP_INJECT_SHARD_FAILURE");  // test hook: "checks" fails this rank's checks
            RPVG_REQUIRE(!(inject && std::strcmp(inject, "checks") == 0), "%s: injected failure of the local checks (RPVG_HIP_INJECT_SHARD_FAILURE)", who);
            return RPVG_HIP_OK;
        };
        const int local_rc = local_checks();
        double peers_failed = 0;
        {
            std::lock_guard<std::mutex> lock(ctx->mutex);
            RPVG_HIP_CHECK(hipSetDevice(ctx->device));
            DeviceBuffer<double> d_status;
            const double mine = local_rc == RPVG_HIP_OK ? 0.0 : 1.0;
            RPVG_HIP_CHECK(d_status.upload(&mine, 1, ctx->stream));
            if (const int rc = ctx->allReduceSumF64(d_status.ptr, 1)) return rc;
            RPVG_HIP_CHECK(hipMemcpyAsync(&peers_failed, d_status.ptr, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
            RPVG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        }
        if (local_rc != RPVG_HIP_OK) return local_rc;
        if (peers_failed != 0) {
            setError("%s: %g rank(s) of the communicator failed their local checks; nobody entered the EM", who, peers_failed);
            return RPVG_HIP_ERR_RUNTIME;
        }
    }
    RPVG_REQUIRE(ctx && abundances && noise_count && iterations && ((device_matrix && device_counts) || (sharded && num_rows == 0)),
                 "%s: NULL argument", who);
    // (a rank of a row-sharded cluster may hold no rows — fewer rows than ranks —: it contributes zero column sums and
    // takes part in every all-reduce, so that its peers do not wait for it)
    RPVG_REQUIRE((num_rows > 0 || sharded) && num_cols >= 2, "%s: need at least one row, one path and the noise column", who);
    RPVG_REQUIRE(ld >= num_cols && (ld % 2) == 0, "%s: ld (%llu) must be even and >= num_cols (%u)", who,
                 static_cast<unsigned long long>(ld), num_cols);
    RPVG_REQUIRE((reinterpret_cast<uintptr_t>(device_matrix) % 16) == 0, "%s: matrix must be 16-byte aligned", who);
    RPVG_REQUIRE(num_cols <= 2048, "%s: %u columns exceed the register-resident row limit (2048)", who, num_cols);
    RPVG_REQUIRE(total_count > 0 && max_em_its > 0, "%s: total_count and max_em_its must be positive", who);
    RPVG_REQUIRE(!sharded || ctx->comm, "%s: the context has no communicator (rpvg_hip_comm_init first)", who);

    std::lock_guard<std::mutex> lock(ctx->mutex);
    RPVG_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const uint32_t C = num_cols;
    DenseEmRun run;
    run.matrix = device_matrix;
    run.num_rows = num_rows;
    run.num_cols = num_cols;
    run.ld = ld;
    run.counts = device_counts;
    run.total_count = total_count;
    run.max_em_its = max_em_its;
    run.max_rel_em_conv = max_rel_em_conv;
    run.sharded = sharded;
    if (const int rc = emDenseIterate(ctx, who, run)) return rc;

    std::vector<double> a(C);
    RPVG_HIP_CHECK(hipMemcpyAsync(a.data(), run.d_a.ptr, sizeof(double) * C, hipMemcpyDeviceToHost, st));
    RPVG_HIP_CHECK(hipStreamSynchronize(st));

    // src/path_abundance_estimator.cpp:100-113
    double nc = 0;
    for (uint32_t j = 0; j + 1 < C; ++j) {
        if (a[j] < kMinEmAbundance) {
            nc += a[j] * total_count;
            abundances[j] = 0;
        } else {
            abundances[j] = a[j] * total_count;
        }
    }
    nc += a[C - 1] * total_count;
    *noise_count = nc;
    *iterations = run.control.iterations;
    return RPVG_HIP_OK;
}

}  // namespace

namespace rpvg_hip_detail {

// The iterations of the dense EM up to the stop rule (the body of rpvg_hip_em_dense[_sharded]; also the dense route of the
// large problems of rpvg_hip_em_solve, em_grid.hip).  Caller holds ctx->mutex, has set the device and checked the shape.
int emDenseIterate(rpvg_hip_ctx * ctx, const char * who, DenseEmRun & run) {
    hipStream_t st = ctx->stream;
    const uint32_t C = run.num_cols;
    const uint64_t num_rows = run.num_rows, ld = run.ld;
    const double * device_matrix = run.matrix;
    const double * device_counts = run.counts;
    const double total_count = run.total_count, max_rel_em_conv = run.max_rel_em_conv, zero_mass = run.zero_mass;
    const uint32_t max_em_its = run.max_em_its;
    const bool sharded = run.sharded;
    DeviceBuffer<double> & d_a = run.d_a;

    const uint32_t cus = ctx->props.multiProcessorCount;
    const bool wide = C > 256;  // row split over the block's waves (emDenseAccumWideKernel)
    // enough waves to cover HBM latency, few enough partial vectors to reduce cheaply
    uint32_t blocks_per_cu = wide ? 4 : 2;
    if (const char * env = RPVG_EXPERIMENT_ENV("RPVG_HIP_DENSE_BLOCKS_PER_CU")) blocks_per_cu = std::max(1, std::atoi(env));
    uint32_t grid = wide ? std::min<uint64_t>((num_rows + 1) / 2, static_cast<uint64_t>(cus) * blocks_per_cu)
                         : std::min<uint64_t>((num_rows + 3) / 4, static_cast<uint64_t>(cus) * blocks_per_cu);
    grid = std::max<uint32_t>(grid, 1);
    const uint32_t partial_ld = wide ? ((C + 511) / 512) * 512 : ((C + 1) & ~1u);
    const uint32_t reduce_slices = 16;

    DeviceBuffer<double> d_partials, d_reduced, d_t;
    if (wide) RPVG_HIP_CHECK(d_reduced.alloc(static_cast<size_t>(reduce_slices) * partial_ld));
    if (sharded) {
        RPVG_HIP_CHECK(d_t.alloc(partial_ld + 2));  // [C column sums | ... | status word at C]
        RPVG_HIP_CHECK(hipMemsetAsync(d_t.ptr, 0, (partial_ld + 2) * sizeof(double), st));
    }
    const dim3 col_grid((C + 255) / 256);
    DeviceBuffer<DenseControl> d_ctl;
    RPVG_HIP_CHECK(d_a.alloc(C));
    RPVG_HIP_CHECK(d_partials.alloc(static_cast<size_t>(grid) * partial_ld));
    RPVG_HIP_CHECK(d_ctl.alloc(1));
    RPVG_HIP_CHECK(hipMemsetAsync(d_ctl.ptr, 0, sizeof(DenseControl), st));
    // src/path_abundance_estimator.cpp:54 — 1 / float(C), widened
    double a0 = static_cast<double>(1.0f / static_cast<float>(C));
    if (sharded) {  // test hook: "nan" poisons this rank's start vector — its column sums are not finite in the first iteration
        const char * inject = std::getenv("RPVG_HIP_INJECT_SHARD_FAILURE");
        if (inject && std::strcmp(inject, "nan") == 0) a0 = std::nan("");
    }
    fillConstantKernel<<<dim3((C + 255) / 256), dim3(256), 0, st>>>(d_a.ptr, C, a0);

    const int nchunk = (C + 127) / 128;
    const uint32_t chunk_its = 8;  // iterations queued between looks at the control word
    DenseControl & h_ctl = run.control;
    h_ctl = DenseControl{};
    uint32_t queued = 0;
    // em_dense_ms (rpvg_hip_kernel_stats) = HIP-event time of the streaming-pass launches only
    while (!h_ctl.done) {
        const uint32_t n = std::min<uint32_t>(chunk_its, max_em_its - queued);
        for (uint32_t i = 0; i < n; ++i) {
            const int span = ctx->spanBegin(FAM_EM_DENSE);
            if (wide) {
                const int wchunk = (C + 511) / 512;
                if (wchunk <= 1) emDenseAccumWideKernel<1><<<dim3(grid), dim3(256), 0, st>>>(device_matrix, num_rows, C, ld, device_counts, d_a.ptr, d_partials.ptr, partial_ld, d_ctl.ptr);
                else if (wchunk <= 2) emDenseAccumWideKernel<2><<<dim3(grid), dim3(256), 0, st>>>(device_matrix, num_rows, C, ld, device_counts, d_a.ptr, d_partials.ptr, partial_ld, d_ctl.ptr);
                else if (wchunk <= 3) emDenseAccumWideKernel<3><<<dim3(grid), dim3(256), 0, st>>>(device_matrix, num_rows, C, ld, device_counts, d_a.ptr, d_partials.ptr, partial_ld, d_ctl.ptr);
                else emDenseAccumWideKernel<4><<<dim3(grid), dim3(256), 0, st>>>(device_matrix, num_rows, C, ld, device_counts, d_a.ptr, d_partials.ptr, partial_ld, d_ctl.ptr);
                ctx->spanEnd(span);
                emDenseReducePartialsKernel<<<dim3((C + 255) / 256, reduce_slices), dim3(256), 0, st>>>(C, grid, partial_ld, d_partials.ptr, d_reduced.ptr, d_ctl.ptr);
                if (sharded) {
                    // this rank's column sums -> sum over ranks (same bits on every rank) -> identical update everywhere
                    emDenseReducePartialsKernel<<<col_grid, dim3(256), 0, st>>>(C, reduce_slices, partial_ld, d_reduced.ptr, d_t.ptr, d_ctl.ptr, d_t.ptr + C);
                    if (const int rc = ctx->allReduceSumF64(d_t.ptr, C + 1)) {
                        (void) hipStreamSynchronize(st);  // the kernels queued so far use the buffers freed on return
                        return rc;
                    }
                    emDenseFinalizeKernel<<<col_grid, dim3(256), 0, st>>>(C, 1, partial_ld, d_t.ptr, d_a.ptr, total_count,
                                                                      max_rel_em_conv, d_ctl.ptr, d_t.ptr + C, zero_mass);
                } else {
                    emDenseFinalizeKernel<<<col_grid, dim3(256), 0, st>>>(C, reduce_slices, partial_ld, d_reduced.ptr, d_a.ptr,
                                                                      total_count, max_rel_em_conv, d_ctl.ptr, nullptr, zero_mass);
                }
                emDenseControlKernel<<<dim3(1), dim3(1), 0, st>>>(d_ctl.ptr, max_em_its);
                continue;
            }
            if (nchunk <= 1) launchAccum<1>(grid, st, device_matrix, num_rows, C, ld, device_counts, d_a.ptr, d_partials.ptr, partial_ld, d_ctl.ptr);
            else if (nchunk <= 2) launchAccum<2>(grid, st, device_matrix, num_rows, C, ld, device_counts, d_a.ptr, d_partials.ptr, partial_ld, d_ctl.ptr);
            else if (nchunk <= 4) launchAccum<4>(grid, st, device_matrix, num_rows, C, ld, device_counts, d_a.ptr, d_partials.ptr, partial_ld, d_ctl.ptr);
            else if (nchunk <= 8) launchAccum<8>(grid, st, device_matrix, num_rows, C, ld, device_counts, d_a.ptr, d_partials.ptr, partial_ld, d_ctl.ptr);
            else launchAccum<16>(grid, st, device_matrix, num_rows, C, ld, device_counts, d_a.ptr, d_partials.ptr, partial_ld, d_ctl.ptr);
            ctx->spanEnd(span);
            if (sharded) {
                emDenseReducePartialsKernel<<<col_grid, dim3(256), 0, st>>>(C, grid, partial_ld, d_partials.ptr, d_t.ptr, d_ctl.ptr, d_t.ptr + C);
                if (const int rc = ctx->allReduceSumF64(d_t.ptr, C + 1)) {
                    (void) hipStreamSynchronize(st);  // the kernels queued so far use the buffers freed on return
                    return rc;
                }
                emDenseFinalizeKernel<<<col_grid, dim3(256), 0, st>>>(C, 1, partial_ld, d_t.ptr, d_a.ptr, total_count,
                                                                  max_rel_em_conv, d_ctl.ptr, d_t.ptr + C, zero_mass);
            } else {
                emDenseFinalizeKernel<<<col_grid, dim3(256), 0, st>>>(C, grid, partial_ld, d_partials.ptr, d_a.ptr,
                                                                  total_count, max_rel_em_conv, d_ctl.ptr, nullptr, zero_mass);
            }
            emDenseControlKernel<<<dim3(1), dim3(1), 0, st>>>(d_ctl.ptr, max_em_its);
        }
        queued += n;
        RPVG_HIP_CHECK(hipGetLastError());
        RPVG_HIP_CHECK(hipMemcpyAsync(&h_ctl, d_ctl.ptr, sizeof(DenseControl), hipMemcpyDeviceToHost, st));
        RPVG_HIP_CHECK(hipStreamSynchronize(st));
    }

    if (h_ctl.error) {
        setError("%s: the column sums of some rank were not finite in EM iteration %u (status word of the all-reduce): every rank stopped there", who,
                 h_ctl.iterations + 1);
        return RPVG_HIP_ERR_RUNTIME;
    }
    // launches that actually streamed the matrix = iterations executed
    ctx->stats.em_dense_launches += h_ctl.iterations;
    ctx->stats.em_dense_alg_bytes += static_cast<double>(h_ctl.iterations) *
                                     (8.0 * static_cast<double>(num_rows) * C + 8.0 * static_cast<double>(num_rows) + 16.0 * C);
    ctx->stats.em_iterations_total += h_ctl.iterations;
    return RPVG_HIP_OK;
}

}  // namespace rpvg_hip_detail

extern "C" int rpvg_hip_em_dense(rpvg_hip_ctx * ctx, const double * device_matrix, uint64_t num_rows, uint32_t num_cols,
                                 uint64_t ld, const double * device_counts, double total_count, uint32_t max_em_its,
                                 double max_rel_em_conv, double * abundances, double * noise_count,
                                 uint32_t * iterations) {
    return emDenseRun(ctx, "rpvg_hip_em_dense", false, device_matrix, num_rows, num_cols, ld, device_counts, total_count,
                      max_em_its, max_rel_em_conv, abundances, noise_count, iterations);
}

extern "C" int rpvg_hip_em_dense_sharded(rpvg_hip_ctx * ctx, const double * device_matrix, uint64_t num_rows,
                                         uint32_t num_cols, uint64_t ld, const double * device_counts, double total_count,
                                         uint32_t max_em_its, double max_rel_em_conv, double * abundances,
                                         double * noise_count, uint32_t * iterations) {
    return emDenseRun(ctx, "rpvg_hip_em_dense_sharded", true, device_matrix, num_rows, num_cols, ld, device_counts,
                      total_count, max_em_its, max_rel_em_conv, abundances, noise_count, iterations);
}

extern "C" int rpvg_hip_dense_from_cluster(rpvg_hip_ctx * ctx, const rpvg_hip_batch * batch, uint32_t cluster,
                                           double * device_matrix, uint64_t ld, double * device_counts,
                                           double * total_count) {
    RPVG_REQUIRE(ctx && batch && device_matrix && device_counts && total_count, "rpvg_hip_dense_from_cluster: NULL argument");
    RPVG_REQUIRE(cluster < batch->num_clusters, "rpvg_hip_dense_from_cluster: cluster %u of %u", cluster, batch->num_clusters);
    const uint64_t r0 = batch->h_cluster_row_off[cluster], r1 = batch->h_cluster_row_off[cluster + 1];
    const uint32_t N = static_cast<uint32_t>(batch->h_cluster_path_off[cluster + 1] - batch->h_cluster_path_off[cluster]);
    RPVG_REQUIRE(r1 > r0 && N > 0, "rpvg_hip_dense_from_cluster: cluster %u is empty", cluster);
    RPVG_REQUIRE(ld >= N + 1 && (ld % 2) == 0, "rpvg_hip_dense_from_cluster: ld must be even and >= paths + 1");

    std::lock_guard<std::mutex> lock(ctx->mutex);
    RPVG_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const uint64_t R = r1 - r0;
    DeviceBuffer<double> d_total;
    RPVG_HIP_CHECK(d_total.alloc(1));
    const int span = ctx->spanBegin(FAM_BUILD);
    RPVG_HIP_CHECK(hipMemsetAsync(d_total.ptr, 0, sizeof(double), st));
    RPVG_HIP_CHECK(hipMemsetAsync(device_matrix, 0, sizeof(double) * R * ld, st));
    denseFromClusterKernel<<<dim3(static_cast<uint32_t>((R + 255) / 256)), dim3(256), 0, st>>>(
        r0, R, N, batch->row_ent_off.ptr, batch->ent_path.ptr, batch->ent_prob.ptr, batch->row_count.ptr,
        batch->row_noise.ptr, device_matrix, ld, device_counts, d_total.ptr);
    ctx->spanEnd(span);
    ctx->stats.build_launches += 1;
    RPVG_HIP_CHECK(hipGetLastError());
    RPVG_HIP_CHECK(hipMemcpyAsync(total_count, d_total.ptr, sizeof(double), hipMemcpyDeviceToHost, st));
    RPVG_HIP_CHECK(hipStreamSynchronize(st));
    return RPVG_HIP_OK;
}

extern "C" int rpvg_hip_synth_dense_cluster(rpvg_hip_ctx * ctx, uint64_t seed, uint64_t num_rows, uint32_t num_paths,
                                            double * device_matrix, uint64_t ld, double * device_counts) {
    return rpvg_hip_synth_dense_rows(ctx, seed, 0, num_rows, num_paths, device_matrix, ld, device_counts);
}

namespace {

// host-side tables of the synthetic cluster (tiny): theta ~ LogNormal(0, 2) normalised, lengths ~ U[200, 5000]
struct SynthHostTables {
    std::vector<double> theta, inv_len, score_prob, deficit_cdf;
    DeviceBuffer<double> d_theta, d_inv_len, d_score, d_def;
    explicit SynthHostTables(const uint64_t seed, const uint32_t N) : theta(N), inv_len(N), score_prob(21), deficit_cdf(20) {
        auto hmix = [](uint64_t x) {
            x += 0x9E3779B97F4A7C15ull;
            x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
            x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
            return x ^ (x >> 31);
        };
        auto hu01 = [](uint64_t h) { return ((h >> 11) + 0.5) * (1.0 / 9007199254740992.0); };
        double tsum = 0;
        for (uint32_t j = 0; j < N; ++j) {
            const double u1 = hu01(hmix(seed ^ (0xA000000000ull + 2 * j))), u2 = hu01(hmix(seed ^ (0xA000000000ull + 2 * j + 1)));
            const double z = std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);  // Box-Muller
            theta[j] = std::exp(2.0 * z);
            tsum += theta[j];
            inv_len[j] = 1.0 / (200.0 + 4800.0 * hu01(hmix(seed ^ (0xB000000000ull + j))));
        }
        double acc = 0;
        for (uint32_t j = 0; j < N; ++j) {
            acc += theta[j] / tsum;
            theta[j] = acc;
        }
        theta[N - 1] = 1.0;
        for (int d = 0; d <= 20; ++d) score_prob[d] = std::exp(-1.383325268738 * d);  // Utils::score_log_base, src/utils.hpp:83
        double pk = std::exp(-3.0), cdf = 0;  // Poisson(3)
        for (int k = 0; k < 20; ++k) {
            cdf += pk;
            deficit_cdf[k] = cdf;  // deficit d = 1 + k
            pk *= 3.0 / (k + 1);
        }
        deficit_cdf[19] = 1.0;
    }
    hipError_t upload(hipStream_t st, SynthTables * tab) {
        hipError_t e = d_theta.upload(theta.data(), theta.size(), st);
        if (e == hipSuccess) e = d_inv_len.upload(inv_len.data(), inv_len.size(), st);
        if (e == hipSuccess) e = d_score.upload(score_prob.data(), 21, st);
        if (e == hipSuccess) e = d_def.upload(deficit_cdf.data(), 20, st);
        *tab = SynthTables{d_theta.ptr, d_inv_len.ptr, d_score.ptr, d_def.ptr};
        return e;
    }
};

}  // namespace

extern "C" int rpvg_hip_synth_dense_rows(rpvg_hip_ctx * ctx, uint64_t seed, uint64_t row_begin, uint64_t num_rows,
                                         uint32_t num_paths, double * device_matrix, uint64_t ld, double * device_counts) {
    RPVG_REQUIRE(ctx && device_matrix && device_counts, "rpvg_hip_synth_dense_rows: NULL argument");
    RPVG_REQUIRE(num_rows > 0 && num_paths > 0, "rpvg_hip_synth_dense_rows: empty cluster");
    RPVG_REQUIRE(ld >= num_paths + 1 && (ld % 2) == 0, "rpvg_hip_synth_dense_rows: ld must be even and >= paths + 1");
    const uint32_t N = num_paths;
    SynthHostTables tables(seed, N);

    std::lock_guard<std::mutex> lock(ctx->mutex);
    RPVG_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    SynthTables tab;
    RPVG_HIP_CHECK(tables.upload(st, &tab));
    const uint64_t threads = num_rows * 64;
    synthDenseKernel<<<dim3(static_cast<uint32_t>((threads + 255) / 256)), dim3(256), 0, st>>>(seed, row_begin, num_rows, N, tab,
                                                                                            device_matrix, ld, device_counts);
    RPVG_HIP_CHECK(hipGetLastError());
    RPVG_HIP_CHECK(hipStreamSynchronize(st));
    return RPVG_HIP_OK;
}

extern "C" int rpvg_hip_synth_dense_cluster_batch(rpvg_hip_ctx * ctx, uint64_t seed, uint64_t num_rows, uint32_t num_paths,
                                                  rpvg_hip_batch ** batch_out) {
    RPVG_REQUIRE(ctx && batch_out, "rpvg_hip_synth_dense_cluster_batch: NULL argument");
    *batch_out = nullptr;
    RPVG_REQUIRE(num_rows > 0 && num_paths > 0, "rpvg_hip_synth_dense_cluster_batch: empty cluster");
    RPVG_REQUIRE(num_rows * num_paths < 0xffffffffull, "rpvg_hip_synth_dense_cluster_batch: %llu x %u entries exceed the 32-bit entry offsets of an EM problem",
                 static_cast<unsigned long long>(num_rows), num_paths);
    const uint32_t N = num_paths;
    const uint64_t R = num_rows, M = R * N;
    SynthHostTables tables(seed, N);

    std::lock_guard<std::mutex> lock(ctx->mutex);
    RPVG_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    std::unique_ptr<rpvg_hip_batch> b(new (std::nothrow) rpvg_hip_batch());
    if (!b) {
        setError("rpvg_hip_synth_dense_cluster_batch: out of host memory");
        return RPVG_HIP_ERR_ALLOC;
    }
    b->num_clusters = 1;
    b->num_rows = R;
    b->num_entries = M;
    b->num_paths = N;
    b->h_cluster_row_off = {0, R};
    b->h_cluster_path_off = {0, N};
    b->h_cluster_ent_off = {0, M};
    RPVG_HIP_CHECK(b->cluster_row_off.upload(b->h_cluster_row_off.data(), 2, st));
    RPVG_HIP_CHECK(b->cluster_path_off.upload(b->h_cluster_path_off.data(), 2, st));
    RPVG_HIP_CHECK(b->row_noise.alloc(R));
    RPVG_HIP_CHECK(b->row_count.alloc(R));
    RPVG_HIP_CHECK(b->row_ent_off.alloc(R + 1));
    RPVG_HIP_CHECK(b->ent_path.alloc(M));
    RPVG_HIP_CHECK(b->ent_prob.alloc(M));
    SynthTables tab;
    RPVG_HIP_CHECK(tables.upload(st, &tab));
    const uint64_t threads = R * 64;
    synthDenseBatchKernel<<<dim3(static_cast<uint32_t>((threads + 255) / 256)), dim3(256), 0, st>>>(seed, 0, R, N, tab, b->row_ent_off.ptr, b->ent_path.ptr,
                                                                                                 b->ent_prob.ptr, b->row_count.ptr, b->row_noise.ptr);
    RPVG_HIP_CHECK(hipGetLastError());
    RPVG_HIP_CHECK(hipStreamSynchronize(st));
    *batch_out = b.release();
    return RPVG_HIP_OK;
}
