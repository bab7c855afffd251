// Batched EM abundance solves on ragged, sparse cluster problems (gfx950).
//
// Takes over, for a whole batch of (cluster, column-subset) problems at once:
//   constructPartialProbabilityMatrix       src/path_estimator.cpp:79-113
//   addNoiseAndNormalizeProbabilityMatrix   src/path_estimator.cpp:156-166
//   EMAbundanceEstimator                    src/path_abundance_estimator.cpp:47-114
//
// Pipeline per rpvg_hip_em_solve() call (all on the context's stream):
//   1. scatterColumnMapKernel   path -> column (or -1) map of every problem
//   2. fillSegmentsKernel       ordered compaction into a per-problem CSR of
//                               row-normalised entries  P_ij/rowsum_i*(1-noise_i),
//                               at offsets the host knows (a problem keeps at most
//                               the rows and entries of its cluster); counts the rows
//                               and entries that survive the column subset and the
//                               read mass of the rows that touch no selected path
//   3. (host) the counts: cost-descending order, size bins
//   4. emSparseKernel<BLOCK>    ONE workgroup per problem runs the whole EM
//                               loop on the GPU: abundance vector a[] and the
//                               M-step accumulators t[] live in LDS, the
//                               problem's CSR streams from L2/HBM every
//                               iteration, convergence is decided on-device.
//
// EM iteration, fused to one pass over the rows:
//   s_i = noise_i*a_noise + sum_e P_e*a[col_e] ;  w_i = count_i / s_i
//   t[col_e] += w_i*P_e ; t_noise += w_i*noise_i
//   a'_j = a_j*t_j/T ;  a'_noise = (a_noise*t_noise + Z)/T
// where Z is the read mass of the rows without any selected path: such a row
// is (0,...,0,noise_i) after normalisation, its posterior is exactly 1 on the
// noise component for every a, so its M-step contribution is the constant
// count_i (this also subsumes what readCollapseProbabilityMatrix,
// src/path_estimator.cpp:219-259, would merge among those rows).

#include "common.hpp"

#include <algorithm>
#include <memory>
#include <numeric>

using namespace rpvg_hip_detail;

namespace {

constexpr double kMinEmAbundance = 1e-8;   // src/path_abundance_estimator.cpp:11
constexpr uint32_t kMinEmConvIts = 10;     // src/path_abundance_estimator.cpp:10

// ---- block-level primitives (wave = 64) -------------------------------------

template <typename T>
__device__ __forceinline__ T waveReduceSum(T v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

template <>
__device__ __forceinline__ double waveReduceSum<double>(double v) {
    return waveSumF64(v);
}

// Sum over the block, result in every thread.  scratch: BLOCK/64 elements.
template <typename T, int BLOCK>
__device__ __forceinline__ T blockReduceSum(T v, T * scratch) {
    v = waveReduceSum(v);
    if (BLOCK == 64) return v;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    T total = scratch[0];
#pragma unroll
    for (int w = 1; w < BLOCK / 64; ++w) total += scratch[w];
    return total;
}

// Exclusive scan of the pair (a, b) over the block; totals to every thread.
// scratch: 2*BLOCK/64 uint32.
template <int BLOCK>
__device__ __forceinline__ void blockExclusiveScanPair(uint32_t & a, uint32_t & b, uint32_t & total_a, uint32_t & total_b,
                                                       uint32_t * scratch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t ia = a, ib = b;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t ta = __shfl_up(ia, d, 64), tb = __shfl_up(ib, d, 64);
        if (lane >= d) {
            ia += ta;
            ib += tb;
        }
    }
    __syncthreads();
    if (lane == 63) {
        scratch[2 * wave] = ia;
        scratch[2 * wave + 1] = ib;
    }
    __syncthreads();
    uint32_t off_a = 0, off_b = 0, ta = 0, tb = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w) {
        const uint32_t xa = scratch[2 * w], xb = scratch[2 * w + 1];
        if (w < wave) {
            off_a += xa;
            off_b += xb;
        }
        ta += xa;
        tb += xb;
    }
    a = off_a + ia - a;
    b = off_b + ib - b;
    total_a = ta;
    total_b = tb;
}

// ---- size bins of the EM kernels -----------------------------------------------
// One kernel variant per bin (rpvg_hip_em_kernel_name); the bin of a problem follows from its columns (paths + noise),
// kept rows and kept entries alone, so the device decides it (fillOffsetsKernel) and the host repeats the decision for
// the statistics:
//   0  LDS-resident, one wave      CSR + vectors fit 8 KB
//   1  LDS-resident, four waves    fit 40 KB
//   2  streamed from L2, 4 waves
//   3  streamed from L2, 16 waves  (a few giant problems)
//   4-6 register-resident dense, one wave: at most 16 columns and 64 / 128 / 256 rows (emRegisterKernel)
//   7  LDS-resident, sixteen waves  CSR + vectors fit 152 KB (one workgroup per CU: the whole LDS)
//   8-9 register-resident dense, one wave: 17 to 32 columns and 64 / 128 rows
//   10 too many columns for LDS-resident vectors (> ~9 700): vectors in global memory, 16 waves
//   11 the grid bin: rows + entries at or above EmBinRule::grid_min_work — not one workgroup but the whole GPU, one round
//      of launches per EM iteration, driven by the host (em_grid.hip); no kernel of this file serves it
constexpr int kEmBins = RPVG_HIP_EM_KERNELS;
constexpr int kEmGridBin = 11;
constexpr int kEmWorkBuckets = 32;   // inside a bin the problems are ordered by floor(log2(rows + entries)), large first
// A FEW mid-size problems — streamed ones (bin 3) of 2^16 rows + entries and more, below the grid threshold — take the grid route
// too: one workgroup walks such a problem at ~30 us per EM iteration (20 000 rows x 3 entries: 33), the whole GPU at ~8, and a real
// cluster of that size runs hundreds to thousands of iterations.  Only a few, because the grid takes its problems a handful at
// a time where the one-workgroup kernels take them all side by side: emOrderKernel moves them when the solve has at most
// kEmMidGridMax of them (the histogram tells it), and leaves them where they are otherwise.
constexpr uint32_t kEmMidGridMax = 8;
constexpr uint32_t kEmMidGridLog2 = 16;  // (a bucket is floor(log2(work + 1)): work + 1 >= 2^16)
constexpr int kEmStreamedBin = 3;
constexpr size_t kEmLdsLimit = 156 * 1024;
constexpr uint32_t kRegColsMax = 32;  // the widest register-resident variant

// LDS bytes of a problem: the abundance vector, one accumulator vector PER WAVEFRONT of the workgroup (the M-step's sums have one
// order of additions: emSparseProblem) and scratch, plus its CSR when resident
__host__ __device__ inline size_t emLdsBytes(uint32_t cols, uint32_t rows, uint32_t entries, int block, bool resident) {
    size_t bytes = sizeof(double) * ((1 + static_cast<size_t>(block) / 64) * cols + block / 64 + 2);
    if (resident) bytes += static_cast<size_t>(rows) * 16 + static_cast<size_t>(entries) * 8 + (static_cast<size_t>(rows) + 1 + entries) * 4 + 8;
    return (bytes + 15) & ~static_cast<size_t>(15);
}

// the grid route's workgroups (em_grid.hip: four wavefronts): abundances + an accumulator vector per wavefront
__host__ __device__ inline size_t emGridLdsBytes(const uint32_t cols) { return sizeof(double) * 5 * static_cast<size_t>(cols); }

struct EmBinRule {
    uint32_t use_register_kernel;   // RPVG_HIP_NO_REGISTER_EM=1 clears it
    uint64_t streamed_small_work;   // a streamed problem above this many rows + entries gets 1 024 threads instead of 256
    uint64_t grid_min_work;         // rows + entries from which a problem goes to the grid bin (0: never; emGridMinWork())
};

__host__ __device__ inline int emBinOf(const EmBinRule rule, const uint32_t C, const uint32_t rows, const uint32_t entries) {
    const uint64_t work = static_cast<uint64_t>(entries) + rows;
    // (the grid kernels keep the vectors in LDS: the few problems too wide for that stay in bin 10)
    if (rule.grid_min_work != 0 && work >= rule.grid_min_work && emGridLdsBytes(C) <= kEmLdsLimit) return kEmGridBin;
    if (rule.use_register_kernel && C <= 16 && rows <= 256) return rows <= 64 ? 4 : rows <= 128 ? 5 : 6;
    if (rule.use_register_kernel && C <= kRegColsMax && rows <= 128) return rows <= 64 ? 8 : 9;
    if (emLdsBytes(C, rows, entries, 64, true) <= 8 * 1024) return 0;
    if (emLdsBytes(C, rows, entries, 256, true) <= 40 * 1024) return 1;
    if (emLdsBytes(C, rows, entries, 1024, true) <= 152 * 1024) return 7;
    // streamed: sixteen wavefronts if their accumulator vectors fit LDS, four if those do (up to ~3 900 columns), else the vectors
    // in global memory
    if (emLdsBytes(C, 0, 0, 256, false) > kEmLdsLimit) return 10;
    if (emLdsBytes(C, 0, 0, 1024, false) > kEmLdsLimit) return 2;
    return work <= rule.streamed_small_work ? 2 : 3;
}

__host__ __device__ inline uint32_t emWorkBucket(const uint32_t rows, const uint32_t entries) {
    // floor(log2(work + 1)), inverted: bucket 0 holds the largest problems
    uint64_t work = static_cast<uint64_t>(entries) + rows + 1;
    uint32_t lg = 0;
    while (work > 1 && lg < static_cast<uint32_t>(kEmWorkBuckets - 1)) {
        work >>= 1;
        ++lg;
    }
    return static_cast<uint32_t>(kEmWorkBuckets - 1) - lg;
}

// Work queues of one rpvg_hip_em_solve on the device (zero-initialised): the fill kernel counts the problems of every
// (bin, bucket); emOrderKernel turns the counts into offsets and lists the problems; the EM kernels — persistent
// workgroups — draw problems of their bin from bin_cursor until bin_count is reached.  The host never needs the counts
// to launch anything.
struct EmQueues {
    uint32_t histogram[kEmBins * kEmWorkBuckets];
    uint32_t bucket_cursor[kEmBins * kEmWorkBuckets];
    uint32_t bin_start[kEmBins];
    uint32_t bin_count[kEmBins];
    uint32_t bin_cursor[kEmBins];
    uint32_t pad;
    unsigned long long wide_cursor;   // doubles handed out of EmLaunchArgs::wide_vectors
    unsigned long long wide_overflow; // set when a wide problem did not get its vectors (capacity exceeded)
};

// ---- 1 + 2. column map, count and fill ------------------------------------------
// The rows of a problem's cluster are cut into segments of kFillSegmentRows; a work item is one segment of one problem.
// Three launches:
//   fillSegmentsKernel<false>  per item: the path -> column map of the problem in LDS (clusters of up to kLdsMapPaths
//                              paths; wider ones look their paths up in the sorted column list by bisection), then the
//                              segment's rows that touch a selected path, their entries, and the read mass — counted
//   fillOffsetsKernel          per problem: exclusive prefix of its segments' counts, the problem's totals, its size bin
//   fillSegmentsKernel<true>   per item: ordered compaction of the segment behind the rows of the segments before it
//                              (block-wide exclusive scan per 256 rows), P / rowsum * (1 - noise) with the reference's two
//                              roundings (addNoiseAndNormalizeProbabilityMatrix, src/path_estimator.cpp:156-166)
// (Round 2 and the first version of round 3: ONE workgroup per problem walked all rows of its cluster, 256 at a time, two
// block scans each — a 100 000-row cluster was 400 dependent steps and the kernel, 0.5-0.8 ms per lane, was as long as
// its largest cluster.)
constexpr uint32_t kLdsMapPaths = 16384;
constexpr uint32_t kFillSegmentRows = 1024;
constexpr uint64_t kFillLongRowEntries = 32;   // mean entries per row from which a cluster's rows take a wavefront each
constexpr size_t kFillLongRowLds = kFillSegmentRows * (3 * sizeof(uint32_t) + sizeof(double));

struct FillArgs {
    uint32_t num_problems;                 // upper bound when num_problems_dev is set
    const uint32_t * num_problems_dev;     // null: num_problems is exact
    uint32_t num_items;                    // segments of all problems; upper bound when num_items_dev is set
    const uint32_t * num_items_dev;
    const uint64_t * seg_first;            // [P+1] first item of each problem
    const uint32_t * item_problem;         // [items]
    const uint32_t * prob_cluster;         // [P]
    const uint64_t * col_off;              // [P+1]
    const uint32_t * col_path;
    const uint64_t * cluster_row_off;
    const uint64_t * cluster_path_off;
    const uint64_t * row_ent_off;
    const uint32_t * ent_path;
    const double * ent_prob;
    const double * row_count;
    const double * row_noise;
    const uint64_t * row_base;             // [P] first compacted row of the problem
    const uint64_t * ent_base;             // [P] first compacted entry of the problem
    uint32_t * prow_off;                   // [rows_total + P] per problem kept_rows+1 offsets relative to its entry base; null: count only
    double * prow_count;
    double * prow_noise;
    uint32_t * pent_col;
    double * pent_val;
    // per item: what the segment keeps (counted), then where it starts inside its problem (fillOffsetsKernel)
    uint32_t * seg_rows;
    uint32_t * seg_entries;
    double * seg_zero_mass;
    double * seg_total_mass;
    uint32_t * kept_rows;                  // [P] the counts of the problem
    uint32_t * kept_entries;
    double * zero_mass;
    double * total_mass;
    uint32_t * prob_bucket;                // [P] bin * kEmWorkBuckets + bucket (with the storage only)
    EmQueues * queues;
    EmBinRule rule;
    uint32_t lds_map_paths;                // capacity of the LDS map of this launch (0: bisection for every problem)
    uint32_t long_row_scratch;             // the launch carries kFillLongRowLds bytes of LDS behind the map: clusters of long rows take a wavefront per row
    // problems whose rows are written as a dense row-major matrix instead of a CSR (the fused build: queueEmSolve)
    uint32_t num_fused;
    EmFusedDense fused[kEmMaxFusedDense];
};

// whether the rows of cluster k take a wavefront each in fillSegmentsKernel
__device__ inline bool fillLongRows(const FillArgs & args, const uint32_t k) {
    const uint64_t c0 = args.cluster_row_off[k], c1 = args.cluster_row_off[k + 1];
    return args.long_row_scratch && args.row_ent_off[c1] - args.row_ent_off[c0] >= kFillLongRowEntries * (c1 - c0);
}

template <bool WRITE>
__global__ __launch_bounds__(256) void fillSegmentsKernel(const FillArgs args) {
    constexpr int BLOCK = 256;
    extern __shared__ __attribute__((aligned(16))) int32_t lds_map[];
    __shared__ uint32_t scratch[2 * (BLOCK / 64)];
    __shared__ double dscratch[BLOCK / 64];
    const uint32_t num_items = args.num_items_dev ? *args.num_items_dev : args.num_items;
    uint32_t mapped_problem = UINT32_MAX;
    for (uint32_t item = blockIdx.x; item < num_items; item += gridDim.x) {
        const uint32_t p = args.item_problem[item];
        if (WRITE) {  // (the rows of these problems go straight into their dense matrices: fillDenseRowsKernel)
            bool fused = false;
            for (uint32_t f = 0; f < args.num_fused; ++f) fused = fused || args.fused[f].problem == p;
            if (fused) continue;
        }
        const uint32_t segment = static_cast<uint32_t>(item - args.seg_first[p]);
        const uint32_t k = args.prob_cluster[p];
        const uint32_t n_paths = static_cast<uint32_t>(args.cluster_path_off[k + 1] - args.cluster_path_off[k]);
        const uint32_t * cols = args.col_path + args.col_off[p];
        const uint32_t n_cols = static_cast<uint32_t>(args.col_off[p + 1] - args.col_off[p]);
        const bool use_map = n_paths <= args.lds_map_paths;
        // every path of the cluster is a column (the `transcripts` model: one problem per cluster over all of its paths — the column
        // list is ascending and without repeats, so as long as the cluster it is 0, 1, 2, ...): no map, and every entry is kept
        const bool identity = n_cols == n_paths;
        __syncthreads();  // (the previous item is done with the map and the scratch)
        if (use_map && !identity && mapped_problem != p) {
            for (uint32_t i = threadIdx.x; i < n_paths; i += BLOCK) lds_map[i] = -1;
            __syncthreads();
            for (uint32_t c = threadIdx.x; c < n_cols; c += BLOCK) lds_map[cols[c]] = static_cast<int32_t>(c);
            __syncthreads();
            mapped_problem = p;
        }
        auto column_of = [&](const uint32_t path) -> int32_t {
            if (identity) return static_cast<int32_t>(path);
            if (use_map) return lds_map[path];
            uint32_t lo = 0, hi = n_cols;  // first column whose path is not below `path`
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (cols[mid] < path) lo = mid + 1;
                else hi = mid;
            }
            return (lo < n_cols && cols[lo] == path) ? static_cast<int32_t>(lo) : -1;
        };
        const uint64_t c0 = args.cluster_row_off[k], c1 = args.cluster_row_off[k + 1];
        const uint64_t r0 = c0 + static_cast<uint64_t>(segment) * kFillSegmentRows, r1 = min(c1, r0 + kFillSegmentRows);
        const uint64_t rb = WRITE ? args.row_base[p] : 0, eb = WRITE ? args.ent_base[p] : 0;
        // the offsets array has one extra slot per problem
        uint32_t * off = WRITE ? args.prow_off + rb + p : nullptr;
        uint32_t run_rows = WRITE ? args.seg_rows[item] : 0, run_ent = WRITE ? args.seg_entries[item] : 0;  // (starts, by now)
        double z = 0, t = 0;  // read counts of the rows without a selected path / of all rows
        if (fillLongRows(args, k)) {
            // A cluster of long rows (the 2 000-path rows of BASELINE.json configs[1]): a wavefront per row, its lanes striding the
            // row's entries — 512-byte requests — instead of a thread walking 2 000 entries 24 KB from its neighbour's (0.2 s for
            // the 1 M x 2 000 cluster, as long as eighty EM iterations over it).  Kept entries of a row and, for the write, its
            // row sum go to LDS; the rows' places come from a scan over the segment; a row's entries are compacted 64 at a time
            // (ballot + prefix count), in order.
            uint32_t * row_n = reinterpret_cast<uint32_t *>(lds_map + args.lds_map_paths);  // [kFillSegmentRows] kept entries
            uint32_t * row_slot = row_n + kFillSegmentRows, * row_eoff = row_slot + kFillSegmentRows;  // exclusive prefixes of (kept ? 1 : 0), kept entries
            double * row_sum = reinterpret_cast<double *>(row_eoff + kFillSegmentRows);
            const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
            const uint32_t seg_n = static_cast<uint32_t>(r1 - r0);
            for (uint32_t i = wave; i < seg_n; i += BLOCK / 64) {
                const uint64_t r = r0 + i;
                const uint64_t e0 = args.row_ent_off[r], e1 = args.row_ent_off[r + 1];
                uint32_t n = 0;
                double sum = 0;
                if (identity) {
                    n = static_cast<uint32_t>(e1 - e0);
                    if (WRITE) {
                        for (uint64_t e = e0 + lane; e < e1; e += 64) sum += args.ent_prob[e];
                    }
                } else {
                    for (uint64_t e = e0 + lane; e < e1; e += 64) {
                        if (column_of(args.ent_path[e]) >= 0) {
                            ++n;
                            sum += args.ent_prob[e];
                        }
                    }
                    for (int d = 32; d >= 1; d >>= 1) n += __shfl_xor(n, d, 64);
                }
                if (WRITE) sum = waveSumF64(sum);
                if (lane == 0) {
                    row_n[i] = n;
                    if (WRITE) row_sum[i] = sum;
                    if (!WRITE) {
                        const double c = args.row_count[r];
                        t += c;
                        if (!n) z += c;
                    }
                }
            }
            __syncthreads();
            // exclusive scan over the segment's rows: a thread takes kFillSegmentRows / BLOCK neighbouring rows
            constexpr uint32_t kPer = kFillSegmentRows / BLOCK;
            uint32_t slots = 0, ents = 0;
            for (uint32_t j = 0; j < kPer; ++j) {
                const uint32_t i = threadIdx.x * kPer + j;
                const uint32_t n = i < seg_n ? row_n[i] : 0u;
                slots += n ? 1u : 0u;
                ents += n;
            }
            uint32_t slot0 = slots, ent0 = ents, tot_rows, tot_ent;
            blockExclusiveScanPair<BLOCK>(slot0, ent0, tot_rows, tot_ent, scratch);
            if (WRITE) {
                for (uint32_t j = 0; j < kPer; ++j) {
                    const uint32_t i = threadIdx.x * kPer + j;
                    if (i >= seg_n) break;
                    row_slot[i] = slot0;
                    row_eoff[i] = ent0;
                    slot0 += row_n[i] ? 1u : 0u;
                    ent0 += row_n[i];
                }
                __syncthreads();
                for (uint32_t i = wave; i < seg_n; i += BLOCK / 64) {
                    if (!row_n[i]) continue;
                    const uint64_t r = r0 + i;
                    const uint32_t my_row = run_rows + row_slot[i];
                    const uint32_t my_ent = run_ent + row_eoff[i];
                    const double nz = args.row_noise[r];
                    if (lane == 0) {
                        off[my_row] = my_ent;
                        args.prow_count[rb + my_row] = args.row_count[r];
                        args.prow_noise[rb + my_row] = nz;
                    }
                    const double keep = 1 - nz, rowsum = row_sum[i];
                    const uint64_t e0 = args.row_ent_off[r], e1 = args.row_ent_off[r + 1];
                    uint32_t base = 0;
                    for (uint64_t eb0 = e0; eb0 < e1; eb0 += 64) {
                        const uint64_t e = eb0 + lane;
                        const int32_t c = e < e1 ? column_of(args.ent_path[e]) : -1;
                        const unsigned long long kept = __ballot(c >= 0);
                        if (c >= 0) {
                            const uint32_t at = my_ent + base + static_cast<uint32_t>(__popcll(kept & ((1ull << lane) - 1)));
                            args.pent_col[eb + at] = static_cast<uint32_t>(c);
                            // addNoiseAndNormalizeProbabilityMatrix: (P / rowsum) * (1 - noise), two roundings
                            args.pent_val[eb + at] = (args.ent_prob[e] / rowsum) * keep;
                        }
                        base += static_cast<uint32_t>(__popcll(kept));
                    }
                }
            } else {
                z = blockReduceSum<double, BLOCK>(z, dscratch);
                t = blockReduceSum<double, BLOCK>(t, dscratch);
                if (threadIdx.x == 0) {
                    args.seg_rows[item] = tot_rows;
                    args.seg_entries[item] = tot_ent;
                    args.seg_zero_mass[item] = z;
                    args.seg_total_mass[item] = t;
                }
            }
            continue;
        }
        for (uint64_t rc = r0; rc < r1; rc += BLOCK) {
            const uint64_t r = rc + threadIdx.x;
            uint32_t n = 0;
            double rowsum = 0;
            uint64_t e0 = 0, e1 = 0;
            if (r < r1) {
                e0 = args.row_ent_off[r];
                e1 = args.row_ent_off[r + 1];
                for (uint64_t e = e0; e < e1; ++e) {
                    if (column_of(args.ent_path[e]) >= 0) {
                        ++n;
                        rowsum += args.ent_prob[e];
                    }
                }
                if (!WRITE) {
                    const double c = args.row_count[r];
                    t += c;
                    if (!n) z += c;
                }
            }
            uint32_t slot = n ? 1u : 0u, epos = n, tot_rows, tot_ent;
            blockExclusiveScanPair<BLOCK>(slot, epos, tot_rows, tot_ent, scratch);
            if (WRITE && n) {
                const uint32_t my_row = run_rows + slot;
                uint32_t my_ent = run_ent + epos;
                off[my_row] = my_ent;
                const double nz = args.row_noise[r];
                args.prow_count[rb + my_row] = args.row_count[r];
                args.prow_noise[rb + my_row] = nz;
                const double keep = 1 - nz;
                for (uint64_t e = e0; e < e1; ++e) {
                    const int32_t c = column_of(args.ent_path[e]);
                    if (c >= 0) {
                        args.pent_col[eb + my_ent] = static_cast<uint32_t>(c);
                        // addNoiseAndNormalizeProbabilityMatrix: (P / rowsum) * (1 - noise), two roundings
                        args.pent_val[eb + my_ent] = (args.ent_prob[e] / rowsum) * keep;
                        ++my_ent;
                    }
                }
            }
            run_rows += tot_rows;
            run_ent += tot_ent;
        }
        if (!WRITE) {
            z = blockReduceSum<double, BLOCK>(z, dscratch);
            t = blockReduceSum<double, BLOCK>(t, dscratch);
            if (threadIdx.x == 0) {
                args.seg_rows[item] = run_rows;
                args.seg_entries[item] = run_ent;
                args.seg_zero_mass[item] = z;
                args.seg_total_mass[item] = t;
            }
        }
    }
}

// The wavefront-per-row path costs 20 KB of LDS per workgroup: only a solve that sits on a cluster large enough to matter
// (the grid threshold of the EM: a batch of small clusters keeps its eight workgroups per CU) carries it.
template <bool WRITE>
hipError_t launchFillSegments(FillArgs & fa, const uint32_t grid, const uint64_t max_cluster_work, hipStream_t st) {
    static const bool never = RPVG_EXPERIMENT_ENV("RPVG_HIP_FILL_THREAD_ROWS") != nullptr;  // A/B knob
    fa.long_row_scratch = (!never && max_cluster_work >= (1ull << 18)) ? 1u : 0u;
    fa.lds_map_paths = (fa.lds_map_paths + 3) & ~3u;  // (the scratch behind the map holds doubles, and fillDenseRowsKernel moves 16 bytes at a time)
    const size_t lds = fa.lds_map_paths * sizeof(int32_t) + (fa.long_row_scratch ? kFillLongRowLds : 0);
    if (lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&fillSegmentsKernel<WRITE>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        if (e != hipSuccess) return e;
    }
    fillSegmentsKernel<WRITE><<<dim3(grid), dim3(256), lds, st>>>(fa);
    return hipSuccess;
}

// The fused build: the rows of problem args.fused[blockIdx.y] from its cluster straight into its dense row-major matrix — what
// fillSegmentsKernel<true> and emGridDenseBuildKernel (em_grid.hip) do in two steps with the compacted CSR and a zeroed matrix
// between them, with the same arithmetic in the same order (row sum: a lane's entries in ascending order, then waveSumF64;
// value = (P / rowsum) * (1 - noise)), so the matrix is the same to the bit.  A workgroup takes segments of the cluster's rows,
// a wavefront a row at a time: the row's entries are loaded (at most 64 x kDenseRowCache: into registers, read once; longer
// rows: read twice), an image of the matrix row in LDS is zeroed meanwhile, the kept entries go to their columns of the image
// and the noise term behind the last path, and the image leaves in 16-byte stores, one behind the other.  12 B read per entry,
// 8 B written per cell, nothing written twice (zeros and values to the matrix itself: 4.1 TB/s of algorithmic bytes; the zeros
// reached HBM).
constexpr int kDenseRowCache = 32;
constexpr size_t kFillDenseLds = kFillSegmentRows * 2 * sizeof(uint32_t);   // + an image of a row per wavefront

__global__ __launch_bounds__(256) void fillDenseRowsKernel(const FillArgs args) {
    constexpr int BLOCK = 256;
    extern __shared__ __attribute__((aligned(16))) int32_t lds_map[];
    __shared__ uint32_t scratch[2 * (BLOCK / 64)];
    const EmFusedDense fd = args.fused[blockIdx.y];
    const uint32_t p = fd.problem;
    const uint32_t k = args.prob_cluster[p];
    const uint32_t n_paths = static_cast<uint32_t>(args.cluster_path_off[k + 1] - args.cluster_path_off[k]);
    const uint32_t * cols = args.col_path + args.col_off[p];
    const uint32_t n_cols = static_cast<uint32_t>(args.col_off[p + 1] - args.col_off[p]);
    const bool identity = n_cols == n_paths;  // (fillSegmentsKernel)
    const bool use_map = !identity && n_paths <= args.lds_map_paths;
    uint32_t * row_n = reinterpret_cast<uint32_t *>(lds_map + args.lds_map_paths);  // [kFillSegmentRows] kept entries of a row
    uint32_t * row_slot = row_n + kFillSegmentRows;                                 // its place among the segment's kept rows
    double * image = reinterpret_cast<double *>(row_slot + kFillSegmentRows) + (threadIdx.x >> 6) * fd.ld;  // [ld] of this wavefront
    if (use_map) {
        for (uint32_t i = threadIdx.x; i < n_paths; i += BLOCK) lds_map[i] = -1;
        __syncthreads();
        for (uint32_t c = threadIdx.x; c < n_cols; c += BLOCK) lds_map[cols[c]] = static_cast<int32_t>(c);
    }
    auto column_of = [&](const uint32_t path) -> int32_t {
        if (identity) return static_cast<int32_t>(path);
        if (use_map) return lds_map[path];
        uint32_t lo = 0, hi = n_cols;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (cols[mid] < path) lo = mid + 1;
            else hi = mid;
        }
        return (lo < n_cols && cols[lo] == path) ? static_cast<int32_t>(lo) : -1;
    };
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t c0 = args.cluster_row_off[k], c1 = args.cluster_row_off[k + 1];
    const uint64_t rb = args.row_base[p];
    for (uint64_t item = args.seg_first[p] + blockIdx.x; item < args.seg_first[p + 1]; item += gridDim.x) {
        const uint32_t segment = static_cast<uint32_t>(item - args.seg_first[p]);
        const uint64_t r0 = c0 + static_cast<uint64_t>(segment) * kFillSegmentRows, r1 = min(c1, r0 + kFillSegmentRows);
        const uint32_t seg_n = static_cast<uint32_t>(r1 - r0);
        const uint32_t run_rows = args.seg_rows[item];  // (the segment's first row among the problem's kept rows: fillOffsetsKernel)
        __syncthreads();  // (the map is complete; the previous segment is done with the scratch)
        for (uint32_t i = wave; i < seg_n; i += BLOCK / 64) {
            const uint64_t e0 = args.row_ent_off[r0 + i], e1 = args.row_ent_off[r0 + i + 1];
            uint32_t n = static_cast<uint32_t>(e1 - e0);
            if (!identity) {
                n = 0;
                for (uint64_t e = e0 + lane; e < e1; e += 64) n += column_of(args.ent_path[e]) >= 0 ? 1u : 0u;
                for (int d = 32; d >= 1; d >>= 1) n += __shfl_xor(n, d, 64);
            }
            if (lane == 0) row_n[i] = n;
        }
        __syncthreads();
        constexpr uint32_t kPer = kFillSegmentRows / BLOCK;
        uint32_t slots = 0, unused = 0, tot_rows, tot_unused;
        for (uint32_t j = 0; j < kPer; ++j) {
            const uint32_t i = threadIdx.x * kPer + j;
            slots += (i < seg_n && row_n[i]) ? 1u : 0u;
        }
        uint32_t slot0 = slots;
        blockExclusiveScanPair<BLOCK>(slot0, unused, tot_rows, tot_unused, scratch);
        for (uint32_t j = 0; j < kPer; ++j) {
            const uint32_t i = threadIdx.x * kPer + j;
            if (i >= seg_n) break;
            row_slot[i] = slot0;
            slot0 += row_n[i] ? 1u : 0u;
        }
        __syncthreads();
        for (uint32_t i = wave; i < seg_n; i += BLOCK / 64) {
            if (!row_n[i]) continue;
            const uint64_t r = r0 + i;
            const uint32_t my_row = run_rows + row_slot[i];
            const double nz = args.row_noise[r];
            if (lane == 0) {
                args.prow_count[rb + my_row] = args.row_count[r];
                args.prow_noise[rb + my_row] = nz;
            }
            const double keep = 1 - nz;
            const uint64_t e0 = args.row_ent_off[r], e1 = args.row_ent_off[r + 1];
            double * out = fd.matrix + static_cast<uint64_t>(my_row) * fd.ld;
            // (a wavefront's LDS instructions execute in order: zeros, values, the read — the compiler keeps the order of accesses
            // that may touch the same words)
            for (uint32_t j = 2 * lane; j < fd.ld; j += 128) *reinterpret_cast<double2 *>(image + j) = double2{0.0, 0.0};
            if (e1 - e0 <= 64ull * kDenseRowCache) {
                int32_t col[kDenseRowCache];
                double val[kDenseRowCache];
                const uint32_t len = static_cast<uint32_t>(e1 - e0);
#pragma unroll
                for (int j = 0; j < kDenseRowCache; ++j) {
                    const uint32_t at = lane + 64u * j;
                    col[j] = -1;
                    val[j] = 0.0;
                    if (64u * j < len && at < len) {
                        col[j] = static_cast<int32_t>(args.ent_path[e0 + at]);
                        val[j] = args.ent_prob[e0 + at];
                    }
                }
                double sum = 0;
#pragma unroll
                for (int j = 0; j < kDenseRowCache; ++j) {
                    if (64u * j < len) {
                        if (col[j] >= 0) col[j] = column_of(static_cast<uint32_t>(col[j]));
                        if (col[j] >= 0) sum += val[j];
                    }
                }
                const double rowsum = waveSumF64(sum);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int j = 0; j < kDenseRowCache; ++j) {
                    // addNoiseAndNormalizeProbabilityMatrix: (P / rowsum) * (1 - noise), two roundings
                    if (64u * j < len && col[j] >= 0) image[col[j]] = (val[j] / rowsum) * keep;
                }
            } else {
                double sum = 0;
                for (uint64_t e = e0 + lane; e < e1; e += 64) {
                    if (column_of(args.ent_path[e]) >= 0) sum += args.ent_prob[e];
                }
                const double rowsum = waveSumF64(sum);
                __builtin_amdgcn_wave_barrier();
                for (uint64_t e = e0 + lane; e < e1; e += 64) {
                    const int32_t c = column_of(args.ent_path[e]);
                    if (c >= 0) image[c] = (args.ent_prob[e] / rowsum) * keep;
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) image[n_cols] = nz;
            __builtin_amdgcn_wave_barrier();
            for (uint32_t j = 2 * lane; j < fd.ld; j += 128) *reinterpret_cast<double2 *>(out + j) = *reinterpret_cast<const double2 *>(image + j);
            __builtin_amdgcn_wave_barrier();  // (the next row's zeros stay behind these reads)
        }
    }
}

// per problem: the counts of its segments become their starts; totals, terminal offset, size bin
__global__ __launch_bounds__(256) void fillOffsetsKernel(const FillArgs args) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (args.num_problems_dev ? *args.num_problems_dev : args.num_problems)) return;
    uint32_t rows = 0, entries = 0;
    double z = 0, t = 0;  // (read counts: integers, exact in any order)
    for (uint64_t item = args.seg_first[p]; item < args.seg_first[p + 1]; ++item) {
        const uint32_t r = args.seg_rows[item], e = args.seg_entries[item];
        args.seg_rows[item] = rows;
        args.seg_entries[item] = entries;
        rows += r;
        entries += e;
        z += args.seg_zero_mass[item];
        t += args.seg_total_mass[item];
    }
    if (args.prow_off) {
        args.prow_off[args.row_base[p] + p + rows] = entries;
        const uint32_t n_cols = static_cast<uint32_t>(args.col_off[p + 1] - args.col_off[p]);
        const uint32_t bucket = static_cast<uint32_t>(emBinOf(args.rule, n_cols + 1, rows, entries)) * kEmWorkBuckets + emWorkBucket(rows, entries);
        args.prob_bucket[p] = bucket;
        atomicAdd(&args.queues->histogram[bucket], 1u);
    }
    args.kept_rows[p] = rows;
    args.kept_entries[p] = entries;
    args.zero_mass[p] = z;
    args.total_mass[p] = t;
}

// The problems of the grid bin that take the dense route (em_grid.hip) and sit on a cluster of long rows: the host gives each a
// matrix and the compaction writes their rows into it — at most kEmMaxFusedDense of them, whichever come first (the others take
// the CSR and the copy: the same matrix either way).
struct DenseCandidate {
    uint32_t problem, rows, columns, entries;
};
struct DenseCandidates {
    uint32_t count, pad;
    DenseCandidate list[kEmMaxFusedDense];
};
__global__ __launch_bounds__(256) void denseCandidatesKernel(const FillArgs args, DenseCandidates * __restrict__ out) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (args.num_problems_dev ? *args.num_problems_dev : args.num_problems)) return;
    if (args.prob_bucket[p] / kEmWorkBuckets != static_cast<uint32_t>(kEmGridBin)) return;
    const uint32_t columns = static_cast<uint32_t>(args.col_off[p + 1] - args.col_off[p]) + 1;
    const uint32_t rows = args.kept_rows[p], entries = args.kept_entries[p];
    if (!emDenseRule(columns, rows, entries) || !fillLongRows(args, args.prob_cluster[p])) return;
    const uint32_t at = atomicAdd(&out->count, 1u);
    if (at < static_cast<uint32_t>(kEmMaxFusedDense)) out->list[at] = DenseCandidate{p, rows, columns, entries};
}

// ---- 3. the work queues ------------------------------------------------------------
// Every workgroup scans the (bin, bucket) histogram in LDS (352 counters) and lists its problems: a problem's place
// inside its bucket is whatever the atomic hands out — the order inside a bucket only decides who starts first.
__global__ __launch_bounds__(256) void emOrderKernel(const uint32_t num_problems, const uint32_t * __restrict__ num_problems_dev,
                                                    const uint32_t * __restrict__ prob_bucket, const uint64_t * __restrict__ col_off,
                                                    EmQueues * __restrict__ queues, uint32_t * __restrict__ order,
                                                    unsigned long long * __restrict__ wide_off, const unsigned long long wide_capacity,
                                                    const uint32_t mid_grid_allowed) {
    constexpr int kCells = kEmBins * kEmWorkBuckets;
    static_assert(kCells <= 512, "two cells per thread");
    __shared__ uint32_t start[kCells + 1];
    __shared__ uint32_t wave_total[4];
    __shared__ uint32_t moved_cells;
    const uint32_t P = num_problems_dev ? *num_problems_dev : num_problems;
    // the few mid-size problems that take the grid route (above): the cells (streamed bin, buckets of 2^16 work units and more)
    // count as cells of the grid bin — every workgroup reaches the same verdict from the same histogram
    constexpr uint32_t kMidBuckets = kEmWorkBuckets - kEmMidGridLog2;  // buckets 0 .. kMidBuckets - 1 hold work + 1 >= 2^16
    if (threadIdx.x == 0) {
        uint32_t mid = 0;
        for (uint32_t b = 0; b < kMidBuckets; ++b) mid += queues->histogram[kEmStreamedBin * kEmWorkBuckets + b];
        moved_cells = (mid_grid_allowed && mid > 0 && mid <= kEmMidGridMax) ? kMidBuckets : 0u;
    }
    __syncthreads();
    const uint32_t moved = moved_cells;
    auto cellCount = [&](const uint32_t c) -> uint32_t {
        if (c >= kCells) return 0u;
        const uint32_t bin = c / kEmWorkBuckets, bucket = c % kEmWorkBuckets;
        if (bucket < moved && bin == kEmStreamedBin) return 0u;
        if (bucket < moved && bin == kEmGridBin) return queues->histogram[c] + queues->histogram[kEmStreamedBin * kEmWorkBuckets + bucket];
        return queues->histogram[c];
    };
    {
        // exclusive prefix of the histogram: thread t owns cells 2 t and 2 t + 1
        const uint32_t c0 = 2 * threadIdx.x, c1 = c0 + 1;
        const uint32_t h0 = cellCount(c0), h1 = cellCount(c1);
        uint32_t incl = h0 + h1;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d, 64);
            if (lane >= d) incl += up;
        }
        if (lane == 63) wave_total[wave] = incl;
        __syncthreads();
        uint32_t before = 0;
        for (int w = 0; w < wave; ++w) before += wave_total[w];
        const uint32_t excl = before + incl - (h0 + h1);
        if (c0 < kCells) start[c0] = excl;
        if (c1 < kCells) start[c1] = excl + h0;
        if (threadIdx.x == 255) start[kCells] = before + incl;
    }
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x < kEmBins) {
        queues->bin_start[threadIdx.x] = start[threadIdx.x * kEmWorkBuckets];
        queues->bin_count[threadIdx.x] = start[(threadIdx.x + 1) * kEmWorkBuckets] - start[threadIdx.x * kEmWorkBuckets];
    }
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    uint32_t cell = prob_bucket[p];
    if (cell / kEmWorkBuckets == kEmStreamedBin && cell % kEmWorkBuckets < moved) cell = kEmGridBin * kEmWorkBuckets + cell % kEmWorkBuckets;
    order[start[cell] + atomicAdd(&queues->bucket_cursor[cell], 1u)] = p;
    if (cell / kEmWorkBuckets == 10) {  // abundance + accumulator vectors in global memory
        const unsigned long long need = 2ull * (static_cast<unsigned long long>(col_off[p + 1] - col_off[p]) + 1);
        const unsigned long long at = atomicAdd(&queues->wide_cursor, need);
        wide_off[p] = at;
        if (at + need > wide_capacity) atomicAdd(&queues->wide_overflow, 1ull);
    }
}

// ---- 4. the problems of the grid bin, described to the host ----------------------------------------
// The grid bin has no kernel of its own here: its problems are solved over the whole GPU with one round of launches per
// EM iteration, which the host drives (em_grid.hip).  One thread per problem of the bin writes what the host needs.
struct GridDescribeArgs {
    const EmQueues * queues;
    const uint32_t * order;
    const uint64_t * col_off;
    const uint64_t * row_base;
    const uint64_t * ent_base;
    const uint32_t * kept_rows;
    const uint32_t * kept_entries;
    const double * zero_mass;
    const double * total_mass;
    const uint32_t * problem_merged;  // NULL: no collapse
    EmGridProblem * out;
    uint32_t capacity;
};

__global__ __launch_bounds__(256) void emGridDescribeKernel(const GridDescribeArgs args) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= args.queues->bin_count[kEmGridBin] || i >= args.capacity) return;
    const uint32_t p = args.order[args.queues->bin_start[kEmGridBin] + i];
    EmGridProblem d;
    d.problem = p;
    d.columns = static_cast<uint32_t>(args.col_off[p + 1] - args.col_off[p]) + 1;
    d.rows = args.kept_rows[p];
    d.entries = args.kept_entries[p];
    d.merged = (args.problem_merged != nullptr && args.problem_merged[p] != 0) ? 1u : 0u;
    d.pad = 0;
    d.row_base = args.row_base[p];
    d.ent_base = args.ent_base[p];
    d.col_begin = args.col_off[p];
    d.total_mass = args.total_mass[p];
    d.zero_mass = args.zero_mass[p];
    args.out[i] = d;
}

// ---- 5. the EM kernel --------------------------------------------------------

struct EmLaunchArgs {
    const uint32_t * order;        // problems, bin by bin (EmQueues::bin_start), large first
    EmQueues * queues;
    uint32_t bin;                  // the bin this launch serves
    const uint64_t * col_off;      // [P+1]
    const uint64_t * row_base;     // [P]
    const uint64_t * ent_base;     // [P]
    const uint32_t * kept_rows;    // [P]
    const double * zero_mass;      // [P]
    const double * total_mass;     // [P]
    const uint32_t * prow_off;
    const double * prow_count;
    const double * merged_count;      // read counts after the row collapse (row_collapse.hip), for the problems with ...
    const uint32_t * problem_merged;  // ... this flag; NULL: no collapse
    const double * prow_noise;
    const uint32_t * pent_col;
    const double * pent_val;
    uint32_t max_em_its;
    uint32_t register_copies;      // emRegisterKernel<1,16>: small problems in copies (RPVG_HIP_EM_COPIES=0: never)
    double max_rel_em_conv;
    double * wide_vectors;         // problems too wide for LDS: abundance + accumulator vectors, 2 C doubles each,
    const unsigned long long * wide_off;  // [P] at this offset (emSparseKernel<..., WIDE>)
    unsigned long long wide_capacity;
    double * abundances;           // [col_off[P]]
    double * noise_count;          // [P]
    uint32_t * iterations;         // [P]
};

// the read counts of problem p's rows: a row whose count the row collapse moved to its run head has none left and takes no part
__device__ __forceinline__ const double * rowCounts(const EmLaunchArgs & args, const uint32_t p, const uint64_t rb) {
    return (args.problem_merged != nullptr && args.problem_merged[p] != 0 ? args.merged_count : args.prow_count) + rb;
}

// The EM kernels are persistent: a launch has as many workgroups as the GPU holds at once (or as the bin can have
// problems, if fewer), and every workgroup draws the next problem of its bin from the queue until the bin is empty —
// the host launches every variant without knowing how many problems each bin got.  Returns the problem, or
// UINT32_MAX when the bin is exhausted; uniform over the workgroup.
template <int BLOCK>
__device__ __forceinline__ uint32_t nextProblem(const EmLaunchArgs & args, uint32_t * slot) {
    uint32_t i;
    if (BLOCK == 64) {
        i = (threadIdx.x == 0) ? atomicAdd(&args.queues->bin_cursor[args.bin], 1u) : 0u;
        i = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(i)));
    } else {
        __syncthreads();  // everybody is done with the previous problem (and with *slot)
        if (threadIdx.x == 0) *slot = atomicAdd(&args.queues->bin_cursor[args.bin], 1u);
        __syncthreads();
        i = *slot;
    }
    if (i >= args.queues->bin_count[args.bin]) return UINT32_MAX;
    return args.order[args.queues->bin_start[args.bin] + i];
}

// RESIDENT: the problem's compacted CSR is copied into LDS once and every EM iteration runs out of LDS
// (small problems need up to thousands of iterations; from L2 each costs ~1.7 us of dependent-load
// latency, from LDS a fraction of that).
// WIDE: the abundance and accumulator vectors do not fit LDS (more than ~10 000 columns: the reference's EM has no
// size limit, and clusters such as HLA exceed this): they live in global memory (L2), the M-step uses global FP64 atomics.
template <int BLOCK, bool RESIDENT, bool WIDE>
__device__ __forceinline__ void emSparseProblem(const EmLaunchArgs & args, const uint32_t p, unsigned char * smem_raw) {
    const uint32_t C = static_cast<uint32_t>(args.col_off[p + 1] - args.col_off[p]) + 1;  // + noise
    if (WIDE && args.wide_off[p] + 2ull * C > args.wide_capacity) return;  // (reported through EmQueues::wide_overflow)
    // The M-step's column sums have ONE order of additions, whatever the wavefronts' timing: every wavefront adds into an
    // accumulator vector of its own — within a wavefront the additions follow the program and, lanes of one instruction that meet on
    // a column, the LDS unit's lane order — and the vectors are added up in wavefront order (registers -> LDS slots -> ordered
    // sum).  Two runs on the same input give the same bits, iteration counts included.  (WIDE — more than ~3 900 columns, the
    // vectors in global memory — keeps one vector and global atomics: reproducible up to the order of those additions.)
    constexpr uint32_t kCopies = WIDE ? 1 : BLOCK / 64;
    double * a = WIDE ? args.wide_vectors + args.wide_off[p] : reinterpret_cast<double *>(smem_raw);  // [C] abundances (last = noise)
    double * t = a + C;                                 // [kCopies x C] M-step accumulators
    double * red = WIDE ? reinterpret_cast<double *>(smem_raw) : t + static_cast<size_t>(kCopies) * C;  // [BLOCK/64] reduction scratch
    double * tw = t + static_cast<size_t>(kCopies == 1 ? 0 : threadIdx.x >> 6) * C;  // this wavefront's

    const uint32_t n_rows = args.kept_rows[p];
    const uint64_t rb = args.row_base[p], eb = args.ent_base[p];
    const uint32_t * off = args.prow_off + rb + p;
    const double * cnt = rowCounts(args, p, rb);
    const double * nzv = args.prow_noise + rb;
    const uint32_t * col = args.pent_col + eb;
    const double * val = args.pent_val + eb;
    if (RESIDENT) {
        const uint32_t n_ent = off[n_rows];
        double * l_cnt = red + (BLOCK / 64 + 2);
        double * l_nz = l_cnt + n_rows;
        double * l_val = l_nz + n_rows;
        uint32_t * l_off = reinterpret_cast<uint32_t *>(l_val + n_ent);
        uint32_t * l_col = l_off + (n_rows + 1);
        for (uint32_t r = threadIdx.x; r < n_rows; r += BLOCK) {
            l_cnt[r] = cnt[r];
            l_nz[r] = nzv[r];
        }
        for (uint32_t r = threadIdx.x; r <= n_rows; r += BLOCK) l_off[r] = off[r];
        for (uint32_t e = threadIdx.x; e < n_ent; e += BLOCK) {
            l_val[e] = val[e];
            l_col[e] = col[e];
        }
        off = l_off;
        cnt = l_cnt;
        nzv = l_nz;
        col = l_col;
        val = l_val;
    }
    const double T = args.total_mass[p];
    const double Z = args.zero_mass[p];
    const double eps = args.max_rel_em_conv;
    const uint32_t noise_col = C - 1;

    // src/path_abundance_estimator.cpp:54 — 1 / float(C), widened
    const double a0 = static_cast<double>(1.0f / static_cast<float>(C));
    for (uint32_t j = threadIdx.x; j < C; j += BLOCK) a[j] = a0;

    for (uint32_t j = threadIdx.x; j < kCopies * C; j += BLOCK) t[j] = 0;
    __syncthreads();  // a[], t[] and (when resident) the problem's CSR are in LDS

    const double inv_T = 1.0 / T;
    uint32_t iters = 0, conv = 0;
    for (uint32_t it = 0; it < args.max_em_its; ++it) {
        const double a_noise = a[noise_col];
        double tn = 0;
        for (uint32_t r = threadIdx.x; r < n_rows; r += BLOCK) {
            const uint32_t e0 = off[r], e1 = off[r + 1];
            const double nz = nzv[r];
            double s = nz * a_noise;
            for (uint32_t e = e0; e < e1; ++e) s += val[e] * a[col[e]];
            // cnt / s: hardware reciprocal, two Newton steps, one residual correction of the quotient
            double y = __builtin_amdgcn_rcp(s);
            y = fma(fma(-s, y, 1.0), y, y);
            y = fma(fma(-s, y, 1.0), y, y);
            const double quot = cnt[r] * y;
            // (a row whose count a row collapse moved to its run head takes no part: row_collapse.hip)
            const double w = cnt[r] == 0.0 ? 0.0 : fma(fma(-s, quot, cnt[r]), y, quot);
            for (uint32_t e = e0; e < e1; ++e) atomicAdd(&tw[col[e]], w * val[e]);
            tn += w * nz;
        }
        // the noise column has no entries: its accumulator takes the per-wave sums of w * noise
        tn = waveSumF64(tn);
        if ((threadIdx.x & 63) == 0 && tn != 0.0) atomicAdd(&tw[noise_col], tn);
        __syncthreads();  // all atomics to t[] done, all reads of a[] done
        int viol = 0;
        for (uint32_t j = threadIdx.x; j < C; j += BLOCK) {
            const double aj = a[j];
            double tj = t[j];
            t[j] = 0;
#pragma unroll
            for (uint32_t w = 1; w < kCopies; ++w) {  // (wavefront order)
                tj += t[w * C + j];
                t[w * C + j] = 0;
            }
            const double an = (j == noise_col) ? (aj * tj + Z) * inv_T : (aj * tj) * inv_T;
            // |an - aj| / an > eps  (src/path_abundance_estimator.cpp:73-75), without the division
            if (an >= kMinEmAbundance && fabs(an - aj) > eps * an) viol = 1;
            a[j] = an;
        }
        int any_viol;
        if (BLOCK == 64) {
            any_viol = __any(viol);
            __syncthreads();
        } else {
            any_viol = __syncthreads_or(viol);
        }
        ++iters;
        if (!any_viol) {
            if (++conv == kMinEmConvIts) break;
        } else {
            conv = 0;
        }
    }

    // src/path_abundance_estimator.cpp:100-113
    double low = 0;
    double * out = args.abundances + args.col_off[p];
    for (uint32_t j = threadIdx.x; j < noise_col; j += BLOCK) {
        const double aj = a[j];
        if (aj < kMinEmAbundance) {
            low += aj * T;
            out[j] = 0;
        } else {
            out[j] = aj * T;
        }
    }
    low = blockReduceSum<double, BLOCK>(low, red);
    if (threadIdx.x == 0) {
        args.noise_count[p] = low + a[noise_col] * T;
        args.iterations[p] = iters;
    }
}

template <int BLOCK, bool RESIDENT, bool WIDE = false>
__global__ __launch_bounds__(BLOCK) void emSparseKernel(const EmLaunchArgs args) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ uint32_t next_slot;
    if (blockIdx.x >= args.queues->bin_count[args.bin]) return;  // more workgroups than the bin has problems
    for (uint32_t p = nextProblem<BLOCK>(args, &next_slot); p != UINT32_MAX; p = nextProblem<BLOCK>(args, &next_slot)) {
        emSparseProblem<BLOCK, RESIDENT, WIDE>(args, p, smem_raw);
        if (BLOCK == 64) __syncthreads();  // (the LDS is reused; wider workgroups meet in nextProblem)
    }
}

// `grid`: workgroups of the persistent launch (the caller's bound on the problems of the bin, capped by what the GPU holds)
template <int BLOCK, bool RESIDENT, bool WIDE = false>
hipError_t launchEm(const EmLaunchArgs & args, uint32_t grid, size_t lds, hipStream_t stream) {
    if (grid == 0) return hipSuccess;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&emSparseKernel<BLOCK, RESIDENT, WIDE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        if (e != hipSuccess) return e;
    }
    emSparseKernel<BLOCK, RESIDENT, WIDE><<<dim3(grid), dim3(BLOCK), lds, stream>>>(args);
    return hipGetLastError();
}


// ---- register-resident EM for small problems ---------------------------------------------------------------
// A problem with at most COLS columns — its paths AND the noise component — and 64 * RPL rows is held DENSE in the
// registers of one wavefront: lane l owns rows l, l + 64, ... (RPL of them) as COLS doubles each, the noise probability
// of the row in the column behind the last path.  An iteration then needs no dependent memory access for the matrix:
// the E-step is RPL x COLS fused multiply-adds against the abundance vector, the M-step RPL x COLS more into COLS
// per-lane partial column sums, which cross the wave once through LDS (64 / COLS lanes per column + one or two DPP
// steps inside the quad).  These problems are the ones that run for thousands of iterations (the batch's EM time is the
// iteration count of its slowest problem times the latency of ONE iteration — a 14-path, 17-row problem with 2 494 of
// them in the configs[2] bench).  Zero entries add exact zeros; the column sums are added in a fixed order (the sparse
// kernels use LDS atomics).  Round 3: the noise component became a column like the others (it was a wave-wide DPP sum
// of its own next to the transposition: 27 instructions of ~230 per iteration), the update runs in every lane of a
// column without a branch, and the transposition is skewed (below).
// value of the lane whose index differs in bit 0 (D = 1) or bit 1 (D = 2) — inside a quad, through DPP
template <int D>
__device__ __forceinline__ double quadSwapF64(const double v) {
    constexpr int perm = (D == 1) ? 0xB1 : 0x4E;  // quad_perm [1,0,3,2] / [2,3,0,1]
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), perm, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), perm, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

// x of the lanes 32 .. 63 <-> y of the lanes 0 .. 31 (v_permlane32_swap_b32, new with gfx950): afterwards x + y is, in a lane
// of the lower half, the sum of the two x of the lanes l and l + 32, in a lane of the upper half that of the two y
__device__ __forceinline__ void swapHalvesF64(double & x, double & y) {
    const auto lo = __builtin_amdgcn_permlane32_swap(static_cast<unsigned>(__double2loint(x)), static_cast<unsigned>(__double2loint(y)), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(static_cast<unsigned>(__double2hiint(x)), static_cast<unsigned>(__double2hiint(y)), false, false);
    x = __hiloint2double(static_cast<int>(hi[0]), static_cast<int>(lo[0]));
    y = __hiloint2double(static_cast<int>(hi[1]), static_cast<int>(lo[1]));
}

// the same between neighbouring rows of 16 lanes: x of the odd rows <-> y of the even rows (v_permlane16_swap_b32)
__device__ __forceinline__ void swapRowsF64(double & x, double & y) {
    const auto lo = __builtin_amdgcn_permlane16_swap(static_cast<unsigned>(__double2loint(x)), static_cast<unsigned>(__double2loint(y)), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(static_cast<unsigned>(__double2hiint(x)), static_cast<unsigned>(__double2hiint(y)), false, false);
    x = __hiloint2double(static_cast<int>(hi[0]), static_cast<int>(lo[0]));
    y = __hiloint2double(static_cast<int>(hi[1]), static_cast<int>(lo[1]));
}

// One halving step inside a row of 16 lanes: the DPP control CTRL pairs every lane with one whose `upper` differs; a
// lane keeps x (upper: y) and adds the partner's copy of it.
template <int CTRL>
__device__ __forceinline__ double halveF64(const double x, const double y, const bool upper) {
    const double send = upper ? x : y, keep = upper ? y : x;
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(send), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(send), CTRL, 0xF, 0xF, true);
    return keep + __hiloint2double(hi, lo);
}

// Column sums over the wave of COLS per-lane partials v[0 .. COLS), without LDS: every step halves the number of values
// a lane carries and doubles the lanes each value has been added over — lanes l and l + 32 (one swap per 32-bit
// register half and one addition for TWO columns), neighbouring rows of 16 lanes, then inside the row through DPP.
// Afterwards lane l holds, in the return value, the sum of column l / (64 / COLS) over all 64 lanes; the order of the
// additions is fixed.  (Round 2 and the first version of round 3 transposed the partials through LDS: 16 writes and 16
// reads per lane, 625 of an iteration's 1 180 cycles in LDS issue and waits — PMC, tools/r03_em_pmc.sh.)
// SKIP: the first SKIP halving steps have nothing to add (problems of at most 32 / 16 rows whose lanes 32 .. 63 / 16 .. 63 carry
// copies of the rows, every copy the partial sums of its share of the columns in v[0 .. COLS >> SKIP): exactly where those steps
// would have put them, the other addends being zeros).
template <int COLS, int SKIP = 0>
__device__ __forceinline__ double columnSumsOverWave(double (&v)[COLS], const uint32_t lane) {
    if (SKIP < 1) {
#pragma unroll
        for (int c = 0; c < COLS / 2; ++c) {
            swapHalvesF64(v[c], v[c + COLS / 2]);
            v[c] += v[c + COLS / 2];
        }
    }
    if (SKIP < 2) {
#pragma unroll
        for (int c = 0; c < COLS / 4; ++c) {
            swapRowsF64(v[c], v[c + COLS / 4]);
            v[c] += v[c + COLS / 4];
        }
    }
    const bool bit3 = (lane & 8) != 0, bit2 = (lane & 4) != 0;
#pragma unroll
    for (int c = 0; c < COLS / 8; ++c) v[c] = halveF64<0x128>(v[c], v[c + COLS / 8], bit3);   // row_ror:8
#pragma unroll
    for (int c = 0; c < COLS / 16; ++c) v[c] = halveF64<0x141>(v[c], v[c + COLS / 16], bit2);  // row_half_mirror
    double t;
    if (COLS == 32) {
        t = halveF64<0x4E>(v[0], v[1], (lane & 2) != 0);  // quad_perm [2,3,0,1]
    } else {
        t = v[0];
        t += quadSwapF64<2>(t);
    }
    t += quadSwapF64<1>(t);
    return t;
}

// COPIES (one row per lane, 16 columns): a problem of at most 64 / COPIES rows is held COPIES times — lane l carries row
// l % (64 / COPIES) — and every copy takes 16 / COPIES of the columns in the M-step: 16 / COPIES multiplications instead of 16, and
// the first log2(COPIES) steps of the column sums (eight and four swaps and additions) fall away.  The E-step, and with it every
// value, is what it is with one copy (the steps left out add zeros).  The batch's EM time is the latency of one iteration times
// the iterations of its slowest problems, and those are small (2 268 iterations on 17 rows in the configs[2] batch).
template <int RPL, int COLS, int COPIES = 1>
__device__ __forceinline__ void emRegisterProblem(const EmLaunchArgs & args, const uint32_t p, double * reg_lds) {
    static_assert(COPIES == 1 || (RPL == 1 && COLS == 16), "copies: one row per lane, 16 columns");
    constexpr uint32_t kCols = COLS;
    constexpr int kSkip = COPIES == 4 ? 2 : (COPIES == 2 ? 1 : 0);
    constexpr int kShare = COLS / COPIES;  // columns of a copy in the M-step
    constexpr int kLanesPerColumn = 64 / COLS;  // 4 (16 columns) or 2 (32 columns)
    const uint32_t np = static_cast<uint32_t>(args.col_off[p + 1] - args.col_off[p]);  // < kCols; column np = noise
    const uint32_t lane = threadIdx.x;
    const uint32_t n_rows = args.kept_rows[p];
    const uint64_t rb = args.row_base[p], eb = args.ent_base[p];
    const uint32_t * off = args.prow_off + rb + p;
    const double * cnt = rowCounts(args, p, rb);
    const double * nzv = args.prow_noise + rb;
    const uint32_t * col = args.pent_col + eb;
    const double * val = args.pent_val + eb;

    // stage the dense tile through LDS (the scatter needs dynamic indexing, registers must not)
    constexpr uint32_t kTile = 64 * RPL * kCols;
    double * tile = reg_lds;
    for (uint32_t idx = lane; idx < kTile; idx += 64) tile[idx] = 0.0;
    __syncthreads();
    for (uint32_t r = lane; r < n_rows; r += 64) {
        for (uint32_t e = off[r]; e < off[r + 1]; ++e) tile[r * kCols + col[e]] += val[e];
        tile[r * kCols + np] = nzv[r];
    }
    __syncthreads();
    double P[RPL][kCols], c[RPL];
    double mine[kShare];  // (COPIES > 1) the row's values in the columns of this copy
    bool valid[RPL];
#pragma unroll
    for (int q = 0; q < RPL; ++q) {
        const uint32_t r = COPIES > 1 ? lane % (64 / COPIES) : q * 64 + lane;
        c[q] = r < n_rows ? cnt[r] : 0.0;
        valid[q] = c[q] != 0.0;  // (also: a row whose count a row collapse moved to its run head, row_collapse.hip)
#pragma unroll
        for (int j = 0; j < static_cast<int>(kCols); ++j) P[q][j] = tile[r * kCols + j];
        if (COPIES > 1) {
            const uint32_t first_column = (lane / (64 / COPIES)) * kShare;
#pragma unroll
            for (int j = 0; j < kShare; ++j) mine[j] = tile[r * kCols + first_column + j];
        }
    }
    const double T = args.total_mass[p];
    const double eps = args.max_rel_em_conv;
    // src/path_abundance_estimator.cpp:54 — 1 / float(C), widened
    const double a0 = static_cast<double>(1.0f / static_cast<float>(np + 1));
    // Column j belongs to lanes kLanesPerColumn * j ...: each holds a_j and applies the update (the same arithmetic on the
    // same values: no lane is special, nothing branches).  Columns behind the noise column are padding: their
    // abundance starts, and therefore stays, at zero.
    const uint32_t my_col = lane / kLanesPerColumn;
    double a_mine = (my_col <= np) ? a0 : 0.0;
    // Z: the read mass of the rows without any selected path, which the noise component takes whole (header of this file)
    const double z_mine = (my_col == np) ? args.zero_mass[p] : 0.0;

    const double inv_T = 1.0 / T;

    uint32_t iters = 0, conv = 0;
    for (uint32_t it = 0; it < args.max_em_its; ++it) {
        // the abundance vector lives in its owner lanes (a_mine): read straight from them (scalar broadcast)
        double av[kCols];
#pragma unroll
        for (int j = 0; j < static_cast<int>(kCols); ++j) av[j] = readLaneF64(a_mine, j * kLanesPerColumn);
        double w[RPL];
#pragma unroll
        for (int q = 0; q < RPL; ++q) {
            // four interleaved partial sums: the latency of one iteration is what this kernel is about
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
            for (int j = 0; j < static_cast<int>(kCols); j += 4) {
                s0 = fma(P[q][j], av[j], s0);
                s1 = fma(P[q][j + 1], av[j + 1], s1);
                s2 = fma(P[q][j + 2], av[j + 2], s2);
                s3 = fma(P[q][j + 3], av[j + 3], s3);
            }
            const double s = (s0 + s1) + (s2 + s3);
            // c / s: hardware reciprocal, two Newton steps, one residual correction of the quotient
            double y = __builtin_amdgcn_rcp(s);
            y = fma(fma(-s, y, 1.0), y, y);
            y = fma(fma(-s, y, 1.0), y, y);
            const double quot = c[q] * y;
            w[q] = valid[q] ? fma(fma(-s, quot, c[q]), y, quot) : 0.0;
        }
        double pj[kCols];
        if (COPIES > 1) {
#pragma unroll
            for (int j = 0; j < kShare; ++j) pj[j] = w[0] * mine[j];
        } else {
#pragma unroll
            for (int j = 0; j < static_cast<int>(kCols); ++j) {
                pj[j] = w[0] * P[0][j];
#pragma unroll
                for (int q = 1; q < RPL; ++q) pj[j] = fma(w[q], P[q][j], pj[j]);
            }
        }
        const double tj = columnSumsOverWave<COLS, kSkip>(pj, lane);

        // a'_j = a_j t_j / T;  a'_noise = (a_noise t_noise + Z) / T  (z_mine is zero off the noise column)
        const double an = fma(a_mine, tj, z_mine) * inv_T;
        // |an - aj| / an > eps  (src/path_abundance_estimator.cpp:73-75), without the division
        const bool viol = (an >= kMinEmAbundance) & (fabs(an - a_mine) > eps * an);
        a_mine = an;
        const int any_viol = __any(viol);
        ++iters;
        if (!any_viol) {
            if (++conv == kMinEmConvIts) break;
        } else {
            conv = 0;
        }
    }

    // src/path_abundance_estimator.cpp:100-113
    double low = 0.0;
    const bool first_of_column = (lane % kLanesPerColumn) == 0;
    if (first_of_column && my_col < np) {
        double * out = args.abundances + args.col_off[p];
        if (a_mine < kMinEmAbundance) {
            low = a_mine * T;
            out[my_col] = 0;
        } else {
            out[my_col] = a_mine * T;
        }
    }
    low = waveSumF64(low);
    if (first_of_column && my_col == np) {
        args.noise_count[p] = low + a_mine * T;
        args.iterations[p] = iters;
    }
}

// workgroup `block` of the ones that serve args.bin
template <int RPL, int COLS>
__device__ __forceinline__ void emRegisterBin(const EmLaunchArgs & args, const uint32_t block, double * reg_lds) {
    if (block >= args.queues->bin_count[args.bin]) return;  // more workgroups than the bin has problems
    // (RPVG_HIP_EM_COPIES=0 in the launch arguments: every problem with one copy, A/B and tests)
    for (uint32_t p = nextProblem<64>(args, nullptr); p != UINT32_MAX; p = nextProblem<64>(args, nullptr)) {
        if (RPL == 1 && COLS == 16 && args.register_copies) {
            const uint32_t n_rows = args.kept_rows[p];
            if (n_rows <= 16) emRegisterProblem<1, 16, 4>(args, p, reg_lds);
            else if (n_rows <= 32) emRegisterProblem<1, 16, 2>(args, p, reg_lds);
            else emRegisterProblem<1, 16, 1>(args, p, reg_lds);
        } else {
            emRegisterProblem<RPL, COLS>(args, p, reg_lds);
        }
        __syncthreads();  // the staging tile is reused
    }
}

// one launch per bin (RPVG_HIP_EM_REGISTER_LAUNCHES=5: A/B, and the per-bin device times of rpvg_hip_kernel_stats)
template <int RPL, int COLS>
__global__ __launch_bounds__(64) void emRegisterBinKernel(const EmLaunchArgs args) {
    extern __shared__ __attribute__((aligned(16))) double reg_lds[];
    emRegisterBin<RPL, COLS>(args, blockIdx.x, reg_lds);
}

// The register-resident bins in ONE launch: workgroup b serves bin kRegisterBins[b % variants] as that bin's workgroup
// b / variants (the variants side by side from the first workgroup on).  A launch lasts as long as its slowest problem iterates
// — a millisecond per bin on the configs[2] batch, on a few wavefronts —, and launches that share a stream, or, with batches
// in flight, a hardware queue with other contexts' streams, run one after the other: five bins in five launches were 4.1 ms
// of queue time per batch, in one launch they are 1.3.
constexpr uint32_t kRegisterBins[5] = {6, 4, 5, 8, 9};  // <4,16> <1,16> <2,16> <1,32> <2,32>: rpvg_hip_em_kernel_name
constexpr uint32_t kRegisterKernelIndex = 4;             // the launch's slot in rpvg_hip_kernel_stats::em_kernel

__global__ __launch_bounds__(64) void emRegisterKernel(const EmLaunchArgs launch_args, const uint32_t variants) {
    extern __shared__ __attribute__((aligned(16))) double reg_lds[];
    EmLaunchArgs args = launch_args;
    const uint32_t variant = blockIdx.x % variants, block = blockIdx.x / variants;
    args.bin = kRegisterBins[variant];
    switch (variant) {
        case 0: emRegisterBin<4, 16>(args, block, reg_lds); break;
        case 1: emRegisterBin<1, 16>(args, block, reg_lds); break;
        case 2: emRegisterBin<2, 16>(args, block, reg_lds); break;
        case 3: emRegisterBin<1, 32>(args, block, reg_lds); break;
        default: emRegisterBin<2, 32>(args, block, reg_lds); break;
    }
}

template <int RPL, int COLS>
hipError_t launchEmRegister(const EmLaunchArgs & args, uint32_t grid, hipStream_t stream) {
    if (grid == 0) return hipSuccess;
    const size_t lds = (64 * RPL * COLS + 2) * sizeof(double);  // the staging tile
    emRegisterBinKernel<RPL, COLS><<<dim3(grid), dim3(64), lds, stream>>>(args);
    return hipGetLastError();
}

// grid_per_bin workgroups for each of the bins (three without problems of more than 16 columns, else five)
hipError_t launchEmRegisterBins(const EmLaunchArgs & args, const uint32_t grid_per_bin, const bool with_32_columns, hipStream_t stream) {
    if (grid_per_bin == 0) return hipSuccess;
    const uint32_t variants = with_32_columns ? 5u : 3u;
    const size_t lds = (64 * 4 * 16 + 2) * sizeof(double);  // the largest staging tile (4 x 16 = 2 x 32)
    emRegisterKernel<<<dim3(grid_per_bin * variants), dim3(64), lds, stream>>>(args, variants);
    return hipGetLastError();
}

bool oneRegisterLaunch() {
    static const bool one = []() {
        const char * env = std::getenv("RPVG_HIP_EM_REGISTER_LAUNCHES");
        return !(env && std::atoi(env) == 5);
    }();
    return one;
}


// ---- Gibbs read-count sampler ----------------------------------------------------------
//
// gibbsReadCountSampler (src/path_abundance_estimator.cpp:116-212) for a batch of problems: per Gibbs
// iteration every row's reads are split multinomially over its columns with probabilities
// P_ij a_j / s_i (the reference draws the multinomial as a chain of binomials, :149-178), then every
// component draws a_j ~ Gamma(count_j + gamma, 1) and the vector is renormalised (:182-190); every
// `thin`-th state is recorded (:192-210).  ONE workgroup per problem runs all iterations.  The
// reference's mt19937 / libstdc++ distribution streams cannot be reproduced on a GPU:
// draws come from the counter-based Philox4x32-10 generator keyed by the problem's seed, so parity with
// the reference is statistical.  Rows without any selected path put all their reads on the noise
// component (their posterior there is exactly 1), as in the EM kernel.

struct Philox {
    uint32_t key[2];
    uint32_t ctr[4];
    uint32_t out[4];
    int have;

    __device__ __forceinline__ void init(const uint64_t seed, const uint32_t stream_hi, const uint32_t stream_lo) {
        key[0] = static_cast<uint32_t>(seed);
        key[1] = static_cast<uint32_t>(seed >> 32);
        ctr[0] = 0;
        ctr[1] = 0;
        ctr[2] = stream_lo;
        ctr[3] = stream_hi;
        have = 0;
    }

    __device__ __forceinline__ void round(uint32_t (&c)[4], const uint32_t k0, const uint32_t k1) {
        const uint64_t p0 = static_cast<uint64_t>(0xD2511F53u) * c[0];
        const uint64_t p1 = static_cast<uint64_t>(0xCD9E8D57u) * c[2];
        const uint32_t n0 = static_cast<uint32_t>(p1 >> 32) ^ c[1] ^ k0;
        const uint32_t n1 = static_cast<uint32_t>(p1);
        const uint32_t n2 = static_cast<uint32_t>(p0 >> 32) ^ c[3] ^ k1;
        const uint32_t n3 = static_cast<uint32_t>(p0);
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    }

    __device__ __forceinline__ void refill() {
        uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
        uint32_t k0 = key[0], k1 = key[1];
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            round(c, k0, k1);
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
        if (++ctr[0] == 0) ++ctr[1];
        have = 4;
    }

    __device__ __forceinline__ uint32_t next() {
        if (have == 0) refill();
        return out[--have];
    }

    // uniform in (0, 1)
    __device__ __forceinline__ double uniform() {
        const uint64_t hi = next(), lo = next();
        return (static_cast<double>(((hi << 32) | lo) >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    }

    __device__ __forceinline__ double normal() {
        const double u1 = uniform(), u2 = uniform();
        return sqrt(-2.0 * log(u1)) * cospi(2.0 * u2);
    }
};

// Binomial(n, p) by inversion: from 0 when the mean is small, otherwise outwards from the mode (expected
// O(sqrt(n p q)) steps; exact up to floating point).
__device__ uint32_t sampleBinomial(Philox & rng, const uint32_t n, double p) {
    if (n == 0 || !(p > 0.0)) return 0;
    if (p >= 1.0) return n;
    const bool flip = p > 0.5;
    if (flip) p = 1.0 - p;
    const double q = 1.0 - p, ratio = p / q;
    uint32_t k;
    if (n * p < 16.0) {
        double pmf = exp(n * log(q));
        double u = rng.uniform();
        k = 0;
        while (u > pmf && k < n) {
            u -= pmf;
            pmf *= ratio * (static_cast<double>(n - k) / (k + 1.0));
            ++k;
        }
    } else {
        const uint32_t mode = static_cast<uint32_t>((n + 1.0) * p);
        const double log_pmf_mode = lgamma(n + 1.0) - lgamma(mode + 1.0) - lgamma(n - mode + 1.0) + mode * log(p) + (n - mode) * log(q);
        const double pmf_mode = exp(log_pmf_mode);
        double u = rng.uniform();
        // walk outwards from the mode, alternating sides, until the accumulated mass passes u
        double up = pmf_mode, down = pmf_mode;
        uint32_t ku = mode, kd = mode;
        k = mode;
        if (u > pmf_mode) {
            u -= pmf_mode;
            while (true) {
                bool moved = false;
                if (ku < n) {
                    up *= ratio * (static_cast<double>(n - ku) / (ku + 1.0));
                    ++ku;
                    moved = true;
                    if (u <= up) { k = ku; break; }
                    u -= up;
                }
                if (kd > 0) {
                    down *= (static_cast<double>(kd) / (n - kd + 1.0)) / ratio;
                    --kd;
                    moved = true;
                    if (u <= down) { k = kd; break; }
                    u -= down;
                }
                if (!moved) { k = mode; break; }
            }
        }
    }
    return flip ? n - k : k;
}

// Gamma(shape >= 1, 1) by Marsaglia and Tsang's squeeze method.
__device__ double sampleGamma(Philox & rng, const double shape) {
    const double d = shape - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    while (true) {
        const double x = rng.normal();
        double v = 1.0 + c * x;
        if (v <= 0.0) continue;
        v = v * v * v;
        const double u = rng.uniform();
        if (log(u) < 0.5 * x * x + d - d * v + d * log(v)) return d * v;
    }
}

struct GibbsLaunchArgs {
    uint32_t count;
    const uint64_t * col_off;
    const uint64_t * row_base;
    const uint64_t * ent_base;
    const uint32_t * kept_rows;
    const double * zero_mass;
    const double * total_mass;
    const uint32_t * prow_off;
    const double * prow_count;
    const double * prow_noise;
    const uint32_t * pent_col;
    const double * pent_val;
    const double * init_abundances;   // [col_off[P]] expected counts (EM result)
    const double * init_noise_count;  // [P]
    const uint32_t * num_samples;     // [P]
    const uint64_t * seed;            // [P]
    const uint64_t * sample_off;      // [P+1]
    const uint64_t * abund_sample_off;  // [P+1] prefix of num_samples * columns
    uint32_t thin;
    double gamma;
    double * noise_samples;
    double * abundance_samples;
};

constexpr double kMinGibbsAbundance = 1e-8;  // src/path_abundance_estimator.cpp:14

__global__ __launch_bounds__(256) void gibbsReadCountKernel(const GibbsLaunchArgs args) {
    constexpr int BLOCK = 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const uint32_t p = blockIdx.x;
    if (p >= args.count) return;
    const uint32_t n_samples = args.num_samples[p];
    if (n_samples == 0) return;
    const uint32_t C = static_cast<uint32_t>(args.col_off[p + 1] - args.col_off[p]) + 1;
    const uint32_t noise_col = C - 1;
    double * a = reinterpret_cast<double *>(smem_raw);          // [C]
    double * red = a + C;                                       // [BLOCK/64 + 2]
    unsigned long long * counts = reinterpret_cast<unsigned long long *>(red + (BLOCK / 64 + 2));  // [C]

    const uint32_t n_rows = args.kept_rows[p];
    const uint64_t rb = args.row_base[p], eb = args.ent_base[p];
    const uint32_t * off = args.prow_off + rb + p;
    const double * cnt = args.prow_count + rb;
    const double * nzv = args.prow_noise + rb;
    const uint32_t * col = args.pent_col + eb;
    const double * val = args.pent_val + eb;
    const double T = args.total_mass[p];
    const unsigned long long Z = static_cast<unsigned long long>(args.zero_mass[p]);

    // start from the EM estimate (:128-136)
    for (uint32_t j = threadIdx.x; j < C; j += BLOCK) {
        a[j] = (j == noise_col ? args.init_noise_count[p] : args.init_abundances[args.col_off[p] + j]) / T;
    }
    __syncthreads();

    Philox rng;
    rng.init(args.seed[p], p, threadIdx.x);

    double * noise_out = args.noise_samples + args.sample_off[p];
    double * abund_out = args.abundance_samples + args.abund_sample_off[p];
    const uint32_t num_its = n_samples * args.thin;
    uint32_t recorded = 0;

    for (uint32_t it = 1; it <= num_its; ++it) {
        for (uint32_t j = threadIdx.x; j < C; j += BLOCK) counts[j] = (j == noise_col) ? Z : 0ull;
        __syncthreads();
        const double a_noise = a[noise_col];
        for (uint32_t r = threadIdx.x; r < n_rows; r += BLOCK) {
            const uint32_t e0 = off[r], e1 = off[r + 1];
            const double nz = nzv[r];
            double s = nz * a_noise;
            for (uint32_t e = e0; e < e1; ++e) s += val[e] * a[col[e]];
            uint32_t remaining = static_cast<uint32_t>(cnt[r]);
            double remaining_prob = 1.0;
            for (uint32_t e = e0; e < e1 && remaining > 0; ++e) {
                const double prob = val[e] * a[col[e]] / s;
                if (prob > 0.0) {
                    const uint32_t drawn = sampleBinomial(rng, remaining, fmin(1.0, prob / remaining_prob));
                    if (drawn) atomicAdd(&counts[col[e]], static_cast<unsigned long long>(drawn));
                    remaining -= drawn;
                }
                remaining_prob -= prob;
            }
            if (remaining) atomicAdd(&counts[noise_col], static_cast<unsigned long long>(remaining));
        }
        __syncthreads();
        double local = 0.0;
        for (uint32_t j = threadIdx.x; j < C; j += BLOCK) {
            const double g = sampleGamma(rng, static_cast<double>(counts[j]) + args.gamma);
            a[j] = g;
            local += g;
        }
        const double total = blockReduceSum<double, BLOCK>(local, red);
        __syncthreads();
        for (uint32_t j = threadIdx.x; j < C; j += BLOCK) a[j] = a[j] / total;
        __syncthreads();
        if (it % args.thin == 0) {
            double low = 0.0;
            for (uint32_t j = threadIdx.x; j < noise_col; j += BLOCK) {
                const double aj = a[j];
                if (aj < kMinGibbsAbundance) {
                    low += aj * T;
                    abund_out[static_cast<uint64_t>(recorded) * noise_col + j] = 0.0;
                } else {
                    abund_out[static_cast<uint64_t>(recorded) * noise_col + j] = aj * T;
                }
            }
            low = blockReduceSum<double, BLOCK>(low, red);
            if (threadIdx.x == 0) noise_out[recorded] = low + a[noise_col] * T;
            ++recorded;
            __syncthreads();
        }
    }
}

// ---- shared host part: the compacted CSR of a list of problems, their work queues, the EM launches -------------

EmBinRule emBinRule() {
    static const bool use_register_kernel = RPVG_EXPERIMENT_ENV("RPVG_HIP_NO_REGISTER_EM") == nullptr;
    // A streamed problem is one workgroup: above this many rows + entries it gets 1 024 threads instead of 256 (round 2:
    // 262 144 — a 200 000-entry problem on 256 threads took 47 us per EM iteration and, at 23 iterations, as long as the
    // thousands of iterations of the slowest register-resident problem; round 3: 24 576, then 0 — a batch has a few dozen
    // streamed problems, far fewer than CUs, and on 256 threads the ones below the limit took 55 us per iteration, twice
    // what the larger ones above it took on 1 024).
    static const uint64_t streamed_small = RPVG_EXPERIMENT_ENV("RPVG_HIP_EM_STREAM_SMALL") ? std::strtoull(RPVG_EXPERIMENT_ENV("RPVG_HIP_EM_STREAM_SMALL"), nullptr, 10) : 0;
    return EmBinRule{use_register_kernel ? 1u : 0u, streamed_small, emGridMinWork()};
}

}  // namespace

namespace rpvg_hip_detail {

size_t emQueuesBytes() { return sizeof(EmQueues); }
uint32_t emFillSegmentRows() { return kFillSegmentRows; }

// Everything between a problem list in device memory and its EM results, queued on the context's streams without a host
// synchronisation: compaction of every problem's rows (fillSegmentsKernel), the work queues (emOrderKernel), one
// persistent launch per kernel variant.  Caller holds ctx->mutex and has set the device; `work` must outlive the kernels.
int queueEmSolve(rpvg_hip_ctx * ctx, const rpvg_hip_batch * batch, const EmProblemList & list, const uint32_t max_em_its,
                 const double max_rel_em_conv, const EmOutputs & out, EmSolveWork & work, const bool fill_only, const double collapse_precision) {
    hipStream_t st = ctx->stream;
    const uint32_t P = list.P_bound;
    const EmBinRule rule = emBinRule();
    static const bool no_em_collapse = RPVG_EXPERIMENT_ENV("RPVG_HIP_NO_EM_COLLAPSE") != nullptr;
    const bool collapse = collapse_precision > 0 && !no_em_collapse && !RPVG_EXPERIMENT_ENV("RPVG_HIP_NO_COLLAPSE") && list.rows_capacity > 0;
    // (the few mid-size problems that may take the grid route: only where the grid route exists at all)
    const bool mid_grid_allowed = rule.grid_min_work > (1ull << kEmMidGridLog2);
    // The grid bin (problems too large for one workgroup, em_grid.hip): the host has to see them.  Only a solve that
    // sits on a cluster large enough to produce one pays for the look (two small copies and their waits).
    const bool grid_possible = rule.grid_min_work != 0 && list.max_cluster_work >= (mid_grid_allowed ? (1ull << kEmMidGridLog2) - 1 : rule.grid_min_work);
    std::unique_ptr<HostScope> stage_scope(new HostScope("em_solve: allocations + fill queued"));
    RPVG_HIP_CHECK(work.d_prow_off.alloc(list.rows_capacity + P));
    RPVG_HIP_CHECK(work.d_prow_count.alloc(list.rows_capacity));
    RPVG_HIP_CHECK(work.d_prow_noise.alloc(list.rows_capacity));
    RPVG_HIP_CHECK(work.d_pent_col.alloc(list.entries_capacity));
    RPVG_HIP_CHECK(work.d_pent_val.alloc(list.entries_capacity));
    RPVG_HIP_CHECK(work.d_zero.alloc(P));
    RPVG_HIP_CHECK(work.d_bucket.alloc(P));
    RPVG_HIP_CHECK(work.d_order.alloc(P));
    if (!work.zeroed_queues) RPVG_HIP_CHECK(work.d_queues.alloc(sizeof(EmQueues)));
    if (list.wide_capacity > 0) {
        RPVG_HIP_CHECK(work.d_wide_vectors.alloc(list.wide_capacity));
        RPVG_HIP_CHECK(work.d_wide_off.alloc(P));
    }
    EmQueues * queues = reinterpret_cast<EmQueues *>(work.zeroed_queues ? work.zeroed_queues : work.d_queues.ptr);

    int span = ctx->spanBegin(FAM_BUILD);
    if (!work.zeroed_queues) RPVG_HIP_CHECK(zeroAsync(work.d_queues.ptr, sizeof(EmQueues), st));
    RPVG_HIP_CHECK(work.d_seg_rows.alloc(list.items_bound));
    RPVG_HIP_CHECK(work.d_seg_entries.alloc(list.items_bound));
    RPVG_HIP_CHECK(work.d_seg_zero.alloc(list.items_bound));
    RPVG_HIP_CHECK(work.d_seg_total.alloc(list.items_bound));
    FillArgs fa;
    fa.num_problems = P;
    fa.num_problems_dev = list.d_num_problems;
    fa.num_items = list.items_bound;
    fa.num_items_dev = list.d_num_items;
    fa.seg_first = list.d_seg_first;
    fa.item_problem = list.d_item_problem;
    fa.prob_cluster = list.d_cluster;
    fa.col_off = list.d_col_off;
    fa.col_path = list.d_col_path;
    fa.cluster_row_off = batch->cluster_row_off.ptr;
    fa.cluster_path_off = batch->cluster_path_off.ptr;
    fa.row_ent_off = batch->row_ent_off.ptr;
    fa.ent_path = batch->ent_path.ptr;
    fa.ent_prob = batch->ent_prob.ptr;
    fa.row_count = batch->row_count.ptr;
    fa.row_noise = batch->row_noise.ptr;
    fa.row_base = list.d_row_base;
    fa.ent_base = list.d_ent_base;
    fa.prow_off = work.d_prow_off.ptr;
    fa.prow_count = work.d_prow_count.ptr;
    fa.prow_noise = work.d_prow_noise.ptr;
    fa.pent_col = work.d_pent_col.ptr;
    fa.pent_val = work.d_pent_val.ptr;
    fa.seg_rows = work.d_seg_rows.ptr;
    fa.seg_entries = work.d_seg_entries.ptr;
    fa.seg_zero_mass = work.d_seg_zero.ptr;
    fa.seg_total_mass = work.d_seg_total.ptr;
    fa.kept_rows = out.d_kept_rows;
    fa.kept_entries = out.d_kept_entries;
    fa.zero_mass = work.d_zero.ptr;
    fa.total_mass = out.d_total;
    fa.prob_bucket = work.d_bucket.ptr;
    fa.queues = queues;
    fa.rule = rule;
    fa.lds_map_paths = std::min<uint32_t>(list.max_cluster_paths, kLdsMapPaths);
    fa.num_fused = 0;
    // (the segment kernels walk the items with a grid of a few workgroups per CU: an item is at most 1 024 rows)
    const uint32_t fill_grid = std::min<uint32_t>(list.items_bound, static_cast<uint32_t>(ctx->props.multiProcessorCount) * 8);
    RPVG_HIP_CHECK(launchFillSegments<false>(fa, fill_grid, list.max_cluster_work, st));
    fillOffsetsKernel<<<dim3((P + 255) / 256), dim3(256), 0, st>>>(fa);
    // The fused build: a problem of the grid bin that will be solved on a dense matrix (em_grid.hip) gets it from the compaction
    // itself — rows -> matrix, 12 B read per entry and 8 B written per cell, against rows -> CSR -> zeroed matrix -> matrix (the
    // 1 M x 2 000 cluster of BASELINE.json configs[1]: 36 ms of a 160 ms call).  The host has to see the counts for it (one small
    // copy and its wait, only in a solve that sits on a cluster large enough for the grid bin); a solve whose problems are
    // collapsed (row_collapse.hip reads the CSR) keeps the CSR.  RPVG_HIP_NO_FUSED_DENSE=1: never (read per call; the tests take both).
    if (grid_possible && !collapse && !fill_only && !std::getenv("RPVG_HIP_NO_FUSED_DENSE")) {
        DeviceBuffer<DenseCandidates> d_candidates;
        DenseCandidates * h_candidates = nullptr;
        RPVG_HIP_CHECK(d_candidates.alloc(1));
        if (pinnedAlloc(reinterpret_cast<void **>(&h_candidates), sizeof(DenseCandidates)) != hipSuccess) {
            setError("rpvg_hip_em_solve: out of page-locked host memory");
            return RPVG_HIP_ERR_ALLOC;
        }
        struct PinnedGuard {
            void * p;
            ~PinnedGuard() { pinnedFree(p); }
        } pinned_guard{h_candidates};
        RPVG_HIP_CHECK(zeroAsync(d_candidates.ptr, sizeof(DenseCandidates), st));
        denseCandidatesKernel<<<dim3((P + 255) / 256), dim3(256), 0, st>>>(fa, d_candidates.ptr);
        RPVG_HIP_CHECK(hipGetLastError());
        RPVG_HIP_CHECK(hipMemcpyAsync(h_candidates, d_candidates.ptr, sizeof(DenseCandidates), hipMemcpyDeviceToHost, st));
        {
            HostScope wait_scope("em_solve: the counts of the large problems");
            RPVG_HIP_CHECK(waitStream(st));
        }
        const uint32_t n = std::min<uint32_t>(h_candidates->count, kEmMaxFusedDense);
        for (uint32_t i = 0; i < n; ++i) {
            const DenseCandidate & c = h_candidates->list[i];
            if (!emGridDenseRoute(c.columns, c.rows, c.entries)) continue;
            const uint64_t ld = (static_cast<uint64_t>(c.columns) + 1) & ~1ull;
            DeviceBuffer<double> & m = work.fused_matrix[work.num_fused];
            // (without the memory for it the problem takes the CSR, and the grid route decides again)
            if (m.alloc(static_cast<size_t>(c.rows) * ld) != hipSuccess) {
                (void) hipGetLastError();
                continue;
            }
            work.fused[work.num_fused] = EmFusedDense{c.problem, 0u, m.ptr, ld};
            fa.fused[work.num_fused] = work.fused[work.num_fused];
            ++work.num_fused;
        }
        fa.num_fused = work.num_fused;
    }
    RPVG_HIP_CHECK(launchFillSegments<true>(fa, fill_grid, list.max_cluster_work, st));
    if (fa.num_fused > 0) {
        uint64_t widest = 0;
        for (uint32_t f = 0; f < fa.num_fused; ++f) widest = std::max(widest, fa.fused[f].ld);
        const size_t lds = fa.lds_map_paths * sizeof(int32_t) + kFillDenseLds + 4 * widest * sizeof(double);
        if (lds > 64 * 1024) RPVG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&fillDenseRowsKernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
        const uint32_t dense_grid = std::min<uint32_t>(list.items_bound, static_cast<uint32_t>(ctx->props.multiProcessorCount) * 4);
        fillDenseRowsKernel<<<dim3(dense_grid, fa.num_fused), dim3(256), lds, st>>>(fa);
        RPVG_HIP_CHECK(hipGetLastError());
        ctx->stats.build_launches += 1;
    }
    RPVG_HIP_CHECK(hipGetLastError());
    ctx->stats.build_launches += 3;
    if (fill_only) {
        ctx->spanEnd(span);
        return RPVG_HIP_OK;
    }
    // readCollapseProbabilityMatrix on the rows of every problem (src/path_abundance_estimator.cpp:266,668), on the collapse
    // stream; the EM kernels wait for it and read the merged counts of the problems it merged rows in (rowCounts).  (A first
    // version solved every problem next to the collapse and the merged ones a second time: on the configs[2] batch the
    // largest problems were the merged ones, the second pass took as long as the first, and the collapse's forty launches
    // took 2.6 ms in between the persistent EM kernels against 1.5 ms without them.)
    // (the collapse indexes rows with 32 bits and matrices with 20; the callers' memory budgets keep a solve far below both —
    // a solve that is not gets an error rather than results without the collapse)
    RPVG_REQUIRE(!collapse || (list.rows_capacity <= 0x7fffffffull && P + 1 < kCollapseMaxMatrices),
                 "EM solve with a row collapse: %llu row slots in %u problems exceed what one solve can collapse (2^31 - 1 rows, 2^20 - 2 problems): split the problem list",
                 static_cast<unsigned long long>(list.rows_capacity), P);
    if (collapse) {
        RPVG_HIP_CHECK(hipEventCreateWithFlags(&work.filled, hipEventDisableTiming));
        RPVG_HIP_CHECK(hipEventRecord(work.filled, st));
    }
    emOrderKernel<<<dim3((P + 255) / 256), dim3(256), 0, st>>>(P, list.d_num_problems, work.d_bucket.ptr, list.d_col_off, queues, work.d_order.ptr,
                                                              work.d_wide_off.ptr, list.wide_capacity, mid_grid_allowed ? 1u : 0u);
    RPVG_HIP_CHECK(hipGetLastError());
    ctx->spanEnd(span);
    ctx->stats.build_launches += 1;
    if (collapse) {
        auto cw = std::make_shared<CsrCollapseWork>();
        work.collapse = cw;
        RPVG_HIP_CHECK(hipEventCreateWithFlags(&work.collapsed, hipEventDisableTiming));
        RPVG_HIP_CHECK(hipStreamWaitEvent(ctx->collapse_stream, work.filled, 0));
        CsrCollapseInput in;
        in.num_problems_bound = P;
        in.num_problems_dev = list.d_num_problems;
        in.rows_capacity = list.rows_capacity;
        in.row_base = list.d_row_base;
        in.ent_base = list.d_ent_base;
        in.kept_rows = out.d_kept_rows;
        in.col_off = list.d_col_off;
        in.prow_off = work.d_prow_off.ptr;
        in.prow_count = work.d_prow_count.ptr;
        in.prow_noise = work.d_prow_noise.ptr;
        in.pent_col = work.d_pent_col.ptr;
        in.pent_val = work.d_pent_val.ptr;
        in.num_items_bound = list.items_bound;
        in.num_items_dev = list.d_num_items;
        in.seg_first = list.d_seg_first;
        in.item_problem = list.d_item_problem;
        in.segment_rows = kFillSegmentRows;
        in.max_rows_bound = std::min<uint64_t>(list.max_cluster_work, list.rows_capacity);  // (rows + entries of the largest cluster: a bound of its rows)
        const int collapse_span = ctx->spanBegin(FAM_COLLAPSE, ctx->collapse_stream);
        RPVG_HIP_CHECK(hipEventCreateWithFlags(&work.collapse_sorted, hipEventDisableTiming));
        stage_scope.reset(new HostScope("em_solve: the problems' collapse queued"));
        RPVG_HIP_CHECK(queueCsrCollapse(ctx, in, collapse_precision, *cw, ctx->collapse_stream, work.collapse_sorted));
        stage_scope.reset(new HostScope("em_solve: EM launches, grid problems, join"));
        ctx->spanEnd(collapse_span);
        RPVG_HIP_CHECK(hipEventRecord(work.collapsed, ctx->collapse_stream));
    }

    EmLaunchArgs args;
    args.order = work.d_order.ptr;
    args.queues = queues;
    args.bin = 0;
    args.col_off = list.d_col_off;
    args.row_base = list.d_row_base;
    args.ent_base = list.d_ent_base;
    args.kept_rows = out.d_kept_rows;
    args.zero_mass = work.d_zero.ptr;
    args.total_mass = out.d_total;
    args.prow_off = work.d_prow_off.ptr;
    args.prow_count = work.d_prow_count.ptr;
    args.prow_noise = work.d_prow_noise.ptr;
    args.pent_col = work.d_pent_col.ptr;
    args.pent_val = work.d_pent_val.ptr;
    args.max_em_its = max_em_its;
    args.max_rel_em_conv = max_rel_em_conv;
    {
        const char * env = std::getenv("RPVG_HIP_EM_COPIES");  // (read per call: the tests take both ways)
        args.register_copies = env ? (std::atoi(env) != 0 ? 1u : 0u) : 1u;
    }
    args.wide_vectors = work.d_wide_vectors.ptr;
    args.wide_off = work.d_wide_off.ptr;
    args.wide_capacity = list.wide_capacity;
    args.abundances = out.d_abundances;
    args.noise_count = out.d_noise_count;
    args.iterations = out.d_iterations;
    args.merged_count = nullptr;
    args.problem_merged = nullptr;
    if (collapse) {
        const CsrCollapseWork * cw = static_cast<const CsrCollapseWork *>(work.collapse.get());
        args.merged_count = cw->merged_count.ptr;
        args.problem_merged = cw->problem_merged.ptr;
        RPVG_HIP_CHECK(hipStreamWaitEvent(st, work.collapsed, 0));
        // The EM launches go to streams of their own, and a stream whose next command waits for an event holds its hardware
        // queue meanwhile — eight queues for every stream of both lanes (hardwareQueues()).  Queued at once, ten launches that
        // wait for the collapse parked most of the queues for its whole millisecond and the other lane's kernels stood behind
        // them (measured: the library's segmented sort, whose host wait for its partition sizes delayed these launches by
        // accident, beat every sort without such a wait by 1 ms per batch).  So the submitting thread waits for the collapse
        // itself and queues the launches then.
        // RPVG_HIP_EM_LAUNCH_EARLY=1: queue them at once (A/B).
        static const bool launch_early = RPVG_EXPERIMENT_ENV("RPVG_HIP_EM_LAUNCH_EARLY") != nullptr;
        static const bool wait_whole = RPVG_EXPERIMENT_ENV("RPVG_HIP_EM_WAIT_SORT_ONLY") == nullptr;  // A/B: only the collapse's sort (11.3 against 10.2 ms per batch)
        if (!launch_early && work.collapse_sorted) {
            HostScope wait_scope("em_solve: wait for the collapse");
            RPVG_HIP_CHECK(waitEvent(wait_whole ? work.collapsed : work.collapse_sorted));
        }
    }

    // The bins are independent, so their tails (a small problem that needs thousands of iterations, a giant one with
    // many rows) should overlap — but only as many kernels run side by side as the runtime has hardware queues.
    // Workgroups of a persistent launch: one or two per CU (or the bound on the problems if smaller) — every workgroup of a
    // grid costs the dispatcher ~40 ns even if it finds its bin empty, and the host launches all variants blindly: grids
    // sized by what the GPU could hold (4 096 waves for the register kernel) were 20 000 idle workgroups per call, 0.7 ms
    // of dispatcher time next to the other lane's kernels.  A queue of 1 200 short problems drains through 512 waves in
    // tens of microseconds; the problems that run for thousands of iterations start that much later at the most.
    const uint32_t cus = static_cast<uint32_t>(ctx->props.multiProcessorCount);
    static const double grid_scale = RPVG_EXPERIMENT_ENV("RPVG_HIP_EM_GRID_SCALE") ? std::atof(RPVG_EXPERIMENT_ENV("RPVG_HIP_EM_GRID_SCALE")) : 1.0;  // A/B knob
    auto grid = [&](const uint32_t per_cu) { return std::min<uint32_t>(P, std::max<uint32_t>(1, static_cast<uint32_t>(cus * per_cu * grid_scale))); };
    const size_t streamed_lds_256 = emLdsBytes(list.max_cols, 0, 0, 256, false), streamed_lds_1024 = emLdsBytes(list.max_cols, 0, 0, 1024, false);
    const bool wide_possible = streamed_lds_256 > kEmLdsLimit;
    static const bool few_streams = RPVG_EXPERIMENT_ENV("RPVG_HIP_EM_FEW_STREAMS") != nullptr;  // A/B knob
    const bool many_queues = hardwareQueues() >= 8 && !few_streams;
    hipStream_t s_reg4 = many_queues ? ctx->aux[3] : ctx->aux[0], s_reg1 = many_queues ? ctx->aux[4] : ctx->aux[1], s_reg2 = many_queues ? ctx->aux[5] : ctx->aux[2];
    // One persistent launch per kernel variant (the register-resident bins are the long ones: they start first; with
    // eight hardware queues — hardwareQueues(), context.hip — they get streams of their own; chains of launches that share
    // a stream run one after the other: balanced by the kernels' usual durations).  with_spans: every launch carries its
    // own HIP events on its own stream (rpvg_hip_kernel_stats::em_kernel).
    // (the default) the register-resident bins in one launch, the others balanced over the side streams by their usual durations:
    // with six side streams everybody has a stream of its own, with three (the contexts of the batch pipeline) the launch of the
    // register bins (1.3 ms on the configs[2] batch) and <1024,true> (0.2) share one, <64,true> (0.9) and <256,true> (0.5)
    // another, and <1024,false> (0.8) and the wide one have the third
    auto launchVariantsOneRegisterLaunch = [&](auto & timed, int & bin_span) -> int {
        const bool own_streams = ctx->aux_count >= rpvg_hip_ctx::kAuxStreams;
        hipStream_t s_register = own_streams ? ctx->aux[3] : ctx->aux[0];
        hipStream_t s_1024_resident = own_streams ? ctx->aux[4] : ctx->aux[0], s_64 = ctx->aux[1], s_256 = own_streams ? ctx->aux[5] : ctx->aux[1];
        hipStream_t s_1024 = own_streams ? ctx->aux[0] : ctx->aux[2], s_wide = ctx->aux[2];
        timed(static_cast<int>(kRegisterKernelIndex), s_register);
        RPVG_HIP_CHECK(launchEmRegisterBins(args, grid(2), list.max_cols > 16, s_register));
        ctx->spanEnd(bin_span);
        timed(2, st);
        RPVG_HIP_CHECK((launchEm<256, false>(args, grid(1), std::min(streamed_lds_256, kEmLdsLimit), st)));
        ctx->spanEnd(bin_span);
        timed(3, s_1024);
        RPVG_HIP_CHECK((launchEm<1024, false>(args, grid(1), std::min(streamed_lds_1024, kEmLdsLimit), s_1024)));
        ctx->spanEnd(bin_span);
        timed(0, s_64);
        RPVG_HIP_CHECK((launchEm<64, true>(args, grid(2), 8 * 1024, s_64)));
        ctx->spanEnd(bin_span);
        timed(1, s_256);
        RPVG_HIP_CHECK((launchEm<256, true>(args, grid(2), 40 * 1024, s_256)));
        ctx->spanEnd(bin_span);
        timed(7, s_1024_resident);
        RPVG_HIP_CHECK((launchEm<1024, true>(args, grid(1), 152 * 1024, s_1024_resident)));
        ctx->spanEnd(bin_span);
        if (wide_possible) {
            timed(10, s_wide);
            RPVG_HIP_CHECK((launchEm<1024, false, true>(args, grid(1), sizeof(double) * (1024 / 64 + 2), s_wide)));
            ctx->spanEnd(bin_span);
        }
        return RPVG_HIP_OK;
    };
    auto launchVariants = [&](const bool with_spans) -> int {
        RPVG_HIP_CHECK(ctx->forkAux());
        int bin_span = -1;
        auto timed = [&](const int b, hipStream_t on) {
            args.bin = static_cast<uint32_t>(b);
            bin_span = with_spans ? ctx->spanBegin(FAM_EM_KERNEL, on, b) : -1;
            return on;
        };
        if (oneRegisterLaunch()) return launchVariantsOneRegisterLaunch(timed, bin_span);
        RPVG_HIP_CHECK((launchEmRegister<4, 16>(args, grid(2), timed(6, s_reg4))));
        ctx->spanEnd(bin_span);
        RPVG_HIP_CHECK((launchEmRegister<1, 16>(args, grid(2), timed(4, s_reg1))));
        ctx->spanEnd(bin_span);
        RPVG_HIP_CHECK((launchEmRegister<2, 16>(args, grid(2), timed(5, s_reg2))));
        ctx->spanEnd(bin_span);
        timed(2, st);
        RPVG_HIP_CHECK((launchEm<256, false>(args, grid(1), std::min(streamed_lds_256, kEmLdsLimit), st)));
        ctx->spanEnd(bin_span);
        timed(3, ctx->aux[0]);
        RPVG_HIP_CHECK((launchEm<1024, false>(args, grid(1), std::min(streamed_lds_1024, kEmLdsLimit), ctx->aux[0])));
        ctx->spanEnd(bin_span);
        timed(7, ctx->aux[0]);
        RPVG_HIP_CHECK((launchEm<1024, true>(args, grid(1), 152 * 1024, ctx->aux[0])));
        ctx->spanEnd(bin_span);
        timed(0, ctx->aux[1]);
        RPVG_HIP_CHECK((launchEm<64, true>(args, grid(2), 8 * 1024, ctx->aux[1])));
        ctx->spanEnd(bin_span);
        timed(1, ctx->aux[2]);
        RPVG_HIP_CHECK((launchEm<256, true>(args, grid(2), 40 * 1024, ctx->aux[2])));
        ctx->spanEnd(bin_span);
        if (list.max_cols > 16) {
            RPVG_HIP_CHECK((launchEmRegister<1, 32>(args, grid(1), timed(8, ctx->aux[2]))));
            ctx->spanEnd(bin_span);
            RPVG_HIP_CHECK((launchEmRegister<2, 32>(args, grid(1), timed(9, ctx->aux[2]))));
            ctx->spanEnd(bin_span);
        }
        if (wide_possible) {
            timed(10, s_reg2);
            RPVG_HIP_CHECK((launchEm<1024, false, true>(args, grid(1), sizeof(double) * (1024 / 64 + 2), s_reg2)));
            ctx->spanEnd(bin_span);
        }
        // (the side streams are joined behind the problems that run over the whole GPU, below: those start while these kernels run)
        return RPVG_HIP_OK;
    };
    DeviceBuffer<EmGridProblem> d_grid_problems;
    hipEvent_t described = nullptr;
    uint32_t * h_grid_count = nullptr;
    struct GridLookGuard {
        hipEvent_t & ev;
        uint32_t *& pinned;
        ~GridLookGuard() {
            if (ev) (void) hipEventDestroy(ev);
            if (pinned) pinnedFree(pinned);
        }
    } grid_look_guard{described, h_grid_count};
    if (grid_possible) {
        RPVG_HIP_CHECK(d_grid_problems.alloc(P));
        if (pinnedAlloc(reinterpret_cast<void **>(&h_grid_count), 64) != hipSuccess) {
            setError("rpvg_hip_em_solve: out of page-locked host memory");
            return RPVG_HIP_ERR_ALLOC;
        }
        GridDescribeArgs da;
        da.queues = queues;
        da.order = work.d_order.ptr;
        da.col_off = list.d_col_off;
        da.row_base = list.d_row_base;
        da.ent_base = list.d_ent_base;
        da.kept_rows = out.d_kept_rows;
        da.kept_entries = out.d_kept_entries;
        da.zero_mass = work.d_zero.ptr;
        da.total_mass = out.d_total;
        da.problem_merged = args.problem_merged;
        da.out = d_grid_problems.ptr;
        da.capacity = P;
        emGridDescribeKernel<<<dim3((P + 255) / 256), dim3(256), 0, st>>>(da);
        RPVG_HIP_CHECK(hipGetLastError());
        RPVG_HIP_CHECK(hipMemcpyAsync(h_grid_count, &queues->bin_count[kEmGridBin], sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        RPVG_HIP_CHECK(hipEventCreateWithFlags(&described, hipEventDisableTiming));
        RPVG_HIP_CHECK(hipEventRecord(described, st));
    }
    span = ctx->spanBegin(FAM_EM_SPARSE);
    {
        const int rc = launchVariants(true);
        if (rc != RPVG_HIP_OK) return rc;
    }
    if (grid_possible) {
        RPVG_HIP_CHECK(waitEvent(described));
        const uint32_t n_grid = std::min<uint32_t>(*h_grid_count, P);
        if (n_grid > 0) {
            std::vector<EmGridProblem> grid_problems(n_grid);
            RPVG_HIP_CHECK(hipMemcpyAsync(grid_problems.data(), d_grid_problems.ptr, sizeof(EmGridProblem) * n_grid, hipMemcpyDeviceToHost, st));
            RPVG_HIP_CHECK(waitStream(st));
            EmGridStorage storage;
            storage.prow_off = work.d_prow_off.ptr;
            storage.prow_count = work.d_prow_count.ptr;
            storage.merged_count = args.merged_count;
            storage.prow_noise = work.d_prow_noise.ptr;
            storage.pent_col = work.d_pent_col.ptr;
            storage.pent_val = work.d_pent_val.ptr;
            storage.abundances = out.d_abundances;
            storage.noise_count = out.d_noise_count;
            storage.iterations = out.d_iterations;
            storage.fused = work.fused;
            storage.num_fused = work.num_fused;
            const int rc = runEmGridProblems(ctx, st, grid_problems.data(), n_grid, storage, max_em_its, max_rel_em_conv);
            if (rc != RPVG_HIP_OK) return rc;
        }
    }
    {
        // (RPVG_HIP_EM_JOIN_ON_STREAM=1: the context's stream waits for the side streams, not the thread — A/B)
        static const bool join_on_stream = RPVG_EXPERIMENT_ENV("RPVG_HIP_EM_JOIN_ON_STREAM") != nullptr;
        RPVG_HIP_CHECK(join_on_stream ? ctx->joinAux() : ctx->joinAuxOnHost());
    }
    ctx->spanEnd(span);  // (behind the join: the span of the solve's kernels on all of its streams)
    if (collapse) {
        const CsrCollapseWork * cw = static_cast<const CsrCollapseWork *>(work.collapse.get());
        static const bool debug = RPVG_EXPERIMENT_ENV("RPVG_HIP_EM_COLLAPSE_DEBUG") != nullptr;
        if (debug) {  // (synchronises: a measuring aid)
            uint32_t info[6] = {0}, merged = 0, problems = P, counts[3] = {0};
            RPVG_HIP_CHECK(waitStream(st));
            RPVG_HIP_CHECK(hipMemcpy(info, cw->info.ptr, sizeof(info), hipMemcpyDeviceToHost));
            RPVG_HIP_CHECK(hipMemcpy(&merged, cw->problem_merged.ptr + P, sizeof(merged), hipMemcpyDeviceToHost));
            if (list.d_num_problems) RPVG_HIP_CHECK(hipMemcpy(&problems, list.d_num_problems, sizeof(problems), hipMemcpyDeviceToHost));
            {   // (layout of the zeroed words: queueCollapseStages)
                const uint64_t mark_words = (list.rows_capacity + 31) / 32;
                RPVG_HIP_CHECK(hipMemcpy(counts, cw->info.ptr + 6 + 2 + 2 * static_cast<uint64_t>(P) + 1 + mark_words, sizeof(counts), hipMemcpyDeviceToHost));
            }
            std::fprintf(stderr, "[em collapse] marked rows %u forward pairs %u (equal up to rounding %u, apart %u) around pairs %u\n", counts[0], counts[1], info[4], info[5], counts[2]);
            std::fprintf(stderr, "[em collapse] problems %u (bound %u) row slots %llu: replayed %u whole %u active rows %u rows replaced %u problems merged %u\n",
                         problems, P, static_cast<unsigned long long>(list.rows_capacity), info[0], info[2], info[3], info[1], merged);
        }
    }
    return RPVG_HIP_OK;
}

// Folds the results of a solve into the context's statistics (after the results have reached the host): the host
// repeats the device's bin decision per problem.  cols[p] = columns of problem p without the noise component.
void accountEmSolve(rpvg_hip_ctx * ctx, const uint32_t P, const uint64_t * col_off, const uint32_t * kept_rows, const uint32_t * kept_entries,
                    const uint32_t * iterations) {
    // algorithmic bytes: per iteration 12 B per entry (value + column), 20 B per row (count, noise, offset), 16 B per
    // column (a read + a' write)
    const EmBinRule rule = emBinRule();
    double bin_bytes[kEmBins] = {};
    uint64_t bin_its[kEmBins] = {}, bin_problems[kEmBins] = {};
    uint32_t bin_slowest[kEmBins] = {};
    // (emOrderKernel's verdict on the few mid-size problems, repeated)
    uint32_t mid_problems = 0;
    for (uint32_t p = 0; p < P; ++p) {
        const uint32_t C = static_cast<uint32_t>(col_off[p + 1] - col_off[p]) + 1;
        if (emBinOf(rule, C, kept_rows[p], kept_entries[p]) == kEmStreamedBin && emWorkBucket(kept_rows[p], kept_entries[p]) < kEmWorkBuckets - kEmMidGridLog2) ++mid_problems;
    }
    const bool mid_moved = rule.grid_min_work > (1ull << kEmMidGridLog2) && mid_problems > 0 && mid_problems <= kEmMidGridMax;
    for (uint32_t p = 0; p < P; ++p) {
        const uint32_t C = static_cast<uint32_t>(col_off[p + 1] - col_off[p]) + 1;
        int b = emBinOf(rule, C, kept_rows[p], kept_entries[p]);
        if (mid_moved && b == kEmStreamedBin && emWorkBucket(kept_rows[p], kept_entries[p]) < kEmWorkBuckets - kEmMidGridLog2) b = kEmGridBin;
        // (a problem of the grid bin on the dense route ran em_dense.hip's kernels: emDenseIterate has accounted for it)
        if (b == kEmGridBin && emGridDenseRoute(C, kept_rows[p], kept_entries[p])) {
            bin_problems[b] += 1;
            continue;
        }
        bin_bytes[b] += static_cast<double>(iterations[p]) * (12.0 * kept_entries[p] + 20.0 * kept_rows[p] + 16.0 * C);
        bin_its[b] += iterations[p];
        bin_problems[b] += 1;
        bin_slowest[b] = std::max(bin_slowest[b], iterations[p]);
    }
    // (the register-resident bins of one launch are one kernel of the statistics: their problems together, the slowest of all)
    double slot_bytes[kEmBins] = {};
    uint64_t slot_its[kEmBins] = {}, slot_problems[kEmBins] = {};
    uint32_t slot_slowest[kEmBins] = {};
    for (int b = 0; b < kEmBins; ++b) {
        const bool register_bin = b == 4 || b == 5 || b == 6 || b == 8 || b == 9;
        const int slot = register_bin && oneRegisterLaunch() ? static_cast<int>(kRegisterKernelIndex) : b;
        slot_bytes[slot] += bin_bytes[b];
        slot_its[slot] += bin_its[b];
        slot_problems[slot] += bin_problems[b];
        slot_slowest[slot] = std::max(slot_slowest[slot], bin_slowest[b]);
    }
    for (int b = 0; b < kEmBins; ++b) {
        if (!slot_problems[b]) continue;
        rpvg_hip_em_kernel_stats & ks = ctx->stats.em_kernel[b];
        ks.launches += 1;
        ks.problems += slot_problems[b];
        ks.iterations += slot_its[b];
        ks.max_iterations += slot_slowest[b];
        ks.alg_bytes += slot_bytes[b];
        ctx->stats.em_sparse_launches += 1;
        ctx->stats.em_sparse_alg_bytes += slot_bytes[b];
        ctx->stats.em_iterations_total += slot_its[b];
    }
}

}  // namespace rpvg_hip_detail

namespace {

// The problems of an rpvg_hip_em_solve / rpvg_hip_gibbs_read_counts call: validated, uploaded, and laid out.
struct HostProblemSet {
    EmProblemList list;
    EmSolveWork work;
    DeviceBuffer<uint32_t> d_cluster, d_col_path, d_item_problem;
    DeviceBuffer<uint64_t> d_col_off, d_row_base, d_ent_base, d_seg_first;
    UploadPack uploads;
    uint64_t n_cols_total = 0;
};

// Free device memory a call may plan with (a share of what the driver reported when the process first asked: other lanes
// allocate too, and hipMemGetInfo costs milliseconds — asked per call it was 3 ms of every batch's critical path).
uint64_t deviceMemoryBudget() {
    static const uint64_t budget = []() {
        size_t free_bytes = 0, total_bytes = 0;
        if (hipMemGetInfo(&free_bytes, &total_bytes) != hipSuccess) {
            (void) hipGetLastError();
            return static_cast<uint64_t>(4ull << 30);
        }
        return static_cast<uint64_t>(free_bytes) * 2 / 5;
    }();
    return budget;
}

// Caller holds ctx->mutex and has set the device.  Validates the problems and uploads their description; the storage of a
// problem starts where that of the problems before it ends at the most (a problem keeps at most the rows and entries of
// its cluster) — offsets the host knows, so one kernel counts and fills.  When that bound does not fit the memory the
// call may use (RPVG_HIP_EM_BOUND_BYTES, at most two fifths of the free device memory), a counting pass comes first and the
// storage is exact.
int prepareHostProblems(rpvg_hip_ctx * ctx, const rpvg_hip_batch * batch, const rpvg_hip_em_problems * problems,
                        HostProblemSet & ps, const EmOutputs & out, const char * who) {
    const uint32_t P = problems->num_problems;
    std::unique_ptr<HostScope> scope(new HostScope("problems: validate"));
    uint64_t rows_bound = 0, entries_bound = 0;
    std::vector<uint64_t> row_base(P), ent_base(P);
    EmProblemList & list = ps.list;
    for (uint32_t p = 0; p < P; ++p) {
        const uint32_t k = problems->cluster[p];
        RPVG_REQUIRE(k < batch->num_clusters, "%s: problem %u refers to cluster %u of %u", who, p, k, batch->num_clusters);
        const uint64_t n_paths = batch->h_cluster_path_off[k + 1] - batch->h_cluster_path_off[k];
        const uint64_t c0 = problems->col_off[p], c1 = problems->col_off[p + 1];
        RPVG_REQUIRE(c1 > c0, "%s: problem %u has no columns", who, p);
        RPVG_REQUIRE(batch->h_cluster_row_off[k + 1] > batch->h_cluster_row_off[k], "%s: problem %u is on cluster %u which has no rows", who, p, k);
        for (uint64_t c = c0; c < c1; ++c) {
            RPVG_REQUIRE(problems->col_path[c] < n_paths, "%s: problem %u column path %u >= %llu", who, p, problems->col_path[c],
                         static_cast<unsigned long long>(n_paths));
            RPVG_REQUIRE(c == c0 || problems->col_path[c] > problems->col_path[c - 1], "%s: problem %u columns are not strictly ascending", who, p);
        }
        const uint32_t C = static_cast<uint32_t>(c1 - c0) + 1;
        list.max_cols = std::max<uint32_t>(list.max_cols, C);
        list.max_cluster_paths = std::max<uint32_t>(list.max_cluster_paths, static_cast<uint32_t>(n_paths));
        list.max_cluster_work = std::max<uint64_t>(list.max_cluster_work, (batch->h_cluster_row_off[k + 1] - batch->h_cluster_row_off[k]) +
                                                                              (batch->h_cluster_ent_off[k + 1] - batch->h_cluster_ent_off[k]));
        if (emLdsBytes(C, 0, 0, 256, false) > kEmLdsLimit) list.wide_capacity += 2ull * C;
        row_base[p] = rows_bound;
        ent_base[p] = entries_bound;
        rows_bound += batch->h_cluster_row_off[k + 1] - batch->h_cluster_row_off[k];
        entries_bound += batch->h_cluster_ent_off[k + 1] - batch->h_cluster_ent_off[k];
    }
    ps.n_cols_total = problems->col_off[P];
    list.P_bound = P;
    // work items of the compaction: the rows of every problem's cluster in segments
    std::vector<uint64_t> seg_first(P + 1, 0);
    for (uint32_t p = 0; p < P; ++p) {
        const uint32_t k = problems->cluster[p];
        const uint64_t rows = batch->h_cluster_row_off[k + 1] - batch->h_cluster_row_off[k];
        seg_first[p + 1] = seg_first[p] + (rows + kFillSegmentRows - 1) / kFillSegmentRows;
    }
    RPVG_REQUIRE(seg_first[P] < 0xffffffffull, "%s: too many row segments (%llu)", who, static_cast<unsigned long long>(seg_first[P]));
    std::vector<uint32_t> item_problem(seg_first[P]);
    for (uint32_t p = 0; p < P; ++p) {
        for (uint64_t item = seg_first[p]; item < seg_first[p + 1]; ++item) item_problem[item] = p;
    }
    list.items_bound = static_cast<uint32_t>(seg_first[P]);

    scope.reset(new HostScope("problems: uploads"));
    hipStream_t st = ctx->stream;
    // (storage by the bound up to a budget — RPVG_HIP_EM_BOUND_BYTES, by default two fifths of the free device memory and at
    // most 32 GiB; beyond it two passes, the first of which only counts)
    const char * budget_env = std::getenv("RPVG_HIP_EM_BOUND_BYTES");  // (read per call: the tests take both ways)
    const uint64_t bound_budget = budget_env ? static_cast<uint64_t>(std::atof(budget_env)) : std::min<uint64_t>(32ull << 30, deviceMemoryBudget());
    const bool by_bound = rows_bound * 20 + entries_bound * 12 <= bound_budget;
    int span = ctx->spanBegin(FAM_H2D);
    ps.uploads.add(ps.d_cluster, problems->cluster, P);
    ps.uploads.add(ps.d_col_off, problems->col_off, P + 1);
    ps.uploads.add(ps.d_col_path, problems->col_path, ps.n_cols_total);
    ps.uploads.add(ps.d_seg_first, seg_first.data(), P + 1);
    ps.uploads.add(ps.d_item_problem, item_problem.data(), item_problem.size());
    if (by_bound) {
        ps.uploads.add(ps.d_row_base, row_base.data(), P);
        ps.uploads.add(ps.d_ent_base, ent_base.data(), P);
    }
    RPVG_HIP_CHECK(ps.uploads.commit(st));
    ctx->spanEnd(span);
    ctx->stats.h2d_bytes += static_cast<double>(P * 4 + (P + 1) * 24 + ps.n_cols_total * 4);
    list.d_cluster = ps.d_cluster.ptr;
    list.d_col_off = ps.d_col_off.ptr;
    list.d_col_path = ps.d_col_path.ptr;
    list.d_seg_first = ps.d_seg_first.ptr;
    list.d_item_problem = ps.d_item_problem.ptr;
    if (!by_bound) {
        scope.reset(new HostScope("problems: counting pass"));
        // counts only: no storage, no queues
        EmProblemList count_list = list;
        count_list.rows_capacity = count_list.entries_capacity = 0;
        FillArgs fa{};
        fa.num_problems = P;
        fa.num_items = list.items_bound;
        fa.seg_first = list.d_seg_first;
        fa.item_problem = list.d_item_problem;
        fa.prob_cluster = list.d_cluster;
        fa.col_off = list.d_col_off;
        fa.col_path = list.d_col_path;
        fa.cluster_row_off = batch->cluster_row_off.ptr;
        fa.cluster_path_off = batch->cluster_path_off.ptr;
        fa.row_ent_off = batch->row_ent_off.ptr;
        fa.ent_path = batch->ent_path.ptr;
        fa.ent_prob = batch->ent_prob.ptr;
        fa.row_count = batch->row_count.ptr;
        fa.row_noise = batch->row_noise.ptr;
        DeviceBuffer<double> d_zero, d_seg_zero, d_seg_total;
        DeviceBuffer<uint32_t> d_seg_rows, d_seg_entries;
        RPVG_HIP_CHECK(d_zero.alloc(P));
        RPVG_HIP_CHECK(d_seg_zero.alloc(list.items_bound));
        RPVG_HIP_CHECK(d_seg_total.alloc(list.items_bound));
        RPVG_HIP_CHECK(d_seg_rows.alloc(list.items_bound));
        RPVG_HIP_CHECK(d_seg_entries.alloc(list.items_bound));
        fa.seg_rows = d_seg_rows.ptr;
        fa.seg_entries = d_seg_entries.ptr;
        fa.seg_zero_mass = d_seg_zero.ptr;
        fa.seg_total_mass = d_seg_total.ptr;
        fa.kept_rows = out.d_kept_rows;
        fa.kept_entries = out.d_kept_entries;
        fa.zero_mass = d_zero.ptr;
        fa.total_mass = out.d_total;
        fa.rule = emBinRule();
        fa.lds_map_paths = std::min<uint32_t>(list.max_cluster_paths, kLdsMapPaths);
        const uint32_t fill_grid = std::min<uint32_t>(list.items_bound, static_cast<uint32_t>(ctx->props.multiProcessorCount) * 8);
        RPVG_HIP_CHECK(launchFillSegments<false>(fa, fill_grid, list.max_cluster_work, st));
        fillOffsetsKernel<<<dim3((P + 255) / 256), dim3(256), 0, st>>>(fa);
        RPVG_HIP_CHECK(hipGetLastError());
        std::vector<uint32_t> kept_rows(P), kept_ent(P);
        RPVG_HIP_CHECK(hipMemcpyAsync(kept_rows.data(), out.d_kept_rows, P * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        RPVG_HIP_CHECK(hipMemcpyAsync(kept_ent.data(), out.d_kept_entries, P * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        RPVG_HIP_CHECK(waitStream(st));
        rows_bound = entries_bound = 0;
        for (uint32_t p = 0; p < P; ++p) {
            row_base[p] = rows_bound;
            ent_base[p] = entries_bound;
            rows_bound += kept_rows[p];
            entries_bound += kept_ent[p];
        }
        RPVG_HIP_CHECK(ps.d_row_base.upload(row_base.data(), P, st));
        RPVG_HIP_CHECK(ps.d_ent_base.upload(ent_base.data(), P, st));
    }
    list.d_row_base = ps.d_row_base.ptr;
    list.d_ent_base = ps.d_ent_base.ptr;
    list.rows_capacity = rows_bound;
    list.entries_capacity = entries_bound;
    return RPVG_HIP_OK;
}

}  // namespace

extern "C" const char * rpvg_hip_em_kernel_name(int index) {
    // (the size bins of rpvg_hip_em_solve, in bin order)
    // (emRegisterKernel: the five register-resident bins — <1,16> <2,16> <4,16> <1,32> <2,32> — in one launch, under the first
    // one's index; with RPVG_HIP_EM_REGISTER_LAUNCHES=5 every bin has its own launch of emRegisterBinKernel and its own index)
    static const char * const names[RPVG_HIP_EM_KERNELS] = {
        "emSparseKernel<64,true>", "emSparseKernel<256,true>", "emSparseKernel<256,false>", "emSparseKernel<1024,false>",
        "emRegisterBinKernel<1,16>", "emRegisterBinKernel<2,16>", "emRegisterBinKernel<4,16>", "emSparseKernel<1024,true>",
        "emRegisterBinKernel<1,32>", "emRegisterBinKernel<2,32>", "emSparseKernel<1024,false,WIDE>", "emGridAccumKernel"};
    if (index == static_cast<int>(kRegisterKernelIndex) && oneRegisterLaunch()) return "emRegisterKernel";
    return (index >= 0 && index < RPVG_HIP_EM_KERNELS) ? names[index] : nullptr;
}

extern "C" int rpvg_hip_em_solve(rpvg_hip_ctx * ctx, const rpvg_hip_batch * batch, uint32_t max_em_its,
                                 double max_rel_em_conv, const rpvg_hip_em_problems * problems,
                                 rpvg_hip_em_results * results) {
    RPVG_REQUIRE(ctx && batch && problems && results, "rpvg_hip_em_solve: NULL argument");
    const uint32_t P = problems->num_problems;
    if (P == 0) return RPVG_HIP_OK;
    RPVG_REQUIRE(problems->cluster && problems->col_off && problems->col_path, "rpvg_hip_em_solve: NULL problem arrays");
    RPVG_REQUIRE(results->abundances && results->noise_count && results->total_count && results->iterations,
                 "rpvg_hip_em_solve: NULL result arrays");
    RPVG_REQUIRE(max_em_its > 0, "rpvg_hip_em_solve: max_em_its must be positive");

    std::lock_guard<std::mutex> lock(ctx->mutex);
    RPVG_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;

    // the results and the per-problem counts in one block, one copy back
    std::vector<uint32_t> kept_rows(P), kept_ent(P);
    DeviceBuffer<uint32_t> d_iters, d_kept_rows, d_kept_ent;
    DeviceBuffer<double> d_abund, d_noise_count, d_total;
    DownloadPack outputs;
    outputs.add(d_abund, results->abundances, problems->col_off[P]);
    outputs.add(d_noise_count, results->noise_count, P);
    outputs.add(d_total, results->total_count, P);
    outputs.add(d_iters, results->iterations, P);
    outputs.add(d_kept_rows, kept_rows.data(), P);
    outputs.add(d_kept_ent, kept_ent.data(), P);
    RPVG_HIP_CHECK(outputs.alloc());
    EmOutputs out{d_abund.ptr, d_noise_count.ptr, d_iters.ptr, d_kept_rows.ptr, d_kept_ent.ptr, d_total.ptr};

    HostProblemSet ps;
    int rc = prepareHostProblems(ctx, batch, problems, ps, out, "rpvg_hip_em_solve");
    if (rc != RPVG_HIP_OK) return rc;
    std::unique_ptr<HostScope> scope(new HostScope("em_solve: launches"));
    rc = queueEmSolve(ctx, batch, ps.list, max_em_its, max_rel_em_conv, out, ps.work, false, problems->collapse_precision);
    if (rc != RPVG_HIP_OK) return rc;

    scope.reset(new HostScope("em_solve: wait for the kernels + download"));
    RPVG_HIP_CHECK(outputs.fetch(st));
    RPVG_HIP_CHECK(waitStream(st));
    outputs.scatter();
    scope.reset();
    accountEmSolve(ctx, P, problems->col_off, kept_rows.data(), kept_ent.data(), results->iterations);
    return RPVG_HIP_OK;
}

extern "C" int rpvg_hip_gibbs_read_counts(rpvg_hip_ctx * ctx, const rpvg_hip_batch * batch,
                                          const rpvg_hip_em_problems * problems, const double * init_abundances,
                                          const double * init_noise_count, const uint32_t * num_samples,
                                          const uint64_t * seeds, uint32_t gibbs_thin_its, double gamma,
                                          double * noise_samples, double * abundance_samples) {
    RPVG_REQUIRE(ctx && batch && problems, "rpvg_hip_gibbs_read_counts: NULL argument");
    const uint32_t P = problems->num_problems;
    if (P == 0) return RPVG_HIP_OK;
    RPVG_REQUIRE(problems->cluster && problems->col_off && problems->col_path, "rpvg_hip_gibbs_read_counts: NULL problem arrays");
    RPVG_REQUIRE(init_abundances && init_noise_count && num_samples && seeds && noise_samples && abundance_samples,
                 "rpvg_hip_gibbs_read_counts: NULL argument");
    RPVG_REQUIRE(gibbs_thin_its > 0, "rpvg_hip_gibbs_read_counts: gibbs_thin_its must be positive");
    RPVG_REQUIRE(gamma >= 1.0, "rpvg_hip_gibbs_read_counts: gamma must be >= 1 (the reference uses 1)");

    std::vector<uint64_t> sample_off(P + 1, 0), abund_sample_off(P + 1, 0);
    for (uint32_t p = 0; p < P; ++p) {
        sample_off[p + 1] = sample_off[p] + num_samples[p];
        abund_sample_off[p + 1] = abund_sample_off[p] + static_cast<uint64_t>(num_samples[p]) * (problems->col_off[p + 1] - problems->col_off[p]);
    }
    if (sample_off[P] == 0) return RPVG_HIP_OK;

    std::lock_guard<std::mutex> lock(ctx->mutex);
    RPVG_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;

    // the compacted CSR of the problems, as for the EM (fill only: no queues, no EM kernels)
    DeviceBuffer<uint32_t> d_iters, d_kept_rows, d_kept_ent;
    DeviceBuffer<double> d_total;
    RPVG_HIP_CHECK(d_kept_rows.alloc(P));
    RPVG_HIP_CHECK(d_kept_ent.alloc(P));
    RPVG_HIP_CHECK(d_total.alloc(P));
    EmOutputs out{nullptr, nullptr, nullptr, d_kept_rows.ptr, d_kept_ent.ptr, d_total.ptr};
    HostProblemSet ps;
    int rc = prepareHostProblems(ctx, batch, problems, ps, out, "rpvg_hip_gibbs_read_counts");
    if (rc != RPVG_HIP_OK) return rc;
    rc = queueEmSolve(ctx, batch, ps.list, 1, 0.0, out, ps.work, true);
    if (rc != RPVG_HIP_OK) return rc;

    DeviceBuffer<double> d_init_abund, d_init_noise, d_noise_samples, d_abund_samples;
    DeviceBuffer<uint32_t> d_num_samples;
    DeviceBuffer<uint64_t> d_seed, d_sample_off, d_abund_sample_off;
    RPVG_HIP_CHECK(d_init_abund.upload(init_abundances, ps.n_cols_total, st));
    RPVG_HIP_CHECK(d_init_noise.upload(init_noise_count, P, st));
    RPVG_HIP_CHECK(d_num_samples.upload(num_samples, P, st));
    RPVG_HIP_CHECK(d_seed.upload(seeds, P, st));
    RPVG_HIP_CHECK(d_sample_off.upload(sample_off.data(), P + 1, st));
    RPVG_HIP_CHECK(d_abund_sample_off.upload(abund_sample_off.data(), P + 1, st));
    RPVG_HIP_CHECK(d_noise_samples.alloc(sample_off[P]));
    RPVG_HIP_CHECK(d_abund_samples.alloc(abund_sample_off[P]));

    GibbsLaunchArgs args;
    args.count = P;
    args.col_off = ps.d_col_off.ptr;
    args.row_base = ps.d_row_base.ptr;
    args.ent_base = ps.d_ent_base.ptr;
    args.kept_rows = d_kept_rows.ptr;
    args.zero_mass = ps.work.d_zero.ptr;
    args.total_mass = d_total.ptr;
    args.prow_off = ps.work.d_prow_off.ptr;
    args.prow_count = ps.work.d_prow_count.ptr;
    args.prow_noise = ps.work.d_prow_noise.ptr;
    args.pent_col = ps.work.d_pent_col.ptr;
    args.pent_val = ps.work.d_pent_val.ptr;
    args.init_abundances = d_init_abund.ptr;
    args.init_noise_count = d_init_noise.ptr;
    args.num_samples = d_num_samples.ptr;
    args.seed = d_seed.ptr;
    args.sample_off = d_sample_off.ptr;
    args.abund_sample_off = d_abund_sample_off.ptr;
    args.thin = gibbs_thin_its;
    args.gamma = gamma;
    args.noise_samples = d_noise_samples.ptr;
    args.abundance_samples = d_abund_samples.ptr;

    const size_t lds = (sizeof(double) * (2 * static_cast<size_t>(ps.list.max_cols) + 256 / 64 + 2) + 15) & ~static_cast<size_t>(15);
    RPVG_REQUIRE(lds <= 160 * 1024, "rpvg_hip_gibbs_read_counts: a problem with %u columns does not fit the sampler's LDS-resident vectors (limit ~10 000 columns)",
                 ps.list.max_cols);
    if (lds > 64 * 1024) {
        RPVG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&gibbsReadCountKernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    }
    const int span = ctx->spanBegin(FAM_EM_SPARSE);
    gibbsReadCountKernel<<<dim3(P), dim3(256), lds, st>>>(args);
    ctx->spanEnd(span);
    RPVG_HIP_CHECK(hipGetLastError());
    RPVG_HIP_CHECK(d_noise_samples.download(noise_samples, st));
    RPVG_HIP_CHECK(d_abund_samples.download(abundance_samples, st));
    RPVG_HIP_CHECK(waitStream(st));
    return RPVG_HIP_OK;
}
