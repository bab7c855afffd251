// Diploid branch-and-bound posterior search entirely on the GPU (gfx950).
//
// Takes over calculatePathGroupPosteriorsBounded (src/path_estimator.cpp:379-473)
// for a batch of group matrices: ONE workgroup per matrix walks the search in
// the reference's sequential order, so the kept set is the reference's kept
// set.  Inside a workgroup
//   - first columns : the next kTileFirst columns that pass the optimistic bound
//              are evaluated together (their base vectors noise_i + M[i][a]/2 and
//              the read counts staged in LDS), then pruned one after the other;
//   - waves  : second columns — a wave each, or 16 lanes each for matrices with
//              few rows; every element of a second column is read once for the
//              kTileFirst first columns;
//   - lanes  : rows of the matrix (column-major, so lane i reads element i of
//              a column: coalesced).
// sum_i count_i log(x_i) runs over rows ordered by class (rpvg_hip_groups): the
// count-1 rows as a running FP64 product with one logarithm at the end
// (LogProduct, common.hpp), the rest one table logarithm per row.
// Bound: FP64 issue and L2 latency, not HBM — a cluster's matrix (mean ~0.3 MB)
// is re-read from L2.

#include "common.hpp"

#include <algorithm>
#include <cstdlib>
#include <map>
#include <memory>
#include <mutex>
#include <numeric>

using namespace rpvg_hip_detail;

// ---- one search at a time per device -------------------------------------------------
// The host lanes of a batch (two contexts of one device) overlap one lane's host work with the other's kernels.
// Two searches side by side take longer than one after the other (their workgroups compete for whole CUs), and a
// lane whose kernels are spread over the step finishes as late as the other: the kernels of a search therefore wait,
// on the device, for the previous search of another context — the host has queued them by then, so the next search
// starts without a host round trip in between.
namespace {
std::mutex g_search_gate_mutex;
const rpvg_hip_ctx * g_search_gate_owner = nullptr;  // context of the last search queued
}

// A stream whose next command waits for an event holds its hardware queue until the event arrives — eight queues for all the
// streams of both lanes (hardwareQueues()) — and the kernels of other streams on that queue stand behind it: a search parked
// behind the other lane's search (0.8 ms) or behind its matrices' collapse (1 ms) stalled kernels that had nothing to do with
// either.  The submitting thread therefore waits for a long dependency itself and queues the kernels when they can run
// (it has nothing else to queue meanwhile); the stream wait stays, as the ordering guarantee.
// RPVG_HIP_SEARCH_LAUNCH_EARLY=1: stream waits only (A/B).
static bool searchLaunchesEarly() {
    static const bool early = RPVG_EXPERIMENT_ENV("RPVG_HIP_SEARCH_LAUNCH_EARLY") != nullptr;
    return early;
}

// Searches of fewer than 2^17 matrix rows stay outside: they do not fill the GPU, and the batches of PathEstimator::estimate()'s call
// combiner — a few dozen small clusters each, three of them on the GPU at once — would otherwise run one after the other with a
// host wait in between.
constexpr uint64_t kSearchGateRows = 1ull << 17;

static void searchGateEnter(rpvg_hip_ctx * ctx, hipStream_t stream, const uint64_t matrix_rows) {
    static const bool open_gate = RPVG_EXPERIMENT_ENV("RPVG_HIP_NO_SEARCH_GATE") != nullptr;  // A/B knob
    ctx->search_gate_held = !open_gate && matrix_rows >= kSearchGateRows;
    if (!ctx->search_gate_held) return;
    hipEvent_t previous = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_search_gate_mutex);
        if (g_search_gate_owner && g_search_gate_owner != ctx && g_search_gate_owner->device == ctx->device) {
            previous = g_search_gate_owner->search_done;
            (void) hipStreamWaitEvent(stream, previous, 0);
        }
    }
    // (outside the lock: the other lane records its own event under it)
    if (previous && !searchLaunchesEarly()) (void) waitEvent(previous);
}

static void searchGateLeave(rpvg_hip_ctx * ctx, hipStream_t stream) {
    if (!ctx->search_gate_held) return;
    ctx->search_gate_held = false;
    std::lock_guard<std::mutex> lock(g_search_gate_mutex);
    (void) hipEventRecord(ctx->search_done, stream);
    g_search_gate_owner = ctx;
}

void searchGateForget(const rpvg_hip_ctx * ctx) {
    std::lock_guard<std::mutex> lock(g_search_gate_mutex);
    if (g_search_gate_owner == ctx) g_search_gate_owner = nullptr;
}

// the kept pairs of every matrix: one page-locked block [posterior: total f64 | first: total u32 | second: total u32],
// filled by one D2H copy
struct rpvg_hip_pair_posteriors {
    std::vector<uint64_t> pair_off;
    void * block = nullptr;
    bool block_pinned = false;
    const uint32_t * first = nullptr, * second = nullptr;
    const double * posterior = nullptr;
    ~rpvg_hip_pair_posteriors() {
        if (block && block_pinned) pinnedFree(block);
        else std::free(block);
    }
};

namespace {

constexpr uint32_t kLdsRows = 2048;   // rows of (base, count) staged in LDS by the search kernel: 32 KB
constexpr uint32_t kSmallRows = 512;  // ... for matrices with at most this many rows: 8 KB
constexpr uint32_t kSmallRowLdsCols = 128;  // ... by the kernel for matrices with few rows (24 KB of LDS per workgroup: six per CU)
constexpr uint32_t kRowLdsCols = 512; // pair log-likelihoods of one first column kept in LDS up to this many columns: 4 KB

__device__ __forceinline__ double waveSum(double v) { return waveSumF64(v); }

// Utils::add_log (src/utils.hpp:300-302)
__device__ __forceinline__ double addLog(const double log_x, const double log_y) {
    return log_x > log_y ? log_x + log1p(exp(log_y - log_x)) : log_y + log1p(exp(log_x - log_y));
}


// log(sum_k exp(v_k)) of n values of a workgroup, v_k given by `value(k)` (values equal to `lowest` carry no
// mass).  The reference folds Utils::add_log over the values one by one (src/path_estimator.cpp:348,453-463);
// here the maximum is taken first and the exponentials are summed in parallel (same value up to rounding;
// a single thread folding a few hundred add_log calls was the longest serial stretch of these kernels).
// scratch: kBlock/64 doubles in LDS.  Result in every thread.
template <int kBlock, typename ValueFn>
__device__ __forceinline__ double blockLogSumExp(const uint32_t n, ValueFn value, double * scratch) {
    const double lowest = -1.7976931348623157e308;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double mx = lowest;
    for (uint32_t k = threadIdx.x; k < n; k += kBlock) mx = fmax(mx, value(k));
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mx = fmax(mx, __shfl_xor(mx, d, 64));
    __syncthreads();
    if (lane == 0) scratch[wave] = mx;
    __syncthreads();
#pragma unroll
    for (int w2 = 0; w2 < kBlock / 64; ++w2) mx = fmax(mx, scratch[w2]);
    double sum = 0.0;
    for (uint32_t k = threadIdx.x; k < n; k += kBlock) {
        const double v = value(k);
        if (v > lowest) sum += exp(v - mx);
    }
    sum = waveSum(sum);
    __syncthreads();
    if (lane == 0) scratch[wave] = sum;
    __syncthreads();
    double total = 0.0;
#pragma unroll
    for (int w2 = 0; w2 < kBlock / 64; ++w2) total += scratch[w2];
    return (n == 0 || !(mx > lowest)) ? lowest : mx + log(total);
}

struct SearchArgs {
    const uint32_t * order;          // matrices, expensive first
    uint32_t count;
    const uint64_t * mat_val_off;
    const uint64_t * mat_row_off;
    const uint32_t * mat_fast;       // row classes of the matrix (LogProduct, common.hpp): end of the fast rows,
    const uint32_t * mat_mid;        // end of the mid rows
    const uint64_t * mat_rows;
    const uint32_t * mat_cols;
    const double * values;
    const double * rowmax;
    const double * row_count;        // in matrix row order (rpvg_hip_groups), like row_noise
    const double * row_noise;
    const uint64_t * col_off;        // [M+1] prefix of columns (scratch + counts)
    const uint32_t * col_count;      // path_counts of every column
    const uint64_t * pair_cap_off;   // [M+1] prefix of G(G+1)/2 (output regions)
    double min_log_likelihood_diff;
    uint32_t stage_rows;             // rows of (base, count) staged in dynamic LDS (16 B each)
    uint32_t row_lds_cols;           // a row of pair log-likelihoods is kept in LDS when G <= this
    // per-column scratch
    double * log_freq;
    double * marginal;
    double * optimistic_raw;
    double * optimistic;             // in search order
    uint32_t * col_order;
    // output regions
    uint32_t * out_first;
    uint32_t * out_second;
    double * out_value;              // log-likelihood, then posterior
    uint32_t * out_count;            // [M]
    unsigned long long * log_evals;  // device counter: FP64 logs evaluated
};

constexpr int kTileFirst = 4;  // first columns evaluated together by the search kernel

// Pair sums of one second column b against the kTileFirst first columns of a tile, over the rows of a wave:
// out[t] = sum_i count_i log((noise_i + col_a[t][i] / 2) + col_b[i] / 2).  Every element of b is read once and feeds
// kTileFirst logs.  The first `staged` rows take base_t = noise + col_a[t] / 2 and the counts from LDS.  The rows
// are dealt to a group of kGroup lanes (`lane` = index within the group); the caller adds the sums up over the group.
template <int kGroup>
__device__ __forceinline__ void tilePairSums(const LogTableEntry * lt, const double * lds_count, const double * lds_base, const uint32_t stride,
                                             const double * __restrict__ cnt, const double * __restrict__ nz,
                                             const double * const (&col_a)[kTileFirst], const double * __restrict__ col_b, const uint32_t staged,
                                             const uint64_t fast_end, const uint64_t mid_end, const uint64_t R, const int lane,
                                             double (&out)[kTileFirst]) {
    LogProduct pr[kTileFirst];
    double acc[kTileFirst];
#pragma unroll
    for (int t = 0; t < kTileFirst; ++t) acc[t] = 0.0;
    auto clip = [&](const uint64_t v) { return static_cast<uint32_t>(v < staged ? v : staged); };
    sumCountLogsMulti<kTileFirst, kGroup, uint32_t>(lt, lds_count, [&](const uint32_t i, double (&xs)[kTileFirst]) {
        const double x = col_b[i] / 2.0;
#pragma unroll
        for (int t = 0; t < kTileFirst; ++t) xs[t] = lds_base[t * stride + i] + x;
    }, 0u, clip(fast_end), clip(mid_end), staged, lane, pr, acc);
    if (staged < R) {
        sumCountLogsMulti<kTileFirst, kGroup, uint64_t>(lt, cnt, [&](const uint64_t i, double (&xs)[kTileFirst]) {
            const double noise = nz[i], x = col_b[i] / 2.0;
#pragma unroll
            for (int t = 0; t < kTileFirst; ++t) xs[t] = (noise + col_a[t][i] / 2.0) + x;
        }, static_cast<uint64_t>(staged), fast_end, mid_end, R, lane, pr, acc);
    }
#pragma unroll
    for (int t = 0; t < kTileFirst; ++t) out[t] = mid_end ? acc[t] + pr[t].value(lt) : acc[t];
}

// kGroup lanes work on one second column at a time: 64 (a wave per column) for matrices with many rows, 16 (four
// columns per wave) for those with few, where the end of a column's sum — one logarithm per product, the
// reduction over the lanes — costs as much as its rows: four columns share those instructions.
template <int kBlock, int kGroup>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4))) void boundedSearchKernel(const SearchArgs args) {
    constexpr int kWaves = kBlock / 64;
    constexpr int kGroupsPerWave = 64 / kGroup;
    extern __shared__ __attribute__((aligned(16))) double lds_dyn[];
    double * lds_base = lds_dyn;                                  // [kTileFirst][stage_rows]
    double * lds_count = lds_dyn + kTileFirst * args.stage_rows;  // [stage_rows]
    double * lds_rows = lds_count + args.stage_rows;              // [kTileFirst][row_lds_cols]
    __shared__ double lds_red[kBlock / 64];
    __shared__ unsigned long long lds_sum;
    __shared__ LogTableEntry lt[kLogTableSize];

    if (blockIdx.x >= args.count) return;
    loadLogTable(lt);  // visible after the first barrier below
    const uint32_t m = args.order[blockIdx.x];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t R = args.mat_rows[m];
    const uint32_t G = args.mat_cols[m];
    const double * M = args.values + args.mat_val_off[m];
    const double * rm = args.rowmax + args.mat_row_off[m];
    const double * cnt = args.row_count + args.mat_row_off[m];
    const double * nz = args.row_noise + args.mat_row_off[m];
    const uint64_t fast_end = args.mat_fast[m], mid_end = args.mat_mid[m];
    const uint64_t c0 = args.col_off[m];
    const uint32_t * ccount = args.col_count + c0;
    double * lf = args.log_freq + c0;
    double * marg = args.marginal + c0;
    double * opt_raw = args.optimistic_raw + c0;
    double * opt = args.optimistic + c0;
    uint32_t * ord = args.col_order + c0;
    const uint64_t p0 = args.pair_cap_off[m];
    uint32_t * out_first = args.out_first + p0;
    uint32_t * out_second = args.out_second + p0;
    double * out_value = args.out_value + p0;
    const double thr = args.min_log_likelihood_diff;
    const double lowest = -1.7976931348623157e308;  // numeric_limits<double>::lowest()
    const double log_two = log(2.0);

    // calcPathLogFrequences (src/path_estimator.cpp:315-330)
    if (threadIdx.x == 0) lds_sum = 0;
    __syncthreads();
    unsigned long long part = 0;
    for (uint32_t g = threadIdx.x; g < G; g += kBlock) part += ccount[g];
    if (part) atomicAdd(&lds_sum, part);
    __syncthreads();
    const double count_sum = static_cast<double>(lds_sum);
    for (uint32_t g = threadIdx.x; g < G; g += kBlock) lf[g] = log(ccount[g] / count_sum);
    __syncthreads();

    // marginal log-posteriors (group size 1) and optimistic bounds of every column: two logs per row
    for (uint32_t g = wave; g < G; g += kWaves) {
        const double * col = M + static_cast<uint64_t>(g) * R;
        double acc1 = sumCountLogs<uint64_t>(lt, cnt, [&](const uint64_t i) { return nz[i] + col[i] / 1.0; }, 0, fast_end, mid_end, R, lane);
        double acc2 = sumCountLogs<uint64_t>(lt, cnt, [&](const uint64_t i) { return (nz[i] + col[i] / 2.0) + rm[i] / 2.0; }, 0, fast_end, mid_end, R, lane);
        acc1 = waveSum(acc1);
        acc2 = waveSum(acc2);
        if (lane == 0) {
            marg[g] = (acc1 + lf[g]) + 0.0;           // + log(numPermutations({g})) = log(1)
            opt_raw[g] = acc2 + (lf[g] + log_two);    // src/path_estimator.cpp:427-428
        }
    }
    __syncthreads();

    // normalise the marginals (:348,370-376)
    {
        const double sum_log = blockLogSumExp<kBlock>(G, [&](uint32_t g) { return marg[g]; }, lds_red);
        for (uint32_t g = threadIdx.x; g < G; g += kBlock) marg[g] = exp(marg[g] - sum_log);
    }
    __syncthreads();

    // descending (posterior, index) order (:412): rank by counting, ties impossible (indices differ)
    for (uint32_t g = threadIdx.x; g < G; g += kBlock) {
        const double pg = marg[g];
        uint32_t rank = 0;
        for (uint32_t h = 0; h < G; ++h) {
            const double ph = marg[h];
            rank += (ph > pg || (ph == pg && h > g)) ? 1u : 0u;
        }
        ord[rank] = g;
        opt[rank] = opt_raw[g];
    }
    __syncthreads();

    // The search (:418-451).  The reference reaches the first columns one by one, in `ord` order, and skips those
    // whose optimistic bound is too far below the running maximum.  Here the next kTileFirst columns that pass the
    // bound under the CURRENT maximum are evaluated together (every second column is read once for all of them);
    // their rows are then pruned one after the other exactly as the reference does it, and a column whose bound
    // fails by the time it is reached is dropped unseen — the kept set is the reference's.
    // Rows of pair log-likelihoods: LDS when they fit, else (one first column at a time) the now free bound scratch.
    const bool rows_in_lds = G <= args.row_lds_cols;
    const uint32_t tile_max = rows_in_lds ? kTileFirst : 1;
    const uint32_t row_stride = rows_in_lds ? args.row_lds_cols : 0;
    double * row_ll = rows_in_lds ? lds_rows : opt_raw;
    const uint32_t stride = args.stage_rows;

    double max_ll = lowest;
    uint32_t kept = 0;
    unsigned long long pairs_evaluated = 0;
    const uint32_t staged = static_cast<uint32_t>(R < args.stage_rows ? R : args.stage_rows);
    uint32_t pos = 0;
    while (pos < G) {
        uint32_t tile_pos[kTileFirst];
        uint32_t nt = 0;
        while (pos < G && nt < tile_max) {
            if (!(opt[pos] - max_ll < thr)) tile_pos[nt++] = pos;
            ++pos;
        }
        if (nt == 0) break;
#pragma unroll
        for (int t = 1; t < kTileFirst; ++t)
            if (t >= static_cast<int>(nt)) tile_pos[t] = tile_pos[0];  // unused slots repeat the first (uniform code)
        uint32_t a[kTileFirst];
        const double * col_a[kTileFirst];
#pragma unroll
        for (int t = 0; t < kTileFirst; ++t) {
            a[t] = ord[tile_pos[t]];
            col_a[t] = M + static_cast<uint64_t>(a[t]) * R;
        }
        __syncthreads();  // previous users of the staged vectors and of the rows are done
        for (uint32_t i = threadIdx.x; i < staged; i += kBlock) {
            const double noise = nz[i];
#pragma unroll
            for (int t = 0; t < kTileFirst; ++t) lds_base[t * stride + i] = noise + col_a[t][i] / 2.0;
            lds_count[i] = cnt[i];
        }
        __syncthreads();
        for (uint32_t t = 0; t < nt; ++t) pairs_evaluated += G - tile_pos[t];
        // every wave evaluates its share of the second columns without waiting for the others ...
        const uint32_t j0 = tile_pos[0];
        const int sub = lane & (kGroup - 1), grp = lane / kGroup;
        for (uint32_t jw = j0 + wave * kGroupsPerWave; jw < G; jw += kWaves * kGroupsPerWave) {
            const uint32_t j = jw + grp;
            const bool valid = j < G;  // idle groups of the last pass repeat the last column (uniform code)
            const uint32_t b = ord[valid ? j : G - 1];
            double sums[kTileFirst];
            tilePairSums<kGroup>(lt, lds_count, lds_base, stride, cnt, nz, col_a, M + static_cast<uint64_t>(b) * R, staged, fast_end, mid_end, R, sub, sums);
            const double lf_b = lf[b];
#pragma unroll
            for (int t = 0; t < kTileFirst; ++t) {
                const double total = kGroup == 64 ? waveSum(sums[t]) : rowSumF64(sums[t]);
                if (sub == 0 && valid && t < static_cast<int>(nt) && j >= tile_pos[t]) {
                    row_ll[t * row_stride + (j - tile_pos[t])] = total + ((lf[a[t]] + lf_b) + (a[t] == b ? 0.0 : log_two));
                }
            }
        }
        __syncthreads();
        // ... then the reference's pruning rule runs over the rows in order (uniformly in all threads)
        for (uint32_t t = 0; t < nt; ++t) {
            if (t > 0 && opt[tile_pos[t]] - max_ll < thr) continue;  // the reference skips this column when it gets there
            const double * row = row_ll + t * row_stride;
            const uint32_t n = G - tile_pos[t];
            if (thr <= 0.0) {
                // a dropped pair is below the running maximum, so "maximum of the kept pairs so far" = "maximum of
                // all pairs so far": exclusive prefix maximum by wave scan, kept pairs compacted by ballot
                for (uint32_t c = 0; c < n; c += 64) {
                    const uint32_t k = c + lane;
                    const double ll = k < n ? row[k] : lowest;
                    double incl = ll;
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) {
                        const double up = __shfl_up(incl, d, 64);
                        if (lane >= d) incl = fmax(incl, up);
                    }
                    double before = __shfl_up(incl, 1, 64);
                    before = lane == 0 ? max_ll : fmax(max_ll, before);
                    const bool keep = k < n && !(ll - before < thr);
                    const unsigned long long ballot = __ballot(keep);
                    if (keep && wave == 0) {
                        const uint32_t slot = kept + __popcll(ballot & ((1ull << lane) - 1ull));
                        out_first[slot] = a[t];
                        out_second[slot] = ord[tile_pos[t] + k];
                        out_value[slot] = ll;
                    }
                    kept += __popcll(ballot);
                    max_ll = fmax(max_ll, __shfl(incl, 63, 64));
                }
            } else {
                for (uint32_t k = 0; k < n; ++k) {
                    const double ll = row[k];
                    if (!(ll - max_ll < thr)) {
                        max_ll = fmax(max_ll, ll);
                        if (threadIdx.x == 0) {
                            out_first[kept] = a[t];
                            out_second[kept] = ord[tile_pos[t] + k];
                            out_value[kept] = ll;
                        }
                        ++kept;
                    }
                }
            }
        }
    }

    // late losers -> weight zero, log-sum-exp, posteriors (:453-470)
    __syncthreads();
    if (threadIdx.x == 0) {
        args.out_count[m] = kept;
        atomicAdd(args.log_evals, (2ull * G + pairs_evaluated) * R);
    }
    {
        auto kept_value = [&](uint32_t k) { const double ll = out_value[k]; return (ll - max_ll < thr) ? lowest : ll; };
        const double sum_log = blockLogSumExp<kBlock>(kept, kept_value, lds_red);
        for (uint32_t k = threadIdx.x; k < kept; k += kBlock) out_value[k] = exp(kept_value(k) - sum_log);
    }
}


// ---- table path for big matrices ----------------------------------------------
//
// The sequential rule "keep pair s iff ll_s - max(kept so far) >= thr" equals
// "keep iff ll_s - max(all pairs before s) >= thr": a dropped pair is below the
// running maximum, so it never changes it.  A first column that the reference
// skips on its optimistic bound only holds pairs that this filter drops (the
// bound dominates every pair of the column).  So for a big matrix every pair
// is evaluated in parallel by many workgroups (row chunks x first columns),
// and one workgroup then applies an exclusive prefix-max scan in the
// reference's pair order.

constexpr uint32_t kChunkRows = 1024;
constexpr int kTileA = 4;

struct TableWork {
    const uint32_t * item_matrix;   // [W]
    const uint32_t * item_col;      // [W]
    const uint32_t * item_chunk;    // [W]
    uint32_t count;
    const uint64_t * mat_val_off;
    const uint64_t * mat_row_off;
    const uint32_t * mat_fast;
    const uint32_t * mat_mid;
    const uint64_t * mat_rows;
    const uint32_t * mat_cols;
    const double * values;
    const double * rowmax;
    const double * row_count;
    const double * row_noise;
    const uint64_t * big_col_part_off;   // [M] offset of the matrix's [chunk][G] partial column sums (0 for small ones)
    const uint64_t * big_pair_part_off;  // [M] offset of the matrix's [chunk][G][G] partial pair sums
    double * part_marginal;
    double * part_optimistic;
    double * part_pair;
    unsigned long long * log_evals;
};

// One workgroup per (matrix, tile of kTileA consecutive first columns, chunk of kChunkRows rows): partial
// sums over the chunk's rows of the tile's marginal log-likelihoods and of every pair (a, b >= a), a in
// the tile.  The tile's base vectors noise_i + M[i][a]/2 and the read counts are staged in LDS; every
// second column b is then read ONCE per row and feeds kTileA logs (kTileA-way ILP for free, and
// kTileA-times less L2/HBM traffic than one first column per workgroup: that version re-read 27 GB per
// step of the bench workload and was memory-bound).  Work items are dealt to XCDs in contiguous ranges
// (workgroup b runs on XCD b % 8), so the tiles of one row chunk share that XCD's L2.
__global__ __launch_bounds__(256) void pairTableKernel(const TableWork w) {
    __shared__ double lds_base[kTileA][kChunkRows];
    __shared__ double lds_count[kChunkRows];
    __shared__ LogTableEntry lt[kLogTableSize];
    const uint32_t per_xcd = (w.count + 7) / 8;
    const uint32_t item = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    if (item >= w.count) return;
    loadLogTable(lt);  // visible after the barrier that publishes the staged rows
    const uint32_t m = w.item_matrix[item], a0 = w.item_col[item], chunk = w.item_chunk[item];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t R = w.mat_rows[m];
    const uint32_t G = w.mat_cols[m];
    const uint32_t ta = (G - a0) < kTileA ? (G - a0) : kTileA;
    const uint64_t r_begin = static_cast<uint64_t>(chunk) * kChunkRows;
    const uint32_t n = static_cast<uint32_t>((R - r_begin) < kChunkRows ? (R - r_begin) : kChunkRows);
    const double * M = w.values + w.mat_val_off[m] + r_begin;
    const double * cnt = w.row_count + w.mat_row_off[m] + r_begin;
    const double * nz = w.row_noise + w.mat_row_off[m] + r_begin;
    auto local = [&](const uint64_t end_row) { return end_row <= r_begin ? 0u : (end_row - r_begin < n ? static_cast<uint32_t>(end_row - r_begin) : n); };
    const uint32_t nf = local(w.mat_fast[m]), nm = local(w.mat_mid[m]);  // class boundaries within the chunk

    for (uint32_t i = threadIdx.x; i < n; i += 256) {
        const double noise = nz[i];
        lds_count[i] = cnt[i];
#pragma unroll
        for (int t = 0; t < kTileA; ++t) {
            const uint32_t a = a0 + (t < static_cast<int>(ta) ? t : 0);
            lds_base[t][i] = noise + M[static_cast<uint64_t>(a) * R + i] / 2.0;
        }
    }
    __syncthreads();
    // tasks dealt round-robin to the 4 waves: task t < ta = marginal partial sum of column a0 + t,
    // task ta + k = pairs (a0 + t, a0 + k) for every t <= k of the tile
    const uint32_t num_tasks = ta + (G - a0);
    for (uint32_t task = wave; task < num_tasks; task += 4) {
        if (task < ta) {
            const uint32_t a = a0 + task;
            const double * col_a = M + static_cast<uint64_t>(a) * R;
            const double acc = waveSum(sumCountLogs<uint32_t>(lt, lds_count, [&](const uint32_t i) { return nz[i] + col_a[i] / 1.0; }, 0u, nf, nm, n, lane));
            if (lane == 0) w.part_marginal[w.big_col_part_off[m] + static_cast<uint64_t>(chunk) * G + a] = acc;
        } else {
            const uint32_t b = a0 + (task - ta);
            const double * col_b = M + static_cast<uint64_t>(b) * R;
            double acc[kTileA];
            LogProduct pr[kTileA];
#pragma unroll
            for (int t = 0; t < kTileA; ++t) acc[t] = 0.0;
            sumCountLogsMulti<kTileA, 64, uint32_t>(lt, lds_count, [&](const uint32_t i, double (&xs)[kTileA]) {
                const double x = col_b[i] / 2.0;
#pragma unroll
                for (int t = 0; t < kTileA; ++t) xs[t] = lds_base[t][i] + x;
            }, 0u, nf, nm, n, lane, pr, acc);
            if (nm) {
#pragma unroll
                for (int t = 0; t < kTileA; ++t) acc[t] += pr[t].value(lt);
            }
#pragma unroll
            for (int t = 0; t < kTileA; ++t) {
                const double total = waveSum(acc[t]);
                if (lane == 0 && t < static_cast<int>(ta) && a0 + t <= b) {
                    w.part_pair[w.big_pair_part_off[m] + (static_cast<uint64_t>(chunk) * G + (a0 + t)) * G + b] = total;
                }
            }
        }
    }
    if (threadIdx.x == 0) atomicAdd(w.log_evals, static_cast<unsigned long long>(ta + (G - a0) * kTileA) * n);
}


// ---- all pairs of a matrix from LDS, a 4 x 4 tile of pairs per lane ----------------------------------------------
//
// The sequential search prunes little where it matters (configs[2] bench: 4.1 of the 4.4 G row-pair evaluations of "all
// pairs" are reached anyway), and "keep iff ll - max(all pairs before) >= thr" needs no order of evaluation (above).  So
// every pair of every matrix is evaluated, by a kernel built around the arithmetic alone.  (A first attempt — lane =
// second column, eight first columns through scalar loads from a row-major copy of the matrices, round 2's
// pairRowsKernel — had the minimal arithmetic in its inner loop and only matched the sequential search: every tile of
// first columns streamed the whole matrix again, half of the lanes idled on the triangle of pairs, a wave owned 64
// second columns whether the matrix had them or not.)  This kernel is laid out like a matrix product whose inner
// dimension is the rows:
//   * a workgroup takes (matrix, chunk of kChunkRows rows) and stages the chunk, some hundred rows at a time, in LDS —
//     halved and transposed to row-major on the way, so the matrices need no second copy — and every pair of the
//     matrix is evaluated from that ONE staging: the matrix is read from memory once per search;
//   * a lane owns a 4 x 4 tile of pairs (a in 4 ta .., b in 4 tb .., ta <= tb: only the diagonal tiles carry unused
//     slots): per row it reads 4 + 4 values and the row's noise from LDS (two 16-byte reads each) and does 16 additions
//     and 16 multiplications into 16 running products (LogProduct) — 2 FP64 instructions per pair and row, 0.25 LDS
//     reads;
//   * matrices with few tiles spread the rows of the chunk over the lanes instead (slices: lane = (tile, slice), the
//     slice takes every S-th row): the partial sums of a slice are one more part of the chunk for the resolving
//     workgroup, which adds parts anyway.
// Rows with read counts 2 .. kMidMaxCount multiply that many times, the others take the table logarithm, as everywhere.
constexpr uint32_t kTileBlock = 256;
constexpr uint32_t kTileLdsDoubles = 6 * 1024;   // staged values + noise + counts: 48 KB (three workgroups per CU)
constexpr uint32_t kTileMaxColumns = 1024;       // wider matrices keep the sequential search (their pair tables would not fit either)

__host__ __device__ inline uint32_t tileColumns(const uint32_t G) { return (G + 3) / 4; }
__host__ __device__ inline uint32_t tileCount(const uint32_t G) { return tileColumns(G) * (tileColumns(G) + 1) / 2; }
// row slices of a chunk: lanes left over by the tiles
__host__ __device__ inline uint32_t tileSlices(const uint32_t G) { return tileCount(G) <= kTileBlock ? kTileBlock / tileCount(G) : 1u; }
// ... for the single columns (marginals): a lane per four columns and slice — all lanes of the workgroup, so that the
// few column sums are not a serial tail behind the pairs (26 lanes walking every second row cost 1.9 of the kernel's 6.5 ms)
__host__ __device__ inline uint32_t marginalSlices(const uint32_t G) { return kTileBlock / tileColumns(G); }

struct PairTileWork {
    const uint32_t * item_matrix;   // [W]
    const uint32_t * item_chunk;    // [W]
    const uint32_t * item_tiles;    // [W] pairTile2Kernel: first tile | (tiles - 1) << 16
    uint32_t chunk_rows;            // pairTile2Kernel: rows of a chunk (kChunkRows for the others)
    uint32_t count;
    const uint64_t * mat_val_off;
    const uint64_t * mat_row_off;
    const uint32_t * mat_fast;
    const uint32_t * mat_mid;
    const uint64_t * mat_rows;
    const uint32_t * mat_cols;
    const double * values;
    const double * row_count;
    const double * row_noise;
    const uint64_t * col_part_off;   // [M] offset of the matrix's [chunk][G] partial column sums
    const uint64_t * pair_part_off;  // [M] offset of the matrix's [chunk][G][G] partial pair sums
    double * part_marginal;
    double * part_pair;
    unsigned long long * log_evals;
#ifdef RPVG_HIP_EXPERIMENTS
    uint32_t debug_skip;  // timing experiments (RPVG_HIP_PAIR_DEBUG): 1 no count-1 rows, 2 no mid rows, 4 no other rows, 8 no marginals, 16 no loads,
                          // 32 no epilogue, 64 the prologue alone, 128 the dispatch of the grid alone
#endif
};

// (the experiments build times the kernel with classes of rows left out: wrong results, on purpose; not in the shipped library)
#ifdef RPVG_HIP_EXPERIMENTS
#define RPVG_PAIR_DEBUG_SKIP(w) ((w).debug_skip)
#else
#define RPVG_PAIR_DEBUG_SKIP(w) 0u
#endif

// ---- the same tiles, lanes balanced and the staging asynchronous (round 4) -----------------------------------------
//
// What pairTileKernel left on the table (configs[2] bench: 99 % of its evaluations are in matrices of 33 .. 64 columns, most
// of them 64): 136 tiles on 256 lanes is one slice — 53 % of the lanes, a wave of 8 lanes walking every row — and its staging
// (loads to registers, transposed stores to LDS, between two barriers) and its arithmetic added up instead of overlapping
// (1.40 ms = 0.47 staging and fixed costs + 0.93 arithmetic; without the loads 0.93).  Here
//   * a work item is (matrix, chunk of rows, RANGE OF TILES): the host cuts the tiles of a matrix so that tiles x slices fills
//     the workgroup — 136 tiles = 128 tiles in two slices + 8 tiles in 32 slices, 0.53 of the rows per lane instead of all of
//     them; an item stages only the columns its tiles touch;
//   * the rows are staged COLUMN-MAJOR, as they lie in memory, by LDS-direct loads (global_load_lds_dwordx4: two rows per
//     lane, no registers, no transposing stores) into one of two buffers: the loads of the next rows are in flight while the
//     lanes work on the current ones, one barrier per staged block;
//   * the values are staged as they are, not halved: a lane multiplies 2 x = (u + 2 noise) + v — the same roundings as
//     (u / 2 + noise) + v / 2 scaled by two — and takes one from the product's exponent per factor.
constexpr uint32_t kTile2BufferDoubles = 3 * 1024;  // two of them: 48 KB (three workgroups per CU)

__host__ __device__ inline uint32_t tileRowOfTile(const uint32_t t, const uint32_t T) {
    uint32_t lo = 0, hi = T - 1;  // row ta of the triangle starts at tile ta * T - ta (ta - 1) / 2
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (mid * T - mid * (mid - 1) / 2 <= t) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__device__ __forceinline__ void ldsDirectLoad16(const double * from, double * lds_wave_base) {
    // (destination: the wave-uniform base + 16 bytes x lane; source: the lane's own address)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)from, (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

// The nine values a lane takes from a staged row — noise, four first-column and four second-column values, the columns
// kSubRows doubles apart — as nine ds_read_b64 at immediate offsets from three addresses.  Written out: the compiler pairs
// such reads into ds_read2_b64, which the LDS serves at half the rate (8 cycles a wave for 1 KB against 2 for a ds_read_b64's
// 512 bytes), and the reads of this loop keep the LDS as busy as its FP64 instructions keep the SIMDs (1.49 against 1.22 ms).
template <uint32_t kSubRows>
__device__ __forceinline__ void ldsReadRow(const uint32_t noise_at, const uint32_t u_at, const uint32_t v_at, double & noise, double (&u)[4], double (&v)[4]) {
    // (one statement: the compiler puts its own waits before or behind it, not between the reads)
    asm volatile("ds_read_b64 %0, %9\n\t"
                 "ds_read_b64 %1, %10\n\t"
                 "ds_read_b64 %2, %10 offset:%12\n\t"
                 "ds_read_b64 %3, %10 offset:%13\n\t"
                 "ds_read_b64 %4, %10 offset:%14\n\t"
                 "ds_read_b64 %5, %11\n\t"
                 "ds_read_b64 %6, %11 offset:%12\n\t"
                 "ds_read_b64 %7, %11 offset:%13\n\t"
                 "ds_read_b64 %8, %11 offset:%14"
                 : "=&v"(noise), "=&v"(u[0]), "=&v"(u[1]), "=&v"(u[2]), "=&v"(u[3]), "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
                 : "v"(noise_at), "v"(u_at), "v"(v_at), "i"(kSubRows * 8), "i"(kSubRows * 16), "i"(kSubRows * 24));
}

// ... and the wait for them: the values pass through it, so that nothing that uses them moves above it
__device__ __forceinline__ void ldsRowLanded(double & noise, double (&u)[4], double (&v)[4]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(noise), "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
}

__device__ __forceinline__ uint32_t ldsByteAddress(const double * const p) {
    return static_cast<uint32_t>(reinterpret_cast<uintptr_t>(p));  // (a generic pointer into LDS: the offset is its low word)
}

// Rows per staged block: a compile-time constant, so that the eight values of a row sit at immediate offsets from one address
// per side (a stride in a register cost nine address additions and nine increments per row next to the 36 FP64 instructions);
// even (a lane loads two).  The largest of the menu whose block fits a buffer.
__host__ __device__ inline uint32_t tileSubRows(const uint32_t ncols) {
    const uint32_t fit = (kTile2BufferDoubles - ncols / 2) / (ncols + 2);
    return fit >= 126 ? 126u : fit >= 94 ? 94u : fit >= 46 ? 46u : fit >= 30 ? 30u : fit >= 14 ? 14u : fit >= 6 ? 6u : 2u;
}

template <uint32_t kSubRows>
__device__ __forceinline__ void pairTile2Item(const PairTileWork & w, double * const tile_lds, const LogTableEntry * const lt, const uint32_t m, const uint32_t chunk,
                                              const uint32_t t0, const uint32_t tcount, const uint32_t c_lo, const uint32_t ncols) {
    const uint64_t R = w.mat_rows[m];
    const uint32_t G = w.mat_cols[m];
    const uint32_t T = tileColumns(G);
    const uint32_t S = kTileBlock / tcount;
    constexpr uint32_t sub_rows = kSubRows;
    // a load instruction fills 16 bytes x 64 lanes of LDS in one piece: 1, 2 or 4 columns of one group of four (the groups are
    // two doubles apart: the lanes of a wave read the same row of up to 16 different groups, whose stride over the 64 banks is
    // 4 (mod 8) words that way — with 8 sub_rows words they would fall on four bank positions)
    const uint32_t lanes_per_column = sub_rows / 2, columns_per_load = lanes_per_column <= 16 ? 4u : lanes_per_column <= 32 ? 2u : 1u;
    auto columnOffset = [&](const uint32_t c) { return static_cast<size_t>(c) * sub_rows + (c / 4) * 2; };
    const uint64_t r_begin = static_cast<uint64_t>(chunk) * w.chunk_rows;
    const uint32_t n = static_cast<uint32_t>((R - r_begin) < w.chunk_rows ? (R - r_begin) : w.chunk_rows);
    const double * M = w.values + w.mat_val_off[m] + r_begin;  // column-major: M[column * R + row]
    const double * cnt = w.row_count + w.mat_row_off[m] + r_begin;
    const double * nz = w.row_noise + w.mat_row_off[m] + r_begin;
    auto local = [&](const uint64_t end_row) { return end_row <= r_begin ? 0u : (end_row - r_begin < n ? static_cast<uint32_t>(end_row - r_begin) : n); };
    const uint32_t nf = local(w.mat_fast[m]), nm = local(w.mat_mid[m]);  // class boundaries within the chunk
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;

    // a buffer: [ncols][sub_rows] values, [sub_rows] noise, [sub_rows] counts
    const uint32_t buffer_doubles = (ncols + 2) * sub_rows + ncols / 2;
    auto stage = [&](const uint32_t s0, const uint32_t ns, double * const buffer) {
        if (RPVG_PAIR_DEBUG_SKIP(w) & 16u) return;
        const uint32_t pairs = (ns + 1) / 2;  // (a row past the block's last is read, not used)
        const uint32_t col_in_load = lane / lanes_per_column, pair = lane % lanes_per_column;
        const bool lane_loads = col_in_load < columns_per_load && pair < pairs;
        for (uint32_t c0 = wave * columns_per_load; c0 < ncols; c0 += 4 * columns_per_load) {
            const uint32_t c = c0 + col_in_load;
            if (lane_loads && c < ncols) {
                const uint32_t column = c_lo + c < G ? c_lo + c : G - 1;  // (columns past the matrix: slots nobody keeps)
                ldsDirectLoad16(M + static_cast<uint64_t>(column) * R + s0 + 2 * pair, buffer + columnOffset(c0));
            }
        }
        if (wave < 2 && lane < pairs) {
            ldsDirectLoad16((wave == 0 ? nz : cnt) + s0 + 2 * lane, buffer + columnOffset(ncols) + wave * sub_rows);
        }
    };

    // the lane's tile and slice; its column of marginals (the item with the matrix's first tiles only)
    const uint32_t t = t0 + threadIdx.x % tcount, slice = threadIdx.x / tcount;
    const bool active = slice < S;
    const uint32_t ta = tileRowOfTile(t, T);
    const uint32_t tb = ta + (t - (ta * T - ta * (ta > 0 ? ta - 1 : 0) / 2));
    const uint32_t SM = marginalSlices(G);
    const uint32_t tc = threadIdx.x % T, marg_slice = threadIdx.x / T;
    const bool with_marginals = t0 == 0 && marg_slice < SM && !(RPVG_PAIR_DEBUG_SKIP(w) & 8u);

    LogProduct pr[4][4], prm[4];
    double acc[4][4], accm[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        accm[i] = 0.0;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
    }
    uint32_t since_fold = 0, since_fold_m = 0;
    int doubled = 0;  // factors of the pairs' products that were 2 x
    auto foldPairs = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) pr[i][j].fold();
        }
        since_fold = 0;
    };
    auto foldMarginals = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) prm[i].fold();
        since_fold_m = 0;
    };

    const uint32_t blocks = (n + sub_rows - 1) / sub_rows;
    if (RPVG_PAIR_DEBUG_SKIP(w) & 64u) return;  // (timing: the prologue alone)
    if (RPVG_PAIR_DEBUG_SKIP(w) & 16u) {  // (timing without the loads: finite values everywhere)
        for (uint32_t i = threadIdx.x; i < 2 * kTile2BufferDoubles; i += kTileBlock) tile_lds[i] = 0.25;
    }
    stage(0, n < sub_rows ? n : sub_rows, tile_lds);
    for (uint32_t blk = 0; blk < blocks; ++blk) {
        const uint32_t s0 = blk * sub_rows;
        const uint32_t ns = (n - s0) < sub_rows ? (n - s0) : sub_rows;
        double * const H = tile_lds + static_cast<size_t>(blk & 1u) * buffer_doubles;
        // this block's rows have landed (every wave waits for its own loads, the barrier for everybody's), and the other buffer
        // has been read for the last time: the next block's loads go out before the arithmetic on this one starts
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (blk + 1 < blocks) {
            const uint32_t s1 = s0 + sub_rows;
            stage(s1, (n - s1) < sub_rows ? (n - s1) : sub_rows, tile_lds + static_cast<size_t>((blk + 1) & 1u) * buffer_doubles);
        }
        const double * const lds_noise = H + columnOffset(ncols);
        const double * const lds_count = lds_noise + sub_rows;
        const uint32_t f1 = nf <= s0 ? 0u : ((nf - s0) < ns ? (nf - s0) : ns);
        const uint32_t m1 = nm <= s0 ? 0u : ((nm - s0) < ns ? (nm - s0) : ns);
        if (active) {
            const double * const ua = H + columnOffset(4 * ta - c_lo), * const vb = H + columnOffset(4 * tb - c_lo);
            // read count 1: one multiplication per pair and row
            uint32_t r = (RPVG_PAIR_DEBUG_SKIP(w) & 1u) ? f1 : slice;
            {
                uint32_t noise_at = ldsByteAddress(lds_noise + r), u_at = ldsByteAddress(ua + r), v_at = ldsByteAddress(vb + r);
                for (; r < f1; r += S, noise_at += 8 * S, u_at += 8 * S, v_at += 8 * S) {
                    double noise, u[4], v[4];
                    ldsReadRow<kSubRows>(noise_at, u_at, v_at, noise, u, v);
                    ldsRowLanded(noise, u, v);
                    const double un[4] = {fma(2.0, noise, u[0]), fma(2.0, noise, u[1]), fma(2.0, noise, u[2]), fma(2.0, noise, u[3])};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) pr[i][j].mul(un[i] + v[j]);
                    }
                    doubled += 1;
                    if (++since_fold == kFoldFactors) foldPairs();
                }
            }
            // read counts 2 .. kMidMaxCount: the factor that many times
            for (r = (RPVG_PAIR_DEBUG_SKIP(w) & 2u) ? m1 : f1 + (slice + S - f1 % S) % S; r < m1; r += S) {
                double noise, u[4], v[4];
                ldsReadRow<kSubRows>(ldsByteAddress(lds_noise + r), ldsByteAddress(ua + r), ldsByteAddress(vb + r), noise, u, v);
                ldsRowLanded(noise, u, v);
                const uint32_t c = static_cast<uint32_t>(lds_count[r]);
                const double un[4] = {fma(2.0, noise, u[0]), fma(2.0, noise, u[1]), fma(2.0, noise, u[2]), fma(2.0, noise, u[3])};
                if (since_fold + c > kFoldFactors) foldPairs();
                since_fold += c;
                doubled += static_cast<int>(c);
                double x[4][4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) x[i][j] = un[i] + v[j];
                }
                for (uint32_t k = 0; k < c; ++k) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) pr[i][j].mul(x[i][j]);
                    }
                }
            }
            // the rest: one logarithm per pair and row
            for (r = (RPVG_PAIR_DEBUG_SKIP(w) & 4u) ? ns : m1 + (slice + S - m1 % S) % S; r < ns; r += S) {
                const double noise = lds_noise[r], c = lds_count[r];
                const double un[4] = {fma(2.0, noise, ua[r]), fma(2.0, noise, ua[sub_rows + r]), fma(2.0, noise, ua[2 * sub_rows + r]), fma(2.0, noise, ua[3 * sub_rows + r])};
                const double v[4] = {vb[r], vb[sub_rows + r], vb[2 * sub_rows + r], vb[3 * sub_rows + r]};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = fma(c, logPositive(0.5 * (un[i] + v[j]), lt), acc[i][j]);
                }
            }
        }
        if (with_marginals) {  // single columns: noise + the whole value
            const double * const ua = H + columnOffset(4 * tc);  // (c_lo = 0 for this item)
            for (uint32_t r = marg_slice; r < ns; r += SM) {
                const double noise = lds_noise[r];
                const double x[4] = {ua[r] + noise, ua[sub_rows + r] + noise, ua[2 * sub_rows + r] + noise, ua[3 * sub_rows + r] + noise};
                if (r < f1) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) prm[i].mul(x[i]);
                    if (++since_fold_m == kFoldFactors) foldMarginals();
                } else if (r < m1) {
                    const uint32_t c = static_cast<uint32_t>(lds_count[r]);
                    if (since_fold_m + c > kFoldFactors) foldMarginals();
                    since_fold_m += c;
                    for (uint32_t k = 0; k < c; ++k) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) prm[i].mul(x[i]);
                    }
                } else {
                    const double c = lds_count[r];
#pragma unroll
                    for (int i = 0; i < 4; ++i) accm[i] = fma(c, logPositive(x[i], lt), accm[i]);
                }
            }
        }
    }
    // The sums of a chunk: one per pair and column.  Slices add theirs up in LDS, in the order of the slices (the staged rows
    // are done with, no load is in flight), so that the resolving workgroup reads one part per chunk.
    if (RPVG_PAIR_DEBUG_SKIP(w) & 32u) return;  // (timing: without the epilogue)
    double * const out_pairs = w.part_pair + w.pair_part_off[m] + static_cast<uint64_t>(chunk) * G * G;
    double * const out_columns = w.part_marginal + w.col_part_off[m] + static_cast<uint64_t>(chunk) * G;
    double * const sums = tile_lds;         // [S][tcount][16], then
    double * const column_sums = tile_lds;  // [SM][T][4]: one after the other in the same LDS
    if (S > 1) __syncthreads();
    if (active) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t a = 4 * ta + i, b = 4 * tb + j;
                pr[i][j].fold();
                pr[i][j].e -= doubled;
                const double total = acc[i][j] + pr[i][j].value(lt);
                if (S > 1) sums[(slice * tcount + (t - t0)) * 16 + i * 4 + j] = total;
                else if (a <= b && b < G) out_pairs[static_cast<uint64_t>(a) * G + b] = total;
            }
        }
    }
    if (S > 1) {
        __syncthreads();
        for (uint32_t e = threadIdx.x; e < tcount * 16; e += kTileBlock) {
            double total = 0.0;
            for (uint32_t sl = 0; sl < S; ++sl) total += sums[sl * tcount * 16 + e];
            const uint32_t te = t0 + e / 16, i = (e % 16) / 4, j = e % 4;
            const uint32_t row = tileRowOfTile(te, T);
            const uint32_t a = 4 * row + i, b = 4 * (row + (te - (row * T - row * (row > 0 ? row - 1 : 0) / 2))) + j;
            if (a <= b && b < G) out_pairs[static_cast<uint64_t>(a) * G + b] = total;
        }
    }
    if (t0 == 0 && !(RPVG_PAIR_DEBUG_SKIP(w) & 8u)) {
        if (SM > 1) __syncthreads();
        if (with_marginals) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t a = 4 * tc + i;
                const double total = accm[i] + prm[i].value(lt);
                if (SM > 1) column_sums[(marg_slice * T + tc) * 4 + i] = total;
                else if (a < G) out_columns[a] = total;
            }
        }
        if (SM > 1) {
            __syncthreads();
            for (uint32_t e = threadIdx.x; e < T * 4; e += kTileBlock) {
                double total = 0.0;
                for (uint32_t sl = 0; sl < SM; ++sl) total += column_sums[sl * T * 4 + e];
                if (e < G) out_columns[e] = total;
            }
        }
    }
    if (threadIdx.x == 0 && t0 == 0) atomicAdd(w.log_evals, static_cast<unsigned long long>(static_cast<uint64_t>(G) * (G + 1) / 2 + G) * n);
}

__global__ __launch_bounds__(kTileBlock) __attribute__((amdgpu_waves_per_eu(3))) void pairTile2Kernel(const PairTileWork w) {
    extern __shared__ __attribute__((aligned(16))) double tile_lds[];
    __shared__ LogTableEntry lt[kLogTableSize];
    const uint32_t item = blockIdx.x;
    if (item >= w.count) return;
    const uint32_t m = w.item_matrix[item], chunk = w.item_chunk[item];
    const uint32_t t0 = w.item_tiles[item] & 0xffffu, tcount = (w.item_tiles[item] >> 16) + 1;  // tiles [t0, t0 + tcount)
    if (RPVG_PAIR_DEBUG_SKIP(w) & 128u) return;  // (timing: the dispatch of the grid alone)
    loadLogTable(lt);  // visible after the first barrier of the item
    const uint32_t T = tileColumns(w.mat_cols[m]);
    const uint32_t c_lo = 4 * tileRowOfTile(t0, T), ncols = 4 * T - c_lo;  // columns the item's tiles touch: [c_lo, 4 T)
    switch (tileSubRows(ncols)) {
        case 126: pairTile2Item<126>(w, tile_lds, lt, m, chunk, t0, tcount, c_lo, ncols); break;
        case 94: pairTile2Item<94>(w, tile_lds, lt, m, chunk, t0, tcount, c_lo, ncols); break;
        case 46: pairTile2Item<46>(w, tile_lds, lt, m, chunk, t0, tcount, c_lo, ncols); break;
        case 30: pairTile2Item<30>(w, tile_lds, lt, m, chunk, t0, tcount, c_lo, ncols); break;
        case 14: pairTile2Item<14>(w, tile_lds, lt, m, chunk, t0, tcount, c_lo, ncols); break;
        case 6: pairTile2Item<6>(w, tile_lds, lt, m, chunk, t0, tcount, c_lo, ncols); break;
        default: pairTile2Item<2>(w, tile_lds, lt, m, chunk, t0, tcount, c_lo, ncols); break;
    }
}

// The tiles of a matrix cut into the ranges of its work items: 256 at a time, and what is left so that tiles x slices fills
// the workgroup — one more slice for as many tiles as fit then, the rest of the tiles in an item of their own (its lanes walk
// 1 / slices of the rows each) — whenever that walks at least a tenth fewer rows per lane than one item with the slices that fit.
inline void planTileRanges(const uint32_t tiles, std::vector<std::pair<uint32_t, uint32_t> > * ranges) {
    uint32_t t0 = 0, left = tiles;
    while (left > 0) {
        if (left >= kTileBlock) {
            ranges->emplace_back(t0, kTileBlock);
            t0 += kTileBlock;
            left -= kTileBlock;
            continue;
        }
        const uint32_t slices = kTileBlock / left;
        const uint32_t more = kTileBlock / (slices + 1), rest = left - more;
        const double one = 1.0 / slices, two = 1.0 / (slices + 1) + 1.0 / (kTileBlock / rest);
        if (two < 0.9 * one) {
            ranges->emplace_back(t0, more);
            t0 += more;
            left = rest;
        } else {
            ranges->emplace_back(t0, left);
            left = 0;
        }
    }
}

struct ResolveArgs {
    const uint32_t * big_matrix;    // [B] matrices on the table path
    uint32_t chunk_rows;            // rows of a chunk of the partial sums
    uint32_t count;
    const uint64_t * mat_rows;
    const uint32_t * mat_cols;
    const uint64_t * col_off;
    const uint32_t * col_count;
    const uint64_t * pair_cap_off;
    const uint64_t * big_col_part_off;
    const uint64_t * big_pair_part_off;
    const double * part_marginal;
    const double * part_optimistic;
    const double * part_pair;
    double min_log_likelihood_diff;
    double * log_freq;
    double * marginal;
    uint32_t * col_order;
    double * seq_value;     // [pair_cap] log-likelihood of every pair in the reference's sequence order
    uint32_t * out_first;
    uint32_t * out_second;
    double * out_value;
    uint32_t * out_count;
};

// first slot of row `pos` in the pair sequence of a matrix with G columns, and the row of a slot
__device__ __forceinline__ uint64_t sequenceRowStart(const uint32_t pos, const uint32_t G) {
    return static_cast<uint64_t>(pos) * G - (static_cast<uint64_t>(pos) * (pos > 0 ? pos - 1 : 0)) / 2;
}

__device__ __forceinline__ uint32_t sequenceRow(const uint64_t slot, const uint32_t G) {
    uint32_t lo = 0, hi = G - 1;  // largest row whose first slot is <= slot
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (sequenceRowStart(mid, G) <= slot) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// one workgroup per big matrix: marginal order, pair sequence, exclusive prefix-max filter, posteriors
__global__ __launch_bounds__(256) void resolveTableKernel(const ResolveArgs args) {
    constexpr int kBlock = 256;
    __shared__ double lds_red[kBlock / 64];
    __shared__ uint32_t lds_cnt[kBlock / 64];
    __shared__ unsigned long long lds_sum;
    if (blockIdx.x >= args.count) return;
    const uint32_t m = args.big_matrix[blockIdx.x];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t R = args.mat_rows[m];
    const uint32_t G = args.mat_cols[m];
    const uint32_t chunks = static_cast<uint32_t>((R + args.chunk_rows - 1) / args.chunk_rows);
    const uint64_t c0 = args.col_off[m];
    const uint32_t * ccount = args.col_count + c0;
    double * lf = args.log_freq + c0;
    double * marg = args.marginal + c0;
    uint32_t * ord = args.col_order + c0;
    const double * pm = args.part_marginal + args.big_col_part_off[m];
    const double * pp = args.part_pair + args.big_pair_part_off[m];
    const uint64_t p0 = args.pair_cap_off[m];
    double * seq = args.seq_value + p0;
    uint32_t * out_first = args.out_first + p0;
    uint32_t * out_second = args.out_second + p0;
    double * out_value = args.out_value + p0;
    const double thr = args.min_log_likelihood_diff;
    const double lowest = -1.7976931348623157e308;
    const double log_two = log(2.0);

    if (threadIdx.x == 0) lds_sum = 0;
    __syncthreads();
    unsigned long long part = 0;
    for (uint32_t g = threadIdx.x; g < G; g += kBlock) part += ccount[g];
    if (part) atomicAdd(&lds_sum, part);
    __syncthreads();
    const double count_sum = static_cast<double>(lds_sum);
    for (uint32_t g = threadIdx.x; g < G; g += kBlock) {
        const double f = log(ccount[g] / count_sum);
        lf[g] = f;
        double acc = 0.0;
        for (uint32_t c = 0; c < chunks; ++c) acc += pm[static_cast<uint64_t>(c) * G + g];
        marg[g] = (acc + f) + 0.0;
    }
    __syncthreads();
    {
        const double sum_log = blockLogSumExp<kBlock>(G, [&](uint32_t g) { return marg[g]; }, lds_red);
        for (uint32_t g = threadIdx.x; g < G; g += kBlock) marg[g] = exp(marg[g] - sum_log);
    }
    __syncthreads();
    for (uint32_t g = threadIdx.x; g < G; g += kBlock) {
        const double pg = marg[g];
        uint32_t rank = 0;
        for (uint32_t h = 0; h < G; ++h) {
            const double ph = marg[h];
            rank += (ph > pg || (ph == pg && h > g)) ? 1u : 0u;
        }
        ord[rank] = g;
    }
    __syncthreads();

    // pair sequence: (pos, j >= pos) row by row; s = pos*G - pos*(pos-1)/2 + (j - pos).  One thread per slot (the
    // chunk partials of a slot are dependent-latency loads: every thread of the block should have some in flight)
    const uint64_t S = static_cast<uint64_t>(G) * (G + 1) / 2;
    for (uint64_t slot = threadIdx.x; slot < S; slot += kBlock) {
        const uint32_t pos = sequenceRow(slot, G);
        const uint32_t j = pos + static_cast<uint32_t>(slot - sequenceRowStart(pos, G));
        const uint32_t a = ord[pos], b = ord[j];
        const uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
        double acc = 0.0;
        for (uint32_t c = 0; c < chunks; ++c) acc += pp[(static_cast<uint64_t>(c) * G + lo) * G + hi];
        seq[slot] = acc + ((lf[a] + lf[b]) + (a == b ? 0.0 : log_two));
    }
    __syncthreads();

    // exclusive prefix max over the sequence, keep flags, ordered compaction
    double carry_max = lowest;
    uint32_t kept = 0;
    for (uint64_t base = 0; base < S; base += kBlock) {
        const uint64_t s = base + threadIdx.x;
        const double v = s < S ? seq[s] : lowest;
        // inclusive max scan inside the wave
        double inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const double t = __shfl_up(inc, d, 64);
            if (lane >= d) inc = fmax(inc, t);
        }
        if (lane == 63) lds_red[wave] = inc;
        __syncthreads();
        double before = carry_max, total = carry_max;
#pragma unroll
        for (int w2 = 0; w2 < kBlock / 64; ++w2) {
            if (w2 < wave) before = fmax(before, lds_red[w2]);
            total = fmax(total, lds_red[w2]);
        }
        double excl = __shfl_up(inc, 1, 64);
        if (lane == 0) excl = lowest;
        excl = fmax(excl, before);
        const bool keep = (s < S) && !(v - excl < thr);
        // ordered compaction
        const unsigned long long ballot = __ballot(keep);
        const uint32_t in_wave = __popcll(ballot & ((1ull << lane) - 1ull));
        if (lane == 0) lds_cnt[wave] = __popcll(ballot);
        __syncthreads();
        uint32_t off = kept, tot = 0;
#pragma unroll
        for (int w2 = 0; w2 < kBlock / 64; ++w2) {
            if (w2 < wave) off += lds_cnt[w2];
            tot += lds_cnt[w2];
        }
        if (keep) {
            const uint32_t pos = sequenceRow(s, G);
            const uint32_t j = pos + static_cast<uint32_t>(s - sequenceRowStart(pos, G));
            out_first[off + in_wave] = ord[pos];
            out_second[off + in_wave] = ord[j];
            out_value[off + in_wave] = v;
        }
        kept += tot;
        carry_max = total;
        __syncthreads();
    }
    const double max_ll = carry_max;

    // late losers -> weight zero; log-sum-exp; posteriors
    __syncthreads();
    if (threadIdx.x == 0) args.out_count[m] = kept;
    {
        auto kept_value = [&](uint32_t k) { const double ll = out_value[k]; return (ll - max_ll < thr) ? lowest : ll; };
        const double sum_log = blockLogSumExp<kBlock>(kept, kept_value, lds_red);
        for (uint32_t k = threadIdx.x; k < kept; k += kBlock) out_value[k] = exp(kept_value(k) - sum_log);
    }
}

// exclusive prefix of the kept-pair counts (one workgroup: a batch has a few thousand matrices); also fetches the
// validity flag of the matrices' build and copies the tail words in front of the dense pairs, so that one D2H copy
// brings everything the host waits for
__global__ void __launch_bounds__(1024) pairOffsetsKernel(const uint32_t num_matrices, const uint32_t * __restrict__ counts,
                                                          uint64_t * __restrict__ pair_off, const uint32_t * __restrict__ build_error,
                                                          uint32_t * __restrict__ build_error_out, uint32_t * __restrict__ tail_copy,
                                                          const uint32_t tail_words) {
    __shared__ uint64_t sums[1024];
    const uint32_t per = (num_matrices + 1023) / 1024;
    const uint32_t lo = min(num_matrices, threadIdx.x * per), hi = min(num_matrices, lo + per);
    uint64_t mine = 0;
    for (uint32_t m = lo; m < hi; ++m) mine += counts[m];
    sums[threadIdx.x] = mine;
    __syncthreads();
    for (uint32_t step = 1; step < 1024; step <<= 1) {
        const uint64_t add = threadIdx.x >= step ? sums[threadIdx.x - step] : 0;
        __syncthreads();
        sums[threadIdx.x] += add;
        __syncthreads();
    }
    uint64_t run = sums[threadIdx.x] - mine;
    for (uint32_t m = lo; m < hi; ++m) {
        pair_off[m] = run;
        run += counts[m];
    }
    if (threadIdx.x == 1023) pair_off[num_matrices] = sums[1023];
    if (threadIdx.x == 0 && build_error) *build_error_out = *build_error;
    __syncthreads();
    // counts, evaluation counter and flag once more in front of the dense pairs: one copy brings both to the host
    for (uint32_t i = threadIdx.x; i < tail_words; i += blockDim.x) tail_copy[i] = counts[i];
}

// copies the kept pairs of every matrix into one dense block [value: total f64 | first: total u32 | second: total u32]
__global__ void compactPairsBlockKernel(const uint32_t num_matrices, const uint64_t * __restrict__ pair_cap_off,
                                        const uint64_t * __restrict__ pair_off, const uint32_t * __restrict__ in_first,
                                        const uint32_t * __restrict__ in_second, const double * __restrict__ in_value,
                                        unsigned char * __restrict__ block) {
    const uint32_t m = blockIdx.x;
    if (m >= num_matrices) return;
    const uint64_t total = pair_off[num_matrices];
    double * out_value = reinterpret_cast<double *>(block);
    uint32_t * out_first = reinterpret_cast<uint32_t *>(block + total * sizeof(double));
    uint32_t * out_second = out_first + total;
    const uint64_t src = pair_cap_off[m], dst = pair_off[m], n = pair_off[m + 1] - pair_off[m];
    for (uint64_t k = threadIdx.x; k < n; k += blockDim.x) {
        out_first[dst + k] = in_first[src + k];
        out_second[dst + k] = in_second[src + k];
        out_value[dst + k] = in_value[src + k];
    }
}

}  // namespace

namespace rpvg_hip_detail {

// The diploid search of every matrix of `groups`, queued on the context's streams (no host synchronisation): afterwards
// w.d_tail[m] holds the number of kept pairs of matrix m and w.d_out_first / second / value, at w.d_pair_cap_off[m], the
// pairs in the order the reference keeps them.  The caller holds ctx->mutex and calls searchGateLeave when what it
// queues behind the search may run next to another context's search.  (rpvg_hip_bounded_pair_posteriors, below, brings
// the pairs to the host; subset_em.hip consumes them on the device.)
int queuePairSearch(rpvg_hip_ctx * ctx, const rpvg_hip_groups * groups, const uint32_t * column_counts, const double min_rel_likelihood,
                    PairSearchWork & w) {
    std::unique_ptr<HostScope> scope(new HostScope("bounded search: host order + work items"));
    const uint32_t M = groups->num_matrices;
    w.M = M;
    std::vector<uint64_t> & col_off = w.col_off, & pair_cap_off = w.pair_cap_off;
    col_off.assign(M + 1, 0);
    pair_cap_off.assign(M + 1, 0);
    for (uint32_t m = 0; m < M; ++m) {
        const uint64_t G = groups->h_num_cols[m];
        col_off[m + 1] = col_off[m] + G;
        pair_cap_off[m + 1] = pair_cap_off[m] + G * (G + 1) / 2;
    }
    // (column_counts == NULL: the matrices were built from the batch's own haplotype columns, whose multiplicities — at least one
    // haplotype each — are on the device already)
    if (!column_counts && M > 0 && !groups->d_column_counts) {
        setError("rpvg_hip_bounded_pair_posteriors: column_counts is NULL");
        return RPVG_HIP_ERR_INVALID;
    }
    for (uint64_t c = 0; column_counts && c < col_off[M]; ++c) {
        if (column_counts[c] == 0) {
            setError("rpvg_hip_bounded_pair_posteriors: column %llu has a zero count", static_cast<unsigned long long>(c));
            return RPVG_HIP_ERR_INVALID;
        }
    }
    // expensive matrices first
    std::vector<uint32_t> & order = w.order;
    order.assign(M, 0);
    std::iota(order.begin(), order.end(), 0);
    std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
        const double wx = static_cast<double>(groups->h_num_rows[x]) * groups->h_num_cols[x] * groups->h_num_cols[x];
        const double wy = static_cast<double>(groups->h_num_rows[y]) * groups->h_num_cols[y] * groups->h_num_cols[y];
        return wx != wy ? wx > wy : x < y;
    });

    // Big matrices (leading part of `order`) take the table path: every pair evaluated in parallel by
    // (column, row chunk) workgroups, then one resolving workgroup; the rest take the in-workgroup search.
    const uint64_t table_budget = 1ull << 28;  // doubles of partial pair sums (2 GiB)
    // rows x columns from which a matrix takes the table path (RPVG_HIP_TABLE_MIN_WORK overrides; tests use 0)
    double table_min_work = 65536.0;
    if (const char * env = std::getenv("RPVG_HIP_TABLE_MIN_WORK")) table_min_work = std::atof(env);
    if (min_rel_likelihood > 1) table_min_work = 1e300;  // positive threshold: the prefix-maximum form of the rule does not hold
    // every pair of every matrix from LDS-staged rows, a tile of pairs per lane (pairTileKernel): the default for a
    // threshold that is a ratio <= 1; RPVG_HIP_PAIR_TILES=0 keeps the sequential search with its table path (A/B)
    const char * tiles_env = std::getenv("RPVG_HIP_PAIR_TILES");  // (read per call: the tests switch between the two searches)
    const int tiles_wanted = tiles_env ? std::atoi(tiles_env) : 2;
    const bool pair_tiles = tiles_wanted != 0 && min_rel_likelihood <= 1;
    const bool tile_ranges = pair_tiles;  // (round 2's tile kernel, one item per chunk, went with round 5: pairTile2Kernel has two rounds of sweeps behind it)
    // rows of a work item's chunk (RPVG_HIP_PAIR_CHUNK_ROWS: the tests cut small matrices into several chunks with it)
    const uint32_t chunk_rows = tile_ranges && std::getenv("RPVG_HIP_PAIR_CHUNK_ROWS") ? std::max(256, std::atoi(std::getenv("RPVG_HIP_PAIR_CHUNK_ROWS"))) : kChunkRows;
    if (pair_tiles) table_min_work = 0.0;
    const uint32_t tile_step = kTileA;
    uint32_t & num_big = w.num_big;
    num_big = 0;
    std::vector<uint64_t> big_col_part_off(M, 0), big_pair_part_off(M, 0);
    std::vector<uint32_t> item_matrix, item_col, item_chunk;  // (item_col: with tile ranges, first tile | (tiles - 1) << 16)
    std::vector<std::pair<uint32_t, uint32_t> > ranges;
    uint64_t col_part_total = 0, pair_part_total = 0;
    {
        std::vector<uint32_t> table_matrices, others;
        bool table_closed = false;  // (the sequential kernels' table path: a prefix of the order, as before)
        for (uint32_t i = 0; i < M; ++i) {
            const uint32_t m = order[i];
            const uint64_t R = groups->h_num_rows[m], G = groups->h_num_cols[m];
            const uint64_t chunks = (R + chunk_rows - 1) / chunk_rows;
            const uint64_t parts = chunks;
            const bool fits = pair_part_total + parts * G * G <= table_budget;
            const bool takes_table = pair_tiles ? (G <= kTileMaxColumns && fits)
                                                : (!table_closed && static_cast<double>(R) * G >= table_min_work && fits);
            if (!takes_table) {
                table_closed = true;
                others.push_back(m);
                continue;
            }
            table_matrices.push_back(m);
            big_col_part_off[m] = col_part_total;
            big_pair_part_off[m] = pair_part_total;
            col_part_total += parts * G;
            pair_part_total += parts * G * G;
            if (tile_ranges) {
                ranges.clear();
                planTileRanges(tileCount(static_cast<uint32_t>(G)), &ranges);
                for (uint32_t c = 0; c < chunks; ++c) {
                    for (auto & range: ranges) {
                        item_matrix.push_back(m);
                        item_col.push_back(range.first | ((range.second - 1) << 16));
                        item_chunk.push_back(c);
                    }
                }
                continue;
            }
            for (uint32_t c = 0; c < chunks; ++c) {
                for (uint32_t a = 0; a < (pair_tiles ? 1u : G); a += tile_step) {
                    item_matrix.push_back(m);
                    item_col.push_back(a);
                    item_chunk.push_back(c);
                }
            }
        }
        num_big = static_cast<uint32_t>(table_matrices.size());
        std::copy(table_matrices.begin(), table_matrices.end(), order.begin());
        std::copy(others.begin(), others.end(), order.begin() + num_big);
    }

    scope.reset(new HostScope("bounded search: uploads + launches"));
    hipError_t e = hipSetDevice(ctx->device);
    hipStream_t st = ctx->stream;
    auto ok = [&](hipError_t r) { if (e == hipSuccess) e = r; return e == hipSuccess; };

    uint32_t num_medium = 0;
    {
        std::vector<uint32_t> medium, small;
        for (uint32_t i = num_big; i < M; ++i) (groups->h_num_rows[order[i]] > kSmallRows ? medium : small).push_back(order[i]);
        num_medium = static_cast<uint32_t>(medium.size());
        std::copy(medium.begin(), medium.end(), order.begin() + num_big);
        std::copy(small.begin(), small.end(), order.begin() + num_big + num_medium);
    }

    if (RPVG_EXPERIMENT_ENV("RPVG_HIP_SEARCH_CLASSES")) {
        auto show = [&](const char * name, uint32_t first, uint32_t count) {
            double evals = 0;
            for (uint32_t i = first; i < first + count; ++i) evals += 0.5 * groups->h_num_rows[order[i]] * groups->h_num_cols[order[i]] * groups->h_num_cols[order[i]];
            std::fprintf(stderr, "[search classes] %-6s %6u matrices, %.3g pair-row evaluations if nothing is skipped; largest:", name, count, evals);
            for (uint32_t i = first; i < first + std::min<uint32_t>(count, 6); ++i) std::fprintf(stderr, " %llux%u", static_cast<unsigned long long>(groups->h_num_rows[order[i]]), groups->h_num_cols[order[i]]);
            std::fprintf(stderr, "\n");
        };
        show("table", 0, num_big);
        if (pair_tiles) {  // the tile kernel's lanes: how many of a workgroup's 256 carry a tile, how many of a tile's 16 slots a pair
            struct Bucket { uint64_t matrices = 0; double rows = 0, evals = 0, lane_rows = 0, slot_rows = 0; };
            std::map<uint32_t, Bucket> buckets;
            for (uint32_t i = 0; i < num_big; ++i) {
                const double R = static_cast<double>(groups->h_num_rows[order[i]]);
                const uint32_t G = groups->h_num_cols[order[i]];
                uint32_t key = 1;
                while (key < G) key <<= 1;
                Bucket & b = buckets[key];
                const uint32_t tiles = tileCount(G);
                b.matrices += 1;
                b.rows += R;
                b.evals += R * (0.5 * G * (G + 1));
                // rows a lane walks x lanes of the workgroup, summed over the matrix's work items (planTileRanges: an item's lanes walk
                // 1 / slices of the rows each) — until round 6 this line priced round 2's kernel, one item per chunk with
                // tileSlices(G) slices: 0.58 for 64 columns, where the ranges of pairTile2Kernel reach 1.00
                std::vector<std::pair<uint32_t, uint32_t> > ranges;
                planTileRanges(tiles, &ranges);
                for (auto & range : ranges) b.lane_rows += R / (kTileBlock / range.second) * kTileBlock;
                b.slot_rows += R * tiles * 16;
            }
            for (auto & kv : buckets) {
                const Bucket & b = kv.second;
                std::fprintf(stderr, "[search classes]   columns <= %4u: %6llu matrices, %9.0f rows, %.3g evaluations, pairs / tile slots %.2f, tile slots / lane slots %.2f\n",
                             kv.first, static_cast<unsigned long long>(b.matrices), b.rows, b.evals, b.evals / b.slot_rows, b.slot_rows / (16.0 * b.lane_rows));
            }
        }
        show("medium", num_big, num_medium);
        show("small", num_big + num_medium, M - num_big - num_medium);
    }

    auto & d_order = w.d_order; auto & d_col_count = w.d_col_count; auto & d_col_order = w.d_col_order;
    auto & d_out_first = w.d_out_first; auto & d_out_second = w.d_out_second;
    auto & d_col_off = w.d_col_off; auto & d_pair_cap_off = w.d_pair_cap_off;
    auto & d_lf = w.d_lf; auto & d_marg = w.d_marg; auto & d_opt_raw = w.d_opt_raw; auto & d_opt = w.d_opt; auto & d_out_value = w.d_out_value;

    // every host array of the search in one block, one copy (UploadPack: a command per array was 1 ms per search)
    UploadPack & pack = w.pack;
    auto & d_item_matrix = w.d_item_matrix; auto & d_item_col = w.d_item_col; auto & d_item_chunk = w.d_item_chunk;
    auto & d_big_col_part_off = w.d_big_col_part_off; auto & d_big_pair_part_off = w.d_big_pair_part_off;
    pack.add(d_order, order.data(), M);
    pack.add(d_col_off, col_off.data(), M + 1);
    pack.add(d_pair_cap_off, pair_cap_off.data(), M + 1);
    if (column_counts) pack.add(d_col_count, column_counts, col_off[M]);
    if (num_big > 0) {
        pack.add(d_item_matrix, item_matrix.data(), item_matrix.size());
        pack.add(d_item_col, item_col.data(), item_col.size());
        pack.add(d_item_chunk, item_chunk.data(), item_chunk.size());
        pack.add(d_big_col_part_off, big_col_part_off.data(), M);
        pack.add(d_big_pair_part_off, big_pair_part_off.data(), M);
    }
    // [kept pairs per matrix: M words | evaluation counter: 2 words | validity flag of the build | -]: what the host reads first
    const uint32_t evals_word = (M + 1) & ~1u, tail_words = evals_word + 4;
    w.evals_word = evals_word;
    w.tail_words = tail_words;
    auto & d_tail = w.d_tail;
    pack.addZero(d_tail, tail_words);
    if (w.extra_u64_count) pack.add(w.d_extra_u64, w.extra_u64, w.extra_u64_count);
    if (w.extra_zero_bytes) pack.addZero(w.d_extra_zero, w.extra_zero_bytes);
    int span = ctx->spanBegin(FAM_H2D);
    ok(pack.commit(st));
    if (!column_counts) d_col_count.borrow(const_cast<uint32_t *>(groups->d_column_counts), col_off[M]);
    ctx->spanEnd(span);
    ctx->stats.h2d_bytes += static_cast<double>(M * 20 + col_off[M] * 4);
    ok(d_lf.alloc(col_off[M]));
    ok(d_marg.alloc(col_off[M]));
    ok(d_opt_raw.alloc(col_off[M]));
    ok(d_opt.alloc(col_off[M]));
    ok(d_col_order.alloc(col_off[M]));
    ok(d_out_first.alloc(pair_cap_off[M]));
    ok(d_out_second.alloc(pair_cap_off[M]));
    ok(d_out_value.alloc(pair_cap_off[M]));
    if (e != hipSuccess) {
        setError("rpvg_hip_bounded_pair_posteriors: %s", hipGetErrorString(e));
        return (e == hipErrorOutOfMemory) ? RPVG_HIP_ERR_ALLOC : RPVG_HIP_ERR_RUNTIME;
    }

    SearchArgs args;
    args.order = d_order.ptr;
    args.count = M;
    args.mat_val_off = groups->mat_val_off.ptr;
    args.mat_row_off = groups->mat_row_off.ptr;
    args.mat_fast = groups->mat_fast.ptr;
    args.mat_mid = groups->mat_mid.ptr;
    args.mat_rows = groups->mat_rows.ptr;
    args.mat_cols = groups->mat_cols.ptr;
    args.values = groups->values.ptr;
    args.rowmax = groups->rowmax.ptr;
    args.row_count = groups->row_count.ptr;
    args.row_noise = groups->row_noise.ptr;
    args.col_off = d_col_off.ptr;
    args.col_count = d_col_count.ptr;
    args.pair_cap_off = d_pair_cap_off.ptr;
    args.min_log_likelihood_diff = std::log(min_rel_likelihood);
    args.log_freq = d_lf.ptr;
    args.marginal = d_marg.ptr;
    args.optimistic_raw = d_opt_raw.ptr;
    args.optimistic = d_opt.ptr;
    args.col_order = d_col_order.ptr;
    args.out_first = d_out_first.ptr;
    args.out_second = d_out_second.ptr;
    args.out_value = d_out_value.ptr;
    args.out_count = d_tail.ptr;

    args.log_evals = reinterpret_cast<unsigned long long *>(d_tail.ptr + evals_word);
    auto & d_part_marg = w.d_part_marg; auto & d_part_opt = w.d_part_opt; auto & d_part_pair = w.d_part_pair; auto & d_seq = w.d_seq;
    if (num_big > 0) {
        ok(d_part_marg.alloc(col_part_total));
        ok(d_part_opt.alloc(col_part_total));
        ok(d_part_pair.alloc(pair_part_total));
        ok(d_seq.alloc(pair_cap_off[M]));
    }
    if (e != hipSuccess) {
        setError("rpvg_hip_bounded_pair_posteriors: %s", hipGetErrorString(e));
        return (e == hipErrorOutOfMemory) ? RPVG_HIP_ERR_ALLOC : RPVG_HIP_ERR_RUNTIME;
    }

    // the table path, the medium and the small matrices are independent: three streams, so that the
    // tail of one does not idle the GPU
    // streams of the medium / small kernels next to the table path on `st` (RPVG_HIP_SEARCH_STREAMS: A/B knob, 3 = one
    // stream each, 2 = medium and small share one, 1 = everything on st)
    static const int search_streams = []() {
        const char * env = RPVG_EXPERIMENT_ENV("RPVG_HIP_SEARCH_STREAMS");
        return env ? std::max(1, std::min(3, std::atoi(env))) : 3;
    }();
    hipStream_t s_medium = search_streams == 1 ? st : ctx->aux[0];
    hipStream_t s_small = search_streams == 1 ? st : (search_streams == 2 ? ctx->aux[0] : ctx->aux[1]);
    // The row collapse of the matrices runs on a stream of its own behind their build (rpvg_hip_groups_build): the uploads
    // above did not wait for it, the kernels do.  (Searching the matrices as built and once more the few the collapse
    // replayed was tried: its small kernels then wait for slots next to the search's large one, no gain.)
    const bool side_kernels = M > num_big;  // the sequential kernels on the aux streams
    // Matrices whose collapse holds its last stage back (rpvg_hip_groups::held_back_runs), all of them on the table path: the tile
    // kernel reads them as built, at once — the twenty launches that find the collapse's runs (2 ms of a batch's chain of
    // kernels, for some hundred rows of three million) run beside it —, and the stage that rewrites rows comes behind both and
    // adjusts the tile kernel's sums for them.  3.8 against 4.6 ms per configs[2] batch in the pipeline.
    const bool runs_behind_search = static_cast<bool>(groups->held_back_runs) && pair_tiles && num_big == M;
    if (!runs_behind_search) {
        ok(groups->waitCollapse(st));
        if (groups->collapse_done && !searchLaunchesEarly()) {  // (see searchGateEnter: the thread waits, not the stream's queue)
            HostScope wait_scope("search: wait for the collapse of the matrices");
            ok(waitEvent(groups->collapse_done));
        }
    }
    span = ctx->spanBegin(FAM_LOGLIK);
    searchGateEnter(ctx, st, std::accumulate(groups->h_num_rows.begin(), groups->h_num_rows.end(), uint64_t(0)));
    if (side_kernels) ok(ctx->forkAux());
    if (num_big > 0) {
        TableWork tw;
        tw.item_matrix = d_item_matrix.ptr;
        tw.item_col = d_item_col.ptr;
        tw.item_chunk = d_item_chunk.ptr;
        tw.count = static_cast<uint32_t>(item_matrix.size());
        tw.mat_val_off = groups->mat_val_off.ptr;
        tw.mat_row_off = groups->mat_row_off.ptr;
        tw.mat_fast = groups->mat_fast.ptr;
        tw.mat_mid = groups->mat_mid.ptr;
        tw.mat_rows = groups->mat_rows.ptr;
        tw.mat_cols = groups->mat_cols.ptr;
        tw.values = groups->values.ptr;
        tw.rowmax = groups->rowmax.ptr;
        tw.row_count = groups->row_count.ptr;
        tw.row_noise = groups->row_noise.ptr;
        tw.big_col_part_off = d_big_col_part_off.ptr;
        tw.big_pair_part_off = d_big_pair_part_off.ptr;
        tw.part_marginal = d_part_marg.ptr;
        tw.part_optimistic = d_part_opt.ptr;
        tw.part_pair = d_part_pair.ptr;
        tw.log_evals = args.log_evals;
        if (pair_tiles) {
            PairTileWork pw;
            pw.item_matrix = d_item_matrix.ptr;
            pw.item_chunk = d_item_chunk.ptr;
            pw.item_tiles = d_item_col.ptr;
            pw.chunk_rows = chunk_rows;
            pw.count = tw.count;
            pw.mat_val_off = groups->mat_val_off.ptr;
            pw.mat_row_off = groups->mat_row_off.ptr;
            pw.mat_fast = groups->mat_fast.ptr;
            pw.mat_mid = groups->mat_mid.ptr;
            pw.mat_rows = groups->mat_rows.ptr;
            pw.mat_cols = groups->mat_cols.ptr;
            pw.values = groups->values.ptr;
            pw.row_count = groups->row_count.ptr;
            pw.row_noise = groups->row_noise.ptr;
            pw.col_part_off = d_big_col_part_off.ptr;
            pw.pair_part_off = d_big_pair_part_off.ptr;
            pw.part_marginal = d_part_marg.ptr;
            pw.part_pair = d_part_pair.ptr;
            pw.log_evals = args.log_evals;
#ifdef RPVG_HIP_EXPERIMENTS
            pw.debug_skip = RPVG_EXPERIMENT_ENV("RPVG_HIP_PAIR_DEBUG") ? static_cast<uint32_t>(std::atoi(RPVG_EXPERIMENT_ENV("RPVG_HIP_PAIR_DEBUG"))) : 0u;
#endif
            // A/B knob RPVG_HIP_PAIR_LDS_KB: more dynamic LDS than the kernel uses = fewer workgroups per CU (64: two instead of
            // three — registers left over for the other lane's kernels while this one runs)
            static const size_t tile_lds_bytes = []() {
                const char * env = RPVG_EXPERIMENT_ENV("RPVG_HIP_PAIR_LDS_KB");
                return std::max<size_t>(kTileLdsDoubles * sizeof(double), env ? static_cast<size_t>(std::atoi(env)) * 1024 : 0);
            }();
            if (tile_lds_bytes > 64 * 1024) {
                RPVG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&pairTile2Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(tile_lds_bytes)));
            }
            // (On the device's lowest stream priority — the slots its workgroups free going to the other lane's short kernels first —
            // the batch took 11.6 against 10.5 ms: the search is itself on its lane's critical path.  Two passes around the replay of
            // the matrices' collapse — the matrices it leaves alone as soon as its first stages have told them apart, the others when
            // it is done — 10.4 against 9.9 ms: the second pass's grid of mostly empty workgroups, and the replay's small kernels next
            // to the lane's own tile kernel.)
            const int tile_span = ctx->spanBegin(FAM_TILE, st);  // (the kernel alone: rpvg_hip_kernel_stats::search_tile_ms)
            pairTile2Kernel<<<dim3(pw.count), dim3(kTileBlock), tile_lds_bytes, st>>>(pw);
            ctx->spanEnd(tile_span);
#ifdef RPVG_HIP_EXPERIMENTS
            // (RPVG_HIP_PAIR_REPEAT=n: the same launch n more times — what a batch pays per millisecond of this kernel)
            for (int k = RPVG_EXPERIMENT_ENV("RPVG_HIP_PAIR_REPEAT") ? std::atoi(RPVG_EXPERIMENT_ENV("RPVG_HIP_PAIR_REPEAT")) : 0; k > 0; --k) {
                pairTile2Kernel<<<dim3(pw.count), dim3(kTileBlock), tile_lds_bytes, st>>>(pw);
            }
#endif
        } else {
            pairTableKernel<<<dim3(((tw.count + 7) / 8) * 8), dim3(256), 0, st>>>(tw);
        }
        if (runs_behind_search && e == hipSuccess) {
            if (groups->collapse_done) {
                HostScope wait_scope("search: wait for the runs of the matrices' collapse");
                ok(searchLaunchesEarly() ? hipStreamWaitEvent(st, groups->collapse_done, 0) : waitEvent(groups->collapse_done));
            }
            rpvg_hip_groups::SearchSums sums;
            sums.part_pair = d_part_pair.ptr;
            sums.part_marginal = d_part_marg.ptr;
            sums.pair_part_off = d_big_pair_part_off.ptr;
            sums.col_part_off = d_big_col_part_off.ptr;
            sums.chunk_rows = chunk_rows;
            ok(groups->held_back_runs(st, &sums));
            groups->held_back_runs = nullptr;
        }

        ResolveArgs ra;
        ra.big_matrix = d_order.ptr;
        ra.count = num_big;
        ra.mat_rows = groups->mat_rows.ptr;
        ra.mat_cols = groups->mat_cols.ptr;
        ra.chunk_rows = chunk_rows;
        ra.col_off = d_col_off.ptr;
        ra.col_count = d_col_count.ptr;
        ra.pair_cap_off = d_pair_cap_off.ptr;
        ra.big_col_part_off = d_big_col_part_off.ptr;
        ra.big_pair_part_off = d_big_pair_part_off.ptr;
        ra.part_marginal = d_part_marg.ptr;
        ra.part_optimistic = d_part_opt.ptr;
        ra.part_pair = d_part_pair.ptr;
        ra.min_log_likelihood_diff = args.min_log_likelihood_diff;
        ra.log_freq = d_lf.ptr;
        ra.marginal = d_marg.ptr;
        ra.col_order = d_col_order.ptr;
        ra.seq_value = d_seq.ptr;
        ra.out_first = d_out_first.ptr;
        ra.out_second = d_out_second.ptr;
        ra.out_value = d_out_value.ptr;
        ra.out_count = d_tail.ptr;
        resolveTableKernel<<<dim3(num_big), dim3(256), 0, st>>>(ra);
    }
    // the rest walk the search inside one workgroup; matrices with few rows stage less LDS (more
    // workgroups per CU).  `order` is [big | medium | small], each part expensive first.
    args.row_lds_cols = kRowLdsCols;
    auto search_lds_bytes = [](uint32_t stage_rows, uint32_t row_cols) { return static_cast<size_t>((kTileFirst + 1) * stage_rows + kTileFirst * row_cols) * sizeof(double); };
    {
        static std::once_flag once;  // 96 KB of dynamic LDS: above the 64 KB a kernel gets without asking
        std::call_once(once, [&]() {
            (void) hipFuncSetAttribute(reinterpret_cast<const void *>(&boundedSearchKernel<1024, 64>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(search_lds_bytes(kLdsRows, kRowLdsCols)));
        });
    }
    if (num_medium > 0) {
        args.order = d_order.ptr + num_big;
        args.count = num_medium;
        args.stage_rows = kLdsRows;
        boundedSearchKernel<1024, 64><<<dim3(num_medium), dim3(1024), search_lds_bytes(kLdsRows, kRowLdsCols), s_medium>>>(args);
    }
    if (M > num_big + num_medium) {
        args.order = d_order.ptr + num_big + num_medium;
        args.count = M - num_big - num_medium;
        args.stage_rows = kSmallRows;
        args.row_lds_cols = kSmallRowLdsCols;
        boundedSearchKernel<256, 16><<<dim3(args.count), dim3(256), search_lds_bytes(kSmallRows, kSmallRowLdsCols), s_small>>>(args);
    }
    if (side_kernels) ok(ctx->joinAux());
    ctx->spanEnd(span);
    ctx->stats.loglik_launches += (num_big > 0 ? 2 : 0) + (num_medium > 0) + (M > num_big + num_medium);
    ok(hipGetLastError());
    if (e != hipSuccess) {
        setError("rpvg_hip_bounded_pair_posteriors: %s", hipGetErrorString(e));
        (void) hipStreamSynchronize(st);
        return RPVG_HIP_ERR_RUNTIME;
    }
    return RPVG_HIP_OK;
}

void leavePairSearch(rpvg_hip_ctx * ctx) { searchGateLeave(ctx, ctx->stream); }

void accountPairSearch(rpvg_hip_ctx * ctx, const rpvg_hip_groups * groups, const PairSearchWork & w, const unsigned long long log_evals,
                       const uint64_t kept_pairs) {
    ctx->stats.loglik_evals += static_cast<double>(log_evals);  // counted by the kernels
    for (uint32_t i = 0; i < w.M; ++i) {
        const double G = groups->h_num_cols[w.order[i]];
        ctx->stats.search_pairs_possible += G * (G + 1) / 2;
        if (i < w.num_big) ctx->stats.search_pairs_table += G * (G + 1) / 2;
    }
    ctx->stats.search_pairs_kept += static_cast<double>(kept_pairs);
}

}  // namespace rpvg_hip_detail

extern "C" int rpvg_hip_bounded_pair_posteriors(rpvg_hip_ctx * ctx, const rpvg_hip_groups * groups,
                                                const uint32_t * column_counts, double min_rel_likelihood,
                                                rpvg_hip_pair_posteriors ** result_out) {
    RPVG_REQUIRE(ctx && groups && result_out, "rpvg_hip_bounded_pair_posteriors: NULL argument");
    *result_out = nullptr;
    RPVG_REQUIRE(min_rel_likelihood > 0, "rpvg_hip_bounded_pair_posteriors: min_rel_likelihood must be positive");
    const uint32_t M = groups->num_matrices;
    rpvg_hip_pair_posteriors * res = new (std::nothrow) rpvg_hip_pair_posteriors();
    if (!res) {
        setError("rpvg_hip_bounded_pair_posteriors: out of host memory");
        return RPVG_HIP_ERR_ALLOC;
    }
    res->pair_off.assign(M + 1, 0);
    if (M == 0) {
        *result_out = res;
        return RPVG_HIP_OK;
    }
    if (!column_counts && !groups->d_column_counts) {
        delete res;
        setError("rpvg_hip_bounded_pair_posteriors: column_counts is NULL");
        return RPVG_HIP_ERR_INVALID;
    }

    std::lock_guard<std::mutex> lock(ctx->mutex);
    PairSearchWork w;
    {
        const int rc = queuePairSearch(ctx, groups, column_counts, min_rel_likelihood, w);
        if (rc != RPVG_HIP_OK) {
            delete res;
            return rc;
        }
    }
    hipError_t e = hipSuccess;
    hipStream_t st = ctx->stream;
    auto ok = [&](hipError_t r) { if (e == hipSuccess) e = r; return e == hipSuccess; };
    std::unique_ptr<HostScope> scope;
    const std::vector<uint64_t> & pair_cap_off = w.pair_cap_off;
    const uint32_t evals_word = w.evals_word, tail_words = w.tail_words;
    auto & d_tail = w.d_tail; auto & d_pair_cap_off = w.d_pair_cap_off; auto & d_out_first = w.d_out_first;
    auto & d_out_second = w.d_out_second; auto & d_out_value = w.d_out_value;
    DeviceBuffer<uint64_t> d_pair_off;

    // Offsets of the kept pairs and their dense copy are queued behind the search, and ONE copy brings the host the counts
    // (with the evaluation counter and the validity flag of the matrices' build) and the first megabyte of the dense
    // pairs behind them — all of them, normally (20 000 pairs of a bench batch are 0.3 MB); more pairs take a second copy.
    const char * with_counts_env = std::getenv("RPVG_HIP_PAIRS_WITH_COUNTS");  // bytes (tests: 0 = always a second copy)
    const size_t kPairsWithTheCounts = with_counts_env ? static_cast<size_t>(std::atof(with_counts_env)) : (size_t(1) << 20);
    const size_t tail_bytes = tail_words * sizeof(uint32_t), tail_room = (tail_bytes + 255) & ~size_t(255);
    const size_t block_capacity = static_cast<size_t>(pair_cap_off[M]) * 16;
    const size_t first_copy = tail_room + std::min(block_capacity, kPairsWithTheCounts);
    DeviceBuffer<unsigned char> d_result;  // [tail words | dense pairs: value f64 x n, first u32 x n, second u32 x n]
    ok(d_pair_off.alloc(M + 1));
    ok(d_result.alloc(tail_room + block_capacity));
    void * host_result = nullptr;
    if (e == hipSuccess && pinnedAlloc(&host_result, first_copy) != hipSuccess) e = hipErrorOutOfMemory;
    if (e == hipSuccess) {
        const bool check_build = !groups->build_checked && groups->build_error_flag.ptr;
        pairOffsetsKernel<<<dim3(1), dim3(1024), 0, st>>>(M, d_tail.ptr, d_pair_off.ptr, check_build ? groups->build_error_flag.ptr : nullptr,
                                                         d_tail.ptr + evals_word + 2, reinterpret_cast<uint32_t *>(d_result.ptr), tail_words);
        compactPairsBlockKernel<<<dim3(M), dim3(64), 0, st>>>(M, d_pair_cap_off.ptr, d_pair_off.ptr, d_out_first.ptr, d_out_second.ptr,
                                                             d_out_value.ptr, d_result.ptr + tail_room);
        ok(hipGetLastError());
        ok(hipMemcpyAsync(host_result, d_result.ptr, first_copy, hipMemcpyDeviceToHost, st));
    }
    // the next search (another lane's) starts behind this one's offsets and compaction, not behind its large kernel alone:
    // two tiny kernels that would otherwise wait for slots next to that search (0.5 ms before this lane saw its results)
    searchGateLeave(ctx, st);
    scope.reset(new HostScope("bounded search: wait for the kernels"));
    ok(waitStream(st));
    if (e == hipSuccess) {
        const uint32_t * counts = static_cast<const uint32_t *>(host_result);
        unsigned long long log_evals = 0;
        std::memcpy(&log_evals, counts + evals_word, sizeof(log_evals));
        const uint32_t build_bad = counts[evals_word + 2];
        for (uint32_t m = 0; m < M; ++m) res->pair_off[m + 1] = res->pair_off[m] + counts[m];
        if (build_bad) {  // the matrices were built without a host sync
            pinnedFree(host_result);
            delete res;
            setError(build_bad == 2 ? "rpvg_hip_groups_build: a group lists a path twice" : "rpvg_hip_groups_build: a group refers to a path outside its cluster");
            return RPVG_HIP_ERR_INVALID;
        }
        groups->build_checked = true;
        const uint64_t total = res->pair_off[M];
        const unsigned char * pairs = nullptr;
        if (total * 16 <= first_copy - tail_room) {  // they came with the counts: the result keeps the block
            res->block = host_result;
            res->block_pinned = true;
            host_result = nullptr;
            pairs = static_cast<const unsigned char *>(res->block) + tail_room;
        } else {
            scope.reset(new HostScope("bounded search: download pairs"));
            if (pinnedAlloc(&res->block, total * 16) == hipSuccess) {
                res->block_pinned = true;
            } else {
                res->block = std::malloc(total * 16);
                if (!res->block) e = hipErrorOutOfMemory;
            }
            if (e == hipSuccess) {
                ok(hipMemcpyAsync(res->block, d_result.ptr + tail_room, total * 16, hipMemcpyDeviceToHost, st));
                ok(waitStream(st));
                pairs = static_cast<const unsigned char *>(res->block);
            }
        }
        if (e == hipSuccess) {
            res->posterior = reinterpret_cast<const double *>(pairs);
            res->first = reinterpret_cast<const uint32_t *>(pairs + total * sizeof(double));
            res->second = res->first + total;
        }
        accountPairSearch(ctx, groups, w, log_evals, total);
    }
    if (host_result) pinnedFree(host_result);
    if (e != hipSuccess) {
        delete res;
        setError("rpvg_hip_bounded_pair_posteriors: %s", hipGetErrorString(e));
        return (e == hipErrorOutOfMemory) ? RPVG_HIP_ERR_ALLOC : RPVG_HIP_ERR_RUNTIME;
    }
    *result_out = res;
    return RPVG_HIP_OK;
}

extern "C" int rpvg_hip_pair_posteriors_get(const rpvg_hip_pair_posteriors * result, rpvg_hip_pair_posteriors_view * view_out) {
    RPVG_REQUIRE(result && view_out, "rpvg_hip_pair_posteriors_get: NULL argument");
    view_out->num_matrices = static_cast<uint32_t>(result->pair_off.size() - 1);
    view_out->pair_off = result->pair_off.data();
    view_out->first = result->first;
    view_out->second = result->second;
    view_out->posterior = result->posterior;
    return RPVG_HIP_OK;
}

extern "C" void rpvg_hip_pair_posteriors_free(rpvg_hip_pair_posteriors * result) { delete result; }
