// readCollapseProbabilityMatrix (src/path_estimator.cpp:197-259, comparator :13-31) replayed on the group matrices
// (callers src/path_abundance_estimator.cpp:380,443) on the GPU (gfx950).
//
// The reference sorts the rows of the normalised matrix (group columns, then the noise column, then the read
// count; every comparison tolerant, Utils::doubleCompare) and merges every row that lies within prob_precision of
// the head of its run — in all columns, absolutely — into that head: the head keeps its values, the counts add up.
// The consumers of a group matrix are sums over rows of count * log(noise + columns), so "merged into the head" is
// "takes the values of the head" with the row's own count: the rows stay where they are (their class order,
// LogProduct, is untouched) and only values move.
//
// Such rows are not rare: a read whose worst candidate path falls below prob_precision moves that mass into its noise
// probability (src/read_path_probabilities.cpp:181-215), the caller's sort compares the noise first
// (:283-322) and so does not bring it next to the row without that candidate, and the two meet again here.  But they
// are few (427 of the 3.28 M rows of the configs[2] workload), so nothing below sorts a whole matrix unless it must:
//   1. close pairs.  Every row carries a projection key (the sum of its values with fixed weights in [0, 2), written
//      by the build kernels as (matrix, fixed-point key): collapseSortKey, common.hpp).  Rows within prob_precision of
//      each other in all columns have keys within 2 (G + 1) prob_precision, so one radix sort over all matrices and
//      a forward window scan find every close pair.  A close pair that is equal up to rounding (<= 1e-13 relative) moves
//      nothing worth moving; a pair that is not marks both rows, and a second window scan adds the rows close to a
//      marked row: the ACTIVE rows.  A run of the reference that changes a value consists of active rows only, and an
//      inactive row is close to no active one.
//   2. replay, for the matrices with active rows:
//      a. (one workgroup per matrix) the active rows are sorted with the reference's comparator, bitonic network in LDS;
//      b. (the rows of those matrices, spread over many workgroups) an inactive row that the order places between two
//         neighbours of that list ends the run there, as it would in the reference (it is compared with the head,
//         fails, and becomes the head) — such a row has the zeros of both neighbours up to the first column in which
//         they differ, and lies between them in that column;
//      c. (one workgroup per matrix) the reference's compare-with-the-head rule, and the values of the head (group
//         columns, noise, row maximum) are copied over its run.
//      A matrix whose windows are too crowded to look through, or with more active rows than the LDS list holds,
//      takes the same code with every row active (the whole matrix sorted; no rows left to lie in between).
// The tolerant comparison is not a strict weak order; where it is inconsistent the reference's own result is
// whatever std::sort makes of it.

#include "common.hpp"

#include <hipcub/hipcub.hpp>

#include <cfloat>

using namespace rpvg_hip_detail;

namespace {

constexpr int kKeyFractionBits = kCollapseKeyFractionBits;
constexpr uint32_t kMaxWindowCompares = 256;   // a row with more candidates than this sends its matrix to the full sort
constexpr double kEquivalentRelative = 1e-13;  // close rows that differ by no more than this are interchangeable
constexpr uint32_t kListLdsRows = 8192;        // row lists up to this size live (and are sorted) in LDS
constexpr uint32_t kSortThreads = 1024;
constexpr uint32_t kBetweenThreads = 256;
constexpr uint32_t kBetweenRows = 2 * kBetweenThreads;  // rows of a matrix per work item of step b
constexpr uint32_t kPairChunk = 256;           // neighbour pairs staged in LDS at a time
constexpr uint32_t kWholeMatrixBit = 0x80000000u;  // mat_list: every row of the matrix is listed
constexpr uint32_t kBigListBit = 0x40000000u;      // mat_list: sorted by the bitonic kernel, compared in depth (no pair table)
constexpr uint32_t kListSizeMask = 0x3FFFFFFFu;
constexpr uint32_t kPairwiseRows = 1024;           // lists up to this size get a table of all their pairs
constexpr uint64_t kPairTableBytes = 16ull << 20;
constexpr uint8_t kPairLess = 1, kPairClose = 2;

enum : uint32_t { kFlagNone = 0, kFlagActiveRows = 1, kFlagWholeMatrix = 2 };
enum : uint32_t { kInfoMatrices = 0, kInfoRowsReplaced = 1, kInfoWholeMatrices = 2, kInfoActiveRows = 3, kInfoWords = 4 };

__device__ __forceinline__ bool tolerantEqual(const double a, const double b) {  // Utils::doubleCompare, src/utils.hpp:87-93
    return a == b || fabs(a - b) < fabs(fmin(a, b)) * (DBL_EPSILON * 100);
}

struct MatrixView {
    const double * values;     // column-major R x G
    const double * noise;
    const double * count;
    const uint64_t * pattern;  // bit c: column c (< 64) of the row is not zero
    uint64_t R;
    uint32_t G;
    // column G = noise.  (One load behind a selected address: a load on either side of a branch would wait for the
    // previous one, and the comparisons below live on having their loads in flight together.)
    __device__ __forceinline__ double at(const uint32_t column, const uint32_t row) const {
        const double * address = column < G ? values + (static_cast<uint64_t>(column) * R + row) : noise + row;
        return __builtin_nontemporal_load(address);
    }
};

// Rows that are compared in depth agree in most columns, and a column costs two strided loads.  The comparisons below
// walk the columns in the reference's order but (1) skip the columns in which both rows are zero — equal for every
// purpose here — by their zero patterns, and (2) fetch kBatch columns of both rows before they look at any of them,
// so that the loads are in flight together.
constexpr uint32_t kBatch = 8;

template <typename Decide>  // decide(x, y) -> true: stop
__device__ __forceinline__ void forEachColumnPair(const MatrixView & mv, const uint32_t a, const uint32_t b, uint64_t live, Decide decide) {
    uint32_t next_wide = 64;  // columns from 64 on carry no pattern: all of them
    bool noise_done = false;
    while (true) {
        uint32_t column[kBatch];
        uint32_t filled = 0;
        while (filled < kBatch && live) {
            column[filled++] = static_cast<uint32_t>(__ffsll(static_cast<long long>(live)) - 1);
            live &= live - 1;
        }
        while (filled < kBatch && next_wide < mv.G) column[filled++] = next_wide++;
        if (filled < kBatch && !noise_done) {
            column[filled++] = mv.G;
            noise_done = true;
        }
        if (filled == 0) return;
        double x[kBatch], y[kBatch];
#pragma unroll
        for (uint32_t k = 0; k < kBatch; ++k) {
            const uint32_t c = column[k < filled ? k : 0];
            x[k] = mv.at(c, a);
            y[k] = mv.at(c, b);
        }
#pragma unroll
        for (uint32_t k = 0; k < kBatch; ++k) {
            if (k < filled && decide(x[k], y[k])) return;
        }
    }
}

// probabilityCountRowSorter (src/path_estimator.cpp:13-31) on rows a, b of one matrix
__device__ bool rowLess(const MatrixView & mv, const uint32_t a, const uint32_t b, const uint64_t pattern_a, const uint64_t pattern_b) {
    int result = -1;
    forEachColumnPair(mv, a, b, pattern_a | pattern_b, [&](const double x, const double y) {
        if (tolerantEqual(x, y)) return false;
        result = x < y ? 1 : 0;
        return true;
    });
    if (result >= 0) return result != 0;
    const double x = mv.count[a], y = mv.count[b];
    if (!tolerantEqual(x, y)) return x < y;
    return false;
}

__device__ __forceinline__ bool rowLess(const MatrixView & mv, const uint32_t a, const uint32_t b) {
    return rowLess(mv, a, b, mv.pattern[a], mv.pattern[b]);
}

// every column (noise included) within `precision` of each other, absolutely (src/path_estimator.cpp:232-239);
// *equivalent: additionally equal up to rounding in every column
__device__ bool rowsClose(const MatrixView & mv, const uint32_t a, const uint32_t b, const double precision, bool * equivalent) {
    bool eq = true, close = true;
    forEachColumnPair(mv, a, b, mv.pattern[a] | mv.pattern[b], [&](const double x, const double y) {
        const double d = fabs(x - y);
        if (d >= precision) {
            close = false;
            return true;
        }
        if (d > kEquivalentRelative * fmin(fabs(x), fabs(y))) eq = false;
        return false;
    });
    if (close && equivalent) *equivalent = eq;
    return close;
}

__device__ bool rowsIdentical(const MatrixView & mv, const uint32_t a, const uint32_t b) {
    bool same = true;
    forEachColumnPair(mv, a, b, mv.pattern[a] | mv.pattern[b], [&](const double x, const double y) {
        if (x == y) return false;
        same = false;
        return true;
    });
    return same;
}

struct MatrixArrays {
    const uint64_t * mat_val_off;
    const uint64_t * mat_row_off;
    const uint64_t * mat_rows;
    const uint32_t * mat_cols;
    double * values;
    double * row_noise;
    const double * row_count;
    const uint64_t * zero_pattern;
};

__device__ __forceinline__ MatrixView viewOf(const uint32_t m, const MatrixArrays & g) {
    MatrixView mv;
    mv.values = g.values + g.mat_val_off[m];
    mv.noise = g.row_noise + g.mat_row_off[m];
    mv.count = g.row_count + g.mat_row_off[m];
    mv.pattern = g.zero_pattern + g.mat_row_off[m];
    mv.R = g.mat_rows[m];
    mv.G = g.mat_cols[m];
    return mv;
}

// ---- sort key fields (collapseSortKey, common.hpp) -------------------------------------------------------------
__device__ __forceinline__ uint32_t keyMatrix(const uint64_t key) { return static_cast<uint32_t>(key >> kCollapseMatrixShift); }
__device__ __forceinline__ uint64_t keySorted(const uint64_t key) { return key >> kCollapseLargestBits; }  // (matrix, projection)
__device__ __forceinline__ int64_t keyLargest(const uint64_t key) { return static_cast<int64_t>(key & ((1ull << kCollapseLargestBits) - 1)); }

// |key_a - key_b| <= sum_c w_c |a_c - b_c| < 2 (G + 1) precision for close rows; + rounding of the keys, + 2 quanta
__device__ __forceinline__ uint64_t windowQuanta(const uint32_t G, const double precision) {
    const double window = 2.0001 * (G + 1) * precision + 1e-12;
    return static_cast<uint64_t>(window * static_cast<double>(1ull << kKeyFractionBits)) + 2;
}

// the largest values of close rows lie within the precision of each other: steps of the key's low field apart
// (the field wraps: half its range or more means "no filter")
__device__ __forceinline__ int64_t largestSteps(const double precision) {
    const double steps = precision * static_cast<double>(1ull << kCollapseLargestFractionBits) + 1.0;
    return steps >= static_cast<double>(1ull << (kCollapseLargestBits - 1)) ? (1ll << kCollapseLargestBits) : static_cast<int64_t>(steps);
}

// `other` (at or behind `key` in the sorted order) can be close to `key`'s row
__device__ __forceinline__ bool inWindow(const uint64_t key, const uint64_t other, const uint64_t window_q, const int64_t largest_steps, bool * beyond) {
    *beyond = keyMatrix(other) != keyMatrix(key) || keySorted(other) - keySorted(key) > window_q;
    if (*beyond) return false;
    int64_t d = (keyLargest(other) - keyLargest(key)) & ((1ll << kCollapseLargestBits) - 1);  // modulo the field
    if (d >= (1ll << (kCollapseLargestBits - 1))) d -= 1ll << kCollapseLargestBits;
    return d <= largest_steps && -d <= largest_steps;
}

// ---- stage 1: close pairs ---------------------------------------------------------------------------------------

struct PairScanArgs {
    uint64_t total_rows;
    double precision;
    const uint64_t * sort_key;   // sorted (matrix, key, largest value)
    const uint32_t * sort_row;   // position of the row in the row arrays (mat_row_off[m] + row)
    MatrixArrays g;
    uint8_t * same_prev;     // [total rows] by sorted position
    uint32_t * marked_bits;  // [total rows / 32] by row: has a close partner that is not its equal up to rounding
    uint32_t * marked_list;  // [total rows] sorted positions of the marked rows
    uint32_t * marked_count;
    uint32_t * pairs;        // [2 x pair_capacity] candidate pairs (sorted positions) of the scan that runs
    uint32_t * pair_count;   // its number of pairs
    uint32_t pair_capacity;
    uint8_t * active;        // [total rows] by row: marked, or close to a marked row
    uint32_t * mat_flag;     // [M]
    uint32_t * info;
};

// The window scans only collect candidate pairs; the comparisons, each a few dependent rounds of strided loads, then
// run one per thread, all at once (a thread that walked its window and compared as it went spent ~1 us per step:
// 0.2-0.4 ms for the longest windows of the batch).
__device__ __forceinline__ bool appendPair(const PairScanArgs & a, const uint64_t p, const uint64_t q) {
    const uint32_t slot = atomicAdd(a.pair_count, 1u);
    if (slot >= a.pair_capacity) return false;
    a.pairs[2 * static_cast<uint64_t>(slot)] = static_cast<uint32_t>(p);
    a.pairs[2 * static_cast<uint64_t>(slot) + 1] = static_cast<uint32_t>(q);
    return true;
}

// same_prev[p] = the row at sorted position p is bit for bit the row at p - 1 (same matrix)
__global__ void collapseSamePrevKernel(const PairScanArgs a) {
    const uint64_t p = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (p >= a.total_rows) return;
    uint8_t same = 0;
    if (p > 0 && a.sort_key[p] == a.sort_key[p - 1]) {  // equal keys: nearly always the same values twice; the comparison leaves at the first difference
        const uint32_t m = keyMatrix(a.sort_key[p]);
        const MatrixView mv = viewOf(m, a.g);
        const uint64_t r0 = a.g.mat_row_off[m];
        same = rowsIdentical(mv, static_cast<uint32_t>(a.sort_row[p] - r0), static_cast<uint32_t>(a.sort_row[p - 1] - r0)) ? 1 : 0;
    }
    a.same_prev[p] = same;
}

// every row lists the rows after it whose keys lie within the window (a window too crowded to list sends the matrix
// to the full sort)
__global__ void collapseForwardPairsKernel(const PairScanArgs a) {
    const uint64_t p = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (p + 1 >= a.total_rows) return;
    const uint64_t key = a.sort_key[p];
    const uint64_t next = a.sort_key[p + 1];
    const uint32_t m = keyMatrix(key);
    if (keyMatrix(next) != m) return;
    const uint64_t window_q = windowQuanta(a.g.mat_cols[m], a.precision);
    if (keySorted(next) - keySorted(key) > window_q) return;  // nearly every row leaves here
    const int64_t largest_steps = largestSteps(a.precision);
    uint32_t listed = 0;
    for (uint64_t q = p + 1; q < a.total_rows; ++q) {
        const uint64_t other = a.sort_key[q];
        bool beyond;
        if (!inWindow(key, other, window_q, largest_steps, &beyond)) {
            if (beyond) break;
            continue;
        }
        if (a.same_prev[q]) continue;  // bit for bit the row before it: this row itself, or a candidate already listed
        if (++listed > kMaxWindowCompares || !appendPair(a, p, q)) {
            atomicMax(&a.mat_flag[m], static_cast<uint32_t>(kFlagWholeMatrix));
            break;
        }
    }
}

// marks the row at sorted position p; the first marker puts it on the list
__device__ __forceinline__ void markRow(const PairScanArgs & a, const uint64_t p) {
    const uint32_t row = a.sort_row[p];
    const uint32_t bit = 1u << (row & 31);
    if ((atomicOr(&a.marked_bits[row >> 5], bit) & bit) == 0) a.marked_list[atomicAdd(a.marked_count, 1u)] = static_cast<uint32_t>(p);
}

__device__ __forceinline__ bool isMarked(const PairScanArgs & a, const uint32_t row) { return (a.marked_bits[row >> 5] >> (row & 31)) & 1u; }

// both rows of a close pair that is not equal up to rounding are marked
__global__ void collapseMarkPairsKernel(const PairScanArgs a) {
    const uint32_t count = min(*a.pair_count, a.pair_capacity);
    for (uint32_t item = blockIdx.x * blockDim.x + threadIdx.x; item < count; item += gridDim.x * blockDim.x) {
        const uint64_t p = a.pairs[2 * static_cast<uint64_t>(item)], q = a.pairs[2 * static_cast<uint64_t>(item) + 1];
        const uint32_t m = keyMatrix(a.sort_key[p]);
        const MatrixView mv = viewOf(m, a.g);
        const uint64_t r0 = a.g.mat_row_off[m];
        bool equivalent = true;
        if (rowsClose(mv, static_cast<uint32_t>(a.sort_row[p] - r0), static_cast<uint32_t>(a.sort_row[q] - r0), a.precision, &equivalent) && !equivalent) {
            markRow(a, p);
            markRow(a, q);
            // the rows that are this candidate bit for bit are marked with it (a row's own copies list the candidate themselves)
            for (uint64_t s = q + 1; s < a.total_rows && a.same_prev[s]; ++s) markRow(a, s);
            atomicMax(&a.mat_flag[m], static_cast<uint32_t>(kFlagActiveRows));
        }
    }
}

// active = marked, or close to a marked row: every marked row (they are few: a list) lists its window, both ways
__global__ void collapseAroundPairsKernel(const PairScanArgs a) {
    const uint32_t count = *a.marked_count;
    for (uint32_t item = blockIdx.x * blockDim.x + threadIdx.x; item < count; item += gridDim.x * blockDim.x) {
        const uint64_t p = a.marked_list[item];
        a.active[a.sort_row[p]] = 1;
        const uint64_t key = a.sort_key[p];
        const uint32_t m = keyMatrix(key);
        const uint64_t window_q = windowQuanta(a.g.mat_cols[m], a.precision);
        const int64_t largest_steps = largestSteps(a.precision);
        uint32_t listed = 0;
        bool crowded = false;
        for (int direction = 0; direction < 2 && !crowded; ++direction) {
            for (uint64_t step = 1; !crowded; ++step) {
                if (direction == 0 ? p + step >= a.total_rows : step > p) break;
                const uint64_t q = direction == 0 ? p + step : p - step;
                const uint64_t other = a.sort_key[q];
                bool beyond;
                const bool candidate = direction == 0 ? inWindow(key, other, window_q, largest_steps, &beyond)
                                                      : inWindow(other, key, window_q, largest_steps, &beyond);
                if (beyond) break;
                // bit for bit the position looked at before it (same_prev[] looks towards lower positions): decided with it
                const bool repeat = direction == 0 ? a.same_prev[q] != 0 : a.same_prev[q + 1] != 0;
                if (!candidate || repeat || isMarked(a, a.sort_row[q])) continue;
                if (++listed > kMaxWindowCompares || !appendPair(a, p, q)) crowded = true;
            }
        }
        if (crowded) atomicMax(&a.mat_flag[m], static_cast<uint32_t>(kFlagWholeMatrix));
    }
}

__global__ void collapseActivePairsKernel(const PairScanArgs a) {
    const uint32_t count = min(*a.pair_count, a.pair_capacity);
    for (uint32_t item = blockIdx.x * blockDim.x + threadIdx.x; item < count; item += gridDim.x * blockDim.x) {
        const uint64_t p = a.pairs[2 * static_cast<uint64_t>(item)], q = a.pairs[2 * static_cast<uint64_t>(item) + 1];
        const uint32_t m = keyMatrix(a.sort_key[p]);
        const MatrixView mv = viewOf(m, a.g);
        const uint64_t r0 = a.g.mat_row_off[m];
        if (!rowsClose(mv, static_cast<uint32_t>(a.sort_row[p] - r0), static_cast<uint32_t>(a.sort_row[q] - r0), a.precision, nullptr)) continue;
        a.active[a.sort_row[q]] = 1;  // (several writers, one value)
        // and the rows that are this one bit for bit, on either side of it
        for (uint64_t s = q + 1; s < a.total_rows && a.same_prev[s]; ++s) a.active[a.sort_row[s]] = 1;
        for (uint64_t s = q; s > 0 && a.same_prev[s]; --s) a.active[a.sort_row[s - 1]] = 1;
    }
}

// ---- stage 2: replay ------------------------------------------------------------------------------------------

constexpr uint32_t kNoRow = 0xFFFFFFFFu;

struct ReplayArgs {
    uint32_t num_matrices;
    double precision;
    MatrixArrays g;
    const uint32_t * mat_flag;
    const uint8_t * active;      // [total rows] by row
    double * rowmax;
    uint32_t * mat_fast;
    uint32_t * mat_mid;
    uint32_t * mat_list;         // [M] size of the matrix's list (| kWholeMatrixBit: every row), 0: nothing to replay
    uint32_t * replay_list;      // [M] the matrices with a list, [M]: their number
    uint32_t * between_items;    // [2 x (total rows / kBetweenRows + M)] (matrix, slice of its rows) of step b
    uint32_t * between_count;
    uint32_t * row_items;        // [2 x total rows] (matrix, list index) of the rows of the lists with a pair table
    uint32_t * row_item_count;
    uint64_t * pair_base;        // [M] offset of the matrix's n x n pair table
    unsigned long long * pair_bytes;  // bytes of pair tables handed out
    uint8_t * pair_table;        // [kPairTableBytes] entry i * n + j: kPairLess (row i sorts before row j), kPairClose
    uint32_t * list_index;       // [total rows] index in the unsorted list (= in the pair table) of every sorted position
    uint32_t * order;            // [2 * total rows] list of matrix m at 2 * mat_row_off[m], sorted
    uint32_t * head_of;          // [total rows] list position of the run head of every list position
    uint8_t * close;             // [total rows] list position p is close to p - 1 and nothing lies between them
    uint8_t * barrier;           // [total rows] an inactive row lies between list positions p - 1 and p
    uint32_t * pair_column;      // [total rows] column that orders list positions p - 1 and p (kNoRow: not looked at)
    double * pair_lo;            // [total rows] their values in it, smaller
    double * pair_hi;            //              and larger
    uint64_t * pair_pattern;     // [total rows] zero pattern they share before that column
    uint32_t * info;
};

// a0. the list of a matrix: its active rows (the replay then only touches those), or all of them
__global__ __launch_bounds__(256) void collapseListKernel(const ReplayArgs a) {
    __shared__ uint32_t list_size;
    const uint32_t m = blockIdx.x;
    if (m >= a.num_matrices) return;
    const uint32_t flag = a.mat_flag[m];
    if (flag == kFlagNone) {
        if (threadIdx.x == 0) a.mat_list[m] = 0;
        return;
    }
    const uint64_t R = a.g.mat_rows[m];
    const uint64_t r0 = a.g.mat_row_off[m];
    const uint8_t * active = a.active + r0;
    uint32_t * list = a.order + 2 * r0;
    if (threadIdx.x == 0) list_size = 0;
    __syncthreads();
    if (flag != kFlagWholeMatrix) {
        for (uint64_t i = threadIdx.x; i < R; i += blockDim.x) {
            if (active[i]) list[atomicAdd(&list_size, 1u)] = static_cast<uint32_t>(i);
        }
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    const bool whole = flag == kFlagWholeMatrix;
    const uint64_t n = whole ? R : list_size;
    if (n < 2) {
        a.mat_list[m] = 0;
        return;
    }
    uint32_t encoded = static_cast<uint32_t>(n);
    if (whole) encoded |= kWholeMatrixBit | kBigListBit;
    else if (n > kPairwiseRows) encoded |= kBigListBit;
    else {
        const unsigned long long base = atomicAdd(a.pair_bytes, static_cast<unsigned long long>(n * n));
        if (base + n * n > kPairTableBytes) encoded |= kBigListBit;
        else {
            a.pair_base[m] = base;
            const uint32_t first = atomicAdd(a.row_item_count, static_cast<uint32_t>(n));
            for (uint32_t i = 0; i < n; ++i) {
                a.row_items[2 * static_cast<uint64_t>(first + i)] = m;
                a.row_items[2 * static_cast<uint64_t>(first + i) + 1] = i;
            }
        }
    }
    a.mat_list[m] = encoded;
    a.replay_list[atomicAdd(&a.replay_list[a.num_matrices], 1u)] = m;
    if (!whole) {  // work items of step b: (matrix, slice of kBetweenRows rows)
        const uint32_t slices = static_cast<uint32_t>((R + kBetweenRows - 1) / kBetweenRows);
        const uint32_t first = atomicAdd(a.between_count, slices);
        for (uint32_t k = 0; k < slices; ++k) {
            a.between_items[2 * static_cast<uint64_t>(first + k)] = m;
            a.between_items[2 * static_cast<uint64_t>(first + k) + 1] = k;
        }
    }
    atomicAdd(&a.info[kInfoMatrices], 1u);
    atomicAdd(&a.info[kInfoActiveRows], static_cast<uint32_t>(n));
    if (whole) atomicAdd(&a.info[kInfoWholeMatrices], 1u);
}

// a1. every pair of rows of a small list, one thread each: order and closeness in one pass over the columns.  (A
// workgroup that sorts its list with a comparison network goes through dozens of dependent rounds of strided loads;
// here every comparison of the batch is in flight at once, and ranks and runs are then read off the table.)
__global__ __launch_bounds__(64) void collapsePairTableKernel(const ReplayArgs a) {
    const uint32_t num_items = *a.row_item_count;
    for (uint32_t item = blockIdx.x; item < num_items; item += gridDim.x) {
        const uint32_t m = a.row_items[2 * static_cast<uint64_t>(item)], i = a.row_items[2 * static_cast<uint64_t>(item) + 1];
        const uint32_t n = a.mat_list[m] & kListSizeMask;
        const MatrixView mv = viewOf(m, a.g);
        const uint32_t * list = a.order + 2 * a.g.mat_row_off[m];
        uint8_t * table = a.pair_table + a.pair_base[m] + static_cast<uint64_t>(i) * n;
        const uint32_t row_i = list[i];
        const uint64_t pattern_i = mv.pattern[row_i];
        for (uint32_t j = threadIdx.x; j < n; j += blockDim.x) {
            const uint32_t row_j = list[j];
            int less = -1;
            bool close = true;
            forEachColumnPair(mv, row_i, row_j, pattern_i | mv.pattern[row_j], [&](const double x, const double y) {
                if (less < 0 && !tolerantEqual(x, y)) less = x < y ? 1 : 0;
                if (fabs(x - y) >= a.precision) close = false;
                return less >= 0 && !close;
            });
            if (less < 0) {
                const double x = mv.count[row_i], y = mv.count[row_j];
                less = (!tolerantEqual(x, y) && x < y) ? 1 : 0;
            }
            table[j] = (less ? kPairLess : 0) | (close ? kPairClose : 0);
        }
    }
}

// a2. small lists sorted by rank: the number of rows that sort before a row (equal rows in list order).  Where the
// tolerant comparison is inconsistent the ranks may collide; such a list is sorted by the comparison network instead.
__global__ __launch_bounds__(256) void collapseRankKernel(const ReplayArgs a) {
    __shared__ uint32_t lds_row[kPairwiseRows], lds_slot[kPairwiseRows];
    __shared__ uint32_t collision;
    const uint32_t num_items = a.replay_list[a.num_matrices];
    for (uint32_t item = blockIdx.x; item < num_items; item += gridDim.x) {
        const uint32_t m = a.replay_list[item];
        const uint32_t encoded = a.mat_list[m];
        if (encoded & kBigListBit) continue;
        const uint32_t n = encoded;
        const uint64_t r0 = a.g.mat_row_off[m];
        uint32_t * list = a.order + 2 * r0;
        uint32_t * list_index = a.list_index + r0;
        const uint8_t * table = a.pair_table + a.pair_base[m];
        __syncthreads();
        if (threadIdx.x == 0) collision = 0;
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
            lds_row[i] = list[i];
            lds_slot[i] = kNoRow;
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
            uint32_t rank = 0;
            for (uint32_t j = 0; j < n; ++j) {
                const bool before = (table[static_cast<uint64_t>(j) * n + i] & kPairLess) != 0;
                const bool after = (table[static_cast<uint64_t>(i) * n + j] & kPairLess) != 0;
                rank += (before || (!after && j < i)) ? 1u : 0u;
            }
            if (rank >= n || atomicExch(&lds_slot[rank], i) != kNoRow) collision = 1;
        }
        __syncthreads();
        if (collision) {  // (never seen; kept correct rather than fast) insertion sort with the comparator itself
            if (threadIdx.x == 0) {
                const MatrixView mv = viewOf(m, a.g);
                for (uint32_t i = 0; i < n; ++i) lds_slot[i] = i;
                for (uint32_t i = 1; i < n; ++i) {
                    const uint32_t moving = lds_slot[i];
                    uint32_t k = i;
                    while (k > 0 && rowLess(mv, lds_row[moving], lds_row[lds_slot[k - 1]])) {
                        lds_slot[k] = lds_slot[k - 1];
                        --k;
                    }
                    lds_slot[k] = moving;
                }
            }
            __syncthreads();
        }
        for (uint32_t p = threadIdx.x; p < n; p += blockDim.x) {
            list[p] = lds_row[lds_slot[p]];
            list_index[p] = lds_slot[p];
            a.barrier[r0 + p] = 0;
        }
        __syncthreads();
        // the column that orders every pair of neighbours, and the interval between them in it (collapseSortKernel)
        const MatrixView mv = viewOf(m, a.g);
        for (uint32_t p = 1 + threadIdx.x; p < n; p += blockDim.x) {
            const uint32_t x = lds_row[lds_slot[p - 1]], y = lds_row[lds_slot[p]];
            uint32_t d = kNoRow;
            bool can_join = true;
            // (forEachColumnPair visits the columns in order but skips those in which both rows are zero: the column
            // index is recovered from the patterns)
            const uint64_t live = mv.pattern[x] | mv.pattern[y];
            uint64_t remaining = live;
            uint32_t wide = 64;
            forEachColumnPair(mv, x, y, live, [&](const double vx, const double vy) {
                uint32_t column;
                if (remaining) {
                    column = static_cast<uint32_t>(__ffsll(static_cast<long long>(remaining)) - 1);
                    remaining &= remaining - 1;
                } else if (wide < mv.G) {
                    column = wide++;
                } else {
                    column = mv.G;
                }
                if (fabs(vx - vy) >= 2 * a.precision) can_join = false;
                if (d == kNoRow && !tolerantEqual(vx, vy)) d = column;
                return !can_join;
            });
            const bool look = can_join && d != kNoRow;
            a.pair_column[r0 + p] = look ? d : kNoRow;
            if (look) {
                const double vx = mv.at(d, x), vy = mv.at(d, y);
                a.pair_lo[r0 + p] = fmin(vx, vy);
                a.pair_hi[r0 + p] = fmax(vx, vy);
                a.pair_pattern[r0 + p] = mv.pattern[x] & (d >= 64 ? ~0ull : (1ull << d) - 1ull);
            }
        }
    }
}

// a3. big lists (more than kPairwiseRows rows, whole matrices): sorted with the reference's comparator in a
// comparison network, and for every pair of neighbours the column that orders them
__global__ __launch_bounds__(kSortThreads) void collapseSortKernel(const ReplayArgs a) {
    __shared__ uint32_t lds_list[kListLdsRows];
    __shared__ uint64_t lds_pattern[kListLdsRows];  // zero pattern of the row at each list position (LDS lists)
    // over the matrices with a list (a workgroup of this kernel takes a CU's worth of LDS: one per matrix of the batch, each
    // only to find that its matrix has no big list, took 0.35-0.8 ms next to the other lane's kernels)
    const uint32_t num_items = a.replay_list[a.num_matrices];
    for (uint32_t item = blockIdx.x; item < num_items; item += gridDim.x) {
    __syncthreads();  // the lists of the matrix before are done with
    const uint32_t m = a.replay_list[item];
    const uint32_t encoded = a.mat_list[m];
    if (!(encoded & kBigListBit)) continue;  // a small list (pair table, collapseRankKernel)
    const bool whole = (encoded & kWholeMatrixBit) != 0;
    const uint64_t n = encoded & kListSizeMask;
    const MatrixView mv = viewOf(m, a.g);
    const uint64_t r0 = a.g.mat_row_off[m];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, num_waves = blockDim.x >> 6;
    uint64_t padded = 1;
    while (padded < n) padded <<= 1;
    uint32_t * global_order = a.order + 2 * r0;
    uint32_t * order = padded <= kListLdsRows ? lds_list : global_order;
    for (uint64_t i = threadIdx.x; i < padded; i += blockDim.x) order[i] = i < n ? (whole ? static_cast<uint32_t>(i) : global_order[i]) : kNoRow;
    __syncthreads();
    const bool in_lds = order == lds_list;
    if (in_lds) {
        for (uint64_t i = threadIdx.x; i < padded; i += blockDim.x) lds_pattern[i] = order[i] == kNoRow ? 0ull : mv.pattern[order[i]];
    }
    __syncthreads();
    // bitonic network with the reference's comparator; kNoRow sorts behind every row
    for (uint64_t k = 2; k <= padded; k <<= 1) {
        for (uint64_t j = k >> 1; j > 0; j >>= 1) {
            for (uint64_t t = threadIdx.x; t < (padded >> 1); t += blockDim.x) {
                const uint64_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1));  // bit j of i is clear
                const uint64_t l = i | j;
                const uint32_t x = order[i], y = order[l];
                const bool ascending = (i & k) == 0;
                bool swap;
                if (x == kNoRow || y == kNoRow) {
                    swap = ascending ? (x == kNoRow && y != kNoRow) : (y == kNoRow && x != kNoRow);
                } else {
                    const uint64_t px = in_lds ? lds_pattern[i] : mv.pattern[x], py = in_lds ? lds_pattern[l] : mv.pattern[y];
                    swap = ascending ? rowLess(mv, y, x, py, px) : rowLess(mv, x, y, px, py);
                }
                if (swap) {
                    order[i] = y;
                    order[l] = x;
                    if (in_lds) {
                        const uint64_t px = lds_pattern[i];
                        lds_pattern[i] = lds_pattern[l];
                        lds_pattern[l] = px;
                    }
                }
            }
            __syncthreads();
        }
    }
    if (order != global_order) {
        for (uint64_t i = threadIdx.x; i < n; i += blockDim.x) global_order[i] = order[i];
    }
    uint8_t * barrier = a.barrier + r0;
    for (uint64_t p = threadIdx.x; p < n; p += blockDim.x) barrier[p] = 0;
    if (whole) continue;
    // One wave per pair of neighbours: d, the first column in which they differ, and the interval between them in it.
    // A row x with a <= x <= b in the reference's order equals both up to rounding before column d and lies between
    // them in column d.  A row that joins a run is within prob_precision of a head its predecessor is within
    // prob_precision of (or the predecessor is that head): only neighbours that close can be parted by a row.
    uint32_t * pair_column = a.pair_column + r0;
    for (uint64_t p = 1 + wave; p < n; p += num_waves) {
        const uint32_t x = order[p - 1], y = order[p];
        uint32_t d = mv.G + 1;
        bool can_join = true;
        for (uint32_t c0 = 0; c0 <= mv.G; c0 += 64) {
            const uint32_t c = c0 + lane;
            const double vx = c <= mv.G ? mv.at(c, x) : 0.0, vy = c <= mv.G ? mv.at(c, y) : 0.0;
            if (__ballot(fabs(vx - vy) >= 2 * a.precision)) can_join = false;
            const unsigned long long differs = __ballot(c <= mv.G && !tolerantEqual(vx, vy));
            if (differs && d > mv.G) d = c0 + static_cast<uint32_t>(__ffsll(static_cast<long long>(differs)) - 1);
        }
        if (lane == 0) {
            const bool look = can_join && d <= mv.G;  // (equal in every column: whatever lies between them is their equal, too)
            pair_column[p] = look ? d : kNoRow;
            if (look) {
                const double vx = mv.at(d, x), vy = mv.at(d, y);
                a.pair_lo[r0 + p] = fmin(vx, vy);
                a.pair_hi[r0 + p] = fmax(vx, vy);
                a.pair_pattern[r0 + p] = mv.pattern[x] & (d >= 64 ? ~0ull : (1ull << d) - 1ull);
            }
        }
    }
    }
}

// b. inactive rows between neighbours of the lists: work item = (matrix with a list, slice of its rows)
__global__ __launch_bounds__(kBetweenThreads) void collapseBetweenKernel(const ReplayArgs a) {
    __shared__ uint32_t lds_column[kPairChunk];
    __shared__ double lds_lo[kPairChunk], lds_hi[kPairChunk];
    __shared__ uint64_t lds_pattern[kPairChunk], lds_before[kPairChunk];
    const uint32_t num_items = *a.between_count;
    for (uint32_t item = blockIdx.x; item < num_items; item += gridDim.x) {
        const uint32_t m = a.between_items[2 * static_cast<uint64_t>(item)], slice = a.between_items[2 * static_cast<uint64_t>(item) + 1];
        const uint64_t n = a.mat_list[m] & kListSizeMask;  // (not a whole matrix: those have no items)
        const MatrixView mv = viewOf(m, a.g);
        const uint64_t r0 = a.g.mat_row_off[m];
        const uint8_t * active = a.active + r0;
        const uint32_t * order = a.order + 2 * r0;
        for (uint64_t p0 = 1; p0 < n; p0 += kPairChunk) {
            const uint32_t chunk = static_cast<uint32_t>(min(static_cast<uint64_t>(kPairChunk), n - p0));
            __syncthreads();
            for (uint32_t k = threadIdx.x; k < chunk; k += blockDim.x) {
                const uint32_t d = a.pair_column[r0 + p0 + k];
                lds_column[k] = d;
                if (d != kNoRow) {
                    lds_lo[k] = a.pair_lo[r0 + p0 + k];
                    lds_hi[k] = a.pair_hi[r0 + p0 + k];
                    lds_pattern[k] = a.pair_pattern[r0 + p0 + k];
                    lds_before[k] = d >= 64 ? ~0ull : (1ull << d) - 1ull;
                }
            }
            __syncthreads();
            const uint64_t x_end = min(mv.R, (slice + 1ull) * kBetweenRows);
            for (uint64_t x = slice * static_cast<uint64_t>(kBetweenRows) + threadIdx.x; x < x_end; x += blockDim.x) {
                if (active[x]) continue;
                const uint64_t pattern_x = mv.pattern[x];
                uint32_t loaded_column = kNoRow;
                double vx = 0.0;
                for (uint32_t k = 0; k < chunk; ++k) {
                    const uint32_t d = lds_column[k];
                    if (d == kNoRow || (pattern_x & lds_before[k]) != lds_pattern[k]) continue;
                    if (d != loaded_column) {
                        vx = mv.at(d, static_cast<uint32_t>(x));
                        loaded_column = d;
                    }
                    const double lo = lds_lo[k], hi = lds_hi[k];
                    if (!((vx >= lo || tolerantEqual(vx, lo)) && (vx <= hi || tolerantEqual(vx, hi)))) continue;
                    const uint64_t p = p0 + k;
                    if (!rowLess(mv, static_cast<uint32_t>(x), order[p - 1]) && !rowLess(mv, order[p], static_cast<uint32_t>(x))) a.barrier[r0 + p] = 1;
                }
            }
        }
    }
}

// c. runs and values: work item = matrix with a list
__global__ __launch_bounds__(256) void collapseRunsKernel(const ReplayArgs a) {
    __shared__ uint32_t lds_next[kListLdsRows];
    __shared__ uint32_t demote;
    const int lane = threadIdx.x & 63;
    const uint32_t num_items = a.replay_list[a.num_matrices];
    for (uint32_t item = blockIdx.x; item < num_items; item += gridDim.x) {
        const uint32_t m = a.replay_list[item];
        const uint32_t encoded = a.mat_list[m];
        const uint64_t n = encoded & kListSizeMask;
        const bool tabled = !(encoded & kBigListBit);
        const MatrixView mv = viewOf(m, a.g);
        const uint64_t R = mv.R;
        const uint64_t r0 = a.g.mat_row_off[m];
        const uint32_t * order = a.order + 2 * r0;
        const uint32_t * list_index = a.list_index + r0;
        const uint8_t * table = a.pair_table + (tabled ? a.pair_base[m] : 0);
        uint32_t * head_of = a.head_of + r0;
        uint32_t * next_head = n <= kListLdsRows ? lds_next : a.pair_column + r0;  // (the pair columns are done with)
        uint8_t * close = a.close + r0;
        const uint8_t * barrier = a.barrier + r0;
        // list positions p, q hold rows within prob_precision of each other: read off the pair table, or compared
        auto closeRows = [&](const uint64_t p, const uint64_t q) {
            if (tabled) return (table[static_cast<uint64_t>(list_index[p]) * n + list_index[q]] & kPairClose) != 0;
            return rowsClose(mv, order[p], order[q], a.precision, nullptr);
        };
        __syncthreads();
        if (threadIdx.x == 0) demote = 0;
        for (uint64_t p = threadIdx.x; p < n; p += blockDim.x) {
            head_of[p] = kNoRow;
            close[p] = (p > 0 && !barrier[p] && closeRows(p - 1, p)) ? 1 : 0;
        }
        __syncthreads();
        // Runs (src/path_estimator.cpp:226-255): a row joins the run of the current head if it is close to the head,
        // otherwise it becomes the head.  Every list position answers "if I were a head, where would the next one be"
        // on its own (the first position behind it that an inactive row parts from it or that is not close to it: a
        // wave per position that has a follower, 64 candidates at a time); one thread then walks from head to head,
        // and every head claims its run.
        for (uint64_t h = threadIdx.x; h < n; h += blockDim.x) {
            if (h + 1 >= n || !close[h + 1]) next_head[h] = static_cast<uint32_t>(h + 1);
        }
        for (uint64_t h = threadIdx.x >> 6; h + 1 < n; h += blockDim.x >> 6) {
            if (!close[h + 1]) continue;
            uint64_t q0 = h + 2, found = n;
            while (q0 < n && found == n) {
                const uint64_t q = q0 + lane;
                const bool stop = q < n && (barrier[q] != 0 || !closeRows(h, q));
                const unsigned long long ballot = __ballot(stop);
                if (ballot) found = q0 + static_cast<uint64_t>(__ffsll(static_cast<long long>(ballot)) - 1);
                q0 += 64;
            }
            if (lane == 0) next_head[h] = static_cast<uint32_t>(found);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (uint64_t h = 0; h < n; h = next_head[h]) head_of[h] = static_cast<uint32_t>(h);
        }
        __syncthreads();
        for (uint64_t h = threadIdx.x; h < n; h += blockDim.x) {
            if (head_of[h] != h) continue;
            for (uint64_t q = h + 1; q < next_head[h]; ++q) head_of[q] = static_cast<uint32_t>(h);
        }
        __syncthreads();
        // the rows of a run take the values of its head: a wave per row, its lanes over the columns
        double * M = a.g.values + a.g.mat_val_off[m];
        double * nz = a.g.row_noise + r0;
        double * rm = a.rowmax + r0;
        const uint32_t fast_mid_end = a.mat_mid[m];
        uint32_t replaced = 0;
        for (uint64_t q = threadIdx.x >> 6; q < n; q += blockDim.x >> 6) {
            const uint32_t h = head_of[q];
            if (h == q) continue;
            const uint32_t dst = order[q], src = order[h];
            for (uint32_t c = lane; c < mv.G; c += 64) M[static_cast<uint64_t>(c) * R + dst] = M[static_cast<uint64_t>(c) * R + src];
            if (lane == 0) {
                const double noise = nz[src];
                nz[dst] = noise;
                rm[dst] = rm[src];
                ++replaced;
                // a product-path row (LogProduct, common.hpp) needs noise >= kProductMinNoise: if the head's is below,
                // the whole matrix takes the logarithm path
                if (dst < fast_mid_end && !(noise >= kProductMinNoise)) demote = 1;
            }
        }
        if (replaced) atomicAdd(&a.info[kInfoRowsReplaced], replaced);
        __syncthreads();
        if (threadIdx.x == 0 && demote) {
            a.mat_fast[m] = 0;
            a.mat_mid[m] = 0;
        }
    }
}

}  // namespace

// Queues the collapse of the matrices of `g` on `st` behind their build (rpvg_hip_groups_build).
hipError_t rpvg_hip_detail::queueRowCollapse(rpvg_hip_ctx * ctx, rpvg_hip_groups * g, const uint64_t total_rows, const double precision,
                                             hipStream_t st) {
    (void) ctx;
    const uint32_t M = g->num_matrices;
    if (M == 0 || total_rows == 0) return hipSuccess;
    if (total_rows > 0x7fffffffull || M > kCollapseMaxMatrices) return hipErrorInvalidValue;
    struct CollapseTemporaries {
        DeviceBuffer<uint64_t> key_out, pair_pattern;
        DeviceBuffer<uint32_t> row_out, order, head, pair_column, marked_list, pairs, between_items, row_items, list_index;
        DeviceBuffer<uint64_t> pair_base;
        DeviceBuffer<uint8_t> pair_table;
        DeviceBuffer<double> pair_bound;
        DeviceBuffer<uint8_t> bytes;  // same_prev, close, barrier: total_rows each
        DeviceBuffer<unsigned char> sort_tmp;
    };
    std::shared_ptr<CollapseTemporaries> tmp = std::make_shared<CollapseTemporaries>();
    g->build_temporaries.emplace_back(tmp);
    hipError_t e = hipSuccess;
    auto ok = [&](hipError_t r) { if (e == hipSuccess) e = r; return e == hipSuccess; };
    // zeroed words: the pair table's byte counter (8 bytes, first: aligned), matrix flags [M], replay list [M] + its
    // count, marked bits, then the counters of the marked list, the two pair lists, the between items, the row items
    const uint64_t mark_words = (total_rows + 31) / 32;
    const uint64_t num_words = 2 + 2 * static_cast<uint64_t>(M) + 1 + mark_words + 5;
    const uint32_t pair_capacity = static_cast<uint32_t>(total_rows / 2 + 4096);
    ok(tmp->key_out.alloc(total_rows));
    ok(tmp->row_out.alloc(total_rows));
    // one block, one memset: [info | the zeroed words | active bytes] + list sizes [M] (written by the list kernel)
    const uint64_t active_words = (total_rows + 3) / 4;
    ok(g->collapse_info.alloc(kInfoWords + num_words + active_words + M));
    ok(tmp->order.alloc(2 * total_rows));
    ok(tmp->head.alloc(total_rows));
    ok(tmp->pair_column.alloc(total_rows));
    ok(tmp->pair_bound.alloc(2 * total_rows));
    ok(tmp->pair_pattern.alloc(total_rows));
    ok(tmp->marked_list.alloc(total_rows));
    ok(tmp->pairs.alloc(2 * static_cast<size_t>(pair_capacity)));
    ok(tmp->between_items.alloc(2 * (total_rows / kBetweenRows + M)));
    ok(tmp->row_items.alloc(2 * total_rows));
    ok(tmp->list_index.alloc(total_rows));
    ok(tmp->pair_base.alloc(M));
    ok(tmp->pair_table.alloc(kPairTableBytes));
    ok(tmp->bytes.alloc(3 * total_rows));
    // The keys carry the matrix above the projection and the rows of a matrix are contiguous: ONE (stable) radix sort of
    // the whole array on the projection and as many matrix bits as there are matrices orders every matrix's rows by their
    // projection.  (Round 2 sorted segment by segment, hipcub::DeviceSegmentedRadixSort: 0.33 ms per 1.6 M rows in 2 500
    // segments, the longest kernel of the collapse; RPVG_HIP_COLLAPSE_SEGMENTED_SORT=1 keeps it for A/B.)
    static const bool segmented = std::getenv("RPVG_HIP_COLLAPSE_SEGMENTED_SORT") != nullptr;
    int matrix_bits = 1;
    while ((1u << matrix_bits) < M) ++matrix_bits;
    const int begin_bit = kCollapseLargestBits, end_bit = segmented ? kCollapseMatrixShift : kCollapseMatrixShift + matrix_bits;
    const uint32_t * segment_off = g->collapse_segment_off.ptr;
    size_t sort_bytes = 0;
    auto sort = [&](void * scratch) {
        return segmented ? hipcub::DeviceSegmentedRadixSort::SortPairs(scratch, sort_bytes, g->collapse_key.ptr, tmp->key_out.ptr, g->collapse_row.ptr,
                                                                       tmp->row_out.ptr, static_cast<int>(total_rows), static_cast<int>(M),
                                                                       segment_off, segment_off + 1, begin_bit, end_bit, st)
                         : hipcub::DeviceRadixSort::SortPairs(scratch, sort_bytes, g->collapse_key.ptr, tmp->key_out.ptr, g->collapse_row.ptr,
                                                              tmp->row_out.ptr, static_cast<int>(total_rows), begin_bit, end_bit, st);
    };
    if (e == hipSuccess) ok(sort(nullptr));
    ok(tmp->sort_tmp.alloc(sort_bytes));
    if (e != hipSuccess) return e;
    uint32_t * pair_bytes = g->collapse_info.ptr + kInfoWords, * mat_flag = pair_bytes + 2, * replay_list = mat_flag + M, * marked_bits = replay_list + M + 1,
             * marked_count = marked_bits + mark_words, * pair_counts = marked_count + 1, * between_count = pair_counts + 2,
             * row_item_count = between_count + 1, * mat_list = pair_bytes + num_words + active_words;
    uint8_t * same_prev = tmp->bytes.ptr, * close = same_prev + total_rows, * barrier = close + total_rows;
    uint8_t * active = reinterpret_cast<uint8_t *>(pair_bytes + num_words);
    ok(hipMemsetAsync(g->collapse_info.ptr, 0, (kInfoWords + num_words + active_words) * sizeof(uint32_t), st));
    ok(sort(tmp->sort_tmp.ptr));
    MatrixArrays arrays;
    arrays.mat_val_off = g->mat_val_off.ptr;
    arrays.mat_row_off = g->mat_row_off.ptr;
    arrays.mat_rows = g->mat_rows.ptr;
    arrays.mat_cols = g->mat_cols.ptr;
    arrays.values = g->values.ptr;
    arrays.row_noise = g->row_noise.ptr;
    arrays.row_count = g->row_count.ptr;
    arrays.zero_pattern = g->collapse_mask.ptr;
    PairScanArgs a;
    a.total_rows = total_rows;
    a.precision = precision;
    a.sort_key = tmp->key_out.ptr;
    a.sort_row = tmp->row_out.ptr;
    a.g = arrays;
    a.same_prev = same_prev;
    a.marked_bits = marked_bits;
    a.marked_list = tmp->marked_list.ptr;
    a.marked_count = marked_count;
    a.pairs = tmp->pairs.ptr;
    a.pair_count = pair_counts;
    a.pair_capacity = pair_capacity;
    a.active = active;
    a.mat_flag = mat_flag;
    a.info = g->collapse_info.ptr;
    const uint32_t row_blocks = static_cast<uint32_t>((total_rows + 255) / 256);
    collapseSamePrevKernel<<<dim3(row_blocks), dim3(256), 0, st>>>(a);
    collapseForwardPairsKernel<<<dim3(row_blocks), dim3(256), 0, st>>>(a);
    collapseMarkPairsKernel<<<dim3(1024), dim3(64), 0, st>>>(a);
    a.pair_count = pair_counts + 1;  // the list is free again
    collapseAroundPairsKernel<<<dim3(256), dim3(64), 0, st>>>(a);
    collapseActivePairsKernel<<<dim3(1024), dim3(64), 0, st>>>(a);
    ReplayArgs r;
    r.num_matrices = M;
    r.precision = precision;
    r.g = arrays;
    r.mat_flag = mat_flag;
    r.active = active;
    r.rowmax = g->rowmax.ptr;
    r.mat_fast = g->mat_fast.ptr;
    r.mat_mid = g->mat_mid.ptr;
    r.mat_list = mat_list;
    r.replay_list = replay_list;
    r.between_items = tmp->between_items.ptr;
    r.between_count = between_count;
    r.row_items = tmp->row_items.ptr;
    r.row_item_count = row_item_count;
    r.pair_base = tmp->pair_base.ptr;
    r.pair_bytes = reinterpret_cast<unsigned long long *>(pair_bytes);
    r.pair_table = tmp->pair_table.ptr;
    r.list_index = tmp->list_index.ptr;
    r.order = tmp->order.ptr;
    r.head_of = tmp->head.ptr;
    r.close = close;
    r.barrier = barrier;
    r.pair_column = tmp->pair_column.ptr;
    r.pair_lo = tmp->pair_bound.ptr;
    r.pair_hi = tmp->pair_bound.ptr + total_rows;
    r.pair_pattern = tmp->pair_pattern.ptr;
    r.info = g->collapse_info.ptr;
    collapseListKernel<<<dim3(M), dim3(256), 0, st>>>(r);
    collapsePairTableKernel<<<dim3(4096), dim3(64), 0, st>>>(r);
    collapseRankKernel<<<dim3(1024), dim3(256), 0, st>>>(r);
    collapseSortKernel<<<dim3(std::min<uint32_t>(M, 128)), dim3(kSortThreads), 0, st>>>(r);
    collapseBetweenKernel<<<dim3(2048), dim3(kBetweenThreads), 0, st>>>(r);
    collapseRunsKernel<<<dim3(1024), dim3(256), 0, st>>>(r);
    ok(hipGetLastError());
    return e;
}

extern "C" int rpvg_hip_groups_collapse_info(rpvg_hip_ctx * ctx, const rpvg_hip_groups * groups, uint32_t * matrices_replayed,
                                             uint32_t * rows_replaced, uint32_t * matrices_sorted_whole, uint32_t * active_rows) {
    RPVG_REQUIRE(ctx && groups, "rpvg_hip_groups_collapse_info: NULL argument");
    uint32_t info[kInfoWords] = {0};
    if (groups->collapse_info.ptr) {
        std::lock_guard<std::mutex> lock(ctx->mutex);
        RPVG_HIP_CHECK(hipSetDevice(ctx->device));
        RPVG_HIP_CHECK(groups->waitCollapse(ctx->stream));
        RPVG_HIP_CHECK(hipMemcpyAsync(info, groups->collapse_info.ptr, sizeof(info), hipMemcpyDeviceToHost, ctx->stream));
        RPVG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
    if (matrices_replayed) *matrices_replayed = info[kInfoMatrices];
    if (rows_replaced) *rows_replaced = info[kInfoRowsReplaced];
    if (matrices_sorted_whole) *matrices_sorted_whole = info[kInfoWholeMatrices];
    if (active_rows) *active_rows = info[kInfoActiveRows];
    return RPVG_HIP_OK;
}
