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
enum : uint8_t { kRowListed = 1, kRowStoodFor = 2 };
constexpr int kCsrCellHashBits = 8;  // of the sort key's low field, for the rows of EM problems (CsrArrays)
constexpr double kEquivalentRelative = 1e-13;  // close rows that differ by no more than this are interchangeable
// (8 192 in rounds 2-3: 96 KB of static LDS and 1 024 threads, a whole CU per workgroup — on a batch without a single big list
// the kernel still waited 0.2-0.4 ms for CUs the other lane's kernels held, on the collapse's critical path)
constexpr uint32_t kListLdsRows = 2048;        // row lists up to this size live (and are sorted) in LDS
constexpr uint32_t kSortThreads = 256;  // (1 024 in rounds 2-3: sixteen waves' worth of registers to find on one CU before the kernel could even see that it has no big list)
constexpr uint32_t kBetweenThreads = 256;
constexpr uint32_t kBetweenRows = 2 * kBetweenThreads;  // rows of a matrix per work item of step b
constexpr uint32_t kPairChunk = 256;           // neighbour pairs staged in LDS at a time
constexpr uint32_t kWholeMatrixBit = 0x80000000u;  // mat_list: every row of the matrix is listed
constexpr uint32_t kBigListBit = 0x40000000u;      // mat_list: sorted by the bitonic kernel, compared in depth (no pair table)
constexpr uint32_t kListSizeMask = 0x3FFFFFFFu;
constexpr uint32_t kPairwiseRows = 1024;           // lists up to this size get a table of all their pairs
constexpr uint64_t kPairTableBytes = 16ull << 20;
constexpr uint8_t kPairLess = 1, kPairClose = 2;
constexpr uint32_t kLdsTableRows = 128;            // pair tables of lists up to this size are read from LDS (16 KB)

enum : uint32_t { kFlagNone = 0, kFlagActiveRows = 1, kFlagWholeMatrix = 2 };
enum : uint32_t { kInfoMatrices = 0, kInfoRowsReplaced = 1, kInfoWholeMatrices = 2, kInfoActiveRows = 3, kInfoPairsEquivalent = 4, kInfoPairsApart = 5, kInfoWords = 6 };

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
    __device__ __forceinline__ uint64_t patternOf(const uint32_t row) const { return pattern[row]; }
    __device__ __forceinline__ double countOf(const uint32_t row) const { return count[row]; }
};

// The rows of one EM problem (em_sparse.hip: the compacted CSR of a cluster restricted to a path subset): the same
// interface over sparse rows.  A row holds a handful of entries, in no particular column order: at() walks them.
struct CsrView {
    const uint32_t * off;      // [R + 1] entry range of every row, relative to `col` / `val`
    const uint32_t * col;
    const double * val;
    const double * noise;
    const double * count;
    const uint64_t * pattern;
    uint64_t R;
    uint32_t G;
    __device__ __forceinline__ double at(const uint32_t column, const uint32_t row) const {
        if (column >= G) return noise[row];
        double value = 0.0;
        for (uint32_t e = off[row]; e < off[row + 1]; ++e) {
            if (col[e] == column) value = val[e];
        }
        return value;
    }
    __device__ __forceinline__ uint64_t patternOf(const uint32_t row) const { return pattern[row]; }
    __device__ __forceinline__ double countOf(const uint32_t row) const { return count[row]; }
};

// Rows that are compared in depth agree in most columns, and a column costs two strided loads.  The comparisons below
// walk the columns in the reference's order but (1) skip the columns in which both rows are zero — equal for every
// purpose here — by their zero patterns, and (2) fetch kBatch columns of both rows before they look at any of them,
// so that the loads are in flight together.
constexpr uint32_t kBatch = 8;

template <typename View, typename Decide>  // decide(x, y) -> true: stop
__device__ __forceinline__ void forEachColumnPair(const View & mv, const uint32_t a, const uint32_t b, uint64_t live, Decide decide) {
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
template <typename View>
__device__ bool rowLess(const View & mv, const uint32_t a, const uint32_t b, const uint64_t pattern_a, const uint64_t pattern_b) {
    int result = -1;
    forEachColumnPair(mv, a, b, pattern_a | pattern_b, [&](const double x, const double y) {
        if (tolerantEqual(x, y)) return false;
        result = x < y ? 1 : 0;
        return true;
    });
    if (result >= 0) return result != 0;
    const double x = mv.countOf(a), y = mv.countOf(b);
    if (!tolerantEqual(x, y)) return x < y;
    return false;
}

template <typename View>
__device__ __forceinline__ bool rowLess(const View & mv, const uint32_t a, const uint32_t b) {
    return rowLess(mv, a, b, mv.patternOf(a), mv.patternOf(b));
}

// every column (noise included) within `precision` of each other, absolutely (src/path_estimator.cpp:232-239);
// *equivalent: additionally equal up to rounding in every column
template <typename View>
__device__ __forceinline__ bool rowsClose(const View & mv, const uint32_t a, const uint32_t b, const double precision, bool * equivalent) {
    bool eq = true, close = true;
    forEachColumnPair(mv, a, b, mv.patternOf(a) | mv.patternOf(b), [&](const double x, const double y) {
        // (both flags updated on every path: a store to whichever flag a branch picked kept them in scratch memory)
        const double d = fabs(x - y);
        const bool apart = d >= precision;
        close = close && !apart;
        eq = eq && !(d > kEquivalentRelative * fmin(fabs(x), fabs(y)));
        return apart;
    });
    if (close && equivalent) *equivalent = eq;
    return close;
}

// The rows of an EM problem are cluster rows cut down to the problem's columns: thousands of them agree up to the rounding of
// their last normalisation, not bit for bit.  A "cell" is a value's multiple of 2^-44 (5.7e-14, five orders below the
// precision of the collapse): rows with the same cells in every column stand for each other in the search for close rows.
constexpr double kCellsPerUnit = 0x1p44;
constexpr double kCellWidth = 0x1p-44;
__device__ __forceinline__ int64_t cellOf(const double v) { return static_cast<int64_t>(v * kCellsPerUnit); }

template <typename View>
__device__ bool rowsSameCells(const View & mv, const uint32_t a, const uint32_t b) {
    bool same = true;
    forEachColumnPair(mv, a, b, mv.patternOf(a) | mv.patternOf(b), [&](const double x, const double y) {
        if (cellOf(x) == cellOf(y)) return false;
        same = false;
        return true;
    });
    return same;
}

// (rows of a problem nearly always list their entries in the same column order: the entry lists are compared as they are, the
// column walk only decides when the lists differ in shape)
__device__ bool rowsSameCells(const CsrView & mv, const uint32_t a, const uint32_t b) {
    const uint32_t a0 = mv.off[a], a1 = mv.off[a + 1], b0 = mv.off[b];
    if (cellOf(mv.noise[a]) != cellOf(mv.noise[b])) return false;
    bool same_shape = a1 - a0 == mv.off[b + 1] - b0;
    for (uint32_t k = 0; same_shape && k < a1 - a0; ++k) same_shape = mv.col[a0 + k] == mv.col[b0 + k];
    if (same_shape) {
        for (uint32_t k = 0; k < a1 - a0; ++k) {
            if (cellOf(mv.val[a0 + k]) != cellOf(mv.val[b0 + k])) return false;
        }
        return true;
    }
    return rowsSameCells<CsrView>(mv, a, b);
}

template <typename View>
__device__ bool rowsIdentical(const View & mv, const uint32_t a, const uint32_t b) {
    bool same = true;
    forEachColumnPair(mv, a, b, mv.patternOf(a) | mv.patternOf(b), [&](const double x, const double y) {
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
    typedef MatrixView View;
    static constexpr int kHashBits = 0;  // stretches: bit for bit equal rows
    __device__ __forceinline__ uint64_t rowOffset(const uint32_t m) const { return mat_row_off[m]; }
    __device__ __forceinline__ uint64_t numRows(const uint32_t m) const { return mat_rows[m]; }
    __device__ __forceinline__ uint32_t numCols(const uint32_t m) const { return mat_cols[m]; }
    __device__ __forceinline__ MatrixView view(const uint32_t m) const {
        MatrixView mv;
        mv.values = values + mat_val_off[m];
        mv.noise = row_noise + mat_row_off[m];
        mv.count = row_count + mat_row_off[m];
        mv.pattern = zero_pattern + mat_row_off[m];
        mv.R = mat_rows[m];
        mv.G = mat_cols[m];
        return mv;
    }
};

// the EM problems of a solve as "matrices": problem p's rows sit at row_base[p] (the layout by the bound: there are unused
// row slots between the problems, whose sort keys carry a matrix index no problem has)
struct CsrArrays {
    const uint64_t * row_base;     // [P]
    const uint64_t * ent_base;     // [P]
    const uint32_t * kept_rows;    // [P]
    const uint64_t * col_off;      // [P+1]
    const uint32_t * prow_off;     // [rows + P]
    const double * prow_count;
    const double * prow_noise;
    const uint32_t * pent_col;
    const double * pent_val;
    const uint64_t * zero_pattern; // [rows]
    double * merged_count;         // [rows] read counts after the merges (written for the problems with a replay)
    uint32_t * problem_merged;     // [P] 1: rows of the problem were merged
    uint32_t * merged_problems;    // [0]: their number
    typedef CsrView View;
    static constexpr int kHashBits = kCsrCellHashBits;  // stretches: rows of equal cells (rowsSameCells), kept together by a hash of them in the key
    __device__ __forceinline__ uint64_t rowOffset(const uint32_t p) const { return row_base[p]; }
    __device__ __forceinline__ uint64_t numRows(const uint32_t p) const { return kept_rows[p]; }
    __device__ __forceinline__ uint32_t numCols(const uint32_t p) const { return static_cast<uint32_t>(col_off[p + 1] - col_off[p]); }
    __device__ __forceinline__ CsrView view(const uint32_t p) const {
        CsrView mv;
        const uint64_t rb = row_base[p], eb = ent_base[p];
        mv.off = prow_off + rb + p;
        mv.col = pent_col + eb;
        mv.val = pent_val + eb;
        mv.noise = prow_noise + rb;
        mv.count = prow_count + rb;
        mv.pattern = zero_pattern + rb;
        mv.R = kept_rows[p];
        mv.G = numCols(p);
        return mv;
    }
};

template <typename Arrays>
__device__ __forceinline__ typename Arrays::View viewOf(const uint32_t m, const Arrays & g) { return g.view(m); }

// ---- sort key fields (collapseSortKey, common.hpp) -------------------------------------------------------------
__device__ __forceinline__ uint32_t keyMatrix(const uint64_t key) { return static_cast<uint32_t>(key >> kCollapseMatrixShift); }
// HASH bits at the top of the low field belong to the sorted part (EM problems: a hash of the row's cells)
template <int HASH>
__device__ __forceinline__ uint64_t keySorted(const uint64_t key) { return key >> (kCollapseLargestBits - HASH); }  // (matrix, projection[, hash])
__device__ __forceinline__ uint64_t keyProjection(const uint64_t key) { return key >> kCollapseLargestBits; }  // (matrix, projection)
template <int HASH>
__device__ __forceinline__ int64_t keyLargest(const uint64_t key) { return static_cast<int64_t>(key & ((1ull << (kCollapseLargestBits - HASH)) - 1)); }

// |key_a - key_b| <= sum_c w_c |a_c - b_c| < 2 (G + 1) precision for close rows; + rounding of the keys, + 2 quanta
__device__ __forceinline__ uint64_t windowQuanta(const uint32_t G, const double precision) {
    const double window = 2.0001 * (G + 1) * precision + 1e-12;
    return static_cast<uint64_t>(window * static_cast<double>(1ull << kKeyFractionBits)) + 2;
}

// the largest values of close rows lie within the precision of each other: steps of the key's low field apart
// (the field wraps: half its range or more means "no filter")
template <int HASH>
__device__ __forceinline__ int64_t largestSteps(const double precision) {
    const double steps = precision * static_cast<double>(1ull << kCollapseLargestFractionBits) + (HASH ? 2.0 : 1.0);  // (rows of equal cells: up to a step from their first)
    return steps >= static_cast<double>(1ull << (kCollapseLargestBits - HASH - 1)) ? (1ll << (kCollapseLargestBits - HASH)) : static_cast<int64_t>(steps);
}

// `other` (at or behind `key` in the sorted order) can be close to `key`'s row
template <int HASH>
__device__ __forceinline__ bool inWindow(const uint64_t key, const uint64_t other, const uint64_t window_q, const int64_t largest_steps, bool * beyond) {
    *beyond = keyMatrix(other) != keyMatrix(key) || keyProjection(other) - keyProjection(key) > window_q;
    if (*beyond) return false;
    constexpr int bits = kCollapseLargestBits - HASH;
    int64_t d = (keyLargest<HASH>(other) - keyLargest<HASH>(key)) & ((1ll << bits) - 1);  // modulo the field
    if (d >= (1ll << (bits - 1))) d -= 1ll << bits;
    return d <= largest_steps && -d <= largest_steps;
}

// ---- stage 1: close pairs ---------------------------------------------------------------------------------------

template <typename Arrays>
struct PairScanArgs {
    uint32_t num_matrices;   // sort keys with a matrix index from here on belong to unused row slots (EM problems)
    uint64_t total_rows;
    double precision;
    const uint64_t * sort_key;   // sorted (matrix, key, largest value)
    const uint32_t * sort_row;   // position of the row in the row arrays (mat_row_off[m] + row)
    Arrays g;
    uint8_t * same_prev;     // [total rows] by sorted position
    uint32_t * stretch_first; // [total rows] by sorted position: first position of the stretch of bit-for-bit equal rows it lies in
    uint32_t * stretch_end;   // [total rows] at the first position of a stretch: one past its last
    uint32_t * position_of;   // [total rows] by row: its sorted position (EM problems)
    uint32_t * marked_bits;  // [total rows / 32] by row: has a close partner that is not its equal up to rounding
    uint32_t * marked_list;  // [total rows] sorted positions of the marked rows
    uint32_t * marked_count;
    uint32_t * pairs;        // [2 x pair_capacity] candidate pairs (sorted positions) of the scan that runs
    uint32_t * pair_count;   // its number of pairs
    uint32_t pair_capacity;
    uint8_t * active;        // [total rows] by row: kRowListed: marked, or close to a marked row; kRowStoodFor (EM problems): the first row of its stretch is
    uint32_t * mat_flag;     // [M]
    uint32_t * info;
    bool count_pairs;        // measuring aid: info[] counts the candidate pairs that were equal up to rounding / not close
};

// The window scans only collect candidate pairs; the comparisons, each a few dependent rounds of strided loads, then
// run one per thread, all at once (a thread that walked its window and compared as it went spent ~1 us per step:
// 0.2-0.4 ms for the longest windows of the batch).
template <typename Arrays>
__device__ __forceinline__ bool appendPair(const PairScanArgs<Arrays> & a, const uint64_t p, const uint64_t q) {
    const uint32_t slot = atomicAdd(a.pair_count, 1u);
    if (slot >= a.pair_capacity) return false;
    a.pairs[2 * static_cast<uint64_t>(slot)] = static_cast<uint32_t>(p);
    a.pairs[2 * static_cast<uint64_t>(slot) + 1] = static_cast<uint32_t>(q);
    return true;
}

// same_prev[p] = the row at sorted position p is bit for bit the row at p - 1 (same matrix)
template <typename Arrays>
__global__ void collapseSamePrevKernel(const PairScanArgs<Arrays> a) {
    const uint64_t p = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (p >= a.total_rows) return;
    uint8_t same = 0;
    constexpr int HASH = Arrays::kHashBits;
    // equal keys (with a hash: equal sorted parts — the largest values of rows of equal cells can lie a step apart):
    // nearly always the same values twice; the comparison leaves at the first difference
    if (p > 0 && (HASH ? keySorted<HASH>(a.sort_key[p]) == keySorted<HASH>(a.sort_key[p - 1]) : a.sort_key[p] == a.sort_key[p - 1]) &&
        keyMatrix(a.sort_key[p]) < a.num_matrices) {
        const uint32_t m = keyMatrix(a.sort_key[p]);
        const typename Arrays::View mv = viewOf(m, a.g);
        const uint64_t r0 = a.g.rowOffset(m);
        const uint32_t row = static_cast<uint32_t>(a.sort_row[p] - r0), before = static_cast<uint32_t>(a.sort_row[p - 1] - r0);
        same = (HASH ? rowsSameCells(mv, row, before) : rowsIdentical(mv, row, before)) ? 1 : 0;
    }
    a.same_prev[p] = same;
    if (HASH) a.position_of[a.sort_row[p]] = static_cast<uint32_t>(p);
    a.stretch_first[p] = same ? 0u : static_cast<uint32_t>(p);  // (the running maximum of these is the stretch's first position)
}

// stretch_end[] of every stretch, written by its last position
template <typename Arrays>
__global__ void collapseStretchEndKernel(const PairScanArgs<Arrays> a) {
    const uint64_t p = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (p >= a.total_rows) return;
    if (p + 1 == a.total_rows || !a.same_prev[p + 1]) a.stretch_end[a.stretch_first[p]] = static_cast<uint32_t>(p + 1);
}

// The first row of every stretch of equal rows lists the stretches after it whose keys lie within the window (a window too
// crowded to list sends the matrix to the full sort).  The scans step from stretch to stretch: the rows of an EM problem are
// cluster rows cut down to a few columns, thousands of them bit for bit the same, and a scan that walked over every copy
// took 3 ms on the problems of one batch.
template <typename Arrays>
__global__ void collapseForwardPairsKernel(const PairScanArgs<Arrays> a) {
    const uint64_t p = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (p + 1 >= a.total_rows || a.same_prev[p]) return;
    const uint64_t first = a.stretch_end[p];
    if (first >= a.total_rows) return;
    const uint64_t key = a.sort_key[p];
    const uint64_t next = a.sort_key[first];
    const uint32_t m = keyMatrix(key);
    if (keyMatrix(next) != m || m >= a.num_matrices) return;
    const uint64_t window_q = windowQuanta(a.g.numCols(m), a.precision);
    if (keyProjection(next) - keyProjection(key) > window_q) return;  // nearly every row leaves here
    constexpr int HASH = Arrays::kHashBits;
    const int64_t largest_steps = largestSteps<HASH>(a.precision);
    uint32_t listed = 0;
    for (uint64_t q = first; q < a.total_rows; q = a.stretch_end[q]) {
        const uint64_t other = a.sort_key[q];
        bool beyond;
        if (!inWindow<HASH>(key, other, window_q, largest_steps, &beyond)) {
            if (beyond) break;
            continue;
        }
        if (++listed > kMaxWindowCompares || !appendPair(a, p, q)) {
            atomicMax(&a.mat_flag[m], static_cast<uint32_t>(kFlagWholeMatrix));
            break;
        }
    }
}

// marks the row at sorted position p; the first marker puts it on the list
template <typename Arrays>
__device__ __forceinline__ void markRow(const PairScanArgs<Arrays> & a, const uint64_t p) {
    const uint32_t row = a.sort_row[p];
    const uint32_t bit = 1u << (row & 31);
    if ((atomicOr(&a.marked_bits[row >> 5], bit) & bit) == 0) a.marked_list[atomicAdd(a.marked_count, 1u)] = static_cast<uint32_t>(p);
}

template <typename Arrays>
__device__ __forceinline__ bool isMarked(const PairScanArgs<Arrays> & a, const uint32_t row) { return (a.marked_bits[row >> 5] >> (row & 31)) & 1u; }

// what a candidate pair of first rows has to pass for their stretches to count as close: the rows of a stretch lie within a
// cell of its first row (bit for bit equal rows: no margin)
template <typename Arrays>
__device__ __forceinline__ double pairPrecision(const double precision) { return Arrays::kHashBits ? precision + 2 * kCellWidth : precision; }

// the wave walks the stretches that start at the sorted positions its lanes hold in `first` (lanes with `wanted`):
// visit(position) for every row of them
template <typename Arrays, typename Visit>
__device__ __forceinline__ void forEachRowOfStretches(const PairScanArgs<Arrays> & a, const bool wanted, const uint32_t first, Visit visit) {
    const uint32_t lane = threadIdx.x & 63;
    for (uint64_t todo = __ballot(wanted); todo; todo &= todo - 1) {
        const int source = __ffsll(static_cast<long long>(todo)) - 1;
        const uint32_t begin = __shfl(first, source), end = a.stretch_end[begin];
        for (uint32_t s = begin + lane; s < end; s += 64) visit(s);
    }
}

// first position of the stretch that sorted position s lies in
template <typename Arrays>
__device__ __forceinline__ uint32_t p0(const PairScanArgs<Arrays> & a, const uint32_t s) { return a.stretch_first[s]; }

// both stretches of a close pair that is not equal up to rounding are marked (blocks of one wave)
template <typename Arrays>
__global__ void collapseMarkPairsKernel(const PairScanArgs<Arrays> a) {
    const uint32_t count = min(*a.pair_count, a.pair_capacity);
    // (lane l of block b takes pair b + l * blocks: a short list spreads over the waves, each of which walks the stretches of its pairs)
    for (uint32_t base = 0; base < count; base += gridDim.x * blockDim.x) {
        const uint32_t item = base + threadIdx.x * gridDim.x + blockIdx.x;
        uint32_t p = 0, q = 0;
        bool mark = false;
        if (item < count) {
            p = a.pairs[2 * static_cast<uint64_t>(item)];
            q = a.pairs[2 * static_cast<uint64_t>(item) + 1];
            const uint32_t m = keyMatrix(a.sort_key[p]);
            const typename Arrays::View mv = viewOf(m, a.g);
            const uint64_t r0 = a.g.rowOffset(m);
            bool equivalent = true;
            const bool close = rowsClose(mv, static_cast<uint32_t>(a.sort_row[p] - r0), static_cast<uint32_t>(a.sort_row[q] - r0), pairPrecision<Arrays>(a.precision), &equivalent);
            if (a.count_pairs) atomicAdd(&a.info[close ? kInfoPairsEquivalent : kInfoPairsApart], close && !equivalent ? 0u : 1u);
            mark = close && !equivalent;
            if (mark) atomicMax(&a.mat_flag[m], static_cast<uint32_t>(kFlagActiveRows));
        }
        if (Arrays::kHashBits) {
            // EM problems: the first row of a stretch goes through the replay for all of it (finishRuns); the others only must
            // not count as rows that part neighbours of the lists
            if (mark) {
                markRow(a, p);
                markRow(a, q);
            }
            forEachRowOfStretches(a, mark, p, [&](const uint32_t s) { if (s != p0(a, s)) a.active[a.sort_row[s]] = kRowStoodFor; });
            forEachRowOfStretches(a, mark, q, [&](const uint32_t s) { if (s != p0(a, s)) a.active[a.sort_row[s]] = kRowStoodFor; });
        } else {
            forEachRowOfStretches(a, mark, p, [&](const uint32_t s) { markRow(a, s); });
            forEachRowOfStretches(a, mark, q, [&](const uint32_t s) { markRow(a, s); });
        }
    }
}

// active = marked, or close to a marked row: the first row of every marked stretch (they are few: a list) lists its
// window, both ways, stretch by stretch
template <typename Arrays>
__global__ void collapseAroundPairsKernel(const PairScanArgs<Arrays> a) {
    const uint32_t count = *a.marked_count;
    for (uint32_t item = blockIdx.x * blockDim.x + threadIdx.x; item < count; item += gridDim.x * blockDim.x) {
        const uint64_t p = a.marked_list[item];
        a.active[a.sort_row[p]] = kRowListed;
        if (a.same_prev[p]) continue;  // a copy: the first row of its stretch lists for it
        const uint64_t key = a.sort_key[p];
        const uint32_t m = keyMatrix(key);
        const uint64_t window_q = windowQuanta(a.g.numCols(m), a.precision);
        constexpr int HASH = Arrays::kHashBits;
        const int64_t largest_steps = largestSteps<HASH>(a.precision);
        uint32_t listed = 0;
        bool crowded = false;
        for (uint64_t q = a.stretch_end[p]; q < a.total_rows && !crowded; q = a.stretch_end[q]) {
            bool beyond;
            const bool candidate = inWindow<HASH>(key, a.sort_key[q], window_q, largest_steps, &beyond);
            if (beyond) break;
            if (!candidate || isMarked(a, a.sort_row[q])) continue;
            if (++listed > kMaxWindowCompares || !appendPair(a, p, q)) crowded = true;
        }
        for (uint64_t below = p; below > 0 && !crowded;) {
            const uint64_t q = a.stretch_first[below - 1];
            below = q;
            bool beyond;
            const bool candidate = inWindow<HASH>(a.sort_key[q], key, window_q, largest_steps, &beyond);
            if (beyond) break;
            if (!candidate || isMarked(a, a.sort_row[q])) continue;
            if (++listed > kMaxWindowCompares || !appendPair(a, p, q)) crowded = true;
        }
        if (crowded) atomicMax(&a.mat_flag[m], static_cast<uint32_t>(kFlagWholeMatrix));
    }
}

template <typename Arrays>
__global__ void collapseActivePairsKernel(const PairScanArgs<Arrays> a) {  // (blocks of one wave)
    const uint32_t count = min(*a.pair_count, a.pair_capacity);
    for (uint32_t base = 0; base < count; base += gridDim.x * blockDim.x) {
        const uint32_t item = base + threadIdx.x * gridDim.x + blockIdx.x;
        uint32_t q = 0;
        bool close = false;
        if (item < count) {
            const uint32_t p = a.pairs[2 * static_cast<uint64_t>(item)];
            q = a.pairs[2 * static_cast<uint64_t>(item) + 1];
            const uint32_t m = keyMatrix(a.sort_key[p]);
            const typename Arrays::View mv = viewOf(m, a.g);
            const uint64_t r0 = a.g.rowOffset(m);
            close = rowsClose(mv, static_cast<uint32_t>(a.sort_row[p] - r0), static_cast<uint32_t>(a.sort_row[q] - r0), pairPrecision<Arrays>(a.precision), nullptr);
        }
        // q is the first row of its stretch: the rows it stands for are active with it (several writers, one value)
        forEachRowOfStretches(a, close, q, [&](const uint32_t s) { a.active[a.sort_row[s]] = Arrays::kHashBits && s != p0(a, s) ? kRowStoodFor : kRowListed; });
    }
}

// ---- stage 2: replay ------------------------------------------------------------------------------------------

constexpr uint32_t kNoRow = 0xFFFFFFFFu;

template <typename Arrays>
struct ReplayArgs {
    uint32_t num_matrices;
    double precision;
    Arrays g;
    const uint32_t * mat_flag;
    const uint8_t * active;      // [total rows] by row
    const uint32_t * sort_row;   // (EM problems: the stretches behind the rows of the lists)
    const uint32_t * position_of;
    const uint32_t * stretch_end;
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
    bool debug;                  // RPVG_HIP_EM_COLLAPSE_DEBUG: slow workgroups of the runs kernel report where their time went
    bool no_lds_tables;          // RPVG_HIP_COLLAPSE_NO_LDS_TABLES (A/B): pair tables read from global memory as in round 3
    // group matrices whose diploid search ran on the values as built (rpvg_hip_groups::held_back_runs): its sums, to be adjusted
    // for every row the runs rewrite; part_pair == NULL: nobody has read the matrices yet
    double * part_pair;
    double * part_marginal;
    const uint64_t * pair_part_off;
    const uint64_t * col_part_off;
    uint32_t chunk_rows;
    uint32_t * rewritten_count;  // [M] rows of the matrix's list that take the values of a head (listRewrittenRows)
    bool walk_only;              // collapseRunsKernel stops behind the runs (collapseAdjustSumsKernel and collapseRewriteKernel follow)
};

// a1. every pair of rows of a small list, one thread each: order and closeness in one pass over the columns.  (A
// workgroup that sorts its list with a comparison network goes through dozens of dependent rounds of strided loads;
// here every comparison of the batch is in flight at once, and ranks and runs are then read off the table.)
// row i of matrix m's pair table; the calling threads stride its columns (thread `first` of `stride`)
template <typename Arrays>
__device__ __forceinline__ void pairTableRow(const ReplayArgs<Arrays> & a, const uint32_t m, const uint32_t i, const uint32_t first, const uint32_t stride) {
    const uint32_t n = a.mat_list[m] & kListSizeMask;
    const typename Arrays::View mv = viewOf(m, a.g);
    const uint32_t * list = a.order + 2 * a.g.rowOffset(m);
    uint8_t * table = a.pair_table + a.pair_base[m] + static_cast<uint64_t>(i) * n;
    const uint32_t row_i = list[i];
    const uint64_t pattern_i = mv.patternOf(row_i);
    for (uint32_t j = first; j < n; j += stride) {
        const uint32_t row_j = list[j];
        int less = -1;
        bool close = true;
        forEachColumnPair(mv, row_i, row_j, pattern_i | mv.patternOf(row_j), [&](const double x, const double y) {
            if (less < 0 && !tolerantEqual(x, y)) less = x < y ? 1 : 0;
            if (fabs(x - y) >= a.precision) close = false;
            return less >= 0 && !close;
        });
        if (less < 0) {
            const double x = mv.countOf(row_i), y = mv.countOf(row_j);
            less = (!tolerantEqual(x, y) && x < y) ? 1 : 0;
        }
        table[j] = (less ? kPairLess : 0) | (close ? kPairClose : 0);
    }
}

template <typename Arrays>
__global__ __launch_bounds__(64) void collapsePairTableKernel(const ReplayArgs<Arrays> a) {
    const uint32_t num_items = *a.row_item_count;
    for (uint32_t item = blockIdx.x; item < num_items; item += gridDim.x) {
        pairTableRow(a, a.row_items[2 * static_cast<uint64_t>(item)], a.row_items[2 * static_cast<uint64_t>(item) + 1], threadIdx.x, blockDim.x);
    }
}

// a2. small lists sorted by rank: the number of rows that sort before a row (equal rows in list order).  Where the
// tolerant comparison is inconsistent the ranks may collide; such a list is sorted by the comparison network instead.
// the list of matrix m (a small one: it has a pair table) sorted by rank, and the column that orders every pair of neighbours
template <typename Arrays>
__device__ void rankList(const ReplayArgs<Arrays> & a, const uint32_t m) {
    __shared__ uint32_t lds_row[kPairwiseRows], lds_slot[kPairwiseRows];
    __shared__ uint8_t lds_table[kLdsTableRows * kLdsTableRows];
    __shared__ uint32_t collision;
    const uint32_t encoded = a.mat_list[m];
    if (encoded & kBigListBit) return;
    const uint32_t n = encoded;
    const uint64_t r0 = a.g.rowOffset(m);
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
    // (the table of a short list — nearly every list — moves to LDS first: a rank is 2 n dependent byte loads, and from global
    // memory the 100-row list of a batch made this kernel 0.1-0.2 ms of the collapse's critical path)
    const bool table_in_lds = n <= kLdsTableRows && !a.no_lds_tables;
    if (table_in_lds) {
        for (uint32_t k = threadIdx.x; k < n * n; k += blockDim.x) lds_table[k] = table[k];
        __syncthreads();
    }
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        uint32_t rank = 0;
        if (table_in_lds) {
            for (uint32_t j = 0; j < n; ++j) {
                const bool before = (lds_table[j * n + i] & kPairLess) != 0;
                const bool after = (lds_table[i * n + j] & kPairLess) != 0;
                rank += (before || (!after && j < i)) ? 1u : 0u;
            }
        } else {
            for (uint32_t j = 0; j < n; ++j) {
                const bool before = (table[static_cast<uint64_t>(j) * n + i] & kPairLess) != 0;
                const bool after = (table[static_cast<uint64_t>(i) * n + j] & kPairLess) != 0;
                rank += (before || (!after && j < i)) ? 1u : 0u;
            }
        }
        if (rank >= n || atomicExch(&lds_slot[rank], i) != kNoRow) collision = 1;
    }
    __syncthreads();
    if (collision) {  // (never seen; kept correct rather than fast) insertion sort with the comparator itself
        if (threadIdx.x == 0) {
            const typename Arrays::View mv = viewOf(m, a.g);
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
    const typename Arrays::View mv = viewOf(m, a.g);
    for (uint32_t p = 1 + threadIdx.x; p < n; p += blockDim.x) {
        const uint32_t x = lds_row[lds_slot[p - 1]], y = lds_row[lds_slot[p]];
        uint32_t d = kNoRow;
        bool can_join = true;
        // (forEachColumnPair visits the columns in order but skips those in which both rows are zero: the column
        // index is recovered from the patterns)
        const uint64_t live = mv.patternOf(x) | mv.patternOf(y);
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
            a.pair_pattern[r0 + p] = mv.patternOf(x) & (d >= 64 ? ~0ull : (1ull << d) - 1ull);
        }
    }
}

template <typename Arrays>
__global__ __launch_bounds__(256) void collapseRankKernel(const ReplayArgs<Arrays> a) {
    const uint32_t num_items = a.replay_list[a.num_matrices];
    for (uint32_t item = blockIdx.x; item < num_items; item += gridDim.x) rankList(a, a.replay_list[item]);
}

// a3. big lists (more than kPairwiseRows rows, whole matrices): sorted with the reference's comparator in a
// comparison network, and for every pair of neighbours the column that orders them
template <typename Arrays>
__global__ __launch_bounds__(kSortThreads) void collapseSortKernel(const ReplayArgs<Arrays> a) {
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
    const typename Arrays::View mv = viewOf(m, a.g);
    const uint64_t r0 = a.g.rowOffset(m);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, num_waves = blockDim.x >> 6;
    uint64_t padded = 1;
    while (padded < n) padded <<= 1;
    uint32_t * global_order = a.order + 2 * r0;
    uint32_t * order = padded <= kListLdsRows ? lds_list : global_order;
    for (uint64_t i = threadIdx.x; i < padded; i += blockDim.x) order[i] = i < n ? (whole ? static_cast<uint32_t>(i) : global_order[i]) : kNoRow;
    __syncthreads();
    const bool in_lds = order == lds_list;
    if (in_lds) {
        for (uint64_t i = threadIdx.x; i < padded; i += blockDim.x) lds_pattern[i] = order[i] == kNoRow ? 0ull : mv.patternOf(order[i]);
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
                    const uint64_t px = in_lds ? lds_pattern[i] : mv.patternOf(x), py = in_lds ? lds_pattern[l] : mv.patternOf(y);
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
                a.pair_pattern[r0 + p] = mv.patternOf(x) & (d >= 64 ? ~0ull : (1ull << d) - 1ull);
            }
        }
    }
    }
}

// b. inactive rows between neighbours of the lists: work item = (matrix with a list, slice of its rows)
// the rows [slice * kBetweenRows, ...) of matrix m against the neighbour pairs of its list
template <typename Arrays>
__device__ void betweenSlice(const ReplayArgs<Arrays> & a, const uint32_t m, const uint32_t slice) {
    __shared__ uint32_t lds_column[kPairChunk];
    __shared__ double lds_lo[kPairChunk], lds_hi[kPairChunk];
    __shared__ uint64_t lds_pattern[kPairChunk], lds_before[kPairChunk];
    const uint64_t n = a.mat_list[m] & kListSizeMask;  // (not a whole matrix: those have no items)
    const typename Arrays::View mv = viewOf(m, a.g);
    const uint64_t r0 = a.g.rowOffset(m);
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
            const uint64_t pattern_x = mv.patternOf(x);
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

template <typename Arrays>
__global__ __launch_bounds__(kBetweenThreads) void collapseBetweenKernel(const ReplayArgs<Arrays> a) {
    const uint32_t num_items = *a.between_count;
    for (uint32_t item = blockIdx.x; item < num_items; item += gridDim.x) {
        betweenSlice(a, a.between_items[2 * static_cast<uint64_t>(item)], a.between_items[2 * static_cast<uint64_t>(item) + 1]);
    }
}

// The diploid search has summed count * log(noise + (u_a + u_b) / 2) over the rows AS BUILT (pairTile2Kernel: per chunk of rows,
// pair of columns a <= b, and column; rpvg_hip_groups::held_back_runs).  A row that takes the values of its head contributes the
// head's term with its own count: the difference, count * log(after / before), goes to the sums of the row's chunk.  Work item =
// (matrix with a list, slice of its G x G table of cells): the rewritten rows are staged in LDS some at a time (old and new
// values: the global loads of a batch of rows in flight together), a thread owns its cells throughout — every sum has one
// writer and the order of its additions is the list's: the result does not depend on how threads run — and keeps a cell's
// difference in a register while the rows stay in one chunk (most matrices have one).  The rows are few (some hundred per
// configs[2] batch) but a matrix may hold most of them: on one workgroup, cell by cell from global memory with two logarithms
// each, they were 1.5 ms behind the search.
constexpr uint32_t kAdjustSlices = 16;

__device__ void adjustSearchSums(const ReplayArgs<MatrixArrays> & a, const uint32_t m, const uint32_t slice) {
    const uint64_t n = a.mat_list[m] & kListSizeMask;
    const MatrixView mv = viewOf(m, a.g);
    const uint64_t R = mv.R;
    const uint64_t r0 = a.g.rowOffset(m);
    const uint32_t * order = a.order + 2 * r0;
    const uint32_t * head_of = a.head_of + r0;
    const double * M = a.g.values + a.g.mat_val_off[m];
    const double * nz = a.g.row_noise + r0;
    const uint32_t G = mv.G;
    constexpr uint32_t kStageDoubles = 4096;
    constexpr uint32_t kStageRows = 32;
    __shared__ double stage[kStageDoubles];                      // [row of the batch][old | new][column]
    __shared__ double stage_noise[2 * kStageRows], stage_count[kStageRows];
    __shared__ uint32_t stage_chunk[kStageRows];
    const uint32_t rows_per_stage = G <= kStageDoubles / 2 ? min(kStageRows, kStageDoubles / (2 * G)) : 0u;
    double * pairs = a.part_pair + a.pair_part_off[m];
    double * singles = a.part_marginal + a.col_part_off[m];
    const double * cnt = a.g.row_count + r0;
    // count * log(after / before): after / before - 1 is tiny for rows within prob_precision of each other unless the values
    // themselves are — the series then, the logarithm otherwise
    auto difference = [](const double c, const double before, const double after) {
        if (before == after) return 0.0;
        const double x = (after - before) / before;
        if (fabs(x) < 1e-4) return c * (x * (1.0 - x * (0.5 - x * (1.0 / 3.0 - 0.25 * x))));
        return c * log1p(x);
    };
    // the list positions whose rows are rewritten, in list order (the walk of the runs listed them: runsOfMatrix)
    const uint32_t * rewritten = a.pair_column + r0;
    const uint32_t num_rewritten = a.rewritten_count[m];
    (void) n;
    for (uint32_t i0 = 0; rows_per_stage && i0 < num_rewritten; i0 += rows_per_stage) {
        const uint32_t rows = min(rows_per_stage, num_rewritten - i0);
        __syncthreads();  // (the batch before is done with the LDS)
        if (threadIdx.x < rows) {
            const uint64_t q = rewritten[i0 + threadIdx.x];
            const uint32_t dst = order[q], src = order[head_of[q]];
            stage_chunk[threadIdx.x] = dst / a.chunk_rows;
            stage_noise[2 * threadIdx.x] = nz[dst];
            stage_noise[2 * threadIdx.x + 1] = nz[src];
            stage_count[threadIdx.x] = cnt[dst];
        }
        __syncthreads();
        for (uint32_t idx = threadIdx.x; idx < rows * G; idx += blockDim.x) {
            const uint32_t row = idx / G, c = idx % G;
            const uint64_t q = rewritten[i0 + row];
            stage[(2 * row) * G + c] = M[static_cast<uint64_t>(c) * R + order[q]];
            stage[(2 * row + 1) * G + c] = M[static_cast<uint64_t>(c) * R + order[head_of[q]]];
        }
        __syncthreads();
        for (uint64_t cell = static_cast<uint64_t>(slice) * blockDim.x + threadIdx.x; cell < static_cast<uint64_t>(G) * G; cell += static_cast<uint64_t>(kAdjustSlices) * blockDim.x) {
            const uint32_t x = static_cast<uint32_t>(cell / G), y = static_cast<uint32_t>(cell % G);
            if (x > y) continue;
            double sum = 0.0, single_sum = 0.0;
            uint32_t sum_chunk = kNoRow;
            auto flush = [&]() {
                if (sum_chunk == kNoRow) return;
                if (sum != 0.0) pairs[static_cast<uint64_t>(sum_chunk) * G * G + cell] += sum;
                if (x == y && single_sum != 0.0) singles[static_cast<uint64_t>(sum_chunk) * G + x] += single_sum;
                sum = 0.0;
                single_sum = 0.0;
            };
            for (uint32_t row = 0; row < rows; ++row) {
                const uint32_t chunk = stage_chunk[row];
                if (chunk != sum_chunk) {
                    flush();
                    sum_chunk = chunk;
                }
                const double * before_row = stage + (2 * row) * G, * after_row = before_row + G;
                const double old_noise = stage_noise[2 * row], new_noise = stage_noise[2 * row + 1], c = stage_count[row];
                // (the search's own arithmetic: 2 x = (u_a + 2 noise) + u_b)
                sum += difference(c, 0.5 * (fma(2.0, old_noise, before_row[x]) + before_row[y]), 0.5 * (fma(2.0, new_noise, after_row[x]) + after_row[y]));
                if (x == y) single_sum += difference(c, before_row[x] + old_noise, after_row[x] + new_noise);
            }
            flush();
        }
    }
}

// the list positions of matrix m whose rows take the values of their head, in list order, into the scratch of a stage that is
// done (pair_column), and their number
__device__ void listRewrittenRows(const ReplayArgs<MatrixArrays> & a, const uint32_t m, const uint64_t n, const uint32_t * head_of) {
    __shared__ uint32_t rewritten_total, wave_rewritten[4];
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t * rewritten = a.pair_column + a.g.rowOffset(m);
    if (threadIdx.x == 0) rewritten_total = 0;
    __syncthreads();
    for (uint64_t q0 = 0; q0 < n; q0 += blockDim.x) {
        const uint64_t q = q0 + threadIdx.x;
        const bool mine = q < n && head_of[q] != q;
        const unsigned long long ballot = __ballot(mine);
        if (lane == 0) wave_rewritten[threadIdx.x >> 6] = __popcll(ballot);
        __syncthreads();
        uint32_t at = rewritten_total + __popcll(ballot & ((1ull << lane) - 1ull));
        for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) at += wave_rewritten[w];
        if (mine) rewritten[at] = static_cast<uint32_t>(q);
        __syncthreads();
        if (threadIdx.x == 0) rewritten_total += wave_rewritten[0] + wave_rewritten[1] + wave_rewritten[2] + wave_rewritten[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) a.rewritten_count[m] = rewritten_total;
}
__device__ void listRewrittenRows(const ReplayArgs<CsrArrays> &, const uint32_t, const uint64_t, const uint32_t *) {}

// What a run does to its rows.  Group matrices: the rows of a run take the values of its head — the consumers are sums over
// rows of count * log(noise + columns), so "merged into the head" is "takes the values of the head" with the row's own
// count; a wave per row, its lanes over the columns.
__device__ __forceinline__ void finishRuns(const ReplayArgs<MatrixArrays> & a, const uint32_t m, const uint64_t n, const uint32_t * order,
                                           const uint32_t * head_of, uint32_t * demote, const bool whole) {
    (void) whole;
    const int lane = threadIdx.x & 63;
    const MatrixView mv = viewOf(m, a.g);
    const uint64_t R = mv.R;
    const uint64_t r0 = a.g.rowOffset(m);
    double * M = a.g.values + a.g.mat_val_off[m];
    double * nz = a.g.row_noise + r0;
    double * rm = a.rowmax + r0;
    const uint32_t fast_mid_end = a.mat_mid[m];
    uint32_t replaced = 0;
    for (uint64_t q = threadIdx.x >> 6; q < n; q += blockDim.x >> 6) {
        const uint32_t h = head_of[q];
        if (h == q) continue;
        const uint32_t dst = order[q], src = order[h];
        // (a column is a cache line of its own, in either row: eight loads in flight per lane — one at a time, a merged row of a
        // 2 000-column matrix took a wave 60 us)
        for (uint32_t c0 = lane; c0 < mv.G; c0 += 64 * 8) {
            double v[8];
#pragma unroll
            for (uint32_t k = 0; k < 8; ++k) {
                const uint32_t c = c0 + 64 * k;
                v[k] = c < mv.G ? __builtin_nontemporal_load(&M[static_cast<uint64_t>(c) * R + src]) : 0.0;
            }
#pragma unroll
            for (uint32_t k = 0; k < 8; ++k) {
                const uint32_t c = c0 + 64 * k;
                if (c < mv.G) M[static_cast<uint64_t>(c) * R + dst] = v[k];
            }
        }
        if (lane == 0) {
            const double noise = nz[src];
            nz[dst] = noise;
            rm[dst] = rm[src];
            ++replaced;
            // a product-path row (LogProduct, common.hpp) needs noise >= kProductMinNoise: if the head's is below,
            // the whole matrix takes the logarithm path
            if (dst < fast_mid_end && !(noise >= kProductMinNoise)) *demote = 1;
        }
    }
    if (replaced) atomicAdd(&a.info[kInfoRowsReplaced], replaced);
    __syncthreads();
    if (threadIdx.x == 0 && *demote) {
        a.mat_fast[m] = 0;
        a.mat_mid[m] = 0;
    }
}

// EM problems (src/path_abundance_estimator.cpp:266,668 -> src/path_estimator.cpp:219-259): the head keeps its values and
// the counts of its run add up — in CSR terms a merged row's count moves to its run head (read counts are integers: the
// sums are exact in any order).  A row of the list stands for its stretch (rows of equal cells: all of them join the run
// with it); in a whole-matrix list every row is there itself.  The counts after the merges go to a second array (the
// first EM pass may still be reading the original ones); the problem is flagged for the second pass.
__device__ __forceinline__ void finishRuns(const ReplayArgs<CsrArrays> & a, const uint32_t p, const uint64_t n, const uint32_t * order,
                                           const uint32_t * head_of, uint32_t * demote, const bool whole) {
    (void) demote;
    __shared__ uint32_t any_merge, members_total;
    const uint64_t r0 = a.g.rowOffset(p);
    const uint64_t R = a.g.numRows(p);
    double * merged = a.g.merged_count + r0;
    const double * count = a.g.prow_count + r0;
    uint8_t * joined = a.barrier + r0;  // (free again: the runs are known) by list position: 1 = a head that rows joined
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, num_waves = blockDim.x >> 6;
    if (threadIdx.x == 0) any_merge = 0, members_total = 0;
    for (uint64_t r = threadIdx.x; r < R; r += blockDim.x) merged[r] = count[r];
    for (uint64_t q = threadIdx.x; q < n; q += blockDim.x) joined[q] = 0;
    __syncthreads();
    for (uint64_t q = threadIdx.x; q < n; q += blockDim.x) {
        if (head_of[q] != q) joined[head_of[q]] = 1, any_merge = 1;  // (several writers, one value)
    }
    __syncthreads();
    if (any_merge) {  // (uniform)
        // the rows of the runs with more than a head, stretch by stretch (a wave each): first their counts go...
        auto forStretchOf = [&](const uint64_t q, auto visit) {  // visit(row of the problem)
            if (whole) {
                if (lane == 0) visit(order[q]);
                return;
            }
            const uint32_t first = a.position_of[r0 + order[q]], end = a.stretch_end[first];
            for (uint32_t s = first + lane; s < end; s += 64) visit(static_cast<uint32_t>(a.sort_row[s] - r0));
        };
        for (uint64_t q = wave; q < n; q += num_waves) {
            if (head_of[q] == q && !joined[q]) continue;
            forStretchOf(q, [&](const uint32_t row) { merged[row] = 0.0; });
        }
        __syncthreads();
        // ... then they arrive at the head
        for (uint64_t q = wave; q < n; q += num_waves) {
            const uint32_t h = head_of[q];
            if (h == q && !joined[q]) continue;
            double sum = 0.0;
            uint32_t rows = 0;
            forStretchOf(q, [&](const uint32_t row) { sum += count[row]; ++rows; });
            for (int step = 32; step; step >>= 1) {
                sum += __shfl_xor(sum, step);
                rows += __shfl_xor(rows, step);
            }
            if (lane == 0) {
                atomicAdd(&merged[order[h]], sum);
                atomicAdd(&members_total, h == q ? rows - 1 : rows);
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            a.g.problem_merged[p] = 1;
            atomicAdd(a.g.merged_problems, 1u);
            atomicAdd(&a.info[kInfoRowsReplaced], members_total);
        }
    }
    __syncthreads();
}

// c. runs and values of matrix m (the workgroup's)
template <typename Arrays>
__device__ void runsOfMatrix(const ReplayArgs<Arrays> & a, const uint32_t m) {
    __shared__ uint32_t demote;
    __shared__ uint32_t walk_at, run_end;
    const uint32_t encoded = a.mat_list[m];
    const uint64_t n = encoded & kListSizeMask;
    const bool tabled = !(encoded & kBigListBit);
    const typename Arrays::View mv = viewOf(m, a.g);
    const uint64_t r0 = a.g.rowOffset(m);
    const uint32_t * order = a.order + 2 * r0;
    const uint32_t * list_index = a.list_index + r0;
    const uint8_t * table = a.pair_table + (tabled ? a.pair_base[m] : 0);
    uint32_t * head_of = a.head_of + r0;
    uint8_t * close = a.close + r0;
    const uint8_t * barrier = a.barrier + r0;
    // list positions p, q hold rows within prob_precision of each other: read off the pair table, or compared
    auto closeRows = [&](const uint64_t p, const uint64_t q) {
        if (tabled) return (table[static_cast<uint64_t>(list_index[p]) * n + list_index[q]] & kPairClose) != 0;
        return rowsClose(mv, order[p], order[q], a.precision, nullptr);
    };
    __syncthreads();
    const long long clock_begin = wall_clock64();
    if (threadIdx.x == 0) demote = 0;
    if (tabled && n <= kLdsTableRows && !a.no_lds_tables) {
        // A short list with a pair table — nearly every list: table, barriers and heads in LDS, and ONE thread walks the runs
        // (src/path_estimator.cpp:226-255: a row joins the run of the current head if nothing parts them and it is close to the
        // head, otherwise it becomes the head).  The walk below, a workgroup stepping from head to head through barriers and
        // global memory, took 0.1-0.3 ms for the 100-row list of a batch.
        __shared__ uint8_t lds_table[kLdsTableRows * kLdsTableRows];
        __shared__ uint8_t lds_barrier[kLdsTableRows];
        __shared__ uint32_t lds_index[kLdsTableRows], lds_head[kLdsTableRows];
        const uint32_t nn = static_cast<uint32_t>(n);
        for (uint32_t k = threadIdx.x; k < nn * nn; k += blockDim.x) lds_table[k] = table[k];
        for (uint32_t p = threadIdx.x; p < nn; p += blockDim.x) {
            lds_barrier[p] = barrier[p];
            lds_index[p] = list_index[p];
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t h = 0;
            while (h < nn) {
                uint32_t q = h + 1;
                while (q < nn && !lds_barrier[q] && (lds_table[lds_index[h] * nn + lds_index[q]] & kPairClose) != 0) ++q;
                for (uint32_t r = h; r < q; ++r) lds_head[r] = h;
                h = q;
            }
        }
        __syncthreads();
        for (uint32_t p = threadIdx.x; p < nn; p += blockDim.x) head_of[p] = lds_head[p];
        __syncthreads();
        if (a.walk_only) listRewrittenRows(a, m, n, head_of);
        else finishRuns(a, m, n, order, head_of, &demote, (encoded & kWholeMatrixBit) != 0);
        return;
    }
    for (uint64_t p = threadIdx.x; p < n; p += blockDim.x) {
        head_of[p] = kNoRow;
        close[p] = (p > 0 && !barrier[p] && closeRows(p - 1, p)) ? 1 : 0;
    }
    __syncthreads();
    // Runs (src/path_estimator.cpp:226-255): a row joins the run of the current head if it is close to the head,
    // otherwise it becomes the head.  The workgroup walks from head to head: a head whose successor is not close to it
    // (nearly every one) is a run of its own and costs one flag; a head with followers has the whole workgroup look
    // for the end of its run, 256 candidates at a time.  (Round 2 had every list position find "where would the next
    // head be if I were one" on its own before one thread walked the heads: with n rows all close to each other —
    // an EM problem over 10^5 reads that differ in the ninth digit — that is n^2 / 2 comparisons for the one run.)
    if (threadIdx.x == 0) walk_at = 0;
    __syncthreads();
    while (walk_at < n) {  // (uniform: walk_at only changes between barriers)
        const uint64_t h = walk_at;
        if (h + 1 >= n || !close[h + 1]) {
            // a stretch of single-row runs, up to the next row with a close successor: the workgroup looks for it, 256
            // positions at a time (one thread stepping through a list of a few hundred single rows, a dependent load per
            // step, was most of this kernel's 0.23 ms on the group matrices)
            if (threadIdx.x == 0) run_end = static_cast<uint32_t>(n);
            __syncthreads();
            for (uint64_t q0 = h + 1; q0 < n && run_end == n; q0 += blockDim.x) {  // (uniform: run_end read behind the barrier below)
                const uint64_t q = q0 + threadIdx.x;
                if (q + 1 < n && close[q + 1]) atomicMin(&run_end, static_cast<uint32_t>(q));
                __syncthreads();
            }
            const uint64_t stretch_end = run_end;
            for (uint64_t q = h + threadIdx.x; q < stretch_end; q += blockDim.x) head_of[q] = static_cast<uint32_t>(q);
            __syncthreads();
            if (threadIdx.x == 0) walk_at = static_cast<uint32_t>(stretch_end);
            __syncthreads();
            continue;
        }
        if (threadIdx.x == 0) run_end = static_cast<uint32_t>(n);
        __syncthreads();
        for (uint64_t q0 = h + 2; q0 < n && run_end == n; q0 += blockDim.x) {  // (uniform: run_end read behind the barrier below)
            const uint64_t q = q0 + threadIdx.x;
            if (q < n && (barrier[q] != 0 || !closeRows(h, q))) atomicMin(&run_end, static_cast<uint32_t>(q));
            __syncthreads();
        }
        const uint64_t end = run_end;
        for (uint64_t q = h + threadIdx.x; q < end; q += blockDim.x) head_of[q] = static_cast<uint32_t>(h);
        __syncthreads();
        if (threadIdx.x == 0) walk_at = static_cast<uint32_t>(end);
        __syncthreads();
    }
    const long long clock_walked = wall_clock64();
    if (a.walk_only) listRewrittenRows(a, m, n, head_of);
    else finishRuns(a, m, n, order, head_of, &demote, (encoded & kWholeMatrixBit) != 0);
    if (a.debug && threadIdx.x == 0 && wall_clock64() - clock_begin > 2000) {
        printf("[runs] matrix %u rows %llu cols %u list %llu tabled %d: walk %lld finish %lld (10 ns)\n", m, static_cast<unsigned long long>(mv.R), mv.G,
               static_cast<unsigned long long>(n), tabled ? 1 : 0, clock_walked - clock_begin, wall_clock64() - clock_walked);
    }
}

// a0. the list of a matrix: its active rows (the replay then only touches those), or all of them.  Persistent workgroups, a
// grid-stride loop over the matrices: most have no flag and cost a load.
// (Round 4 tried the whole replay of a short list right here, twice — pair table, ranks, rows between neighbours, runs, one stage
// after the other behind workgroup barriers instead of six dependent launches; first a workgroup per matrix with the tables in
// global memory, then persistent workgroups with the tables in LDS: 12.5 against 11.5 and 11.4 against 10.3 ms per configs[2]
// batch.  The staged kernels spread a matrix's pairs and row slices over many workgroups; one workgroup walking the stages of
// its matrices in turn is the longer chain, and the batch waits for the longest.)
template <typename Arrays>
__global__ __launch_bounds__(256) void collapseListKernel(const ReplayArgs<Arrays> a) {
    __shared__ uint32_t list_size;
    for (uint32_t m = blockIdx.x; m < a.num_matrices; m += gridDim.x) {
        const uint32_t flag = a.mat_flag[m];
        if (flag == kFlagNone) {
            if (threadIdx.x == 0) a.mat_list[m] = 0;
            continue;
        }
        const uint64_t R = a.g.numRows(m);
        const uint64_t r0 = a.g.rowOffset(m);
        const uint8_t * active = a.active + r0;
        uint32_t * list = a.order + 2 * r0;
        __syncthreads();  // (the matrix before is done with the shared words)
        if (threadIdx.x == 0) list_size = 0;
        __syncthreads();
        if (flag != kFlagWholeMatrix) {
            // (eight flags per load where the alignment allows: the largest matrix of a batch has 10^5 rows, and a byte per thread
            // and step was 50 us of this kernel)
            const uint64_t head = min(R, static_cast<uint64_t>((8 - (reinterpret_cast<uintptr_t>(active) & 7)) & 7));
            for (uint64_t i = threadIdx.x; i < head; i += blockDim.x) {
                if (active[i] == kRowListed) list[atomicAdd(&list_size, 1u)] = static_cast<uint32_t>(i);
            }
            const uint64_t words = (R - head) / 8;
            const uint64_t * active_words = reinterpret_cast<const uint64_t *>(active + head);
            for (uint64_t w = threadIdx.x; w < words; w += blockDim.x) {
                uint64_t flags = active_words[w];
                for (uint32_t k = 0; flags; ++k, flags >>= 8) {
                    if ((flags & 0xFF) == kRowListed) list[atomicAdd(&list_size, 1u)] = static_cast<uint32_t>(head + 8 * w + k);
                }
            }
            for (uint64_t i = head + 8 * words + threadIdx.x; i < R; i += blockDim.x) {
                if (active[i] == kRowListed) list[atomicAdd(&list_size, 1u)] = static_cast<uint32_t>(i);
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            const bool whole = flag == kFlagWholeMatrix;
            const uint64_t n = whole ? R : list_size;
            if (n < 2) {
                a.mat_list[m] = 0;
            } else {
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
        }
    }
}

// work item = matrix with a list
template <typename Arrays>
__global__ __launch_bounds__(256) void collapseRunsKernel(const ReplayArgs<Arrays> a) {
    const uint32_t num_items = a.replay_list[a.num_matrices];
    for (uint32_t item = blockIdx.x; item < num_items; item += gridDim.x) runsOfMatrix(a, a.replay_list[item]);
}

// behind a walk-only collapseRunsKernel: the search's sums ((matrix with a list, slice of its cells) per work item), ...
__global__ __launch_bounds__(256) void collapseAdjustSumsKernel(const ReplayArgs<MatrixArrays> a) {
    const uint32_t num_items = a.replay_list[a.num_matrices] * kAdjustSlices;
    for (uint32_t item = blockIdx.x; item < num_items; item += gridDim.x) {
        const uint32_t m = a.replay_list[item / kAdjustSlices];
        if (a.rewritten_count[m] == 0) continue;  // (uniform)
        __syncthreads();
        adjustSearchSums(a, m, item % kAdjustSlices);
    }
}

// ... and the values of the heads over their runs
__global__ __launch_bounds__(256) void collapseRewriteKernel(const ReplayArgs<MatrixArrays> a) {
    __shared__ uint32_t demote;
    const uint32_t num_items = a.replay_list[a.num_matrices];
    for (uint32_t item = blockIdx.x; item < num_items; item += gridDim.x) {
        const uint32_t m = a.replay_list[item];
        const uint32_t encoded = a.mat_list[m];
        __syncthreads();
        if (threadIdx.x == 0) demote = 0;
        __syncthreads();
        finishRuns(a, m, encoded & kListSizeMask, a.order + 2 * a.g.rowOffset(m), a.head_of + a.g.rowOffset(m), &demote, (encoded & kWholeMatrixBit) != 0);
    }
}

}  // namespace

namespace {

// ---- stage 0: the rows of every matrix in the order of their sort keys — without a library sort ---------------------------
// What the window scans need: the rows of a matrix ordered by (projection[, hash of the cells]), rows of equal keys in row
// order.  The keys carry the matrix above the projection and the rows of a matrix are contiguous, so the order is one
// segment per matrix, and nearly every segment fits a few KB of LDS: a workgroup loads up to 1 024 rows as 64-bit words
// [projection 24 | hash H | row 20 | largest value 20 - H], sorts them with a bitonic network and — the usual case, the whole
// matrix in one go — writes the sorted keys and rows, every row's sorted position, "equal to its predecessor" and the
// stretch starts itself.  A larger matrix lists its chunks of 2 048 rows; they are sorted the same way into a buffer and
// merged pairwise, round by round (merge path: every thread finds its eight output words by a binary search on its
// diagonal), ceil(log2(chunks)) rounds for the largest matrix there can be; a last kernel writes what the one-chunk matrices
// wrote in the sort.  (Rounds 2-3: hipcub::DeviceRadixSort over all rows of all matrices — five passes of three launches
// each, 0.31 ms of a lane's 1.2 ms of collapse standing alone — and DeviceSegmentedRadixSort for the EM problems, 0.5 ms in
// one kernel behind a host wait for its partition sizes, plus a kernel for the keys and one for "equal to its predecessor"
// that walked every row slot.  A first version of this one sorted chunks of 8 192 rows on 512 threads whatever the matrix —
// 0.28 ms for the 5 000 matrices of a configs[2] batch, two workgroups per CU — and ranked every element of a large matrix in
// every other chunk of it: chunks squared, 0.17-0.64 ms.)
constexpr uint32_t kSmallSortRows = 1024;
constexpr uint32_t kSmallSortThreads = 128;
constexpr uint32_t kSortChunkRows = 2048;
constexpr uint32_t kSortChunkThreads = 256;
constexpr uint32_t kMergeTile = 1024;                       // output words of a merge round per work item
constexpr uint32_t kMergeThreads = 128;
constexpr uint32_t kSortRowBits = 20;                       // rows of a matrix the packed word can tell apart
constexpr uint64_t kSortMaxSegmentRows = (1ull << kSortRowBits) - 1;

template <int HASH>
__device__ __forceinline__ uint64_t packSortWord(const uint64_t key, const uint32_t row) {
    constexpr int low = kCollapseLargestBits - HASH;  // bits of the key that are not sorted on
    const uint64_t sorted = (key & ((1ull << kCollapseMatrixShift) - 1)) >> low;
    return (sorted << (kSortRowBits + low)) | (static_cast<uint64_t>(row) << low) | (key & ((1ull << low) - 1));
}
template <int HASH>
__device__ __forceinline__ uint32_t sortWordRow(const uint64_t word) { return static_cast<uint32_t>(word >> (kCollapseLargestBits - HASH)) & ((1u << kSortRowBits) - 1); }
template <int HASH>
__device__ __forceinline__ uint64_t sortWordKey(const uint64_t word, const uint32_t matrix) {
    constexpr int low = kCollapseLargestBits - HASH;
    return (static_cast<uint64_t>(matrix) << kCollapseMatrixShift) | ((word >> (kSortRowBits + low)) << low) | (word & ((1ull << low) - 1));
}
template <int HASH>
__device__ __forceinline__ bool sortWordsSameSortedPart(const uint64_t x, const uint64_t y) {
    constexpr int shift = kSortRowBits + kCollapseLargestBits - HASH;
    return HASH ? (x >> shift) == (y >> shift) : ((x >> shift) == (y >> shift) && ((x ^ y) & ((1ull << (kCollapseLargestBits - HASH)) - 1)) == 0);
}

// a column's share of the hash of a row's cells (summed: the entries of a row come in no particular order; an empty cell adds nothing)
__device__ __forceinline__ uint64_t cellHashTerm(const uint32_t column, const double value) {
    const uint64_t cell = static_cast<uint64_t>(cellOf(value));
    uint64_t x = (cell + 0x632BE59BD9B4E019ull * (column + 1)) * 0xD6E8FEB86659FD93ull;
    x ^= x >> 32;
    return cell ? x * 0xC2B2AE3D27D4EB4Full : 0;
}

// the key of row i of matrix m: group matrices carry it (written by the build kernels), the rows of an EM problem get it here
// (projection of the normalised row, its largest value, a hash of its cells) together with their zero pattern
__device__ __forceinline__ uint64_t sortKeyOfRow(const MatrixArrays & g, const uint32_t m, const uint64_t r0, const uint32_t i, const uint64_t * key_in,
                                                 uint64_t * pattern_out) {
    (void) g; (void) m; (void) pattern_out;
    return key_in[r0 + i];
}
__device__ __forceinline__ uint64_t sortKeyOfRow(const CsrArrays & g, const uint32_t p, const uint64_t r0, const uint32_t i, const uint64_t * key_in,
                                                 uint64_t * pattern_out) {
    (void) key_in;
    const CsrView mv = g.view(p);
    const double noise = mv.noise[i];
    double projection = collapseWeight(mv.G) * noise, largest = 0.0;
    uint64_t cells = cellHashTerm(mv.G, noise), pattern = 0;
    for (uint32_t e = mv.off[i]; e < mv.off[i + 1]; ++e) {
        const uint32_t c = mv.col[e];
        const double v = mv.val[e];
        projection = fma(collapseWeight(c), v, projection);
        largest = fmax(largest, v);
        cells += cellHashTerm(c, v);
        if (c < 64 && v != 0.0) pattern |= 1ull << c;
    }
    pattern_out[r0 + i] = pattern;
    // the top of the low field: a hash of the row's cells, sorted on (rows of equal cells end up next to each other)
    constexpr int low_bits = kCollapseLargestBits - kCsrCellHashBits;
    uint64_t k = collapseSortKey(p, projection, largest);
    k = (k & ~((1ull << kCollapseLargestBits) - 1)) | (((cells * 0x9E3779B97F4A7C15ull) >> (64 - kCsrCellHashBits)) << low_bits) | (k & ((1ull << low_bits) - 1));
    return k;
}

template <typename Arrays>
struct SegmentSortArgs {
    Arrays g;
    uint32_t num_matrices;               // group matrices: their number; EM problems: the bound of their number
    const uint32_t * num_matrices_dev;   // EM problems on a device-built list: their number
    // EM problems: the work items of the fill (em_sparse.hip: segment_rows row slots of a problem) — collapseUnusedSlotsKernel
    uint32_t num_items_bound;
    const uint32_t * num_items_dev;
    const uint64_t * seg_first;
    const uint32_t * item_problem;
    uint32_t segment_rows;
    uint64_t total_rows;
    const uint64_t * key_in;             // group matrices: the keys of the build
    uint64_t * pattern_out;              // EM problems: zero pattern of every row slot
    uint64_t * sort_key;                 // sorted (matrix, key, largest value), by position
    uint32_t * sort_row;
    uint32_t * position_of;              // by row: its sorted position
    uint8_t * same_prev;
    uint32_t * stretch_start;
    uint64_t * words_a, * words_b;       // [total rows] each: the sorted runs of the larger matrices, source and target of a merge round in turn
    uint32_t * chunks;                   // [2 x capacity] (matrix, chunk of kSortChunkRows rows) of the matrices beyond kSmallSortRows rows
    uint32_t * chunk_count;
    uint32_t chunk_capacity;
    uint32_t round;                      // of the merge
    uint32_t rounds;
};

template <typename Arrays>
__device__ __forceinline__ void writeSortedPosition(const SegmentSortArgs<Arrays> & a, const uint32_t m, const uint64_t r0, const uint64_t position, const uint64_t word,
                                                    const uint64_t word_before, const bool has_before) {
    constexpr int HASH = Arrays::kHashBits;
    const uint32_t row = sortWordRow<HASH>(word);
    uint8_t same = 0;
    if (has_before && sortWordsSameSortedPart<HASH>(word, word_before)) {
        const typename Arrays::View mv = viewOf(m, a.g);
        const uint32_t before = sortWordRow<HASH>(word_before);
        same = (HASH ? rowsSameCells(mv, row, before) : rowsIdentical(mv, row, before)) ? 1 : 0;
    }
    a.sort_key[position] = sortWordKey<HASH>(word, m);
    a.sort_row[position] = static_cast<uint32_t>(r0 + row);
    a.position_of[r0 + row] = static_cast<uint32_t>(position);
    a.same_prev[position] = same;
    a.stretch_start[position] = same ? 0u : static_cast<uint32_t>(position);
}

// the rows [c0, c0 + n) of matrix m, sorted in `words` (LDS, `padded` >= n a power of two) by the calling workgroup
template <typename Arrays>
__device__ __forceinline__ void sortChunkInLds(const SegmentSortArgs<Arrays> & a, const uint32_t m, const uint64_t r0, const uint64_t c0, const uint32_t n,
                                               const uint32_t padded, uint64_t * words) {
    constexpr int HASH = Arrays::kHashBits;
    for (uint32_t i = threadIdx.x; i < padded; i += blockDim.x) {
        words[i] = i < n ? packSortWord<HASH>(sortKeyOfRow(a.g, m, r0, static_cast<uint32_t>(c0 + i), a.key_in, a.pattern_out), static_cast<uint32_t>(c0 + i)) : ~0ull;
    }
    __syncthreads();
    for (uint32_t k = 2; k <= padded; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = threadIdx.x; t < (padded >> 1); t += blockDim.x) {
                const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1));  // bit j of i is clear
                const uint32_t l = i | j;
                const uint64_t x = words[i], y = words[l];
                if (((i & k) == 0) == (x > y)) {
                    words[i] = y;
                    words[l] = x;
                }
            }
            __syncthreads();
        }
    }
}

// Matrices of up to kSmallSortRows rows — nearly all of a batch: sixteen workgroups of two waves per CU, each with its matrix
// in 8 KB of LDS, drawing matrices with a grid-stride loop.  A larger matrix lists its chunks for the kernel below.
template <typename Arrays>
__global__ __launch_bounds__(kSmallSortThreads) void collapseSmallSortKernel(const SegmentSortArgs<Arrays> a) {
    __shared__ uint64_t words[kSmallSortRows];
    const uint32_t M = a.num_matrices_dev ? min(*a.num_matrices_dev, a.num_matrices) : a.num_matrices;
    for (uint32_t m = blockIdx.x; m < M; m += gridDim.x) {
        const uint64_t R = a.g.numRows(m);
        if (R == 0) continue;
        if (R > kSmallSortRows) {
            if (threadIdx.x == 0) {
                const uint32_t chunks = static_cast<uint32_t>((R + kSortChunkRows - 1) / kSortChunkRows);
                const uint32_t first = atomicAdd(a.chunk_count, chunks);
                for (uint32_t c = 0; c < chunks && first + c < a.chunk_capacity; ++c) {
                    a.chunks[2 * static_cast<uint64_t>(first + c)] = m;
                    a.chunks[2 * static_cast<uint64_t>(first + c) + 1] = c;
                }
            }
            continue;
        }
        const uint64_t r0 = a.g.rowOffset(m);
        const uint32_t n = static_cast<uint32_t>(R);
        uint32_t padded = 2;
        while (padded < n) padded <<= 1;
        __syncthreads();  // (the matrix before is done with the LDS)
        sortChunkInLds(a, m, r0, 0, n, padded, words);
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) writeSortedPosition(a, m, r0, r0 + i, words[i], i ? words[i - 1] : 0ull, i != 0);
    }
}

// the chunks of the larger matrices: sorted runs of kSortChunkRows words in words_a
template <typename Arrays>
__global__ __launch_bounds__(kSortChunkThreads) void collapseChunkSortKernel(const SegmentSortArgs<Arrays> a) {
    __shared__ uint64_t words[kSortChunkRows];
    const uint32_t count = min(*a.chunk_count, a.chunk_capacity);
    for (uint32_t item = blockIdx.x; item < count; item += gridDim.x) {
        const uint32_t m = a.chunks[2 * static_cast<uint64_t>(item)], chunk = a.chunks[2 * static_cast<uint64_t>(item) + 1];
        const uint64_t R = a.g.numRows(m), r0 = a.g.rowOffset(m);
        const uint64_t c0 = static_cast<uint64_t>(chunk) * kSortChunkRows;
        const uint32_t n = static_cast<uint32_t>(min(static_cast<uint64_t>(kSortChunkRows), R - c0));
        uint32_t padded = 64;
        while (padded < n) padded <<= 1;
        __syncthreads();  // (the chunk before is done with the LDS)
        sortChunkInLds(a, m, r0, c0, n, padded, words);
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) a.words_a[r0 + c0 + i] = words[i];
    }
}

// One round of the merge: the sorted runs of length kSortChunkRows << round of every larger matrix, two by two.  Work item =
// kMergeTile output words of a matrix (two per chunk of the list); a thread finds where its eight outputs begin in the two
// runs by a binary search along its diagonal (merge path; the words are distinct: they carry the row) and merges them out.
template <typename Arrays>
__global__ __launch_bounds__(kMergeThreads) void collapseMergeRoundKernel(const SegmentSortArgs<Arrays> a) {
    constexpr uint32_t kPerThread = kMergeTile / kMergeThreads;
    constexpr uint32_t kTilesPerChunk = kSortChunkRows / kMergeTile;
    const uint32_t count = min(*a.chunk_count, a.chunk_capacity) * kTilesPerChunk;
    const uint64_t * __restrict__ source = (a.round & 1) ? a.words_b : a.words_a;
    uint64_t * __restrict__ target = (a.round & 1) ? a.words_a : a.words_b;
    const uint64_t run = static_cast<uint64_t>(kSortChunkRows) << a.round;
    for (uint32_t item = blockIdx.x; item < count; item += gridDim.x) {
        const uint32_t m = a.chunks[2 * static_cast<uint64_t>(item / kTilesPerChunk)];
        const uint64_t R = a.g.numRows(m), r0 = a.g.rowOffset(m);
        const uint64_t out0 = static_cast<uint64_t>(a.chunks[2 * static_cast<uint64_t>(item / kTilesPerChunk) + 1]) * kSortChunkRows + (item % kTilesPerChunk) * kMergeTile;
        if (out0 >= R) continue;
        const uint64_t pair0 = out0 / (2 * run) * (2 * run);
        const uint64_t a1 = min(R, pair0 + run), b1 = min(R, pair0 + 2 * run);
        const uint64_t * A = source + r0 + pair0, * B = source + r0 + a1;
        const uint64_t nA = a1 - pair0, nB = b1 - a1;
        const uint64_t d = (out0 - pair0) + static_cast<uint64_t>(threadIdx.x) * kPerThread;  // this thread's diagonal
        if (d >= nA + nB) continue;
        // i = words of A among the first d of the merged sequence: the smallest i with A[i] > B[d - i - 1]
        uint64_t lo = d > nB ? d - nB : 0, hi = min(d, nA);
        while (lo < hi) {
            const uint64_t mid = (lo + hi) >> 1;
            if (A[mid] < B[d - mid - 1]) lo = mid + 1;
            else hi = mid;
        }
        uint64_t i = lo, j = d - lo;
        uint64_t * out = target + r0 + pair0 + d;
        const uint32_t outputs = static_cast<uint32_t>(min(static_cast<uint64_t>(kPerThread), nA + nB - d));
        uint64_t x = i < nA ? A[i] : ~0ull, y = j < nB ? B[j] : ~0ull;
        for (uint32_t k = 0; k < outputs; ++k) {
            if (x < y) {
                out[k] = x;
                ++i;
                x = i < nA ? A[i] : ~0ull;
            } else {
                out[k] = y;
                ++j;
                y = j < nB ? B[j] : ~0ull;
            }
        }
    }
}

// the larger matrices' rows at their sorted positions (what the small kernel writes itself)
template <typename Arrays>
__global__ __launch_bounds__(256) void collapseMergedPositionsKernel(const SegmentSortArgs<Arrays> a) {
    const uint32_t count = min(*a.chunk_count, a.chunk_capacity);
    const uint64_t * __restrict__ sorted = (a.rounds & 1) ? a.words_b : a.words_a;
    for (uint32_t item = blockIdx.x; item < count; item += gridDim.x) {
        const uint32_t m = a.chunks[2 * static_cast<uint64_t>(item)], chunk = a.chunks[2 * static_cast<uint64_t>(item) + 1];
        const uint64_t R = a.g.numRows(m), r0 = a.g.rowOffset(m);
        const uint64_t c0 = static_cast<uint64_t>(chunk) * kSortChunkRows;
        const uint32_t n = static_cast<uint32_t>(min(static_cast<uint64_t>(kSortChunkRows), R - c0));
        for (uint32_t e = threadIdx.x; e < n; e += blockDim.x) {
            const uint64_t i = c0 + e;
            writeSortedPosition(a, m, r0, r0 + i, sorted[r0 + i], i ? sorted[r0 + i - 1] : 0ull, i != 0);
        }
    }
}

// EM problems: the row slots no problem's rows fill (the storage is laid out by a bound) get keys that sort behind every row
// and are close to nothing.  A workgroup per work item of the fill (em_sparse.hip: segment_rows slots of a problem): the
// unused slots of its own range, the last item of a problem up to the next problem's first slot; everybody shares the
// slots behind the last problem's items.
template <typename Arrays>
__global__ __launch_bounds__(256) void collapseUnusedSlotsKernel(const SegmentSortArgs<Arrays> a) {
    const uint32_t M = a.num_matrices_dev ? min(*a.num_matrices_dev, a.num_matrices) : a.num_matrices;
    const uint32_t items = a.num_items_dev ? min(*a.num_items_dev, a.num_items_bound) : a.num_items_bound;
    const uint64_t unused = collapseSortKey(a.num_matrices, 0.0, 0.0);
    auto fill = [&](const uint64_t r) {
        a.sort_key[r] = unused;
        a.sort_row[r] = static_cast<uint32_t>(r);
        a.position_of[r] = static_cast<uint32_t>(r);
        a.same_prev[r] = 0;
        a.stretch_start[r] = static_cast<uint32_t>(r);
        a.pattern_out[r] = 0;
    };
    for (uint32_t item = blockIdx.x; item < items; item += gridDim.x) {
        const uint32_t m = a.item_problem[item];
        if (m >= M) continue;
        const uint64_t s = item - a.seg_first[m];
        const bool last = item + 1 == a.seg_first[m + 1];
        const uint64_t r0 = a.g.rowOffset(m), R = a.g.numRows(m);
        const uint64_t lo = max(R, s * a.segment_rows);
        const uint64_t own = static_cast<uint64_t>(a.seg_first[m + 1] - a.seg_first[m]) * a.segment_rows;
        const uint64_t bound = min((m + 1 < M) ? a.g.rowOffset(m + 1) - r0 : own, a.total_rows - r0);
        const uint64_t hi = last ? bound : min(bound, (s + 1) * a.segment_rows);
        for (uint64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) fill(r0 + i);
    }
    const uint64_t tail = M == 0 ? 0 : min(a.total_rows, a.g.rowOffset(M - 1) + static_cast<uint64_t>(a.seg_first[M] - a.seg_first[M - 1]) * a.segment_rows);
    for (uint64_t r = tail + blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x; r < a.total_rows; r += static_cast<uint64_t>(gridDim.x) * blockDim.x) fill(r);
}

}  // namespace

namespace {

inline void launchAdjustAndRewrite(const ReplayArgs<MatrixArrays> & r, const uint32_t grid, hipStream_t on) {
    collapseAdjustSumsKernel<<<dim3(grid * 4), dim3(256), 0, on>>>(r);
    ReplayArgs<MatrixArrays> rewrite = r;
    rewrite.walk_only = false;
    collapseRewriteKernel<<<dim3(grid), dim3(256), 0, on>>>(rewrite);
}
inline void launchAdjustAndRewrite(const ReplayArgs<CsrArrays> &, const uint32_t, hipStream_t) {}
// the rows of the runs take their heads' values, behind a walk-only collapseRunsKernel (no search has read the matrices as built)
inline void launchRewrite(const ReplayArgs<MatrixArrays> & r, const uint32_t grid, hipStream_t on) {
    ReplayArgs<MatrixArrays> rewrite = r;
    rewrite.walk_only = false;
    collapseRewriteKernel<<<dim3(grid), dim3(256), 0, on>>>(rewrite);
}
inline void launchRewrite(const ReplayArgs<CsrArrays> &, const uint32_t, hipStream_t) {}

struct CollapseTemporaries {
    DeviceBuffer<uint32_t> rewritten_count;
    DeviceBuffer<uint64_t> key_out, pair_pattern;
    DeviceBuffer<uint32_t> row_out, order, head, pair_column, marked_list, pairs, between_items, row_items, list_index;
    DeviceBuffer<uint32_t> stretch;  // stretch starts by position, their running maximum, stretch ends, positions by row: total_rows each
    DeviceBuffer<uint64_t> pair_base;
    DeviceBuffer<uint8_t> pair_table;
    DeviceBuffer<double> pair_bound;
    DeviceBuffer<uint8_t> bytes;  // same_prev, close, barrier: total_rows each
    DeviceBuffer<unsigned char> sort_tmp;
    // EM problems only
    DeviceBuffer<uint64_t> csr_key, csr_pattern;
    DeviceBuffer<uint32_t> csr_row, csr_segments;
    // the hand-written segment sort (stage 0)
    DeviceBuffer<uint64_t> words_a, words_b;
    DeviceBuffer<uint32_t> chunks, chunk_count;
};

// the hand-written segment sort instead of the library's (null): what it needs beyond the stages' own arguments
struct SegmentSortPlan {
    const uint32_t * num_matrices_dev = nullptr;   // EM problems: their number on the device, and the work items of the fill
    uint32_t num_items_bound = 0;
    const uint32_t * num_items_dev = nullptr;
    const uint64_t * seg_first = nullptr;
    const uint32_t * item_problem = nullptr;
    uint32_t segment_rows = 0;
    uint64_t * pattern_out = nullptr;
    uint64_t max_segment_rows = 0;                 // a bound of the rows of the largest matrix (the rounds of the merge)
};

// The stages behind the keys, for group matrices and for EM problems alike: `key` / `row` hold, per row slot, the sort key
// (matrix, projection, largest value) and the slot itself; `info` receives [counters | zeroed words | active bytes | list sizes].
template <typename Arrays>
hipError_t queueCollapseStages(const Arrays & arrays, const uint32_t M, const uint64_t total_rows, const double precision, const uint64_t * key,
                               const uint32_t * row, const uint32_t * segment_begin, const uint32_t * segment_end, const bool segmented, DeviceBuffer<uint32_t> & info, double * rowmax, uint32_t * mat_fast,
                               uint32_t * mat_mid, CollapseTemporaries * tmp, hipStream_t st, const SegmentSortPlan * plan = nullptr, hipEvent_t sorted = nullptr,
                               std::function<hipError_t(hipStream_t, const rpvg_hip_groups::SearchSums *)> * held_back_runs = nullptr) {
    hipError_t e = hipSuccess;
    auto ok = [&](hipError_t r) { if (e == hipSuccess) e = r; return e == hipSuccess; };
    // zeroed words: the pair table's byte counter (8 bytes, first: aligned), matrix flags [M], replay list [M] + its
    // count, marked bits, then the counters of the marked list, the two pair lists, the between items, the row items
    const uint64_t mark_words = (total_rows + 31) / 32;
    const uint64_t num_words = 2 + 2 * static_cast<uint64_t>(M) + 1 + mark_words + 5;
    const uint32_t pair_capacity = static_cast<uint32_t>(std::min<uint64_t>(4 * total_rows + 4096, 0x20000000ull));
    if (tmp->key_out.count != total_rows) ok(tmp->key_out.alloc(total_rows));  // (the EM problems' caller has them already)
    if (tmp->row_out.count != total_rows) ok(tmp->row_out.alloc(total_rows));
    // one block, one memset: [info | the zeroed words | active bytes] + list sizes [M] (written by the list kernel)
    const uint64_t active_words = (total_rows + 3) / 4;
    ok(info.alloc(kInfoWords + num_words + active_words + M));
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
    ok(tmp->stretch.alloc(4 * total_rows));
    // The keys carry the matrix above the projection and the rows of a matrix are contiguous: ONE (stable) radix sort of
    // the whole array on the projection and as many matrix bits as there are matrices orders every matrix's rows by their
    // projection.  (The library path: RPVG_HIP_COLLAPSE_LIBRARY_SORT=1 for A/B, and matrices of 2^20 rows or more.  Round 2
    // sorted segment by segment, hipcub::DeviceSegmentedRadixSort: 0.33 ms per 1.6 M rows in 2 500
    // segments, the longest kernel of the collapse; RPVG_HIP_COLLAPSE_SEGMENTED_SORT=1 keeps it for A/B on the group matrices.)
    int matrix_bits = 1;
    while ((1u << matrix_bits) < M + 1) ++matrix_bits;  // (+ 1: the index unused row slots of the EM problems carry)
    const int begin_bit = kCollapseLargestBits - Arrays::kHashBits, end_bit = segmented ? kCollapseMatrixShift : kCollapseMatrixShift + matrix_bits;
    size_t sort_bytes = 0;
    auto sort = [&](void * scratch) {
        // (hipcub's segmented sort partitions the segments by size and reads the partition sizes back — the lane's thread waits
        // there for the fill; rocprim's unpartitioned configuration, one kernel and no wait, was 1 ms slower per lane all the same)
        return segmented ? hipcub::DeviceSegmentedRadixSort::SortPairs(scratch, sort_bytes, key, tmp->key_out.ptr, row,
                                                                       tmp->row_out.ptr, static_cast<int>(total_rows), static_cast<int>(M),
                                                                       segment_begin, segment_end, begin_bit, end_bit, st)
                         : hipcub::DeviceRadixSort::SortPairs(scratch, sort_bytes, key, tmp->key_out.ptr, row,
                                                              tmp->row_out.ptr, static_cast<int>(total_rows), begin_bit, end_bit, st);
    };
    if (e == hipSuccess && !plan) ok(sort(nullptr));
    uint32_t * stretch_start = tmp->stretch.ptr, * stretch_first = stretch_start + total_rows, * stretch_end = stretch_first + total_rows,
             * position_of = stretch_end + total_rows;
    size_t scan_bytes = 0;
    auto scan = [&](void * scratch) {
        return hipcub::DeviceScan::InclusiveScan(scratch, scan_bytes, stretch_start, stretch_first, hipcub::Max(), static_cast<int>(total_rows), st);
    };
    if (e == hipSuccess) ok(scan(nullptr));
    ok(tmp->sort_tmp.alloc(std::max(sort_bytes, scan_bytes)));
    // the hand-written segment sort: the chunk list of the matrices beyond the small sort, the two buffers of their merge
    uint32_t chunk_capacity = 0, merge_rounds = 0;
    if (plan) {
        chunk_capacity = static_cast<uint32_t>(std::min<uint64_t>(total_rows / kSortChunkRows + M + 1, 0x7fffffffull));
        const uint64_t max_chunks = (plan->max_segment_rows + kSortChunkRows - 1) / kSortChunkRows;
        while ((1ull << merge_rounds) < max_chunks) ++merge_rounds;
        ok(tmp->words_a.alloc(total_rows));
        ok(tmp->words_b.alloc(total_rows));
        ok(tmp->chunks.alloc(2 * static_cast<size_t>(chunk_capacity)));
        ok(tmp->chunk_count.alloc(1));
    }
    if (e != hipSuccess) return e;
    uint32_t * pair_bytes = info.ptr + kInfoWords, * mat_flag = pair_bytes + 2, * replay_list = mat_flag + M, * marked_bits = replay_list + M + 1,
             * marked_count = marked_bits + mark_words, * pair_counts = marked_count + 1, * between_count = pair_counts + 2,
             * row_item_count = between_count + 1, * mat_list = pair_bytes + num_words + active_words;
    uint8_t * same_prev = tmp->bytes.ptr, * close = same_prev + total_rows, * barrier = close + total_rows;
    uint8_t * active = reinterpret_cast<uint8_t *>(pair_bytes + num_words);
    ok(zeroAsync(info.ptr, (kInfoWords + num_words + active_words) * sizeof(uint32_t), st));
    if (!plan) ok(sort(tmp->sort_tmp.ptr));
    PairScanArgs<Arrays> a;
    a.total_rows = total_rows;
    a.precision = precision;
    a.sort_key = tmp->key_out.ptr;
    a.sort_row = tmp->row_out.ptr;
    a.g = arrays;
    a.same_prev = same_prev;
    a.stretch_first = stretch_start;  // (the kernel in front of the scan writes the starts)
    a.stretch_end = stretch_end;
    a.position_of = position_of;
    a.marked_bits = marked_bits;
    a.marked_list = tmp->marked_list.ptr;
    a.marked_count = marked_count;
    a.pairs = tmp->pairs.ptr;
    a.pair_count = pair_counts;
    a.pair_capacity = pair_capacity;
    a.active = active;
    a.mat_flag = mat_flag;
    a.info = info.ptr;
    a.num_matrices = M;
    a.count_pairs = RPVG_EXPERIMENT_ENV("RPVG_HIP_EM_COLLAPSE_DEBUG") != nullptr;
    const uint32_t row_blocks = static_cast<uint32_t>((total_rows + 255) / 256);
    // The kernels over pairs, lists and matrices with a list draw their (few) items from device-side counts with grid-stride
    // loops: a workgroup per CU or two, not thousands that find nothing (a workgroup costs the dispatcher ~40 ns whatever it does:
    // 4 096 of them were most of a 60 us kernel).  RPVG_HIP_COLLAPSE_GRID overrides (A/B).
    static const uint32_t small_grid = RPVG_EXPERIMENT_ENV("RPVG_HIP_COLLAPSE_GRID") ? std::max(1, std::atoi(RPVG_EXPERIMENT_ENV("RPVG_HIP_COLLAPSE_GRID"))) : 256;
    if (plan) {
        ok(zeroAsync(tmp->chunk_count.ptr, sizeof(uint32_t), st));
        SegmentSortArgs<Arrays> c;
        c.g = arrays;
        c.num_matrices = M;
        c.num_matrices_dev = plan->num_matrices_dev;
        c.num_items_bound = plan->num_items_bound;
        c.num_items_dev = plan->num_items_dev;
        c.seg_first = plan->seg_first;
        c.item_problem = plan->item_problem;
        c.segment_rows = plan->segment_rows;
        c.total_rows = total_rows;
        c.key_in = key;
        c.pattern_out = plan->pattern_out;
        c.sort_key = tmp->key_out.ptr;
        c.sort_row = tmp->row_out.ptr;
        c.position_of = position_of;
        c.same_prev = same_prev;
        c.stretch_start = stretch_start;
        c.words_a = tmp->words_a.ptr;
        c.words_b = tmp->words_b.ptr;
        c.chunks = tmp->chunks.ptr;
        c.chunk_count = tmp->chunk_count.ptr;
        c.chunk_capacity = chunk_capacity;
        c.round = 0;
        c.rounds = merge_rounds;
        if (Arrays::kHashBits) {
            collapseUnusedSlotsKernel<Arrays><<<dim3(std::max<uint32_t>(1, std::min<uint32_t>(plan->num_items_bound, 8 * small_grid))), dim3(256), 0, st>>>(c);
        }
        collapseSmallSortKernel<Arrays><<<dim3(std::max<uint32_t>(1, std::min<uint32_t>(M, 16 * small_grid))), dim3(kSmallSortThreads), 0, st>>>(c);
        // (the matrices beyond the small sort — a few dozen per batch: the chunks of their list, log2(chunks of the largest) rounds
        // of merging, their positions; with no such matrix possible, nothing)
        if (plan->max_segment_rows > kSmallSortRows) {
            collapseChunkSortKernel<Arrays><<<dim3(4 * small_grid), dim3(kSortChunkThreads), 0, st>>>(c);
            for (uint32_t round = 0; round < merge_rounds; ++round) {
                c.round = round;
                collapseMergeRoundKernel<Arrays><<<dim3(8 * small_grid), dim3(kMergeThreads), 0, st>>>(c);
            }
            collapseMergedPositionsKernel<Arrays><<<dim3(4 * small_grid), dim3(256), 0, st>>>(c);
        }
    } else {
        collapseSamePrevKernel<Arrays><<<dim3(row_blocks), dim3(256), 0, st>>>(a);
    }
    if (sorted) ok(hipEventRecord(sorted, st));
    ok(scan(tmp->sort_tmp.ptr));
    a.stretch_first = stretch_first;
    collapseStretchEndKernel<Arrays><<<dim3(row_blocks), dim3(256), 0, st>>>(a);
    collapseForwardPairsKernel<Arrays><<<dim3(row_blocks), dim3(256), 0, st>>>(a);
    collapseMarkPairsKernel<Arrays><<<dim3(small_grid), dim3(64), 0, st>>>(a);
    a.pair_count = pair_counts + 1;  // the list is free again
    collapseAroundPairsKernel<Arrays><<<dim3(256), dim3(64), 0, st>>>(a);
    collapseActivePairsKernel<Arrays><<<dim3(small_grid), dim3(64), 0, st>>>(a);
    ReplayArgs<Arrays> r;
    r.num_matrices = M;
    r.precision = precision;
    r.g = arrays;
    r.mat_flag = mat_flag;
    r.active = active;
    r.sort_row = tmp->row_out.ptr;
    r.position_of = position_of;
    r.stretch_end = stretch_end;
    r.rowmax = rowmax;
    r.mat_fast = mat_fast;
    r.mat_mid = mat_mid;
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
    r.info = info.ptr;
    r.debug = RPVG_EXPERIMENT_ENV("RPVG_HIP_EM_COLLAPSE_DEBUG") != nullptr;
    r.no_lds_tables = RPVG_EXPERIMENT_ENV("RPVG_HIP_COLLAPSE_NO_LDS_TABLES") != nullptr;
    collapseListKernel<Arrays><<<dim3(std::min<uint32_t>(M, small_grid)), dim3(256), 0, st>>>(r);
    const uint32_t staged_grid = small_grid;
    collapsePairTableKernel<Arrays><<<dim3(staged_grid), dim3(64), 0, st>>>(r);
    collapseRankKernel<Arrays><<<dim3(staged_grid), dim3(256), 0, st>>>(r);
    collapseSortKernel<Arrays><<<dim3(std::min<uint32_t>(M, 32)), dim3(kSortThreads), 0, st>>>(r);
    collapseBetweenKernel<Arrays><<<dim3(2 * staged_grid), dim3(kBetweenThreads), 0, st>>>(r);
    r.part_pair = nullptr;
    r.part_marginal = nullptr;
    r.pair_part_off = nullptr;
    r.col_part_off = nullptr;
    r.chunk_rows = 0;
    r.rewritten_count = nullptr;
    r.walk_only = false;
    if (held_back_runs) {
        // The walk over the runs writes nothing a reader of the matrices sees (the heads of the list positions, the list of the
        // rows that will take their head's values): it runs here, on the collapse's stream, beside the search — one matrix of a
        // configs[2] batch walks for 0.8 ms, which stood behind the tile kernel on the search's stream (round 5) — and what is
        // held back is the adjustment of the search's sums and the rewrite of the rows.
        ok(tmp->rewritten_count.alloc(M));
        r.rewritten_count = tmp->rewritten_count.ptr;
        r.walk_only = true;
        collapseRunsKernel<Arrays><<<dim3(staged_grid), dim3(256), 0, st>>>(r);
        *held_back_runs = [=](hipStream_t on, const rpvg_hip_groups::SearchSums * sums) {
            ReplayArgs<Arrays> mine = r;
            if (sums) {  // the search has read the matrices as built: its sums, then the rows
                mine.part_pair = sums->part_pair;
                mine.part_marginal = sums->part_marginal;
                mine.pair_part_off = sums->pair_part_off;
                mine.col_part_off = sums->col_part_off;
                mine.chunk_rows = sums->chunk_rows;
                launchAdjustAndRewrite(mine, staged_grid, on);
            } else {
                launchRewrite(mine, staged_grid, on);
            }
            return hipGetLastError();
        };
    } else {
        collapseRunsKernel<Arrays><<<dim3(staged_grid), dim3(256), 0, st>>>(r);
    }
    ok(hipGetLastError());
    return e;
}

// sort key, slot and zero pattern of every row slot of the EM problems' storage: the slots of a problem's kept rows carry
// (problem, projection of the normalised row, its largest value); the others the index `num_problems`, which no problem
// has — they sort behind every row and are close to nothing
__global__ __launch_bounds__(256) void csrCollapseKeysKernel(const CsrArrays g, const uint32_t num_problems_bound, const uint32_t * __restrict__ num_problems_dev,
                                                           const uint32_t num_items_bound, const uint32_t * __restrict__ num_items_dev,
                                                           const uint64_t * __restrict__ seg_first, const uint32_t * __restrict__ item_problem,
                                                           const uint32_t segment_rows, const uint64_t total_rows, uint64_t * __restrict__ key,
                                                           uint32_t * __restrict__ row, uint64_t * __restrict__ pattern_out, uint32_t * __restrict__ segment_begin,
                                                           uint32_t * __restrict__ segment_end, uint64_t * __restrict__ sorted_key,
                                                           uint32_t * __restrict__ sorted_row) {
    // Work item = a problem's stretch of `segment_rows` row slots (the items of the fill, em_sparse.hip: a problem has at least
    // as many as its slots need); the last item of a problem also covers the unused slots up to the next problem's.
    // (segment_begin != NULL: the rows are sorted problem by problem, a segment each, and the unused slots — which no segment
    // covers — keep what is written here in the sorted arrays.)
    const uint32_t P = num_problems_dev ? *num_problems_dev : num_problems_bound;
    const uint32_t items = num_items_dev ? min(*num_items_dev, num_items_bound) : num_items_bound;
    const uint64_t unused = collapseSortKey(num_problems_bound, 0.0, 0.0);
    for (uint32_t item = blockIdx.x; item < items; item += gridDim.x) {
        const uint32_t p = item_problem[item];
        if (p >= P) continue;
        const uint64_t s = item - seg_first[p];
        const bool last = item + 1 == seg_first[p + 1];
        const CsrView mv = g.view(p);
        const uint64_t r0 = g.row_base[p];
        const uint64_t slots = (p + 1 < P ? g.row_base[p + 1] : r0 + mv.R) - r0;
        if (s == 0 && threadIdx.x == 0 && segment_begin) {
            segment_begin[p] = static_cast<uint32_t>(r0);
            segment_end[p] = static_cast<uint32_t>(r0 + mv.R);
        }
        const uint64_t lo = s * segment_rows, hi = last ? slots : min(slots, (s + 1) * segment_rows);
        for (uint64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
            uint64_t k = unused, pattern = 0;
            if (i < mv.R) {
                const double noise = mv.noise[i];
                double projection = collapseWeight(mv.G) * noise, largest = 0.0;
                uint64_t cells = cellHashTerm(mv.G, noise);
                for (uint32_t e = mv.off[i]; e < mv.off[i + 1]; ++e) {
                    const uint32_t c = mv.col[e];
                    const double v = mv.val[e];
                    projection = fma(collapseWeight(c), v, projection);
                    largest = fmax(largest, v);
                    cells += cellHashTerm(c, v);
                    if (c < 64 && v != 0.0) pattern |= 1ull << c;
                }
                // the top of the low field: a hash of the row's cells, sorted on (rows of equal cells end up next to each other)
                constexpr int low_bits = kCollapseLargestBits - kCsrCellHashBits;
                k = collapseSortKey(p, projection, largest);
                k = (k & ~((1ull << kCollapseLargestBits) - 1)) | (((cells * 0x9E3779B97F4A7C15ull) >> (64 - kCsrCellHashBits)) << low_bits) | (k & ((1ull << low_bits) - 1));
            }
            key[r0 + i] = k;
            row[r0 + i] = static_cast<uint32_t>(r0 + i);
            pattern_out[r0 + i] = pattern;
            if (segment_begin && i >= mv.R) {
                sorted_key[r0 + i] = unused;
                sorted_row[r0 + i] = static_cast<uint32_t>(r0 + i);
            }
        }
    }
    const uint64_t thread = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x, threads = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    if (segment_begin) {  // the segments of the bound's problems that do not exist
        for (uint64_t p = P + thread; p < num_problems_bound; p += threads) segment_begin[p] = segment_end[p] = 0;
    }
    // the tail behind the last problem (everything, if there is no problem)
    const uint64_t from = P == 0 ? 0 : g.row_base[P - 1] + g.kept_rows[P - 1];
    for (uint64_t r = from + thread; r < total_rows; r += threads) {
        key[r] = unused;
        row[r] = static_cast<uint32_t>(r);
        pattern_out[r] = 0;
        if (segment_begin) {
            sorted_key[r] = unused;
            sorted_row[r] = static_cast<uint32_t>(r);
        }
    }
}

}  // namespace

// RPVG_HIP_COLLAPSE_LIBRARY_SORT = 1 | matrices | problems: the library's radix sorts instead of the hand-written segment sort (A/B)
static bool librarySortFor(const char * what) {
    const char * env = RPVG_EXPERIMENT_ENV("RPVG_HIP_COLLAPSE_LIBRARY_SORT");
    return env != nullptr && (std::strcmp(env, "1") == 0 || std::strcmp(env, what) == 0);
}

// Queues the collapse of the matrices of `g` on `st` behind their build (rpvg_hip_groups_build).
hipError_t rpvg_hip_detail::queueRowCollapse(rpvg_hip_ctx * ctx, rpvg_hip_groups * g, const uint64_t total_rows, const double precision,
                                             hipStream_t st, const bool hold_back_runs) {
    (void) ctx;
    const uint32_t M = g->num_matrices;
    if (M == 0 || total_rows == 0) return hipSuccess;
    if (total_rows > 0x7fffffffull || M >= kCollapseMaxMatrices) return hipErrorInvalidValue;
    std::shared_ptr<CollapseTemporaries> tmp = std::make_shared<CollapseTemporaries>();
    g->build_temporaries.emplace_back(tmp);
    MatrixArrays arrays;
    arrays.mat_val_off = g->mat_val_off.ptr;
    arrays.mat_row_off = g->mat_row_off.ptr;
    arrays.mat_rows = g->mat_rows.ptr;
    arrays.mat_cols = g->mat_cols.ptr;
    arrays.values = g->values.ptr;
    arrays.row_noise = g->row_noise.ptr;
    arrays.row_count = g->row_count.ptr;
    arrays.zero_pattern = g->collapse_mask.ptr;
    static const bool segmented = RPVG_EXPERIMENT_ENV("RPVG_HIP_COLLAPSE_SEGMENTED_SORT") != nullptr;  // (A/B: slower on the group matrices)
    const bool library_sort = librarySortFor("matrices") || segmented;  // A/B knob (read per call: the tests take both ways)
    SegmentSortPlan plan;
    for (uint32_t m = 0; m < M; ++m) plan.max_segment_rows = std::max<uint64_t>(plan.max_segment_rows, g->h_num_rows[m]);
    const bool use_plan = !library_sort && plan.max_segment_rows <= kSortMaxSegmentRows;
    return queueCollapseStages(arrays, M, total_rows, precision, g->collapse_key.ptr, g->collapse_row.ptr, g->collapse_segment_off.ptr,
                               g->collapse_segment_off.ptr + 1, segmented && g->collapse_segment_off.ptr != nullptr, g->collapse_info,
                               g->rowmax.ptr, g->mat_fast.ptr, g->mat_mid.ptr, tmp.get(), st, use_plan ? &plan : nullptr, nullptr,
                               hold_back_runs ? &g->held_back_runs : nullptr);
}

// readCollapseProbabilityMatrix on the rows of every EM problem of a solve (src/path_abundance_estimator.cpp:266,668): queued
// on `st` behind the compaction of the problems' rows.  Afterwards work.problem_merged[p] != 0 marks the problems in which a
// run joined rows that were not equal up to rounding, merged_count holds their read counts after the merges (a merged
// row's count moved to its run head) and merged_problems[0] their number.
hipError_t rpvg_hip_detail::queueCsrCollapse(rpvg_hip_ctx * ctx, const CsrCollapseInput & in, const double precision, CsrCollapseWork & work, hipStream_t st,
                                             hipEvent_t sorted) {
    const uint32_t P = in.num_problems_bound;
    const uint64_t total_rows = in.rows_capacity;
    if (P == 0 || total_rows == 0) return hipSuccess;
    if (total_rows > 0x7fffffffull || P + 1 >= kCollapseMaxMatrices) return hipErrorInvalidValue;
    std::shared_ptr<CollapseTemporaries> tmp = std::make_shared<CollapseTemporaries>();
    work.temporaries = tmp;
    hipError_t e = hipSuccess;
    auto ok = [&](hipError_t r) { if (e == hipSuccess) e = r; return e == hipSuccess; };
    const bool library_sort = librarySortFor("problems");  // A/B knob (read per call: the tests take both ways)
    // The hand-written segment sort computes the keys itself; problems of 2^20 rows or more (by the bound) and the A/B knob take
    // the library sort behind csrCollapseKeysKernel.
    const bool use_plan = !library_sort && in.max_rows_bound <= kSortMaxSegmentRows && in.max_rows_bound > 0 && in.seg_first && in.item_problem;
    ok(tmp->csr_pattern.alloc(total_rows));
    if (!use_plan) {
        ok(tmp->csr_key.alloc(total_rows));
        ok(tmp->csr_row.alloc(total_rows));
    }
    ok(work.merged_count.alloc(total_rows));
    ok(work.problem_merged.alloc(P + 1));  // [P]: the number of merged problems
    if (e != hipSuccess) return e;
    ok(zeroAsync(work.problem_merged.ptr, (P + 1) * sizeof(uint32_t), st));
    CsrArrays arrays;
    arrays.row_base = in.row_base;
    arrays.ent_base = in.ent_base;
    arrays.kept_rows = in.kept_rows;
    arrays.col_off = in.col_off;
    arrays.prow_off = in.prow_off;
    arrays.prow_count = in.prow_count;
    arrays.prow_noise = in.prow_noise;
    arrays.pent_col = in.pent_col;
    arrays.pent_val = in.pent_val;
    arrays.zero_pattern = tmp->csr_pattern.ptr;
    arrays.merged_count = work.merged_count.ptr;
    arrays.problem_merged = work.problem_merged.ptr;
    arrays.merged_problems = work.problem_merged.ptr + P;
    if (use_plan) {
        SegmentSortPlan plan;
        plan.num_matrices_dev = in.num_problems_dev;
        plan.num_items_bound = in.num_items_bound;
        plan.num_items_dev = in.num_items_dev;
        plan.seg_first = in.seg_first;
        plan.item_problem = in.item_problem;
        plan.segment_rows = in.segment_rows;
        plan.pattern_out = tmp->csr_pattern.ptr;
        plan.max_segment_rows = in.max_rows_bound;
        return queueCollapseStages(arrays, P, total_rows, precision, nullptr, nullptr, nullptr, nullptr, false, work.info, nullptr, nullptr, nullptr, tmp.get(), st, &plan, sorted);
    }
    // The rows of a problem are sorted as a segment: a handful of launches against the global sort's seven passes of three
    // launches each (same box, configs[2] batch: 12.0-12.5 ms per step against 14.6).  RPVG_HIP_EM_COLLAPSE_GLOBAL_SORT=1: the
    // global sort (A/B).
    static const bool segmented = RPVG_EXPERIMENT_ENV("RPVG_HIP_EM_COLLAPSE_GLOBAL_SORT") == nullptr;
    if (segmented) {
        ok(tmp->csr_segments.alloc(2 * static_cast<size_t>(P)));
        ok(tmp->key_out.alloc(total_rows));
        ok(tmp->row_out.alloc(total_rows));
        if (e != hipSuccess) return e;
    }
    uint32_t * segment_begin = segmented ? tmp->csr_segments.ptr : nullptr, * segment_end = segmented ? segment_begin + P : nullptr;
    const uint32_t keys_grid = std::max<uint32_t>(1, std::min<uint32_t>(in.num_items_bound, static_cast<uint32_t>(ctx->props.multiProcessorCount) * 8));
    csrCollapseKeysKernel<<<dim3(keys_grid), dim3(256), 0, st>>>(arrays, P, in.num_problems_dev, in.num_items_bound, in.num_items_dev, in.seg_first, in.item_problem,
                                                                in.segment_rows, total_rows, tmp->csr_key.ptr, tmp->csr_row.ptr, tmp->csr_pattern.ptr, segment_begin,
                                                                segment_end, tmp->key_out.ptr, tmp->row_out.ptr);
    ok(hipGetLastError());
    if (e != hipSuccess) return e;
    return queueCollapseStages(arrays, P, total_rows, precision, tmp->csr_key.ptr, tmp->csr_row.ptr, segment_begin, segment_end, segmented, work.info, nullptr,
                               nullptr, nullptr, tmp.get(), st, nullptr, sorted);
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
