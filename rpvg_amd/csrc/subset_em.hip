// From the diploid search to the EM solutions of the retained path subsets without a host round trip (gfx950).
//
// NestedPathAbundanceEstimator::inferAbundancesCollapsedGroups (src/path_abundance_estimator.cpp:428-471) is, per cluster,
//   calculatePathGroupPosteriorsBounded      src/path_estimator.cpp:379-473          (bounded_search.hip)
//   selectPathSubsetIndices                  src/path_abundance_estimator.cpp:569-606   <- subsetSelectKernel
//   inferPathSubsetAbundance, per subset:    :625-671  collapsed_path_subset, constructPartialProbabilityMatrix,
//                                            addNoiseAndNormalizeProbabilityMatrix, EMAbundanceEstimator (em_sparse.hip)
//   the posterior-weighted merge             :702-749  (host: rpvg_amd/host/path_abundance_estimator.cpp)
// Round 2 brought the kept diplotypes to the host between the first two lines and the EM problem list back to the device
// between the second and the third: 1.2 ms of every lane's critical path with the GPU idle, 8 ms with four host threads.
// Here the search's kept pairs stay where the search left them:
//   subsetSelectKernel    one workgroup per matrix: the diplotypes with posterior >= min_hap_prob (:576), each expanded to the
//                         sorted list of the paths of its two haplotype columns (:583-593; a merge of two ascending lists,
//                         walked, never stored), identical lists merged by summing their posteriors in the order the
//                         reference adds them (:595-596), weights normalised by the sum over the selected diplotypes
//                         (:598-605), subsets below min_hap_prob dropped (:627-630), the rest ranked in lexicographic order
//                         of their lists (the order of the host classes' ordered map; the reference's is that of a hash map)
//   subsetOffsetsKernel   one workgroup: prefix sums over the matrices — subsets, list lengths, columns, and the rows and
//                         entries the EM problems may keep (a problem keeps at most those of its cluster) — against the
//                         capacities the host reserved; writes the number of problems the kernels behind it read
//   subsetExpandKernel    one workgroup per matrix: the subsets' path lists (with the homozygous paths twice), their distinct
//                         paths = the columns of their EM problems, cluster, weight and storage offsets
//   queueEmSolve          em_sparse.hip: compaction, size bins, work queues, persistent EM kernels — all sized on the device
// The host reads a 64-byte header while the EM runs (how many subsets, how long the lists) and queues the copies of
// exactly that behind the EM kernels: one synchronisation per call.

#include "common.hpp"

#include <cmath>
#include <memory>

using namespace rpvg_hip_detail;

namespace {

constexpr uint32_t kMaxSelected = 1024;  // selected diplotypes per matrix the select kernel holds in LDS (1 / min_hap_prob of the default)

// ascending walk over the union (with repetition) of the path lists of two columns
struct MergedList {
    const uint32_t * a;
    const uint32_t * b;
    uint32_t na, nb, i, j;
    __device__ __forceinline__ MergedList(const uint32_t * a_in, uint32_t na_in, const uint32_t * b_in, uint32_t nb_in)
        : a(a_in), b(b_in), na(na_in), nb(nb_in), i(0), j(0) {}
    __device__ __forceinline__ bool done() const { return i >= na && j >= nb; }
    __device__ __forceinline__ uint32_t next() {
        if (j >= nb || (i < na && a[i] <= b[j])) return a[i++];
        return b[j++];
    }
};

struct Columns {  // the columns of one matrix as path lists
    const uint64_t * group_path_off;
    const uint32_t * group_path;
    uint64_t first_column;
    __device__ __forceinline__ MergedList list(const uint32_t first, const uint32_t second) const {
        const uint64_t a0 = group_path_off[first_column + first], a1 = group_path_off[first_column + first + 1];
        const uint64_t b0 = group_path_off[first_column + second], b1 = group_path_off[first_column + second + 1];
        return MergedList(group_path + a0, static_cast<uint32_t>(a1 - a0), group_path + b0, static_cast<uint32_t>(b1 - b0));
    }
};

// lexicographic comparison of two merged lists (std::vector's operator<): -1, 0, 1
__device__ __forceinline__ int compareLists(MergedList x, MergedList y) {
    while (!x.done() && !y.done()) {
        const uint32_t px = x.next(), py = y.next();
        if (px != py) return px < py ? -1 : 1;
    }
    if (x.done() && y.done()) return 0;
    return x.done() ? -1 : 1;
}

struct MatrixTotals {  // what subsetOffsetsKernel adds up over the matrices
    unsigned long long subsets, list_length, columns, rows, entries, segments;
};

struct SubsetHeader {  // device -> host, one small copy
    unsigned long long subsets, list_length, columns, rows, entries, segments;  // totals (what the capacities must hold)
    unsigned long long overflow;    // != 0: something did not fit (num_problems is 0 then, nothing behind runs)
    uint32_t num_problems;
    uint32_t num_items;             // row segments of all problems (em_sparse.hip: the work items of the compaction)
    uint32_t build_bad, pad;             // validity flag of the matrices' build (they were built without a host synchronisation)
    unsigned long long select_overflow;  // matrices with more selected diplotypes than kMaxSelected
    unsigned long long log_evals, kept_pairs;  // the search's counters (statistics)
    uint32_t big_count, pad2;        // matrices with more kept pairs than the one-wave select kernel takes
};

// The results of a call as ONE block (device, then page-locked host memory): every array 64-byte aligned, in this order.
// (A D2H copy costs the stream ~40 us next to running kernels whatever its size: twelve arrays were 0.5 ms at the end of
// every lane.)
struct PackedLayout {
    size_t subset_off, weight, path_off, col_off, abundances, noise, total, path, col_path, iterations, kept_rows, kept_entries, bytes;
    // the posterior-weighted merge (subsetMergeKernel): per matrix the number of path group sets and the noise count, per set (the
    // sets of matrix m from slot path_off[subset_off[m]] on: a matrix has at most as many as its subsets list paths) its one or
    // two paths, its posterior and the abundance of either path; flags[0]: a transcript with more than two paths in one subset
    size_t set_count, cluster_noise, set_first, set_second, set_posterior, set_abund, merge_flags;
};
__host__ __device__ inline PackedLayout packedLayout(const uint64_t M, const uint64_t S, const uint64_t L, const uint64_t C) {
    PackedLayout p;
    size_t at = 0;
    auto place = [&](const size_t bytes) {
        const size_t here = at;
        at += (bytes + 63) & ~static_cast<size_t>(63);
        return here;
    };
    p.subset_off = place((M + 1) * 8);
    p.weight = place(S * 8);
    p.path_off = place((S + 1) * 8);
    p.col_off = place((S + 1) * 8);
    p.abundances = place(C * 8);
    p.noise = place(S * 8);
    p.total = place(S * 8);
    p.path = place(L * 4);
    p.col_path = place(C * 4);
    p.iterations = place(S * 4);
    p.kept_rows = place(S * 4);
    p.kept_entries = place(S * 4);
    p.set_count = place(M * 4);
    p.cluster_noise = place(M * 8);
    p.set_first = place(L * 4);
    p.set_second = place(L * 4);
    p.set_posterior = place(L * 8);
    p.set_abund = place(L * 16);
    p.merge_flags = place(64);
    p.bytes = at;
    return p;
}

struct SelectArgs {
    uint32_t num_matrices;
    const uint32_t * pair_count;      // [M] kept pairs of each matrix (the search's tail words)
    const uint64_t * pair_cap_off;    // [M+1]
    const uint32_t * pair_first;
    const uint32_t * pair_second;
    const double * pair_value;        // posteriors
    const uint64_t * group_off;       // [M+1]
    const uint64_t * group_path_off;
    const uint32_t * group_path;
    const uint32_t * cluster;         // [M]
    const uint64_t * mat_rows;        // [M]
    const uint64_t * cluster_row_off;
    const uint64_t * row_ent_off;
    const uint64_t * slot_off;        // [M+1] slots of each matrix: min(pairs possible, kMaxSelected)
    double min_hap_prob;
    uint32_t segment_rows;            // emFillSegmentRows()
    uint32_t * slot_first;
    uint32_t * slot_second;
    double * slot_weight;
    uint32_t * slot_length;
    uint32_t * slot_columns;
    MatrixTotals * totals;            // [M]
    SubsetHeader * header;            // zero-initialised
    uint32_t * big_matrices;          // [M] matrices the one-wave launch left to the wide one (count: header->big_count)
};

template <int BLOCK>
__device__ __forceinline__ uint32_t blockExclusiveScan(const uint32_t v, uint32_t & total, uint32_t * scratch) {
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

// ---- selectPathSubsetIndices (src/path_abundance_estimator.cpp:569-606) + the retained subsets in order ----------
// Two launches: one wavefront per matrix for the matrices the search left at most kSmallSelected pairs (nearly all: a bench
// batch keeps eight per matrix), and a few 256-thread workgroups that walk the list of the others (a matrix with a flat
// posterior keeps hundreds: one wave over global scratch spent 3 ms on one of them).  Both copy the path lists of the
// matrix's columns into LDS first when they fit — every comparison of two subsets walks four of them — and both order
// the retained subsets by a bitonic network over their indices (rank by counting is quadratic in list walks).
constexpr uint32_t kSmallSelected = 64;
constexpr uint32_t kSmallCachedPaths = 1024, kWideCachedPaths = 6144;  // column path lists cached in LDS (entries)

struct CachedColumns {  // the columns of one matrix as path lists, in LDS when they fit
    const uint64_t * group_path_off;  // global offsets (relative to all matrices)
    const uint32_t * paths;           // LDS copy (offsets relative to `first_offset`) or the global array
    uint64_t first_column, base;      // base: what to subtract from a global offset to index `paths`
    __device__ __forceinline__ MergedList list(const uint32_t first, const uint32_t second) const {
        const uint64_t a0 = group_path_off[first_column + first], a1 = group_path_off[first_column + first + 1];
        const uint64_t b0 = group_path_off[first_column + second], b1 = group_path_off[first_column + second + 1];
        return MergedList(paths + (a0 - base), static_cast<uint32_t>(a1 - a0), paths + (b0 - base), static_cast<uint32_t>(b1 - b0));
    }
};

template <uint32_t CAP, uint32_t CACHE, int BLOCK>
__device__ __forceinline__ void selectSubsetsOfMatrix(const SelectArgs & args, const uint32_t m) {
    __shared__ uint32_t sel[CAP];          // selected pair -> index among the matrix's kept pairs
    __shared__ unsigned long long hash[CAP];
    __shared__ uint32_t length[CAP];
    __shared__ uint32_t leader[CAP];       // first selected pair with the identical path list
    __shared__ double weight[CAP];
    __shared__ uint32_t order[CAP];        // retained subsets, then sorted by their path lists
    __shared__ uint32_t cached_paths[CACHE];
    __shared__ uint32_t scratch[BLOCK / 64];
    __shared__ double sum_posterior;
    __shared__ unsigned long long red[3];
    __shared__ uint32_t n_retained;
    const uint32_t n_pairs = args.pair_count[m];
    const uint64_t base = args.pair_cap_off[m];
    const uint32_t * first = args.pair_first + base;
    const uint32_t * second = args.pair_second + base;
    const double * value = args.pair_value + base;
    const uint32_t k = args.cluster[m];
    MatrixTotals * out_totals = args.totals + m;
    // the path lists of the matrix's columns
    const uint64_t col0 = args.group_off[m], col1 = args.group_off[m + 1];
    const uint64_t path0 = args.group_path_off[col0], path1 = args.group_path_off[col1];
    const bool cached = path1 - path0 <= CACHE;
    if (cached) {
        for (uint32_t i = threadIdx.x; i < path1 - path0; i += BLOCK) cached_paths[i] = args.group_path[path0 + i];
    }
    const CachedColumns columns{args.group_path_off, cached ? cached_paths : args.group_path, col0, cached ? path0 : 0};

    // the selected diplotypes, in the order the search kept them (:574-576)
    uint32_t n_sel = 0;
    for (uint32_t c = 0; c < n_pairs; c += BLOCK) {
        const uint32_t i = c + threadIdx.x;
        const uint32_t take = (i < n_pairs && value[i] >= args.min_hap_prob) ? 1u : 0u;
        uint32_t total;
        const uint32_t at = n_sel + blockExclusiveScan<BLOCK>(take, total, scratch);
        if (take && at < CAP) sel[at] = i;
        n_sel += total;
    }
    if (n_sel > CAP) {  // (a threshold below 1 / kMaxSelected: the caller takes the host-driven path)
        if (threadIdx.x == 0) {
            atomicAdd(&args.header->select_overflow, 1ull);
            *out_totals = MatrixTotals{0, 0, 0, 0, 0, 0};
        }
        return;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double sum = 0;  // in pair order, as the reference adds it up (:598)
        for (uint32_t s = 0; s < n_sel; ++s) sum += value[sel[s]];
        sum_posterior = sum;
        red[0] = red[1] = red[2] = 0;
        n_retained = 0;
    }
    // every selected diplotype's path list: length and a hash (FNV-1a over the merged walk)
    for (uint32_t s = threadIdx.x; s < n_sel; s += BLOCK) {
        MergedList list = columns.list(first[sel[s]], second[sel[s]]);
        unsigned long long h = 1469598103934665603ull;
        uint32_t n = 0;
        while (!list.done()) {
            h = (h ^ list.next()) * 1099511628211ull;
            ++n;
        }
        hash[s] = h;
        length[s] = n;
    }
    __syncthreads();
    // identical lists: the first of them leads (:595, emplace finds the existing key)
    for (uint32_t s = threadIdx.x; s < n_sel; s += BLOCK) {
        uint32_t lead = s;
        for (uint32_t t = 0; t < s; ++t) {
            if (hash[t] == hash[s] && length[t] == length[s] &&
                compareLists(columns.list(first[sel[t]], second[sel[t]]), columns.list(first[sel[s]], second[sel[s]])) == 0) {
                lead = t;
                break;
            }
        }
        leader[s] = lead;
    }
    __syncthreads();
    // weight of a subset: the posteriors of its diplotypes added in pair order (:596), over the sum of all selected
    // (:602-605); the retained ones (:627-630: subsets below the threshold are skipped), in any order for now
    for (uint32_t s = threadIdx.x; s < n_sel; s += BLOCK) {
        double w = 0;
        if (leader[s] == s) {
            for (uint32_t t = s; t < n_sel; ++t) {
                if (leader[t] == s) w += value[sel[t]];
            }
            w /= sum_posterior;
            if (w >= args.min_hap_prob) order[atomicAdd(&n_retained, 1u)] = s;
        }
        weight[s] = w;
    }
    __syncthreads();
    // ... then in lexicographic order of their path lists: a bitonic network over `order` (padding sorts last)
    const uint32_t n_ret = n_retained;
    uint32_t padded = 1;
    while (padded < n_ret) padded <<= 1;
    for (uint32_t i = n_ret + threadIdx.x; i < padded; i += BLOCK) order[i] = 0xffffffffu;
    __syncthreads();
    auto before = [&](const uint32_t x, const uint32_t y) {  // subset x sorts before subset y
        if (x == 0xffffffffu) return false;
        if (y == 0xffffffffu) return true;
        return compareLists(columns.list(first[sel[x]], second[sel[x]]), columns.list(first[sel[y]], second[sel[y]])) < 0;
    };
    for (uint32_t size = 2; size <= padded; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t i = threadIdx.x; i < padded; i += BLOCK) {
                const uint32_t partner = i ^ stride;
                if (partner > i) {
                    const bool ascending = (i & size) == 0;
                    const uint32_t x = order[i], y = order[partner];
                    if (before(y, x) == ascending) {
                        order[i] = y;
                        order[partner] = x;
                    }
                }
            }
            __syncthreads();
        }
    }
    // their slots
    const uint64_t slot0 = args.slot_off[m];
    unsigned long long my_subsets = 0, my_length = 0, my_columns = 0;
    for (uint32_t rank = threadIdx.x; rank < n_ret; rank += BLOCK) {
        const uint32_t s = order[rank];
        uint32_t distinct = 0, previous = 0xffffffffu;
        MergedList walk = columns.list(first[sel[s]], second[sel[s]]);
        while (!walk.done()) {
            const uint32_t path = walk.next();
            distinct += (path != previous) ? 1u : 0u;
            previous = path;
        }
        args.slot_first[slot0 + rank] = first[sel[s]];
        args.slot_second[slot0 + rank] = second[sel[s]];
        args.slot_weight[slot0 + rank] = weight[s];
        args.slot_length[slot0 + rank] = length[s];
        args.slot_columns[slot0 + rank] = distinct;
        my_subsets += 1;
        my_length += length[s];
        my_columns += distinct;
    }
    if (my_subsets) {
        atomicAdd(&red[0], my_subsets);
        atomicAdd(&red[1], my_length);
        atomicAdd(&red[2], my_columns);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint64_t r0 = args.cluster_row_off[k], r1 = args.cluster_row_off[k + 1];
        const unsigned long long rows = r1 - r0, entries = args.row_ent_off[r1] - args.row_ent_off[r0];
        *out_totals = MatrixTotals{red[0], red[1], red[2], red[0] * rows, red[0] * entries, red[0] * ((rows + args.segment_rows - 1) / args.segment_rows)};
    }
}

__global__ __launch_bounds__(64) void subsetSelectSmallKernel(const SelectArgs args) {
    const uint32_t m = blockIdx.x;
    if (m >= args.num_matrices) return;
    if (args.pair_count[m] > kSmallSelected) {  // left to the wide launch: a few workgroups that walk this list
        if (threadIdx.x == 0) args.big_matrices[atomicAdd(&args.header->big_count, 1u)] = m;
        return;
    }
    selectSubsetsOfMatrix<kSmallSelected, kSmallCachedPaths, 64>(args, m);
}

__global__ __launch_bounds__(256) void subsetSelectWideKernel(const SelectArgs args) {
    const uint32_t count = args.header->big_count;
    for (uint32_t item = blockIdx.x; item < count; item += gridDim.x) {
        selectSubsetsOfMatrix<kMaxSelected, kWideCachedPaths, 256>(args, args.big_matrices[item]);
        __syncthreads();  // the LDS is reused
    }
}

struct OffsetsArgs {
    const uint32_t * pair_count;      // [M] (the search's tail words)
    const unsigned long long * log_evals;
    const uint32_t * build_error;     // null: already checked
    uint32_t num_matrices;
    const MatrixTotals * totals;   // [M]
    MatrixTotals * bases;          // [M] exclusive prefix
    SubsetHeader * header;
    unsigned long long cap_subsets, cap_length, cap_columns, cap_rows, cap_entries, cap_items;
    uint64_t * seg_first;          // [cap_subsets + 1]
    uint64_t * path_off;           // [cap_subsets + 1]: the terminal entries are written here
    uint64_t * col_off;
};

__global__ __launch_bounds__(1024) void subsetOffsetsKernel(const OffsetsArgs args) {
    __shared__ MatrixTotals sums[1024];
    const uint32_t M = args.num_matrices;
    const uint32_t per = (M + 1023) / 1024;
    const uint32_t lo = min(M, threadIdx.x * per), hi = min(M, lo + per);
    MatrixTotals mine{0, 0, 0, 0, 0, 0};
    unsigned long long my_pairs = 0;
    for (uint32_t m = lo; m < hi; ++m) {
        my_pairs += args.pair_count[m];
        const MatrixTotals t = args.totals[m];
        mine.subsets += t.subsets;
        mine.list_length += t.list_length;
        mine.columns += t.columns;
        mine.rows += t.rows;
        mine.entries += t.entries;
        mine.segments += t.segments;
    }
    sums[threadIdx.x] = mine;
    if (my_pairs) atomicAdd(&args.header->kept_pairs, my_pairs);
    __syncthreads();
    for (uint32_t step = 1; step < 1024; step <<= 1) {
        MatrixTotals add{0, 0, 0, 0, 0, 0};
        if (threadIdx.x >= step) add = sums[threadIdx.x - step];
        __syncthreads();
        MatrixTotals & s = sums[threadIdx.x];
        s.subsets += add.subsets;
        s.list_length += add.list_length;
        s.columns += add.columns;
        s.rows += add.rows;
        s.entries += add.entries;
        s.segments += add.segments;
        __syncthreads();
    }
    MatrixTotals run = sums[threadIdx.x];
    run.subsets -= mine.subsets;
    run.list_length -= mine.list_length;
    run.columns -= mine.columns;
    run.rows -= mine.rows;
    run.entries -= mine.entries;
    run.segments -= mine.segments;
    for (uint32_t m = lo; m < hi; ++m) {
        args.bases[m] = run;
        const MatrixTotals t = args.totals[m];
        run.subsets += t.subsets;
        run.list_length += t.list_length;
        run.columns += t.columns;
        run.rows += t.rows;
        run.entries += t.entries;
        run.segments += t.segments;
    }
    if (threadIdx.x == 1023) {
        const MatrixTotals all = sums[1023];
        SubsetHeader * h = args.header;
        h->subsets = all.subsets;
        h->list_length = all.list_length;
        h->columns = all.columns;
        h->rows = all.rows;
        h->entries = all.entries;
        h->segments = all.segments;
        const bool fits = h->select_overflow == 0 && all.subsets <= args.cap_subsets && all.list_length <= args.cap_length &&
                          all.columns <= args.cap_columns && all.rows <= args.cap_rows && all.entries <= args.cap_entries &&
                          all.segments <= args.cap_items && all.subsets < 0xffffffffull && all.segments < 0xffffffffull;
        h->overflow = fits ? 0ull : 1ull;
        h->num_problems = fits ? static_cast<uint32_t>(all.subsets) : 0u;
        h->num_items = fits ? static_cast<uint32_t>(all.segments) : 0u;
        h->log_evals = *args.log_evals;
        h->build_bad = args.build_error ? *args.build_error : 0u;
        if (fits) {
            args.path_off[all.subsets] = all.list_length;
            args.col_off[all.subsets] = all.columns;
            args.seg_first[all.subsets] = all.segments;
        }
    }
}

struct ExpandArgs {
    uint32_t num_matrices;
    const SubsetHeader * header;
    const MatrixTotals * totals;
    const MatrixTotals * bases;
    const uint64_t * slot_off;
    const uint32_t * slot_first;
    const uint32_t * slot_second;
    const double * slot_weight;
    const uint32_t * slot_length;
    const uint32_t * slot_columns;
    const uint64_t * group_off;
    const uint64_t * group_path_off;
    const uint32_t * group_path;
    const uint32_t * cluster;
    const uint64_t * cluster_row_off;
    const uint64_t * row_ent_off;
    // per subset
    uint32_t * sub_cluster;
    double * sub_weight;
    uint64_t * path_off;
    uint32_t * path;
    uint64_t * col_off;
    uint32_t * col_path;
    uint64_t * row_base;
    uint64_t * ent_base;
    uint64_t * seg_first;
    uint32_t * item_problem;
    uint32_t segment_rows;
    uint64_t * subset_off;   // [M+1]
};

__global__ __launch_bounds__(256) void subsetExpandKernel(const ExpandArgs args) {
    constexpr int BLOCK = 256;
    __shared__ uint32_t scratch[BLOCK / 64];
    const uint32_t m = blockIdx.x;
    if (m >= args.num_matrices) return;
    const MatrixTotals base = args.bases[m];
    if (threadIdx.x == 0) {
        args.subset_off[m] = args.header->overflow ? 0 : base.subsets;
        if (m + 1 == args.num_matrices) args.subset_off[m + 1] = args.header->overflow ? 0 : args.header->subsets;
    }
    if (args.header->overflow) return;
    const uint32_t n = static_cast<uint32_t>(args.totals[m].subsets);
    const uint64_t slot0 = args.slot_off[m];
    const Columns columns{args.group_path_off, args.group_path, args.group_off[m]};
    const uint32_t k = args.cluster[m];
    const uint64_t r0 = args.cluster_row_off[k], r1 = args.cluster_row_off[k + 1];
    const uint64_t rows = r1 - r0, entries = args.row_ent_off[r1] - args.row_ent_off[r0];
    uint64_t length_run = base.list_length, columns_run = base.columns;
    for (uint32_t c = 0; c < n; c += BLOCK) {
        const uint32_t r = c + threadIdx.x;
        const uint32_t len = r < n ? args.slot_length[slot0 + r] : 0u, cols = r < n ? args.slot_columns[slot0 + r] : 0u;
        uint32_t len_total, cols_total;
        const uint32_t len_before = blockExclusiveScan<BLOCK>(len, len_total, scratch);
        const uint32_t cols_before = blockExclusiveScan<BLOCK>(cols, cols_total, scratch);
        if (r < n) {
            const uint64_t s = base.subsets + r;
            const uint64_t p0 = length_run + len_before, c0 = columns_run + cols_before;
            args.sub_cluster[s] = k;
            args.sub_weight[s] = args.slot_weight[slot0 + r];
            args.path_off[s] = p0;
            args.col_off[s] = c0;
            args.row_base[s] = base.rows + static_cast<uint64_t>(r) * rows;
            args.ent_base[s] = base.entries + static_cast<uint64_t>(r) * entries;
            const uint64_t segments = (rows + args.segment_rows - 1) / args.segment_rows, item0 = base.segments + static_cast<uint64_t>(r) * segments;
            args.seg_first[s] = item0;
            for (uint64_t item = item0; item < item0 + segments; ++item) args.item_problem[item] = static_cast<uint32_t>(s);
            MergedList list = columns.list(args.slot_first[slot0 + r], args.slot_second[slot0 + r]);
            uint32_t previous = 0xffffffffu;
            uint64_t pw = p0, cw = c0;
            while (!list.done()) {
                const uint32_t path = list.next();
                args.path[pw++] = path;
                // collapsed_path_subset (:637-656): the distinct paths, ascending = the columns of the subset's EM problem
                if (path != previous) args.col_path[cw++] = path;
                previous = path;
            }
        }
        length_run += len_total;
        columns_run += cols_total;
    }
}

struct PackArgs {
    const SubsetHeader * header;
    uint32_t num_matrices;
    const uint64_t * subset_off;
    const double * weight;
    const uint64_t * path_off;
    const uint64_t * col_off;
    const double * abundances;
    const double * noise;
    const double * total;
    const uint32_t * path;
    const uint32_t * col_path;
    const uint32_t * iterations;
    const uint32_t * kept_rows;
    const uint32_t * kept_entries;
    const uint32_t * merge_bad;  // null: no merge on the device
    unsigned char * packed;
};

__global__ __launch_bounds__(256) void packResultsKernel(const PackArgs args) {
    const SubsetHeader h = *args.header;
    if (h.overflow) return;
    const uint64_t M = args.num_matrices, S = h.subsets, L = h.list_length, C = h.columns;
    const PackedLayout lay = packedLayout(M, S, L, C);
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x, first = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    auto copy = [&](const size_t offset, const void * from, const size_t words) {  // 4-byte words
        uint32_t * to = reinterpret_cast<uint32_t *>(args.packed + offset);
        const uint32_t * src = static_cast<const uint32_t *>(from);
        for (size_t i = first; i < words; i += stride) to[i] = src[i];
    };
    copy(lay.subset_off, args.subset_off, (M + 1) * 2);
    copy(lay.weight, args.weight, S * 2);
    copy(lay.path_off, args.path_off, (S + 1) * 2);
    copy(lay.col_off, args.col_off, (S + 1) * 2);
    copy(lay.abundances, args.abundances, C * 2);
    copy(lay.noise, args.noise, S * 2);
    copy(lay.total, args.total, S * 2);
    copy(lay.path, args.path, L);
    copy(lay.col_path, args.col_path, C);
    copy(lay.iterations, args.iterations, S);
    copy(lay.kept_rows, args.kept_rows, S);
    copy(lay.kept_entries, args.kept_entries, S);
    if (args.merge_bad && first == 0) *reinterpret_cast<uint32_t *>(args.packed + lay.merge_flags) = *args.merge_bad;
}

// ---- the posterior-weighted merge of the subsets' EM solutions (src/path_abundance_estimator.cpp:702-749) ----------------
// Per cluster the reference walks its retained subsets in order and, for every transcript (PathInfo::group_id) that has paths in
// the subset, adds the subset's weight to the posterior of that transcript's path set and weight x abundance / multiplicity to
// the set's abundances; the sets come out in lexicographic order of their path lists (the ordered map of the host classes).
// One workgroup per cluster does the same: every position of every subset's path list looks for the other path of its
// transcript in the list (a diplotype's list holds at most two: its haplotypes' paths) and the first one emits a record
//   key = [ path + 1 : 27 | other path + 1 or 0 : 27 | rank of the subset in the cluster : 10 ]
// with the two weighted abundances; the records are sorted by key (bitonic network, in LDS up to 2 048 list entries per cluster,
// in device memory beyond) — which puts the sets in the reference's order and the records of a set in subset order — and one
// thread per set adds its records up one after the other: the additions of the host's merge, in its order, with
// separately rounded multiplies and adds (mulRounded / addRounded, common.hpp: no fused multiply-add, the host has none), so the results are bit-equal.
constexpr uint32_t kMergeLdsRecords = 2048;
constexpr unsigned long long kMergeNoRecord = ~0ull;
constexpr uint32_t kMergePathBits = 27, kMergeRankBits = 10;

struct MergeArgs {
    const SubsetHeader * header;
    uint32_t num_matrices;
    const uint64_t * subset_off;
    const double * weight;
    const uint64_t * path_off;
    const uint32_t * path;
    const uint64_t * col_off;
    const uint32_t * col_path;
    const double * abundances;
    const double * noise;
    const double * total;
    const uint32_t * cluster;           // [M]
    const uint64_t * cluster_path_off;  // of the batch
    const uint32_t * path_group_id;     // of the batch
    unsigned long long * sort_key;      // [2 x capacity of the lists] scratch of the clusters beyond LDS
    uint32_t * sort_pos;
    double * rec_a0;                    // [capacity of the lists]
    double * rec_a1;
    uint32_t * merge_bad;               // zero-initialised: set when a transcript has more than two paths in one subset
    unsigned char * packed;
};

__global__ __launch_bounds__(256) void subsetMergeKernel(const MergeArgs args) {
    constexpr int BLOCK = 256;
    __shared__ unsigned long long s_key[kMergeLdsRecords];
    __shared__ uint32_t s_pos[kMergeLdsRecords];
    __shared__ uint32_t s_scan[BLOCK / 64];
    const SubsetHeader h = *args.header;
    if (h.overflow) return;
    const uint32_t m = blockIdx.x, tid = threadIdx.x;
    const uint64_t M = args.num_matrices;
    if (m >= M) return;
    const PackedLayout lay = packedLayout(M, h.subsets, h.list_length, h.columns);
    uint32_t * set_count = reinterpret_cast<uint32_t *>(args.packed + lay.set_count);
    double * cluster_noise = reinterpret_cast<double *>(args.packed + lay.cluster_noise);
    uint32_t * set_first = reinterpret_cast<uint32_t *>(args.packed + lay.set_first);
    uint32_t * set_second = reinterpret_cast<uint32_t *>(args.packed + lay.set_second);
    double * set_posterior = reinterpret_cast<double *>(args.packed + lay.set_posterior);
    double * set_abund = reinterpret_cast<double *>(args.packed + lay.set_abund);
    const uint64_t s0 = args.subset_off[m];
    const uint32_t n = static_cast<uint32_t>(args.subset_off[m + 1] - s0);
    if (n == 0) {  // (no diplotype reached min_hap_prob: all reads are noise, src/path_abundance_estimator.cpp:749 — the host fills that in)
        if (tid == 0) {
            set_count[m] = 0;
            cluster_noise[m] = 0.0;
        }
        return;
    }
    const uint64_t base = args.path_off[s0];
    const uint32_t Lm = static_cast<uint32_t>(args.path_off[s0 + n] - base);
    const uint32_t * gid = args.path_group_id + args.cluster_path_off[args.cluster[m]];
    if (tid == BLOCK - 1) {  // the noise count: the subsets' noise counts by their weights, then the mass of the dropped diplotypes (:712,749)
        double noise = 0.0, sum_hap_prob = 0.0;
        for (uint32_t r = 0; r < n; ++r) {
            const double w = args.weight[s0 + r];
            sum_hap_prob = addRounded(sum_hap_prob, w);
            noise = addRounded(noise, mulRounded(args.noise[s0 + r], w));
        }
        cluster_noise[m] = addRounded(noise, mulRounded(addRounded(1.0, -sum_hap_prob), args.total[s0]));
    }
    uint32_t P2 = 2;
    while (P2 < Lm) P2 <<= 1;
    unsigned long long * keys = s_key;
    uint32_t * order = s_pos;
    if (P2 > kMergeLdsRecords) {
        keys = args.sort_key + 2 * base;
        order = args.sort_pos + 2 * base;
    }
    for (uint32_t pos = tid; pos < P2; pos += BLOCK) {
        unsigned long long key = kMergeNoRecord;
        if (pos < Lm) {
            uint32_t lo = 0, hi = n - 1;
            while (lo < hi) {
                const uint32_t mid = (lo + hi + 1) >> 1;
                if (args.path_off[s0 + mid] - base <= pos) lo = mid; else hi = mid - 1;
            }
            const uint32_t r = lo;
            const uint32_t l0 = static_cast<uint32_t>(args.path_off[s0 + r] - base), L = static_cast<uint32_t>(args.path_off[s0 + r + 1] - base) - l0;
            const uint32_t * list = args.path + base + l0;
            const uint32_t j = pos - l0;
            const uint32_t g = gid[list[j]];
            uint32_t members = 0, first = 0, second = 0;
            for (uint32_t t = 0; t < L; ++t) {
                if (gid[list[t]] == g) {
                    if (members == 0) first = t;
                    else if (members == 1) second = t;
                    ++members;
                }
            }
            if (members > 2) *args.merge_bad = 1;
            if (first == j && members <= 2) {
                const uint32_t p0 = list[first], p1 = (members == 2) ? list[second] : 0xffffffffu;
                const double w = args.weight[s0 + r];
                const uint64_t c0 = args.col_off[s0 + r];
                const uint32_t C = static_cast<uint32_t>(args.col_off[s0 + r + 1] - c0);
                const uint32_t * cols = args.col_path + c0;
                auto column = [&](const uint32_t p) {
                    uint32_t a = 0, b = C;
                    while (a < b) {
                        const uint32_t mid = (a + b) >> 1;
                        if (cols[mid] < p) a = mid + 1; else b = mid;
                    }
                    return a;
                };
                const double multiplicity = (members == 2 && p0 == p1) ? 2.0 : 1.0;
                args.rec_a0[base + pos] = mulRounded(args.abundances[c0 + column(p0)], w) / multiplicity;
                args.rec_a1[base + pos] = (members == 2) ? mulRounded(args.abundances[c0 + column(p1)], w) / multiplicity : 0.0;
                key = (static_cast<unsigned long long>(p0 + 1) << (kMergePathBits + kMergeRankBits)) |
                      (static_cast<unsigned long long>(members == 2 ? p1 + 1 : 0u) << kMergeRankBits) | r;
            }
        }
        keys[pos] = key;
        order[pos] = pos;
    }
    __syncthreads();
    for (uint32_t k2 = 2; k2 <= P2; k2 <<= 1) {
        for (uint32_t j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
            for (uint32_t i = tid; i < P2; i += BLOCK) {
                const uint32_t x = i ^ j2;
                if (x > i) {
                    const unsigned long long a = keys[i], b = keys[x];
                    if ((a > b) == ((i & k2) == 0)) {
                        keys[i] = b;
                        keys[x] = a;
                        const uint32_t oa = order[i];
                        order[i] = order[x];
                        order[x] = oa;
                    }
                }
            }
            __syncthreads();
        }
    }
    uint32_t Q = 0;
    for (uint32_t c0 = 0; c0 < Lm; c0 += BLOCK) {
        const uint32_t i = c0 + tid;
        const unsigned long long key = i < Lm ? keys[i] : kMergeNoRecord;
        const bool start = key != kMergeNoRecord && (i == 0 || (keys[i - 1] >> kMergeRankBits) != (key >> kMergeRankBits));
        uint32_t total;
        const uint32_t before = blockExclusiveSum<BLOCK>(start ? 1u : 0u, total, s_scan);
        if (start) {
            const unsigned long long set = key >> kMergeRankBits;
            double posterior = 0.0, a0 = 0.0, a1 = 0.0;
            for (uint32_t e = i; e < Lm && keys[e] != kMergeNoRecord && (keys[e] >> kMergeRankBits) == set; ++e) {
                const uint32_t r = static_cast<uint32_t>(keys[e] & ((1u << kMergeRankBits) - 1u)), pos = order[e];
                posterior = addRounded(posterior, args.weight[s0 + r]);
                a0 = addRounded(a0, args.rec_a0[base + pos]);
                a1 = addRounded(a1, args.rec_a1[base + pos]);
            }
            const uint64_t q = base + Q + before;
            const uint32_t second = static_cast<uint32_t>(set & ((1u << kMergePathBits) - 1u));
            set_first[q] = static_cast<uint32_t>(set >> kMergePathBits) - 1;
            set_second[q] = second ? second - 1 : 0xffffffffu;
            set_posterior[q] = posterior;
            set_abund[2 * q] = a0;
            set_abund[2 * q + 1] = a1;
        }
        Q += total;
    }
    if (tid == 0) set_count[m] = Q;
}

}  // namespace

// the result: everything in one page-locked block
struct rpvg_hip_subset_em {
    uint32_t num_matrices = 0;
    uint64_t num_subsets = 0;
    void * block = nullptr;
    rpvg_hip_subset_em_view view{};
    ~rpvg_hip_subset_em() {
        if (block) pinnedFree(block);
    }
};

// One attempt with the capacities the context's hints give.  *did_not_fit: the reserved capacity was too small — the hints now hold
// what this very call needs and a second attempt fits.
static int nestedSubsetEmAttempt(rpvg_hip_ctx * ctx, const rpvg_hip_batch * batch, const rpvg_hip_groups * groups,
                                 const uint32_t * column_counts, double min_rel_likelihood, double min_hap_prob,
                                 uint32_t max_em_its, double max_rel_em_conv, double collapse_precision,
                                 rpvg_hip_subset_em ** result_out, bool * did_not_fit, PairSearchWork & search, const bool second_attempt) {
    *did_not_fit = false;
    RPVG_REQUIRE(min_rel_likelihood > 0, "rpvg_hip_nested_subset_em: min_rel_likelihood must be positive");
    RPVG_REQUIRE(max_em_its > 0, "rpvg_hip_nested_subset_em: max_em_its must be positive");
    RPVG_REQUIRE(groups->batch == batch, "rpvg_hip_nested_subset_em: the matrices were built on another batch");
    const uint32_t M = groups->num_matrices;
    RPVG_REQUIRE(M == 0 || column_counts || groups->d_column_counts, "rpvg_hip_nested_subset_em: column_counts is NULL");
    std::unique_ptr<rpvg_hip_subset_em> res(new (std::nothrow) rpvg_hip_subset_em());
    if (!res) {
        setError("rpvg_hip_nested_subset_em: out of host memory");
        return RPVG_HIP_ERR_ALLOC;
    }
    res->num_matrices = M;

    // What the device path does not take (the caller runs the three calls it replaces): a subset threshold that lets a
    // matrix select more diplotypes than the select kernel holds, clusters whose EM vectors do not fit LDS, the A/B knob.
    const bool disabled = std::getenv("RPVG_HIP_NO_DEVICE_SUBSETS") != nullptr;  // (read per call: the tests take both ways)
    if (disabled || !(min_hap_prob * kMaxSelected >= 1.0)) {
        setError("rpvg_hip_nested_subset_em: not taken (min_hap_prob %g admits more than %u subsets per cluster, or RPVG_HIP_NO_DEVICE_SUBSETS)",
                 min_hap_prob, kMaxSelected);
        return RPVG_HIP_ERR_UNSUPPORTED;
    }
    std::unique_ptr<HostScope> scope(new HostScope("subset em: bounds"));
    uint64_t slots = 0, lane_rows = 0, lane_entries = 0, lane_paths = 0, max_cluster_work = 0;
    uint32_t max_paths = 0;
    std::vector<uint64_t> slot_off(M + 1, 0);
    for (uint32_t m = 0; m < M; ++m) {
        const uint64_t G = groups->h_num_cols[m];
        slot_off[m + 1] = slot_off[m] + std::min<uint64_t>(G * (G + 1) / 2, kMaxSelected);
        const uint32_t k = groups->h_cluster[m];
        lane_rows += batch->h_cluster_row_off[k + 1] - batch->h_cluster_row_off[k];
        lane_entries += batch->h_cluster_ent_off[k + 1] - batch->h_cluster_ent_off[k];
        lane_paths += groups->h_num_paths[m];
        max_paths = std::max(max_paths, groups->h_num_paths[m]);
        max_cluster_work = std::max<uint64_t>(max_cluster_work, (batch->h_cluster_row_off[k + 1] - batch->h_cluster_row_off[k]) +
                                                                    (batch->h_cluster_ent_off[k + 1] - batch->h_cluster_ent_off[k]));
    }
    slots = slot_off[M];
    // (em_sparse.hip, emLdsBytes: the streamed kernel of four wavefronts — abundances and an accumulator vector per wavefront)
    if (M > 0 && 8ull * (5ull * (static_cast<uint64_t>(max_paths) + 1) + 6) > 156 * 1024) {
        setError("rpvg_hip_nested_subset_em: not taken (a cluster with %u paths: EM vectors in global memory)", max_paths);
        return RPVG_HIP_ERR_UNSUPPORTED;
    }

    std::lock_guard<std::mutex> lock(ctx->mutex);
    RPVG_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    if (M == 0) {
        *result_out = res.release();
        return RPVG_HIP_OK;
    }
    // Capacities: what the last call on this context needed, with headroom; a first call plans with eight subsets' worth of
    // every cluster.  A call that does not fit reports what it needed (the next one fits) and is not taken.
    SubsetEmHints & hints = ctx->subset_hints;
    auto planned = [](const double per_unit, const unsigned long long units, const double headroom) {
        return static_cast<unsigned long long>(std::ceil(per_unit * headroom * static_cast<double>(units)));
    };
    unsigned long long cap_subsets = std::min<unsigned long long>(slots, hints.subsets_per_matrix > 0 ? planned(hints.subsets_per_matrix, M, 1.25) + 256 : std::max<unsigned long long>(8ull * M, 16384));
    unsigned long long cap_length = std::max<unsigned long long>(planned(hints.list_per_path, lane_paths, 2.0), std::max<unsigned long long>(8 * lane_paths, 1u << 20));
    unsigned long long cap_rows = hints.rows_per_row > 0 ? planned(hints.rows_per_row, lane_rows, 1.25) + 65536 : 8 * lane_rows;
    unsigned long long cap_entries = hints.entries_per_entry > 0 ? planned(hints.entries_per_entry, lane_entries, 1.25) + 65536 : 8 * lane_entries;
    if (hints.retry) {  // the second attempt of a call that did not fit: what it reported, exactly
        cap_subsets = std::min<unsigned long long>(slots, std::max(cap_subsets, hints.retry_subsets + 64));
        cap_length = std::max(cap_length, hints.retry_list_length + 64);
        cap_rows = std::max(cap_rows, hints.retry_rows + 64);
        cap_entries = std::max(cap_entries, hints.retry_entries + 64);
        hints.retry = false;
    }
    const unsigned long long cap_columns = cap_length;
    const unsigned long long cap_items = cap_rows / emFillSegmentRows() + cap_subsets;

    scope.reset(new HostScope("subset em: search + kernels queued"));
    // with the search's own small arrays (one copy, one memset): the slot offsets; the header and the EM work queues, zeroed
    const size_t header_room = 256;
    int rc = RPVG_HIP_OK;
    if (!second_attempt) {
        search.extra_u64 = slot_off.data();
        search.extra_u64_count = M + 1;
        search.extra_zero_bytes = header_room + emQueuesBytes();
        rc = queuePairSearch(ctx, groups, column_counts, min_rel_likelihood, search);
        if (rc != RPVG_HIP_OK) return rc;
    } else {
        // the search is the first attempt's: its pairs are on the device (a second search would meet the matrices with their
        // collapse's last stage done and leave sums that differ in their last bits — the estimates of a cluster must not depend on
        // what shared its batch); the header and the EM work queues start from zero again
        const hipError_t zeroed = zeroAsync(search.d_extra_zero.ptr, search.extra_zero_bytes, st);
        if (zeroed != hipSuccess) {
            setError("rpvg_hip_nested_subset_em: %s", hipGetErrorString(zeroed));
            return RPVG_HIP_ERR_RUNTIME;
        }
    }

    hipError_t e = hipSuccess;
    auto ok = [&](hipError_t r) { if (e == hipSuccess) e = r; return e == hipSuccess; };
    const uint64_t * d_slot_off = search.d_extra_u64.ptr;
    SubsetHeader * header = reinterpret_cast<SubsetHeader *>(search.d_extra_zero.ptr);
    static_assert(sizeof(SubsetHeader) <= 256, "room for the header");
    DeviceBuffer<uint32_t> d_slot_first, d_slot_second, d_slot_length, d_slot_columns;
    DeviceBuffer<double> d_slot_weight;
    DeviceBuffer<unsigned char> d_totals, d_bases, d_packed;
    ok(d_slot_first.alloc(slots));
    ok(d_slot_second.alloc(slots));
    ok(d_slot_length.alloc(slots));
    ok(d_slot_columns.alloc(slots));
    ok(d_slot_weight.alloc(slots));
    ok(d_totals.alloc(sizeof(MatrixTotals) * M));
    ok(d_bases.alloc(sizeof(MatrixTotals) * M));
    DeviceBuffer<uint32_t> d_big_matrices;
    ok(d_big_matrices.alloc(M));
    // per subset, and the lists
    DeviceBuffer<uint32_t> d_sub_cluster, d_path, d_col_path, d_iters, d_kept_rows, d_kept_entries;
    DeviceBuffer<double> d_sub_weight, d_abund, d_noise, d_total;
    DeviceBuffer<uint64_t> d_path_off, d_col_off, d_row_base, d_ent_base, d_subset_off, d_seg_first;
    DeviceBuffer<uint32_t> d_item_problem;
    ok(d_sub_cluster.alloc(cap_subsets));
    ok(d_sub_weight.alloc(cap_subsets));
    ok(d_path_off.alloc(cap_subsets + 1));
    ok(d_col_off.alloc(cap_subsets + 1));
    ok(d_row_base.alloc(cap_subsets));
    ok(d_ent_base.alloc(cap_subsets));
    ok(d_subset_off.alloc(M + 1));
    ok(d_seg_first.alloc(cap_subsets + 1));
    ok(d_item_problem.alloc(cap_items));
    ok(d_path.alloc(cap_length));
    ok(d_col_path.alloc(cap_columns));
    ok(d_abund.alloc(cap_columns));
    ok(d_noise.alloc(cap_subsets));
    ok(d_total.alloc(cap_subsets));
    ok(d_iters.alloc(cap_subsets));
    ok(d_kept_rows.alloc(cap_subsets));
    ok(d_kept_entries.alloc(cap_subsets));
    const size_t packed_capacity = packedLayout(M, cap_subsets, cap_length, cap_columns).bytes;
    ok(d_packed.alloc(packed_capacity));
    // the posterior-weighted merge on the device: when the batch carries the transcripts of its paths (PathInfo::group_id)
    const bool merge_here = batch->path_group_id.ptr != nullptr && max_paths < (1u << kMergePathBits) - 2;
    DeviceBuffer<unsigned long long> d_merge_key;
    DeviceBuffer<uint32_t> d_merge_pos;
    DeviceBuffer<double> d_merge_a0, d_merge_a1;
    if (merge_here) {
        ok(d_merge_key.alloc(2 * cap_length + 2));
        ok(d_merge_pos.alloc(2 * cap_length + 2));
        ok(d_merge_a0.alloc(cap_length));
        ok(d_merge_a1.alloc(cap_length));
    }
    void * pinned_header = nullptr;
    if (e == hipSuccess && pinnedAlloc(&pinned_header, sizeof(SubsetHeader)) != hipSuccess) e = hipErrorOutOfMemory;
    hipEvent_t header_here = nullptr;
    if (e == hipSuccess) ok(hipEventCreateWithFlags(&header_here, hipEventDisableTiming));
    if (e != hipSuccess) {
        setError("rpvg_hip_nested_subset_em: %s", hipGetErrorString(e));
        (void) hipStreamSynchronize(st);
        leavePairSearch(ctx);
        if (pinned_header) pinnedFree(pinned_header);
        if (header_here) (void) hipEventDestroy(header_here);
        return (e == hipErrorOutOfMemory) ? RPVG_HIP_ERR_ALLOC : RPVG_HIP_ERR_RUNTIME;
    }
    MatrixTotals * totals = reinterpret_cast<MatrixTotals *>(d_totals.ptr);
    MatrixTotals * bases = reinterpret_cast<MatrixTotals *>(d_bases.ptr);

    std::unique_ptr<HostScope> part_scope(new HostScope("subset em: select + offsets + expand queued"));
    int span = ctx->spanBegin(FAM_BUILD);
    SelectArgs sa;
    sa.num_matrices = M;
    sa.pair_count = search.d_tail.ptr;
    sa.pair_cap_off = search.d_pair_cap_off.ptr;
    sa.pair_first = search.d_out_first.ptr;
    sa.pair_second = search.d_out_second.ptr;
    sa.pair_value = search.d_out_value.ptr;
    sa.group_off = groups->d_group_off;
    sa.group_path_off = groups->d_group_path_off;
    sa.group_path = groups->d_group_path;
    sa.cluster = groups->d_cluster;
    sa.mat_rows = groups->mat_rows.ptr;
    sa.cluster_row_off = batch->cluster_row_off.ptr;
    sa.row_ent_off = batch->row_ent_off.ptr;
    sa.slot_off = d_slot_off;
    sa.min_hap_prob = min_hap_prob;
    sa.segment_rows = emFillSegmentRows();
    sa.slot_first = d_slot_first.ptr;
    sa.slot_second = d_slot_second.ptr;
    sa.slot_weight = d_slot_weight.ptr;
    sa.slot_length = d_slot_length.ptr;
    sa.slot_columns = d_slot_columns.ptr;
    sa.totals = totals;
    sa.header = header;
    sa.big_matrices = d_big_matrices.ptr;
    subsetSelectSmallKernel<<<dim3(M), dim3(64), 0, st>>>(sa);
    subsetSelectWideKernel<<<dim3(std::min<uint32_t>(M, 128)), dim3(256), 0, st>>>(sa);
    const bool check_build = !groups->build_checked && groups->build_error_flag.ptr;
    OffsetsArgs oa;
    oa.pair_count = search.d_tail.ptr;
    oa.log_evals = reinterpret_cast<const unsigned long long *>(search.d_tail.ptr + search.evals_word);
    oa.build_error = check_build ? groups->build_error_flag.ptr : nullptr;
    oa.num_matrices = M;
    oa.totals = totals;
    oa.bases = bases;
    oa.header = header;
    oa.cap_subsets = cap_subsets;
    oa.cap_length = cap_length;
    oa.cap_columns = cap_columns;
    oa.cap_rows = cap_rows;
    oa.cap_entries = cap_entries;
    oa.cap_items = cap_items;
    oa.seg_first = d_seg_first.ptr;
    oa.path_off = d_path_off.ptr;
    oa.col_off = d_col_off.ptr;
    subsetOffsetsKernel<<<dim3(1), dim3(1024), 0, st>>>(oa);
    // the header reaches the host while the rest runs
    SubsetHeader * h_header = static_cast<SubsetHeader *>(pinned_header);
    ok(hipMemcpyAsync(h_header, header, sizeof(SubsetHeader), hipMemcpyDeviceToHost, st));
    ok(hipEventRecord(header_here, st));
    ExpandArgs ea;
    ea.num_matrices = M;
    ea.header = header;
    ea.totals = totals;
    ea.bases = bases;
    ea.slot_off = d_slot_off;
    ea.slot_first = d_slot_first.ptr;
    ea.slot_second = d_slot_second.ptr;
    ea.slot_weight = d_slot_weight.ptr;
    ea.slot_length = d_slot_length.ptr;
    ea.slot_columns = d_slot_columns.ptr;
    ea.group_off = groups->d_group_off;
    ea.group_path_off = groups->d_group_path_off;
    ea.group_path = groups->d_group_path;
    ea.cluster = groups->d_cluster;
    ea.cluster_row_off = batch->cluster_row_off.ptr;
    ea.row_ent_off = batch->row_ent_off.ptr;
    ea.sub_cluster = d_sub_cluster.ptr;
    ea.sub_weight = d_sub_weight.ptr;
    ea.path_off = d_path_off.ptr;
    ea.path = d_path.ptr;
    ea.col_off = d_col_off.ptr;
    ea.col_path = d_col_path.ptr;
    ea.row_base = d_row_base.ptr;
    ea.ent_base = d_ent_base.ptr;
    ea.seg_first = d_seg_first.ptr;
    ea.item_problem = d_item_problem.ptr;
    ea.segment_rows = emFillSegmentRows();
    ea.subset_off = d_subset_off.ptr;
    subsetExpandKernel<<<dim3(M), dim3(256), 0, st>>>(ea);
    ok(hipGetLastError());
    ctx->spanEnd(span);
    ctx->stats.build_launches += 4;
    // the next search (another lane's) may start: what follows are this lane's EM problems
    leavePairSearch(ctx);

    part_scope.reset();
    EmProblemList list;
    list.P_bound = static_cast<uint32_t>(std::min<unsigned long long>(cap_subsets, 0xfffffffeull));
    list.d_num_problems = &header->num_problems;
    list.d_cluster = d_sub_cluster.ptr;
    list.d_col_off = d_col_off.ptr;
    list.d_col_path = d_col_path.ptr;
    list.d_row_base = d_row_base.ptr;
    list.d_ent_base = d_ent_base.ptr;
    list.rows_capacity = cap_rows;
    list.entries_capacity = cap_entries;
    list.d_seg_first = d_seg_first.ptr;
    list.d_item_problem = d_item_problem.ptr;
    list.items_bound = static_cast<uint32_t>(std::min<unsigned long long>(cap_items, 0xfffffffeull));
    list.d_num_items = &header->num_items;
    list.max_cols = max_paths + 1;
    list.max_cluster_paths = max_paths;
    list.max_cluster_work = max_cluster_work;  // (a cluster large enough for the grid bin: the solve waits for its problems' descriptions)
    list.wide_capacity = 0;
    EmOutputs out{d_abund.ptr, d_noise.ptr, d_iters.ptr, d_kept_rows.ptr, d_kept_entries.ptr, d_total.ptr};
    EmSolveWork work;
    work.zeroed_queues = search.d_extra_zero.ptr + header_room;
    if (e == hipSuccess) {
        rc = queueEmSolve(ctx, batch, list, max_em_its, max_rel_em_conv, out, work, false, collapse_precision);
        if (rc != RPVG_HIP_OK) {
            (void) hipStreamSynchronize(st);
            (void) hipEventDestroy(header_here);
            pinnedFree(pinned_header);
            return rc;
        }
    }
    part_scope.reset(new HostScope("subset em: merge + pack queued"));
    static_assert(sizeof(SubsetHeader) <= 192, "the merge's flag word sits behind the header");
    uint32_t * d_merge_bad = reinterpret_cast<uint32_t *>(search.d_extra_zero.ptr + 192);
    if (merge_here && e == hipSuccess) {
        MergeArgs ma;
        ma.header = header;
        ma.num_matrices = M;
        ma.subset_off = d_subset_off.ptr;
        ma.weight = d_sub_weight.ptr;
        ma.path_off = d_path_off.ptr;
        ma.path = d_path.ptr;
        ma.col_off = d_col_off.ptr;
        ma.col_path = d_col_path.ptr;
        ma.abundances = d_abund.ptr;
        ma.noise = d_noise.ptr;
        ma.total = d_total.ptr;
        ma.cluster = groups->d_cluster;
        ma.cluster_path_off = batch->cluster_path_off.ptr;
        ma.path_group_id = batch->path_group_id.ptr;
        ma.sort_key = d_merge_key.ptr;
        ma.sort_pos = d_merge_pos.ptr;
        ma.rec_a0 = d_merge_a0.ptr;
        ma.rec_a1 = d_merge_a1.ptr;
        ma.merge_bad = d_merge_bad;
        ma.packed = d_packed.ptr;
        const int merge_span = ctx->spanBegin(FAM_BUILD);
        subsetMergeKernel<<<dim3(M), dim3(256), 0, st>>>(ma);
        ctx->spanEnd(merge_span);
        ctx->stats.build_launches += 1;
    }
    PackArgs pa;
    pa.merge_bad = merge_here ? d_merge_bad : nullptr;
    pa.header = header;
    pa.num_matrices = M;
    pa.subset_off = d_subset_off.ptr;
    pa.weight = d_sub_weight.ptr;
    pa.path_off = d_path_off.ptr;
    pa.col_off = d_col_off.ptr;
    pa.abundances = d_abund.ptr;
    pa.noise = d_noise.ptr;
    pa.total = d_total.ptr;
    pa.path = d_path.ptr;
    pa.col_path = d_col_path.ptr;
    pa.iterations = d_iters.ptr;
    pa.kept_rows = d_kept_rows.ptr;
    pa.kept_entries = d_kept_entries.ptr;
    pa.packed = d_packed.ptr;
    packResultsKernel<<<dim3(256), dim3(256), 0, st>>>(pa);
    ok(hipGetLastError());

    part_scope.reset();
    scope.reset(new HostScope("subset em: header"));
    if (e == hipSuccess) ok(waitEvent(header_here));
    (void) hipEventDestroy(header_here);
    if (e != hipSuccess) {
        setError("rpvg_hip_nested_subset_em: %s", hipGetErrorString(e));
        (void) hipStreamSynchronize(st);
        pinnedFree(pinned_header);
        return (e == hipErrorOutOfMemory) ? RPVG_HIP_ERR_ALLOC : RPVG_HIP_ERR_RUNTIME;
    }
    const SubsetHeader got = *h_header;
    pinnedFree(pinned_header);
    {   // per unit of input, the largest of the recent calls (a call's figure fades by a tenth with every call after it)
        auto fold = [](double & kept, const unsigned long long needed, const unsigned long long units) {
            kept = std::max(0.9 * kept, static_cast<double>(needed) / static_cast<double>(std::max<unsigned long long>(units, 1)));
        };
        fold(hints.subsets_per_matrix, got.subsets, M);
        fold(hints.list_per_path, got.list_length, lane_paths);
        fold(hints.rows_per_row, got.rows, lane_rows);
        fold(hints.entries_per_entry, got.entries, lane_entries);
    }
    if (got.build_bad) {
        (void) hipStreamSynchronize(st);
        setError(got.build_bad == 2 ? "rpvg_hip_groups_build: a group lists a path twice" : "rpvg_hip_groups_build: a group refers to a path outside its cluster");
        return RPVG_HIP_ERR_INVALID;
    }
    groups->build_checked = true;
    if (got.overflow) {
        (void) hipStreamSynchronize(st);  // (the kernels behind the header saw zero problems)
        setError("rpvg_hip_nested_subset_em: not taken (%llu subsets, %llu rows, %llu entries, %llu selected-diplotype overflows: over the reserved "
                 "capacity)", got.subsets, got.rows, got.entries, got.select_overflow);
        if (!got.select_overflow) {  // (a matrix that selects more diplotypes than the select kernel holds fits no capacity)
            hints.retry = true;
            hints.retry_subsets = got.subsets;
            hints.retry_list_length = std::max(got.list_length, got.columns);
            hints.retry_rows = got.rows;
            hints.retry_entries = got.entries;
            *did_not_fit = true;
        }
        return RPVG_HIP_ERR_UNSUPPORTED;
    }
    // exactly what there is: one block, one copy, behind the EM kernels
    scope.reset(new HostScope("subset em: wait for the EM + download"));
    const uint64_t S = got.subsets;
    const PackedLayout lay = packedLayout(M, S, got.list_length, got.columns);
    if (pinnedAlloc(&res->block, std::max<size_t>(lay.bytes, 64)) != hipSuccess) {
        (void) hipStreamSynchronize(st);
        setError("rpvg_hip_nested_subset_em: out of page-locked host memory");
        return RPVG_HIP_ERR_ALLOC;
    }
    unsigned char * host = static_cast<unsigned char *>(res->block);
    ok(hipMemcpyAsync(host, d_packed.ptr, lay.bytes, hipMemcpyDeviceToHost, st));
    ok(waitStream(st));
    if (e != hipSuccess) {
        setError("rpvg_hip_nested_subset_em: %s", hipGetErrorString(e));
        return RPVG_HIP_ERR_RUNTIME;
    }
    accountPairSearch(ctx, groups, search, got.log_evals, got.kept_pairs);

    rpvg_hip_subset_em_view & v = res->view;
    v.num_matrices = M;
    v.subset_off = reinterpret_cast<const uint64_t *>(host + lay.subset_off);
    v.weight = reinterpret_cast<const double *>(host + lay.weight);
    v.path_off = reinterpret_cast<const uint64_t *>(host + lay.path_off);
    v.path = reinterpret_cast<const uint32_t *>(host + lay.path);
    v.col_off = reinterpret_cast<const uint64_t *>(host + lay.col_off);
    v.col_path = reinterpret_cast<const uint32_t *>(host + lay.col_path);
    v.abundances = reinterpret_cast<const double *>(host + lay.abundances);
    v.noise_count = reinterpret_cast<const double *>(host + lay.noise);
    v.total_count = reinterpret_cast<const double *>(host + lay.total);
    v.iterations = reinterpret_cast<const uint32_t *>(host + lay.iterations);
    if (merge_here) {
        if (*reinterpret_cast<const uint32_t *>(host + lay.merge_flags)) {
            setError("rpvg_hip_nested_subset_em: a diplotype's path subset holds more than two paths of one transcript (PathInfo::group_id): "
                     "the reference asserts against it (src/path_abundance_estimator.cpp:722)");
            return RPVG_HIP_ERR_INVALID;
        }
        v.set_count = reinterpret_cast<const uint32_t *>(host + lay.set_count);
        v.cluster_noise_count = reinterpret_cast<const double *>(host + lay.cluster_noise);
        v.set_first = reinterpret_cast<const uint32_t *>(host + lay.set_first);
        v.set_second = reinterpret_cast<const uint32_t *>(host + lay.set_second);
        v.set_posterior = reinterpret_cast<const double *>(host + lay.set_posterior);
        v.set_abundance = reinterpret_cast<const double *>(host + lay.set_abund);
    }
    res->num_subsets = S;
    accountEmSolve(ctx, static_cast<uint32_t>(S), v.col_off, reinterpret_cast<const uint32_t *>(host + lay.kept_rows),
                   reinterpret_cast<const uint32_t *>(host + lay.kept_entries), v.iterations);
    *result_out = res.release();
    return RPVG_HIP_OK;
}

extern "C" int rpvg_hip_nested_subset_em(rpvg_hip_ctx * ctx, const rpvg_hip_batch * batch, const rpvg_hip_groups * groups,
                                         const uint32_t * column_counts, double min_rel_likelihood, double min_hap_prob,
                                         uint32_t max_em_its, double max_rel_em_conv, double collapse_precision,
                                         rpvg_hip_subset_em ** result_out) {
    RPVG_REQUIRE(ctx && batch && groups && result_out, "rpvg_hip_nested_subset_em: NULL argument");
    *result_out = nullptr;
    // Capacities are planned from what recent calls needed per unit of input; a call that outgrows them learns what it needs from
    // its own header and selects, expands and solves again — cheap next to what the caller would otherwise do (the three separate
    // calls with the subsets on the host).
    bool did_not_fit = false;
    PairSearchWork search;  // (the diploid search runs once: the second attempt selects from the same pairs)
    int rc = nestedSubsetEmAttempt(ctx, batch, groups, column_counts, min_rel_likelihood, min_hap_prob, max_em_its, max_rel_em_conv, collapse_precision,
                                   result_out, &did_not_fit, search, false);
    if (rc == RPVG_HIP_ERR_UNSUPPORTED && did_not_fit) {
        static const bool trace = std::getenv("RPVG_AMD_TRACE") != nullptr;
        if (trace) std::fprintf(stderr, "[rpvg_hip trace]   subset em: second attempt (%s)\n", rpvg_hip_last_error());
        rc = nestedSubsetEmAttempt(ctx, batch, groups, column_counts, min_rel_likelihood, min_hap_prob, max_em_its, max_rel_em_conv, collapse_precision,
                                   result_out, &did_not_fit, search, true);
    }
    return rc;
}

extern "C" int rpvg_hip_subset_em_get(const rpvg_hip_subset_em * result, rpvg_hip_subset_em_view * view_out) {
    RPVG_REQUIRE(result && view_out, "rpvg_hip_subset_em_get: NULL argument");
    *view_out = result->view;
    view_out->num_matrices = result->num_matrices;
    return RPVG_HIP_OK;
}

extern "C" void rpvg_hip_subset_em_free(rpvg_hip_subset_em * result) { delete result; }
