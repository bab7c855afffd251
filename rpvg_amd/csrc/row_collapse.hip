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
// Two stages, both queued behind the build without a host round trip:
//   1. detection (always): every row carries a projection key (sum of its values with fixed weights in [1, 2), written
//      by the build kernels as (matrix, fixed-point key): collapseSortKey, common.hpp).  Rows within prob_precision of each other in all columns have keys within
//      2 (G + 1) prob_precision, so one radix sort of (matrix, key) over all matrices and a forward window scan find
//      every such pair.  A matrix whose close pairs are all equal up to rounding (<= 1e-13 relative) is left alone:
//      whichever row heads a run there, the values that would move are the same to 13 digits.
//   2. replay (flagged matrices only, one workgroup each): bitonic network over the row indices with the
//      reference's comparator, run heads found with the reference's compare-with-the-head rule, values of the
//      head copied over its run (group columns, noise, row maximum).  The tolerant comparison is not a strict weak
//      order; where it is inconsistent the reference's own result is whatever std::sort makes of it.

#include "common.hpp"

#include <hipcub/hipcub.hpp>

#include <cfloat>

using namespace rpvg_hip_detail;

namespace {

constexpr int kKeyFractionBits = kCollapseKeyFractionBits;
constexpr int kKeyBits = kCollapseKeyBits;
constexpr uint32_t kMaxWindowCompares = 64;    // a row with more candidates than this sends its matrix to the replay
constexpr double kEquivalentRelative = 1e-13;  // close rows that differ by no more than this are interchangeable
constexpr uint32_t kSortLdsRows = 8192;        // index arrays up to this size are sorted in LDS

__device__ __forceinline__ bool tolerantEqual(const double a, const double b) {  // Utils::doubleCompare, src/utils.hpp:87-93
    return a == b || fabs(a - b) < fabs(fmin(a, b)) * (DBL_EPSILON * 100);
}

struct MatrixView {
    const double * values;  // column-major R x G
    const double * noise;
    const double * count;
    uint64_t R;
    uint32_t G;
};

// probabilityCountRowSorter (src/path_estimator.cpp:13-31) on rows a, b of one matrix
__device__ bool rowLess(const MatrixView & mv, const uint32_t a, const uint32_t b) {
    for (uint32_t c = 0; c < mv.G; ++c) {
        const double x = mv.values[static_cast<uint64_t>(c) * mv.R + a], y = mv.values[static_cast<uint64_t>(c) * mv.R + b];
        if (!tolerantEqual(x, y)) return x < y;
    }
    {
        const double x = mv.noise[a], y = mv.noise[b];
        if (!tolerantEqual(x, y)) return x < y;
    }
    const double x = mv.count[a], y = mv.count[b];
    if (!tolerantEqual(x, y)) return x < y;
    return false;
}

// every column (noise included) within `precision` of each other, absolutely (src/path_estimator.cpp:232-239);
// *equivalent: additionally equal up to rounding in every column
__device__ bool rowsClose(const MatrixView & mv, const uint32_t a, const uint32_t b, const double precision, bool * equivalent) {
    bool eq = true;
    for (uint32_t c = 0; c <= mv.G; ++c) {
        const double x = c < mv.G ? mv.values[static_cast<uint64_t>(c) * mv.R + a] : mv.noise[a];
        const double y = c < mv.G ? mv.values[static_cast<uint64_t>(c) * mv.R + b] : mv.noise[b];
        const double d = fabs(x - y);
        if (d >= precision) return false;
        if (d > kEquivalentRelative * fmin(fabs(x), fabs(y))) eq = false;
    }
    if (equivalent) *equivalent = eq;
    return true;
}

__device__ bool rowsIdentical(const MatrixView & mv, const uint32_t a, const uint32_t b) {
    for (uint32_t c = 0; c < mv.G; ++c)
        if (mv.values[static_cast<uint64_t>(c) * mv.R + a] != mv.values[static_cast<uint64_t>(c) * mv.R + b]) return false;
    return mv.noise[a] == mv.noise[b];
}

__device__ __forceinline__ MatrixView viewOf(const uint32_t m, const uint64_t * __restrict__ mat_val_off, const uint64_t * __restrict__ mat_row_off,
                                             const uint64_t * __restrict__ mat_rows, const uint32_t * __restrict__ mat_cols,
                                             const double * __restrict__ values, const double * __restrict__ row_noise,
                                             const double * __restrict__ row_count) {
    MatrixView mv;
    mv.values = values + mat_val_off[m];
    mv.noise = row_noise + mat_row_off[m];
    mv.count = row_count + mat_row_off[m];
    mv.R = mat_rows[m];
    mv.G = mat_cols[m];
    return mv;
}

// ---- stage 1: detection -------------------------------------------------------------------------------------

// same_prev[p] = the row at sorted position p is bit for bit the row at p - 1 (same matrix)
__global__ void collapseSamePrevKernel(const uint64_t total_rows, const uint64_t * __restrict__ sort_key, const uint32_t * __restrict__ sort_row,
                                       const uint64_t * __restrict__ mat_val_off,
                                       const uint64_t * __restrict__ mat_row_off, const uint64_t * __restrict__ mat_rows,
                                       const uint32_t * __restrict__ mat_cols, const double * __restrict__ values,
                                       const double * __restrict__ row_noise, const double * __restrict__ row_count,
                                       uint8_t * __restrict__ same_prev) {
    const uint64_t p = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (p >= total_rows) return;
    uint8_t same = 0;
    if (p > 0 && sort_key[p] == sort_key[p - 1]) {  // equal keys: nearly always the same row twice; the comparison leaves at the first difference
        const uint32_t m = static_cast<uint32_t>(sort_key[p] >> kKeyBits);
        const MatrixView mv = viewOf(m, mat_val_off, mat_row_off, mat_rows, mat_cols, values, row_noise, row_count);
        const uint64_t r0 = mat_row_off[m];
        same = rowsIdentical(mv, static_cast<uint32_t>(sort_row[p] - r0), static_cast<uint32_t>(sort_row[p - 1] - r0)) ? 1 : 0;
    }
    same_prev[p] = same;
}

// every row looks at the rows after it whose keys lie within the window; a close pair that is not equal up to
// rounding (or a window too crowded to look through) flags the matrix
__global__ void collapseWindowKernel(const uint64_t total_rows, const double precision, const uint64_t * __restrict__ sort_key,
                                     const uint32_t * __restrict__ sort_row, const uint8_t * __restrict__ same_prev,
                                     const uint64_t * __restrict__ mat_val_off, const uint64_t * __restrict__ mat_row_off,
                                     const uint64_t * __restrict__ mat_rows, const uint32_t * __restrict__ mat_cols,
                                     const double * __restrict__ values, const double * __restrict__ row_noise,
                                     const double * __restrict__ row_count, uint32_t * __restrict__ mat_flag,
                                     uint32_t * __restrict__ info) {
    const uint64_t p = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (p >= total_rows) return;
    const uint64_t key = sort_key[p];
    const uint32_t m = static_cast<uint32_t>(key >> kKeyBits);
    const MatrixView mv = viewOf(m, mat_val_off, mat_row_off, mat_rows, mat_cols, values, row_noise, row_count);
    // |key_a - key_b| <= sum_c w_c |a_c - b_c| < 2 (G + 1) precision for close rows; + rounding of the keys, + 2 quanta
    const double window = 2.0001 * (mv.G + 1) * precision + 1e-12;
    const uint64_t window_q = static_cast<uint64_t>(window * static_cast<double>(1ull << kKeyFractionBits)) + 2;
    const uint64_t r0 = mat_row_off[m];
    const uint32_t a = static_cast<uint32_t>(sort_row[p] - r0);
    uint32_t compares = 0;
    bool flag = false;
    for (uint64_t q = p + 1; q < total_rows; ++q) {
        const uint64_t other = sort_key[q];
        if ((other >> kKeyBits) != m || other - key > window_q) break;
        if (q > p + 1 && same_prev[q]) continue;  // bit for bit the candidate before it
        if (++compares > kMaxWindowCompares) {
            flag = true;
            break;
        }
        bool equivalent = true;
        if (rowsClose(mv, a, static_cast<uint32_t>(sort_row[q] - r0), precision, &equivalent) && !equivalent) {
            flag = true;
            break;
        }
    }
    if (flag && atomicExch(&mat_flag[m], 1u) == 0u) atomicAdd(&info[0], 1u);
}

// ---- stage 2: replay ------------------------------------------------------------------------------------------

constexpr uint32_t kNoRow = 0xFFFFFFFFu;

// first index in [begin, end) for which pred holds, `end` if none; block-uniform result
template <typename Pred>
__device__ uint64_t blockFirstTrue(const uint64_t begin, const uint64_t end, unsigned long long * shared_min, Pred pred) {
    for (uint64_t base = begin; base < end; base += blockDim.x) {
        if (threadIdx.x == 0) *shared_min = end;
        __syncthreads();
        const uint64_t q = base + threadIdx.x;
        if (q < end && pred(q)) atomicMin(shared_min, static_cast<unsigned long long>(q));
        __syncthreads();
        const uint64_t found = *shared_min;
        __syncthreads();
        if (found < end) return found;
    }
    return end;
}

__global__ __launch_bounds__(1024) void collapseReplayKernel(
    const uint32_t num_matrices, const double precision, const uint32_t * __restrict__ mat_flag,
    const uint64_t * __restrict__ mat_val_off, const uint64_t * __restrict__ mat_row_off, const uint64_t * __restrict__ mat_rows,
    const uint32_t * __restrict__ mat_cols, double * __restrict__ values, double * __restrict__ row_noise,
    const double * __restrict__ row_count, double * __restrict__ rowmax, uint32_t * __restrict__ mat_fast, uint32_t * __restrict__ mat_mid,
    uint32_t * __restrict__ order_scratch,  // [2 * total rows]: sorted row indices of matrix m at 2 * mat_row_off[m]
    uint32_t * __restrict__ head_scratch,   // [total rows]: sorted position of the run head of every sorted position
    uint8_t * __restrict__ close_prev,      // [total rows]: sorted position p is close to p - 1
    uint32_t * __restrict__ info) {
    __shared__ uint32_t lds_order[kSortLdsRows];
    __shared__ unsigned long long shared_min;
    __shared__ uint32_t demote;
    const uint32_t m = blockIdx.x;
    if (m >= num_matrices || !mat_flag[m]) return;
    const MatrixView mv = viewOf(m, mat_val_off, mat_row_off, mat_rows, mat_cols, values, row_noise, row_count);
    const uint64_t R = mv.R;
    if (R < 2) return;
    uint64_t padded = 1;
    while (padded < R) padded <<= 1;
    uint32_t * order = padded <= kSortLdsRows ? lds_order : order_scratch + 2 * mat_row_off[m];
    if (threadIdx.x == 0) demote = 0;
    for (uint64_t i = threadIdx.x; i < padded; i += blockDim.x) order[i] = i < R ? static_cast<uint32_t>(i) : kNoRow;
    __syncthreads();
    // bitonic network; kNoRow sorts behind every row
    for (uint64_t k = 2; k <= padded; k <<= 1) {
        for (uint64_t j = k >> 1; j > 0; j >>= 1) {
            for (uint64_t t = threadIdx.x; t < (padded >> 1); t += blockDim.x) {
                const uint64_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1));  // bit j of i is clear
                const uint64_t l = i | j;
                const uint32_t a = order[i], b = order[l];
                const bool ascending = (i & k) == 0;
                bool swap;
                if (a == kNoRow || b == kNoRow) {
                    swap = ascending ? (a == kNoRow && b != kNoRow) : (b == kNoRow && a != kNoRow);
                } else {
                    swap = ascending ? rowLess(mv, b, a) : rowLess(mv, a, b);
                }
                if (swap) {
                    order[i] = b;
                    order[l] = a;
                }
            }
            __syncthreads();
        }
    }
    uint32_t * head_of = head_scratch + mat_row_off[m];
    uint8_t * close = close_prev + mat_row_off[m];
    for (uint64_t p = threadIdx.x; p < R; p += blockDim.x) {
        head_of[p] = static_cast<uint32_t>(p);
        close[p] = (p > 0 && rowsClose(mv, order[p - 1], order[p], precision, nullptr)) ? 1 : 0;
    }
    __syncthreads();
    // Runs (src/path_estimator.cpp:226-255): a row joins the run of the current head if it is close to the head,
    // otherwise it becomes the head.  A row whose predecessor is a head of its own is decided by close[]; the rows
    // after a join are compared with the head, a block-wide chunk at a time.
    uint64_t p = 1;
    while (p < R) {
        const uint64_t joiner = blockFirstTrue(p, R, &shared_min, [&](const uint64_t q) { return close[q] != 0; });
        if (joiner >= R) break;
        const uint64_t head = joiner - 1;  // every row in [p, joiner) is a head; so is joiner - 1 (p - 1 is one)
        const uint32_t head_row = order[head];
        const uint64_t next_head = blockFirstTrue(joiner, R, &shared_min, [&](const uint64_t q) {
            return !rowsClose(mv, head_row, order[q], precision, nullptr);
        });
        for (uint64_t q = joiner + threadIdx.x; q < next_head; q += blockDim.x) head_of[q] = static_cast<uint32_t>(head);
        p = next_head + 1;  // next_head heads a run of its own; close[next_head + 1] compares with it
        __syncthreads();
    }
    __syncthreads();
    // the rows of a run take the values of its head
    double * M = values + mat_val_off[m];
    double * nz = row_noise + mat_row_off[m];
    double * rm = rowmax + mat_row_off[m];
    const uint32_t fast_mid_end = mat_mid[m];
    uint32_t replaced = 0;
    for (uint64_t q = threadIdx.x; q < R; q += blockDim.x) {
        const uint32_t h = head_of[q];
        if (h == q) continue;
        const uint32_t dst = order[q], src = order[h];
        for (uint32_t c = 0; c < mv.G; ++c) M[static_cast<uint64_t>(c) * R + dst] = M[static_cast<uint64_t>(c) * R + src];
        const double noise = nz[src];
        nz[dst] = noise;
        rm[dst] = rm[src];
        ++replaced;
        // a product-path row (LogProduct, common.hpp) needs noise >= kProductMinNoise: if the head's is below, the
        // whole matrix takes the logarithm path
        if (dst < fast_mid_end && !(noise >= kProductMinNoise)) demote = 1;
    }
    if (replaced) atomicAdd(&info[1], replaced);
    __syncthreads();
    if (threadIdx.x == 0 && demote) {
        mat_fast[m] = 0;
        mat_mid[m] = 0;
    }
}

}  // namespace

// Queues the collapse of the matrices of `g` on `st` behind their build (rpvg_hip_groups_build).
hipError_t rpvg_hip_detail::queueRowCollapse(rpvg_hip_ctx * ctx, rpvg_hip_groups * g, const uint64_t total_rows, const double precision,
                                             hipStream_t st) {
    (void) ctx;
    const uint32_t M = g->num_matrices;
    if (M == 0 || total_rows == 0) return hipSuccess;
    if (total_rows > 0x7fffffffull || (static_cast<uint64_t>(M) >> (64 - kKeyBits)) != 0) return hipErrorInvalidValue;
    struct CollapseTemporaries {
        DeviceBuffer<uint64_t> key_out;
        DeviceBuffer<uint32_t> row_out, mat_flag, order, head;
        DeviceBuffer<uint8_t> same_prev, close_prev;
        DeviceBuffer<unsigned char> sort_tmp;
    };
    std::shared_ptr<CollapseTemporaries> tmp = std::make_shared<CollapseTemporaries>();
    g->build_temporaries.emplace_back(tmp);
    hipError_t e = hipSuccess;
    auto ok = [&](hipError_t r) { if (e == hipSuccess) e = r; return e == hipSuccess; };
    ok(tmp->key_out.alloc(total_rows));
    ok(tmp->row_out.alloc(total_rows));
    ok(tmp->mat_flag.alloc(M));
    ok(tmp->order.alloc(2 * total_rows));
    ok(tmp->head.alloc(total_rows));
    ok(tmp->same_prev.alloc(total_rows));
    ok(tmp->close_prev.alloc(total_rows));
    ok(g->collapse_info.alloc(2));
    int matrix_bits = 1;
    while ((1ull << matrix_bits) < M) ++matrix_bits;
    const int end_bit = kKeyBits + matrix_bits;
    size_t sort_bytes = 0;
    if (e == hipSuccess) ok(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, g->collapse_key.ptr, tmp->key_out.ptr, g->collapse_row.ptr, tmp->row_out.ptr,
                                                               static_cast<int>(total_rows), 0, end_bit, st));
    ok(tmp->sort_tmp.alloc(sort_bytes));
    if (e != hipSuccess) return e;
    ok(hipMemsetAsync(tmp->mat_flag.ptr, 0, M * sizeof(uint32_t), st));
    ok(hipMemsetAsync(g->collapse_info.ptr, 0, 2 * sizeof(uint32_t), st));
    const uint32_t row_blocks = static_cast<uint32_t>((total_rows + 255) / 256);
    ok(hipcub::DeviceRadixSort::SortPairs(tmp->sort_tmp.ptr, sort_bytes, g->collapse_key.ptr, tmp->key_out.ptr, g->collapse_row.ptr, tmp->row_out.ptr,
                                          static_cast<int>(total_rows), 0, end_bit, st));
    collapseSamePrevKernel<<<dim3(row_blocks), dim3(256), 0, st>>>(total_rows, tmp->key_out.ptr, tmp->row_out.ptr, g->mat_val_off.ptr,
                                                                 g->mat_row_off.ptr, g->mat_rows.ptr, g->mat_cols.ptr, g->values.ptr,
                                                                 g->row_noise.ptr, g->row_count.ptr, tmp->same_prev.ptr);
    collapseWindowKernel<<<dim3(row_blocks), dim3(256), 0, st>>>(total_rows, precision, tmp->key_out.ptr, tmp->row_out.ptr, tmp->same_prev.ptr,
                                                               g->mat_val_off.ptr, g->mat_row_off.ptr, g->mat_rows.ptr, g->mat_cols.ptr,
                                                               g->values.ptr, g->row_noise.ptr, g->row_count.ptr, tmp->mat_flag.ptr,
                                                               g->collapse_info.ptr);
    collapseReplayKernel<<<dim3(M), dim3(1024), 0, st>>>(M, precision, tmp->mat_flag.ptr, g->mat_val_off.ptr, g->mat_row_off.ptr, g->mat_rows.ptr,
                                                        g->mat_cols.ptr, g->values.ptr, g->row_noise.ptr, g->row_count.ptr, g->rowmax.ptr,
                                                        g->mat_fast.ptr, g->mat_mid.ptr, tmp->order.ptr, tmp->head.ptr, tmp->close_prev.ptr,
                                                        g->collapse_info.ptr);
    ok(hipGetLastError());
    return e;
}

extern "C" int rpvg_hip_groups_collapse_info(rpvg_hip_ctx * ctx, const rpvg_hip_groups * groups, uint32_t * matrices_replayed,
                                             uint32_t * rows_replaced) {
    RPVG_REQUIRE(ctx && groups, "rpvg_hip_groups_collapse_info: NULL argument");
    uint32_t info[2] = {0, 0};
    if (groups->collapse_info.ptr) {
        std::lock_guard<std::mutex> lock(ctx->mutex);
        RPVG_HIP_CHECK(hipSetDevice(ctx->device));
        RPVG_HIP_CHECK(hipMemcpyAsync(info, groups->collapse_info.ptr, sizeof(info), hipMemcpyDeviceToHost, ctx->stream));
        RPVG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
    if (matrices_replayed) *matrices_replayed = info[0];
    if (rows_replaced) *rows_replaced = info[1];
    return RPVG_HIP_OK;
}
