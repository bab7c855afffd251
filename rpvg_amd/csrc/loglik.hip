// Group (haplotype / diplotype) matrices and their log-likelihood
// contraction on the GPU (gfx950).
//
// Takes over
//   constructGroupedProbabilityMatrix / constructProbabilityMatrix
//                                            src/path_estimator.cpp:55-77,115-154
//   addNoiseAndNormalizeProbabilityMatrix    src/path_estimator.cpp:156-166
//   rowwise().maxCoeff()                     src/path_estimator.cpp:414
//   read_counts * (noise + sum cols / g).array().log().matrix()
//                                            src/path_estimator.cpp:354-361,424-427,439,527-545
//
// Layout: one column-major R_m x G_m matrix per requested (cluster, grouping),
// so that a wave walking the rows of one or two columns reads contiguous
// memory.  The rows of a matrix are its cluster's rows ordered by class (count
// 1 first: partitionRowsKernel), with counts and noise copied in that order, so
// that the contraction — FP64-issue bound, not HBM bound: a cluster's matrix is
// re-read from L2 by every request that touches it — replaces most logarithms
// by a running product (LogProduct, common.hpp).

#include "common.hpp"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <memory>
#include <type_traits>

using namespace rpvg_hip_detail;


namespace {

constexpr uint32_t kNoMember = 0xFFFFFFFFu;

// Row order of every matrix: the rows of its cluster by class (rowClass, common.hpp) — count 1 first, then the mid
// counts ascending, then the rest — in cluster order within a class (stable, deterministic).  One workgroup per
// matrix: class histogram, then placement chunk by chunk with ballot ranks.
// (BLOCK threads per matrix: 1 024 — a matrix of 20 000 rows is 80 rounds of three barriers on 256 threads, 0.16 ms at the head of
// a lane's chain of kernels)
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void partitionRowsKernel(
    const uint32_t num_matrices, const uint64_t * __restrict__ mat_row_off, const uint64_t * __restrict__ mat_row0,
    const uint64_t * __restrict__ mat_rows, const double * __restrict__ row_count, const double * __restrict__ row_noise,
    uint32_t * __restrict__ row_perm, double * __restrict__ count_out, double * __restrict__ noise_out,
    uint32_t * __restrict__ mat_fast, uint32_t * __restrict__ mat_mid, const uint32_t mid_min_rows) {
    __shared__ uint32_t class_base[kNumRowClasses];       // next free slot of every class
    __shared__ uint32_t wave_count[BLOCK / 64][kNumRowClasses];
    const uint32_t m = blockIdx.x;
    if (m >= num_matrices) return;
    const uint32_t R = static_cast<uint32_t>(mat_rows[m]);
    const double * cnt = row_count + mat_row0[m];
    const double * nz = row_noise + mat_row0[m];
    const uint64_t out0 = mat_row_off[m];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < kNumRowClasses) class_base[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < R; i += BLOCK) atomicAdd(&class_base[rowClass(cnt[i], nz[i], R >= mid_min_rows)], 1u);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t running = 0;
        for (uint32_t c = 0; c < kNumRowClasses; ++c) {
            const uint32_t n = class_base[c];
            class_base[c] = running;
            running += n;
            if (c == 0) mat_fast[m] = running;
            if (c + 2 == kNumRowClasses) mat_mid[m] = running;
        }
    }
    __syncthreads();
    for (uint32_t c0 = 0; c0 < R; c0 += BLOCK) {
        const uint32_t i = c0 + threadIdx.x;
        const bool in = i < R;
        const double c = in ? cnt[i] : 0.0, z = in ? nz[i] : 0.0;
        const uint32_t mine = in ? rowClass(c, z, R >= mid_min_rows) : kNumRowClasses;
        uint32_t rank = 0;
#pragma unroll
        for (uint32_t k = 0; k < kNumRowClasses; ++k) {
            const unsigned long long ballot = __ballot(mine == k);
            if (mine == k) rank = __popcll(ballot & ((1ull << lane) - 1ull));
            if (lane == 0) wave_count[wave][k] = __popcll(ballot);
        }
        __syncthreads();
        if (in) {
            uint32_t dest = class_base[mine] + rank;
            for (int w = 0; w < wave; ++w) dest += wave_count[w][mine];
            row_perm[out0 + dest] = i;
            count_out[out0 + dest] = c;
            noise_out[out0 + dest] = z;
        }
        __syncthreads();
        if (threadIdx.x < kNumRowClasses) {
            uint32_t added = 0;
            for (int w = 0; w < BLOCK / 64; ++w) added += wave_count[w][threadIdx.x];
            class_base[threadIdx.x] += added;
        }
        __syncthreads();
    }
}

// one workgroup per (matrix, chunk of 256 rows), thread per row
__global__ __launch_bounds__(256) void groupsBuildKernel(
    const uint32_t num_items, const uint32_t * __restrict__ item_matrix, const uint32_t * __restrict__ item_chunk,
    const uint64_t * __restrict__ mat_val_off, const uint64_t * __restrict__ mat_row_off,
    const uint64_t * __restrict__ mat_row0, const uint64_t * __restrict__ mat_rows, const uint32_t * __restrict__ mat_cols,
    const uint64_t * __restrict__ mat_inc_off,   // [M] offset of the matrix's path->groups CSR offsets
    const uint64_t * __restrict__ path_grp_off,  // per matrix N_k+1 offsets (absolute into path_grp)
    const uint32_t * __restrict__ path_grp, const uint64_t * __restrict__ row_ent_off,
    const uint32_t * __restrict__ ent_path, const double * __restrict__ ent_prob, const double * __restrict__ row_noise,
    const uint32_t * __restrict__ row_perm, const int normalise, double * __restrict__ values, double * __restrict__ rowmax,
    uint64_t * __restrict__ collapse_key, uint32_t * __restrict__ collapse_row, uint64_t * __restrict__ collapse_mask) {  // null: no row collapse
    if (blockIdx.x >= num_items) return;
    const uint32_t m = item_matrix[blockIdx.x];
    const uint64_t R = mat_rows[m], r0 = mat_row0[m];
    const uint32_t G = mat_cols[m];
    double * M = values + mat_val_off[m];
    double * rm = rowmax + mat_row_off[m];
    const uint64_t * pgo = path_grp_off + mat_inc_off[m];
    const uint64_t i = static_cast<uint64_t>(item_chunk[blockIdx.x]) * 256 + threadIdx.x;
    if (i < R) {
        const uint64_t r = r0 + row_perm[mat_row_off[m] + i];  // matrix row i = this row of the cluster
        for (uint64_t e = row_ent_off[r]; e < row_ent_off[r + 1]; ++e) {
            const uint32_t p = ent_path[e];
            const double v = ent_prob[e];
            for (uint64_t x = pgo[p]; x < pgo[p + 1]; ++x) M[static_cast<uint64_t>(path_grp[x]) * R + i] += v;
        }
        double mx = 0.0;
        if (normalise) {
            double rowsum = 0.0;
            for (uint32_t g = 0; g < G; ++g) rowsum += M[static_cast<uint64_t>(g) * R + i];
            const double keep = 1 - row_noise[r];
            double key = collapseWeight(G) * row_noise[r];
            uint64_t pattern = 0;
            for (uint32_t g = 0; g < G; ++g) {
                double v = (M[static_cast<uint64_t>(g) * R + i] / rowsum) * keep;
                if (v != v) v = 0.0;  // 0/0 rows -> 0 (src/path_estimator.cpp:162)
                M[static_cast<uint64_t>(g) * R + i] = v;
                mx = (g == 0) ? v : fmax(mx, v);
                key = fma(collapseWeight(g), v, key);
                if (g < 64 && v != 0.0) pattern |= 1ull << g;
            }
            if (collapse_key) {
                collapse_key[mat_row_off[m] + i] = collapseSortKey(m, key, mx);
                collapse_row[mat_row_off[m] + i] = static_cast<uint32_t>(mat_row_off[m] + i);
                collapse_mask[mat_row_off[m] + i] = pattern;
            }
        } else {
            for (uint32_t g = 0; g < G; ++g) {
                const double v = M[static_cast<uint64_t>(g) * R + i];
                mx = (g == 0) ? v : fmax(mx, v);
            }
        }
        rm[i] = mx;
    }
}


// Rows per LDS tile for a matrix with G columns: the tile (G x rows doubles) must fit kTileDoubles; a power of two
// of at least 16 rows (128-byte runs per column when the tile is written out).  0 = too wide: global-memory kernel.
constexpr uint32_t kTileDoubles = 4096;  // 32 KB: four workgroups per CU (64 KB tiles, two per CU, were 11 % slower: the rows hang on dependent loads)

__host__ __device__ inline uint32_t tileRows(const uint32_t G) {
    if (G * 16u > kTileDoubles) return 0;
    uint32_t rows = 256;
    while (rows * G > kTileDoubles) rows >>= 1;
    return rows;
}

// The same construction through an LDS tile: one workgroup per (matrix, tile of rows).  The rows of the tile are
// accumulated, normalised and scanned for their maximum in LDS by the thread that owns the row (same order of
// additions as groupsBuildKernel: the results are bit-identical), then the tile leaves in one coalesced write —
// one HBM write per matrix element instead of zero-fill + read-modify-writes + two normalisation passes.
__global__ __launch_bounds__(256) void groupsBuildTileKernel(
    const uint32_t num_items, const uint32_t * __restrict__ item_matrix, const uint32_t * __restrict__ item_chunk,
    const uint64_t * __restrict__ mat_val_off, const uint64_t * __restrict__ mat_row_off,
    const uint64_t * __restrict__ mat_row0, const uint64_t * __restrict__ mat_rows, const uint32_t * __restrict__ mat_cols,
    const uint64_t * __restrict__ mat_inc_off, const uint64_t * __restrict__ path_grp_off,
    const uint32_t * __restrict__ path_grp, const uint64_t * __restrict__ row_ent_off,
    const uint32_t * __restrict__ ent_path, const double * __restrict__ ent_prob, const double * __restrict__ row_noise,
    const uint32_t * __restrict__ row_perm, const int normalise, double * __restrict__ values, double * __restrict__ rowmax,
    uint64_t * __restrict__ collapse_key, uint32_t * __restrict__ collapse_row, uint64_t * __restrict__ collapse_mask) {  // null: no row collapse
    extern __shared__ double tile[];
    if (blockIdx.x >= num_items) return;
    const uint32_t m = item_matrix[blockIdx.x];
    const uint64_t R = mat_rows[m], r0 = mat_row0[m];
    const uint32_t G = mat_cols[m];
    const uint32_t Rc = tileRows(G);
    const uint64_t i0 = static_cast<uint64_t>(item_chunk[blockIdx.x]) * Rc;
    const uint32_t nrows = static_cast<uint32_t>(min(static_cast<uint64_t>(Rc), R - i0));
    const uint32_t Rs = Rc + 1;  // odd stride: a walk along a row of the tile (second loop below) meets every LDS bank
    const uint32_t cells = G * Rc;
    for (uint32_t idx = threadIdx.x; idx < G * Rs; idx += blockDim.x) tile[idx] = 0.0;
    __syncthreads();
    const uint32_t t = threadIdx.x;
    if (t < nrows) {
        const uint64_t r = r0 + row_perm[mat_row_off[m] + i0 + t];  // matrix row i0 + t = this row of the cluster
        const uint64_t * pgo = path_grp_off + mat_inc_off[m];
        // The row's entries hang on a chain of dependent loads (entry -> path -> its columns' offsets -> column): four
        // entries walk it side by side, the additions stay in entry order.
        const uint64_t e_end = row_ent_off[r + 1];
        for (uint64_t e = row_ent_off[r]; e < e_end; e += 4) {
            uint32_t p[4];
            double v[4];
            uint64_t x_begin[4], x_end[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool in = e + k < e_end;
                p[k] = in ? ent_path[e + k] : 0u;
                v[k] = in ? ent_prob[e + k] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool in = e + k < e_end;
                x_begin[k] = in ? pgo[p[k]] : 0;
                x_end[k] = in ? pgo[p[k] + 1] : 0;
            }
            // A path lies in as many columns as haplotype groups carry it — dozens for a common allele — and every column
            // index is a load of its own: the first kAhead of all four entries are fetched before anything is added (32
            // loads in flight), longer lists eight at a time.  The additions keep their order: entry after entry, column
            // after column.
            constexpr int kAhead = 8;
            uint32_t columns[4][kAhead];
#pragma unroll
            for (int j = 0; j < kAhead; ++j) {
#pragma unroll
                for (int k = 0; k < 4; ++k) columns[k][j] = x_begin[k] + j < x_end[k] ? path_grp[x_begin[k] + j] : 0u;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
#pragma unroll
                for (int j = 0; j < kAhead; ++j) {
                    if (x_begin[k] + j < x_end[k]) tile[columns[k][j] * Rs + t] += v[k];
                }
                for (uint64_t x = x_begin[k] + kAhead; x < x_end[k]; x += kAhead) {
                    uint32_t more[kAhead];
#pragma unroll
                    for (int j = 0; j < kAhead; ++j) more[j] = x + j < x_end[k] ? path_grp[x + j] : 0u;
#pragma unroll
                    for (int j = 0; j < kAhead; ++j) {
                        if (x + j < x_end[k]) tile[more[j] * Rs + t] += v[k];
                    }
                }
            }
        }
    }
    // The rows' second half — row sum, normalisation, maximum, sort key, zero pattern — with as many lanes per row as the
    // workgroup has to spare (a tile of 64 columns is 64 rows: four lanes each, neighbours in a wave, every one a quarter of
    // the columns; the owner thread alone spent 8 of the tile's 23 us here, one division and three dependent chains per
    // column).  The row sum keeps its order (column after column, by the row's first lane); the key is a sum of the lanes'
    // partial keys in a fixed order.
    __syncthreads();
    uint32_t parts = 1;
    while (parts < 8 && 2 * parts * Rc <= blockDim.x) parts *= 2;
    const uint32_t row = threadIdx.x / parts, part = threadIdx.x % parts;
    if (row < Rc) {  // (whole groups of `parts` lanes: the shuffles below stay inside them)
        const bool valid = row < nrows;
        const uint64_t r = valid ? r0 + row_perm[mat_row_off[m] + i0 + row] : 0;
        double mx = 0.0;
        if (normalise) {
            double rowsum = 0.0;
            if (valid && part == 0) {
                for (uint32_t g = 0; g < G; ++g) rowsum += tile[g * Rs + row];
            }
            rowsum = __shfl(rowsum, static_cast<int>(threadIdx.x & 63u) - static_cast<int>(part));
            const double noise = valid ? row_noise[r] : 0.0;
            const double keep = 1 - noise;
            double key = part == 0 ? collapseWeight(G) * noise : 0.0;
            uint64_t pattern = 0;
            if (valid) {
                for (uint32_t g = part; g < G; g += parts) {
                    double v = (tile[g * Rs + row] / rowsum) * keep;
                    if (v != v) v = 0.0;  // 0/0 rows -> 0 (src/path_estimator.cpp:162)
                    tile[g * Rs + row] = v;
                    mx = fmax(mx, v);
                    key = fma(collapseWeight(g), v, key);
                    if (g < 64 && v != 0.0) pattern |= 1ull << g;
                }
            }
            for (uint32_t step = 1; step < parts; step *= 2) {
                mx = fmax(mx, __shfl_xor(mx, static_cast<int>(step)));
                key += __shfl_xor(key, static_cast<int>(step));
                pattern |= __shfl_xor(pattern, static_cast<int>(step));
            }
            if (valid && part == 0 && collapse_key) {
                collapse_key[mat_row_off[m] + i0 + row] = collapseSortKey(m, key, mx);
                collapse_row[mat_row_off[m] + i0 + row] = static_cast<uint32_t>(mat_row_off[m] + i0 + row);
                collapse_mask[mat_row_off[m] + i0 + row] = pattern;
            }
        } else {
            if (valid) {
                for (uint32_t g = part; g < G; g += parts) mx = fmax(mx, tile[g * Rs + row]);
            }
            for (uint32_t step = 1; step < parts; step *= 2) mx = fmax(mx, __shfl_xor(mx, static_cast<int>(step)));
        }
        if (valid && part == 0) rowmax[mat_row_off[m] + i0 + row] = mx;
    }
    __syncthreads();
    double * M = values + mat_val_off[m];
    const uint32_t shift = 31 - __clz(Rc);
    for (uint32_t idx = threadIdx.x; idx < cells; idx += blockDim.x) {
        const uint32_t g = idx >> shift, tt = idx & (Rc - 1);
        if (tt < nrows) M[static_cast<uint64_t>(g) * R + i0 + tt] = tile[g * Rs + tt];
    }
}

// ---- path -> groups incidence, inverted on the device ---------------------------------
// The caller gives, per matrix, the paths of every column (group).  The build kernel needs the
// opposite: the columns of every path.  One thread per column counts / scatters; the per-path lists
// come out in arbitrary column order, which does not matter (a row adds each of its entries to every
// column of the entry's path; the order of those columns does not change any sum).

__device__ __forceinline__ uint32_t matrixOfColumn(const uint64_t * __restrict__ group_off, const uint32_t num_matrices,
                                                   const uint64_t column) {
    uint32_t lo = 0, hi = num_matrices - 1;
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (group_off[mid] <= column) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__global__ void incidenceCountKernel(const uint32_t num_matrices, const uint64_t num_columns,
                                     const uint64_t * __restrict__ group_off, const uint64_t * __restrict__ group_path_off,
                                     const uint32_t * __restrict__ group_path, const uint64_t * __restrict__ inc_off,
                                     const uint64_t * __restrict__ num_paths, uint32_t * __restrict__ degree,
                                     uint32_t * __restrict__ error_flag) {
    const uint64_t column = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (column >= num_columns) return;
    const uint32_t m = matrixOfColumn(group_off, num_matrices, column);
    for (uint64_t x = group_path_off[column]; x < group_path_off[column + 1]; ++x) {
        const uint32_t p = group_path[x];
        if (p >= num_paths[m]) {
            *error_flag = 1;
            continue;
        }
        atomicAdd(&degree[inc_off[m] + p], 1u);
    }
}

__global__ void incidenceFillKernel(const uint32_t num_matrices, const uint64_t num_columns,
                                    const uint64_t * __restrict__ group_off, const uint64_t * __restrict__ group_path_off,
                                    const uint32_t * __restrict__ group_path, const uint64_t * __restrict__ inc_off,
                                    const uint64_t * __restrict__ num_paths, const uint64_t * __restrict__ path_grp_off,
                                    uint32_t * __restrict__ cursor, uint32_t * __restrict__ path_grp) {
    const uint64_t column = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (column >= num_columns) return;
    const uint32_t m = matrixOfColumn(group_off, num_matrices, column);
    const uint32_t local = static_cast<uint32_t>(column - group_off[m]);
    for (uint64_t x = group_path_off[column]; x < group_path_off[column + 1]; ++x) {
        const uint32_t p = group_path[x];
        if (p >= num_paths[m]) continue;
        const uint64_t slot = inc_off[m] + p;
        path_grp[path_grp_off[slot] + atomicAdd(&cursor[slot], 1u)] = local;
    }
}

// ---- the same matrices from the columns' path sets as bit masks (round 4) -----------------------------------------------
//
// groupsBuildTileKernel walks, for every entry of a row, a chain of dependent loads (entry -> path -> the path's list of
// columns -> column) and adds into an LDS tile that holds 64 rows of 64 columns: a wave of the workgroup's four works, four
// workgroups fit a CU, and 23 us per tile is what the chain takes (0.95 ms per configs[2] batch, a third of the HBM time of
// the 1.5 GB it writes).  The incidence the other way round needs no lists: a column IS a set of paths, and a cluster has tens
// of paths — one or a few 64-bit words per column.  A lane owns a row, a wave a quarter of (up to 64) columns: the lane loads
// its row's entries once and, for each of its columns, adds the probabilities of the entries whose path is in the column's
// set, entry after entry (the additions of groupsBuildKernel in the same order: bit-identical values).  No tile: the values
// stay in registers until they leave, 64 consecutive rows of a column per store.  Row sum, normalisation, largest value,
// sort key and zero pattern as before (the sum over the columns in ascending order, handed from wave to wave).
constexpr uint32_t kMaskMaxColumns = 1024;  // (wider matrices, and sets of more than kMaskMaxWords words: the list kernels)
constexpr uint32_t kMaskMaxWords = 16;
constexpr uint32_t kMaskRows = 64;

// one thread per column: the set of its paths
__global__ void columnMaskKernel(const uint32_t num_matrices, const uint64_t num_columns, const uint64_t * __restrict__ group_off,
                                 const uint64_t * __restrict__ group_path_off, const uint32_t * __restrict__ group_path,
                                 const uint64_t * __restrict__ num_paths, const uint64_t * __restrict__ mask_off, uint64_t * __restrict__ masks,
                                 uint32_t * __restrict__ error_flag) {
    const uint64_t column = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (column >= num_columns) return;
    const uint32_t m = matrixOfColumn(group_off, num_matrices, column);
    if (mask_off[m] == ~0ull) return;  // (a matrix of the other kernels)
    const uint32_t words = static_cast<uint32_t>((num_paths[m] + 63) / 64);
    uint64_t * const out = masks + mask_off[m] + (column - group_off[m]) * words;
    for (uint32_t k = 0; k < words; ++k) out[k] = 0;
    for (uint64_t x = group_path_off[column]; x < group_path_off[column + 1]; ++x) {
        const uint32_t p = group_path[x];
        if (p >= num_paths[m]) {
            *error_flag = 1;
            continue;
        }
        const uint64_t bit = 1ull << (p & 63u);
        if (out[p >> 6] & bit) *error_flag = 2;  // (the lists of the other kernels would add such a path twice)
        out[p >> 6] |= bit;
    }
}

struct MaskBuildArgs {
    uint32_t num_items;
    const uint32_t * item_matrix;
    const uint32_t * item_chunk;
    const uint64_t * mat_val_off;
    const uint64_t * mat_row_off;
    const uint64_t * mat_row0;
    const uint64_t * mat_rows;
    const uint32_t * mat_cols;
    const uint64_t * num_paths;
    const uint64_t * mask_off;
    const uint64_t * masks;
    const uint64_t * row_ent_off;
    const uint32_t * ent_path;
    const double * ent_prob;
    const double * row_noise;
    const uint32_t * row_perm;
    int normalise;
#ifdef RPVG_HIP_EXPERIMENTS
    int debug_no_long_rows;    // timing experiment (RPVG_HIP_BUILD_DEBUG=1): entries past the held ones are dropped
#endif
    double * values;
    double * rowmax;
    uint64_t * collapse_key;   // null: no row collapse
    uint32_t * collapse_row;
    uint64_t * collapse_mask;
};

// A wave per (matrix, 64 rows) item, a lane per row, and nothing shared between lanes: the lane walks the columns of its row
// twice — once for the row sum (column after column: the order of the other kernels), once more for the values, which it
// computes again rather than keeping them (a cell is a handful of bit tests and additions; a tile of 64 x 64 values in LDS was
// what limited a CU to four workgroups, and the kernel's time went with the workgroups a CU held: 0.97 ms with four, 1.31 with
// three, 2.35 with two — latency of its loads and of its short dependent loops, not bandwidth).  No LDS, no barrier; the set of
// a column is the same word for every lane (a scalar load).
constexpr int kMaskHeld = 8;   // entries of a row in registers (3.3 a row on the configs[2] batch), in two halves

__global__ __launch_bounds__(256) void groupsBuildMaskKernel(const MaskBuildArgs a) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t item = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (item >= a.num_items) return;
    const uint32_t m = a.item_matrix[item];
    const uint64_t R = a.mat_rows[m], r0 = a.mat_row0[m], row_off = a.mat_row_off[m];
    const uint32_t G = a.mat_cols[m];
    const uint32_t words = static_cast<uint32_t>((a.num_paths[m] + 63) / 64);
    const uint64_t * const sets = a.masks + a.mask_off[m];
    double * const M = a.values + a.mat_val_off[m];
    const uint64_t i = static_cast<uint64_t>(a.item_chunk[item]) * kMaskRows + lane;
    const bool valid = i < R;
    const uint64_t r = valid ? r0 + a.row_perm[row_off + i] : 0;
    const uint64_t e_begin = valid ? a.row_ent_off[r] : 0, e_end = valid ? a.row_ent_off[r + 1] : 0;
    const double noise = valid && a.normalise ? a.row_noise[r] : 0.0;
    uint32_t p[kMaskHeld];
    double v[kMaskHeld];
#pragma unroll
    for (int k = 0; k < kMaskHeld; ++k) {
        p[k] = 0u;
        v[k] = 0.0;  // (an entry past the row's last adds + 0.0: nothing)
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (e_begin + k < e_end) {
            p[k] = a.ent_path[e_begin + k];
            v[k] = a.ent_prob[e_begin + k];
        }
    }
    const bool second_half = __ballot(e_begin + 4 < e_end) != 0ull;  // (of the wave: the same for every lane)
    if (second_half) {
#pragma unroll
        for (int k = 4; k < kMaskHeld; ++k) {
            if (e_begin + k < e_end) {
                p[k] = a.ent_path[e_begin + k];
                v[k] = a.ent_prob[e_begin + k];
            }
        }
    }
#ifdef RPVG_HIP_EXPERIMENTS
    const bool longer = !a.debug_no_long_rows && __ballot(e_begin + kMaskHeld < e_end) != 0ull;
#else
    const bool longer = __ballot(e_begin + kMaskHeld < e_end) != 0ull;
#endif

    // the row's value in column c: the probabilities of its entries whose path is in the column's set, entry after entry
    auto cell = [&](const uint32_t c) {
        double sum = 0.0;
        if (words == 1) {
            const uint64_t set = sets[c];
#pragma unroll
            for (int k = 0; k < 4; ++k) sum += ((set >> (p[k] & 63u)) & 1ull) ? v[k] : 0.0;
            if (second_half) {
#pragma unroll
                for (int k = 4; k < kMaskHeld; ++k) sum += ((set >> (p[k] & 63u)) & 1ull) ? v[k] : 0.0;
            }
            if (longer) {
                for (uint64_t e = e_begin + kMaskHeld; e < e_end; ++e) sum += ((set >> (a.ent_path[e] & 63u)) & 1ull) ? a.ent_prob[e] : 0.0;
            }
        } else {
            const uint64_t * const set = sets + static_cast<size_t>(c) * words;
#pragma unroll
            for (int k = 0; k < 4; ++k) sum += ((set[p[k] >> 6] >> (p[k] & 63u)) & 1ull) ? v[k] : 0.0;
            if (second_half) {
#pragma unroll
                for (int k = 4; k < kMaskHeld; ++k) sum += ((set[p[k] >> 6] >> (p[k] & 63u)) & 1ull) ? v[k] : 0.0;
            }
            if (longer) {
                for (uint64_t e = e_begin + kMaskHeld; e < e_end; ++e) {
                    const uint32_t path = a.ent_path[e];
                    sum += ((set[path >> 6] >> (path & 63u)) & 1ull) ? a.ent_prob[e] : 0.0;
                }
            }
        }
        return sum;
    };

    double rowsum = 0.0;
    if (a.normalise) {
#pragma unroll 4
        for (uint32_t c = 0; c < G; ++c) rowsum += cell(c);
    }
    const double keep = 1 - noise;
    double key = collapseWeight(G) * noise, mx = 0.0;
    uint64_t pattern = 0;
#pragma unroll 4
    for (uint32_t c = 0; c < G; ++c) {
        double value = cell(c);
        if (a.normalise) {
            value = (value / rowsum) * keep;
            if (value != value) value = 0.0;  // 0/0 rows -> 0 (src/path_estimator.cpp:162)
            key = fma(collapseWeight(c), value, key);
            if (c < 64 && value != 0.0) pattern |= 1ull << c;
        }
        mx = fmax(mx, value);
        if (valid) M[static_cast<uint64_t>(c) * R + i] = value;
    }
    if (valid) {
        a.rowmax[row_off + i] = mx;
        if (a.normalise && a.collapse_key) {
            a.collapse_key[row_off + i] = collapseSortKey(m, key, mx);
            a.collapse_row[row_off + i] = static_cast<uint32_t>(row_off + i);
            a.collapse_mask[row_off + i] = pattern;
        }
    }
}

// ---- every column one path of its own, values as they are (rpvg_hip_groups_build_single_paths without normalisation: the raw path
// posteriors of configs[4]) ------------------------------------------------------------------------------------------------------
// The general kernels ask, cell by cell, which entries of the row lie in the column's set: columns x entries bit tests per row, for
// matrices of up to thousands of columns (groupsBuildMaskKernel: 5.1 ms per lane of a configs[4] batch, a sixth of the batch's kernel
// time).  Here a cell is an entry or nothing: the lane zeroes its row column after column (the lanes of a wave write 64 neighbouring
// doubles of a column) and stores its entries where they belong — the same values, the same row maxima.  A wave per (matrix, 64 rows).
__global__ __launch_bounds__(256) void groupsBuildSinglePathKernel(const MaskBuildArgs a) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t item = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (item >= a.num_items) return;
    const uint32_t m = a.item_matrix[item];
    const uint64_t R = a.mat_rows[m], r0 = a.mat_row0[m], row_off = a.mat_row_off[m];
    const uint32_t G = a.mat_cols[m];
    double * const M = a.values + a.mat_val_off[m];
    const uint64_t i = static_cast<uint64_t>(a.item_chunk[item]) * kMaskRows + lane;
    if (i >= R) return;
    const uint64_t r = r0 + a.row_perm[row_off + i];
    for (uint32_t c = 0; c < G; ++c) M[static_cast<uint64_t>(c) * R + i] = 0.0;
    double mx = 0.0;
    for (uint64_t e = a.row_ent_off[r]; e < a.row_ent_off[r + 1]; ++e) {  // (behind the zeros of the same lane)
        const double value = a.ent_prob[e];
        M[static_cast<uint64_t>(a.ent_path[e]) * R + i] = value;
        mx = fmax(mx, value);
    }
    a.rowmax[row_off + i] = mx;
}

// ---- matrices of up to 64 columns: the columns of a path as ONE word (round 5) ---------------------------------------------
//
// With several batches in flight the GPU is busy throughout, and groupsBuildMaskKernel is a quarter of what its SIMDs issue per
// configs[2] batch: a cell costs a test of every held entry of the row against the column's set — twice, once for the row sum and
// once for the value —, four or eight entries whatever the row has, and a set is a load per cell.  The matrices of that batch
// have up to 64 columns (99 % of its cells), few enough for the incidence the other way round to be a word per PATH: the columns
// that contain it.  A lane (= a row) then takes its entries one after the other — as many as the wave's longest row has — and
// adds each entry's probability to the cells whose bit is set in the entry's word, all 64 cells in registers: a bit field
// extract, two ANDs and an addition per entry and cell (the additions of a cell are those of the other kernels in the same
// order: entry after entry; an entry outside the column adds + 0.0), no load after the entries', and the cells are still there
// when the row sum is known.
constexpr uint32_t kWordMaxColumns = 64;
constexpr int kWordHeld = 4;   // entries loaded ahead of the loop (3.3 a row on the configs[2] batch; the wave's longest row: 4.4)

// one thread per column: its bit into the word of every path it contains (words zeroed before)
__global__ void pathColumnWordKernel(const uint32_t num_matrices, const uint64_t num_columns, const uint64_t * __restrict__ group_off,
                                     const uint64_t * __restrict__ group_path_off, const uint32_t * __restrict__ group_path,
                                     const uint64_t * __restrict__ num_paths, const uint64_t * __restrict__ word_off,
                                     unsigned long long * __restrict__ words, uint32_t * __restrict__ error_flag) {
    const uint64_t column = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (column >= num_columns) return;
    const uint32_t m = matrixOfColumn(group_off, num_matrices, column);
    if (word_off[m] == ~0ull) return;  // (a matrix of the other kernels)
    const unsigned long long bit = 1ull << (column - group_off[m]);
    for (uint64_t x = group_path_off[column]; x < group_path_off[column + 1]; ++x) {
        const uint32_t p = group_path[x];
        if (p >= num_paths[m]) {
            *error_flag = 1;
            continue;
        }
        if (atomicOr(&words[word_off[m] + p], bit) & bit) *error_flag = 2;  // (the lists of the other kernels would add such a path twice)
    }
}

// cells[c] += the entry's probability where bit c of its word is set
__device__ __forceinline__ void addEntryToCells(double (&cells)[kWordMaxColumns], const uint32_t blocks, const uint64_t word, const double prob) {
    const int lo = static_cast<int>(static_cast<uint32_t>(word)), hi = static_cast<int>(static_cast<uint32_t>(word >> 32));
    const int prob_lo = __double2loint(prob), prob_hi = __double2hiint(prob);
#pragma unroll
    for (uint32_t b = 0; b < kWordMaxColumns / 8; ++b) {
        if (b < blocks) {  // (of the wave; a column past the matrix's last has no bit anywhere)
#pragma unroll
            for (uint32_t j = 0; j < 8; ++j) {
                const uint32_t c = 8 * b + j;
                const int all = __builtin_amdgcn_sbfe(c < 32 ? lo : hi, c & 31u, 1);  // 0 or ~0
                cells[c] += __hiloint2double(prob_hi & all, prob_lo & all);
            }
        }
    }
}

// A wave per (matrix, 64 rows) item, a lane per row, as groupsBuildMaskKernel; MaskBuildArgs with `masks` = the words of the
// paths and `mask_off` = the first word of a matrix.
__global__ __launch_bounds__(256) void groupsBuildWordKernel(const MaskBuildArgs a) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t item = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (item >= a.num_items) return;
    const uint32_t m = a.item_matrix[item];
    const uint64_t R = a.mat_rows[m], r0 = a.mat_row0[m], row_off = a.mat_row_off[m];
    const uint32_t G = a.mat_cols[m], blocks = (G + 7) / 8;
    const uint64_t * const words = a.masks + a.mask_off[m];
    double * const M = a.values + a.mat_val_off[m];
    const uint64_t i = static_cast<uint64_t>(a.item_chunk[item]) * kMaskRows + lane;
    const bool valid = i < R;
    const uint64_t r = valid ? r0 + a.row_perm[row_off + i] : 0;
    const uint64_t e_begin = valid ? a.row_ent_off[r] : 0, e_end = valid ? a.row_ent_off[r + 1] : 0;
    const double noise = valid && a.normalise ? a.row_noise[r] : 0.0;
    const uint32_t n = static_cast<uint32_t>(e_end - e_begin);
    uint32_t longest = n;
#pragma unroll
    for (int offset = 32; offset > 0; offset >>= 1) {
        const uint32_t other = __shfl_xor(longest, offset);
        longest = other > longest ? other : longest;
    }
    longest = __builtin_amdgcn_readfirstlane(longest);

    uint64_t word[kWordHeld];
    double prob[kWordHeld];
    bool tiny_entry = false;  // a non-zero entry below 2^-400 (or not a number): see the quotients below
#pragma unroll
    for (int k = 0; k < kWordHeld; ++k) {
        word[k] = 0;   // (an entry past the row's last: no column)
        prob[k] = 0.0;
        if (static_cast<uint32_t>(k) < n) {
            word[k] = words[a.ent_path[e_begin + k]];
            prob[k] = a.ent_prob[e_begin + k];
            tiny_entry |= !(prob[k] == 0.0 || prob[k] >= 0x1p-400);
        }
    }
    double cells[kWordMaxColumns];
#pragma unroll
    for (uint32_t c = 0; c < kWordMaxColumns; ++c) cells[c] = 0.0;
#pragma unroll
    for (int k = 0; k < kWordHeld; ++k) {
        if (static_cast<uint32_t>(k) < longest) addEntryToCells(cells, blocks, word[k], prob[k]);
    }
    for (uint32_t k = kWordHeld; k < longest; ++k) {
        uint64_t w = 0;
        double v = 0.0;
        if (k < n) {
            w = words[a.ent_path[e_begin + k]];
            v = a.ent_prob[e_begin + k];
            tiny_entry |= !(v == 0.0 || v >= 0x1p-400);
        }
        addEntryToCells(cells, blocks, w, v);
    }

    double rowsum = 0.0;
    if (a.normalise) {
#pragma unroll
        for (uint32_t b = 0; b < kWordMaxColumns / 8; ++b) {
            if (b < blocks) {
#pragma unroll
                for (uint32_t j = 0; j < 8; ++j) rowsum += cells[8 * b + j];  // (in column order; a column past the last adds + 0.0)
            }
        }
    }
    // The quotients cell / rowsum.  The division the compiler emits is eleven instructions a cell, six of them on the
    // denominator alone: its reciprocal (v_rcp_f64, two Newton steps) — once per row here —, then quotient = cell x reciprocal,
    // remainder = cell - rowsum x quotient, quotient + remainder x reciprocal: the hardware's own sequence, which differs from
    // this one by the scaling of operands near the ends of the exponent range (v_div_scale / v_div_fmas) and the special cases
    // of v_div_fixup.  Neither arises while the row sum lies in [2^-400, 2^400] and every non-zero cell is at least 2^-400 (a
    // cell is a sum of the row's non-negative entries: at least its smallest non-zero one, at most the row sum): such waves —
    // all of them, in practice — take the three instructions, the others the division.  0 / rowsum is 0 either way, and a row
    // sum of 0 (all cells 0) gives not-a-number, hence 0, either way.
    const double keep = 1 - noise;
    constexpr double kSmall = 0x1p-400, kLarge = 0x1p400;
    const bool plain = __ballot(valid && a.normalise && (!(rowsum == 0.0 || (rowsum >= kSmall && rowsum <= kLarge)) || tiny_entry)) == 0ull;
    double reciprocal = 0.0;
    if (plain) {
        reciprocal = __builtin_amdgcn_rcp(rowsum);
        reciprocal = fma(reciprocal, fma(-rowsum, reciprocal, 1.0), reciprocal);
        reciprocal = fma(reciprocal, fma(-rowsum, reciprocal, 1.0), reciprocal);
    }
    double key = collapseWeight(G) * noise, mx = 0.0;
    uint32_t pattern_lo = 0, pattern_hi = 0;
    double * out = M + i;
    // eight columns at a time: without a branch between them their quotients overlap
    auto columns = [&](const uint32_t b, auto all_eight, auto plain_division) {
#pragma unroll
        for (uint32_t j = 0; j < 8; ++j) {
            const uint32_t c = 8 * b + j;
            if (decltype(all_eight)::value || c < G) {
                double value = cells[c];
                if (a.normalise) {
                    if (decltype(plain_division)::value) {
                        const double quotient = value * reciprocal;
                        value = fma(fma(-rowsum, quotient, value), reciprocal, quotient);
                    } else {
                        value = value / rowsum;
                    }
                    value *= keep;
                    if (value != value) value = 0.0;  // 0/0 rows -> 0 (src/path_estimator.cpp:162)
                    key = fma(collapseWeight(c), value, key);
                    if (c < 32) pattern_lo |= value != 0.0 ? 1u << (c & 31u) : 0u;
                    else pattern_hi |= value != 0.0 ? 1u << (c & 31u) : 0u;
                }
                asm("v_max_f64 %0, %1, %2" : "=v"(mx) : "v"(mx), "v"(value));  // (no NaN here: fmax() would quiet both operands first)
                if (valid) out[static_cast<uint64_t>(j) * R] = value;
            }
        }
        out += 8 * R;
    };
#pragma unroll
    for (uint32_t b = 0; b < kWordMaxColumns / 8; ++b) {
        if (8 * b + 8 <= G) {
            if (plain) columns(b, std::true_type(), std::true_type()); else columns(b, std::true_type(), std::false_type());
        } else if (8 * b < G) {
            if (plain) columns(b, std::false_type(), std::true_type()); else columns(b, std::false_type(), std::false_type());
        }
    }
    if (valid) {
        a.rowmax[row_off + i] = mx;
        if (a.normalise && a.collapse_key) {
            a.collapse_key[row_off + i] = collapseSortKey(m, key, mx);
            a.collapse_row[row_off + i] = static_cast<uint32_t>(row_off + i);
            a.collapse_mask[row_off + i] = (static_cast<uint64_t>(pattern_hi) << 32) | pattern_lo;
        }
    }
}

// storage of the matrices groupsBuildKernel accumulates in global memory: one work item = 256 rows of one matrix
__global__ __launch_bounds__(256) void zeroWideMatricesKernel(const uint32_t * __restrict__ item_matrix, const uint32_t * __restrict__ item_chunk,
                                                              const uint64_t * __restrict__ mat_val_off, const uint64_t * __restrict__ mat_rows,
                                                              const uint32_t * __restrict__ mat_cols, double * __restrict__ values) {
    const uint32_t m = item_matrix[blockIdx.x];
    const uint64_t R = mat_rows[m];
    const uint64_t row = static_cast<uint64_t>(item_chunk[blockIdx.x]) * 256 + threadIdx.x;
    if (row >= R) return;
    double * M = values + mat_val_off[m];
    const uint32_t G = mat_cols[m];
    for (uint32_t g = 0; g < G; ++g) M[static_cast<uint64_t>(g) * R + row] = 0.0;
}

// one wave per request
template <int WIDTH>
__global__ __launch_bounds__(256) void groupLoglikKernel(
    const uint32_t num_requests, const uint32_t * __restrict__ req_matrix, const uint32_t * __restrict__ req_members,
    const uint8_t * __restrict__ req_rowmax, const double divisor, const uint64_t * __restrict__ mat_val_off,
    const uint64_t * __restrict__ mat_row_off, const uint32_t * __restrict__ mat_fast, const uint32_t * __restrict__ mat_mid,
    const uint64_t * __restrict__ mat_rows, const double * __restrict__ values, const double * __restrict__ rowmax,
    const double * __restrict__ row_count, const double * __restrict__ row_noise, double * __restrict__ out) {
    __shared__ LogTableEntry lt[kLogTableSize];
    loadLogTable(lt);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint32_t q = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (q >= num_requests) return;
    const uint32_t m = req_matrix[q];
    const uint64_t R = mat_rows[m];
    const double * M = values + mat_val_off[m];
    const double * cnt = row_count + mat_row_off[m];
    const double * nz = row_noise + mat_row_off[m];
    const double * rm = req_rowmax && req_rowmax[q] ? rowmax + mat_row_off[m] : nullptr;
    const double * col[WIDTH];
#pragma unroll
    for (int w = 0; w < WIDTH; ++w) {
        const uint32_t g = req_members[static_cast<uint64_t>(q) * WIDTH + w];
        col[w] = (g == kNoMember) ? nullptr : M + static_cast<uint64_t>(g) * R;
    }
    auto x = [&](const uint64_t i) {
        double v = nz[i];
#pragma unroll
        for (int w = 0; w < WIDTH; ++w)
            if (col[w]) v += col[w][i] / divisor;
        if (rm) v += rm[i] / divisor;
        return v;
    };
    const double acc = waveSumF64(sumCountLogs<uint64_t>(lt, cnt, x, 0, mat_fast[m], mat_mid[m], R, lane));
    if (lane == 0) out[q] = acc;
}

// Conditionals of the Gibbs sampler (src/path_estimator.cpp:527-545): request q fixes the other WIDTH-1 members of a
// group on its matrix and asks for the log-likelihood of every candidate column.  One wave per (request, 4 candidate
// columns): the base vector noise + sum(others)/g is formed once per row and shared by the four logs (same order
// of additions as the reference: noise, the others in slot order, then the candidate).
template <int WIDTH>
__global__ __launch_bounds__(256) void groupConditionalKernel(
    const uint32_t num_requests, const uint64_t num_items, const uint64_t * __restrict__ item_off,
    const uint64_t * __restrict__ out_off, const uint32_t * __restrict__ req_matrix,
    const uint32_t * __restrict__ req_others, const double divisor, const uint64_t * __restrict__ mat_val_off,
    const uint64_t * __restrict__ mat_row_off, const uint32_t * __restrict__ mat_fast, const uint32_t * __restrict__ mat_mid,
    const uint64_t * __restrict__ mat_rows, const uint32_t * __restrict__ mat_cols, const double * __restrict__ values, const double * __restrict__ row_count,
    const double * __restrict__ row_noise, double * __restrict__ out) {
    constexpr int kCand = 4;
    __shared__ LogTableEntry lt[kLogTableSize];
    loadLogTable(lt);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint64_t item = (blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x) >> 6;
    if (item >= num_items) return;
    uint32_t lo = 0, hi = num_requests - 1;  // last q with item_off[q] <= item
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (item_off[mid] <= item) lo = mid; else hi = mid - 1;
    }
    const uint32_t q = lo;
    const uint32_t m = req_matrix[q];
    const uint64_t R = mat_rows[m];
    const uint32_t G = mat_cols[m];
    const uint32_t k0 = static_cast<uint32_t>(item - item_off[q]) * kCand;
    const double * M = values + mat_val_off[m];
    const double * cnt = row_count + mat_row_off[m];
    const double * nz = row_noise + mat_row_off[m];
    const double * other[WIDTH > 1 ? WIDTH - 1 : 1];
#pragma unroll
    for (int w = 0; w + 1 < WIDTH; ++w) other[w] = M + static_cast<uint64_t>(req_others[static_cast<uint64_t>(q) * (WIDTH - 1) + w]) * R;
    const double * cand[kCand];
#pragma unroll
    for (int c = 0; c < kCand; ++c) cand[c] = M + static_cast<uint64_t>(min(k0 + c, G - 1)) * R;
    double acc[kCand] = {0.0, 0.0, 0.0, 0.0};
    LogProduct pr[kCand];
    const uint64_t fast_end = mat_fast[m], mid_end = mat_mid[m];
    auto x = [&](const uint64_t i, double (&xs)[kCand]) {
        double base = nz[i];
#pragma unroll
        for (int w = 0; w + 1 < WIDTH; ++w) base += other[w][i] / divisor;
#pragma unroll
        for (int c = 0; c < kCand; ++c) xs[c] = base + cand[c][i] / divisor;
    };
    sumCountLogsMulti<kCand, 64, uint64_t>(lt, cnt, x, 0, fast_end, mid_end, R, lane, pr, acc);
    if (mid_end) {
#pragma unroll
        for (int c = 0; c < kCand; ++c) acc[c] += pr[c].value(lt);
    }
#pragma unroll
    for (int c = 0; c < kCand; ++c) {
        const double total = waveSumF64(acc[c]);
        if (lane == 0 && k0 + c < G) out[out_off[q] + k0 + c] = total;
    }
}

__global__ void debugLogKernel(const uint64_t n, const double * __restrict__ x, double * __restrict__ out, const int use_table) {
    __shared__ LogTableEntry lt[kLogTableSize];
    loadLogTable(lt);
    __syncthreads();
    const uint64_t i = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (i < n) out[i] = use_table ? logPositive(x[i], lt) : logPositive(x[i]);
}

}  // namespace

extern "C" int rpvg_hip_debug_log(rpvg_hip_ctx * ctx, uint64_t n, const double * x, double * out, int32_t use_table) {
    RPVG_REQUIRE(ctx && x && out, "rpvg_hip_debug_log: NULL argument");
    if (n == 0) return RPVG_HIP_OK;
    std::lock_guard<std::mutex> lock(ctx->mutex);
    RPVG_HIP_CHECK(hipSetDevice(ctx->device));
    DeviceBuffer<double> d_x, d_out;
    RPVG_HIP_CHECK(d_x.upload(x, n, ctx->stream));
    RPVG_HIP_CHECK(d_out.alloc(n));
    debugLogKernel<<<dim3(static_cast<uint32_t>((n + 255) / 256)), dim3(256), 0, ctx->stream>>>(n, d_x.ptr, d_out.ptr, use_table);
    RPVG_HIP_CHECK(hipGetLastError());
    RPVG_HIP_CHECK(d_out.download(out, ctx->stream));
    RPVG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return RPVG_HIP_OK;
}

// The haplotype columns of the batch (path_sources.hip: laid out by bounds, per cluster) for the clusters of a build, as the
// compact lists the build kernels take.  One workgroup per matrix.
__global__ __launch_bounds__(256) void gatherSourceColumnsKernel(const uint32_t num_matrices, const uint32_t * __restrict__ cluster,
                                                                 const uint64_t * __restrict__ cluster_src_off, const uint32_t * __restrict__ src_col_count,
                                                                 const uint32_t * __restrict__ src_col_end, const uint32_t * __restrict__ src_col_path,
                                                                 const uint64_t * __restrict__ group_off, const uint64_t * __restrict__ first_path,
                                                                 uint64_t * __restrict__ group_path_off, uint32_t * __restrict__ group_path,
                                                                 uint32_t * __restrict__ column_counts) {
    const uint32_t m = blockIdx.x;
    if (m >= num_matrices) return;
    const uint64_t slot0 = cluster_src_off[cluster[m]];
    const uint64_t g0 = group_off[m], p0 = first_path[m];
    const uint32_t G = static_cast<uint32_t>(group_off[m + 1] - g0), L = static_cast<uint32_t>(first_path[m + 1] - p0);
    if (threadIdx.x == 0 && m == 0) group_path_off[0] = 0;
    for (uint32_t c = threadIdx.x; c < G; c += 256) {
        group_path_off[g0 + c + 1] = p0 + src_col_end[slot0 + c];
        column_counts[g0 + c] = src_col_count[slot0 + c];
    }
    for (uint32_t x = threadIdx.x; x < L; x += 256) group_path[p0 + x] = src_col_path[slot0 + x];
}

// column c of a matrix = path c of its cluster, alone (rpvg_hip_groups_build_single_paths)
__global__ __launch_bounds__(256) void singlePathColumnsKernel(const uint32_t num_matrices, const uint64_t * __restrict__ group_off,
                                                               uint64_t * __restrict__ group_path_off, uint32_t * __restrict__ group_path) {
    const uint32_t m = blockIdx.x;
    if (m >= num_matrices) return;
    const uint64_t g0 = group_off[m];
    const uint32_t G = static_cast<uint32_t>(group_off[m + 1] - g0);
    if (threadIdx.x == 0 && m == 0) group_path_off[0] = 0;
    for (uint32_t c = threadIdx.x; c < G; c += 256) {
        group_path_off[g0 + c + 1] = g0 + c + 1;
        group_path[g0 + c] = c;
    }
}

static int buildGroups(rpvg_hip_ctx * ctx, const rpvg_hip_batch * batch, const rpvg_hip_group_spec * spec, bool from_sources,
                       rpvg_hip_groups ** groups_out, bool single_paths = false);

extern "C" int rpvg_hip_groups_build(rpvg_hip_ctx * ctx, const rpvg_hip_batch * batch, const rpvg_hip_group_spec * spec,
                                     rpvg_hip_groups ** groups_out) {
    RPVG_REQUIRE(ctx && batch && spec && groups_out, "rpvg_hip_groups_build: NULL argument");
    *groups_out = nullptr;
    RPVG_REQUIRE(spec->num_matrices == 0 || (spec->cluster && spec->group_off && spec->group_path_off && spec->group_path),
                 "rpvg_hip_groups_build: NULL spec arrays");
    return buildGroups(ctx, batch, spec, false, groups_out);
}

extern "C" int rpvg_hip_groups_build_from_sources(rpvg_hip_ctx * ctx, const rpvg_hip_batch * batch, uint32_t num_matrices, const uint32_t * clusters,
                                                  int32_t normalise, double collapse_precision, rpvg_hip_groups ** groups_out) {
    RPVG_REQUIRE(ctx && batch && groups_out && (clusters || num_matrices == 0), "rpvg_hip_groups_build_from_sources: NULL argument");
    *groups_out = nullptr;
    if (!batch->has_source_columns) {
        setError("rpvg_hip_groups_build_from_sources: not taken (the batch was uploaded without PathInfo::source_ids, or their id ranges "
                 "outgrew the device scratch)");
        return RPVG_HIP_ERR_UNSUPPORTED;
    }
    for (uint32_t m = 0; m < num_matrices; ++m) {
        RPVG_REQUIRE(clusters[m] < batch->num_clusters, "rpvg_hip_groups_build_from_sources: matrix %u refers to cluster %u of %u", m, clusters[m],
                     batch->num_clusters);
        RPVG_REQUIRE(batch->h_src_num_cols[clusters[m]] > 0, "rpvg_hip_groups_build_from_sources: cluster %u has no source (haplotype) ids on its paths",
                     clusters[m]);
    }
    // the columns of matrix m are those of its cluster: the offsets are sums of sizes the upload brought back
    std::vector<uint64_t> group_off(num_matrices + 1, 0), first_path(num_matrices + 1, 0);
    for (uint32_t m = 0; m < num_matrices; ++m) {
        group_off[m + 1] = group_off[m] + batch->h_src_num_cols[clusters[m]];
        first_path[m + 1] = first_path[m] + batch->h_src_col_paths[clusters[m]];
    }
    rpvg_hip_group_spec spec;
    spec.num_matrices = num_matrices;
    spec.cluster = clusters;
    spec.group_off = group_off.data();
    spec.group_path_off = first_path.data();  // (from_sources: per MATRIX, the first list entry of its columns)
    spec.group_path = nullptr;
    spec.normalise = normalise;
    spec.collapse_precision = collapse_precision;
    return buildGroups(ctx, batch, &spec, true, groups_out);
}

extern "C" int rpvg_hip_groups_build_single_paths(rpvg_hip_ctx * ctx, const rpvg_hip_batch * batch, uint32_t num_matrices, const uint32_t * clusters,
                                                  int32_t normalise, double collapse_precision, rpvg_hip_groups ** groups_out) {
    RPVG_REQUIRE(ctx && batch && groups_out && (clusters || num_matrices == 0), "rpvg_hip_groups_build_single_paths: NULL argument");
    *groups_out = nullptr;
    std::vector<uint64_t> group_off(num_matrices + 1, 0);
    for (uint32_t m = 0; m < num_matrices; ++m) {
        RPVG_REQUIRE(clusters[m] < batch->num_clusters, "rpvg_hip_groups_build_single_paths: matrix %u refers to cluster %u of %u", m, clusters[m],
                     batch->num_clusters);
        group_off[m + 1] = group_off[m] + (batch->h_cluster_path_off[clusters[m] + 1] - batch->h_cluster_path_off[clusters[m]]);
    }
    rpvg_hip_group_spec spec;
    spec.num_matrices = num_matrices;
    spec.cluster = clusters;
    spec.group_off = group_off.data();
    spec.group_path_off = group_off.data();  // (per MATRIX, as from the batch's columns: a column is one list entry)
    spec.group_path = nullptr;
    spec.normalise = normalise;
    spec.collapse_precision = collapse_precision;
    return buildGroups(ctx, batch, &spec, false, groups_out, true);
}

static int buildGroups(rpvg_hip_ctx * ctx, const rpvg_hip_batch * batch, const rpvg_hip_group_spec * spec, const bool from_sources,
                       rpvg_hip_groups ** groups_out, const bool single_paths) {
    const uint32_t M = spec->num_matrices;
    RPVG_REQUIRE(spec->collapse_precision >= 0 && spec->collapse_precision < 1, "rpvg_hip_groups_build: collapse_precision outside [0, 1)");
    RPVG_REQUIRE(spec->collapse_precision == 0 || spec->normalise,
                 "rpvg_hip_groups_build: the row collapse applies to normalised matrices (src/path_abundance_estimator.cpp:379-380,442-443)");
    static const bool no_collapse = RPVG_EXPERIMENT_ENV("RPVG_HIP_NO_COLLAPSE") != nullptr;  // A/B knob: matrices as built
    const bool collapse = spec->collapse_precision > 0 && !no_collapse;

    rpvg_hip_groups * g = new (std::nothrow) rpvg_hip_groups();
    if (!g) {
        setError("rpvg_hip_groups_build: out of host memory");
        return RPVG_HIP_ERR_ALLOC;
    }
    g->batch = batch;
    g->num_matrices = M;
    g->normalise = spec->normalise;

    std::unique_ptr<HostScope> scope_host(new HostScope("groups_build: host sizes"));
    // host: sizes and offsets only (O(M)); the path -> groups incidence is inverted on the device
    std::vector<uint64_t> val_off(M), row_off(M), row0(M), rows(M), inc_off(M), num_paths(M);
    std::vector<uint32_t> cols(M), item_matrix, item_chunk, tile_matrix, tile_chunk, mask_matrix, mask_chunk, word_matrix, word_chunk;
    std::vector<uint64_t> mask_off(M, ~0ull), word_off(M, ~0ull);  // (~0: a matrix of other kernels)
    uint64_t val_total = 0, row_total = 0, inc_total = 0, mask_total = 0, word_total = 0;
    bool lists_needed = false;  // the path -> columns lists of the tile / global-memory kernels
    // Which kernel builds a matrix: up to kWordMaxColumns columns groupsBuildWordKernel, up to kMaskMaxColumns
    // groupsBuildMaskKernel, the rest the list kernels.  RPVG_HIP_BUILD_MASKS=1: no word kernel, =0: the list kernels for
    // everything (the three produce the identical values: switches for the tests and for A/B timing — 0.84 (masks) against 1.05 ms
    // (lists) per configs[2] batch standing alone, but 9.71 against 9.65 ms per batch in round 4's two-lane bench: the list kernels
    // wait, and the other lane's kernels run meanwhile; with batches in flight the GPU has no such gaps).
    const char * masks_env = std::getenv("RPVG_HIP_BUILD_MASKS");
    const bool build_masks = masks_env ? std::atoi(masks_env) != 0 : true;
    const bool build_words = masks_env ? std::atoi(masks_env) >= 2 : true;
    // single-path columns, values as they are: one kernel for every width (RPVG_HIP_NO_DIRECT_BUILD=1: the general kernels — the tests take both)
    const bool direct = single_paths && !spec->normalise && !std::getenv("RPVG_HIP_NO_DIRECT_BUILD");
    for (uint32_t m = 0; m < M; ++m) {
        const uint32_t k = spec->cluster[m];
        if (k >= batch->num_clusters) {
            setError("rpvg_hip_groups_build: matrix %u refers to cluster %u of %u", m, k, batch->num_clusters);
            delete g;
            return RPVG_HIP_ERR_INVALID;
        }
        const uint64_t R = batch->h_cluster_row_off[k + 1] - batch->h_cluster_row_off[k];
        const uint64_t N = batch->h_cluster_path_off[k + 1] - batch->h_cluster_path_off[k];
        const uint64_t g0 = spec->group_off[m], g1 = spec->group_off[m + 1];
        if (R > 0xffffffffull) {
            setError("rpvg_hip_groups_build: matrix %u has %llu rows (limit 2^32 - 1)", m, static_cast<unsigned long long>(R));
            delete g;
            return RPVG_HIP_ERR_INVALID;
        }
        if (R == 0 || g1 <= g0) {
            setError("rpvg_hip_groups_build: matrix %u has no rows or no columns", m);
            delete g;
            return RPVG_HIP_ERR_INVALID;
        }
        rows[m] = R;
        cols[m] = static_cast<uint32_t>(g1 - g0);
        {
            uint64_t longest = 0;
            if (from_sources) longest = batch->h_src_max_col_paths[k];
            else if (single_paths) longest = 1;
            else for (uint64_t c = g0; c < g1; ++c) longest = std::max<uint64_t>(longest, spec->group_path_off[c + 1] - spec->group_path_off[c]);
            g->h_max_col_paths.push_back(static_cast<uint32_t>(std::min<uint64_t>(longest, 0xffffffffu)));
            g->h_num_paths.push_back(static_cast<uint32_t>(N));
            g->h_cluster.push_back(k);
        }
        row0[m] = batch->h_cluster_row_off[k];
        num_paths[m] = N;
        val_off[m] = val_total;
        row_off[m] = row_total;
        inc_off[m] = inc_total;
        val_total += R * (g1 - g0);
        row_total += R;
        inc_total += N + 1;
        const uint64_t mask_words = (N + 63) / 64;
        if (direct) {  // (groupsBuildSinglePathKernel: the items of the mask kernel, no masks)
            for (uint64_t c = 0; c * kMaskRows < R; ++c) {
                mask_matrix.push_back(m);
                mask_chunk.push_back(static_cast<uint32_t>(c));
            }
            continue;
        }
        if (build_words && cols[m] <= kWordMaxColumns) {
            word_off[m] = word_total;
            word_total += N;
            for (uint64_t c = 0; c * kMaskRows < R; ++c) {
                word_matrix.push_back(m);
                word_chunk.push_back(static_cast<uint32_t>(c));
            }
            continue;
        }
        if (build_masks && cols[m] <= kMaskMaxColumns && mask_words <= kMaskMaxWords) {
            mask_off[m] = mask_total;
            mask_total += cols[m] * mask_words;
            for (uint64_t c = 0; c * kMaskRows < R; ++c) {
                mask_matrix.push_back(m);
                mask_chunk.push_back(static_cast<uint32_t>(c));
            }
            continue;
        }
        lists_needed = true;
        const uint32_t tile_rows = tileRows(cols[m]);
        if (tile_rows) {
            for (uint64_t c = 0; c * tile_rows < R; ++c) {
                tile_matrix.push_back(m);
                tile_chunk.push_back(static_cast<uint32_t>(c));
            }
        } else {
            for (uint64_t c = 0; c * 256 < R; ++c) {
                item_matrix.push_back(m);
                item_chunk.push_back(static_cast<uint32_t>(c));
            }
        }
    }
    g->h_num_cols = cols;
    g->h_num_rows = rows;
    // the row collapse keeps the matrix of a row in 20 bits of its sort key and rows in 31 (row_collapse.hip)
    if (collapse && (M >= kCollapseMaxMatrices || row_total > 0x7fffffffull)) {
        setError("rpvg_hip_groups_build: the row collapse takes up to %u matrices and 2^31 - 1 rows per call (got %u matrices, %llu rows): "
                 "build the matrices of a batch in several calls", kCollapseMaxMatrices - 1, M, static_cast<unsigned long long>(row_total));
        delete g;
        return RPVG_HIP_ERR_INVALID;
    }
    if (M == 0) {
        *groups_out = g;
        return RPVG_HIP_OK;
    }
    const uint64_t num_columns = spec->group_off[M];
    const uint64_t num_incidences = (from_sources || single_paths) ? spec->group_path_off[M] : spec->group_path_off[num_columns];

    scope_host.reset();
    HostScope scope_dev("groups_build: upload + kernels + sync");
    std::lock_guard<std::mutex> lock(ctx->mutex);
    hipError_t e = hipSetDevice(ctx->device);
    hipStream_t st = ctx->stream;
    // temporaries: owned by the matrices object (the kernels that use them may still be queued on return)
    struct BuildTemporaries {
        DeviceBuffer<uint64_t> inc_off, path_grp_off, group_off, group_path_off, num_paths;
        DeviceBuffer<uint32_t> path_grp, item_matrix, item_chunk, tile_matrix, tile_chunk, group_path, degree, cursor, cluster, mask_matrix, mask_chunk;
        DeviceBuffer<uint64_t> mask_off, masks, first_path, word_off, path_words;
        DeviceBuffer<uint32_t> word_matrix, word_chunk;
        DeviceBuffer<uint32_t> column_counts;
        DeviceBuffer<unsigned char> scan_tmp;
    };
    std::shared_ptr<BuildTemporaries> tmp = std::make_shared<BuildTemporaries>();
    g->build_temporaries.emplace_back(tmp);
    auto & d_inc_off = tmp->inc_off; auto & d_path_grp_off = tmp->path_grp_off; auto & d_group_off = tmp->group_off;
    auto & d_group_path_off = tmp->group_path_off; auto & d_num_paths = tmp->num_paths; auto & d_path_grp = tmp->path_grp;
    auto & d_item_matrix = tmp->item_matrix; auto & d_item_chunk = tmp->item_chunk; auto & d_tile_matrix = tmp->tile_matrix;
    auto & d_tile_chunk = tmp->tile_chunk; auto & d_group_path = tmp->group_path; auto & d_degree = tmp->degree;
    auto & d_cursor = tmp->cursor; auto & d_scan_tmp = tmp->scan_tmp;
    auto & d_error = g->build_error_flag;
    auto ok = [&](hipError_t r) { if (e == hipSuccess) e = r; return e == hipSuccess; };
    std::unique_ptr<HostScope> sub(new HostScope("groups_build: uploads"));
    int span = ctx->spanBegin(FAM_H2D);
    // one block, one copy for the host arrays (UploadPack); the zero-initialised counters ride behind them (one memset)
    UploadPack & pack = g->uploads;
    pack.add(g->mat_val_off, val_off.data(), M);
    pack.add(g->mat_row_off, row_off.data(), M);
    pack.add(g->mat_row0, row0.data(), M);
    pack.add(g->mat_rows, rows.data(), M);
    pack.add(g->mat_cols, cols.data(), M);
    pack.add(d_inc_off, inc_off.data(), M);
    pack.add(d_num_paths, num_paths.data(), M);
    pack.add(d_group_off, spec->group_off, M + 1);
    if (from_sources) {
        pack.add(tmp->first_path, spec->group_path_off, M + 1);
    } else if (single_paths) {
        // (nothing: the lists are written on the device)
    } else {
        pack.add(d_group_path_off, spec->group_path_off, num_columns + 1);
        pack.add(d_group_path, spec->group_path, num_incidences);
    }
    if (!item_matrix.empty()) {
        pack.add(d_item_matrix, item_matrix.data(), item_matrix.size());
        pack.add(d_item_chunk, item_chunk.data(), item_chunk.size());
    }
    if (!tile_matrix.empty()) {
        pack.add(d_tile_matrix, tile_matrix.data(), tile_matrix.size());
        pack.add(d_tile_chunk, tile_chunk.data(), tile_chunk.size());
    }
    if (!mask_matrix.empty()) {
        pack.add(tmp->mask_matrix, mask_matrix.data(), mask_matrix.size());
        pack.add(tmp->mask_chunk, mask_chunk.data(), mask_chunk.size());
        pack.add(tmp->mask_off, mask_off.data(), M);
    }
    if (!word_matrix.empty()) {
        pack.add(tmp->word_matrix, word_matrix.data(), word_matrix.size());
        pack.add(tmp->word_chunk, word_chunk.data(), word_chunk.size());
        pack.add(tmp->word_off, word_off.data(), M);
        pack.addZero(tmp->path_words, word_total);
    }
    std::vector<uint32_t> segment_off;
    if (collapse) {
        segment_off.resize(M + 1);
        for (uint32_t m = 0; m < M; ++m) segment_off[m] = static_cast<uint32_t>(row_off[m]);
        segment_off[M] = static_cast<uint32_t>(row_total);
        if (row_total > 0x7fffffffull) e = hipErrorInvalidValue;
        pack.add(g->collapse_segment_off, segment_off.data(), M + 1);
    }
    DeviceBuffer<uint32_t> & d_cluster = tmp->cluster;
    pack.add(d_cluster, spec->cluster, M);
    if (lists_needed) {
        pack.addZero(d_degree, inc_total);
        pack.addZero(d_cursor, inc_total);
    }
    pack.addZero(d_error, 1);
    ok(pack.commit(st));
    if (from_sources) {  // the lists stay on the device: gathered from the batch's columns
        ok(d_group_path_off.alloc(num_columns + 1));
        ok(d_group_path.alloc(num_incidences));
        ok(tmp->column_counts.alloc(num_columns));
        if (e == hipSuccess) {
            gatherSourceColumnsKernel<<<dim3(M), dim3(256), 0, st>>>(M, d_cluster.ptr, batch->cluster_src_off.ptr, batch->src_col_count.ptr,
                                                                   batch->src_col_end.ptr, batch->src_col_path.ptr, d_group_off.ptr, tmp->first_path.ptr,
                                                                   d_group_path_off.ptr, d_group_path.ptr, tmp->column_counts.ptr);
            ok(hipGetLastError());
        }
        g->d_column_counts = tmp->column_counts.ptr;
    }
    if (single_paths) {
        ok(d_group_path_off.alloc(num_columns + 1));
        ok(d_group_path.alloc(num_incidences));
        if (e == hipSuccess) {
            singlePathColumnsKernel<<<dim3(M), dim3(256), 0, st>>>(M, d_group_off.ptr, d_group_path_off.ptr, d_group_path.ptr);
            ok(hipGetLastError());
        }
    }
    g->d_group_off = d_group_off.ptr;
    g->d_group_path_off = d_group_path_off.ptr;
    g->d_group_path = d_group_path.ptr;
    g->d_cluster = d_cluster.ptr;
    ctx->spanEnd(span);
    ctx->stats.h2d_bytes += static_cast<double>(M * 60 + num_columns * 8 + num_incidences * 4 + (item_matrix.size() + tile_matrix.size()) * 8);
    sub.reset(new HostScope("groups_build: allocations"));
    ok(g->values.alloc(val_total + 2));
    ok(g->rowmax.alloc(row_total));
    ok(g->row_perm.alloc(row_total));
    ok(g->row_count.alloc(row_total + 2));
    ok(g->row_noise.alloc(row_total + 2));
    ok(g->mat_fast.alloc(M));
    ok(g->mat_mid.alloc(M));
    if (collapse) {
        ok(g->collapse_key.alloc(row_total));
        ok(g->collapse_row.alloc(row_total));
        ok(g->collapse_mask.alloc(row_total));
    }
    size_t scan_bytes = 0;
    if (lists_needed) {
        ok(d_path_grp_off.alloc(inc_total));
        ok(d_path_grp.alloc(num_incidences));
        if (e == hipSuccess) ok(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, d_degree.ptr, d_path_grp_off.ptr, static_cast<int>(inc_total), st));
        ok(d_scan_tmp.alloc(scan_bytes));
        if (e == hipSuccess && inc_total > 0x7fffffffull) e = hipErrorInvalidValue;
    }
    if (!mask_matrix.empty()) ok(tmp->masks.alloc(mask_total));
    sub.reset(new HostScope("groups_build: launches"));
    if (e == hipSuccess) {
        span = ctx->spanBegin(FAM_BUILD);
        if (!item_matrix.empty()) {
            // (a hipMemsetAsync per wide matrix was up to four fill kernels each: 370 commands per lane on the configs[4] batch)
            zeroWideMatricesKernel<<<dim3(static_cast<uint32_t>(item_matrix.size())), dim3(256), 0, st>>>(
                d_item_matrix.ptr, d_item_chunk.ptr, g->mat_val_off.ptr, g->mat_rows.ptr, g->mat_cols.ptr, g->values.ptr);
        }
        partitionRowsKernel<1024><<<dim3(M), dim3(1024), 0, st>>>(M, g->mat_row_off.ptr, g->mat_row0.ptr, g->mat_rows.ptr, batch->row_count.ptr,
                                                           batch->row_noise.ptr, g->row_perm.ptr, g->row_count.ptr, g->row_noise.ptr,
                                                           g->mat_fast.ptr, g->mat_mid.ptr,
                                                           kMidMinRows);
        const uint32_t col_blocks = static_cast<uint32_t>((num_columns + 255) / 256);
        MaskBuildArgs ma;
        memset(&ma, 0, sizeof(ma));
        {
            ma.mat_val_off = g->mat_val_off.ptr;
            ma.mat_row_off = g->mat_row_off.ptr;
            ma.mat_row0 = g->mat_row0.ptr;
            ma.mat_rows = g->mat_rows.ptr;
            ma.mat_cols = g->mat_cols.ptr;
            ma.num_paths = d_num_paths.ptr;
            ma.row_ent_off = batch->row_ent_off.ptr;
            ma.ent_path = batch->ent_path.ptr;
            ma.ent_prob = batch->ent_prob.ptr;
            ma.row_noise = batch->row_noise.ptr;
            ma.row_perm = g->row_perm.ptr;
            ma.normalise = spec->normalise ? 1 : 0;
#ifdef RPVG_HIP_EXPERIMENTS
            ma.debug_no_long_rows = RPVG_EXPERIMENT_ENV("RPVG_HIP_BUILD_DEBUG") ? std::atoi(RPVG_EXPERIMENT_ENV("RPVG_HIP_BUILD_DEBUG")) & 1 : 0;
#endif
            ma.values = g->values.ptr;
            ma.rowmax = g->rowmax.ptr;
            ma.collapse_key = g->collapse_key.ptr;
            ma.collapse_row = g->collapse_row.ptr;
            ma.collapse_mask = g->collapse_mask.ptr;
        }
        if (!mask_matrix.empty() && direct) {
            ma.num_items = static_cast<uint32_t>(mask_matrix.size());
            ma.item_matrix = tmp->mask_matrix.ptr;
            ma.item_chunk = tmp->mask_chunk.ptr;
            groupsBuildSinglePathKernel<<<dim3((ma.num_items + 3) / 4), dim3(256), 0, st>>>(ma);
        } else if (!mask_matrix.empty()) {
            columnMaskKernel<<<dim3(col_blocks), dim3(256), 0, st>>>(M, num_columns, d_group_off.ptr, d_group_path_off.ptr, d_group_path.ptr,
                                                                    d_num_paths.ptr, tmp->mask_off.ptr, tmp->masks.ptr, d_error.ptr);
            ma.num_items = static_cast<uint32_t>(mask_matrix.size());
            ma.item_matrix = tmp->mask_matrix.ptr;
            ma.item_chunk = tmp->mask_chunk.ptr;
            ma.mask_off = tmp->mask_off.ptr;
            ma.masks = tmp->masks.ptr;
            groupsBuildMaskKernel<<<dim3((ma.num_items + 3) / 4), dim3(256), 0, st>>>(ma);
        }
        if (!word_matrix.empty()) {
            pathColumnWordKernel<<<dim3(col_blocks), dim3(256), 0, st>>>(M, num_columns, d_group_off.ptr, d_group_path_off.ptr, d_group_path.ptr,
                                                                        d_num_paths.ptr, tmp->word_off.ptr,
                                                                        reinterpret_cast<unsigned long long *>(tmp->path_words.ptr), d_error.ptr);
            ma.num_items = static_cast<uint32_t>(word_matrix.size());
            ma.item_matrix = tmp->word_matrix.ptr;
            ma.item_chunk = tmp->word_chunk.ptr;
            ma.mask_off = tmp->word_off.ptr;
            ma.masks = tmp->path_words.ptr;
            groupsBuildWordKernel<<<dim3((ma.num_items + 3) / 4), dim3(256), 0, st>>>(ma);
        }
        if (lists_needed) {
            incidenceCountKernel<<<dim3(col_blocks), dim3(256), 0, st>>>(M, num_columns, d_group_off.ptr, d_group_path_off.ptr,
                                                                       d_group_path.ptr, d_inc_off.ptr, d_num_paths.ptr,
                                                                       d_degree.ptr, d_error.ptr);
            ok(hipcub::DeviceScan::ExclusiveSum(d_scan_tmp.ptr, scan_bytes, d_degree.ptr, d_path_grp_off.ptr, static_cast<int>(inc_total), st));
            incidenceFillKernel<<<dim3(col_blocks), dim3(256), 0, st>>>(M, num_columns, d_group_off.ptr, d_group_path_off.ptr,
                                                                      d_group_path.ptr, d_inc_off.ptr, d_num_paths.ptr,
                                                                      d_path_grp_off.ptr, d_cursor.ptr, d_path_grp.ptr);
        }
        if (!tile_matrix.empty()) {
            groupsBuildTileKernel<<<dim3(static_cast<uint32_t>(tile_matrix.size())), dim3(256), (kTileDoubles + 256) * sizeof(double), st>>>(
                static_cast<uint32_t>(tile_matrix.size()), d_tile_matrix.ptr, d_tile_chunk.ptr, g->mat_val_off.ptr,
                g->mat_row_off.ptr, g->mat_row0.ptr, g->mat_rows.ptr, g->mat_cols.ptr, d_inc_off.ptr, d_path_grp_off.ptr,
                d_path_grp.ptr, batch->row_ent_off.ptr, batch->ent_path.ptr, batch->ent_prob.ptr, batch->row_noise.ptr,
                g->row_perm.ptr, spec->normalise ? 1 : 0, g->values.ptr, g->rowmax.ptr, g->collapse_key.ptr, g->collapse_row.ptr, g->collapse_mask.ptr);
        }
        if (!item_matrix.empty()) {
            groupsBuildKernel<<<dim3(static_cast<uint32_t>(item_matrix.size())), dim3(256), 0, st>>>(
                static_cast<uint32_t>(item_matrix.size()), d_item_matrix.ptr, d_item_chunk.ptr, g->mat_val_off.ptr,
                g->mat_row_off.ptr, g->mat_row0.ptr, g->mat_rows.ptr, g->mat_cols.ptr, d_inc_off.ptr, d_path_grp_off.ptr,
                d_path_grp.ptr, batch->row_ent_off.ptr, batch->ent_path.ptr, batch->ent_prob.ptr, batch->row_noise.ptr,
                g->row_perm.ptr, spec->normalise ? 1 : 0, g->values.ptr, g->rowmax.ptr, g->collapse_key.ptr, g->collapse_row.ptr, g->collapse_mask.ptr);
        }
        sub.reset(new HostScope("groups_build: row collapse launches"));
        if (collapse && e == hipSuccess) {
            // on the context's collapse stream, behind the build (rpvg_hip_groups::collapse_done)
            hipStream_t collapse_stream = ctx->collapse_stream;
            static const bool same_stream = RPVG_EXPERIMENT_ENV("RPVG_HIP_COLLAPSE_INLINE") != nullptr;  // A/B knob: on the build's stream (11.5-12.1 vs 10.2-10.4 ms per batch)
            if (same_stream) collapse_stream = st;
            ok(hipEventCreateWithFlags(&g->built, hipEventDisableTiming));
            ok(hipEventCreateWithFlags(&g->collapse_done, hipEventDisableTiming));
            if (e == hipSuccess && collapse_stream != st) {
                ok(hipEventRecord(g->built, st));
                ok(hipStreamWaitEvent(collapse_stream, g->built, 0));
            }
            const int collapse_span = (e == hipSuccess) ? ctx->spanBegin(FAM_COLLAPSE, collapse_stream) : -1;
            // (matrices from the batch's own columns are the diploid search's: it reads them while the collapse finds its runs,
            // rpvg_hip_groups::held_back_runs; RPVG_HIP_COLLAPSE_BEFORE_SEARCH=1: the whole collapse first, as for everybody else)
            const char * before_env = std::getenv("RPVG_HIP_COLLAPSE_BEFORE_SEARCH");  // (read per call: the tests take both ways)
            const bool before_search = before_env != nullptr && std::atoi(before_env) != 0;
            if (e == hipSuccess) ok(queueRowCollapse(ctx, g, row_total, spec->collapse_precision, collapse_stream, from_sources && !before_search));
            ctx->spanEnd(collapse_span);
            if (e == hipSuccess) ok(hipEventRecord(g->collapse_done, collapse_stream));
        }
        sub.reset();
        ctx->spanEnd(span);
        ctx->stats.build_launches += 3;
        ok(hipGetLastError());
        static const bool wait_here = RPVG_EXPERIMENT_ENV("RPVG_HIP_BUILD_SYNC") != nullptr;  // A/B knob: host sync before returning
        if (wait_here) ok(hipStreamSynchronize(st));
    }
    if (e != hipSuccess) {
        setError("rpvg_hip_groups_build: %s", hipGetErrorString(e));
        (void) hipStreamSynchronize(st);  // whatever was queued before the failure still uses the buffers
        delete g;
        return (e == hipErrorOutOfMemory) ? RPVG_HIP_ERR_ALLOC : RPVG_HIP_ERR_RUNTIME;
    }
    *groups_out = g;
    return RPVG_HIP_OK;
}

int rpvg_hip_groups::buildError(hipStream_t stream) const {
    if (build_checked || !build_error_flag.ptr) return RPVG_HIP_OK;
    uint32_t bad = 0;
    hipError_t e = hipMemcpyAsync(&bad, build_error_flag.ptr, sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) {
        setError("rpvg_hip_groups_build: %s", hipGetErrorString(e));
        return RPVG_HIP_ERR_RUNTIME;
    }
    if (bad) {
        setError(bad == 2 ? "rpvg_hip_groups_build: a group lists a path twice" : "rpvg_hip_groups_build: a group refers to a path outside its cluster");
        return RPVG_HIP_ERR_INVALID;
    }
    build_checked = true;
    return RPVG_HIP_OK;
}

extern "C" void rpvg_hip_groups_free(rpvg_hip_ctx * ctx, rpvg_hip_groups * groups) {
    if (!groups) return;
    if (ctx) {
        std::lock_guard<std::mutex> lock(ctx->mutex);
        (void) hipSetDevice(ctx->device);
        (void) hipStreamSynchronize(ctx->stream);
        static const bool trace = std::getenv("RPVG_AMD_TRACE") != nullptr;
        if (trace && groups->collapse_info.ptr && groups->waitCollapse(ctx->stream) == hipSuccess) {
            uint32_t info[4] = {0, 0, 0, 0};
            // (on the context's stream: a plain hipMemcpy waits for every stream of the device, the other lane's search included)
            if (hipMemcpyAsync(info, groups->collapse_info.ptr, sizeof(info), hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
                hipStreamSynchronize(ctx->stream) == hipSuccess) {
                std::fprintf(stderr, "[rpvg_hip trace]   row collapse: %u of %u matrices replayed (%u sorted whole), %u active rows, %u rows took their head's values\n",
                             info[0], groups->num_matrices, info[2], info[3], info[1]);
            }
        }
        delete groups;
    } else {
        delete groups;
    }
}

extern "C" int rpvg_hip_group_loglik(rpvg_hip_ctx * ctx, const rpvg_hip_groups * groups, uint32_t num_requests,
                                     const uint32_t * matrix, const uint32_t * members, uint32_t width, double divisor,
                                     const uint8_t * add_rowmax, double * out) {
    RPVG_REQUIRE(ctx && groups, "rpvg_hip_group_loglik: NULL argument");
    if (num_requests == 0) return RPVG_HIP_OK;
    RPVG_REQUIRE(matrix && members && out, "rpvg_hip_group_loglik: NULL request arrays");
    RPVG_REQUIRE(width >= 1 && width <= 4, "rpvg_hip_group_loglik: width %u outside [1, 4]", width);
    RPVG_REQUIRE(divisor > 0, "rpvg_hip_group_loglik: divisor must be positive");
    double evals = 0;
    for (uint32_t q = 0; q < num_requests; ++q) {
        RPVG_REQUIRE(matrix[q] < groups->num_matrices, "rpvg_hip_group_loglik: request %u refers to matrix %u of %u", q,
                     matrix[q], groups->num_matrices);
        for (uint32_t w = 0; w < width; ++w) {
            const uint32_t mem = members[static_cast<uint64_t>(q) * width + w];
            RPVG_REQUIRE(mem == kNoMember || mem < groups->h_num_cols[matrix[q]],
                         "rpvg_hip_group_loglik: request %u member %u >= %u columns", q, mem, groups->h_num_cols[matrix[q]]);
        }
        evals += static_cast<double>(groups->h_num_rows[matrix[q]]);
    }

    std::lock_guard<std::mutex> lock(ctx->mutex);
    RPVG_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    RPVG_HIP_CHECK(groups->waitCollapse(st));
    DeviceBuffer<uint32_t> d_matrix, d_members;
    DeviceBuffer<uint8_t> d_flag;
    DeviceBuffer<double> d_out;
    int span = ctx->spanBegin(FAM_H2D);
    RPVG_HIP_CHECK(d_matrix.upload(matrix, num_requests, st));
    RPVG_HIP_CHECK(d_members.upload(members, static_cast<size_t>(num_requests) * width, st));
    if (add_rowmax) RPVG_HIP_CHECK(d_flag.upload(add_rowmax, num_requests, st));
    ctx->spanEnd(span);
    ctx->stats.h2d_bytes += static_cast<double>(num_requests) * (4 + 4 * width + (add_rowmax ? 1 : 0));
    RPVG_HIP_CHECK(d_out.alloc(num_requests));

    const uint32_t blocks = (num_requests + 3) / 4;
    span = ctx->spanBegin(FAM_LOGLIK);
#define RPVG_LAUNCH_LOGLIK(W)                                                                                              \
    groupLoglikKernel<W><<<dim3(blocks), dim3(256), 0, st>>>(num_requests, d_matrix.ptr, d_members.ptr, d_flag.ptr, divisor, \
                                                            groups->mat_val_off.ptr, groups->mat_row_off.ptr,               \
                                                            groups->mat_fast.ptr, groups->mat_mid.ptr, groups->mat_rows.ptr, groups->values.ptr, \
                                                            groups->rowmax.ptr, groups->row_count.ptr, groups->row_noise.ptr, d_out.ptr)
    switch (width) {
        case 1: RPVG_LAUNCH_LOGLIK(1); break;
        case 2: RPVG_LAUNCH_LOGLIK(2); break;
        case 3: RPVG_LAUNCH_LOGLIK(3); break;
        default: RPVG_LAUNCH_LOGLIK(4); break;
    }
#undef RPVG_LAUNCH_LOGLIK
    ctx->spanEnd(span);
    ctx->stats.loglik_launches += 1;
    ctx->stats.loglik_evals += evals;
    RPVG_HIP_CHECK(hipGetLastError());
    RPVG_HIP_CHECK(d_out.download(out, st));
    RPVG_HIP_CHECK(waitStream(st));
    return groups->buildError(st);  // the matrices were built without a host sync
}

extern "C" int rpvg_hip_group_conditionals(rpvg_hip_ctx * ctx, const rpvg_hip_groups * groups, uint32_t num_requests,
                                           const uint32_t * matrix, const uint32_t * others, uint32_t width,
                                           double divisor, double * out) {
    RPVG_REQUIRE(ctx && groups, "rpvg_hip_group_conditionals: NULL argument");
    if (num_requests == 0) return RPVG_HIP_OK;
    RPVG_REQUIRE(matrix && out && (others || width == 1), "rpvg_hip_group_conditionals: NULL request arrays");
    RPVG_REQUIRE(width >= 1 && width <= 4, "rpvg_hip_group_conditionals: width %u outside [1, 4]", width);
    RPVG_REQUIRE(divisor > 0, "rpvg_hip_group_conditionals: divisor must be positive");
    std::vector<uint64_t> item_off(num_requests + 1, 0), out_off(num_requests + 1, 0);
    double evals = 0;
    for (uint32_t q = 0; q < num_requests; ++q) {
        RPVG_REQUIRE(matrix[q] < groups->num_matrices, "rpvg_hip_group_conditionals: request %u refers to matrix %u of %u",
                     q, matrix[q], groups->num_matrices);
        const uint32_t G = groups->h_num_cols[matrix[q]];
        for (uint32_t w = 0; w + 1 < width; ++w) {
            RPVG_REQUIRE(others[static_cast<uint64_t>(q) * (width - 1) + w] < G,
                         "rpvg_hip_group_conditionals: request %u member %u >= %u columns", q,
                         others[static_cast<uint64_t>(q) * (width - 1) + w], G);
        }
        item_off[q + 1] = item_off[q] + (G + 3) / 4;
        out_off[q + 1] = out_off[q] + G;
        evals += static_cast<double>(groups->h_num_rows[matrix[q]]) * G;
    }
    const uint64_t num_items = item_off[num_requests];

    std::lock_guard<std::mutex> lock(ctx->mutex);
    RPVG_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    RPVG_HIP_CHECK(groups->waitCollapse(st));
    DeviceBuffer<uint32_t> d_matrix, d_others;
    DeviceBuffer<uint64_t> d_item_off, d_out_off;
    DeviceBuffer<double> d_out;
    int span = ctx->spanBegin(FAM_H2D);
    RPVG_HIP_CHECK(d_matrix.upload(matrix, num_requests, st));
    if (width > 1) RPVG_HIP_CHECK(d_others.upload(others, static_cast<size_t>(num_requests) * (width - 1), st));
    RPVG_HIP_CHECK(d_item_off.upload(item_off.data(), item_off.size(), st));
    RPVG_HIP_CHECK(d_out_off.upload(out_off.data(), out_off.size(), st));
    ctx->spanEnd(span);
    ctx->stats.h2d_bytes += static_cast<double>(num_requests) * (4 + 4 * (width - 1) + 16);
    RPVG_HIP_CHECK(d_out.alloc(out_off[num_requests]));

    const uint64_t blocks = (num_items + 3) / 4;
    RPVG_REQUIRE(blocks <= 0x7fffffffull, "rpvg_hip_group_conditionals: %llu work items exceed one launch",
                 static_cast<unsigned long long>(num_items));
    span = ctx->spanBegin(FAM_LOGLIK);
#define RPVG_LAUNCH_COND(W)                                                                                                  \
    groupConditionalKernel<W><<<dim3(static_cast<uint32_t>(blocks)), dim3(256), 0, st>>>(                                    \
        num_requests, num_items, d_item_off.ptr, d_out_off.ptr, d_matrix.ptr, d_others.ptr, divisor, groups->mat_val_off.ptr, \
        groups->mat_row_off.ptr, groups->mat_fast.ptr, groups->mat_mid.ptr, groups->mat_rows.ptr, groups->mat_cols.ptr,      \
        groups->values.ptr,                                                                                                  \
        groups->row_count.ptr, groups->row_noise.ptr, d_out.ptr)
    switch (width) {
        case 1: RPVG_LAUNCH_COND(1); break;
        case 2: RPVG_LAUNCH_COND(2); break;
        case 3: RPVG_LAUNCH_COND(3); break;
        default: RPVG_LAUNCH_COND(4); break;
    }
#undef RPVG_LAUNCH_COND
    ctx->spanEnd(span);
    ctx->stats.loglik_launches += 1;
    ctx->stats.loglik_evals += evals;
    RPVG_HIP_CHECK(hipGetLastError());
    RPVG_HIP_CHECK(d_out.download(out, st));
    RPVG_HIP_CHECK(waitStream(st));
    return groups->buildError(st);  // the matrices were built without a host sync
}
