// EM problems too large for one workgroup: every iteration runs over the whole GPU (gfx950).
//
// rpvg_hip_em_solve / rpvg_hip_nested_subset_em put every problem on ONE workgroup (em_sparse.hip): right for the
// thousands of small problems of a batch, wrong for a cluster of 10^5 - 10^6 rows (a 200 000-row problem ran at 563 us
// per EM iteration, 20 GB/s, on one of 256 CUs).  The reference's EMAbundanceEstimator
// (src/path_abundance_estimator.cpp:47-114) takes a cluster of any size through the same call
// (PathAbundanceEstimator::estimate, :18-45, src/main.cpp:977); here the size bin of a problem decides: a problem whose
// kept rows + entries reach emGridMinWork() lands in the grid bin (emBinOf, em_sparse.hip), the device describes those
// problems to the host (emGridDescribeKernel) and the host solves them one after the other with one round of launches
// per EM iteration:
//
//   CSR route    emGridAccumKernel<LANES,UNROLL> grid-wide pass over the problem's compacted CSR; a workgroup owns a
//                                               contiguous range of rows, abundance vector and accumulators in LDS,
//                                               one partial column-sum vector per workgroup
//                emGridUpdateKernel             column sums over the partials in a fixed order, the update
//                                               a'_j = a_j t_j / T (noise: + Z), the convergence test, and — in the last
//                                               workgroup through — the reference's stop rule
//   dense route  a row-major copy of the problem (emGridDenseBuildKernel) and the streaming kernels of em_dense.hip,
//                when the dense matrix is the smaller representation (8 B per cell against 12 B per entry) and a row
//                fits the registers of a workgroup (C <= 2048): BASELINE.json configs[1], one 1M x 2k cluster, behind
//                `-i transcripts` at the HBM roofline
//
// A dependent kernel boundary costs ~1.5 us on this GPU, a grid-wide barrier inside a persistent kernel 4-7 us
// (MI355X_MICROARCH.md, price list: boundary / barrier-xcd), so the iterations are launches, not barriers; a device-side
// `done` word makes the launches behind the last iteration return at once, the host queues iterations in chunks and
// looks at the word between chunks (two chunks in flight), and the loop stops at exactly the reference's iteration.
// Same arithmetic per row as emSparseProblem (em_sparse.hip): reciprocal + two Newton steps + residual correction.

#include "common.hpp"

#include <atomic>

#include <algorithm>
#include <cstdlib>
#include <vector>

using namespace rpvg_hip_detail;

namespace {

constexpr double kMinEmAbundance = 1e-8;   // src/path_abundance_estimator.cpp:11
constexpr uint32_t kMinEmConvIts = 10;     // src/path_abundance_estimator.cpp:10
constexpr int kGridBlock = 256;

struct GridAccumArgs {
    const uint32_t * off;    // [rows + 1] entry offsets of the problem's rows (relative to col / val)
    const double * cnt;      // [rows] read counts (after the row collapse, if it merged rows of the problem)
    const double * nz;       // [rows] noise probabilities
    const uint32_t * col;    // [entries]
    const double * val;      // [entries] normalised probabilities
    uint32_t rows, C;
    uint32_t rows_per_block;
    const double * a;        // [C] abundances (last = noise)
    double * partials;       // [gridDim.x x partial_ld]
    uint32_t partial_ld;
    const EmGridControl * ctl;
};

// c / s as the one-workgroup kernels compute it (em_sparse.hip): hardware reciprocal, two Newton steps, one residual
// correction of the quotient
__device__ __forceinline__ double countOverSum(const double c, const double s) {
    double y = __builtin_amdgcn_rcp(s);
    y = fma(fma(-s, y, 1.0), y, y);
    y = fma(fma(-s, y, 1.0), y, y);
    const double quot = c * y;
    // (a row whose count a row collapse moved to its run head takes no part: row_collapse.hip)
    return c == 0.0 ? 0.0 : fma(fma(-s, quot, c), y, quot);
}

// sum over the LANES neighbouring lanes that share a row
template <int LANES>
__device__ __forceinline__ double rowLanesSum(double v) {
    if (LANES == 64) return waveSumF64(v);
#pragma unroll
    for (int d = LANES / 2; d >= 1; d >>= 1) v += __shfl_xor(v, d, LANES);
    return v;
}

// (The accumulators: one vector per wavefront, added up in wavefront order, and the workgroups' partial vectors in workgroup order
// below — the CSR route has one order of additions, like the one-workgroup kernels; the dense route has no atomics at all.)
// One streaming pass over the problem's CSR.  LANES lanes share a row (1: a thread per row — short rows, the entries of
// neighbouring rows are neighbours in memory; 4 / 16 / 64: the lanes stride the row's entries, the row sum by shuffles
// or DPP), and every row slot walks UNROLL rows at a time: their offsets, counts and noise are loaded together and their
// entry loops follow each other without a dependent load in between — a pass is a chain of two dependent loads per row
// (offsets, then entries), ~1 us each from L2 or memory, and with one row at a time that chain was the pass.
template <int LANES, int UNROLL>
__global__ __launch_bounds__(kGridBlock) void emGridAccumKernel(const GridAccumArgs args) {
    if (args.ctl->done) return;
    extern __shared__ __attribute__((aligned(16))) double grid_lds[];
    const uint32_t C = args.C, noise_col = C - 1;
    constexpr uint32_t kWaves = kGridBlock / 64;
    double * a = grid_lds;   // [C]
    double * t = a + C;      // [kWaves x C]: an accumulator vector per wavefront, added up in wavefront order (em_sparse.hip, emSparseProblem)
    double * tw = t + static_cast<size_t>(threadIdx.x >> 6) * C;
    for (uint32_t j = threadIdx.x; j < C; j += kGridBlock) a[j] = args.a[j];
    for (uint32_t j = threadIdx.x; j < kWaves * C; j += kGridBlock) t[j] = 0.0;
    __syncthreads();
    constexpr uint32_t kSlots = kGridBlock / LANES;  // rows the workgroup holds at once
    const uint32_t slot = threadIdx.x / LANES, sl = threadIdx.x % LANES;
    const uint32_t r0 = blockIdx.x * args.rows_per_block;
    const uint32_t r1 = min(args.rows, r0 + args.rows_per_block);
    const double a_noise = a[noise_col];
    double tn = 0.0;
    for (uint32_t base = r0 + slot; base < r1; base += kSlots * UNROLL) {
        uint32_t e0[UNROLL], e1[UNROLL];
        double nz[UNROLL], c[UNROLL], s[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const uint32_t r = base + u * kSlots;
            const bool ok = r < r1;
            e0[u] = ok ? args.off[r] : 0u;
            e1[u] = ok ? args.off[r + 1] : 0u;
            nz[u] = ok ? args.nz[r] : 0.0;
            c[u] = ok ? args.cnt[r] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            double x = 0.0;
            for (uint32_t e = e0[u] + sl; e < e1[u]; e += LANES) x += args.val[e] * a[args.col[e]];
            s[u] = x;
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const double w = countOverSum(c[u], rowLanesSum<LANES>(s[u]) + nz[u] * a_noise);  // (no row: count 0, weight 0)
            // (the columns of one row are distinct: the lanes of a row never meet on an accumulator)
            for (uint32_t e = e0[u] + sl; e < e1[u]; e += LANES) atomicAdd(&tw[args.col[e]], w * args.val[e]);
            if (sl == 0) tn += w * nz[u];
        }
    }
    // the noise column has no entries: its accumulator takes the per-wave sums of w * noise
    tn = waveSumF64(tn);
    if ((threadIdx.x & 63) == 0 && tn != 0.0) atomicAdd(&tw[noise_col], tn);
    __syncthreads();
    double * out = args.partials + static_cast<uint64_t>(blockIdx.x) * args.partial_ld;
    for (uint32_t j = threadIdx.x; j < C; j += kGridBlock) {
        double tj = t[j];
#pragma unroll
        for (uint32_t w = 1; w < kWaves; ++w) tj += t[w * C + j];
        out[j] = tj;
    }
}

struct GridUpdateArgs {
    uint32_t C, num_partials, partial_ld;
    const double * partials;
    double * a;
    double inv_total, zero_mass, max_rel_em_conv;
    uint32_t max_em_its;
    EmGridControl * ctl;
};

// 16 columns per workgroup, 64 slices of the partials (thread = (column, slice): 128-byte requests, four loads in flight per
// thread for 256 partials — a partial comes from another XCD's L2 or from memory, ~1 us each: a first version with four waves
// and one load at a time spent 40 us here per iteration, 64 columns x 16 slices 7.3), the slices meet in LDS in slice order —
// the order of the additions is fixed.  The last workgroup through applies the stop rule.
constexpr int kUpdateBlock = 1024;
constexpr int kUpdateColumns = 16;
constexpr int kUpdateSlices = kUpdateBlock / kUpdateColumns;

__global__ __launch_bounds__(kUpdateBlock) void emGridUpdateKernel(const GridUpdateArgs args) {
    if (args.ctl->done) return;
    __shared__ double slice_sum[kUpdateSlices][kUpdateColumns];
    const uint32_t column = threadIdx.x % kUpdateColumns, slice = threadIdx.x / kUpdateColumns;
    const uint32_t j = blockIdx.x * kUpdateColumns + column;
    double acc = 0.0;
    if (j < args.C) {
        const double * partial = args.partials + j;
        const uint64_t ld = args.partial_ld;
        uint32_t b = slice;
        for (; b + 3 * kUpdateSlices < args.num_partials; b += 4 * kUpdateSlices) {
            double v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = partial[static_cast<uint64_t>(b + u * kUpdateSlices) * ld];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += v[u];
        }
        for (; b < args.num_partials; b += kUpdateSlices) acc += partial[static_cast<uint64_t>(b) * ld];
    }
    slice_sum[slice][column] = acc;
    __syncthreads();
    int viol = 0;
    if (slice == 0 && j < args.C) {
        double tj = slice_sum[0][column];
#pragma unroll 8
        for (int w = 1; w < kUpdateSlices; ++w) tj += slice_sum[w][column];
        const double aj = args.a[j];
        // a'_j = a_j t_j / T;  a'_noise = (a_noise t_noise + Z) / T  (em_sparse.hip: the same roundings)
        const double an = (j + 1 == args.C) ? (aj * tj + args.zero_mass) * args.inv_total : (aj * tj) * args.inv_total;
        // |an - aj| / an > eps  (src/path_abundance_estimator.cpp:73-75), without the division
        if (an >= kMinEmAbundance && fabs(an - aj) > args.max_rel_em_conv * an) viol = 1;
        args.a[j] = an;
    }
    const int any_viol = __syncthreads_or(viol);
    if (threadIdx.x == 0) {
        EmGridControl * ctl = args.ctl;
        if (any_viol) atomicOr(&ctl->viol, 1u);
        __threadfence();
        if (atomicAdd(&ctl->arrived, 1u) == gridDim.x - 1) {  // every workgroup's violations are in
            const uint32_t v = atomicOr(&ctl->viol, 0u);
            ctl->iterations += 1;
            if (v == 0) {
                ctl->conv_its += 1;
                if (ctl->conv_its == kMinEmConvIts) ctl->done = 1;
            } else {
                ctl->conv_its = 0;
            }
            if (ctl->iterations >= args.max_em_its) ctl->done = 1;
            ctl->viol = 0;
            ctl->arrived = 0;
        }
    }
}

// src/path_abundance_estimator.cpp:100-113: expected read counts, sub-threshold components moved to the noise count
__global__ __launch_bounds__(kGridBlock) void emGridFinishKernel(const uint32_t C, const double * __restrict__ a, const double T,
                                                               const EmGridControl * __restrict__ ctl, double * __restrict__ abundances,
                                                               double * __restrict__ noise_count, uint32_t * __restrict__ iterations) {
    __shared__ double wave_low[kGridBlock / 64];
    const uint32_t noise_col = C - 1;
    double low = 0.0;
    for (uint32_t j = threadIdx.x; j < noise_col; j += kGridBlock) {
        const double aj = a[j];
        if (aj < kMinEmAbundance) {
            low += aj * T;
            abundances[j] = 0;
        } else {
            abundances[j] = aj * T;
        }
    }
    low = waveSumF64(low);
    if ((threadIdx.x & 63) == 0) wave_low[threadIdx.x >> 6] = low;
    __syncthreads();
    if (threadIdx.x == 0) {
        double total_low = wave_low[0];
        for (int w = 1; w < kGridBlock / 64; ++w) total_low += wave_low[w];
        *noise_count = total_low + a[noise_col] * T;
        *iterations = ctl->iterations;
    }
}

__global__ void gridFillKernel(double * x, const uint32_t n, const double v) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = v;
}

// dense row-major copy of a problem's CSR (zero-filled before): a wavefront per row
__global__ __launch_bounds__(kGridBlock) void emGridDenseBuildKernel(const uint32_t rows, const uint32_t C, const uint64_t ld,
                                                                   const uint32_t * __restrict__ off, const double * __restrict__ nz,
                                                                   const uint32_t * __restrict__ col, const double * __restrict__ val,
                                                                   double * __restrict__ matrix) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t waves = gridDim.x * (kGridBlock / 64);
    for (uint32_t r = blockIdx.x * (kGridBlock / 64) + (threadIdx.x >> 6); r < rows; r += waves) {
        double * out = matrix + static_cast<uint64_t>(r) * ld;
        const uint32_t e0 = off[r], e1 = off[r + 1];
        for (uint32_t e = e0 + lane; e < e1; e += 64) out[col[e]] = val[e];
        if (lane == 0) out[C - 1] = nz[r];
    }
}

template <int LANES, int UNROLL>
hipError_t launchAccumVariant(const GridAccumArgs & args, const uint32_t grid, const size_t lds, hipStream_t st) {
    if (lds > 64 * 1024) {
        // (once per variant and size reached, not per launch: this runs every EM iteration)
        static std::atomic<size_t> granted{0};
        if (lds > granted.load(std::memory_order_relaxed)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&emGridAccumKernel<LANES, UNROLL>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               static_cast<int>(lds));
            if (e != hipSuccess) return e;
            granted.store(lds, std::memory_order_relaxed);
        }
    }
    emGridAccumKernel<LANES, UNROLL><<<dim3(grid), dim3(kGridBlock), lds, st>>>(args);
    return hipSuccess;
}

// lanes per row by the mean row length
inline int gridRowLanes(const uint32_t rows, const uint32_t entries) {
    if (const char * env = RPVG_EXPERIMENT_ENV("RPVG_HIP_EM_GRID_ROW_LANES")) return std::atoi(env);  // A/B knob: 1, 4, 16, 64
    const double mean = static_cast<double>(entries) / std::max(1u, rows);
    return mean < 6.0 ? 1 : mean < 24.0 ? 4 : mean < 96.0 ? 16 : 64;
}

hipError_t launchAccum(const int lanes, const GridAccumArgs & args, const uint32_t grid, const size_t lds, hipStream_t st) {
    switch (lanes) {
        case 1: return launchAccumVariant<1, 4>(args, grid, lds, st);
        case 4: return launchAccumVariant<4, 4>(args, grid, lds, st);
        case 16: return launchAccumVariant<16, 4>(args, grid, lds, st);
        default: return launchAccumVariant<64, 4>(args, grid, lds, st);
    }
}

}  // namespace

namespace rpvg_hip_detail {

uint64_t emGridMinWork() {
    // (read per call: the tests take both ways)  The one-workgroup kernels stream ~1.4 work units (rows + entries) per
    // nanosecond; an iteration over the whole GPU costs ~6 us of launches and reductions whatever the size: from ~10^5
    // units the grid wins per iteration — but it takes its problems one after the other where the one-workgroup kernels
    // take them side by side, so the threshold sits where ONE problem on one workgroup (>= 180 us per iteration) would
    // outlast a batch's other work.
    const char * env = std::getenv("RPVG_HIP_EM_GRID_MIN_WORK");
    return env ? std::strtoull(env, nullptr, 10) : (1ull << 18);
}

bool emGridDenseRoute(const uint32_t columns, const uint32_t rows, const uint32_t entries) {
    if (RPVG_EXPERIMENT_ENV("RPVG_HIP_EM_GRID_NO_DENSE")) return false;  // A/B knob
    return emDenseRule(columns, rows, entries);
}

int runEmGridProblems(rpvg_hip_ctx * ctx, hipStream_t st, const EmGridProblem * problems, const uint32_t count, const EmGridStorage & storage,
                      const uint32_t max_em_its, const double max_rel_em_conv) {
    const uint32_t cus = static_cast<uint32_t>(ctx->props.multiProcessorCount);
    std::vector<uint32_t> csr_route;  // the problems that stay on their CSR
    for (uint32_t i = 0; i < count; ++i) {
        const EmGridProblem & d = problems[i];
        const uint32_t p = d.problem, C = d.columns, rows = d.rows;
        const uint32_t * off = storage.prow_off + d.row_base + p;
        const double * cnt = ((d.merged && storage.merged_count) ? storage.merged_count : storage.prow_count) + d.row_base;
        const double * nz = storage.prow_noise + d.row_base;
        const uint32_t * col = storage.pent_col + d.ent_base;
        const double * val = storage.pent_val + d.ent_base;
        double * out_abundances = storage.abundances + d.col_begin;

        const uint64_t dense_ld = (static_cast<uint64_t>(C) + 1) & ~1ull;
        DeviceBuffer<double> d_matrix;
        const double * prebuilt = nullptr;  // (the compaction wrote the matrix itself: em_sparse.hip, the fused build)
        for (uint32_t f = 0; f < storage.num_fused; ++f) {
            if (storage.fused[f].problem == p && storage.fused[f].ld == dense_ld) prebuilt = storage.fused[f].matrix;
        }
        // (the dense copy is the faster route, not a needed one: without the memory for it the problem stays on its CSR)
        if (prebuilt || (emGridDenseRoute(C, rows, d.entries) && d_matrix.alloc(static_cast<size_t>(rows) * dense_ld) == hipSuccess)) {
            const uint64_t ld = dense_ld;
            if (!prebuilt) {
                const int span = ctx->spanBegin(FAM_BUILD, st);
                hipError_t build_error = hipMemsetAsync(d_matrix.ptr, 0, sizeof(double) * rows * ld, st);
                const uint32_t build_grid = static_cast<uint32_t>(std::min<uint64_t>((static_cast<uint64_t>(rows) + 3) / 4, static_cast<uint64_t>(cus) * 16));
                if (build_error == hipSuccess) {
                    emGridDenseBuildKernel<<<dim3(std::max(1u, build_grid)), dim3(kGridBlock), 0, st>>>(rows, C, ld, off, nz, col, val, d_matrix.ptr);
                    build_error = hipGetLastError();
                }
                ctx->spanEnd(span);  // (closed on the error path too)
                RPVG_HIP_CHECK(build_error);
                ctx->stats.build_launches += 1;
            }
            DenseEmRun run;
            run.matrix = prebuilt ? prebuilt : d_matrix.ptr;
            run.num_rows = rows;
            run.num_cols = C;
            run.ld = ld;
            run.counts = cnt;
            run.total_count = d.total_mass;
            run.zero_mass = d.zero_mass;
            run.max_em_its = max_em_its;
            run.max_rel_em_conv = max_rel_em_conv;
            if (const int rc = emDenseIterate(ctx, "rpvg_hip_em_solve (dense route of a large problem)", run)) return rc;
            DeviceBuffer<EmGridControl> d_ctl;
            RPVG_HIP_CHECK(d_ctl.upload(&run.control, 1, st));
            emGridFinishKernel<<<dim3(1), dim3(kGridBlock), 0, st>>>(C, run.d_a.ptr, d.total_mass, d_ctl.ptr, out_abundances, storage.noise_count + p,
                                                                     storage.iterations + p);
            RPVG_HIP_CHECK(hipGetLastError());
            RPVG_HIP_CHECK(waitStream(st));  // (the buffers of this problem go back to the pool)
            continue;
        }

        csr_route.push_back(i);
    }

    // ---- CSR route: two problems at a time, each on a stream of its own (a handful of mid-size problems of one solve — em_sparse.hip,
    // kEmMidGridMax — would otherwise wait for each other: an iteration is two short launches, and most of its time is their boundaries)
    if (csr_route.empty()) return RPVG_HIP_OK;
    hipError_t e = hipSuccess;
    for (int g = 0; g < 2 && e == hipSuccess; ++g) {
        if (!ctx->grid_stream[g]) e = hipStreamCreateWithFlags(&ctx->grid_stream[g], hipStreamNonBlocking);
    }
    if (e == hipSuccess && !ctx->grid_ready) e = hipEventCreateWithFlags(&ctx->grid_ready, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(ctx->grid_ready, st);  // (the problems' rows are in place behind what `st` holds so far)
    for (int g = 0; g < 2 && e == hipSuccess; ++g) e = hipStreamWaitEvent(ctx->grid_stream[g], ctx->grid_ready, 0);
    RPVG_HIP_CHECK(e);

    struct CsrRun {
        bool active = false;
        hipStream_t stream = nullptr;
        const EmGridProblem * d = nullptr;
        DeviceBuffer<double> d_a, d_partials;
        DeviceBuffer<EmGridControl> d_ctl;
        GridAccumArgs aa;
        GridUpdateArgs ua;
        uint32_t grid = 0, update_grid = 0, chunk_its = 0, queued = 0, chunk = 0;
        int row_lanes = 1, span = -1;
        size_t lds = 0;
        EmGridControl * h_ctl = nullptr;  // two pinned slots
        hipEvent_t looked[2] = {nullptr, nullptr};
        void release() {
            for (hipEvent_t & ev : looked) {
                if (ev) (void) hipEventDestroy(ev);
                ev = nullptr;
            }
            if (h_ctl) pinnedFree(h_ctl);
            h_ctl = nullptr;
            d_a.release();
            d_partials.release();
            d_ctl.release();
            active = false;
        }
        ~CsrRun() {
            if (active && stream) (void) hipStreamSynchronize(stream);
            release();
        }
    } runs[2];
    runs[0].stream = ctx->grid_stream[0];
    runs[1].stream = ctx->grid_stream[1];

    auto start = [&](CsrRun & r, const EmGridProblem & d) -> hipError_t {
        const uint32_t p = d.problem, C = d.columns, rows = d.rows;
        r.d = &d;
        r.row_lanes = gridRowLanes(rows, d.entries);
        // workgroups: enough to fill the GPU, few enough that the partial vectors (written and read once per iteration,
        // 16 B per column and workgroup) stay below the CSR's own bytes
        const uint64_t csr_bytes = 12ull * d.entries + 20ull * rows;
        const uint64_t slots = static_cast<uint64_t>(kGridBlock / r.row_lanes);
        uint64_t blocks = (static_cast<uint64_t>(rows) + slots - 1) / slots;
        blocks = std::min<uint64_t>(blocks, static_cast<uint64_t>(cus) * (r.row_lanes == 1 ? 2 : 4));
        blocks = std::min<uint64_t>(blocks, std::max<uint64_t>(16, csr_bytes / (16ull * C)));
        if (const char * env = RPVG_EXPERIMENT_ENV("RPVG_HIP_EM_GRID_BLOCKS")) blocks = std::max<uint64_t>(1, std::strtoull(env, nullptr, 10));  // A/B knob
        blocks = std::max<uint64_t>(1, std::min<uint64_t>(blocks, rows));
        const uint32_t rows_per_block = static_cast<uint32_t>((static_cast<uint64_t>(rows) + blocks - 1) / blocks);
        r.grid = (rows + rows_per_block - 1) / rows_per_block;
        const uint32_t partial_ld = (C + 63) & ~63u;
        r.lds = sizeof(double) * (1 + kGridBlock / 64) * static_cast<size_t>(C);
        hipError_t err = r.d_a.alloc(C);
        if (err == hipSuccess) err = r.d_partials.alloc(static_cast<size_t>(r.grid) * partial_ld);
        if (err == hipSuccess) err = r.d_ctl.alloc(1);
        if (err == hipSuccess) err = hipMemsetAsync(r.d_ctl.ptr, 0, sizeof(EmGridControl), r.stream);
        if (err != hipSuccess) return err;
        // src/path_abundance_estimator.cpp:54 — 1 / float(C), widened
        gridFillKernel<<<dim3((C + 255) / 256), dim3(256), 0, r.stream>>>(r.d_a.ptr, C, static_cast<double>(1.0f / static_cast<float>(C)));
        r.aa.off = storage.prow_off + d.row_base + p;
        r.aa.cnt = ((d.merged && storage.merged_count) ? storage.merged_count : storage.prow_count) + d.row_base;
        r.aa.nz = storage.prow_noise + d.row_base;
        r.aa.col = storage.pent_col + d.ent_base;
        r.aa.val = storage.pent_val + d.ent_base;
        r.aa.rows = rows;
        r.aa.C = C;
        r.aa.rows_per_block = rows_per_block;
        r.aa.a = r.d_a.ptr;
        r.aa.partials = r.d_partials.ptr;
        r.aa.partial_ld = partial_ld;
        r.aa.ctl = r.d_ctl.ptr;
        r.ua.C = C;
        r.ua.num_partials = r.grid;
        r.ua.partial_ld = partial_ld;
        r.ua.partials = r.d_partials.ptr;
        r.ua.a = r.d_a.ptr;
        r.ua.inv_total = 1.0 / d.total_mass;
        r.ua.zero_mass = d.zero_mass;
        r.ua.max_rel_em_conv = max_rel_em_conv;
        r.ua.max_em_its = max_em_its;
        r.ua.ctl = r.d_ctl.ptr;
        r.update_grid = (C + kUpdateColumns - 1) / kUpdateColumns;
        // Iterations in chunks, two chunks in flight: the control word of chunk k is looked at while chunk k + 1 runs.
        // A short problem's iteration is a few microseconds, a giant one's milliseconds: the chunk holds about half a
        // millisecond of the problem's streaming time at the HBM rate, 4 to 32 iterations.
        const double iteration_us = static_cast<double>(csr_bytes) / 4.0e6 + 8.0;
        r.chunk_its = static_cast<uint32_t>(std::min(32.0, std::max(4.0, 500.0 / iteration_us)));
        if (pinnedAlloc(reinterpret_cast<void **>(&r.h_ctl), 2 * sizeof(EmGridControl)) != hipSuccess) return hipErrorOutOfMemory;
        err = hipEventCreateWithFlags(&r.looked[0], hipEventDisableTiming);
        if (err == hipSuccess) err = hipEventCreateWithFlags(&r.looked[1], hipEventDisableTiming);
        r.queued = 0;
        r.chunk = 0;
        r.span = ctx->spanBegin(FAM_EM_KERNEL, r.stream, RPVG_HIP_EM_KERNELS - 1);
        r.active = true;
        return err;
    };
    // queues the next chunk of iterations of a run (nothing waits here)
    auto queueChunk = [&](CsrRun & r) -> hipError_t {
        const uint32_t n = std::min<uint32_t>(r.chunk_its, max_em_its - r.queued);
        hipError_t err = hipSuccess;
        for (uint32_t it = 0; it < n && err == hipSuccess; ++it) {
            err = launchAccum(r.row_lanes, r.aa, r.grid, r.lds, r.stream);
            emGridUpdateKernel<<<dim3(r.update_grid), dim3(kUpdateBlock), 0, r.stream>>>(r.ua);
        }
        r.queued += n;
        if (err == hipSuccess) err = hipGetLastError();
        if (err == hipSuccess) err = hipMemcpyAsync(&r.h_ctl[r.chunk & 1], r.d_ctl.ptr, sizeof(EmGridControl), hipMemcpyDeviceToHost, r.stream);
        if (err == hipSuccess) err = hipEventRecord(r.looked[r.chunk & 1], r.stream);
        return err;
    };
    // looks at the chunk before the one just queued; true: the run has stopped (finish kernel queued, stream waited for, buffers released)
    auto settle = [&](CsrRun & r, bool & finished) -> hipError_t {
        hipError_t err = hipSuccess;
        bool done = false;
        if (r.chunk > 0) {
            err = waitEvent(r.looked[(r.chunk - 1) & 1]);
            done = err == hipSuccess && r.h_ctl[(r.chunk - 1) & 1].done != 0;
        }
        if (err == hipSuccess && !done && r.queued >= max_em_its) {  // the last chunk there can be
            err = waitEvent(r.looked[r.chunk & 1]);
            done = true;
        }
        ++r.chunk;
        finished = done;
        if (err != hipSuccess || !done) return err;
        ctx->spanEnd(r.span);
        const EmGridProblem & d = *r.d;
        emGridFinishKernel<<<dim3(1), dim3(kGridBlock), 0, r.stream>>>(d.columns, r.d_a.ptr, d.total_mass, r.d_ctl.ptr, storage.abundances + d.col_begin,
                                                                       storage.noise_count + d.problem, storage.iterations + d.problem);
        err = hipGetLastError();
        const hipError_t waited = waitStream(r.stream);  // (the buffers of this problem go back to the pool)
        r.release();
        return err != hipSuccess ? err : waited;
    };

    size_t next = 0;
    while (e == hipSuccess && (next < csr_route.size() || runs[0].active || runs[1].active)) {
        for (CsrRun & r : runs) {
            if (e == hipSuccess && !r.active && next < csr_route.size()) e = start(r, problems[csr_route[next++]]);
        }
        for (CsrRun & r : runs) {
            if (e == hipSuccess && r.active) e = queueChunk(r);
        }
        for (CsrRun & r : runs) {
            bool finished = false;
            if (e == hipSuccess && r.active) e = settle(r, finished);
        }
    }
    if (e != hipSuccess) {
        for (CsrRun & r : runs) {
            if (r.active) (void) hipStreamSynchronize(r.stream);
        }
        setError("rpvg_hip_em_solve (problems over the whole GPU): %s", hipGetErrorString(e));
        return (e == hipErrorOutOfMemory) ? RPVG_HIP_ERR_ALLOC : RPVG_HIP_ERR_RUNTIME;
    }
    return RPVG_HIP_OK;
}

}  // namespace rpvg_hip_detail
