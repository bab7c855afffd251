// read_rows.hip — the step right before the inference hot path, on the GPU: the alignment paths of every read of a
// cluster -> the cluster's merged ReadPathProbabilities rows (interface include/rpvg_rows.h).
//
// Takes over   ReadPathProbabilities::addPathProbs / calcAlignPathLogProbs   src/read_path_probabilities.cpp:39-221
//              the caller's sort + quickMergeIdentical of adjacent rows      src/main.cpp:953-973
//              (operator<, quickMergeIdentical                               src/read_path_probabilities.cpp:223-322)
//
// One wavefront per read.  The reference fills a dense vector over ALL paths of the cluster per read; here only the
// paths the read touches exist:
//   A  per (alignment, path) entry: log prob = score * base (+ fragment log density) - log(effective length); of the
//      entries of one path the one with the longest alignment, then the highest log prob, survives (:128-139) —
//      found by binary search in the other alignments' ascending path lists, no scratch table;
//   B  survivors are ranked into ascending path order from the same sorted lists (prefix counts), optionally folded
//      into name groups (--path-info with -i transcripts, :151-168);
//   C  normalisation by a wave-wide log-sum-exp (the reference folds add_log sequentially: equal up to rounding);
//   D  the precision bucketing of :181-206 is inherently sequential in path order and is kept so: one step per path,
//      the lanes scan the buckets (LDS for the first 256, the read's scratch beyond);
//   E  buckets are ordered as std::sort orders pair<double, vector<uint32_t>> (:219) and written to the read's padded
//      slice; a scan + copy pass packs all reads into the CSR of rpvg_cluster_batch.
// Merging: row ids are merge-sorted with the reference's own tolerant operator< behind the cluster id, runs of rows
// that quickMergeIdentical accepts are found with the reference's compare-with-the-run-head rule, counts are summed.

#include <cfloat>
#include <cmath>
#include <cstring>
#include <memory>

#include <hipcub/hipcub.hpp>

#include "../../include/rpvg_rows.h"
#include "common.hpp"

using namespace rpvg_hip_detail;

// Device-resident alignment batch (validated copy of a rpvg_alignment_batch).
struct rpvg_hip_alignments {
    uint32_t num_clusters = 0;
    uint64_t num_reads = 0, num_aligns = 0, num_entries = 0, num_paths = 0;
    bool collapse = false;
    std::vector<uint64_t> h_cluster_read_off;  // [K+1]
    std::vector<uint64_t> h_out_path_off;      // [K+1] output columns of each cluster (paths, or name groups)
    rpvg_hip_detail::DeviceBuffer<uint32_t> read_cluster, read_count, source_count, path_group, path_idx;
    rpvg_hip_detail::DeviceBuffer<uint32_t> small_reads, large_reads;  // at most / more than 16 (alignment, path) entries
    uint64_t num_small = 0, num_large = 0;
    rpvg_hip_detail::DeviceBuffer<uint64_t> cluster_path_off, cluster_read_off, read_align_off, align_path_off;
    rpvg_hip_detail::DeviceBuffer<double> eff_len;
    rpvg_hip_detail::DeviceBuffer<uint8_t> mapq;
    rpvg_hip_detail::DeviceBuffer<int32_t> noise_score, score;
    rpvg_hip_detail::DeviceBuffer<uint16_t> align_length, frag_length;
};

// Device-resident rows in the grouped layout of rpvg_cluster_batch; host copies are made on demand (view).
struct rpvg_hip_read_rows {
    uint32_t num_clusters = 0;
    uint64_t num_rows = 0, num_groups = 0, num_members = 0;
    std::vector<uint64_t> h_cluster_row_off, h_cluster_path_off;
    rpvg_hip_detail::DeviceBuffer<uint64_t> cluster_row_off, row_grp_off, grp_idx_off;
    rpvg_hip_detail::DeviceBuffer<uint32_t> row_count, path_idx;
    rpvg_hip_detail::DeviceBuffer<double> row_noise, grp_prob;
    // host copies (rpvg_hip_read_rows_view)
    bool downloaded = false;
    std::vector<uint64_t> h_row_grp_off, h_grp_idx_off;
    std::vector<uint32_t> h_row_count, h_path_idx;
    std::vector<double> h_row_noise, h_grp_prob;
    double build_ms = 0, merge_ms = 0;
};

namespace {

constexpr double kScoreLogBase = 1.383325268738;  // Utils::score_log_base, src/utils.hpp:83
constexpr double kNoiseScoreLogBase = 1e-6;       // Utils::noise_score_log_base, src/utils.hpp:84
constexpr double kDoublePrecision = 2.220446049250313e-16 * 100;  // Utils::double_precision, src/utils.hpp:81
constexpr uint32_t kNone = 0xffffffffu;
constexpr int kLdsBuckets = 256;
constexpr int kWavesPerBlock = 4;

struct RowsIn {
    uint64_t num_reads;
    const uint32_t * read_cluster;
    const uint64_t * cluster_path_off;
    const double * path_eff_len;
    const uint32_t * path_source_count;
    const uint32_t * path_group;  // null = no collapsing
    const uint8_t * read_min_mapq;
    const int32_t * read_noise_score;
    const uint64_t * read_align_off;
    const int32_t * align_score_sum;
    const uint16_t * align_length;
    const uint16_t * align_frag_length;
    const uint64_t * align_path_off;
    const uint32_t * align_path_idx;
    const double * frag_table;  // null = single end
    const double * phred_prob;  // [256] 10^(-q/10)
    double prob_precision, min_noise_prob;
};

// per-read padded slices: entry e of the read's (alignment, path) pairs owns slot e of every array
struct RowsScratch {
    double * align_log_prob;    // [A]
    uint32_t * surv_prefix;     // [E + N] exclusive count of survivors before an entry (slot e + read index)
    uint32_t * unit_idx;        // [E] output column of unit t (path, or name group)
    double * unit_val;          // [E] log prob, then probability of unit t
    uint32_t * tmp_idx;         // [E] (collapse) path-sorted survivors
    double * tmp_val;           // [E]
    uint32_t * bucket_of;       // [E] bucket of unit t
    double * bucket_mean;       // [E] buckets beyond the LDS capacity
    uint32_t * bucket_count;    // [E]
    uint32_t * bucket_first;    // [E] first unit (t) of bucket b
    uint32_t * bucket_rank;     // [E] position of bucket b in the sorted output
    uint32_t * bucket_cursor;   // [E]
    // outputs, padded
    double * row_noise;         // [N]
    uint32_t * row_ngroups;     // [N]
    uint32_t * row_nmembers;    // [N]
    double * grp_prob;          // [E] sorted groups of the read
    uint32_t * grp_size;        // [E]
    uint32_t * grp_moff;        // [E] exclusive member offset of the group inside the read
    uint32_t * member;          // [E]
};

__device__ __forceinline__ bool doubleCompareDev(const double a, const double b) {
    return (a == b) || (fabs(a - b) < fabs(fmin(a, b)) * kDoublePrecision);
}

__device__ __forceinline__ double waveMaxF64(double v) {
    for (int d = 32; d >= 1; d >>= 1) v = fmax(v, __shfl_xor(v, d, 64));
    return v;
}

// first position in [lo, hi) of the ascending list whose value is >= p
__device__ __forceinline__ uint64_t lowerBound(const uint32_t * __restrict__ list, uint64_t lo, uint64_t hi, const uint32_t p) {
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (list[mid] < p) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ void alignLogProbKernel(const uint64_t num_aligns, const int32_t * __restrict__ score_sum,
                                   const uint16_t * __restrict__ frag_length, const double * __restrict__ frag_table,
                                   double * __restrict__ out) {
    const uint64_t a = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (a >= num_aligns) return;
    // src/read_path_probabilities.cpp:56-61
    double lp = score_sum[a] * kScoreLogBase;
    if (frag_table) lp += frag_table[frag_length[a]];
    out[a] = lp;
}

__device__ __forceinline__ double addLogDev(const double x, const double y) {  // Utils::add_log, src/utils.hpp:300-302
    return x > y ? x + log1p(exp(y - x)) : y + log1p(exp(x - y));
}

__global__ __launch_bounds__(64 * kWavesPerBlock) void readRowKernel(const RowsIn in, const RowsScratch sc,
                                                                    const uint64_t num_listed, const uint32_t * __restrict__ listed) {
    __shared__ double lds_mean[kWavesPerBlock][kLdsBuckets];
    __shared__ uint32_t lds_count[kWavesPerBlock][kLdsBuckets];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const uint64_t slot = blockIdx.x * static_cast<uint64_t>(kWavesPerBlock) + wave;
    if (slot >= num_listed) return;
    const uint64_t r = listed ? listed[slot] : slot;
    volatile double * wmean = lds_mean[wave];
    volatile uint32_t * wcount = lds_count[wave];

    const uint64_t a0 = in.read_align_off[r], a1 = in.read_align_off[r + 1];
    const uint64_t e0 = in.align_path_off[a0], e1 = in.align_path_off[a1];
    const uint32_t E = static_cast<uint32_t>(e1 - e0);
    const uint8_t mapq = in.read_min_mapq[r];
    const int32_t noise_score = in.read_noise_score[r];

    // src/read_path_probabilities.cpp:89-108
    double noise = 1.0;
    bool has_paths = false;
    if (mapq > 0) {
        noise = fmax(in.prob_precision, fmax(in.min_noise_prob, in.phred_prob[mapq]));
        noise += (1 - noise) * exp(noise_score * kNoiseScoreLogBase);
        has_paths = (noise_score != 0);
    }
    if (!has_paths) {
        if (lane == 0) {
            sc.row_noise[r] = noise;
            sc.row_ngroups[r] = 0;
            sc.row_nmembers[r] = 0;
        }
        return;
    }

    const uint32_t k = in.read_cluster[r];
    const uint64_t cp0 = in.cluster_path_off[k];
    const uint32_t * __restrict__ plist = in.align_path_idx;
    uint32_t * S = sc.surv_prefix + r;  // slot of entry e: S[e]

    // ---- A: survivors (:110-149) and their exclusive prefix count -------------------------------------------
    uint32_t num_surv = 0;
    for (uint32_t base = 0; base < E; base += 64) {
        const uint32_t i = base + lane;
        bool keep = false;
        if (i < E) {
            const uint64_t e = e0 + i;
            uint64_t lo = a0, hi = a1 - 1;  // last alignment whose first entry is <= e
            while (lo < hi) {
                const uint64_t mid = (lo + hi + 1) >> 1;
                if (in.align_path_off[mid] <= e) lo = mid; else hi = mid - 1;
            }
            const uint64_t a = lo;
            const uint32_t p = plist[e];
            keep = in.path_eff_len[cp0 + p] != 0;  // Utils::doubleCompare(x, 0) holds for x == 0 only
            if (keep) {
                const uint16_t al = in.align_length[a];
                const double alp = sc.align_log_prob[a];
                for (uint64_t b = a0; b < a1 && keep; ++b) {
                    if (b == a) continue;
                    const uint64_t b0 = in.align_path_off[b], b1 = in.align_path_off[b + 1];
                    const uint64_t pos = lowerBound(plist, b0, b1, p);
                    if (pos < b1 && plist[pos] == p) {
                        const uint16_t bl = in.align_length[b];
                        const double blp = sc.align_log_prob[b];
                        if (bl > al || (bl == al && (blp > alp || (blp == alp && b < a)))) keep = false;
                    }
                }
            }
        }
        const uint64_t mask = __ballot(keep);
        if (i < E) S[e0 + i] = num_surv + __popcll(mask & ((1ull << lane) - 1ull));
        num_surv += __popcll(mask);
    }
    if (lane == 0) S[e1] = num_surv;
    __threadfence_block();

    // ---- B: survivors in ascending path order ---------------------------------------------------------------
    const bool collapse = in.path_group != nullptr;
    uint32_t * sorted_idx = collapse ? sc.tmp_idx + e0 : sc.unit_idx + e0;
    double * sorted_val = collapse ? sc.tmp_val + e0 : sc.unit_val + e0;
    for (uint32_t base = 0; base < E; base += 64) {
        const uint32_t i = base + lane;
        if (i >= E) continue;
        const uint64_t e = e0 + i;
        if (S[e + 1] == S[e]) continue;  // not a survivor
        const uint32_t p = plist[e];
        uint32_t rank = 0;
        uint64_t own = a0;
        for (uint64_t b = a0; b < a1; ++b) {
            const uint64_t b0 = in.align_path_off[b], b1 = in.align_path_off[b + 1];
            if (e >= b0 && e < b1) own = b;
            const uint64_t pos = (e >= b0 && e < b1) ? e : lowerBound(plist, b0, b1, p);
            rank += S[pos] - S[b0];
        }
        sorted_idx[rank] = p;
        sorted_val[rank] = sc.align_log_prob[own] - log(in.path_eff_len[cp0 + p]);  // :126
    }
    __threadfence_block();

    uint32_t T = num_surv;
    if (collapse && T > 0) {
        // :151-168 — a name group's log prob = add_log, in path order, of member log prob + log(source count);
        // units leave in ascending group index.  Quadratic in the paths the read touches (this mode is
        // `-i transcripts` with --path-info only).
        uint32_t * is_head = sc.bucket_of + e0;  // free until phase D
        for (uint32_t t = lane; t < T; t += 64) {
            const uint32_t g = in.path_group[cp0 + sorted_idx[t]];
            bool head = true;
            for (uint32_t u = 0; u < t && head; ++u) head = in.path_group[cp0 + sorted_idx[u]] != g;
            is_head[t] = head;
        }
        __threadfence_block();
        uint32_t num_units = 0;
        for (uint32_t base = 0; base < T; base += 64) {
            const uint32_t t = base + lane;
            const bool head = (t < T) && is_head[t];
            if (head) {
                const uint32_t g = in.path_group[cp0 + sorted_idx[t]];
                uint32_t rank = 0;
                double glp = -DBL_MAX;
                for (uint32_t u = 0; u < T; ++u) {
                    const uint32_t pu = sorted_idx[u];
                    const uint32_t gu = in.path_group[cp0 + pu];
                    if (gu == g) glp = addLogDev(glp, sorted_val[u] + log(static_cast<double>(in.path_source_count[cp0 + pu])));
                    rank += (gu < g) && is_head[u];
                }
                sc.unit_idx[e0 + rank] = g;
                sc.unit_val[e0 + rank] = glp;
            }
            num_units += __popcll(__ballot(head));
        }
        T = num_units;
        __threadfence_block();
    }

    uint32_t * uidx = sc.unit_idx + e0;
    double * uval = sc.unit_val + e0;

    if (T == 0) {  // every touched path has zero effective length: the reference would assert (:174)
        if (lane == 0) {
            sc.row_noise[r] = noise;
            sc.row_ngroups[r] = 0;
            sc.row_nmembers[r] = 0;
        }
        return;
    }

    // ---- C: normalise (:170-179) -------------------------------------------------------------------------------
    double vmax = -DBL_MAX;
    for (uint32_t t = lane; t < T; t += 64) vmax = fmax(vmax, uval[t]);
    vmax = waveMaxF64(vmax);
    double vsum = 0.0;
    for (uint32_t t = lane; t < T; t += 64) vsum += exp(uval[t] - vmax);
    vsum = waveSumF64(vsum);
    const double log_sum = vmax + log(vsum);
    for (uint32_t t = lane; t < T; t += 64) uval[t] = exp(uval[t] - log_sum);
    __threadfence_block();

    // ---- D: precision buckets, sequential in unit order (:181-211) ---------------------------------------------
    uint32_t nb = 0;
    double low_sum = 0.0;
    for (uint32_t t = 0; t < T; ++t) {
        const double p = uval[t];
        if (p >= in.prob_precision) {
            uint32_t found = kNone;
            for (uint32_t base = 0; base < nb && found == kNone; base += 64) {
                const uint32_t b = base + lane;
                bool hit = false;
                if (b < nb) {
                    const double m = (b < kLdsBuckets) ? wmean[b] : __hip_atomic_load(sc.bucket_mean + e0 + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    hit = fabs(m - p) < in.prob_precision;
                }
                const uint64_t mask = __ballot(hit);
                if (mask) found = base + __ffsll(static_cast<unsigned long long>(mask)) - 1;
            }
            if (lane == 0) {
                if (found == kNone) {
                    if (nb < kLdsBuckets) {
                        wmean[nb] = p;
                        wcount[nb] = 1;
                    } else {
                        __hip_atomic_store(sc.bucket_mean + e0 + nb, p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        sc.bucket_count[e0 + nb] = 1;
                    }
                    sc.bucket_first[e0 + nb] = t;
                    sc.bucket_of[e0 + t] = nb;
                } else {
                    // running mean (:189)
                    if (found < kLdsBuckets) {
                        const uint32_t c = wcount[found];
                        wmean[found] = (wmean[found] * c + p) / (c + 1);
                        wcount[found] = c + 1;
                    } else {
                        const uint32_t c = sc.bucket_count[e0 + found];
                        const double m = __hip_atomic_load(sc.bucket_mean + e0 + found, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(sc.bucket_mean + e0 + found, (m * c + p) / (c + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        sc.bucket_count[e0 + found] = c + 1;
                    }
                    sc.bucket_of[e0 + t] = found;
                }
            }
            if (found == kNone) ++nb;
        } else {
            low_sum += p;
            if (lane == 0) sc.bucket_of[e0 + t] = kNone;
        }
    }
    __threadfence_block();

    // ---- E: scale, order the buckets, write the read's slice (:213-219) ----------------------------------------
    const double scale = 1 - noise;
    auto bucketMean = [&](const uint32_t b) -> double {
        return ((b < kLdsBuckets) ? wmean[b] : __hip_atomic_load(sc.bucket_mean + e0 + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) * scale;
    };
    auto bucketCount = [&](const uint32_t b) -> uint32_t { return (b < kLdsBuckets) ? wcount[b] : sc.bucket_count[e0 + b]; };
    for (uint32_t base = 0; base < nb; base += 64) {
        const uint32_t b = base + lane;
        if (b >= nb) continue;
        const double m = bucketMean(b);
        const uint32_t first = uidx[sc.bucket_first[e0 + b]];
        uint32_t rank = 0;
        for (uint32_t o = 0; o < nb; ++o) {
            if (o == b) continue;
            const double mo = bucketMean(o);
            // pair<double, vector> ordering; member lists are disjoint, so their first elements decide ties
            if (mo < m || (mo == m && uidx[sc.bucket_first[e0 + o]] < first)) ++rank;
        }
        sc.bucket_rank[e0 + b] = rank;
        sc.grp_prob[e0 + rank] = m;
        sc.grp_size[e0 + rank] = bucketCount(b);
        sc.bucket_cursor[e0 + b] = 0;
    }
    __threadfence_block();
    // exclusive member offsets of the sorted groups
    uint32_t running = 0;
    for (uint32_t base = 0; base < nb; base += 64) {
        const uint32_t j = base + lane;
        const uint32_t size = (j < nb) ? sc.grp_size[e0 + j] : 0;
        uint32_t incl = size;
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d, 64);
            if (lane >= d) incl += up;
        }
        if (j < nb) sc.grp_moff[e0 + j] = running + incl - size;
        running += __shfl(incl, 63, 64);
    }
    __threadfence_block();
    // members in ascending unit order inside each group
    if (nb <= 8) {
        for (uint32_t b = 0; b < nb; ++b) {
            const uint32_t moff = sc.grp_moff[e0 + sc.bucket_rank[e0 + b]];
            uint32_t placed = 0;
            for (uint32_t base = 0; base < T; base += 64) {
                const uint32_t t = base + lane;
                const bool mine = (t < T) && sc.bucket_of[e0 + t] == b;
                const uint64_t mask = __ballot(mine);
                if (mine) sc.member[e0 + moff + placed + __popcll(mask & ((1ull << lane) - 1ull))] = uidx[t];
                placed += __popcll(mask);
            }
        }
    } else if (lane == 0) {
        for (uint32_t t = 0; t < T; ++t) {
            const uint32_t b = sc.bucket_of[e0 + t];
            if (b == kNone) continue;
            const uint32_t pos = sc.grp_moff[e0 + sc.bucket_rank[e0 + b]] + sc.bucket_cursor[e0 + b]++;
            sc.member[e0 + pos] = uidx[t];
        }
    }
    if (lane == 0) {
        sc.row_noise[r] = noise + low_sum * scale;  // :216
        sc.row_ngroups[r] = nb;
        sc.row_nmembers[r] = running;
    }
}


// ---- small reads: sixteen lanes per read -----------------------------------------------------------------------
// Most reads touch a handful of paths (3.3 (alignment, path) entries on average in the configs[2] workload); a whole
// wavefront per read then spends its time waiting on the scratch arrays between phases.  A read with at most 16 entries
// is handled by a 16-lane group — four reads per wavefront — entirely in registers: lane = entry, then lane = unit,
// then lane = bucket; the phases talk through width-16 shuffles and ballots.  Same arithmetic, same order of the
// sequential bucketing, same output slices as readRowKernel (which keeps the larger reads and the collapsing mode).
constexpr int kGroupLanes = 16;

__device__ __forceinline__ uint32_t groupBallot(const bool pred, const int group_shift) {
    return static_cast<uint32_t>((__ballot(pred) >> group_shift) & 0xffffull);
}

__global__ __launch_bounds__(256) void readRowSmallKernel(const RowsIn in, const RowsScratch sc, const uint64_t num_listed,
                                                          const uint32_t * __restrict__ listed) {
    const int g = threadIdx.x & (kGroupLanes - 1);                  // lane inside the group
    const int group_shift = (threadIdx.x & 63) & ~(kGroupLanes - 1);  // first lane of the group inside the wave
    const uint64_t slot = (blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x) / kGroupLanes;
    const bool active = slot < num_listed;
    const uint64_t r = active ? listed[slot] : 0;

    const uint64_t a0 = in.read_align_off[r], a1 = in.read_align_off[r + 1];
    const uint64_t e0 = in.align_path_off[a0], e1 = in.align_path_off[a1];
    const uint32_t E = active ? static_cast<uint32_t>(e1 - e0) : 0;  // <= kGroupLanes
    const uint8_t mapq = in.read_min_mapq[r];
    const int32_t noise_score = in.read_noise_score[r];

    // src/read_path_probabilities.cpp:89-108
    double noise = 1.0;
    bool has_paths = false;
    if (active && mapq > 0) {
        noise = fmax(in.prob_precision, fmax(in.min_noise_prob, in.phred_prob[mapq]));
        noise += (1 - noise) * exp(noise_score * kNoiseScoreLogBase);
        has_paths = (noise_score != 0);
    }
    // every lane of the wave takes part in the shuffles below: groups without work carry E = 0
    const uint32_t n_ent = has_paths ? E : 0;
    // The loops over entries / units / buckets stop at the largest entry count of the wave's four reads (3.3 entries
    // per read on average, 16 at most): wave-uniform trip counts, a fifth of the shuffles.
    uint32_t n_max = n_ent;
    n_max = max(n_max, static_cast<uint32_t>(__shfl_xor(static_cast<int>(n_max), 16, 64)));
    n_max = max(n_max, static_cast<uint32_t>(__shfl_xor(static_cast<int>(n_max), 32, 64)));
    const int n_loop = static_cast<int>(__builtin_amdgcn_readfirstlane(static_cast<int>(n_max)));

    // ---- A: one entry per lane; the survivor of every path (:110-149) -------------------------------------------
    const bool is_entry = static_cast<uint32_t>(g) < n_ent;
    uint32_t p = kNone;
    uint32_t al = 0;
    double alp = 0.0, lp = 0.0;
    uint32_t a_idx = 0;
    bool keep = false;
    if (is_entry) {
        const uint64_t e = e0 + g;
        uint64_t a = a0;
        while (a + 1 < a1 && in.align_path_off[a + 1] <= e) ++a;
        a_idx = static_cast<uint32_t>(a - a0);
        p = in.align_path_idx[e];
        al = in.align_length[a];
        alp = sc.align_log_prob[a];
        const double len = in.path_eff_len[in.cluster_path_off[in.read_cluster[r]] + p];
        keep = len != 0;
        if (keep) lp = alp - log(len);  // :126
    }
    for (int j = 0; j < n_loop; ++j) {
        const uint32_t pj = __shfl(p, j, kGroupLanes);
        const uint32_t alj = __shfl(al, j, kGroupLanes);
        const double alpj = __shfl(alp, j, kGroupLanes);
        const uint32_t aj = __shfl(a_idx, j, kGroupLanes);
        if (is_entry && j != g && static_cast<uint32_t>(j) < n_ent && pj == p) {
            if (alj > al || (alj == al && (alpj > alp || (alpj == alp && aj < a_idx)))) keep = false;
        }
    }
    // ---- B: ascending path order ---------------------------------------------------------------------------------
    uint32_t rank = 0;
    for (int j = 0; j < n_loop; ++j) {
        const uint32_t pj = __shfl(p, j, kGroupLanes);
        const bool kj = __shfl(static_cast<int>(keep), j, kGroupLanes) != 0;
        rank += (kj && pj < p);
    }
    const uint32_t T = __popc(groupBallot(keep, group_shift));
    // unit t lives on lane t: pull (path, log prob) from the survivor whose rank is t
    uint32_t uidx = kNone;
    double uval = -DBL_MAX;
    for (int j = 0; j < n_loop; ++j) {
        const bool kj = __shfl(static_cast<int>(keep), j, kGroupLanes) != 0;
        const uint32_t rj = __shfl(rank, j, kGroupLanes);
        const uint32_t pj = __shfl(p, j, kGroupLanes);
        const double lpj = __shfl(lp, j, kGroupLanes);
        if (kj && rj == static_cast<uint32_t>(g)) {
            uidx = pj;
            uval = lpj;
        }
    }
    const bool is_unit = static_cast<uint32_t>(g) < T;

    // ---- C: normalise (:170-179) ---------------------------------------------------------------------------------
    double vmax = uval;
    for (int d = kGroupLanes / 2; d >= 1; d >>= 1) vmax = fmax(vmax, __shfl_xor(vmax, d, kGroupLanes));
    double vsum = is_unit ? exp(uval - vmax) : 0.0;
    for (int d = kGroupLanes / 2; d >= 1; d >>= 1) vsum += __shfl_xor(vsum, d, kGroupLanes);
    const double prob = is_unit ? exp(uval - (vmax + log(vsum))) : 0.0;

    // ---- D: precision buckets, sequential in unit order (:181-211); bucket b lives on lane b ----------------------
    uint32_t nb = 0, my_bucket = kNone, b_count = 0, b_first = 0;
    double b_mean = 0.0, low_sum = 0.0;
    for (int t = 0; t < n_loop; ++t) {
        const double pt = __shfl(prob, t, kGroupLanes);
        const bool exists = static_cast<uint32_t>(t) < T;  // uniform inside a group
        const bool big = exists && pt >= in.prob_precision;
        const bool hit = big && static_cast<uint32_t>(g) < nb && fabs(b_mean - pt) < in.prob_precision;
        const uint32_t mask = groupBallot(hit, group_shift);
        const uint32_t found = mask ? static_cast<uint32_t>(__ffs(mask) - 1) : nb;
        if (big) {
            if (static_cast<uint32_t>(g) == found) {
                if (mask) {
                    b_mean = (b_mean * b_count + pt) / (b_count + 1);  // running mean (:189)
                    b_count += 1;
                } else {
                    b_mean = pt;
                    b_count = 1;
                    b_first = t;
                }
            }
            if (g == t) my_bucket = found;
            if (!mask) ++nb;
        } else if (exists) {
            low_sum += pt;
        }
    }

    // ---- E: scale, order the buckets, write the read's slice (:213-219) -----------------------------------------
    const double scale = 1 - noise;
    const bool is_bucket = static_cast<uint32_t>(g) < nb;
    const double m = b_mean * scale;
    const uint32_t first_member = __shfl(uidx, static_cast<int>(b_first), kGroupLanes);
    uint32_t brank = 0, moff = 0;
    for (int o = 0; o < n_loop; ++o) {
        const double mo = __shfl(m, o, kGroupLanes);
        const uint32_t fo = __shfl(first_member, o, kGroupLanes);
        if (static_cast<uint32_t>(o) < nb && o != g && (mo < m || (mo == m && fo < first_member))) ++brank;
    }
    for (int o = 0; o < n_loop; ++o) {
        const uint32_t ro = __shfl(brank, o, kGroupLanes);
        const uint32_t co = __shfl(b_count, o, kGroupLanes);
        if (static_cast<uint32_t>(o) < nb && ro < brank) moff += co;
    }
    if (is_bucket) {
        sc.grp_prob[e0 + brank] = m;
        sc.grp_size[e0 + brank] = b_count;
        sc.grp_moff[e0 + brank] = moff;
    }
    // members: unit t goes behind the earlier units of its bucket
    const uint32_t bucket_moff = __shfl(moff, static_cast<int>(my_bucket == kNone ? 0 : my_bucket), kGroupLanes);
    uint32_t before = 0;
    for (int t = 0; t < n_loop; ++t) {
        const uint32_t bt = __shfl(my_bucket, t, kGroupLanes);
        if (t < g && bt == my_bucket) ++before;
    }
    if (is_unit && my_bucket != kNone) sc.member[e0 + bucket_moff + before] = uidx;
    const uint32_t n_members = __popc(groupBallot(is_unit && my_bucket != kNone, group_shift));
    if (active && g == 0) {
        sc.row_noise[r] = has_paths && T > 0 ? noise + low_sum * scale : noise;  // :216
        sc.row_ngroups[r] = nb;
        sc.row_nmembers[r] = n_members;
    }
}

// packs the padded per-read slices into the CSR of rpvg_cluster_batch; sixteen lanes per row.  `source` (optional)
// lists the reads to copy (the merged rows); row j of the output comes from read source[j].
__global__ __launch_bounds__(256) void packRowsKernel(const uint64_t num_rows, const uint32_t * __restrict__ source,
                                                      const uint64_t * __restrict__ read_align_off,
                                                      const uint64_t * __restrict__ align_path_off, const RowsScratch sc,
                                                      const uint64_t * __restrict__ row_grp_off,
                                                      const uint64_t * __restrict__ row_member_off, double * __restrict__ row_noise,
                                                      double * __restrict__ grp_prob, uint64_t * __restrict__ grp_idx_off,
                                                      uint32_t * __restrict__ path_idx) {
    const int lane = threadIdx.x & (kGroupLanes - 1);  // sixteen lanes per row: most rows have a few groups
    const uint64_t j = (blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x) / kGroupLanes;
    if (j >= num_rows) return;
    const uint64_t r = source ? source[j] : j;
    const uint64_t e0 = align_path_off[read_align_off[r]];
    const uint64_t g0 = row_grp_off[j], m0 = row_member_off[j];
    const uint32_t nb = sc.row_ngroups[r], nm = sc.row_nmembers[r];
    for (uint32_t b = lane; b < nb; b += kGroupLanes) {
        grp_prob[g0 + b] = sc.grp_prob[e0 + b];
        grp_idx_off[g0 + b] = m0 + sc.grp_moff[e0 + b];
    }
    for (uint32_t m = lane; m < nm; m += kGroupLanes) path_idx[m0 + m] = sc.member[e0 + m];
    if (lane == 0) {
        row_noise[j] = sc.row_noise[r];
        if (j + 1 == num_rows) grp_idx_off[g0 + nb] = m0 + nm;
    }
}

// ---- merging ------------------------------------------------------------------------------------------------
// The caller of the reference sorts a cluster's rows with ReadPathProbabilities::operator< and merges each row into the
// head of the current run while quickMergeIdentical accepts it (src/main.cpp:953-973).  operator< compares doubles with
// Utils::doubleCompare (relative 2.2e-14) so that rows that agree up to rounding — reads whose alignments differ by
// a common score shift — become neighbours.  Such a comparison is not a strict weak order when values chain
// (a ~ b, b ~ c, a < c), which the sub-precision mass moved into the noise term does produce; merge-path sorts may
// then duplicate elements.  The rows of a cluster are therefore sorted with a bitonic NETWORK — data-oblivious
// compare-exchanges always yield a permutation — using the reference's comparison literally: one workgroup per
// cluster in LDS up to 2048 rows, global-memory stages for larger clusters.  Where the comparison is consistent the
// order equals the reference's (up to ties, which merge anyway); where it is not, the reference's own order is
// whatever std::sort happens to produce.

struct RowView {
    const uint32_t * read_cluster;
    const uint32_t * read_count;
    const uint64_t * read_align_off;
    const uint64_t * align_path_off;
    RowsScratch sc;
    double prob_precision;
};

// ReadPathProbabilities::operator<, src/read_path_probabilities.cpp:283-322 (+ the row id as last key: deterministic)
struct RowLess {
    RowView v;
    __device__ bool operator()(const uint32_t lhs, const uint32_t rhs) const {
        const double nl = v.sc.row_noise[lhs], nr = v.sc.row_noise[rhs];
        if (!doubleCompareDev(nl, nr)) return nl < nr;
        const uint32_t bl = v.sc.row_ngroups[lhs], br = v.sc.row_ngroups[rhs];
        if (bl != br) return bl < br;
        const uint64_t el = v.align_path_off[v.read_align_off[lhs]], er = v.align_path_off[v.read_align_off[rhs]];
        for (uint32_t i = 0; i < bl; ++i) {
            const double pl = v.sc.grp_prob[el + i], pr = v.sc.grp_prob[er + i];
            if (!doubleCompareDev(pl, pr)) return pl < pr;
            const uint32_t sl = v.sc.grp_size[el + i], sr = v.sc.grp_size[er + i];
            if (sl != sr) return sl < sr;
            const uint32_t ml = v.sc.grp_moff[el + i], mr = v.sc.grp_moff[er + i];
            for (uint32_t j = 0; j < sl; ++j) {
                const uint32_t xl = v.sc.member[el + ml + j], xr = v.sc.member[er + mr + j];
                if (xl != xr) return xl < xr;
            }
        }
        const uint32_t tl = v.read_count[lhs], tr = v.read_count[rhs];
        if (tl != tr) return tl < tr;
        return lhs < rhs;
    }
};

// ReadPathProbabilities::quickMergeIdentical without the count update (src/read_path_probabilities.cpp:223-250)
__device__ bool mergeable(const RowView & v, const uint32_t a, const uint32_t b) {
    if (fabs(v.sc.row_noise[a] - v.sc.row_noise[b]) >= v.prob_precision) return false;
    const uint32_t nb = v.sc.row_ngroups[a];
    if (nb != v.sc.row_ngroups[b]) return false;
    const uint64_t ea = v.align_path_off[v.read_align_off[a]], eb = v.align_path_off[v.read_align_off[b]];
    for (uint32_t i = 0; i < nb; ++i) {
        if (fabs(v.sc.grp_prob[ea + i] - v.sc.grp_prob[eb + i]) >= v.prob_precision) return false;
        const uint32_t size = v.sc.grp_size[ea + i];
        if (size != v.sc.grp_size[eb + i]) return false;
        const uint32_t ma = v.sc.grp_moff[ea + i], mb = v.sc.grp_moff[eb + i];
        for (uint32_t j = 0; j < size; ++j) {
            if (v.sc.member[ea + ma + j] != v.sc.member[eb + mb + j]) return false;
        }
    }
    return true;
}

__global__ void iotaKernel(const uint64_t n, uint32_t * out) {
    const uint64_t i = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (i < n) out[i] = static_cast<uint32_t>(i);
}

constexpr uint32_t kSmallSort = 2048;

// Pair t of step (k, j) of the all-ascending bitonic network: the first step of a merge of width k pairs i with its
// mirror image inside the block of k, the later steps pair i with i + j.
__device__ __forceinline__ void bitonicPair(const uint32_t t, const uint32_t k, const uint32_t j, uint32_t * i, uint32_t * l) {
    if (j == (k >> 1)) {
        const uint32_t block = t / j, off = t % j;
        *i = block * k + off;
        *l = block * k + (k - 1 - off);
    } else {
        *i = 2 * j * (t / j) + (t % j);
        *l = *i + j;
    }
}

// The positions of a tile in LDS: row id plus the row's first sort keys (noise, number of groups, probability of the
// first group), which travel with the id.  RowLess reads its keys from global memory through a chain of dependent
// loads (row -> alignment offsets -> groups); most comparisons are decided by the first keys, which are the same
// values, so the outcome is RowLess's.
struct SortTile {
    uint32_t * id;
    uint32_t * ngroups;
    double * noise;
    double * prob0;

    __device__ void load(const RowLess & less, const uint32_t pos, const uint32_t row) {
        id[pos] = row;
        if (row == kNone) return;
        const uint32_t nb = less.v.sc.row_ngroups[row];
        noise[pos] = less.v.sc.row_noise[row];
        ngroups[pos] = nb;
        prob0[pos] = nb ? less.v.sc.grp_prob[less.v.align_path_off[less.v.read_align_off[row]]] : 0.0;
    }

    // is the row at position l before the row at position i?
    __device__ bool before(const RowLess & less, const uint32_t l, const uint32_t i) const {
        const double nl = noise[l], nr = noise[i];
        if (!doubleCompareDev(nl, nr)) return nl < nr;
        const uint32_t bl = ngroups[l], br = ngroups[i];
        if (bl != br) return bl < br;
        if (bl > 0) {
            const double pl = prob0[l], pr = prob0[i];
            if (!doubleCompareDev(pl, pr)) return pl < pr;
        }
        return less(id[l], id[i]);
    }

    __device__ void exchange(const RowLess & less, const uint32_t i, const uint32_t l) {
        const uint32_t a = id[i], b = id[l];
        if (b != kNone && (a == kNone || before(less, l, i))) {
            const double na = noise[i], pa = prob0[i];
            const uint32_t ga = ngroups[i];
            id[i] = b;
            noise[i] = noise[l];
            prob0[i] = prob0[l];
            ngroups[i] = ngroups[l];
            id[l] = a;
            noise[l] = na;
            prob0[l] = pa;
            ngroups[l] = ga;
        }
    }
};

#define RPVG_SORT_TILE_LDS(tile)                          \
    __shared__ uint32_t tile##_id[kSmallSort];            \
    __shared__ uint32_t tile##_ngroups[kSmallSort];       \
    __shared__ double tile##_noise[kSmallSort];           \
    __shared__ double tile##_prob0[kSmallSort];           \
    SortTile tile{tile##_id, tile##_ngroups, tile##_noise, tile##_prob0}

// one workgroup per cluster with at most kSmallSort rows; positions beyond the cluster's rows behave as +infinity
__global__ __launch_bounds__(256) void sortSmallClustersKernel(const uint32_t num_listed, const uint32_t * __restrict__ listed,
                                                               const uint64_t * __restrict__ cluster_read_off, const RowLess less,
                                                               uint32_t * __restrict__ sorted) {
    RPVG_SORT_TILE_LDS(tile);
    if (blockIdx.x >= num_listed) return;
    const uint32_t c = listed[blockIdx.x];
    const uint64_t r0 = cluster_read_off[c];
    const uint32_t n = static_cast<uint32_t>(cluster_read_off[c + 1] - r0);
    uint32_t L = 2;
    while (L < n) L <<= 1;
    for (uint32_t i = threadIdx.x; i < L; i += blockDim.x) tile.load(less, i, (i < n) ? static_cast<uint32_t>(r0 + i) : kNone);
    __syncthreads();
    for (uint32_t k = 2; k <= L; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = threadIdx.x; t < (L >> 1); t += blockDim.x) {
                uint32_t i, l;
                bitonicPair(t, k, j, &i, &l);
                tile.exchange(less, i, l);
            }
            __syncthreads();
        }
    }
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) sorted[r0 + i] = tile.id[i];
}

// Clusters with more than kSmallSort rows: tiles of kSmallSort positions are sorted (first = true: all steps up to
// k = kSmallSort) or finished (first = false: the steps j = kSmallSort / 2 .. 1 of the merge of width k, which stay
// inside a tile) in LDS; only the steps that span tiles run through global memory (bitonicStepBigKernel).
__global__ __launch_bounds__(256) void sortTilesBigKernel(const uint32_t num_tiles, const uint32_t * __restrict__ tile_cluster,
                                                          const uint32_t * __restrict__ tile_index,
                                                          const uint64_t * __restrict__ cluster_read_off, const RowLess less,
                                                          const bool first, const uint32_t k_merge, uint32_t * __restrict__ sorted) {
    RPVG_SORT_TILE_LDS(tile);
    if (blockIdx.x >= num_tiles) return;
    const uint32_t c = tile_cluster[blockIdx.x];
    const uint64_t r0 = cluster_read_off[c];
    const uint32_t n = static_cast<uint32_t>(cluster_read_off[c + 1] - r0);
    const uint32_t base = tile_index[blockIdx.x] * kSmallSort;
    if (base >= n) return;  // a tile of padding only
    if (!first) {
        uint32_t padded = kSmallSort;
        while (padded < n) padded <<= 1;
        if (k_merge > padded) return;  // this cluster's network is complete
    }
    for (uint32_t i = threadIdx.x; i < kSmallSort; i += blockDim.x) tile.load(less, i, (base + i < n) ? sorted[r0 + base + i] : kNone);
    __syncthreads();
    if (first) {
        for (uint32_t k = 2; k <= kSmallSort; k <<= 1) {
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                for (uint32_t t = threadIdx.x; t < (kSmallSort >> 1); t += blockDim.x) {
                    uint32_t i, l;
                    bitonicPair(t, k, j, &i, &l);
                    tile.exchange(less, i, l);
                }
                __syncthreads();
            }
        }
    } else {
        for (uint32_t j = kSmallSort >> 1; j > 0; j >>= 1) {
            for (uint32_t t = threadIdx.x; t < (kSmallSort >> 1); t += blockDim.x) {
                uint32_t i, l;
                bitonicPair(t, k_merge, j, &i, &l);  // j < k_merge / 2: the plain (i, i + j) pairing
                tile.exchange(less, i, l);
            }
            __syncthreads();
        }
    }
    for (uint32_t i = threadIdx.x; i < kSmallSort; i += blockDim.x) {
        if (base + i < n) sorted[r0 + base + i] = tile.id[i];
    }
}

// one step (k, j) of the network for the clusters with more than kSmallSort rows: thread = (big cluster, pair)
__global__ void bitonicStepBigKernel(const uint32_t num_big, const uint32_t * __restrict__ big_cluster,
                                     const uint64_t * __restrict__ big_pair_off, const uint32_t * __restrict__ big_padded,
                                     const uint64_t * __restrict__ cluster_read_off, const RowLess less, const uint32_t k,
                                     const uint32_t j, uint32_t * __restrict__ sorted) {
    const uint64_t g = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (g >= big_pair_off[num_big]) return;
    uint32_t lo = 0, hi = num_big - 1;
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (big_pair_off[mid] <= g) lo = mid; else hi = mid - 1;
    }
    if (k > big_padded[lo]) return;
    const uint32_t c = big_cluster[lo];
    const uint64_t r0 = cluster_read_off[c];
    const uint32_t n = static_cast<uint32_t>(cluster_read_off[c + 1] - r0);
    uint32_t i, l;
    bitonicPair(static_cast<uint32_t>(g - big_pair_off[lo]), k, j, &i, &l);
    if (l >= n) return;  // the partner is +infinity: already in order
    const uint32_t a = sorted[r0 + i], b = sorted[r0 + l];
    if (less(b, a)) {
        sorted[r0 + i] = b;
        sorted[r0 + l] = a;
    }
}

// head[i] = 1 when sorted row i cannot be merged into its predecessor (first guess at the runs); head_pos[i] = i then.
__global__ void adjacentHeadsKernel(const uint64_t n, const RowView v, const uint32_t * __restrict__ sorted,
                                    uint32_t * __restrict__ head, uint32_t * __restrict__ head_pos) {
    const uint64_t i = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (i >= n) return;
    bool h = true;
    if (i > 0) {
        const uint32_t a = sorted[i - 1], b = sorted[i];
        h = v.read_cluster[a] != v.read_cluster[b] || !mergeable(v, a, b);
    }
    head[i] = h;
    head_pos[i] = h ? static_cast<uint32_t>(i) : 0;
}

// The reference merges into the HEAD of the run (src/main.cpp:958-968), not into the predecessor: a row that
// matched its predecessor but not the head starts a new run.  Flags the clusters where that happens.
__global__ void checkRunHeadsKernel(const uint64_t n, const RowView v, const uint32_t * __restrict__ sorted,
                                    const uint32_t * __restrict__ head, const uint32_t * __restrict__ run_head_pos,
                                    uint32_t * __restrict__ cluster_needs_walk, uint32_t * __restrict__ any) {
    const uint64_t i = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (i >= n || head[i]) return;
    if (!mergeable(v, sorted[run_head_pos[i]], sorted[i])) {
        cluster_needs_walk[v.read_cluster[sorted[i]]] = 1;
        *any = 1;
    }
}

// Exact sequential replay of src/main.cpp:953-973 for the flagged clusters (one thread each; rare).
__global__ void walkClustersKernel(const uint32_t num_clusters, const uint64_t * __restrict__ cluster_read_off, const RowView v,
                                   const uint32_t * __restrict__ sorted, const uint32_t * __restrict__ cluster_needs_walk,
                                   uint32_t * __restrict__ head) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= num_clusters || !cluster_needs_walk[k]) return;
    const uint64_t r0 = cluster_read_off[k], r1 = cluster_read_off[k + 1];
    uint64_t cur = r0;
    for (uint64_t i = r0; i < r1; ++i) {
        if (i == r0) {
            head[i] = 1;
        } else if (mergeable(v, sorted[cur], sorted[i])) {
            head[i] = 0;
        } else {
            head[i] = 1;
            cur = i;
        }
    }
}

// run_incl[i] (inclusive sum of head flags) - 1 = merged row of sorted row i
__global__ void mergeRunsKernel(const uint64_t n, const RowView v, const uint32_t * __restrict__ sorted,
                                const uint32_t * __restrict__ head, const uint32_t * __restrict__ run_incl,
                                uint32_t * __restrict__ source, uint32_t * __restrict__ merged_count) {
    const uint64_t i = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (i >= n) return;
    const uint32_t run = run_incl[i] - 1;
    const uint32_t r = sorted[i];
    atomicAdd(merged_count + run, v.read_count[r]);  // integer: order-free
    if (head[i]) source[run] = r;
}

__global__ void clusterRowOffKernel(const uint32_t num_clusters, const uint64_t num_reads, const uint64_t num_merged,
                                    const uint64_t * __restrict__ cluster_read_off, const uint32_t * __restrict__ run_incl,
                                    uint64_t * __restrict__ out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > num_clusters) return;
    const uint64_t r = cluster_read_off[k];
    out[k] = (r < num_reads) ? run_incl[r] - 1 : num_merged;
}

__global__ void gatherSizesKernel(const uint64_t n, const uint32_t * __restrict__ source, const uint32_t * __restrict__ ngroups,
                                  const uint32_t * __restrict__ nmembers, uint64_t * __restrict__ out_ngroups,
                                  uint64_t * __restrict__ out_nmembers) {
    const uint64_t i = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (i >= n) return;
    const uint32_t r = source ? source[i] : static_cast<uint32_t>(i);
    out_ngroups[i] = ngroups[r];
    out_nmembers[i] = nmembers[r];
}

// rows -> the expanded entries of rpvg_hip_batch (as rpvg_hip_batch_upload does for host rows)
__global__ void rowsExpandGroupsKernel(const uint64_t num_groups, const uint64_t * __restrict__ grp_idx_off,
                                       const double * __restrict__ grp_prob, double * __restrict__ ent_prob) {
    const uint64_t g = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (g >= num_groups) return;
    const double p = grp_prob[g];
    for (uint64_t e = grp_idx_off[g]; e < grp_idx_off[g + 1]; ++e) ent_prob[e] = p;
}

__global__ void rowsMetaKernel(const uint64_t num_rows, const uint64_t * __restrict__ row_grp_off,
                               const uint64_t * __restrict__ grp_idx_off, const uint32_t * __restrict__ row_count_u32,
                               uint64_t * __restrict__ row_ent_off, double * __restrict__ row_count) {
    const uint64_t r = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (r > num_rows) return;
    row_ent_off[r] = grp_idx_off[row_grp_off[r]];
    if (r < num_rows) row_count[r] = static_cast<double>(row_count_u32[r]);
}

__global__ void clusterEntryOffKernel(const uint32_t num_clusters, const uint64_t * __restrict__ cluster_row_off,
                                      const uint64_t * __restrict__ row_ent_off, uint64_t * __restrict__ out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k <= num_clusters) out[k] = row_ent_off[cluster_row_off[k]];
}

struct MaxOp {
    __device__ uint32_t operator()(const uint32_t a, const uint32_t b) const { return a > b ? a : b; }
};

template <typename T>
hipError_t exclusiveSum(const T * in, T * out, const size_t n, hipStream_t st) {
    size_t bytes = 0;
    hipError_t err = hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, in, out, n, st);
    if (err != hipSuccess) return err;
    DeviceBuffer<uint8_t> tmp;
    err = tmp.alloc(bytes ? bytes : 1);
    if (err != hipSuccess) return err;
    err = hipcub::DeviceScan::ExclusiveSum(tmp.ptr, bytes, in, out, n, st);
    if (err != hipSuccess) return err;
    return hipStreamSynchronize(st);  // tmp is released at scope end
}

inline dim3 gridFor(const uint64_t n, const uint32_t block) { return dim3(static_cast<uint32_t>((n + block - 1) / block)); }

// The per-read slices of one build (RowsScratch points into them).
struct RowsScratchBuffers {
    DeviceBuffer<double> alp, unit_val, tmp_val, bucket_mean, row_noise, grp_prob;
    DeviceBuffer<uint32_t> prefix, unit_idx, tmp_idx, bucket_of, bucket_count, bucket_first, bucket_rank, bucket_cursor, ngroups,
        nmembers, grp_size, grp_moff, member;

    int alloc(const uint64_t N, const uint64_t A, const uint64_t E, const bool collapse) {
        RPVG_HIP_CHECK(alp.alloc(A));
        RPVG_HIP_CHECK(prefix.alloc(E + N + 1));
        RPVG_HIP_CHECK(unit_idx.alloc(E));
        RPVG_HIP_CHECK(unit_val.alloc(E));
        if (collapse) {
            RPVG_HIP_CHECK(tmp_idx.alloc(E));
            RPVG_HIP_CHECK(tmp_val.alloc(E));
        }
        RPVG_HIP_CHECK(bucket_of.alloc(E));
        RPVG_HIP_CHECK(bucket_mean.alloc(E));
        RPVG_HIP_CHECK(bucket_count.alloc(E));
        RPVG_HIP_CHECK(bucket_first.alloc(E));
        RPVG_HIP_CHECK(bucket_rank.alloc(E));
        RPVG_HIP_CHECK(bucket_cursor.alloc(E));
        RPVG_HIP_CHECK(row_noise.alloc(N));
        RPVG_HIP_CHECK(ngroups.alloc(N + 1));
        RPVG_HIP_CHECK(nmembers.alloc(N + 1));
        RPVG_HIP_CHECK(grp_prob.alloc(E));
        RPVG_HIP_CHECK(grp_size.alloc(E));
        RPVG_HIP_CHECK(grp_moff.alloc(E));
        RPVG_HIP_CHECK(member.alloc(E));
        return RPVG_HIP_OK;
    }

    RowsScratch view() const {
        RowsScratch sc;
        sc.align_log_prob = alp.ptr;
        sc.surv_prefix = prefix.ptr;
        sc.unit_idx = unit_idx.ptr;
        sc.unit_val = unit_val.ptr;
        sc.tmp_idx = tmp_idx.ptr;
        sc.tmp_val = tmp_val.ptr;
        sc.bucket_of = bucket_of.ptr;
        sc.bucket_mean = bucket_mean.ptr;
        sc.bucket_count = bucket_count.ptr;
        sc.bucket_first = bucket_first.ptr;
        sc.bucket_rank = bucket_rank.ptr;
        sc.bucket_cursor = bucket_cursor.ptr;
        sc.row_noise = row_noise.ptr;
        sc.row_ngroups = ngroups.ptr;
        sc.row_nmembers = nmembers.ptr;
        sc.grp_prob = grp_prob.ptr;
        sc.grp_size = grp_size.ptr;
        sc.grp_moff = grp_moff.ptr;
        sc.member = member.ptr;
        return sc;
    }
};

// The caller's sort + quickMergeIdentical (src/main.cpp:953-973) over the padded row slices: sorts the rows of every
// cluster, finds the runs and leaves out->row_count / cluster_row_off of the merged rows; source[j] = the read whose
// slice holds merged row j.
int mergeRows(rpvg_hip_ctx * ctx, const rpvg_hip_alignments * al, const RowsScratch & sc, const double prob_precision,
              rpvg_hip_read_rows * out, DeviceBuffer<uint32_t> * d_source, uint64_t * num_out) {
    hipStream_t st = ctx->stream;
    const uint32_t K = al->num_clusters;
    const uint64_t N = al->num_reads;
    RowView view;
    view.read_cluster = al->read_cluster.ptr;
    view.read_count = al->read_count.ptr;
    view.read_align_off = al->read_align_off.ptr;
    view.align_path_off = al->align_path_off.ptr;
    view.sc = sc;
    view.prob_precision = prob_precision;
    const RowLess less{view};

    DeviceBuffer<uint32_t> d_sorted, d_head, d_head_pos, d_run_head, d_walk, d_any, d_run_incl;
    RPVG_HIP_CHECK(d_sorted.alloc(N));
    RPVG_HIP_CHECK(d_head.alloc(N));
    RPVG_HIP_CHECK(d_head_pos.alloc(N));
    RPVG_HIP_CHECK(d_run_head.alloc(N));
    RPVG_HIP_CHECK(d_run_incl.alloc(N));
    RPVG_HIP_CHECK(d_walk.alloc(K));
    RPVG_HIP_CHECK(d_any.alloc(1));

    // ---- sort the rows of every cluster ------------------------------------------------------------------
    std::vector<uint32_t> small_clusters, big_clusters, big_padded;
    std::vector<uint64_t> big_pair_off(1, 0);
    uint32_t max_padded = 0;
    for (uint32_t k = 0; k < K; ++k) {
        const uint64_t n = al->h_cluster_read_off[k + 1] - al->h_cluster_read_off[k];
        if (n < 2) continue;
        if (n <= kSmallSort) {
            small_clusters.push_back(k);
        } else {
            uint32_t L = kSmallSort;
            while (L < n) L <<= 1;
            big_clusters.push_back(k);
            big_padded.push_back(L);
            big_pair_off.push_back(big_pair_off.back() + L / 2);
            max_padded = std::max(max_padded, L);
        }
    }
    iotaKernel<<<gridFor(N, 256), dim3(256), 0, st>>>(N, d_sorted.ptr);
    DeviceBuffer<uint32_t> d_small, d_big, d_big_padded;
    DeviceBuffer<uint64_t> d_big_pair_off;
    if (!small_clusters.empty()) {
        RPVG_HIP_CHECK(d_small.upload(small_clusters.data(), small_clusters.size(), st));
        sortSmallClustersKernel<<<dim3(static_cast<uint32_t>(small_clusters.size())), dim3(256), 0, st>>>(
            static_cast<uint32_t>(small_clusters.size()), d_small.ptr, al->cluster_read_off.ptr, less, d_sorted.ptr);
    }
    DeviceBuffer<uint32_t> d_tile_cluster, d_tile_index;
    if (!big_clusters.empty()) {
        RPVG_HIP_CHECK(d_big.upload(big_clusters.data(), big_clusters.size(), st));
        RPVG_HIP_CHECK(d_big_padded.upload(big_padded.data(), big_padded.size(), st));
        RPVG_HIP_CHECK(d_big_pair_off.upload(big_pair_off.data(), big_pair_off.size(), st));
        const uint32_t num_big = static_cast<uint32_t>(big_clusters.size());
        std::vector<uint32_t> tile_cluster, tile_index;
        for (uint32_t b = 0; b < num_big; ++b) {
            for (uint32_t t = 0; t < big_padded[b] / kSmallSort; ++t) {
                tile_cluster.push_back(big_clusters[b]);
                tile_index.push_back(t);
            }
        }
        const uint32_t num_tiles = static_cast<uint32_t>(tile_cluster.size());
        RPVG_HIP_CHECK(d_tile_cluster.upload(tile_cluster.data(), num_tiles, st));
        RPVG_HIP_CHECK(d_tile_index.upload(tile_index.data(), num_tiles, st));
        sortTilesBigKernel<<<dim3(num_tiles), dim3(256), 0, st>>>(num_tiles, d_tile_cluster.ptr, d_tile_index.ptr,
                                                                 al->cluster_read_off.ptr, less, true, 0, d_sorted.ptr);
        for (uint32_t k = 2 * kSmallSort; k <= max_padded; k <<= 1) {
            for (uint32_t j = k >> 1; j >= kSmallSort; j >>= 1) {
                bitonicStepBigKernel<<<gridFor(big_pair_off.back(), 256), dim3(256), 0, st>>>(
                    num_big, d_big.ptr, d_big_pair_off.ptr, d_big_padded.ptr, al->cluster_read_off.ptr, less, k, j, d_sorted.ptr);
            }
            sortTilesBigKernel<<<dim3(num_tiles), dim3(256), 0, st>>>(num_tiles, d_tile_cluster.ptr, d_tile_index.ptr,
                                                                     al->cluster_read_off.ptr, less, false, k, d_sorted.ptr);
        }
    }
    RPVG_HIP_CHECK(hipGetLastError());

    // ---- runs ---------------------------------------------------------------------------------------------
    adjacentHeadsKernel<<<gridFor(N, 256), dim3(256), 0, st>>>(N, view, d_sorted.ptr, d_head.ptr, d_head_pos.ptr);
    {
        size_t bytes = 0;
        RPVG_HIP_CHECK(hipcub::DeviceScan::InclusiveScan(nullptr, bytes, d_head_pos.ptr, d_run_head.ptr, MaxOp(), static_cast<int>(N), st));
        DeviceBuffer<uint8_t> tmp;
        RPVG_HIP_CHECK(tmp.alloc(bytes ? bytes : 1));
        RPVG_HIP_CHECK(hipcub::DeviceScan::InclusiveScan(tmp.ptr, bytes, d_head_pos.ptr, d_run_head.ptr, MaxOp(), static_cast<int>(N), st));
        RPVG_HIP_CHECK(hipStreamSynchronize(st));
    }
    RPVG_HIP_CHECK(hipMemsetAsync(d_walk.ptr, 0, sizeof(uint32_t) * K, st));
    RPVG_HIP_CHECK(hipMemsetAsync(d_any.ptr, 0, sizeof(uint32_t), st));
    checkRunHeadsKernel<<<gridFor(N, 256), dim3(256), 0, st>>>(N, view, d_sorted.ptr, d_head.ptr, d_run_head.ptr, d_walk.ptr, d_any.ptr);
    uint32_t any = 0;
    RPVG_HIP_CHECK(hipMemcpyAsync(&any, d_any.ptr, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    RPVG_HIP_CHECK(hipStreamSynchronize(st));
    if (any) {
        walkClustersKernel<<<gridFor(K, 64), dim3(64), 0, st>>>(K, al->cluster_read_off.ptr, view, d_sorted.ptr, d_walk.ptr, d_head.ptr);
    }
    {
        size_t bytes = 0;
        RPVG_HIP_CHECK(hipcub::DeviceScan::InclusiveSum(nullptr, bytes, d_head.ptr, d_run_incl.ptr, static_cast<int>(N), st));
        DeviceBuffer<uint8_t> tmp;
        RPVG_HIP_CHECK(tmp.alloc(bytes ? bytes : 1));
        RPVG_HIP_CHECK(hipcub::DeviceScan::InclusiveSum(tmp.ptr, bytes, d_head.ptr, d_run_incl.ptr, static_cast<int>(N), st));
        RPVG_HIP_CHECK(hipStreamSynchronize(st));
    }
    uint32_t num_runs = 0;
    RPVG_HIP_CHECK(hipMemcpyAsync(&num_runs, d_run_incl.ptr + (N - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    RPVG_HIP_CHECK(hipStreamSynchronize(st));
    *num_out = num_runs;
    RPVG_HIP_CHECK(d_source->alloc(*num_out));
    RPVG_HIP_CHECK(out->row_count.alloc(*num_out));
    RPVG_HIP_CHECK(out->cluster_row_off.alloc(K + 1));
    RPVG_HIP_CHECK(hipMemsetAsync(out->row_count.ptr, 0, sizeof(uint32_t) * *num_out, st));
    mergeRunsKernel<<<gridFor(N, 256), dim3(256), 0, st>>>(N, view, d_sorted.ptr, d_head.ptr, d_run_incl.ptr, d_source->ptr,
                                                         out->row_count.ptr);
    clusterRowOffKernel<<<gridFor(K + 1, 256), dim3(256), 0, st>>>(K, N, *num_out, al->cluster_read_off.ptr, d_run_incl.ptr,
                                                                 out->cluster_row_off.ptr);
    RPVG_HIP_CHECK(hipGetLastError());
    out->h_cluster_row_off.resize(K + 1);
    RPVG_HIP_CHECK(hipMemcpyAsync(out->h_cluster_row_off.data(), out->cluster_row_off.ptr, sizeof(uint64_t) * (K + 1), hipMemcpyDeviceToHost, st));
    RPVG_HIP_CHECK(hipStreamSynchronize(st));  // the sort / scan buffers above go out of scope here
    return RPVG_HIP_OK;
}

// Packs the padded slices of the listed rows (source == nullptr: all reads in order) into the grouped CSR of `out`.
int packRows(rpvg_hip_ctx * ctx, const rpvg_hip_alignments * al, const RowsScratch & sc, const uint32_t * pack_source,
             const uint64_t num_rows, rpvg_hip_read_rows * out) {
    hipStream_t st = ctx->stream;
    DeviceBuffer<uint64_t> d_ng64, d_nm64, d_row_member_off;
    RPVG_HIP_CHECK(d_ng64.alloc(num_rows + 1));
    RPVG_HIP_CHECK(d_nm64.alloc(num_rows + 1));
    RPVG_HIP_CHECK(out->row_grp_off.alloc(num_rows + 1));
    RPVG_HIP_CHECK(d_row_member_off.alloc(num_rows + 1));
    RPVG_HIP_CHECK(hipMemsetAsync(d_ng64.ptr, 0, sizeof(uint64_t) * (num_rows + 1), st));
    RPVG_HIP_CHECK(hipMemsetAsync(d_nm64.ptr, 0, sizeof(uint64_t) * (num_rows + 1), st));
    gatherSizesKernel<<<gridFor(num_rows, 256), dim3(256), 0, st>>>(num_rows, pack_source, sc.row_ngroups, sc.row_nmembers, d_ng64.ptr,
                                                                 d_nm64.ptr);
    RPVG_HIP_CHECK(exclusiveSum(d_ng64.ptr, out->row_grp_off.ptr, num_rows + 1, st));
    RPVG_HIP_CHECK(exclusiveSum(d_nm64.ptr, d_row_member_off.ptr, num_rows + 1, st));
    uint64_t totals[2] = {0, 0};
    RPVG_HIP_CHECK(hipMemcpyAsync(&totals[0], out->row_grp_off.ptr + num_rows, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    RPVG_HIP_CHECK(hipMemcpyAsync(&totals[1], d_row_member_off.ptr + num_rows, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    RPVG_HIP_CHECK(hipStreamSynchronize(st));
    const uint64_t G = totals[0], M = totals[1];

    RPVG_HIP_CHECK(out->row_noise.alloc(num_rows));
    RPVG_HIP_CHECK(out->grp_prob.alloc(G + 1));
    RPVG_HIP_CHECK(out->grp_idx_off.alloc(G + 1));
    RPVG_HIP_CHECK(out->path_idx.alloc(M + 1));
    RPVG_HIP_CHECK(hipMemsetAsync(out->grp_idx_off.ptr + G, 0, sizeof(uint64_t), st));  // G == 0: the terminator is written here only
    packRowsKernel<<<gridFor(num_rows * kGroupLanes, 256), dim3(256), 0, st>>>(num_rows, pack_source, al->read_align_off.ptr, al->align_path_off.ptr,
                                                                    sc, out->row_grp_off.ptr, d_row_member_off.ptr, out->row_noise.ptr,
                                                                    out->grp_prob.ptr, out->grp_idx_off.ptr, out->path_idx.ptr);
    RPVG_HIP_CHECK(hipGetLastError());
    RPVG_HIP_CHECK(hipStreamSynchronize(st));  // the offset buffers above go out of scope
    out->num_rows = num_rows;
    out->num_groups = G;
    out->num_members = M;
    return RPVG_HIP_OK;
}

}  // namespace

extern "C" int rpvg_hip_alignments_upload(rpvg_hip_ctx * ctx, const rpvg_alignment_batch * in, rpvg_hip_alignments ** out_handle) {
    RPVG_REQUIRE(ctx && in && out_handle, "rpvg_hip_alignments_upload: NULL argument");
    *out_handle = nullptr;
    const uint32_t K = in->num_clusters;
    RPVG_REQUIRE(in->cluster_read_off && in->cluster_path_off, "rpvg_hip_alignments_upload: NULL cluster offsets");
    const uint64_t N = in->cluster_read_off[K], P = in->cluster_path_off[K];
    RPVG_REQUIRE(N < 0xffffffffull, "rpvg_hip_alignments_upload: %llu reads exceed one call", static_cast<unsigned long long>(N));
    RPVG_REQUIRE(N == 0 || (in->read_count && in->read_min_mapq && in->read_noise_score && in->read_align_off),
                 "rpvg_hip_alignments_upload: NULL read arrays");
    RPVG_REQUIRE(P == 0 || in->path_effective_length, "rpvg_hip_alignments_upload: NULL path_effective_length");
    const bool collapse = in->path_group != nullptr;
    RPVG_REQUIRE(!collapse || (in->cluster_group_off && in->path_source_count),
                 "rpvg_hip_alignments_upload: collapsing needs cluster_group_off and path_source_count");
    const uint64_t A = N ? in->read_align_off[N] : 0;
    RPVG_REQUIRE(A == 0 || (in->align_score_sum && in->align_length && in->align_frag_length && in->align_path_off),
                 "rpvg_hip_alignments_upload: NULL alignment arrays");
    const uint64_t E = A ? in->align_path_off[A] : 0;
    RPVG_REQUIRE(E == 0 || in->align_path_idx, "rpvg_hip_alignments_upload: NULL align_path_idx");
    RPVG_REQUIRE(E < 0xffffffffull, "rpvg_hip_alignments_upload: %llu path entries exceed one call", static_cast<unsigned long long>(E));

    // validation of the invariants the kernels rely on (O(input), as rpvg_hip_batch_upload does)
    std::vector<uint32_t> read_cluster(N);
    int bad = 0;
    for (uint32_t k = 0; k < K && !bad; ++k) {
        int why = 0;
        if (in->cluster_read_off[k] > in->cluster_read_off[k + 1] || in->cluster_path_off[k] > in->cluster_path_off[k + 1]) why = 1;
        const uint64_t np = in->cluster_path_off[k + 1] - in->cluster_path_off[k];
        const uint64_t ng = collapse ? in->cluster_group_off[k + 1] - in->cluster_group_off[k] : 0;
        for (uint64_t p = in->cluster_path_off[k]; !why && collapse && p < in->cluster_path_off[k + 1]; ++p) {
            if (in->path_group[p] >= ng || in->path_source_count[p] == 0) why = 2;
        }
        for (uint64_t r = in->cluster_read_off[k]; !why && r < in->cluster_read_off[k + 1]; ++r) {
            read_cluster[r] = k;
            if (in->read_align_off[r] >= in->read_align_off[r + 1]) why = 3;
            if (in->read_noise_score[r] > 0) why = 4;
            for (uint64_t a = in->read_align_off[r]; !why && a < in->read_align_off[r + 1]; ++a) {
                if (in->align_path_off[a] >= in->align_path_off[a + 1] || in->align_length[a] == 0) why = 5;
                for (uint64_t e = in->align_path_off[a]; !why && e < in->align_path_off[a + 1]; ++e) {
                    if (in->align_path_idx[e] >= np || (e > in->align_path_off[a] && in->align_path_idx[e - 1] >= in->align_path_idx[e])) why = 6;
                }
            }
        }
        if (why) bad = why;
    }
    static const char * const reasons[] = {"", "descending cluster offsets", "a path group outside its cluster or a zero source count",
                                           "a read without alignments", "a read with a positive noise score",
                                           "an alignment without paths or with zero length",
                                           "path indices of an alignment must be ascending and inside the cluster"};
    RPVG_REQUIRE(bad == 0, "rpvg_hip_alignments_upload: %s", reasons[bad]);

    std::unique_ptr<rpvg_hip_alignments> al(new (std::nothrow) rpvg_hip_alignments());
    if (!al) {
        setError("rpvg_hip_alignments_upload: out of host memory");
        return RPVG_HIP_ERR_ALLOC;
    }
    al->num_clusters = K;
    al->num_reads = N;
    al->num_aligns = A;
    al->num_entries = E;
    al->num_paths = P;
    al->collapse = collapse;
    al->h_cluster_read_off.assign(in->cluster_read_off, in->cluster_read_off + K + 1);
    al->h_out_path_off.assign(K + 1, 0);
    for (uint32_t k = 0; k < K; ++k) {
        al->h_out_path_off[k + 1] = al->h_out_path_off[k] + (collapse ? in->cluster_group_off[k + 1] - in->cluster_group_off[k]
                                                                      : in->cluster_path_off[k + 1] - in->cluster_path_off[k]);
    }

    std::lock_guard<std::mutex> lock(ctx->mutex);
    RPVG_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int span = ctx->spanBegin(FAM_H2D);
    RPVG_HIP_CHECK(al->cluster_path_off.upload(in->cluster_path_off, K + 1, st));
    RPVG_HIP_CHECK(al->cluster_read_off.upload(in->cluster_read_off, K + 1, st));
    RPVG_HIP_CHECK(al->eff_len.upload(in->path_effective_length, P, st));
    if (collapse) {
        RPVG_HIP_CHECK(al->source_count.upload(in->path_source_count, P, st));
        RPVG_HIP_CHECK(al->path_group.upload(in->path_group, P, st));
    }
    if (N) {
        RPVG_HIP_CHECK(al->read_cluster.upload(read_cluster.data(), N, st));
        RPVG_HIP_CHECK(al->read_count.upload(in->read_count, N, st));
        RPVG_HIP_CHECK(al->mapq.upload(in->read_min_mapq, N, st));
        RPVG_HIP_CHECK(al->noise_score.upload(in->read_noise_score, N, st));
        RPVG_HIP_CHECK(al->read_align_off.upload(in->read_align_off, N + 1, st));
        RPVG_HIP_CHECK(al->score.upload(in->align_score_sum, A, st));
        RPVG_HIP_CHECK(al->align_length.upload(in->align_length, A, st));
        RPVG_HIP_CHECK(al->frag_length.upload(in->align_frag_length, A, st));
        RPVG_HIP_CHECK(al->align_path_off.upload(in->align_path_off, A + 1, st));
        RPVG_HIP_CHECK(al->path_idx.upload(in->align_path_idx, E, st));
    }
    {
        std::vector<uint32_t> small_reads, large_reads;
        for (uint64_t r = 0; r < N; ++r) {
            const uint64_t entries = in->align_path_off[in->read_align_off[r + 1]] - in->align_path_off[in->read_align_off[r]];
            (entries <= static_cast<uint64_t>(kGroupLanes) && !collapse ? small_reads : large_reads).push_back(static_cast<uint32_t>(r));
        }
        al->num_small = small_reads.size();
        al->num_large = large_reads.size();
        if (!small_reads.empty()) RPVG_HIP_CHECK(al->small_reads.upload(small_reads.data(), small_reads.size(), st));
        if (!large_reads.empty()) RPVG_HIP_CHECK(al->large_reads.upload(large_reads.data(), large_reads.size(), st));
        RPVG_HIP_CHECK(hipStreamSynchronize(st));
    }
    ctx->spanEnd(span);
    ctx->stats.h2d_bytes += static_cast<double>(N) * 25 + static_cast<double>(A) * 16 + static_cast<double>(E) * 4 + static_cast<double>(P) * 8;
    RPVG_HIP_CHECK(hipStreamSynchronize(st));  // read_cluster leaves scope
    *out_handle = al.release();
    return RPVG_HIP_OK;
}

extern "C" void rpvg_hip_alignments_free(rpvg_hip_ctx * ctx, rpvg_hip_alignments * alignments) {
    if (!alignments) return;
    if (ctx) {
        std::lock_guard<std::mutex> lock(ctx->mutex);
        (void) hipSetDevice(ctx->device);
        (void) hipStreamSynchronize(ctx->stream);
    }
    delete alignments;
}

extern "C" int rpvg_hip_read_rows_build(rpvg_hip_ctx * ctx, const rpvg_hip_alignments * al, const rpvg_row_params * prm,
                                        int32_t merge, rpvg_hip_read_rows ** rows_out) {
    RPVG_REQUIRE(ctx && al && prm && rows_out, "rpvg_hip_read_rows_build: NULL argument");
    *rows_out = nullptr;
    RPVG_REQUIRE(prm->is_single_end || prm->frag_length_log_prob, "rpvg_hip_read_rows_build: paired-end rows need frag_length_log_prob");
    RPVG_REQUIRE(prm->prob_precision > 0 && prm->min_noise_prob >= 0 && prm->min_noise_prob <= 1,
                 "rpvg_hip_read_rows_build: prob_precision / min_noise_prob out of range");
    const uint32_t K = al->num_clusters;
    const uint64_t N = al->num_reads, A = al->num_aligns, E = al->num_entries;
    const bool collapse = al->collapse;

    std::unique_ptr<rpvg_hip_read_rows> out(new (std::nothrow) rpvg_hip_read_rows());
    if (!out) {
        setError("rpvg_hip_read_rows_build: out of host memory");
        return RPVG_HIP_ERR_ALLOC;
    }
    out->num_clusters = K;
    out->h_cluster_path_off = al->h_out_path_off;

    std::lock_guard<std::mutex> lock(ctx->mutex);
    RPVG_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;

    if (N == 0) {
        out->h_cluster_row_off.assign(K + 1, 0);
        RPVG_HIP_CHECK(out->cluster_row_off.upload(out->h_cluster_row_off.data(), K + 1, st));
        const uint64_t zero = 0;
        RPVG_HIP_CHECK(out->row_grp_off.upload(&zero, 1, st));
        RPVG_HIP_CHECK(out->grp_idx_off.upload(&zero, 1, st));
        RPVG_HIP_CHECK(hipStreamSynchronize(st));
        *rows_out = out.release();
        return RPVG_HIP_OK;
    }

    std::vector<double> phred(256);
    for (int q = 0; q < 256; ++q) phred[q] = std::pow(10, -static_cast<double>(q) / 10);  // Utils::phred_to_prob, src/utils.hpp:131-133
    DeviceBuffer<double> d_frag, d_phred;
    if (!prm->is_single_end) RPVG_HIP_CHECK(d_frag.upload(prm->frag_length_log_prob, RPVG_FRAG_LENGTH_TABLE_SIZE, st));
    RPVG_HIP_CHECK(d_phred.upload(phred.data(), 256, st));

    RowsScratchBuffers buffers;
    if (const int rc = buffers.alloc(N, A, E, collapse)) return rc;

    RowsIn rin;
    rin.num_reads = N;
    rin.read_cluster = al->read_cluster.ptr;
    rin.cluster_path_off = al->cluster_path_off.ptr;
    rin.path_eff_len = al->eff_len.ptr;
    rin.path_source_count = al->source_count.ptr;
    rin.path_group = collapse ? al->path_group.ptr : nullptr;
    rin.read_min_mapq = al->mapq.ptr;
    rin.read_noise_score = al->noise_score.ptr;
    rin.read_align_off = al->read_align_off.ptr;
    rin.align_score_sum = al->score.ptr;
    rin.align_length = al->align_length.ptr;
    rin.align_frag_length = al->frag_length.ptr;
    rin.align_path_off = al->align_path_off.ptr;
    rin.align_path_idx = al->path_idx.ptr;
    rin.frag_table = prm->is_single_end ? nullptr : d_frag.ptr;
    rin.phred_prob = d_phred.ptr;
    rin.prob_precision = prm->prob_precision;
    rin.min_noise_prob = prm->min_noise_prob;

    const RowsScratch sc = buffers.view();

    // ---- rows ----------------------------------------------------------------------------------------------
    hipEvent_t ev0, ev1, ev2;
    RPVG_HIP_CHECK(hipEventCreate(&ev0));
    RPVG_HIP_CHECK(hipEventCreate(&ev1));
    RPVG_HIP_CHECK(hipEventCreate(&ev2));
    RPVG_HIP_CHECK(hipEventRecord(ev0, st));
    int span = ctx->spanBegin(FAM_BUILD);
    alignLogProbKernel<<<gridFor(A, 256), dim3(256), 0, st>>>(A, al->score.ptr, al->frag_length.ptr, rin.frag_table, sc.align_log_prob);
    static const bool no_small_kernel = RPVG_EXPERIMENT_ENV("RPVG_HIP_NO_SMALL_READ_KERNEL") != nullptr;
    if (no_small_kernel) {
        readRowKernel<<<gridFor(N, kWavesPerBlock), dim3(64 * kWavesPerBlock), 0, st>>>(rin, sc, N, nullptr);
    } else {
        if (al->num_small) {
            readRowSmallKernel<<<gridFor(al->num_small * kGroupLanes, 256), dim3(256), 0, st>>>(rin, sc, al->num_small, al->small_reads.ptr);
        }
        if (al->num_large) {
            readRowKernel<<<gridFor(al->num_large, kWavesPerBlock), dim3(64 * kWavesPerBlock), 0, st>>>(rin, sc, al->num_large,
                                                                                                  al->large_reads.ptr);
        }
    }
    ctx->spanEnd(span);
    ctx->stats.build_launches += 3;
    RPVG_HIP_CHECK(hipGetLastError());
    RPVG_HIP_CHECK(hipEventRecord(ev1, st));

    // rows to pack: all reads, or the heads of the merged runs
    uint64_t num_out = N;
    DeviceBuffer<uint32_t> d_source;
    const uint32_t * pack_source = nullptr;

    if (merge) {
        if (const int rc = mergeRows(ctx, al, sc, prm->prob_precision, out.get(), &d_source, &num_out)) return rc;
        pack_source = d_source.ptr;
    } else {
        RPVG_HIP_CHECK(out->row_count.alloc(N));
        RPVG_HIP_CHECK(hipMemcpyAsync(out->row_count.ptr, al->read_count.ptr, sizeof(uint32_t) * N, hipMemcpyDeviceToDevice, st));
        out->h_cluster_row_off = al->h_cluster_read_off;
        RPVG_HIP_CHECK(out->cluster_row_off.upload(out->h_cluster_row_off.data(), K + 1, st));
    }

    if (const int rc = packRows(ctx, al, sc, pack_source, num_out, out.get())) return rc;
    RPVG_HIP_CHECK(hipEventRecord(ev2, st));
    RPVG_HIP_CHECK(hipStreamSynchronize(st));  // the scratch slices go out of scope
    float ms01 = 0, ms12 = 0;
    (void) hipEventElapsedTime(&ms01, ev0, ev1);
    (void) hipEventElapsedTime(&ms12, ev1, ev2);
    out->build_ms = ms01;
    out->merge_ms = ms12;
    (void) hipEventDestroy(ev0);
    (void) hipEventDestroy(ev1);
    (void) hipEventDestroy(ev2);
    *rows_out = out.release();
    return RPVG_HIP_OK;
}

extern "C" int rpvg_hip_read_rows_view(rpvg_hip_ctx * ctx, rpvg_hip_read_rows * rows, rpvg_cluster_batch * view, double * build_ms,
                                       double * merge_ms) {
    RPVG_REQUIRE(ctx && rows && view, "rpvg_hip_read_rows_view: NULL argument");
    if (!rows->downloaded) {
        std::lock_guard<std::mutex> lock(ctx->mutex);
        RPVG_HIP_CHECK(hipSetDevice(ctx->device));
        hipStream_t st = ctx->stream;
        const uint64_t R = rows->num_rows, G = rows->num_groups, M = rows->num_members;
        rows->h_row_count.resize(R);
        rows->h_row_noise.resize(R);
        rows->h_row_grp_off.assign(R + 1, 0);
        rows->h_grp_prob.resize(G);
        rows->h_grp_idx_off.assign(G + 1, 0);
        rows->h_path_idx.resize(M);
        if (R) {
            RPVG_HIP_CHECK(hipMemcpyAsync(rows->h_row_count.data(), rows->row_count.ptr, sizeof(uint32_t) * R, hipMemcpyDeviceToHost, st));
            RPVG_HIP_CHECK(hipMemcpyAsync(rows->h_row_noise.data(), rows->row_noise.ptr, sizeof(double) * R, hipMemcpyDeviceToHost, st));
            RPVG_HIP_CHECK(hipMemcpyAsync(rows->h_row_grp_off.data(), rows->row_grp_off.ptr, sizeof(uint64_t) * (R + 1), hipMemcpyDeviceToHost, st));
            RPVG_HIP_CHECK(hipMemcpyAsync(rows->h_grp_idx_off.data(), rows->grp_idx_off.ptr, sizeof(uint64_t) * (G + 1), hipMemcpyDeviceToHost, st));
        }
        if (G) RPVG_HIP_CHECK(hipMemcpyAsync(rows->h_grp_prob.data(), rows->grp_prob.ptr, sizeof(double) * G, hipMemcpyDeviceToHost, st));
        if (M) RPVG_HIP_CHECK(hipMemcpyAsync(rows->h_path_idx.data(), rows->path_idx.ptr, sizeof(uint32_t) * M, hipMemcpyDeviceToHost, st));
        RPVG_HIP_CHECK(hipStreamSynchronize(st));
        rows->downloaded = true;
    }
    std::memset(view, 0, sizeof(*view));
    view->num_clusters = rows->num_clusters;
    view->cluster_row_off = rows->h_cluster_row_off.data();
    view->cluster_path_off = rows->h_cluster_path_off.data();
    view->row_count = rows->h_row_count.data();
    view->row_noise = rows->h_row_noise.data();
    view->row_grp_off = rows->h_row_grp_off.data();
    view->grp_prob = rows->h_grp_prob.data();
    view->grp_idx_off = rows->h_grp_idx_off.data();
    view->path_idx = rows->h_path_idx.data();
    if (build_ms) *build_ms = rows->build_ms;
    if (merge_ms) *merge_ms = rows->merge_ms;
    return RPVG_HIP_OK;
}

extern "C" int rpvg_hip_read_rows_sizes(rpvg_hip_ctx * ctx, const rpvg_hip_read_rows * rows, rpvg_cluster_batch * view) {
    RPVG_REQUIRE(ctx && rows && view, "rpvg_hip_read_rows_sizes: NULL argument");
    std::memset(view, 0, sizeof(*view));
    view->num_clusters = rows->num_clusters;
    view->cluster_row_off = rows->h_cluster_row_off.data();
    view->cluster_path_off = rows->h_cluster_path_off.data();
    return RPVG_HIP_OK;
}

// rows -> the device-resident batch the estimators take, without leaving the GPU
extern "C" int rpvg_hip_read_rows_to_batch(rpvg_hip_ctx * ctx, const rpvg_hip_read_rows * rows, rpvg_hip_batch ** batch_out) {
    RPVG_REQUIRE(ctx && rows && batch_out, "rpvg_hip_read_rows_to_batch: NULL argument");
    *batch_out = nullptr;
    const uint32_t K = rows->num_clusters;
    const uint64_t R = rows->num_rows, G = rows->num_groups, M = rows->num_members;
    std::lock_guard<std::mutex> lock(ctx->mutex);
    RPVG_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    std::unique_ptr<rpvg_hip_batch> b(new (std::nothrow) rpvg_hip_batch());
    if (!b) {
        setError("rpvg_hip_read_rows_to_batch: out of host memory");
        return RPVG_HIP_ERR_ALLOC;
    }
    b->num_clusters = K;
    b->num_rows = R;
    b->num_entries = M;
    b->num_paths = rows->h_cluster_path_off[K];
    b->h_cluster_row_off = rows->h_cluster_row_off;
    b->h_cluster_path_off = rows->h_cluster_path_off;
    b->h_cluster_ent_off.assign(K + 1, 0);
    RPVG_HIP_CHECK(b->cluster_row_off.upload(rows->h_cluster_row_off.data(), K + 1, st));
    RPVG_HIP_CHECK(b->cluster_path_off.upload(rows->h_cluster_path_off.data(), K + 1, st));
    RPVG_HIP_CHECK(b->row_noise.alloc(R));
    RPVG_HIP_CHECK(b->row_count.alloc(R));
    RPVG_HIP_CHECK(b->row_ent_off.alloc(R + 1));
    RPVG_HIP_CHECK(b->ent_path.alloc(M));
    RPVG_HIP_CHECK(b->ent_prob.alloc(M));
    DeviceBuffer<uint64_t> d_cluster_ent_off;
    RPVG_HIP_CHECK(d_cluster_ent_off.alloc(K + 1));
    const int span = ctx->spanBegin(FAM_BUILD);
    if (R) RPVG_HIP_CHECK(hipMemcpyAsync(b->row_noise.ptr, rows->row_noise.ptr, sizeof(double) * R, hipMemcpyDeviceToDevice, st));
    if (M) RPVG_HIP_CHECK(hipMemcpyAsync(b->ent_path.ptr, rows->path_idx.ptr, sizeof(uint32_t) * M, hipMemcpyDeviceToDevice, st));
    if (G) rowsExpandGroupsKernel<<<gridFor(G, 256), dim3(256), 0, st>>>(G, rows->grp_idx_off.ptr, rows->grp_prob.ptr, b->ent_prob.ptr);
    rowsMetaKernel<<<gridFor(R + 1, 256), dim3(256), 0, st>>>(R, rows->row_grp_off.ptr, rows->grp_idx_off.ptr, rows->row_count.ptr,
                                                            b->row_ent_off.ptr, b->row_count.ptr);
    clusterEntryOffKernel<<<gridFor(K + 1, 256), dim3(256), 0, st>>>(K, b->cluster_row_off.ptr, b->row_ent_off.ptr, d_cluster_ent_off.ptr);
    ctx->spanEnd(span);
    ctx->stats.build_launches += 3;
    RPVG_HIP_CHECK(hipGetLastError());
    RPVG_HIP_CHECK(hipMemcpyAsync(b->h_cluster_ent_off.data(), d_cluster_ent_off.ptr, sizeof(uint64_t) * (K + 1), hipMemcpyDeviceToHost, st));
    RPVG_HIP_CHECK(hipStreamSynchronize(st));
    *batch_out = b.release();
    return RPVG_HIP_OK;
}

extern "C" void rpvg_hip_read_rows_free(rpvg_hip_ctx * ctx, rpvg_hip_read_rows * rows) {
    if (!rows) return;
    if (ctx) {
        std::lock_guard<std::mutex> lock(ctx->mutex);
        (void) hipSetDevice(ctx->device);
        (void) hipStreamSynchronize(ctx->stream);
    }
    delete rows;
}
