// rpvg_oracle_capi.cpp — C entry points of the CPU ORACLE (TEST
// INFRASTRUCTURE, NOT PRODUCT CODE; see rpvg_oracle.hpp).  Loaded with ctypes
// by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.
//
// rpvg_oracle_run() reproduces the caller loop of the reference,
// src/main.cpp:829-998: `#pragma omp parallel for schedule(dynamic, 1)` over
// clusters, one shared set of estimator parameters, mt19937(rng_seed + i) per
// cluster, serial inside a cluster.

#include <chrono>
#include <cstring>
#include <string>
#include <vector>

#include <omp.h>

#include "../include/rpvg_batch.h"
#include "../include/rpvg_rows.h"
#include "rpvg_oracle.hpp"

using namespace rpvg_oracle;

namespace {

struct FlatResult {
    std::vector<uint64_t> set_off, member_off, abund_off, em_off, em_col_off;
    std::vector<uint32_t> members, em_iters, em_cols;
    std::vector<double> posteriors, abundances, noise_count, total_count;
    std::vector<uint64_t> gibbs_off, gibbs_path_off, gibbs_noise_off, gibbs_abund_off;
    std::vector<uint32_t> gibbs_path;
    std::vector<double> gibbs_noise, gibbs_abund;
};

std::vector<ReadRow> unpackRows(const rpvg_cluster_batch * b, uint32_t k) {
    std::vector<ReadRow> rows;
    rows.reserve(b->cluster_row_off[k + 1] - b->cluster_row_off[k]);
    for (uint64_t r = b->cluster_row_off[k]; r < b->cluster_row_off[k + 1]; ++r) {
        ReadRow row;
        row.read_count = b->row_count[r];
        row.noise_prob = b->row_noise[r];
        for (uint64_t g = b->row_grp_off[r]; g < b->row_grp_off[r + 1]; ++g) {
            row.path_probs.emplace_back(b->grp_prob[g],
                                        std::vector<uint32_t>(b->path_idx + b->grp_idx_off[g], b->path_idx + b->grp_idx_off[g + 1]));
        }
        rows.emplace_back(std::move(row));
    }
    return rows;
}

std::vector<PathInfo> unpackPaths(const rpvg_cluster_batch * b, uint32_t k) {
    std::vector<PathInfo> paths;
    for (uint64_t p = b->cluster_path_off[k]; p < b->cluster_path_off[k + 1]; ++p) {
        PathInfo info;
        info.group_id = b->path_group_id ? b->path_group_id[p] : 0;
        info.source_count = b->path_source_count ? b->path_source_count[p] : 1;
        if (b->path_source_off) {
            info.source_ids.assign(b->source_id + b->path_source_off[p], b->source_id + b->path_source_off[p + 1]);
        }
        info.effective_length = b->path_effective_length ? b->path_effective_length[p] : 0;
        paths.emplace_back(std::move(info));
    }
    return paths;
}

Params toParams(const rpvg_params * p) {
    Params q;
    q.max_em_its = p->max_em_its;
    q.max_rel_em_conv = p->max_rel_em_conv;
    q.num_gibbs_samples = p->num_gibbs_samples;
    q.gibbs_thin_its = p->gibbs_thin_its;
    q.prob_precision = p->prob_precision;
    q.ploidy = p->ploidy;
    q.min_hap_prob = p->min_hap_prob;
    q.ind_hap_inference = p->ind_hap_inference != 0;
    q.use_hap_gibbs = p->use_hap_gibbs != 0;
    return q;
}

}  // namespace

extern "C" {

// Run `model` on every cluster of the batch with num_threads OpenMP threads.
// seconds_out (optional) receives the wall time of the estimate() loop only.
void * rpvg_oracle_run(const char * model, const rpvg_params * params, const rpvg_cluster_batch * batch,
                       int num_threads, double * seconds_out) {
    const uint32_t K = batch->num_clusters;
    const Params prm = toParams(params);
    const std::string model_s(model);

    std::vector<Estimates> ests(K);
    std::vector<std::vector<ReadRow>> rows(K);
    for (uint32_t k = 0; k < K; ++k) {
        ests[k].paths = unpackPaths(batch, k);
        rows[k] = unpackRows(batch, k);
    }
    if (num_threads < 1) num_threads = 1;

    auto t0 = std::chrono::steady_clock::now();
#pragma omp parallel for schedule(dynamic, 1) num_threads(num_threads)
    for (uint32_t k = 0; k < K; ++k) {
        std::mt19937 rng(params->rng_seed + k);
        estimate(model_s, prm, &ests[k], rows[k], &rng);
    }
    auto t1 = std::chrono::steady_clock::now();
    if (seconds_out) *seconds_out = std::chrono::duration<double>(t1 - t0).count();

    FlatResult * res = new FlatResult();
    res->set_off.push_back(0);
    res->member_off.push_back(0);
    res->abund_off.push_back(0);
    res->em_off.push_back(0);
    res->em_col_off.push_back(0);
    res->gibbs_off.push_back(0);
    res->gibbs_path_off.push_back(0);
    res->gibbs_noise_off.push_back(0);
    res->gibbs_abund_off.push_back(0);
    for (uint32_t k = 0; k < K; ++k) {
        const Estimates & e = ests[k];
        for (auto & cs : e.gibbs_read_count_samples) {
            res->gibbs_path.insert(res->gibbs_path.end(), cs.path_ids.begin(), cs.path_ids.end());
            res->gibbs_path_off.push_back(res->gibbs_path.size());
            res->gibbs_noise.insert(res->gibbs_noise.end(), cs.noise_samples.begin(), cs.noise_samples.end());
            res->gibbs_noise_off.push_back(res->gibbs_noise.size());
            res->gibbs_abund.insert(res->gibbs_abund.end(), cs.abundance_samples.begin(), cs.abundance_samples.end());
            res->gibbs_abund_off.push_back(res->gibbs_abund.size());
        }
        res->gibbs_off.push_back(res->gibbs_path_off.size() - 1);
        for (size_t s = 0; s < e.path_group_sets.size(); ++s) {
            res->members.insert(res->members.end(), e.path_group_sets[s].begin(), e.path_group_sets[s].end());
            res->member_off.push_back(res->members.size());
            res->posteriors.push_back(e.posteriors[s]);
        }
        res->set_off.push_back(res->posteriors.size());
        res->abundances.insert(res->abundances.end(), e.abundances.begin(), e.abundances.end());
        res->abund_off.push_back(res->abundances.size());
        res->noise_count.push_back(e.noise_count);
        res->total_count.push_back(e.total_count);
        for (size_t i = 0; i < e.em_iters.size(); ++i) {
            res->em_iters.push_back(e.em_iters[i]);
            res->em_cols.insert(res->em_cols.end(), e.em_problem_paths[i].begin(), e.em_problem_paths[i].end());
            res->em_col_off.push_back(res->em_cols.size());
        }
        res->em_off.push_back(res->em_iters.size());
    }
    return res;
}

void rpvg_oracle_view(void * handle, rpvg_estimates_view * out) {
    FlatResult * r = static_cast<FlatResult *>(handle);
    out->num_clusters = r->noise_count.size();
    out->set_off = r->set_off.data();
    out->member_off = r->member_off.data();
    out->members = r->members.data();
    out->posteriors = r->posteriors.data();
    out->abund_off = r->abund_off.data();
    out->abundances = r->abundances.data();
    out->noise_count = r->noise_count.data();
    out->total_count = r->total_count.data();
    out->em_off = r->em_off.data();
    out->em_iters = r->em_iters.data();
    out->em_col_off = r->em_col_off.data();
    out->em_cols = r->em_cols.data();
    out->gibbs_off = r->gibbs_off.data();
    out->gibbs_path_off = r->gibbs_path_off.data();
    out->gibbs_path = r->gibbs_path.data();
    out->gibbs_noise_off = r->gibbs_noise_off.data();
    out->gibbs_noise = r->gibbs_noise.data();
    out->gibbs_abund_off = r->gibbs_abund_off.data();
    out->gibbs_abund = r->gibbs_abund.data();
}

void rpvg_oracle_free(void * handle) { delete static_cast<FlatResult *>(handle); }

// EMAbundanceEstimator on an already-normalised dense column-major R x C
// matrix (last column = noise).  abundances_out has C-1 entries.  Returns the
// iteration count.  seconds_out (optional) = wall time of the EM loop.
uint32_t rpvg_oracle_em_dense(const double * P_colmajor, uint64_t R, uint64_t C, const double * counts,
                              uint32_t max_em_its, double max_rel_em_conv, double * abundances_out,
                              double * noise_count_out, double * total_count_out, double * seconds_out) {
    ColMatrix P(R, C);
    std::memcpy(P.v.data(), P_colmajor, sizeof(double) * R * C);
    std::vector<double> c(counts, counts + R);
    Estimates est;
    est.resetEstimates(C - 1, 1);
    double total = 0;
    for (auto x : c) total += x;
    est.total_count = total;
    auto t0 = std::chrono::steady_clock::now();
    uint32_t its = EMAbundanceEstimator(&est, P, c, max_em_its, max_rel_em_conv);
    auto t1 = std::chrono::steady_clock::now();
    if (seconds_out) *seconds_out = std::chrono::duration<double>(t1 - t0).count();
    for (uint64_t j = 0; j + 1 < C; ++j) abundances_out[j] = est.abundances[j];
    *noise_count_out = est.noise_count;
    *total_count_out = est.total_count;
    return its;
}

// weightedMinimumPathCover; cover is row-major R x N bytes.  Returns the size
// of the cover written to out (capacity N).
uint32_t rpvg_oracle_min_path_cover(const uint8_t * cover, uint64_t R, uint64_t N, const double * read_counts,
                                    const double * path_weights, uint32_t * out) {
    std::vector<uint8_t> cv(cover, cover + R * N);
    std::vector<double> rc(read_counts, read_counts + R), pw(path_weights, path_weights + N);
    auto res = weightedMinimumPathCover(cv, R, N, rc, pw);
    for (size_t i = 0; i < res.size(); ++i) out[i] = res[i];
    return res.size();
}

// Group posteriors on a dense column-major R x N matrix with separate noise
// vector.  mode 0 = Full, 1 = Bounded(min_rel_likelihood).  Returns the number
// of sets; members_out (capacity max_sets*group_size), posteriors_out
// (capacity max_sets).  Returns UINT32_MAX if capacity is too small.
uint32_t rpvg_oracle_group_posteriors(const double * P_colmajor, uint64_t R, uint64_t N, const double * noise,
                                      const double * counts, const uint32_t * path_counts, uint32_t group_size,
                                      int mode, double min_rel_lik, uint32_t max_sets, uint32_t * members_out,
                                      double * posteriors_out) {
    ColMatrix P(R, N);
    std::memcpy(P.v.data(), P_colmajor, sizeof(double) * R * N);
    std::vector<double> nz(noise, noise + R), c(counts, counts + R);
    std::vector<uint32_t> pc(path_counts, path_counts + N);
    Estimates est;
    if (mode == 0) {
        calculatePathGroupPosteriorsFull(&est, P, nz, c, pc, group_size);
    } else {
        calculatePathGroupPosteriorsBounded(&est, P, nz, c, pc, group_size, min_rel_lik);
    }
    if (est.path_group_sets.size() > max_sets) return UINT32_MAX;
    for (size_t s = 0; s < est.path_group_sets.size(); ++s) {
        for (uint32_t m = 0; m < group_size; ++m) members_out[s * group_size + m] = est.path_group_sets[s][m];
        posteriors_out[s] = est.posteriors[s];
    }
    return est.path_group_sets.size();
}

uint32_t rpvg_oracle_num_permutations(const uint32_t * values, uint32_t n) {
    return numPermutations(std::vector<uint32_t>(values, values + n));
}

double rpvg_oracle_add_log(double x, double y) { return add_log(x, y); }

int rpvg_oracle_double_compare(double a, double b) { return doubleCompare(a, b) ? 1 : 0; }

int rpvg_oracle_max_threads(void) { return omp_get_max_threads(); }

// ---- row construction ---------------------------------------------------------------

namespace {

struct FlatRows {
    std::vector<uint64_t> cluster_row_off, cluster_path_off, row_grp_off, grp_idx_off;
    std::vector<uint32_t> row_count, path_idx;
    std::vector<double> row_noise, grp_prob;
};

}  // namespace

// FragmentLengthDist::logProb(v) for v = 0 .. RPVG_FRAG_LENGTH_TABLE_SIZE-1 (src/fragment_length_dist.cpp:385-394).
void rpvg_oracle_frag_length_table(double loc, double scale, double shape, uint32_t sd_max_multi, double * out) {
    const FragmentLengthDist fld(loc, scale, shape, sd_max_multi);
    for (uint32_t v = 0; v < RPVG_FRAG_LENGTH_TABLE_SIZE; ++v) out[v] = fld.logProb(v);
}

// The caller's loop of src/main.cpp:889-973 over every cluster of the batch: one addPathProbs per read, then (merge != 0)
// sort + quickMergeIdentical.  OpenMP over clusters as the reference (src/main.cpp:829).
void * rpvg_oracle_build_rows(const rpvg_alignment_batch * b, const rpvg_row_params * prm, int merge, int num_threads,
                              double * seconds_out) {
    const uint32_t K = b->num_clusters;
    FragmentLengthDist fld;
    if (!prm->is_single_end) {
        fld.log_prob_buffer.assign(prm->frag_length_log_prob, prm->frag_length_log_prob + RPVG_FRAG_LENGTH_TABLE_SIZE);
    }
    const bool collapse = b->path_group != nullptr;
    std::vector<std::vector<ReadRow>> rows(K);
    if (num_threads < 1) num_threads = 1;

    auto t0 = std::chrono::steady_clock::now();
#pragma omp parallel for schedule(dynamic, 1) num_threads(num_threads)
    for (uint32_t k = 0; k < K; ++k) {
        const uint64_t p0 = b->cluster_path_off[k], p1 = b->cluster_path_off[k + 1];
        std::vector<PathInfo> paths(p1 - p0);
        std::vector<uint32_t> path_group;
        for (uint64_t p = p0; p < p1; ++p) {
            paths[p - p0].effective_length = b->path_effective_length[p];
            paths[p - p0].source_count = b->path_source_count ? b->path_source_count[p] : 1;
            if (collapse) path_group.emplace_back(b->path_group[p]);
        }
        const uint32_t num_groups = collapse ? b->cluster_group_off[k + 1] - b->cluster_group_off[k] : 0;
        rows[k].reserve(b->cluster_read_off[k + 1] - b->cluster_read_off[k]);
        for (uint64_t r = b->cluster_read_off[k]; r < b->cluster_read_off[k + 1]; ++r) {
            std::vector<AlignPath> align_paths;
            for (uint64_t a = b->read_align_off[r]; a < b->read_align_off[r + 1]; ++a) {
                AlignPath ap;
                ap.min_mapq = b->read_min_mapq[r];
                ap.score_sum = b->align_score_sum[a];
                ap.align_length = b->align_length[a];
                ap.frag_length = b->align_frag_length[a];
                ap.path_idx.assign(b->align_path_idx + b->align_path_off[a], b->align_path_idx + b->align_path_off[a + 1]);
                align_paths.emplace_back(std::move(ap));
            }
            AlignPath noise;
            noise.min_mapq = b->read_min_mapq[r];
            noise.score_sum = b->read_noise_score[r];
            align_paths.emplace_back(noise);
            rows[k].emplace_back(addPathProbs(b->read_count[r], prm->prob_precision, align_paths, paths, fld,
                                              prm->is_single_end != 0, prm->min_noise_prob, collapse, path_group, num_groups));
        }
        if (merge) sortAndMergeRows(&rows[k], prm->prob_precision);
    }
    auto t1 = std::chrono::steady_clock::now();
    if (seconds_out) *seconds_out = std::chrono::duration<double>(t1 - t0).count();

    FlatRows * out = new FlatRows();
    out->cluster_row_off.push_back(0);
    out->cluster_path_off.push_back(0);
    out->row_grp_off.push_back(0);
    out->grp_idx_off.push_back(0);
    for (uint32_t k = 0; k < K; ++k) {
        for (auto & row : rows[k]) {
            out->row_count.push_back(row.read_count);
            out->row_noise.push_back(row.noise_prob);
            for (auto & g : row.path_probs) {
                out->grp_prob.push_back(g.first);
                out->path_idx.insert(out->path_idx.end(), g.second.begin(), g.second.end());
                out->grp_idx_off.push_back(out->path_idx.size());
            }
            out->row_grp_off.push_back(out->grp_prob.size());
        }
        out->cluster_row_off.push_back(out->row_count.size());
        const uint64_t cols = collapse ? b->cluster_group_off[k + 1] - b->cluster_group_off[k]
                                       : b->cluster_path_off[k + 1] - b->cluster_path_off[k];
        out->cluster_path_off.push_back(out->cluster_path_off.back() + cols);
    }
    return out;
}

void rpvg_oracle_rows_view(void * handle, rpvg_cluster_batch * out) {
    FlatRows * r = static_cast<FlatRows *>(handle);
    std::memset(out, 0, sizeof(*out));
    out->num_clusters = r->cluster_row_off.size() - 1;
    out->cluster_row_off = r->cluster_row_off.data();
    out->cluster_path_off = r->cluster_path_off.data();
    out->row_count = r->row_count.data();
    out->row_noise = r->row_noise.data();
    out->row_grp_off = r->row_grp_off.data();
    out->grp_prob = r->grp_prob.data();
    out->grp_idx_off = r->grp_idx_off.data();
    out->path_idx = r->path_idx.data();
}

void rpvg_oracle_rows_free(void * handle) { delete static_cast<FlatRows *>(handle); }

// createPathClusters over id sets (CSR).  cluster_off needs num_paths + 1 slots, cluster_paths num_paths.
uint32_t rpvg_oracle_path_clusters(uint32_t num_paths, uint64_t num_sets, const uint64_t * set_off, const uint32_t * set_path,
                                   uint32_t * path_to_cluster, uint64_t * cluster_off, uint32_t * cluster_paths, double * seconds_out) {
    std::vector<std::vector<uint32_t>> sets(num_sets);
    for (uint64_t s = 0; s < num_sets; ++s) sets[s].assign(set_path + set_off[s], set_path + set_off[s + 1]);
    std::vector<uint32_t> p2c;
    auto t0 = std::chrono::steady_clock::now();
    auto clusters = createPathClusters(num_paths, sets, &p2c);
    auto t1 = std::chrono::steady_clock::now();
    if (seconds_out) *seconds_out = std::chrono::duration<double>(t1 - t0).count();
    cluster_off[0] = 0;
    uint64_t n = 0;
    for (size_t c = 0; c < clusters.size(); ++c) {
        for (auto p : clusters[c]) cluster_paths[n++] = p;
        cluster_off[c + 1] = n;
    }
    for (uint32_t i = 0; i < num_paths; ++i) path_to_cluster[i] = p2c[i];
    return clusters.size();
}

}  // extern "C"
