// rpvg_oracle.cpp — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
// See rpvg_oracle.hpp for the parity status ("parity unpinned" except the
// minimum-path-cover KAT) and the rules on who may load this.
//
// The reference evaluates everything with Eigen dense expressions; here each
// expression is spelled out as explicit loops over a column-major matrix in
// the same pass structure (one loop nest per Eigen statement), so that the
// CPU baseline timed from this file does the same number of sweeps over the
// matrix as the reference does.

#include "rpvg_oracle.hpp"

#include <algorithm>
#include <cassert>
#include <cmath>
#include <limits>
#include <numeric>
#include <queue>
#include <set>
#include <unordered_map>

namespace rpvg_oracle {

using std::pair;
using std::vector;

// src/path_abundance_estimator.cpp:10-14
static const uint32_t min_em_conv_its = 10;
static const double min_em_abundance = 1e-8;
static const double abundance_gibbs_gamma = 1;
static const double min_gibbs_abundance = 1e-8;

// src/path_estimator.cpp:3-11
static const uint32_t min_gibbs_chains = 10;
static const double gibbs_chain_scaling = 0.01;
static const uint32_t min_burn_it = 50;
static const double burn_it_scaling = 0.025;
static const uint32_t min_gibbs_it = 100;
static const double gibbs_it_scaling = 0.05;

// src/path_posterior_estimator.cpp:5
static const double min_rel_likelihood = 1e-8;

static const double double_precision = std::numeric_limits<double>::epsilon() * 100;  // utils.hpp:81
static const double kLowest = std::numeric_limits<double>::lowest();

// ---------------------------------------------------------------------------
// scalar helpers
// ---------------------------------------------------------------------------

// src/utils.hpp:87-93
bool doubleCompare(const double a, const double b) {
    assert(std::isfinite(a));
    assert(std::isfinite(b));
    return ((a == b) || (std::abs(a - b) < std::abs(std::min(a, b)) * double_precision));
}

// src/utils.hpp:95-117 — n!/(n-u+1)! with u = #distinct values (kept as is,
// including being "wrong" from ploidy 4: SURVEY.md appendix A.8).
uint32_t numPermutations(vector<uint32_t> values) {
    assert(!values.empty());
    if (values.size() == 1) {
        return 1;
    }
    std::sort(values.begin(), values.end());
    uint32_t num_unique_values = 1;
    for (size_t i = 1; i < values.size(); ++i) {
        if (values[i - 1] != values[i]) {
            num_unique_values++;
        }
    }
    return (std::tgamma(values.size() + 1) / std::tgamma(values.size() - num_unique_values + 2));
}

// src/utils.hpp:300-302
double add_log(double log_x, double log_y) {
    return log_x > log_y ? log_x + std::log1p(std::exp(log_y - log_x)) : log_y + std::log1p(std::exp(log_x - log_y));
}

// src/path_cluster_estimates.hpp:65-110 — multisets with repetition of size
// group_size over [0, num_components) in lexicographic order.
static void generateGroupsRecursive(vector<vector<uint32_t>> * out, uint32_t num_components, uint32_t group_size,
                                    vector<uint32_t> cur_group) {
    if (cur_group.size() < group_size) {
        uint32_t start_idx = cur_group.empty() ? 0 : cur_group.back();
        for (uint32_t i = start_idx; i < num_components; ++i) {
            vector<uint32_t> new_group = cur_group;
            new_group.push_back(i);
            generateGroupsRecursive(out, num_components, group_size, new_group);
        }
    } else {
        out->emplace_back(cur_group);
    }
}

void Estimates::resetEstimates(uint32_t num_components, const uint32_t group_size) {
    path_group_sets.clear();
    posteriors.clear();
    abundances.clear();
    noise_count = 0;
    total_count = 0;
    gibbs_read_count_samples.clear();
    if (group_size > 0) {
        generateGroupsRecursive(&path_group_sets, num_components, group_size, vector<uint32_t>());
        posteriors = vector<double>(path_group_sets.size(), 0);
        abundances = vector<double>(path_group_sets.size() * group_size, 0);
    }
}

// ---------------------------------------------------------------------------
// matrix builders
// ---------------------------------------------------------------------------

// src/path_estimator.cpp:55-77
void constructProbabilityMatrix(ColMatrix * P, vector<double> * noise, vector<double> * counts,
                                const vector<ReadRow> & rows, const uint32_t num_paths) {
    assert(!rows.empty());
    *P = ColMatrix(rows.size(), num_paths);
    noise->assign(rows.size(), 0);
    counts->assign(rows.size(), 0);
    for (size_t i = 0; i < rows.size(); ++i) {
        for (auto & pp : rows[i].path_probs) {
            for (auto & path : pp.second) {
                assert(path < num_paths);
                P->at(i, path) = pp.first;
            }
        }
        (*noise)[i] = rows[i].noise_prob;
        (*counts)[i] = rows[i].read_count;
    }
}

// src/path_estimator.cpp:79-113
void constructPartialProbabilityMatrix(ColMatrix * P, vector<double> * noise, vector<double> * counts,
                                       const vector<ReadRow> & rows, const vector<uint32_t> & path_ids,
                                       const uint32_t num_paths) {
    assert(!rows.empty());
    assert(!path_ids.empty());
    vector<int32_t> path_id_idx(num_paths, -1);
    for (size_t i = 0; i < path_ids.size(); ++i) {
        path_id_idx.at(path_ids[i]) = i;
    }
    *P = ColMatrix(rows.size(), path_ids.size());
    noise->assign(rows.size(), 0);
    counts->assign(rows.size(), 0);
    for (size_t i = 0; i < rows.size(); ++i) {
        for (auto & pp : rows[i].path_probs) {
            for (auto & path : pp.second) {
                assert(path < num_paths);
                if (path_id_idx[path] >= 0) {
                    P->at(i, path_id_idx[path]) = pp.first;
                }
            }
        }
        (*noise)[i] = rows[i].noise_prob;
        (*counts)[i] = rows[i].read_count;
    }
}

// src/path_estimator.cpp:115-154
void constructGroupedProbabilityMatrix(ColMatrix * P, vector<double> * noise, vector<double> * counts,
                                       const vector<ReadRow> & rows, const vector<vector<uint32_t>> & path_groups,
                                       const uint32_t num_paths) {
    assert(!rows.empty());
    assert(!path_groups.empty());
    vector<vector<uint32_t>> path_id_group_idx(num_paths);
    for (size_t i = 0; i < path_groups.size(); ++i) {
        assert(!path_groups[i].empty());
        for (auto & path : path_groups[i]) {
            path_id_group_idx.at(path).emplace_back(i);
        }
    }
    *P = ColMatrix(rows.size(), path_groups.size());
    noise->assign(rows.size(), 0);
    counts->assign(rows.size(), 0);
    for (size_t i = 0; i < rows.size(); ++i) {
        for (auto & pp : rows[i].path_probs) {
            for (auto & path : pp.second) {
                assert(path < num_paths);
                for (auto & group_id : path_id_group_idx[path]) {
                    P->at(i, group_id) += pp.first;
                }
            }
        }
        (*noise)[i] = rows[i].noise_prob;
        (*counts)[i] = rows[i].read_count;
    }
}

// src/path_estimator.cpp:156-166 — three full-matrix statements + a resize
// that reallocates and copies (conservativeResize), kept as separate passes.
void addNoiseAndNormalizeProbabilityMatrix(ColMatrix * P, const vector<double> & noise) {
    assert(P->rows == noise.size());
    const size_t R = P->rows, N = P->cols;
    vector<double> row_sum(R, 0.0);
    for (size_t j = 0; j < N; ++j) {
        const double * c = P->col(j);
        for (size_t i = 0; i < R; ++i) row_sum[i] += c[i];
    }
    for (size_t j = 0; j < N; ++j)
        for (size_t i = 0; i < R; ++i) P->at(i, j) = P->at(i, j) / row_sum[i];
    for (size_t j = 0; j < N; ++j)
        for (size_t i = 0; i < R; ++i) P->at(i, j) = P->at(i, j) * (1 - noise[i]);
    for (size_t j = 0; j < N; ++j)
        for (size_t i = 0; i < R; ++i)
            if (std::isnan(P->at(i, j))) P->at(i, j) = 0;
    ColMatrix Q(R, N + 1);
    std::copy(P->v.begin(), P->v.end(), Q.v.begin());
    for (size_t i = 0; i < R; ++i) Q.at(i, N) = noise[i];
    *P = std::move(Q);
}

// src/path_estimator.cpp:13-31 — tolerant lexicographic row comparator.
static bool probabilityCountRowSorter(const pair<vector<double>, double> & lhs, const pair<vector<double>, double> & rhs) {
    assert(lhs.first.size() == rhs.first.size());
    for (size_t i = 0; i < lhs.first.size(); ++i) {
        if (!doubleCompare(lhs.first[i], rhs.first[i])) {
            return (lhs.first[i] < rhs.first[i]);
        }
    }
    if (!doubleCompare(lhs.second, rhs.second)) {
        return (lhs.second < rhs.second);
    }
    return false;
}

// src/path_estimator.cpp:197-259 — row sort, then merge runs of rows whose
// every entry is within prob_precision of the run's first row.
void readCollapseProbabilityMatrix(ColMatrix * P, vector<double> * counts, const double prob_precision) {
    assert(P->rows > 0);
    assert(P->rows == counts->size());
    const size_t R = P->rows, C = P->cols;

    vector<pair<vector<double>, double>> sorted_rows;
    sorted_rows.reserve(R);
    for (size_t i = 0; i < R; ++i) {
        vector<double> row(C);
        for (size_t j = 0; j < C; ++j) row[j] = P->at(i, j);
        sorted_rows.emplace_back(std::move(row), (*counts)[i]);
    }
    std::sort(sorted_rows.begin(), sorted_rows.end(), probabilityCountRowSorter);
    for (size_t i = 0; i < R; ++i) {
        for (size_t j = 0; j < C; ++j) P->at(i, j) = sorted_rows[i].first[j];
        (*counts)[i] = sorted_rows[i].second;
    }

    uint32_t prev_unique_probs_row = 0;
    for (size_t i = 1; i < R; ++i) {
        bool is_identical = true;
        for (size_t j = 0; j < C; ++j) {
            if (std::abs(P->at(prev_unique_probs_row, j) - P->at(i, j)) >= prob_precision) {
                is_identical = false;
                break;
            }
        }
        if (is_identical) {
            (*counts)[prev_unique_probs_row] += (*counts)[i];
        } else {
            if (prev_unique_probs_row + 1 < i) {
                for (size_t j = 0; j < C; ++j) P->at(prev_unique_probs_row + 1, j) = P->at(i, j);
                (*counts)[prev_unique_probs_row + 1] = (*counts)[i];
            }
            prev_unique_probs_row++;
        }
    }

    const size_t newR = prev_unique_probs_row + 1;
    ColMatrix Q(newR, C);
    for (size_t j = 0; j < C; ++j)
        for (size_t i = 0; i < newR; ++i) Q.at(i, j) = P->at(i, j);
    *P = std::move(Q);
    counts->resize(newR);
}

// ---------------------------------------------------------------------------
// EM
// ---------------------------------------------------------------------------

// src/path_abundance_estimator.cpp:47-114.  Per iteration, as the reference:
//   :61  read_posteriors = P .* abundances (row-broadcast)   -> temp R x C
//   :62  read_posteriors /= rowwise sum                       -> 2 sweeps
//   :64  abundances = read_counts * read_posteriors           -> 1 sweep
//   :65  abundances /= total_count
//   :67-95 convergence scan, :97 prev <- cur
uint32_t EMAbundanceEstimator(Estimates * est, const ColMatrix & P, const vector<double> & counts,
                              const uint32_t max_em_its, const double max_rel_em_conv) {
    assert(!est->abundances.empty());
    assert(est->noise_count == 0);
    assert(est->total_count > 0);

    const size_t R = P.rows, C = P.cols;
    assert(C == est->abundances.size() + 1);

    // :54 — note the float division.
    vector<double> abundances(C, 1 / static_cast<float>(est->abundances.size() + 1));
    vector<double> prev_abundances = abundances;

    uint32_t em_conv_its = 0;
    uint32_t its_done = 0;

    vector<double> row_sum(R);

    for (uint32_t it = 0; it < max_em_its; ++it) {
        ++its_done;
        ColMatrix post(R, C);  // the reference allocates this temporary every iteration
        for (size_t j = 0; j < C; ++j) {
            const double a = abundances[j];
            const double * pc = P.col(j);
            double * qc = post.v.data() + j * R;
            for (size_t i = 0; i < R; ++i) qc[i] = pc[i] * a;
        }
        std::fill(row_sum.begin(), row_sum.end(), 0.0);
        for (size_t j = 0; j < C; ++j) {
            const double * qc = post.col(j);
            for (size_t i = 0; i < R; ++i) row_sum[i] += qc[i];
        }
        for (size_t j = 0; j < C; ++j) {
            double * qc = post.v.data() + j * R;
            for (size_t i = 0; i < R; ++i) qc[i] = qc[i] / row_sum[i];
        }
        for (size_t j = 0; j < C; ++j) {
            const double * qc = post.col(j);
            double acc = 0;
            for (size_t i = 0; i < R; ++i) acc += counts[i] * qc[i];
            abundances[j] = acc;
        }
        for (size_t j = 0; j < C; ++j) abundances[j] /= est->total_count;

        bool has_converged = true;
        for (size_t j = 0; j < C; ++j) {
            if (abundances[j] >= min_em_abundance) {
                auto rel_abundance_diff = std::fabs(abundances[j] - prev_abundances[j]) / abundances[j];
                if (rel_abundance_diff > max_rel_em_conv) {
                    has_converged = false;
                    break;
                }
            }
        }
        if (has_converged) {
            em_conv_its++;
            if (em_conv_its == min_em_conv_its) {
                break;
            }
        } else {
            em_conv_its = 0;
        }
        prev_abundances = abundances;
    }

    for (size_t j = 0; j < C - 1; ++j) {
        if (abundances[j] < min_em_abundance) {
            est->noise_count += abundances[j] * est->total_count;
            est->abundances.at(j) = 0;
        } else {
            est->abundances.at(j) = abundances[j] * est->total_count;
        }
    }
    est->noise_count += abundances[C - 1] * est->total_count;
    return its_done;
}

// src/path_abundance_estimator.cpp:116-212 (libstdc++ <random>, same streams
// as a gcc build of the reference).
void gibbsReadCountSampler(Estimates * est, const ColMatrix & P, const vector<double> & counts, const double gamma,
                           std::mt19937 * rng, const uint32_t num_samples, const uint32_t gibbs_thin_its) {
    assert(est->total_count > 0);
    assert(!est->gibbs_read_count_samples.empty());
    CountSamples & cs = est->gibbs_read_count_samples.back();
    assert(cs.path_ids.size() == est->abundances.size());
    cs.noise_samples.reserve(num_samples);
    cs.abundance_samples.reserve(est->abundances.size() * num_samples);

    const size_t R = P.rows, C = P.cols;
    vector<double> gibbs_abundances(C);
    for (size_t j = 0; j < C - 1; ++j) gibbs_abundances[j] = est->abundances[j] / est->total_count;
    gibbs_abundances[C - 1] = est->noise_count / est->total_count;

    const uint32_t num_gibbs_its = num_samples * gibbs_thin_its;
    vector<double> row_sum(R);
    ColMatrix post(R, C);

    for (uint32_t gibbs_it = 1; gibbs_it <= num_gibbs_its; ++gibbs_it) {
        for (size_t j = 0; j < C; ++j)
            for (size_t i = 0; i < R; ++i) post.at(i, j) = P.at(i, j) * gibbs_abundances[j];
        std::fill(row_sum.begin(), row_sum.end(), 0.0);
        for (size_t j = 0; j < C; ++j)
            for (size_t i = 0; i < R; ++i) row_sum[i] += post.at(i, j);
        for (size_t j = 0; j < C; ++j)
            for (size_t i = 0; i < R; ++i) post.at(i, j) /= row_sum[i];

        vector<uint32_t> gibbs_path_read_counts(C, 0);
        for (size_t i = 0; i < R; ++i) {
            uint32_t row_reads_counts = counts[i];
            double row_sum_probs = 1;
            for (size_t j = 0; j < C; ++j) {
                auto cur_prob = post.at(i, j);
                if (cur_prob > 0) {
                    assert(row_sum_probs > 0);
                    std::binomial_distribution<uint32_t> sampler(row_reads_counts, std::min(1.0, cur_prob / row_sum_probs));
                    auto path_read_count = sampler(*rng);
                    gibbs_path_read_counts[j] += path_read_count;
                    row_reads_counts -= path_read_count;
                    if (row_reads_counts == 0) {
                        break;
                    }
                }
                row_sum_probs -= cur_prob;
            }
            assert(row_reads_counts == 0);
        }

        double gibbs_abundances_sum = 0;
        for (size_t j = 0; j < C; ++j) {
            std::gamma_distribution<double> gamma_count_dist(gibbs_path_read_counts[j] + gamma, 1);
            gibbs_abundances[j] = gamma_count_dist(*rng);
            gibbs_abundances_sum += gibbs_abundances[j];
        }
        for (size_t j = 0; j < C; ++j) gibbs_abundances[j] = gibbs_abundances[j] / gibbs_abundances_sum;

        if (gibbs_it % gibbs_thin_its == 0) {
            cs.noise_samples.emplace_back(0);
            for (size_t j = 0; j < C - 1; ++j) {
                if (gibbs_abundances[j] < min_gibbs_abundance) {
                    cs.noise_samples.back() += gibbs_abundances[j] * est->total_count;
                    cs.abundance_samples.emplace_back(0);
                } else {
                    cs.abundance_samples.emplace_back(gibbs_abundances[j] * est->total_count);
                }
            }
            cs.noise_samples.back() += gibbs_abundances[C - 1] * est->total_count;
        }
    }
}

// ---------------------------------------------------------------------------
// group posteriors
// ---------------------------------------------------------------------------

// src/path_estimator.cpp:315-330
static vector<double> calcPathLogFrequences(const vector<uint32_t> & path_counts) {
    vector<double> path_log_freqs;
    path_log_freqs.reserve(path_counts.size());
    uint32_t count_sum = std::accumulate(path_counts.begin(), path_counts.end(), 0);
    assert(count_sum > 0);
    for (auto & count : path_counts) {
        assert(count > 0);
        path_log_freqs.emplace_back(std::log(count / static_cast<double>(count_sum)));
    }
    return path_log_freqs;
}

// c . log(v)  (read_counts * v.array().log().matrix())
static double countLogDot(const vector<double> & counts, const vector<double> & v) {
    double acc = 0;
    for (size_t i = 0; i < v.size(); ++i) acc += counts[i] * std::log(v[i]);
    return acc;
}

// src/path_estimator.cpp:332-377
void calculatePathGroupPosteriorsFull(Estimates * est, const ColMatrix & P, const vector<double> & noise,
                                      const vector<double> & counts, const vector<uint32_t> & path_counts,
                                      const uint32_t group_size) {
    assert(P.rows > 0);
    assert(P.rows == noise.size());
    assert(P.rows == counts.size());
    assert(P.cols == path_counts.size());
    assert(group_size > 0);

    auto path_log_freqs = calcPathLogFrequences(path_counts);
    est->resetEstimates(P.cols, group_size);
    assert(est->posteriors.size() > 0);

    const size_t R = P.rows;
    double sum_log_posterior = kLowest;
    vector<double> group_read_probs(R);

    for (uint32_t s = 0; s < est->path_group_sets.size(); ++s) {
        group_read_probs = noise;
        for (auto & path_idx : est->path_group_sets[s]) {
            const double * c = P.col(path_idx);
            for (size_t i = 0; i < R; ++i) group_read_probs[i] += (c[i] / static_cast<double>(group_size));
        }
        est->posteriors[s] = countLogDot(counts, group_read_probs);
        for (auto & path_idx : est->path_group_sets[s]) {
            est->posteriors[s] += path_log_freqs.at(path_idx);
        }
        est->posteriors[s] += std::log(numPermutations(est->path_group_sets[s]));
        sum_log_posterior = add_log(sum_log_posterior, est->posteriors[s]);
    }
    for (size_t s = 0; s < est->posteriors.size(); ++s) {
        est->posteriors[s] = std::exp(est->posteriors[s] - sum_log_posterior);
    }
}

// src/path_estimator.cpp:379-473
void calculatePathGroupPosteriorsBounded(Estimates * est, const ColMatrix & P, const vector<double> & noise,
                                         const vector<double> & counts, const vector<uint32_t> & path_counts,
                                         const uint32_t group_size, const double min_rel_lik) {
    assert(P.rows > 0);
    assert(P.cols == path_counts.size());
    assert(group_size == 2);

    const double min_log_likelihood_diff = std::log(min_rel_lik);
    auto path_log_freqs = calcPathLogFrequences(path_counts);
    est->resetEstimates(0, 0);

    Estimates marginal;
    calculatePathGroupPosteriorsFull(&marginal, P, noise, counts, path_counts, 1);
    assert(marginal.posteriors.size() == P.cols);

    vector<pair<double, uint32_t>> marginal_posteriors;
    marginal_posteriors.reserve(marginal.posteriors.size());
    for (size_t i = 0; i < marginal.posteriors.size(); ++i) {
        marginal_posteriors.emplace_back(marginal.posteriors[i], marginal.path_group_sets[i].front());
    }
    std::sort(marginal_posteriors.rbegin(), marginal_posteriors.rend());

    const size_t R = P.rows;
    vector<double> max_read_probs(R, 0.0);  // :414 rowwise max / group_size
    for (size_t i = 0; i < R; ++i) {
        double m = P.at(i, 0);
        for (size_t j = 1; j < P.cols; ++j) m = std::max(m, P.at(i, j));
        max_read_probs[i] = m / static_cast<double>(group_size);
    }

    vector<double> log_likelihoods;
    double max_log_likelihood = kLowest;
    vector<double> base(R), tmp(R);

    for (uint32_t i = 0; i < marginal_posteriors.size(); ++i) {
        const uint32_t first_path_idx = marginal_posteriors[i].second;
        const double * c1 = P.col(first_path_idx);
        for (size_t r = 0; r < R; ++r) base[r] = noise[r] + (c1[r] / static_cast<double>(group_size));

        for (size_t r = 0; r < R; ++r) tmp[r] = base[r] + max_read_probs[r];
        double optimal_log_likelihood = countLogDot(counts, tmp);
        optimal_log_likelihood += path_log_freqs.at(first_path_idx) + std::log(2);

        if (optimal_log_likelihood - max_log_likelihood < min_log_likelihood_diff) {
            continue;
        }

        for (uint32_t j = i; j < marginal_posteriors.size(); ++j) {
            const uint32_t second_path_idx = marginal_posteriors[j].second;
            const double * c2 = P.col(second_path_idx);
            for (size_t r = 0; r < R; ++r) tmp[r] = base[r] + (c2[r] / static_cast<double>(group_size));
            log_likelihoods.emplace_back(countLogDot(counts, tmp));
            log_likelihoods.back() += path_log_freqs.at(first_path_idx) + path_log_freqs.at(second_path_idx) +
                                      std::log(numPermutations(vector<uint32_t>({first_path_idx, second_path_idx})));

            if (log_likelihoods.back() - max_log_likelihood < min_log_likelihood_diff) {
                log_likelihoods.pop_back();
                continue;
            }
            max_log_likelihood = std::max(max_log_likelihood, log_likelihoods.back());
            est->path_group_sets.emplace_back(vector<uint32_t>({first_path_idx, second_path_idx}));
        }
    }

    double sum_log_posterior = kLowest;
    for (size_t i = 0; i < log_likelihoods.size(); ++i) {
        if (log_likelihoods[i] - max_log_likelihood < min_log_likelihood_diff) {
            log_likelihoods[i] = kLowest;
        }
        sum_log_posterior = add_log(sum_log_posterior, log_likelihoods[i]);
    }
    assert(est->posteriors.empty());
    for (auto & ll : log_likelihoods) {
        est->posteriors.emplace_back(std::exp(ll - sum_log_posterior));
    }
    assert(est->posteriors.size() == est->path_group_sets.size());
}

struct VecHash {
    size_t operator()(const vector<uint32_t> & v) const {
        size_t seed = 0;
        for (auto x : v) seed ^= std::hash<uint32_t>()(x) + 0x9e3779b9 + (seed << 6) + (seed >> 2);
        return seed;
    }
};

// src/path_estimator.cpp:475-589 — hash maps are only looked up, never
// iterated, so the container choice does not affect the result.
void estimatePathGroupPosteriorsGibbs(Estimates * est, const ColMatrix & P, const vector<double> & noise,
                                      const vector<double> & counts, const vector<uint32_t> & path_counts,
                                      const uint32_t group_size, std::mt19937 * rng) {
    assert(P.rows > 0);
    assert(P.cols == path_counts.size());
    assert(group_size > 0);

    auto path_log_freqs = calcPathLogFrequences(path_counts);
    est->resetEstimates(0, 0);

    std::uniform_int_distribution<uint32_t> init_path_sampler(0, path_log_freqs.size() - 1);
    vector<uint32_t> cur;
    cur.reserve(group_size);

    std::unordered_map<vector<uint32_t>, std::discrete_distribution<uint32_t>, VecHash> sampler_cache;
    std::unordered_map<vector<uint32_t>, uint32_t, VecHash> set_indices;
    vector<uint32_t> sample_counts;

    const uint32_t num_gibbs_chains = min_gibbs_chains + std::round(gibbs_chain_scaling * group_size * path_log_freqs.size());
    const uint32_t num_burn_its = min_burn_it + std::round(burn_it_scaling * group_size * path_log_freqs.size());
    const uint32_t num_gibbs_its = min_gibbs_it + std::round(gibbs_it_scaling * group_size * path_log_freqs.size());

    const size_t R = P.rows;
    vector<double> group_read_probs(R), tmp(R);

    for (uint32_t c = 0; c < num_gibbs_chains; ++c) {
        cur.clear();
        for (uint32_t i = 0; i < group_size; ++i) cur.emplace_back(init_path_sampler(*rng));

        for (uint32_t it = 0; it < num_burn_its + num_gibbs_its; ++it) {
            for (uint32_t j = 0; j < group_size; ++j) {
                vector<uint32_t> key = cur;
                key.at(j) = P.cols;
                std::sort(key.begin(), key.end());
                auto cache_it = sampler_cache.emplace(key, std::discrete_distribution<uint32_t>());
                if (cache_it.second) {
                    group_read_probs = noise;
                    for (uint32_t k = 0; k < group_size; ++k) {
                        if (j != k) {
                            const double * col = P.col(cur[k]);
                            for (size_t r = 0; r < R; ++r) group_read_probs[r] += (col[r] / static_cast<double>(group_size));
                        }
                    }
                    vector<double> group_probs;
                    group_probs.reserve(P.cols);
                    double sum_log_group_probs = kLowest;
                    for (uint32_t k = 0; k < P.cols; ++k) {
                        const double * col = P.col(k);
                        for (size_t r = 0; r < R; ++r) tmp[r] = group_read_probs[r] + (col[r] / static_cast<double>(group_size));
                        group_probs.emplace_back(countLogDot(counts, tmp));
                        group_probs.back() += path_log_freqs.at(k);
                        sum_log_group_probs = add_log(sum_log_group_probs, group_probs.back());
                    }
                    for (auto & prob : group_probs) prob = std::exp(prob - sum_log_group_probs);
                    cache_it.first->second = std::discrete_distribution<uint32_t>(group_probs.begin(), group_probs.end());
                }
                cur.at(j) = cache_it.first->second(*rng);
            }
            if (it >= num_burn_its) {
                vector<uint32_t> sorted_cur = cur;
                std::sort(sorted_cur.begin(), sorted_cur.end());
                auto idx_it = set_indices.emplace(sorted_cur, est->path_group_sets.size());
                if (idx_it.second) {
                    est->path_group_sets.emplace_back(sorted_cur);
                    sample_counts.emplace_back(1);
                } else {
                    sample_counts.at(idx_it.first->second)++;
                }
            }
        }
    }
    assert(est->posteriors.empty());
    for (auto & sc : sample_counts) {
        est->posteriors.emplace_back(sc / static_cast<double>(num_gibbs_chains * num_gibbs_its));
    }
}

// ---------------------------------------------------------------------------
// minimum path cover  (src/path_abundance_estimator.cpp:297-340)
// ---------------------------------------------------------------------------
vector<uint32_t> weightedMinimumPathCover(const vector<uint8_t> & cover, const size_t num_rows, const size_t num_paths,
                                          const vector<double> & read_counts, const vector<double> & path_weights) {
    assert(num_rows == read_counts.size());
    assert(num_paths == path_weights.size());
    if (num_paths == 1) {
        return vector<uint32_t>({0});
    }
    auto uncovered = read_counts;
    vector<uint32_t> min_path_cover;
    min_path_cover.reserve(num_paths);

    while (*std::max_element(uncovered.begin(), uncovered.end()) > 0) {
        double best = 0;
        int32_t best_idx = -1;
        for (size_t j = 0; j < num_paths; ++j) {
            double acc = 0;
            for (size_t i = 0; i < num_rows; ++i) acc += uncovered[i] * static_cast<double>(cover[i * num_paths + j]);
            const double weighted = acc / path_weights[j];
            if (weighted > best) {
                best = weighted;
                best_idx = j;
            }
        }
        assert(best > 0);
        assert(best_idx >= 0);
        min_path_cover.emplace_back(best_idx);
        for (size_t i = 0; i < num_rows; ++i) {
            uncovered[i] = uncovered[i] * static_cast<double>(!cover[i * num_paths + best_idx]);
        }
    }
    std::sort(min_path_cover.begin(), min_path_cover.end());
    return min_path_cover;
}

// ---------------------------------------------------------------------------
// estimators
// ---------------------------------------------------------------------------

static double sumOf(const vector<double> & v) {
    double s = 0;
    for (auto x : v) s += x;
    return s;
}

static void recordEM(Estimates * est, uint32_t iters, const vector<uint32_t> & cols) {
    est->em_iters.push_back(iters);
    est->em_problem_paths.push_back(cols);
}

// src/path_abundance_estimator.cpp:18-45
static void estimateTranscripts(const Params & prm, Estimates * est, const vector<ReadRow> & rows, std::mt19937 * rng) {
    est->resetEstimates(est->paths.size(), 1);
    est->em_iters.clear();
    est->em_problem_paths.clear();
    if (!rows.empty()) {
        ColMatrix P;
        vector<double> noise, counts;
        constructProbabilityMatrix(&P, &noise, &counts, rows, est->paths.size());
        addNoiseAndNormalizeProbabilityMatrix(&P, noise);
        est->total_count = sumOf(counts);
        uint32_t its = EMAbundanceEstimator(est, P, counts, prm.max_em_its, prm.max_rel_em_conv);
        vector<uint32_t> all(est->paths.size());
        std::iota(all.begin(), all.end(), 0);
        recordEM(est, its, all);
        if (prm.num_gibbs_samples > 0) {
            est->gibbs_read_count_samples.emplace_back(CountSamples());
            est->gibbs_read_count_samples.back().path_ids = all;
            gibbsReadCountSampler(est, P, counts, abundance_gibbs_gamma, rng, prm.num_gibbs_samples, prm.gibbs_thin_its);
        }
    }
}

// src/path_abundance_estimator.cpp:217-295
static void estimateStrains(const Params & prm, Estimates * est, const vector<ReadRow> & rows, std::mt19937 * rng) {
    est->resetEstimates(est->paths.size(), 1);
    est->em_iters.clear();
    est->em_problem_paths.clear();
    if (rows.empty()) return;

    ColMatrix P;
    vector<double> noise, counts;
    constructProbabilityMatrix(&P, &noise, &counts, rows, est->paths.size());

    const size_t R = P.rows, N = P.cols;
    vector<uint8_t> cover(R * N, 0);
    vector<double> path_weights(N, 0.0);
    for (size_t i = 0; i < R; ++i) {
        if (doubleCompare(noise[i], 1)) {
            counts[i] = 0;
        }
        for (auto & pp : rows[i].path_probs) {
            for (auto & path : pp.second) {
                assert(pp.first > 0);
                cover[i * N + path] = 1;
                path_weights[path] += std::log(pp.first) * counts[i];
            }
        }
    }
    for (auto & w : path_weights) w *= -1;
    vector<uint32_t> min_path_cover = weightedMinimumPathCover(cover, R, N, counts, path_weights);

    if (!min_path_cover.empty()) {
        ColMatrix mP;
        vector<double> mnoise, mcounts;
        constructPartialProbabilityMatrix(&mP, &mnoise, &mcounts, rows, min_path_cover, est->paths.size());
        Estimates sub;
        sub.resetEstimates(mP.cols, 1);
        addNoiseAndNormalizeProbabilityMatrix(&mP, mnoise);
        readCollapseProbabilityMatrix(&mP, &mcounts, prm.prob_precision);
        sub.total_count = sumOf(mcounts);
        uint32_t its = EMAbundanceEstimator(&sub, mP, mcounts, prm.max_em_its, prm.max_rel_em_conv);
        recordEM(est, its, min_path_cover);
        if (prm.num_gibbs_samples > 0) {
            sub.gibbs_read_count_samples.emplace_back(CountSamples());
            sub.gibbs_read_count_samples.back().path_ids = min_path_cover;
            gibbsReadCountSampler(&sub, mP, mcounts, abundance_gibbs_gamma, rng, prm.num_gibbs_samples, prm.gibbs_thin_its);
            est->gibbs_read_count_samples.emplace_back(std::move(sub.gibbs_read_count_samples.front()));
        }
        for (size_t i = 0; i < min_path_cover.size(); ++i) {
            est->abundances.at(min_path_cover[i]) += sub.abundances.at(i);
        }
        est->noise_count = sub.noise_count;
        est->total_count = sub.total_count;
    }
}

// src/path_abundance_estimator.cpp:473-491 — groups in first-seen order.
static vector<vector<uint32_t>> findPathGroups(const vector<PathInfo> & paths) {
    vector<vector<uint32_t>> path_groups;
    std::map<uint32_t, uint32_t> path_group_indexes;
    for (size_t i = 0; i < paths.size(); ++i) {
        auto it = path_group_indexes.emplace(paths[i].group_id, path_group_indexes.size());
        if (it.second) path_groups.emplace_back(vector<uint32_t>());
        path_groups.at(it.first->second).emplace_back(i);
    }
    return path_groups;
}

// src/path_abundance_estimator.cpp:493-546.  The reference walks a sparsepp
// hash map (iteration order unknown — SURVEY.md H4); the oracle walks source
// ids in ascending order.  Only the ORDER of the groups depends on this.
static pair<vector<vector<uint32_t>>, vector<uint32_t>> findPathSourceGroups(const vector<PathInfo> & paths) {
    std::map<uint32_t, vector<uint32_t>> source_id_paths;
    for (size_t i = 0; i < paths.size(); ++i) {
        for (auto & id : paths[i].source_ids) {
            source_id_paths[id].emplace_back(i);
        }
    }
    pair<vector<vector<uint32_t>>, vector<uint32_t>> groups;
    for (auto it = source_id_paths.begin(); it != source_id_paths.end(); ++it) {
        if (it->second.empty()) continue;
        groups.second.emplace_back(1);
        auto it2 = it;
        ++it2;
        for (; it2 != source_id_paths.end(); ++it2) {
            if (!it2->second.empty() && it->second == it2->second) {
                groups.second.back()++;
                it2->second.clear();
            }
        }
        groups.first.emplace_back(it->second);
        it->second.clear();
    }
    return groups;
}

typedef std::map<vector<uint32_t>, double> SubsetMap;

// src/path_abundance_estimator.cpp:548-567
static void sampleGroupPathIndices(vector<vector<uint32_t>> * path_subset_samples, const Estimates & group_est,
                                   const vector<uint32_t> & group, uint32_t group_size, std::mt19937 * rng) {
    std::discrete_distribution<uint32_t> sampler(group_est.posteriors.begin(), group_est.posteriors.end());
    for (auto & sample : *path_subset_samples) {
        vector<uint32_t> set = group_est.path_group_sets.at(sampler(*rng));
        assert(set.size() == group_size);
        (void) group_size;
        std::sort(set.begin(), set.end());
        for (auto & g : set) sample.emplace_back(group.at(g));
    }
}

// src/path_abundance_estimator.cpp:569-606
static void selectPathSubsetIndices(SubsetMap * path_subset_samples, const Estimates & group_est,
                                    const vector<vector<uint32_t>> & path_groups, double min_hap_prob) {
    double sum_posterior = 0;
    for (size_t i = 0; i < group_est.posteriors.size(); ++i) {
        if (group_est.posteriors[i] >= min_hap_prob) {
            vector<uint32_t> path_subset;
            for (auto & group : group_est.path_group_sets[i]) {
                for (auto & path : path_groups.at(group)) path_subset.emplace_back(path);
            }
            std::sort(path_subset.begin(), path_subset.end());
            (*path_subset_samples)[path_subset] += group_est.posteriors[i];
            sum_posterior += group_est.posteriors[i];
        }
    }
    for (auto & s : *path_subset_samples) s.second /= sum_posterior;
}

// src/path_abundance_estimator.cpp:608-750 (ordered maps instead of sparsepp:
// output order = lexicographic in the path tuple; compare keyed).
static void inferPathSubsetAbundance(const Params & prm, Estimates * est, const vector<ReadRow> & rows,
                                     std::mt19937 * rng, const SubsetMap & path_subset_samples) {
    assert(est->noise_count == 0);
    assert(est->total_count == 0);
    for (auto & row : rows) est->total_count += row.read_count;

    std::map<vector<uint32_t>, pair<double, vector<double>>> path_group_estimates;
    double sum_hap_prob = 0;
    uint32_t subset_gibbs_samples = prm.num_gibbs_samples;
    double subset_gibbs_prob = 1;

    for (auto & path_subset : path_subset_samples) {
        if (path_subset.second < prm.min_hap_prob) continue;
        sum_hap_prob += path_subset.second;
        assert(!path_subset.first.empty());

        vector<uint32_t> collapsed;
        std::map<uint32_t, pair<uint32_t, uint32_t>> collapsed_index;  // path -> (column, multiplicity)
        collapsed.emplace_back(path_subset.first.front());
        collapsed_index.emplace(path_subset.first.front(), std::make_pair(0u, 1u));
        for (size_t i = 1; i < path_subset.first.size(); ++i) {
            if (path_subset.first[i] != collapsed.back()) {
                collapsed.emplace_back(path_subset.first[i]);
                collapsed_index.emplace(path_subset.first[i], std::make_pair(static_cast<uint32_t>(collapsed.size() - 1), 1u));
            } else {
                collapsed_index.at(path_subset.first[i]).second++;
            }
        }

        ColMatrix sP;
        vector<double> snoise, scounts;
        constructPartialProbabilityMatrix(&sP, &snoise, &scounts, rows, collapsed, est->paths.size());
        Estimates sub;
        sub.resetEstimates(sP.cols, 1);
        addNoiseAndNormalizeProbabilityMatrix(&sP, snoise);
        readCollapseProbabilityMatrix(&sP, &scounts, prm.prob_precision);
        sub.total_count = sumOf(scounts);
        uint32_t its = EMAbundanceEstimator(&sub, sP, scounts, prm.max_em_its, prm.max_rel_em_conv);
        recordEM(est, its, collapsed);

        if (subset_gibbs_samples > 0) {
            std::binomial_distribution<uint32_t> sampler(subset_gibbs_samples, std::min(1.0, path_subset.second / subset_gibbs_prob));
            uint32_t cur = sampler(*rng);
            subset_gibbs_samples -= cur;
            subset_gibbs_prob -= path_subset.second;
            if (cur > 0) {
                sub.gibbs_read_count_samples.emplace_back(CountSamples());
                sub.gibbs_read_count_samples.back().path_ids = collapsed;
                gibbsReadCountSampler(&sub, sP, scounts, abundance_gibbs_gamma, rng, cur, prm.gibbs_thin_its);
                est->gibbs_read_count_samples.emplace_back(std::move(sub.gibbs_read_count_samples.front()));
            }
        }

        est->noise_count += sub.noise_count * path_subset.second;

        std::map<uint32_t, vector<uint32_t>> subset_path_group_index;
        for (auto & path : path_subset.first) {
            subset_path_group_index[est->paths.at(path).group_id].emplace_back(path);
        }
        for (auto & path_group : subset_path_group_index) {
            assert(path_group.second.size() <= prm.ploidy);
            auto it = path_group_estimates.emplace(path_group.second,
                                                   pair<double, vector<double>>(0, vector<double>(path_group.second.size(), 0)));
            it.first->second.first += path_subset.second;
            for (size_t i = 0; i < path_group.second.size(); ++i) {
                auto ci = collapsed_index.find(path_group.second[i]);
                assert(ci != collapsed_index.end());
                it.first->second.second.at(i) += (sub.abundances.at(ci->second.first) * path_subset.second / ci->second.second);
            }
        }
    }

    for (auto & e : path_group_estimates) {
        est->path_group_sets.emplace_back(e.first);
        est->posteriors.emplace_back(e.second.first);
        est->abundances.insert(est->abundances.end(), e.second.second.begin(), e.second.second.end());
    }
    assert(sum_hap_prob < 1 || doubleCompare(sum_hap_prob, 1));
    est->noise_count += (1 - sum_hap_prob) * est->total_count;
}

static void groupPosteriors(const Params & prm, Estimates * group_est, ColMatrix * gP, vector<double> * gcounts,
                            const vector<uint32_t> & path_counts, std::mt19937 * rng) {
    // shared tail of :380-407 and :442-464 — split the noise column back out,
    // then Gibbs / Bounded / Full.
    const size_t R = gP->rows, Cn = gP->cols;
    vector<double> gnoise(gP->col(Cn - 1), gP->col(Cn - 1) + R);
    ColMatrix M(R, Cn - 1);
    std::copy(gP->v.begin(), gP->v.begin() + R * (Cn - 1), M.v.begin());
    if (prm.use_hap_gibbs) {
        estimatePathGroupPosteriorsGibbs(group_est, M, gnoise, *gcounts, path_counts, prm.ploidy, rng);
    } else if (prm.ploidy == 2) {
        calculatePathGroupPosteriorsBounded(group_est, M, gnoise, *gcounts, path_counts, prm.ploidy, prm.min_hap_prob);
    } else {
        calculatePathGroupPosteriorsFull(group_est, M, gnoise, *gcounts, path_counts, prm.ploidy);
    }
}

// src/path_abundance_estimator.cpp:428-471
static void inferAbundancesCollapsedGroups(const Params & prm, Estimates * est, const vector<ReadRow> & rows,
                                           std::mt19937 * rng) {
    est->resetEstimates(0, 0);
    if (rows.empty()) return;
    auto path_source_groups = findPathSourceGroups(est->paths);

    ColMatrix gP;
    vector<double> gnoise, gcounts;
    constructGroupedProbabilityMatrix(&gP, &gnoise, &gcounts, rows, path_source_groups.first, est->paths.size());
    addNoiseAndNormalizeProbabilityMatrix(&gP, gnoise);
    readCollapseProbabilityMatrix(&gP, &gcounts, prm.prob_precision);

    Estimates group_est;
    groupPosteriors(prm, &group_est, &gP, &gcounts, path_source_groups.second, rng);

    SubsetMap path_subset_samples;
    selectPathSubsetIndices(&path_subset_samples, group_est, path_source_groups.first, prm.min_hap_prob);
    inferPathSubsetAbundance(prm, est, rows, rng, path_subset_samples);
}

// src/path_abundance_estimator.cpp:356-426
static void inferAbundancesIndependentGroups(const Params & prm, Estimates * est, const vector<ReadRow> & rows,
                                             std::mt19937 * rng) {
    est->resetEstimates(0, 0);
    if (rows.empty()) return;
    auto path_groups = findPathGroups(est->paths);
    vector<vector<uint32_t>> path_subset_samples(std::floor(1 / prm.min_hap_prob));

    for (auto & group : path_groups) {
        ColMatrix gP;
        vector<double> gnoise, gcounts;
        constructPartialProbabilityMatrix(&gP, &gnoise, &gcounts, rows, group, est->paths.size());
        addNoiseAndNormalizeProbabilityMatrix(&gP, gnoise);
        readCollapseProbabilityMatrix(&gP, &gcounts, prm.prob_precision);
        vector<uint32_t> group_path_counts;
        for (size_t i = 0; i < group.size(); ++i) group_path_counts.emplace_back(est->paths.at(group[i]).source_count);
        Estimates group_est;
        groupPosteriors(prm, &group_est, &gP, &gcounts, group_path_counts, rng);
        sampleGroupPathIndices(&path_subset_samples, group_est, group, prm.ploidy, rng);
    }

    SubsetMap clustered;
    for (auto & s : path_subset_samples) {
        std::sort(s.begin(), s.end());
        clustered[s] += 1 / static_cast<double>(path_subset_samples.size());
    }
    inferPathSubsetAbundance(prm, est, rows, rng, clustered);
}

static vector<uint32_t> sourceCounts(const Estimates & est) {
    vector<uint32_t> path_counts;
    path_counts.reserve(est.paths.size());
    for (auto & p : est.paths) path_counts.emplace_back(p.source_count);
    return path_counts;
}

// src/path_posterior_estimator.cpp:35-71 (and :9-31 when ploidy == 1 is
// requested through PathPosteriorEstimator — same Full call with g = 1).
static void estimateHaplotypes(const Params & prm, Estimates * est, const vector<ReadRow> & rows, std::mt19937 * rng) {
    est->resetEstimates(0, 0);
    if (rows.empty()) return;
    ColMatrix P;
    vector<double> noise, counts;
    constructProbabilityMatrix(&P, &noise, &counts, rows, est->paths.size());
    auto path_counts = sourceCounts(*est);
    vector<PathInfo> keep = est->paths;
    if (prm.use_hap_gibbs) {
        estimatePathGroupPosteriorsGibbs(est, P, noise, counts, path_counts, prm.ploidy, rng);
    } else if (prm.ploidy == 2) {
        calculatePathGroupPosteriorsBounded(est, P, noise, counts, path_counts, prm.ploidy, min_rel_likelihood);
    } else {
        calculatePathGroupPosteriorsFull(est, P, noise, counts, path_counts, prm.ploidy);
    }
    est->paths = keep;
}

void estimate(const std::string & model, const Params & prm, Estimates * est, const vector<ReadRow> & rows,
              std::mt19937 * rng) {
    est->em_iters.clear();
    est->em_problem_paths.clear();
    if (model == "transcripts") {
        estimateTranscripts(prm, est, rows, rng);
    } else if (model == "strains") {
        estimateStrains(prm, est, rows, rng);
    } else if (model == "haplotype-transcripts") {
        if (!prm.ind_hap_inference) {
            inferAbundancesCollapsedGroups(prm, est, rows, rng);
        } else {
            inferAbundancesIndependentGroups(prm, est, rows, rng);
        }
    } else if (model == "haplotypes") {
        estimateHaplotypes(prm, est, rows, rng);
    } else {
        assert(false);
    }
}

// ---------------------------------------------------------------------------
// caller-side row ordering and merging
// ---------------------------------------------------------------------------

// src/read_path_probabilities.cpp:283-322
bool rowLess(const ReadRow & lhs, const ReadRow & rhs) {
    if (!doubleCompare(lhs.noise_prob, rhs.noise_prob)) {
        return (lhs.noise_prob < rhs.noise_prob);
    }
    if (lhs.path_probs.size() != rhs.path_probs.size()) {
        return (lhs.path_probs.size() < rhs.path_probs.size());
    }
    for (size_t i = 0; i < lhs.path_probs.size(); ++i) {
        if (!doubleCompare(lhs.path_probs[i].first, rhs.path_probs[i].first)) {
            return (lhs.path_probs[i].first < rhs.path_probs[i].first);
        }
        if (lhs.path_probs[i].second.size() != rhs.path_probs[i].second.size()) {
            return (lhs.path_probs[i].second.size() < rhs.path_probs[i].second.size());
        }
        for (size_t j = 0; j < lhs.path_probs[i].second.size(); ++j) {
            if (lhs.path_probs[i].second[j] != rhs.path_probs[i].second[j]) {
                return (lhs.path_probs[i].second[j] < rhs.path_probs[i].second[j]);
            }
        }
    }
    if (lhs.read_count != rhs.read_count) {
        return (lhs.read_count < rhs.read_count);
    }
    return false;
}

// src/read_path_probabilities.cpp:223-250
bool quickMergeIdentical(ReadRow * a, const ReadRow & b, const double prob_precision) {
    if (std::abs(a->noise_prob - b.noise_prob) >= prob_precision) {
        return false;
    }
    if (a->path_probs.size() == b.path_probs.size()) {
        for (size_t i = 0; i < a->path_probs.size(); ++i) {
            if (std::abs(a->path_probs[i].first - b.path_probs[i].first) >= prob_precision) {
                return false;
            }
            if (a->path_probs[i].second != b.path_probs[i].second) {
                return false;
            }
        }
        a->read_count += b.read_count;
        return true;
    }
    return false;
}

// src/main.cpp:953-973
void sortAndMergeRows(vector<ReadRow> * rows, const double prob_precision) {
    std::sort(rows->begin(), rows->end(), rowLess);
    if (!rows->empty()) {
        uint32_t prev_unique_probs_idx = 0;
        for (size_t i = 1; i < rows->size(); ++i) {
            if (!quickMergeIdentical(&rows->at(prev_unique_probs_idx), rows->at(i), prob_precision)) {
                if (prev_unique_probs_idx + 1 < i) {
                    rows->at(prev_unique_probs_idx + 1) = rows->at(i);
                }
                prev_unique_probs_idx++;
            }
        }
        rows->resize(prev_unique_probs_idx + 1);
    }
}

// ---------------------------------------------------------------------------
// row construction  (src/read_path_probabilities.cpp:39-221, src/fragment_length_dist.cpp)
// ---------------------------------------------------------------------------
static const double score_log_base = 1.383325268738;  // src/utils.hpp:83
static const double noise_score_log_base = 1e-6;      // src/utils.hpp:84
static const double kPi = 3.141592653589793238462643383279;

// src/utils.hpp:143-161
static double Phi(double z) {
    static const double root_1_2 = std::sqrt(0.5);
    const double x = z * root_1_2;
    const double a = std::fabs(x);
    double y;
    if (a < root_1_2) {
        y = 0.5 + 0.5 * std::erf(x);
    } else {
        y = 0.5 * std::erfc(a);
        if (x > 0) y = 1.0 - y;
    }
    return y;
}

// src/utils.hpp:165-194
static double log_Phi(double z) {
    if (z > 6.0) return -Phi(-z);
    if (z > -20.0) return std::log(Phi(z));
    double last_total = 0, right_hand_side = 1, numerator = 1, denom_factor = 1;
    const double denom_cons = 1.0 / (z * z);
    long sign = 1, i = 0;
    const double log_LHS = -0.5 * z * z - std::log(-z) - 0.5 * std::log(2 * kPi);
    while (std::fabs(last_total - right_hand_side) > std::numeric_limits<double>::epsilon()) {
        i += 1;
        last_total = right_hand_side;
        sign = -sign;
        denom_factor *= denom_cons;
        numerator *= 2 * i - 1;
        right_hand_side += sign * numerator * denom_factor;
    }
    return log_LHS + std::log(right_hand_side);
}

// src/utils.hpp:206-220
static double log_normal_pdf(double x, double m, double s) {
    static const double inv_sqrt_2pi = 0.3989422804014327;
    const double z = (x - m) / s;
    return std::log(inv_sqrt_2pi) - std::log(s) - 0.5 * z * z;
}

static double log_skew_normal_pdf(double x, double m, double s, double a) {
    static const double log_const = std::log(2.0 / std::sqrt(2.0 * kPi));
    const double z = (x - m) / s;
    return log_const + log_Phi(a * z) - std::log(s) - 0.5 * z * z;
}

// src/fragment_length_dist.cpp:21-27,396-427
FragmentLengthDist::FragmentLengthDist(const double loc_in, const double scale_in, const double shape_in,
                                       const uint32_t sd_max_multi) : loc(loc_in), scale(scale_in), shape(shape_in) {
    assert(loc >= 0 && scale > 0);
    const double delta = shape / std::sqrt(1.0 + shape * shape);
    const double sd = scale * (1.0 - 2.0 * delta * delta / kPi);
    max_length = std::ceil(loc + sd * sd_max_multi);
    assert(max_length > 0);
    log_prob_buffer = vector<double>(max_length + 1);
    for (size_t i = 0; i < log_prob_buffer.size(); ++i) {
        log_prob_buffer[i] = doubleCompare(shape, 0.0) ? log_normal_pdf(i, loc, scale) : log_skew_normal_pdf(i, loc, scale, shape);
    }
}

// src/fragment_length_dist.cpp:385-394
double FragmentLengthDist::logProb(const uint32_t value) const {
    if (value < log_prob_buffer.size()) return log_prob_buffer[value];
    if (doubleCompare(shape, 0.0)) return log_normal_pdf(value, loc, scale);
    return log_skew_normal_pdf(value, loc, scale, shape);
}

// src/read_path_probabilities.cpp:39-67
vector<double> calcAlignPathLogProbs(const vector<AlignPath> & align_paths, const FragmentLengthDist & fld,
                                     const bool is_single_end) {
    assert(align_paths.size() > 1);
    assert(align_paths.back().path_idx.empty());
    assert(align_paths.back().frag_length == 0);
    assert(align_paths.back().align_length == 0);
    assert(align_paths.back().score_sum <= 0);

    vector<double> align_paths_log_probs;
    align_paths_log_probs.reserve(align_paths.size());
    for (size_t i = 0; i < align_paths.size() - 1; ++i) {
        const AlignPath & align_path = align_paths[i];
        assert(align_paths.front().min_mapq == align_path.min_mapq);
        align_paths_log_probs.emplace_back(align_path.score_sum * score_log_base);
        if (!is_single_end) {
            align_paths_log_probs.back() += fld.logProb(align_path.frag_length);
        }
    }
    align_paths_log_probs.emplace_back(align_paths.back().score_sum * noise_score_log_base);
    return align_paths_log_probs;
}

// src/read_path_probabilities.cpp:74-221
ReadRow addPathProbs(const uint32_t read_count, const double prob_precision, const vector<AlignPath> & align_paths,
                     const vector<PathInfo> & cluster_paths, const FragmentLengthDist & fld, const bool is_single_end,
                     const double min_noise_prob, const bool collapse_groups, const vector<uint32_t> & path_group,
                     const uint32_t num_groups) {
    ReadRow row;
    row.read_count = read_count;
    row.noise_prob = 1;
    assert(align_paths.size() > 1);

    if (align_paths.front().min_mapq > 0) {
        // Utils::phred_to_prob, src/utils.hpp:131-133
        row.noise_prob = std::max(prob_precision, std::max(min_noise_prob, std::pow(10, -((double) align_paths.front().min_mapq) / 10)));
        assert(row.noise_prob < 1 && row.noise_prob > 0);

        auto align_paths_log_probs = calcAlignPathLogProbs(align_paths, fld, is_single_end);
        row.noise_prob += (1 - row.noise_prob) * std::exp(align_paths_log_probs.back());

        if (align_paths.back().score_sum == 0) {
            assert(doubleCompare(row.noise_prob, 1));
            return row;
        }

        vector<double> read_path_log_probs(cluster_paths.size(), kLowest);
        vector<double> read_path_max_align_lengths(cluster_paths.size(), 0);

        for (size_t i = 0; i + 1 < align_paths.size(); ++i) {
            assert(!align_paths[i].path_idx.empty());
            for (auto path_idx: align_paths[i].path_idx) {
                assert(path_idx < cluster_paths.size());
                if (doubleCompare(cluster_paths[path_idx].effective_length, 0)) {
                    assert(doubleCompare(read_path_log_probs[path_idx], kLowest));
                } else {
                    const double log_prob = align_paths_log_probs[i] - std::log(cluster_paths[path_idx].effective_length);
                    assert(align_paths[i].align_length > 0);
                    if (align_paths[i].align_length > read_path_max_align_lengths[path_idx]) {
                        read_path_log_probs[path_idx] = log_prob;
                        read_path_max_align_lengths[path_idx] = align_paths[i].align_length;
                    } else if (align_paths[i].align_length == read_path_max_align_lengths[path_idx]) {
                        read_path_log_probs[path_idx] = std::max(read_path_log_probs[path_idx], log_prob);
                    }
                }
            }
        }

        if (collapse_groups) {
            assert(path_group.size() == cluster_paths.size());
            assert(num_groups > 0);
            vector<double> read_path_log_probs_groups(num_groups, kLowest);
            for (size_t i = 0; i < read_path_log_probs.size(); ++i) {
                double & slot = read_path_log_probs_groups.at(path_group[i]);
                slot = add_log(slot, read_path_log_probs[i] + std::log(cluster_paths[i].source_count));
            }
            read_path_log_probs = read_path_log_probs_groups;
        }

        double read_path_log_probs_sum = kLowest;
        for (auto & log_prob: read_path_log_probs) {
            read_path_log_probs_sum = add_log(read_path_log_probs_sum, log_prob);
        }
        double low_prob_sum = 0;
        assert(read_path_log_probs_sum > kLowest);

        auto & path_probs = row.path_probs;
        for (size_t i = 0; i < read_path_log_probs.size(); ++i) {
            read_path_log_probs[i] = std::exp(read_path_log_probs[i] - read_path_log_probs_sum);
            if (read_path_log_probs[i] >= prob_precision) {
                auto it = path_probs.begin();
                while (it != path_probs.end()) {
                    if (std::abs(it->first - read_path_log_probs[i]) < prob_precision) {
                        it->first = ((it->first * it->second.size() + read_path_log_probs[i]) / (it->second.size() + 1));
                        it->second.emplace_back(i);
                        break;
                    }
                    ++it;
                }
                if (it == path_probs.end()) {
                    path_probs.emplace_back(read_path_log_probs[i], vector<uint32_t>({static_cast<uint32_t>(i)}));
                }
            } else {
                low_prob_sum += read_path_log_probs[i];
            }
        }
        for (auto & prob: path_probs) prob.first *= (1 - row.noise_prob);
        row.noise_prob += low_prob_sum * (1 - row.noise_prob);
        std::sort(path_probs.begin(), path_probs.end());
    }
    return row;
}

// ---------------------------------------------------------------------------
// path clustering  (src/path_clusters.cpp:12-86, 163-207)
// ---------------------------------------------------------------------------
vector<vector<uint32_t>> createPathClusters(const uint32_t num_paths, const vector<vector<uint32_t>> & id_sets,
                                            vector<uint32_t> * path_to_cluster_index) {
    // constructor :12-86 — connected_paths: anchor = first id of the set, every other id is linked to it both ways
    vector<std::set<uint32_t>> connected_paths(num_paths);
    for (auto & ids: id_sets) {
        assert(!ids.empty());
        const uint32_t anchor_path_id = ids.front();
        for (auto path_id: ids) {
            if (anchor_path_id != path_id) {
                if (connected_paths.at(anchor_path_id).emplace(path_id).second) {
                    connected_paths.at(path_id).emplace(anchor_path_id);
                }
            }
        }
    }
    // createPathClusters :163-207
    const uint32_t unset = static_cast<uint32_t>(-1);
    *path_to_cluster_index = vector<uint32_t>(num_paths, unset);
    vector<vector<uint32_t>> cluster_to_paths_index;
    for (uint32_t i = 0; i < num_paths; ++i) {
        if (path_to_cluster_index->at(i) == unset) {
            std::queue<uint32_t> search_queue;
            search_queue.push(i);
            cluster_to_paths_index.emplace_back(vector<uint32_t>());
            while (!search_queue.empty()) {
                auto cur_path = search_queue.front();
                const bool is_first_visit = (path_to_cluster_index->at(cur_path) == unset);
                assert(is_first_visit || path_to_cluster_index->at(cur_path) == cluster_to_paths_index.size() - 1);
                path_to_cluster_index->at(cur_path) = cluster_to_paths_index.size() - 1;
                if (is_first_visit) {
                    cluster_to_paths_index.back().emplace_back(cur_path);
                    for (auto & next_path: connected_paths.at(cur_path)) {
                        if (path_to_cluster_index->at(next_path) == unset) {
                            search_queue.push(next_path);
                        }
                    }
                }
                search_queue.pop();
            }
            std::sort(cluster_to_paths_index.back().begin(), cluster_to_paths_index.back().end());
        }
    }
    return cluster_to_paths_index;
}

}  // namespace rpvg_oracle
