// rpvg_oracle.hpp — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
//
// A plain C++17 restatement (no Eigen, no sparsepp) of the rpvg inference hot
// path: the per-cluster EM abundance / haplotype-posterior estimators.  Only
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it;
// the product path (rpvg_amd/) never links, imports or calls anything in here.
//
// PARITY STATUS: "parity unpinned" for EM / posteriors / nested inference: the
// reference cannot be compiled in this image (Eigen, sparsepp, gbwt, protobuf,
// libvgio, htslib absent — SURVEY.md F2) and its own test-suite pins nothing on
// this path except MinimumPathAbundanceEstimator::weightedMinimumPathCover
// (src/tests/path_abundance_estimator_test.cpp:8-28), which IS pinned here
// (tests/test_oracle_golden.py).  PINNED as well: row construction
// (addPathProbs + quickMergeIdentical), against every case of the reference's
// src/tests/read_path_probabilities_test.cpp:9-205 (tests/test_row_construction.py).  The restatement is additionally cross-checked
// against an independent numpy restatement (oracle/np_oracle.py) and the
// hand-derivable known-answer cases of SURVEY.md §8c.
//
// Every function cites the reference file:line it follows (paths relative to
// the reference checkout, src/...).
#ifndef RPVG_ORACLE_HPP
#define RPVG_ORACLE_HPP

#include <cstdint>
#include <map>
#include <random>
#include <string>
#include <utility>
#include <vector>

namespace rpvg_oracle {

// Dense column-major double matrix (Utils::ColMatrixXd, src/utils.hpp:58-65).
struct ColMatrix {
    size_t rows = 0, cols = 0;
    std::vector<double> v;
    ColMatrix() {}
    ColMatrix(size_t r, size_t c) : rows(r), cols(c), v(r * c, 0.0) {}
    double & at(size_t i, size_t j) { return v[j * rows + i]; }
    double at(size_t i, size_t j) const { return v[j * rows + i]; }
    const double * col(size_t j) const { return v.data() + j * rows; }
};

// One merged class of read pairs (ReadPathProbabilities,
// src/read_path_probabilities.hpp:19-44): count, noise probability and
// (probability -> cluster-local path indices) groups sorted ascending.
struct ReadRow {
    uint32_t read_count = 0;
    double noise_prob = 1;
    std::vector<std::pair<double, std::vector<uint32_t>>> path_probs;
};

// src/path_cluster_estimates.hpp:15-33 (source_ids kept as a sorted vector).
struct PathInfo {
    uint32_t group_id = 0;
    uint32_t source_count = 1;
    std::vector<uint32_t> source_ids;
    double effective_length = 0;
};

// src/path_cluster_estimates.hpp:35-43.
struct CountSamples {
    std::vector<uint32_t> path_ids;
    std::vector<double> noise_samples;
    std::vector<double> abundance_samples;
};

// src/path_cluster_estimates.hpp:45-111.  em_iters is oracle-only
// instrumentation: the number of EM iterations of every EMAbundanceEstimator
// call in the order the calls were made (used for iteration-count parity).
struct Estimates {
    std::vector<PathInfo> paths;
    std::vector<std::vector<uint32_t>> path_group_sets;
    std::vector<double> posteriors;
    std::vector<double> abundances;
    double noise_count = 0;
    double total_count = 0;
    std::vector<CountSamples> gibbs_read_count_samples;
    std::vector<uint32_t> em_iters;
    std::vector<std::vector<uint32_t>> em_problem_paths;  // column paths of each EM call

    void resetEstimates(uint32_t num_components, uint32_t group_size);
};

// Every knob main.cpp:364-419 exposes for this path, same defaults.
struct Params {
    uint32_t max_em_its = 10000;        // --max-em-its      main.cpp:416
    double max_rel_em_conv = 0.001;     // --max-rel-em-conv main.cpp:417
    uint32_t num_gibbs_samples = 0;     // -n                main.cpp:415
    uint32_t gibbs_thin_its = 25;       // --gibbs-thin-its  main.cpp:418
    double prob_precision = 1e-8;       // --prob-precision  main.cpp:402
    uint32_t ploidy = 2;                // -y                main.cpp:407
    double min_hap_prob = 0.001;        // --min-hap-prob    main.cpp:409
    bool ind_hap_inference = false;     // --ind-hap-inference main.cpp:410
    bool use_hap_gibbs = false;         // --use-hap-gibbs   main.cpp:411
};

// ---- scalar helpers (src/utils.hpp:81-117, 300-302) -----------------------
bool doubleCompare(double a, double b);
uint32_t numPermutations(std::vector<uint32_t> values);
double add_log(double log_x, double log_y);

// ---- matrix builders (src/path_estimator.cpp:55-166, 197-259) --------------
void constructProbabilityMatrix(ColMatrix * P, std::vector<double> * noise, std::vector<double> * counts,
                                const std::vector<ReadRow> & rows, uint32_t num_paths);
void constructPartialProbabilityMatrix(ColMatrix * P, std::vector<double> * noise, std::vector<double> * counts,
                                       const std::vector<ReadRow> & rows, const std::vector<uint32_t> & path_ids,
                                       uint32_t num_paths);
void constructGroupedProbabilityMatrix(ColMatrix * P, std::vector<double> * noise, std::vector<double> * counts,
                                       const std::vector<ReadRow> & rows,
                                       const std::vector<std::vector<uint32_t>> & path_groups, uint32_t num_paths);
void addNoiseAndNormalizeProbabilityMatrix(ColMatrix * P, const std::vector<double> & noise);
void readCollapseProbabilityMatrix(ColMatrix * P, std::vector<double> * counts, double prob_precision);

// ---- EM (src/path_abundance_estimator.cpp:47-114) --------------------------
// Returns the number of iterations executed.
uint32_t EMAbundanceEstimator(Estimates * est, const ColMatrix & P, const std::vector<double> & counts,
                              uint32_t max_em_its, double max_rel_em_conv);

// ---- Gibbs read-count sampler (src/path_abundance_estimator.cpp:116-212) ---
void gibbsReadCountSampler(Estimates * est, const ColMatrix & P, const std::vector<double> & counts, double gamma,
                           std::mt19937 * rng, uint32_t num_samples, uint32_t gibbs_thin_its);

// ---- group posteriors (src/path_estimator.cpp:315-589) ---------------------
void calculatePathGroupPosteriorsFull(Estimates * est, const ColMatrix & P, const std::vector<double> & noise,
                                      const std::vector<double> & counts, const std::vector<uint32_t> & path_counts,
                                      uint32_t group_size);
void calculatePathGroupPosteriorsBounded(Estimates * est, const ColMatrix & P, const std::vector<double> & noise,
                                         const std::vector<double> & counts, const std::vector<uint32_t> & path_counts,
                                         uint32_t group_size, double min_rel_likelihood);
void estimatePathGroupPosteriorsGibbs(Estimates * est, const ColMatrix & P, const std::vector<double> & noise,
                                      const std::vector<double> & counts, const std::vector<uint32_t> & path_counts,
                                      uint32_t group_size, std::mt19937 * rng);

// ---- minimum path cover (src/path_abundance_estimator.cpp:297-340) ---------
// cover is row-major bool R x N.
std::vector<uint32_t> weightedMinimumPathCover(const std::vector<uint8_t> & cover, size_t num_rows, size_t num_paths,
                                               const std::vector<double> & read_counts,
                                               const std::vector<double> & path_weights);

// ---- the estimate() entry points -------------------------------------------
// model: "transcripts" | "strains" | "haplotype-transcripts" | "haplotypes"
// (factory main.cpp:766-788).  est->paths must be filled by the caller.
void estimate(const std::string & model, const Params & prm, Estimates * est, const std::vector<ReadRow> & rows,
              std::mt19937 * rng);

// ---- caller-side row merging (src/main.cpp:953-973) ------------------------
bool rowLess(const ReadRow & lhs, const ReadRow & rhs);                        // read_path_probabilities.cpp:283-322
bool quickMergeIdentical(ReadRow * a, const ReadRow & b, double prob_precision);  // :223-250
void sortAndMergeRows(std::vector<ReadRow> * rows, double prob_precision);

// ---- row construction: the step before the path -----
// src/fragment_length_dist.cpp:19-27,385-427 (normal / skew-normal; log-density table up to max_length).
struct FragmentLengthDist {
    double loc = 0, scale = 0, shape = 0;
    uint32_t max_length = 0;
    std::vector<double> log_prob_buffer;
    FragmentLengthDist() {}
    FragmentLengthDist(double mean, double sd, uint32_t sd_max_multi) : FragmentLengthDist(mean, sd, 0.0, sd_max_multi) {}
    FragmentLengthDist(double loc_in, double scale_in, double shape_in, uint32_t sd_max_multi);
    double logProb(uint32_t value) const;
};

// The AlignmentPath fields addPathProbs reads (src/alignment_path.hpp:22-39) with the gbwt search state replaced
// by the path ids it locates, already mapped to cluster-local indices (align_paths_ids + clustered_path_index).
struct AlignPath {
    uint8_t min_mapq = 0;
    int32_t score_sum = 0;
    uint16_t align_length = 0;
    uint16_t frag_length = 0;
    std::vector<uint32_t> path_idx;
};

// src/read_path_probabilities.cpp:39-67
std::vector<double> calcAlignPathLogProbs(const std::vector<AlignPath> & align_paths, const FragmentLengthDist & fld,
                                          bool is_single_end);
// src/read_path_probabilities.cpp:74-221.  align_paths.back() is the noise entry.  path_group (one cluster-local
// group index per path) and num_groups are only read when collapse_groups is set.
ReadRow addPathProbs(uint32_t read_count, double prob_precision, const std::vector<AlignPath> & align_paths,
                     const std::vector<PathInfo> & cluster_paths, const FragmentLengthDist & fld, bool is_single_end,
                     double min_noise_prob, bool collapse_groups = false,
                     const std::vector<uint32_t> & path_group = std::vector<uint32_t>(), uint32_t num_groups = 0);


// ---- path clustering --------------------------------------------------------------------
// src/path_clusters.cpp:12-86 (every id set connects its members to its first member) + :163-207
// (createPathClusters: BFS per unvisited path in ascending id; members sorted).  Returns cluster_to_paths_index and
// fills path_to_cluster_index.
std::vector<std::vector<uint32_t>> createPathClusters(uint32_t num_paths, const std::vector<std::vector<uint32_t>> & id_sets,
                                                      std::vector<uint32_t> * path_to_cluster_index);

}  // namespace rpvg_oracle

#endif
