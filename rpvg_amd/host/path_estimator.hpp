// GPU-backed PathEstimator: the reference's abstract estimator interface
// (src/path_estimator.hpp:16-49) kept as is — same constructor argument,
// same virtual estimate() — with the per-cluster Eigen algebra replaced by
// batched calls into the C ABI of the GPU engine (include/rpvg_hip.h).
//
// Two ways in:
//   estimate()       one cluster, synchronous; signature of the reference
//                    (src/path_estimator.hpp:23), callable from the reference's
//                    own loop (src/main.cpp:977) — from all threads of its OpenMP
//                    team at once: the calls that are in flight together are
//                    joined into batches (CallCombiner, path_estimator.cpp), up to
//                    three of them on the GPU at a time; every caller returns
//                    with its own cluster's estimates and its own generator
//                    advanced.  A lone caller gets a batch of one.
//   estimateBatch()  all clusters of a batch at once; what a GPU wants, and
//                    what the reference's `omp parallel for` over clusters
//                    (src/main.cpp:829) becomes.
#ifndef RPVG_AMD_PATH_ESTIMATOR_HPP
#define RPVG_AMD_PATH_ESTIMATOR_HPP

#include <cstdint>
#include <functional>
#include <memory>
#include <mutex>
#include <random>
#include <vector>

#include "hip_engine.hpp"
#include "path_cluster_estimates.hpp"
#include "read_path_probabilities.hpp"

namespace rpvg_amd {

// One posterior problem over the columns of a group matrix of one cluster:
// column g is the set of cluster-local paths column_paths[g] (a single path
// for `-i haplotypes`, a haplotype's HST set for `-i haplotype-transcripts`).
// Stored flat (offsets + one array): a batch holds hundreds of thousands of columns.
struct GroupPosteriorProblem {

    uint32_t cluster = 0;

    std::vector<uint32_t> column_path_off;  // empty (no allocation: thousands are constructed at once) or [columns + 1]
    std::vector<uint32_t> column_path;
    std::vector<uint32_t> column_counts;

    uint32_t numColumns() const { return column_counts.size(); }

    // the leading offset, before the first column is added through column_path_off directly
    void beginColumns(const size_t expected_columns = 0) {

        column_path_off.reserve(expected_columns + 1);

        if (column_path_off.empty()) {

            column_path_off.emplace_back(0);
        }
    }

    void addColumn(const uint32_t * first_path, const uint32_t * last_path, const uint32_t count) {

        beginColumns();
        column_path.insert(column_path.end(), first_path, last_path);
        column_path_off.emplace_back(column_path.size());
        column_counts.emplace_back(count);
    }

    // columns 0 .. n-1, column j = path j alone (the raw path posteriors: src/path_posterior_estimator.cpp:9-31): only the
    // counts are kept — the lists 0 | 1 | 2 ... are implied (single_paths), the device writes them itself
    // (rpvg_hip_groups_build_single_paths), and nothing on the host walks the lists of such a problem
    bool single_paths = false;

    template <typename CountOf>
    void singlePathColumns(const uint32_t n, CountOf count_of) {

        single_paths = true;
        column_path_off.clear();
        column_path.clear();
        column_counts.resize(n);

        for (uint32_t j = 0; j < n; ++j) {

            column_counts[j] = count_of(j);
        }
    }

    // the implied lists written out (a caller that mixes such problems with others)
    void materialiseSinglePaths() {

        if (single_paths) {

            const uint32_t n = column_counts.size();
            column_path_off.resize(static_cast<size_t>(n) + 1);
            column_path.resize(n);

            for (uint32_t j = 0; j < n; ++j) {

                column_path_off[j] = j;
                column_path[j] = j;
            }

            column_path_off[n] = n;
            single_paths = false;
        }
    }

    const uint32_t * columnBegin(const uint32_t column) const { return column_path.data() + column_path_off[column]; }
    const uint32_t * columnEnd(const uint32_t column) const { return column_path.data() + column_path_off[column + 1]; }
};

// Group sets (multisets of `group_size` column indices, flat) and their posteriors.
struct GroupPosteriors {

    uint32_t group_size = 0;

    std::vector<uint32_t> members;
    std::vector<double> posteriors;

    size_t size() const { return posteriors.size(); }
    const uint32_t * set(const size_t idx) const { return members.data() + idx * group_size; }
};

class PathEstimator {

    public:

        PathEstimator(const double prob_precision_in, std::shared_ptr<HipEngine> engine_in = HipEngine::processDefault());
        virtual ~PathEstimator() {};

        // Reference interface (src/path_estimator.hpp:23).  path_cluster_estimates->paths
        // must be filled by the caller, as src/main.cpp:855-887 does.
        virtual void estimate(PathClusterEstimates * path_cluster_estimates, const std::vector<ReadPathProbabilities> & cluster_probs, std::mt19937 * mt_rng);

        // Batched form.  path_cluster_estimates->at(i).paths must be filled for
        // every cluster i of the device batch; cluster i draws from rngs->at(i)
        // (rngs may be null for models that consume no random numbers).
        virtual void estimateBatch(std::vector<PathClusterEstimates> * path_cluster_estimates, const DeviceClusterBatch & cluster_batch, std::vector<std::mt19937> * rngs) = 0;

        // The host side of the device sampler reads and writes the state of the callers' std::mt19937 directly where the
        // library's layout allows it (path_estimator.cpp, GeneratorLayout).  1: it does, and `rounds` generators at random
        // positions gave the same next 624 outputs and went on with the same words as through the standard's interface;
        // 0: the portable route is in use; -1: a mismatch (which the first use would have turned into the portable route).
        static int generatorStateSelfTest(uint32_t rounds);

        // Whether estimateBatch() draws random numbers with the current settings (the reference's default
        // `transcripts`, `haplotype-transcripts` and `haplotypes` runs draw none: SURVEY.md F7).
        virtual bool usesRandomNumbers() const { return false; }

        // Whether estimateBatch() reads the haplotype columns of a batch (DeviceClusterBatch::hasSourceColumns: the batch was
        // uploaded with PathInfo::group_id and PathInfo::source_ids, and the device formed the columns behind the copy) with the
        // current settings: estimate() ships the paths' ids with a cluster only then.
        virtual bool wantsSourceColumns() const { return false; }

        // The haplotype columns of a cluster's paths (findPathSourceGroups, src/path_abundance_estimator.cpp:493-546) for an
        // estimator that wantsSourceColumns(): estimate() forms them on the calling thread, as the reference does, and ships them
        // with the cluster — the upload of the calls' batch then waits for nothing but its own kernel.  False: not formed.
        virtual bool sourceColumnsOf(GroupPosteriorProblem * columns, const std::vector<PathInfo> & paths) const { (void) columns; (void) paths; return false; }

        // Same, seeding cluster i with mt19937(rng_seed + i) as src/main.cpp:976 does.
        void estimateBatchSeeded(std::vector<PathClusterEstimates> * path_cluster_estimates, const DeviceClusterBatch & cluster_batch, const uint32_t rng_seed);

        // estimate() without the call combiner: the cluster as a batch of one, on the calling thread.
        void estimateAlone(PathClusterEstimates * path_cluster_estimates, const std::vector<ReadPathProbabilities> & cluster_probs, std::mt19937 * mt_rng);

    protected:

        const double prob_precision;
        const std::shared_ptr<HipEngine> engine;

        // Runs `work` on `clusters` (cluster indices of a batch, ordered by size) cut into the engine's host
        // lanes (pipeline_lanes.hpp); returns when every lane is done, rethrowing the first failure.
        // `work` receives its clusters and a callback to invoke once its first host phase is done and its first
        // device stage is about to start: the next lane begins then (LaneStagger).
        void runInLanes(const std::vector<uint32_t> & clusters, const std::function<void(const std::vector<uint32_t> &, const std::function<void()> &)> & work) const;

        // calculatePathGroupPosteriorsFull (src/path_estimator.cpp:332-377) for many
        // problems at once; log-likelihood contractions on the GPU.
        void calculatePathGroupPosteriorsFull(std::vector<GroupPosteriors> * group_posteriors, const DeviceClusterBatch & cluster_batch, const std::vector<GroupPosteriorProblem> & problems, const uint32_t group_size, const bool normalise) const;

        // calculatePathGroupPosteriorsBounded (src/path_estimator.cpp:379-473) for many
        // problems at once: one GPU workgroup per problem walks the branch-and-bound in
        // the reference's sequential order (rpvg_hip_bounded_pair_posteriors).
        void calculatePathGroupPosteriorsBounded(std::vector<GroupPosteriors> * group_posteriors, const DeviceClusterBatch & cluster_batch, const std::vector<GroupPosteriorProblem> & problems, const uint32_t group_size, const double min_rel_likelihood, const bool normalise) const;

        // estimatePathGroupPosteriorsGibbs (src/path_estimator.cpp:475-589) for many problems at once.  The
        // chains run on the host with the reference's generator and distributions (so a problem consumes its
        // generator exactly as the reference would); every conditional a chain has not seen yet — N
        // log-likelihood contractions — is evaluated on the GPU, all problems advancing in lock-step so that
        // one device call serves the pending conditionals of every problem.  rngs.at(i) drives problem i.
        void estimatePathGroupPosteriorsGibbs(std::vector<GroupPosteriors> * group_posteriors, const DeviceClusterBatch & cluster_batch, const std::vector<GroupPosteriorProblem> & problems, const uint32_t group_size, const bool normalise, const std::vector<std::mt19937 *> & rngs) const;

        // The same search driven from the host: pair log-likelihoods are fetched from the
        // GPU a block of first paths at a time and the reference's sequential pruning is
        // replayed on them here.  Kept as the cross-check of the on-device search
        // (RPVG_AMD_HOST_BOUNDED=1 selects it).
        void calculatePathGroupPosteriorsBoundedHostDriven(std::vector<GroupPosteriors> * group_posteriors, const DeviceClusterBatch & cluster_batch, const std::vector<GroupPosteriorProblem> & problems, const uint32_t group_size, const double min_rel_likelihood, const bool normalise) const;

        // The diploid search, the selection of path subsets and the EM of every retained subset in ONE device call
        // (rpvg_hip_nested_subset_em; src/path_abundance_estimator.cpp:440-469 and :625-671): problem i of `problems`
        // is matrix i of the result.  False when the device does not take the input (the caller runs the separate
        // calls then); the result owns page-locked host memory.
        class SubsetEmResult {

            public:

                SubsetEmResult() : result(nullptr) {}
                ~SubsetEmResult();

                SubsetEmResult(const SubsetEmResult &) = delete;
                SubsetEmResult & operator=(const SubsetEmResult &) = delete;

                rpvg_hip_subset_em_view view;

            private:

                friend class PathEstimator;
                rpvg_hip_subset_em * result;
        };

        bool nestedSubsetAbundances(SubsetEmResult * result, const DeviceClusterBatch & cluster_batch, const std::vector<GroupPosteriorProblem> & problems, const double min_rel_likelihood, const double min_hap_prob, const uint32_t max_em_its, const double max_rel_em_conv) const;

        // The same from the haplotype columns the batch holds on the device (DeviceClusterBatch::hasSourceColumns()): matrix i of
        // the result belongs to clusters.at(i); nothing but the cluster list goes up, and the result carries the
        // posterior-weighted merge of the solutions as well (rpvg_hip_subset_em_view::set_*).
        bool nestedSubsetAbundances(SubsetEmResult * result, const DeviceClusterBatch & cluster_batch, const std::vector<uint32_t> & clusters, const double min_rel_likelihood, const double min_hap_prob, const uint32_t max_em_its, const double max_rel_em_conv) const;

        // src/path_estimator.cpp:315-330
        static std::vector<double> calcPathLogFrequences(const std::vector<uint32_t> & path_counts);


    private:

        // Joins the estimate() calls of concurrent threads into batches (path_estimator.cpp).
        class CallCombiner;
        std::shared_ptr<CallCombiner> call_combiner;
        std::mutex call_combiner_mutex;
};

}

#endif
