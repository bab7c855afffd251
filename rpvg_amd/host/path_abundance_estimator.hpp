// GPU-backed abundance estimators with the constructor parameter lists of the
// reference's classes (src/path_abundance_estimator.hpp:18-76):
//   PathAbundanceEstimator        `-i transcripts`
//   NestedPathAbundanceEstimator  `-i haplotype-transcripts`
// The EM solves of a whole batch of clusters — one per cluster for
// `transcripts`, one per retained diplotype path subset for
// `haplotype-transcripts` — are issued as ONE call into the GPU engine.
#ifndef RPVG_AMD_PATH_ABUNDANCE_ESTIMATOR_HPP
#define RPVG_AMD_PATH_ABUNDANCE_ESTIMATOR_HPP

#include <map>
#include <utility>
#include <vector>

#include "path_estimator.hpp"

namespace rpvg_amd {

class PathAbundanceEstimator : public PathEstimator {

    public:

        PathAbundanceEstimator(const uint32_t max_em_its_in, const double max_rel_em_conv_in, const uint32_t num_gibbs_samples_in, const uint32_t gibbs_thin_its_in, const double prob_precision, std::shared_ptr<HipEngine> engine = HipEngine::processDefault());
        virtual ~PathAbundanceEstimator() {};

        bool usesRandomNumbers() const { return num_gibbs_samples > 0; }

        void estimateBatch(std::vector<PathClusterEstimates> * path_cluster_estimates, const DeviceClusterBatch & cluster_batch, std::vector<std::mt19937> * rngs);

    protected:

        const uint32_t max_em_its;
        const double max_rel_em_conv;

        const uint32_t num_gibbs_samples;
        const uint32_t gibbs_thin_its;

        // One EM problem: a cluster restricted to an ascending list of its paths.
        struct EMProblem {

            uint32_t cluster;
            std::vector<uint32_t> path_ids;
        };

        struct EMSolution {

            std::vector<double> abundances;
            double noise_count;
            double total_count;
            uint32_t iterations;
        };

        // EMAbundanceEstimator (src/path_abundance_estimator.cpp:47-114), together with
        // the matrix construction and normalisation in front of it, for a batch of problems.
        // read_collapse: the rows of every problem go through readCollapseProbabilityMatrix first (the callers at
        // src/path_abundance_estimator.cpp:266,668 do that, the one at :18-45 does not)
        void EMAbundanceEstimator(std::vector<EMSolution> * solutions, const DeviceClusterBatch & cluster_batch, const std::vector<EMProblem> & problems, const bool read_collapse) const;

        // gibbsReadCountSampler (src/path_abundance_estimator.cpp:116-212) for a batch of solved problems, on the
        // GPU (Philox generator keyed by `seeds`; statistical parity with the reference's mt19937 streams).
        // Problems with num_samples == 0 get an empty CountSamples.
        void gibbsReadCountSampler(std::vector<CountSamples> * count_samples, const DeviceClusterBatch & cluster_batch, const std::vector<EMProblem> & problems, const std::vector<EMSolution> & solutions, const std::vector<uint32_t> & num_samples, const std::vector<uint64_t> & seeds) const;

        static uint64_t drawSeed(std::mt19937 * mt_rng);
};

// `-i strains` (src/path_abundance_estimator.hpp:39-49): greedy weighted minimum path cover of the cluster's
// reads, then the EM on the covering paths only.
class MinimumPathAbundanceEstimator : public PathAbundanceEstimator {

    public:

        MinimumPathAbundanceEstimator(const uint32_t max_em_its, const double max_rel_em_conv, const uint32_t num_gibbs_samples, const uint32_t gibbs_thin_its, const double prob_precision, std::shared_ptr<HipEngine> engine = HipEngine::processDefault());
        ~MinimumPathAbundanceEstimator() {};

        void estimateBatch(std::vector<PathClusterEstimates> * path_cluster_estimates, const DeviceClusterBatch & cluster_batch, std::vector<std::mt19937> * rngs);

        // weightedMinimumPathCover (src/path_abundance_estimator.cpp:297-340) of the listed clusters, on the GPU.
        std::vector<std::vector<uint32_t> > weightedMinimumPathCover(const DeviceClusterBatch & cluster_batch, const std::vector<uint32_t> & clusters) const;
};

class NestedPathAbundanceEstimator : public PathAbundanceEstimator {

    public:

        NestedPathAbundanceEstimator(const uint32_t group_size_in, const double min_hap_prob_in, const bool infer_collapsed_in, const bool use_group_post_gibbs_in, const uint32_t max_em_its, const double max_rel_em_conv, const uint32_t num_gibbs_samples, const uint32_t gibbs_thin_its, const double prob_precision, std::shared_ptr<HipEngine> engine = HipEngine::processDefault());
        ~NestedPathAbundanceEstimator() {};

        bool usesRandomNumbers() const { return !infer_collapsed || use_group_post_gibbs || num_gibbs_samples > 0; }

        // (the one-call device path of estimateClusters: collapsed groups, diploid, branch and bound, no read-count samples)
        bool wantsSourceColumns() const { return infer_collapsed && !use_group_post_gibbs && group_size == 2 && num_gibbs_samples == 0; }

        bool sourceColumnsOf(GroupPosteriorProblem * columns, const std::vector<PathInfo> & paths) const;

        void estimateBatch(std::vector<PathClusterEstimates> * path_cluster_estimates, const DeviceClusterBatch & cluster_batch, std::vector<std::mt19937> * rngs);

    private:

        const uint32_t group_size;
        const double min_hap_prob;

        const bool infer_collapsed;
        const bool use_group_post_gibbs;

        // path subset (sorted, homozygous paths repeated) -> weight
        typedef std::map<std::vector<uint32_t>, double> PathSubsetWeights;

        std::vector<std::vector<uint32_t> > findPathGroups(const std::vector<PathInfo> & paths) const;
        void findPathSourceGroups(GroupPosteriorProblem * problem, const std::vector<PathInfo> & paths) const;

        void estimateClusters(std::vector<PathClusterEstimates> * path_cluster_estimates, const DeviceClusterBatch & cluster_batch, const std::vector<uint32_t> & clusters, std::vector<std::mt19937> * rngs, const std::function<void()> & first_device_stage) const;

        void pathGroupPosteriors(std::vector<GroupPosteriors> * group_posteriors, const DeviceClusterBatch & cluster_batch, const std::vector<GroupPosteriorProblem> & problems, std::vector<std::mt19937> * rngs) const;

        void sampleGroupPathIndices(std::vector<std::vector<uint32_t> > * path_subset_samples, const GroupPosteriors & group_posteriors, const std::vector<uint32_t> & group, std::mt19937 * mt_rng) const;
        void selectPathSubsetIndices(PathSubsetWeights * path_subset_samples, const GroupPosteriors & group_posteriors, const GroupPosteriorProblem & problem) const;

        // The weighted merge (src/path_abundance_estimator.cpp:702-749) of the subsets and EM solutions the device left
        // (PathEstimator::nestedSubsetAbundances): matrix i of the result belongs to clusters.at(i).
        void mergeSubsetSolutions(std::vector<PathClusterEstimates> * path_cluster_estimates, const DeviceClusterBatch & cluster_batch, const std::vector<uint32_t> & clusters, const rpvg_hip_subset_em_view & subsets, bool reset_first) const;

        // The estimates of a device call that merged on the device too (rpvg_hip_subset_em_view::set_*): copied into the
        // clusters' containers in place — a caller that hands the same containers in again pays no allocation.
        void unpackMergedSolutions(std::vector<PathClusterEstimates> * path_cluster_estimates, const DeviceClusterBatch & cluster_batch, const std::vector<uint32_t> & clusters, const rpvg_hip_subset_em_view & subsets) const;

        void inferPathSubsetAbundance(std::vector<PathClusterEstimates> * path_cluster_estimates, const DeviceClusterBatch & cluster_batch, const std::vector<uint32_t> & clusters, const std::vector<PathSubsetWeights> & path_subset_samples, std::vector<std::mt19937> * rngs, bool reset_first) const;
};

}

#endif
