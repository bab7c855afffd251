// GPU-backed posterior estimators with the constructor parameter lists of the
// reference's classes (src/path_posterior_estimator.hpp:18-42):
//   PathPosteriorEstimator       single-path posteriors (group size 1)
//   PathGroupPosteriorEstimator  `-i haplotypes`, ploidy -y
#ifndef RPVG_AMD_PATH_POSTERIOR_ESTIMATOR_HPP
#define RPVG_AMD_PATH_POSTERIOR_ESTIMATOR_HPP

#include <vector>

#include "path_estimator.hpp"

namespace rpvg_amd {

class PathPosteriorEstimator : public PathEstimator {

    public:

        PathPosteriorEstimator(const double prob_precision, std::shared_ptr<HipEngine> engine = HipEngine::processDefault());
        virtual ~PathPosteriorEstimator() {};

        void estimateBatch(std::vector<PathClusterEstimates> * path_cluster_estimates, const DeviceClusterBatch & cluster_batch, std::vector<std::mt19937> * rngs);

    protected:

        // One problem per non-empty cluster on the raw (un-normalised) probability
        // matrix: a column per path, weighted by PathInfo::source_count
        // (src/path_posterior_estimator.cpp:19-27, 45-53).
        std::vector<GroupPosteriorProblem> rawPathProblems(const std::vector<PathClusterEstimates> & path_cluster_estimates, const DeviceClusterBatch & cluster_batch, const std::vector<uint32_t> & clusters) const;

        // The clusters of the batch that have at least one row.
        static std::vector<uint32_t> clustersWithRows(const DeviceClusterBatch & cluster_batch);
};

class PathGroupPosteriorEstimator : public PathPosteriorEstimator {

    public:

        PathGroupPosteriorEstimator(const uint32_t group_size_in, const bool use_group_post_gibbs_in, const double prob_precision, std::shared_ptr<HipEngine> engine = HipEngine::processDefault());
        ~PathGroupPosteriorEstimator() {};

        bool usesRandomNumbers() const { return use_group_post_gibbs; }

        void estimateBatch(std::vector<PathClusterEstimates> * path_cluster_estimates, const DeviceClusterBatch & cluster_batch, std::vector<std::mt19937> * rngs);

    private:

        const uint32_t group_size;
        const bool use_group_post_gibbs;

        void estimateClusters(std::vector<PathClusterEstimates> * path_cluster_estimates, const DeviceClusterBatch & cluster_batch, const std::vector<uint32_t> & clusters, std::vector<std::mt19937> * rngs) const;
};

}

#endif
