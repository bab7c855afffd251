#include "path_posterior_estimator.hpp"

#include <cassert>

#include "trace.hpp"

namespace rpvg_amd {

// src/path_posterior_estimator.cpp:5
static const double min_rel_likelihood = 1e-8;

PathPosteriorEstimator::PathPosteriorEstimator(const double prob_precision, std::shared_ptr<HipEngine> engine) : PathEstimator(prob_precision, engine) {}

std::vector<uint32_t> PathPosteriorEstimator::clustersWithRows(const DeviceClusterBatch & cluster_batch) {

    std::vector<uint32_t> clusters;

    for (uint32_t i = 0; i < cluster_batch.numClusters(); ++i) {

        if (cluster_batch.numRows(i) > 0) {

            clusters.emplace_back(i);
        }
    }

    return clusters;
}

std::vector<GroupPosteriorProblem> PathPosteriorEstimator::rawPathProblems(const std::vector<PathClusterEstimates> & path_cluster_estimates, const DeviceClusterBatch & cluster_batch, const std::vector<uint32_t> & clusters) const {

    ScopedPhase phase("posteriors: problems");

    std::vector<GroupPosteriorProblem> problems;

    for (auto & i: clusters) {

        assert(path_cluster_estimates.at(i).paths.size() == cluster_batch.numPaths(i));
        assert(cluster_batch.numRows(i) > 0);

        problems.emplace_back(GroupPosteriorProblem());
        problems.back().cluster = i;
    }

    #pragma omp parallel for schedule(dynamic, 16) num_threads(shortLoopThreads())
    for (size_t p = 0; p < problems.size(); ++p) {

        auto & problem = problems[p];
        const auto & paths = path_cluster_estimates.at(problem.cluster).paths;

        problem.singlePathColumns(paths.size(), [&](const uint32_t j) { return paths[j].source_count; });
    }

    return problems;
}

// src/path_posterior_estimator.cpp:9-31 over a batch of clusters.
void PathPosteriorEstimator::estimateBatch(std::vector<PathClusterEstimates> * path_cluster_estimates, const DeviceClusterBatch & cluster_batch, std::vector<std::mt19937> * rngs) {

    assert(path_cluster_estimates->size() == cluster_batch.numClusters());

    for (auto & estimates: *path_cluster_estimates) {

        estimates.resetEstimates(estimates.paths.size(), 1);
    }

    const auto problems = rawPathProblems(*path_cluster_estimates, cluster_batch, clustersWithRows(cluster_batch));

    std::vector<GroupPosteriors> group_posteriors;
    calculatePathGroupPosteriorsFull(&group_posteriors, cluster_batch, problems, 1, false);

    for (size_t i = 0; i < problems.size(); ++i) {

        auto & estimates = path_cluster_estimates->at(problems.at(i).cluster);

        // calculatePathGroupPosteriorsFull re-creates the sets and zeroes the rest (src/path_estimator.cpp:343)
        estimates.resetEstimates(estimates.paths.size(), 1);
        assert(estimates.path_group_sets.size() == group_posteriors.at(i).size());

        estimates.posteriors = std::move(group_posteriors.at(i).posteriors);
    }
}

PathGroupPosteriorEstimator::PathGroupPosteriorEstimator(const uint32_t group_size_in, const bool use_group_post_gibbs_in, const double prob_precision, std::shared_ptr<HipEngine> engine) : PathPosteriorEstimator(prob_precision, engine), group_size(group_size_in), use_group_post_gibbs(use_group_post_gibbs_in) {}

// src/path_posterior_estimator.cpp:35-71 over a batch of clusters.
void PathGroupPosteriorEstimator::estimateBatch(std::vector<PathClusterEstimates> * path_cluster_estimates, const DeviceClusterBatch & cluster_batch, std::vector<std::mt19937> * rngs) {

    if (use_group_post_gibbs && !rngs) {

        throw EngineError("Gibbs haplotype posteriors draw random numbers: a generator per cluster is required");
    }

    assert(path_cluster_estimates->size() == cluster_batch.numClusters());

    {
        ScopedPhase phase("posteriors: reset estimates");

        #pragma omp parallel for schedule(dynamic, 16) num_threads(shortLoopThreads())
        for (size_t i = 0; i < path_cluster_estimates->size(); ++i) {

            (*path_cluster_estimates)[i].resetEstimates(0, 0);
        }
    }

    runInLanes(clustersWithRows(cluster_batch), [&](const std::vector<uint32_t> & lane_clusters, const std::function<void()> & first_device_stage) {

        first_device_stage();  // the host phase before the first device call is negligible here
        estimateClusters(path_cluster_estimates, cluster_batch, lane_clusters, rngs);
    });
}

// src/path_posterior_estimator.cpp:35-71 for a subset of the batch's clusters (all with at least one row).
void PathGroupPosteriorEstimator::estimateClusters(std::vector<PathClusterEstimates> * path_cluster_estimates, const DeviceClusterBatch & cluster_batch, const std::vector<uint32_t> & clusters, std::vector<std::mt19937> * rngs) const {

    const auto problems = rawPathProblems(*path_cluster_estimates, cluster_batch, clusters);
    std::vector<GroupPosteriors> group_posteriors;

    if (use_group_post_gibbs) {

        std::vector<std::mt19937 *> problem_rngs;

        for (auto & problem: problems) {

            problem_rngs.emplace_back(&rngs->at(problem.cluster));
        }

        estimatePathGroupPosteriorsGibbs(&group_posteriors, cluster_batch, problems, group_size, false, problem_rngs);

    } else if (group_size == 2) {

        calculatePathGroupPosteriorsBounded(&group_posteriors, cluster_batch, problems, group_size, min_rel_likelihood, false);

    } else {

        calculatePathGroupPosteriorsFull(&group_posteriors, cluster_batch, problems, group_size, false);
    }

    ScopedPhase pack_phase("posteriors: fill estimates");

    #pragma omp parallel for schedule(dynamic, 16) num_threads(shortLoopThreads())
    for (size_t i = 0; i < problems.size(); ++i) {

        auto & estimates = path_cluster_estimates->at(problems.at(i).cluster);

        const auto & result = group_posteriors.at(i);

        estimates.path_group_sets.reserve(result.size());

        for (size_t j = 0; j < result.size(); ++j) {

            estimates.path_group_sets.emplace_back(result.set(j), result.set(j) + result.group_size);
        }

        estimates.posteriors = std::move(group_posteriors.at(i).posteriors);

        if (group_size != 2 && !use_group_post_gibbs) {

            // the Full routine leaves zero-filled abundances behind (resetEstimates(n, g), src/path_estimator.cpp:343)
            estimates.abundances.assign(estimates.path_group_sets.size() * group_size, 0);
        }
    }
}

}
