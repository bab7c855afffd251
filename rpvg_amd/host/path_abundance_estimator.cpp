#include "path_abundance_estimator.hpp"

#include <algorithm>
#include <array>
#include <cassert>
#include <cmath>
#include <limits>
#include <memory>
#include <numeric>

#include "numeric_utils.hpp"
#include "pipeline_lanes.hpp"
#include "trace.hpp"

namespace rpvg_amd {

PathAbundanceEstimator::PathAbundanceEstimator(const uint32_t max_em_its_in, const double max_rel_em_conv_in, const uint32_t num_gibbs_samples_in, const uint32_t gibbs_thin_its_in, const double prob_precision, std::shared_ptr<HipEngine> engine) : PathEstimator(prob_precision, engine), max_em_its(max_em_its_in), max_rel_em_conv(max_rel_em_conv_in), num_gibbs_samples(num_gibbs_samples_in), gibbs_thin_its(gibbs_thin_its_in) {}

// src/path_abundance_estimator.cpp:13
static const double abundance_gibbs_gamma = 1;

uint64_t PathAbundanceEstimator::drawSeed(std::mt19937 * mt_rng) {

    const uint64_t high = (*mt_rng)();
    const uint64_t low = (*mt_rng)();

    return (high << 32) | low;
}

void PathAbundanceEstimator::gibbsReadCountSampler(std::vector<CountSamples> * count_samples, const DeviceClusterBatch & cluster_batch, const std::vector<EMProblem> & problems, const std::vector<EMSolution> & solutions, const std::vector<uint32_t> & num_samples, const std::vector<uint64_t> & seeds) const {

    ScopedPhase phase("Gibbs read counts: rpvg_hip_gibbs_read_counts");

    assert(problems.size() == solutions.size());
    assert(problems.size() == num_samples.size());
    assert(problems.size() == seeds.size());

    count_samples->assign(problems.size(), CountSamples());

    if (problems.empty()) {

        return;
    }

    std::vector<uint32_t> clusters;
    std::vector<uint64_t> col_off(1, 0);
    std::vector<uint32_t> col_path;
    std::vector<double> init_abundances;
    std::vector<double> init_noise_count;

    std::vector<uint64_t> sample_off(1, 0);
    std::vector<uint64_t> abundance_sample_off(1, 0);

    for (size_t i = 0; i < problems.size(); ++i) {

        clusters.emplace_back(problems.at(i).cluster);
        col_path.insert(col_path.end(), problems.at(i).path_ids.begin(), problems.at(i).path_ids.end());
        col_off.emplace_back(col_path.size());

        init_abundances.insert(init_abundances.end(), solutions.at(i).abundances.begin(), solutions.at(i).abundances.end());
        init_noise_count.emplace_back(solutions.at(i).noise_count);

        sample_off.emplace_back(sample_off.back() + num_samples.at(i));
        abundance_sample_off.emplace_back(abundance_sample_off.back() + static_cast<uint64_t>(num_samples.at(i)) * problems.at(i).path_ids.size());
    }

    std::vector<double> noise_samples(sample_off.back());
    std::vector<double> abundance_samples(abundance_sample_off.back());

    rpvg_hip_em_problems em_problems;
    em_problems.num_problems = problems.size();
    em_problems.cluster = clusters.data();
    em_problems.col_off = col_off.data();
    em_problems.col_path = col_path.data();
    em_problems.collapse_precision = 0;

    HipEngine::check(rpvg_hip_gibbs_read_counts(engine->ctx(), cluster_batch.handle(), &em_problems, init_abundances.data(), init_noise_count.data(), num_samples.data(), seeds.data(), gibbs_thin_its, abundance_gibbs_gamma, noise_samples.data(), abundance_samples.data()), "rpvg_hip_gibbs_read_counts");

    for (size_t i = 0; i < problems.size(); ++i) {

        auto & samples = count_samples->at(i);

        samples.path_ids = problems.at(i).path_ids;
        samples.noise_samples.assign(noise_samples.begin() + sample_off.at(i), noise_samples.begin() + sample_off.at(i + 1));
        samples.abundance_samples.assign(abundance_samples.begin() + abundance_sample_off.at(i), abundance_samples.begin() + abundance_sample_off.at(i + 1));
    }
}

void PathAbundanceEstimator::EMAbundanceEstimator(std::vector<EMSolution> * solutions, const DeviceClusterBatch & cluster_batch, const std::vector<EMProblem> & problems, const bool read_collapse) const {

    ScopedPhase phase("EM: flatten + rpvg_hip_em_solve + unpack");

    solutions->assign(problems.size(), EMSolution());

    if (problems.empty()) {

        return;
    }

    std::unique_ptr<ScopedPhase> part(new ScopedPhase("EM: flatten"));

    std::vector<uint32_t> clusters;
    std::vector<uint64_t> col_off(1, 0);
    std::vector<uint32_t> col_path;

    clusters.reserve(problems.size());
    col_off.reserve(problems.size() + 1);

    for (auto & problem: problems) {

        assert(!problem.path_ids.empty());

        clusters.emplace_back(problem.cluster);
        col_off.emplace_back(col_off.back() + problem.path_ids.size());
    }

    col_path.resize(col_off.back());

    #pragma omp parallel for schedule(static) num_threads(hostThreads())
    for (size_t i = 0; i < problems.size(); ++i) {

        std::copy(problems.at(i).path_ids.begin(), problems.at(i).path_ids.end(), col_path.begin() + col_off.at(i));
    }

    std::vector<double> abundances(col_path.size());
    std::vector<double> noise_counts(problems.size());
    std::vector<double> total_counts(problems.size());
    std::vector<uint32_t> iterations(problems.size());

    rpvg_hip_em_problems em_problems;
    em_problems.num_problems = problems.size();
    em_problems.cluster = clusters.data();
    em_problems.col_off = col_off.data();
    em_problems.col_path = col_path.data();
    em_problems.collapse_precision = read_collapse ? prob_precision : 0;

    rpvg_hip_em_results em_results;
    em_results.abundances = abundances.data();
    em_results.noise_count = noise_counts.data();
    em_results.total_count = total_counts.data();
    em_results.iterations = iterations.data();

    part.reset();
    HipEngine::check(rpvg_hip_em_solve(engine->ctx(), cluster_batch.handle(), max_em_its, max_rel_em_conv, &em_problems, &em_results), "rpvg_hip_em_solve");
    part.reset(new ScopedPhase("EM: unpack"));

    #pragma omp parallel for schedule(static) num_threads(hostThreads())
    for (size_t i = 0; i < problems.size(); ++i) {

        auto & solution = solutions->at(i);

        solution.abundances.assign(abundances.begin() + col_off.at(i), abundances.begin() + col_off.at(i + 1));
        solution.noise_count = noise_counts.at(i);
        solution.total_count = total_counts.at(i);
        solution.iterations = iterations.at(i);
    }
}

// src/path_abundance_estimator.cpp:18-45 over a batch of clusters.
void PathAbundanceEstimator::estimateBatch(std::vector<PathClusterEstimates> * path_cluster_estimates, const DeviceClusterBatch & cluster_batch, std::vector<std::mt19937> * rngs) {

    if (num_gibbs_samples > 0 && !rngs) {

        throw EngineError("read-count Gibbs sampling draws random numbers: a generator per cluster is required");
    }

    assert(path_cluster_estimates->size() == cluster_batch.numClusters());

    std::vector<EMProblem> problems;
    problems.reserve(cluster_batch.numClusters());

    for (uint32_t i = 0; i < cluster_batch.numClusters(); ++i) {

        auto & estimates = path_cluster_estimates->at(i);

        assert(estimates.paths.size() == cluster_batch.numPaths(i));
        estimates.resetEstimates(estimates.paths.size(), 1);

        if (cluster_batch.numRows(i) > 0) {

            problems.emplace_back(EMProblem());
            problems.back().cluster = i;

            problems.back().path_ids.resize(estimates.paths.size());
            std::iota(problems.back().path_ids.begin(), problems.back().path_ids.end(), 0);
        }
    }

    std::vector<EMSolution> solutions;
    EMAbundanceEstimator(&solutions, cluster_batch, problems, false);  // (src/path_abundance_estimator.cpp:18-45: no row collapse)

    if (num_gibbs_samples > 0) {

        // src/path_abundance_estimator.cpp:34-43
        std::vector<uint32_t> num_samples(problems.size(), num_gibbs_samples);
        std::vector<uint64_t> seeds;

        for (auto & problem: problems) {

            seeds.emplace_back(drawSeed(&rngs->at(problem.cluster)));
        }

        std::vector<CountSamples> count_samples;
        gibbsReadCountSampler(&count_samples, cluster_batch, problems, solutions, num_samples, seeds);

        for (size_t i = 0; i < problems.size(); ++i) {

            path_cluster_estimates->at(problems.at(i).cluster).gibbs_read_count_samples.emplace_back(std::move(count_samples.at(i)));
        }
    }

    for (size_t i = 0; i < problems.size(); ++i) {

        auto & estimates = path_cluster_estimates->at(problems.at(i).cluster);

        estimates.abundances = std::move(solutions.at(i).abundances);
        estimates.noise_count = solutions.at(i).noise_count;
        estimates.total_count = solutions.at(i).total_count;

        estimates.em_iterations.emplace_back(solutions.at(i).iterations);
        estimates.em_problem_paths.emplace_back(std::move(problems.at(i).path_ids));
    }
}

MinimumPathAbundanceEstimator::MinimumPathAbundanceEstimator(const uint32_t max_em_its, const double max_rel_em_conv, const uint32_t num_gibbs_samples, const uint32_t gibbs_thin_its, const double prob_precision, std::shared_ptr<HipEngine> engine) : PathAbundanceEstimator(max_em_its, max_rel_em_conv, num_gibbs_samples, gibbs_thin_its, prob_precision, engine) {}

std::vector<std::vector<uint32_t> > MinimumPathAbundanceEstimator::weightedMinimumPathCover(const DeviceClusterBatch & cluster_batch, const std::vector<uint32_t> & clusters) const {

    std::vector<std::vector<uint32_t> > covers(clusters.size());

    if (clusters.empty()) {

        return covers;
    }

    std::vector<uint64_t> cover_off(1, 0);

    for (auto & cluster: clusters) {

        cover_off.emplace_back(cover_off.back() + cluster_batch.numPaths(cluster));
    }

    std::vector<uint32_t> cover(cover_off.back());
    std::vector<uint32_t> cover_size(clusters.size());

    HipEngine::check(rpvg_hip_min_path_cover(engine->ctx(), cluster_batch.handle(), clusters.size(), clusters.data(), cover_off.data(), cover.data(), cover_size.data()), "rpvg_hip_min_path_cover");

    for (size_t i = 0; i < clusters.size(); ++i) {

        covers.at(i).assign(cover.begin() + cover_off.at(i), cover.begin() + cover_off.at(i) + cover_size.at(i));
    }

    return covers;
}

// src/path_abundance_estimator.cpp:217-295 over a batch of clusters.
void MinimumPathAbundanceEstimator::estimateBatch(std::vector<PathClusterEstimates> * path_cluster_estimates, const DeviceClusterBatch & cluster_batch, std::vector<std::mt19937> * rngs) {

    if (num_gibbs_samples > 0 && !rngs) {

        throw EngineError("read-count Gibbs sampling draws random numbers: a generator per cluster is required");
    }

    assert(path_cluster_estimates->size() == cluster_batch.numClusters());

    std::vector<uint32_t> clusters;

    for (uint32_t i = 0; i < cluster_batch.numClusters(); ++i) {

        auto & estimates = path_cluster_estimates->at(i);

        assert(estimates.paths.size() == cluster_batch.numPaths(i));
        estimates.resetEstimates(estimates.paths.size(), 1);

        if (cluster_batch.numRows(i) > 0) {

            clusters.emplace_back(i);
        }
    }

    const auto covers = weightedMinimumPathCover(cluster_batch, clusters);

    // EM on the covering paths of every cluster that has any (:262-293)
    std::vector<EMProblem> problems;

    for (size_t i = 0; i < clusters.size(); ++i) {

        if (!covers.at(i).empty()) {

            problems.emplace_back(EMProblem());
            problems.back().cluster = clusters.at(i);
            problems.back().path_ids = covers.at(i);
        }
    }

    std::vector<EMSolution> solutions;
    EMAbundanceEstimator(&solutions, cluster_batch, problems, true);  // (:266)

    std::vector<CountSamples> count_samples;

    if (num_gibbs_samples > 0) {

        std::vector<uint32_t> num_samples(problems.size(), num_gibbs_samples);
        std::vector<uint64_t> seeds;

        for (auto & problem: problems) {

            seeds.emplace_back(drawSeed(&rngs->at(problem.cluster)));
        }

        gibbsReadCountSampler(&count_samples, cluster_batch, problems, solutions, num_samples, seeds);
    }

    for (size_t i = 0; i < problems.size(); ++i) {

        auto & estimates = path_cluster_estimates->at(problems.at(i).cluster);

        for (size_t j = 0; j < problems.at(i).path_ids.size(); ++j) {

            estimates.abundances.at(problems.at(i).path_ids.at(j)) += solutions.at(i).abundances.at(j);
        }

        estimates.noise_count = solutions.at(i).noise_count;
        estimates.total_count = solutions.at(i).total_count;

        estimates.em_iterations.emplace_back(solutions.at(i).iterations);
        estimates.em_problem_paths.emplace_back(problems.at(i).path_ids);

        if (!count_samples.empty()) {

            estimates.gibbs_read_count_samples.emplace_back(std::move(count_samples.at(i)));
        }
    }
}

NestedPathAbundanceEstimator::NestedPathAbundanceEstimator(const uint32_t group_size_in, const double min_hap_prob_in, const bool infer_collapsed_in, const bool use_group_post_gibbs_in, const uint32_t max_em_its, const double max_rel_em_conv, const uint32_t num_gibbs_samples, const uint32_t gibbs_thin_its, const double prob_precision, std::shared_ptr<HipEngine> engine) : PathAbundanceEstimator(max_em_its, max_rel_em_conv, num_gibbs_samples, gibbs_thin_its, prob_precision, engine), group_size(group_size_in), min_hap_prob(min_hap_prob_in), infer_collapsed(infer_collapsed_in), use_group_post_gibbs(use_group_post_gibbs_in) {}

// src/path_abundance_estimator.cpp:344-471 over a batch of clusters: posteriors of
// all clusters first (one set of GPU calls), then the EM solves of every retained
// path subset of every cluster (one GPU call), then the weighted merge.
void NestedPathAbundanceEstimator::estimateBatch(std::vector<PathClusterEstimates> * path_cluster_estimates, const DeviceClusterBatch & cluster_batch, std::vector<std::mt19937> * rngs) {

    if (num_gibbs_samples > 0 && !rngs) {

        throw EngineError("read-count Gibbs sampling draws random numbers: a generator per cluster is required");
    }

    if (use_group_post_gibbs && !rngs) {

        throw EngineError("Gibbs haplotype posteriors draw random numbers: a generator per cluster is required");
    }

    assert(path_cluster_estimates->size() == cluster_batch.numClusters());

    ScopedPhase whole_phase("nested: estimateBatch incl. teardown");
    std::unique_ptr<ScopedPhase> list_phase(new ScopedPhase("nested: cluster list"));

    std::vector<uint32_t> clusters;

    // clusters without reads keep empty estimates (src/path_abundance_estimator.cpp:358-360); the others are
    // reset by the lane that owns them, under the kernels of the lane before it
    for (uint32_t i = 0; i < cluster_batch.numClusters(); ++i) {

        assert(path_cluster_estimates->at(i).paths.size() == cluster_batch.numPaths(i));

        if (cluster_batch.numRows(i) > 0) {

            clusters.emplace_back(i);

        } else {

            path_cluster_estimates->at(i).resetEstimates(0, 0);
        }
    }

    list_phase.reset();

    runInLanes(clusters, [&](const std::vector<uint32_t> & lane_clusters, const std::function<void()> & first_device_stage) {

        estimateClusters(path_cluster_estimates, cluster_batch, lane_clusters, rngs, first_device_stage);
    });
}

// (Loops over a lane's clusters with little work per cluster use schedule(static, 1): the clusters come ordered by size, and
// contiguous blocks would give the first thread all the large ones.)
// Chunk of the dynamic schedules over a lane's clusters.  The clusters come ordered by size: a chunk of 16 put the 16
// largest on one thread (weighted merge of the last lane, the end of the batch: 0.6 ms, 0.4 ms with chunks of one).
// A/B knob RPVG_AMD_CLUSTER_CHUNK.
// Clusters per turn of a team's thread in the two loops a batch's host time is made of (findPathSourceGroups, the weighted
// merge): dealt round-robin (static), not drawn from a counter (dynamic) — the same time per batch (9.54 against 9.74 ms,
// interleaved medians) for a quarter less CPU time (80 against 110 ms per batch: no shared counter, and a thread meets the
// clusters it had the batch before when a caller hands the same containers in again).
static int clusterChunk() {

    static const int chunk = []() {

        const char * env = RPVG_AMD_EXPERIMENT_ENV("RPVG_AMD_CLUSTER_CHUNK");
        return env ? std::max(1, std::atoi(env)) : 1;
    }();

    return chunk;
}

// Tens of thousands of small containers are freed by a team instead of one by one — now by the first lane (it never
// waits, and finishes first), at the start of its next batch by any other lane (RetiredContainers, pipeline_lanes.hpp).
// between_device_stages: the call sits between two GPU stages of the lane (search and EM) — the first lane keeps the
// containers too then, until its work is done (PathEstimator::runInLanes drops them while it waits for the other lanes).
static void dropNowOrLater(std::function<void(int)> drop, const bool between_device_stages = false) {

    static const bool never_later = RPVG_AMD_EXPERIMENT_ENV("RPVG_AMD_NO_DEFERRED_TEARDOWN") != nullptr;

    if (never_later || !LaneScope::active() || (HipEngine::currentLane() == 0 && !between_device_stages)) {

        drop(hostThreads());

    } else {

        // (the lane's usual team size: libgomp lets the threads of a team go when the next team is smaller, and creating
        // them again cost the lane's next parallel region 1.5 ms)
        RetiredContainers::ofThisThread().keep([drop]() { drop(hostThreads()); });
    }
}

// The estimator on a subset of the batch's clusters (all with at least one row).
void NestedPathAbundanceEstimator::estimateClusters(std::vector<PathClusterEstimates> * path_cluster_estimates, const DeviceClusterBatch & cluster_batch, const std::vector<uint32_t> & clusters, std::vector<std::mt19937> * rngs, const std::function<void()> & first_device_stage) const {

    // The defaults of `-i haplotype-transcripts` — collapsed groups, diploid, branch and bound, no read-count samples — on a
    // batch that carries its haplotype columns (findPathSourceGroups ran on the device with the upload): matrices, search,
    // subsets, EM and the weighted merge are ONE device call; the host lists the lane's clusters and copies the estimates out.
    if (infer_collapsed && !use_group_post_gibbs && group_size == 2 && num_gibbs_samples == 0 && cluster_batch.hasSourceColumns() && !std::getenv("RPVG_AMD_HOST_SOURCE_GROUPS")) {

        first_device_stage();

        SubsetEmResult device_result;

        if (nestedSubsetAbundances(&device_result, cluster_batch, clusters, min_hap_prob, min_hap_prob, max_em_its, max_rel_em_conv) && device_result.view.set_count) {

            unpackMergedSolutions(path_cluster_estimates, cluster_batch, clusters, device_result.view);
            return;
        }
    }

    // The previous estimates of a cluster are dropped before the new ones are written.  The first lane's prologue is
    // the batch's (the GPU waits for it): that lane drops them in its merge loop, which nobody waits for.
    const bool reset_in_merge = (HipEngine::currentLane() == 0);

    if (!reset_in_merge) {

        ScopedPhase reset_phase("nested: resetEstimates");

        #pragma omp parallel for schedule(static, 1) num_threads(hostThreads())
        for (size_t i = 0; i < clusters.size(); ++i) {

            path_cluster_estimates->at(clusters[i]).resetEstimates(0, 0);
        }
    }

    std::unique_ptr<ScopedPhase> containers_phase(new ScopedPhase("nested: containers"));
    std::vector<PathSubsetWeights> path_subset_samples(clusters.size());
    containers_phase.reset();

    if (infer_collapsed) {

        // inferAbundancesCollapsedGroups (:428-471)
        containers_phase.reset(new ScopedPhase("nested: problem containers"));
        std::vector<GroupPosteriorProblem> problems(clusters.size());
        containers_phase.reset();

        std::unique_ptr<ScopedPhase> groups_phase(new ScopedPhase("nested: findPathSourceGroups"));

        #pragma omp parallel for schedule(static, clusterChunk()) num_threads(hostThreads())
        for (size_t i = 0; i < clusters.size(); ++i) {

            problems.at(i).cluster = clusters.at(i);
            findPathSourceGroups(&problems.at(i), path_cluster_estimates->at(clusters.at(i)).paths);
        }

        for (auto & problem: problems) {

            if (problem.numColumns() == 0) {

                throw EngineError("haplotype-transcripts inference needs source (haplotype) ids on the paths of every cluster");
            }
        }

        groups_phase.reset();
        first_device_stage();

        // The defaults of `-i haplotype-transcripts` — diploid, branch and bound, no read-count samples — run from the
        // matrices to the EM solutions in one device call; the host only merges.
        if (!use_group_post_gibbs && group_size == 2 && num_gibbs_samples == 0) {

            SubsetEmResult device_result;

            if (nestedSubsetAbundances(&device_result, cluster_batch, problems, min_hap_prob, min_hap_prob, max_em_its, max_rel_em_conv)) {

                mergeSubsetSolutions(path_cluster_estimates, cluster_batch, clusters, device_result.view, reset_in_merge);

                ScopedPhase teardown_phase("nested: teardown posterior containers");
                auto old_problems = std::make_shared<std::vector<GroupPosteriorProblem> >(std::move(problems));

                dropNowOrLater([old_problems](const int threads) {

                    #pragma omp parallel for schedule(static) num_threads(threads)
                    for (size_t i = 0; i < old_problems->size(); ++i) {

                        std::vector<uint32_t>().swap(old_problems->at(i).column_path);
                        std::vector<uint32_t>().swap(old_problems->at(i).column_path_off);
                        std::vector<uint32_t>().swap(old_problems->at(i).column_counts);
                    }
                });

                return;
            }
        }

        std::vector<GroupPosteriors> group_posteriors;
        pathGroupPosteriors(&group_posteriors, cluster_batch, problems, rngs);

        std::unique_ptr<ScopedPhase> select_phase(new ScopedPhase("nested: selectPathSubsetIndices"));

        #pragma omp parallel for schedule(dynamic, clusterChunk()) num_threads(hostThreads())
        for (size_t i = 0; i < clusters.size(); ++i) {

            selectPathSubsetIndices(&path_subset_samples.at(i), group_posteriors.at(i), problems.at(i));
        }

        select_phase.reset();

        ScopedPhase teardown_phase("nested: teardown posterior containers");

        auto old_problems = std::make_shared<std::vector<GroupPosteriorProblem> >(std::move(problems));
        auto old_posteriors = std::make_shared<std::vector<GroupPosteriors> >(std::move(group_posteriors));

        dropNowOrLater([old_problems, old_posteriors](const int threads) {

            #pragma omp parallel for schedule(static) num_threads(threads)
            for (size_t i = 0; i < old_problems->size(); ++i) {

                std::vector<uint32_t>().swap(old_problems->at(i).column_path);
                std::vector<uint32_t>().swap(old_problems->at(i).column_path_off);
                std::vector<uint32_t>().swap(old_problems->at(i).column_counts);
                std::vector<uint32_t>().swap(old_posteriors->at(i).members);
                std::vector<double>().swap(old_posteriors->at(i).posteriors);
            }
        }, true);

    } else {

        // inferAbundancesIndependentGroups (:356-426)
        if (!rngs) {

            throw EngineError("independent haplotype inference draws random numbers: a generator per cluster is required");
        }

        std::vector<GroupPosteriorProblem> problems;
        std::vector<std::vector<std::vector<uint32_t> > > cluster_path_groups(clusters.size());
        std::vector<size_t> first_problem(clusters.size() + 1, 0);

        for (size_t i = 0; i < clusters.size(); ++i) {

            const auto & paths = path_cluster_estimates->at(clusters.at(i)).paths;
            cluster_path_groups.at(i) = findPathGroups(paths);

            for (auto & group: cluster_path_groups.at(i)) {

                problems.emplace_back(GroupPosteriorProblem());
                problems.back().cluster = clusters.at(i);

                for (auto & path: group) {

                    problems.back().addColumn(&path, &path + 1, paths.at(path).source_count);
                }
            }

            first_problem.at(i + 1) = problems.size();
        }

        std::vector<std::vector<std::vector<uint32_t> > > cluster_samples(clusters.size());

        for (auto & samples: cluster_samples) {

            samples.resize(std::floor(1 / min_hap_prob));
        }

        if (use_group_post_gibbs) {

            // The reference runs a transcript's chains and draws that transcript's subsets from the same generator before
            // the next transcript's chains start (src/path_abundance_estimator.cpp:371-412): where the next chains start
            // in the generator's stream depends on how many sets the chains before them found (a discrete distribution
            // over one set draws nothing).  Round j of the batch is transcript j of every cluster that has one — one
            // problem per generator and device call, as with collapsed groups — and its subsets are drawn before round
            // j + 1 is queued: draw for draw the reference's order, at the price of one device call per round — as many rounds as the
            // batch's largest cluster has transcripts, the late rounds with a problem or two each: a batch with one cluster of thousands
            // of transcripts serialises thousands of small calls (no bound; a caller with such clusters batches them apart).
            size_t max_groups = 0;

            for (auto & path_groups: cluster_path_groups) {

                max_groups = std::max(max_groups, path_groups.size());
            }

            for (size_t j = 0; j < max_groups; ++j) {

                std::vector<GroupPosteriorProblem> round_problems;
                std::vector<size_t> round_clusters;

                for (size_t i = 0; i < clusters.size(); ++i) {

                    if (j < cluster_path_groups.at(i).size()) {

                        round_problems.emplace_back(std::move(problems.at(first_problem.at(i) + j)));
                        round_clusters.emplace_back(i);
                    }
                }

                std::vector<GroupPosteriors> round_posteriors;
                pathGroupPosteriors(&round_posteriors, cluster_batch, round_problems, rngs);

                #pragma omp parallel for schedule(dynamic, clusterChunk()) num_threads(hostThreads())
                for (size_t k = 0; k < round_clusters.size(); ++k) {

                    const size_t i = round_clusters.at(k);
                    sampleGroupPathIndices(&cluster_samples.at(i), round_posteriors.at(k), cluster_path_groups.at(i).at(j), &rngs->at(clusters.at(i)));
                }
            }

        } else {

            // exact posteriors draw nothing: all transcripts of the batch in one device call, the subsets after it
            std::vector<GroupPosteriors> group_posteriors;
            pathGroupPosteriors(&group_posteriors, cluster_batch, problems, rngs);

            #pragma omp parallel for schedule(dynamic, clusterChunk()) num_threads(hostThreads())
            for (size_t i = 0; i < clusters.size(); ++i) {

                for (size_t j = 0; j < cluster_path_groups.at(i).size(); ++j) {

                    sampleGroupPathIndices(&cluster_samples.at(i), group_posteriors.at(first_problem.at(i) + j), cluster_path_groups.at(i).at(j), &rngs->at(clusters.at(i)));
                }
            }
        }

        for (size_t i = 0; i < clusters.size(); ++i) {

            auto & samples = cluster_samples.at(i);

            for (auto & sample: samples) {

                std::sort(sample.begin(), sample.end());
                path_subset_samples.at(i)[sample] += 1 / static_cast<double>(samples.size());
            }
        }
    }

    inferPathSubsetAbundance(path_cluster_estimates, cluster_batch, clusters, path_subset_samples, rngs, reset_in_merge);

    ScopedPhase teardown_phase("nested: teardown subset weights");

    auto old_samples = std::make_shared<std::vector<PathSubsetWeights> >(std::move(path_subset_samples));

    dropNowOrLater([old_samples](const int threads) {

        #pragma omp parallel for schedule(static) num_threads(threads)
        for (size_t i = 0; i < old_samples->size(); ++i) {

            PathSubsetWeights().swap(old_samples->at(i));
        }
    });
}

void NestedPathAbundanceEstimator::pathGroupPosteriors(std::vector<GroupPosteriors> * group_posteriors, const DeviceClusterBatch & cluster_batch, const std::vector<GroupPosteriorProblem> & problems, std::vector<std::mt19937> * rngs) const {

    if (use_group_post_gibbs) {

        // One problem per cluster and call (collapsed groups; independent groups call this once per transcript
        // round): every problem consumes its cluster's generator exactly as the reference does.
        std::vector<std::mt19937 *> problem_rngs;

        for (auto & problem: problems) {

            problem_rngs.emplace_back(&rngs->at(problem.cluster));
        }

        estimatePathGroupPosteriorsGibbs(group_posteriors, cluster_batch, problems, group_size, true, problem_rngs);

    } else if (group_size == 2) {

        calculatePathGroupPosteriorsBounded(group_posteriors, cluster_batch, problems, group_size, min_hap_prob, true);

    } else {

        calculatePathGroupPosteriorsFull(group_posteriors, cluster_batch, problems, group_size, true);
    }
}

// The paths of every transcript (PathInfo::group_id), transcripts in the order their first path appears, a transcript's paths
// ascending (what src/path_abundance_estimator.cpp:473-491 returns).  Two flat passes instead of a map of vectors grown path by
// path: the transcripts are numbered in a sorted id list, counted, and every list is written once at its final size.
std::vector<std::vector<uint32_t> > NestedPathAbundanceEstimator::findPathGroups(const std::vector<PathInfo> & paths) const {

    const uint32_t num_paths = paths.size();

    // distinct transcript ids, ascending; `order_of[k]`: position of id k in first-appearance order
    std::vector<uint32_t> ids(num_paths);

    for (uint32_t p = 0; p < num_paths; ++p) {

        ids[p] = paths[p].group_id;
    }

    std::sort(ids.begin(), ids.end());
    ids.erase(std::unique(ids.begin(), ids.end()), ids.end());

    const uint32_t unseen = std::numeric_limits<uint32_t>::max();
    std::vector<uint32_t> order_of(ids.size(), unseen), transcript_of(num_paths), sizes;

    for (uint32_t p = 0; p < num_paths; ++p) {

        const uint32_t k = std::lower_bound(ids.begin(), ids.end(), paths[p].group_id) - ids.begin();

        if (order_of[k] == unseen) {

            order_of[k] = sizes.size();
            sizes.emplace_back(0);
        }

        transcript_of[p] = order_of[k];
        ++sizes[order_of[k]];
    }

    std::vector<std::vector<uint32_t> > transcripts(sizes.size());

    for (size_t t = 0; t < transcripts.size(); ++t) {

        transcripts[t].reserve(sizes[t]);
    }

    for (uint32_t p = 0; p < num_paths; ++p) {

        transcripts[transcript_of[p]].emplace_back(p);
    }

    return transcripts;
}

// Haplotypes (source ids) carrying the identical list of paths form one column;
// its multiplicity is the number of such haplotypes
// (src/path_abundance_estimator.cpp:493-546).  Columns come out in ascending
// order of their smallest source id (the reference's order is that of its hash map).
bool NestedPathAbundanceEstimator::sourceColumnsOf(GroupPosteriorProblem * columns, const std::vector<PathInfo> & paths) const {

    if (!wantsSourceColumns()) {

        return false;
    }

    findPathSourceGroups(columns, paths);
    return columns->numColumns() > 0;  // (paths without source ids: the batch's own error message)
}

void NestedPathAbundanceEstimator::findPathSourceGroups(GroupPosteriorProblem * problem, const std::vector<PathInfo> & paths) const {

    // (source id, path) incidences as one key each, grouped by source id with the paths of an id ascending.
    // The id sets are node-based containers (the boundary type): they are walked ONCE, into a flat array — three walks
    // (range, histogram, scatter) were 30 ms of host time per 200 k-path batch, most of a step's host work.
    // Haplotype ids are small consecutive integers in practice: a counting sort over the id range (two passes
    // over the flat incidences, which come in ascending path order) replaces the comparison sort then.
    size_t num_incidences = 0;

    for (auto & path: paths) {

        num_incidences += path.source_ids.size();
    }

    // (scratch of the calling thread, kept from cluster to cluster: six allocations per cluster were a fifth of this function)
    static thread_local std::vector<uint32_t> flat_ids, flat_end, first_of_id;
    static thread_local std::vector<uint64_t> source_paths, column_hash;
    static thread_local std::vector<int32_t> table;

    // (a thread keeps what the usual cluster needs; after a very large one the scratch goes back: a 100-thread team would otherwise
    // hold the peak of every thread for the life of the process)
    constexpr size_t scratch_kept = size_t(1) << 20;

    if (flat_ids.capacity() > scratch_kept && num_incidences <= scratch_kept / 4) {

        std::vector<uint32_t>().swap(flat_ids);
        std::vector<uint32_t>().swap(first_of_id);
        std::vector<uint64_t>().swap(source_paths);
        std::vector<uint64_t>().swap(column_hash);
        std::vector<int32_t>().swap(table);
    }

    flat_ids.resize(num_incidences);
    flat_end.resize(paths.size());
    uint32_t min_id = std::numeric_limits<uint32_t>::max();
    uint32_t max_id = 0;

    {
        size_t next = 0;

        for (size_t i = 0; i < paths.size(); ++i) {

            for (auto & id: paths[i].source_ids) {

                flat_ids[next++] = id;
            }

            flat_end[i] = next;

            if (!paths[i].source_ids.empty()) {  // ordered set: its ends are its range

                min_id = std::min(min_id, *paths[i].source_ids.begin());
                max_id = std::max(max_id, *paths[i].source_ids.rbegin());
            }
        }
    }

    source_paths.resize(num_incidences);

    if (num_incidences > 0 && static_cast<uint64_t>(max_id) - min_id < 4 * static_cast<uint64_t>(num_incidences) + 1024) {

        first_of_id.assign(static_cast<size_t>(max_id - min_id) + 2, 0);

        for (auto & id: flat_ids) {

            first_of_id[id - min_id + 1]++;
        }

        for (size_t i = 1; i < first_of_id.size(); ++i) {

            first_of_id[i] += first_of_id[i - 1];
        }

        size_t next = 0;

        for (size_t i = 0; i < paths.size(); ++i) {

            for (; next < flat_end[i]; ++next) {

                const uint32_t id = flat_ids[next];
                source_paths[first_of_id[id - min_id]++] = (static_cast<uint64_t>(id) << 32) | i;
            }
        }

    } else {

        size_t next = 0;

        for (size_t i = 0; i < paths.size(); ++i) {

            for (; next < flat_end[i]; ++next) {

                source_paths[next] = (static_cast<uint64_t>(flat_ids[next]) << 32) | i;
            }
        }

        std::sort(source_paths.begin(), source_paths.end());
    }

    // every run of one source id is that haplotype's path list; identical lists are found through a
    // small open-addressing table keyed by a hash of the list (column = first haplotype with the list)
    size_t table_size = 16;

    while (table_size < 2 * paths.size() + 16 && table_size < 4 * source_paths.size() + 16) {

        table_size <<= 1;
    }

    table.assign(table_size, -1);
    column_hash.clear();

    // upper bounds: every haplotype its own column
    const size_t max_columns = std::min<size_t>(num_incidences, static_cast<size_t>(max_id - min_id) + 1);
    column_hash.reserve(max_columns);
    problem->column_counts.reserve(max_columns);
    problem->beginColumns(max_columns);
    problem->column_path.reserve(num_incidences);

    size_t run_begin = 0;

    while (run_begin < source_paths.size()) {

        size_t run_end = run_begin;
        uint64_t hash = 1469598103934665603ull;

        while (run_end < source_paths.size() && (source_paths[run_end] >> 32) == (source_paths[run_begin] >> 32)) {

            hash = (hash ^ (source_paths[run_end] & 0xFFFFFFFFull)) * 1099511628211ull;
            ++run_end;
        }

        const size_t run_length = run_end - run_begin;
        size_t slot = hash & (table_size - 1);

        while (true) {

            const int32_t column = table[slot];

            if (column < 0) {

                table[slot] = problem->numColumns();
                column_hash.emplace_back(hash);

                problem->column_counts.emplace_back(1);

                for (size_t j = run_begin; j < run_end; ++j) {

                    problem->column_path.emplace_back(static_cast<uint32_t>(source_paths[j] & 0xFFFFFFFFull));
                }

                problem->column_path_off.emplace_back(problem->column_path.size());
                break;
            }

            bool identical = (column_hash[column] == hash) && (problem->column_path_off[column + 1] - problem->column_path_off[column] == run_length);

            for (size_t j = 0; identical && j < run_length; ++j) {

                identical = (problem->column_path[problem->column_path_off[column] + j] == static_cast<uint32_t>(source_paths[run_begin + j] & 0xFFFFFFFFull));
            }

            if (identical) {

                problem->column_counts[column]++;
                break;
            }

            slot = (slot + 1) & (table_size - 1);
        }

        run_begin = run_end;
    }
}

// One diplotype (set of `group_size` columns) of a transcript per requested sample, drawn from the transcript's posteriors — the
// draws of src/path_abundance_estimator.cpp:548-567, one per sample slot in slot order, from the caller's generator — and the
// paths its columns stand for (ascending column order) appended to the slot.
void NestedPathAbundanceEstimator::sampleGroupPathIndices(std::vector<std::vector<uint32_t> > * path_subset_samples, const GroupPosteriors & group_posteriors, const std::vector<uint32_t> & group, std::mt19937 * mt_rng) const {

    assert(group_posteriors.group_size == group_size);

    const std::vector<double> & weights = group_posteriors.posteriors;
    std::discrete_distribution<uint32_t> draw_set(weights.begin(), weights.end());

    uint32_t columns[8];
    assert(group_size <= 8);

    for (size_t slot = 0; slot < path_subset_samples->size(); ++slot) {

        const uint32_t * members = group_posteriors.set(draw_set(*mt_rng));

        std::copy(members, members + group_size, columns);
        std::sort(columns, columns + group_size);

        std::vector<uint32_t> & sample = (*path_subset_samples)[slot];

        for (uint32_t j = 0; j < group_size; ++j) {

            sample.emplace_back(group.at(columns[j]));
        }
    }
}

// src/path_abundance_estimator.cpp:569-606
void NestedPathAbundanceEstimator::selectPathSubsetIndices(PathSubsetWeights * path_subset_samples, const GroupPosteriors & group_posteriors, const GroupPosteriorProblem & problem) const {

    double sum_posterior = 0;

    for (size_t i = 0; i < group_posteriors.posteriors.size(); ++i) {

        if (group_posteriors.posteriors.at(i) < min_hap_prob) {

            continue;
        }

        std::vector<uint32_t> path_subset;
        size_t num_paths = 0;

        for (uint32_t j = 0; j < group_posteriors.group_size; ++j) {

            const uint32_t group = group_posteriors.set(i)[j];
            num_paths += problem.columnEnd(group) - problem.columnBegin(group);
        }

        path_subset.reserve(num_paths);

        for (uint32_t j = 0; j < group_posteriors.group_size; ++j) {

            const uint32_t group = group_posteriors.set(i)[j];
            path_subset.insert(path_subset.end(), problem.columnBegin(group), problem.columnEnd(group));
        }

        std::sort(path_subset.begin(), path_subset.end());

        // (the key moves into the map: no second copy of the subset)
        path_subset_samples->emplace(std::move(path_subset), 0.0).first->second += group_posteriors.posteriors.at(i);
        sum_posterior += group_posteriors.posteriors.at(i);
    }

    for (auto & subset_sample: *path_subset_samples) {

        subset_sample.second /= sum_posterior;
    }
}

// src/path_abundance_estimator.cpp:702-749 over the subsets and EM solutions of a device call.
void NestedPathAbundanceEstimator::mergeSubsetSolutions(std::vector<PathClusterEstimates> * path_cluster_estimates, const DeviceClusterBatch & cluster_batch, const std::vector<uint32_t> & clusters, const rpvg_hip_subset_em_view & subsets, const bool reset_first) const {

    assert(subsets.num_matrices == clusters.size());

    ScopedPhase merge_phase("nested: weighted merge");

    // (the device path is the diploid one; the keys below hold up to four paths: ploidy <= 4, as everywhere on the device)
    constexpr uint32_t max_group_size = 4;

    if (group_size > max_group_size) {

        throw EngineError("weighted merge of device-built subsets: group size above 4");
    }

    #pragma omp parallel for schedule(static, clusterChunk()) num_threads(hostThreads())
    for (size_t i = 0; i < clusters.size(); ++i) {

        auto & estimates = path_cluster_estimates->at(clusters.at(i));

        if (reset_first) {

            estimates.resetEstimates(0, 0);
        }

        assert(estimates.noise_count == 0);
        estimates.total_count = cluster_batch.totalReadCount(clusters.at(i));

        // (paths of one transcript inside a subset) -> (probability, abundance per path): the reference keeps a hash map of vectors
        // (src/path_abundance_estimator.cpp:702-745), the separate-calls path below an ordered map of them.  Here the entries of all
        // subsets go to one flat array — key = the paths + 1, zero-padded, which orders like the vectors do —, a stable sort brings
        // equal keys together in subset order, and the runs are added up in that order: the sums of the ordered map, addition for
        // addition, without a node allocation per (subset, transcript) — two maps per subset were 6.5 ms of single-thread time per
        // 5 000-cluster batch.
        struct Entry {

            std::array<uint32_t, max_group_size> key;
            uint32_t size;
            double weight;
            std::array<double, max_group_size> abundance;
        };

        static thread_local std::vector<Entry> entries;
        static thread_local std::vector<std::pair<uint32_t, uint32_t> > grouped;  // (group id, position in the subset's list)

        if (entries.capacity() > (size_t(1) << 18)) {  // (after a cluster with a hundred thousand entries: see findPathSourceGroups' scratch)

            std::vector<Entry>().swap(entries);
        }

        entries.clear();

        double sum_hap_prob = 0;

        const uint64_t first_subset = subsets.subset_off[i];
        const uint64_t last_subset = subsets.subset_off[i + 1];

        estimates.em_iterations.reserve(last_subset - first_subset);
        estimates.em_problem_paths.reserve(last_subset - first_subset);

        // the subsets come in lexicographic order of their path lists: the order of the ordered map of the separate calls
        for (uint64_t s = first_subset; s < last_subset; ++s) {

            const double weight = subsets.weight[s];
            assert(weight >= min_hap_prob);

            sum_hap_prob += weight;

            const uint32_t * path_begin = subsets.path + subsets.path_off[s];
            const uint32_t * path_end = subsets.path + subsets.path_off[s + 1];
            const uint32_t * col_begin = subsets.col_path + subsets.col_off[s];
            const uint32_t * col_end = subsets.col_path + subsets.col_off[s + 1];
            const double * abundances = subsets.abundances + subsets.col_off[s];

            assert(subsets.total_count[s] == estimates.total_count);

            estimates.em_iterations.emplace_back(subsets.iterations[s]);
            estimates.em_problem_paths.emplace_back(col_begin, col_end);

            estimates.noise_count += subsets.noise_count[s] * weight;

            // the subset's paths by transcript (group id ascending, the paths of a transcript in list order)
            grouped.clear();

            for (const uint32_t * path = path_begin; path != path_end; ++path) {

                grouped.emplace_back(estimates.paths.at(*path).group_id, static_cast<uint32_t>(path - path_begin));
            }

            std::stable_sort(grouped.begin(), grouped.end(), [](const std::pair<uint32_t, uint32_t> & lhs, const std::pair<uint32_t, uint32_t> & rhs) { return lhs.first < rhs.first; });

            for (size_t g0 = 0; g0 < grouped.size();) {

                size_t g1 = g0;

                while (g1 < grouped.size() && grouped[g1].first == grouped[g0].first) {

                    ++g1;
                }

                assert(g1 - g0 <= group_size && g1 - g0 <= max_group_size);

                Entry entry;
                entry.key.fill(0);
                entry.abundance.fill(0);
                entry.size = g1 - g0;
                entry.weight = weight;

                for (size_t j = g0; j < g1; ++j) {

                    const uint32_t path = path_begin[grouped[j].second];

                    const auto column_it = std::lower_bound(col_begin, col_end, path);
                    assert(column_it != col_end && *column_it == path);

                    const uint32_t multiplicity = std::count(path_begin, path_end, path);

                    entry.key[j - g0] = path + 1;
                    entry.abundance[j - g0] = (abundances[column_it - col_begin] * weight / multiplicity);
                }

                entries.emplace_back(entry);
                g0 = g1;
            }
        }

        std::stable_sort(entries.begin(), entries.end(), [](const Entry & lhs, const Entry & rhs) { return lhs.key < rhs.key; });

        for (size_t e0 = 0; e0 < entries.size();) {

            size_t e1 = e0;
            double posterior = 0;
            std::array<double, max_group_size> sums;
            sums.fill(0);

            while (e1 < entries.size() && entries[e1].key == entries[e0].key) {

                posterior += entries[e1].weight;

                for (uint32_t j = 0; j < entries[e0].size; ++j) {

                    sums[j] += entries[e1].abundance[j];
                }

                ++e1;
            }

            estimates.path_group_sets.emplace_back();

            for (uint32_t j = 0; j < entries[e0].size; ++j) {

                estimates.path_group_sets.back().emplace_back(entries[e0].key[j] - 1);
            }

            estimates.posteriors.emplace_back(posterior);
            estimates.abundances.insert(estimates.abundances.end(), sums.begin(), sums.begin() + entries[e0].size);

            e0 = e1;
        }

        // (see inferPathSubsetAbundance for the tolerance)
        assert(sum_hap_prob < 1 + 1e-9);
        estimates.noise_count += (1 - sum_hap_prob) * estimates.total_count;  // (src/path_abundance_estimator.cpp:749: unclamped, as the reference adds it)
    }
}

void NestedPathAbundanceEstimator::unpackMergedSolutions(std::vector<PathClusterEstimates> * path_cluster_estimates, const DeviceClusterBatch & cluster_batch, const std::vector<uint32_t> & clusters, const rpvg_hip_subset_em_view & subsets) const {

    assert(subsets.num_matrices == clusters.size());
    assert(subsets.set_count);

    ScopedPhase unpack_phase("nested: estimates out of the device block");

    // Every field of the estimates is overwritten (what resetEstimates(0, 0) clears, src/path_abundance_estimator.cpp:351), the
    // vectors of vectors element by element: their allocations survive from one batch to the next.
    for (size_t i = 0; i < clusters.size(); ++i) {

        auto & estimates = path_cluster_estimates->at(clusters[i]);

        const uint64_t first_subset = subsets.subset_off[i];
        const uint64_t num_subsets = subsets.subset_off[i + 1] - first_subset;
        const uint64_t first_set = subsets.path_off[first_subset];
        const uint32_t num_sets = subsets.set_count[i];

        estimates.total_count = cluster_batch.totalReadCount(clusters[i]);
        estimates.noise_count = (num_subsets > 0) ? subsets.cluster_noise_count[i] : estimates.total_count;  // (:749 with no subset kept)
        estimates.gibbs_read_count_samples.clear();

        estimates.path_group_sets.resize(num_sets);
        estimates.posteriors.assign(subsets.set_posterior + first_set, subsets.set_posterior + first_set + num_sets);
        estimates.abundances.clear();

        for (uint32_t q = 0; q < num_sets; ++q) {

            const uint64_t slot = first_set + q;
            auto & path_group_set = estimates.path_group_sets[q];

            if (subsets.set_second[slot] == 0xFFFFFFFFu) {

                path_group_set.assign(1, subsets.set_first[slot]);
                estimates.abundances.emplace_back(subsets.set_abundance[2 * slot]);

            } else {

                path_group_set.resize(2);
                path_group_set[0] = subsets.set_first[slot];
                path_group_set[1] = subsets.set_second[slot];
                estimates.abundances.emplace_back(subsets.set_abundance[2 * slot]);
                estimates.abundances.emplace_back(subsets.set_abundance[2 * slot + 1]);
            }
        }

        estimates.em_iterations.assign(subsets.iterations + first_subset, subsets.iterations + first_subset + num_subsets);
        estimates.em_problem_paths.resize(num_subsets);

        for (uint64_t s = 0; s < num_subsets; ++s) {

            estimates.em_problem_paths[s].assign(subsets.col_path + subsets.col_off[first_subset + s], subsets.col_path + subsets.col_off[first_subset + s + 1]);
        }
    }
}

// src/path_abundance_estimator.cpp:608-750 for all clusters at once.
void NestedPathAbundanceEstimator::inferPathSubsetAbundance(std::vector<PathClusterEstimates> * path_cluster_estimates, const DeviceClusterBatch & cluster_batch, const std::vector<uint32_t> & clusters, const std::vector<PathSubsetWeights> & path_subset_samples, std::vector<std::mt19937> * rngs, const bool reset_first) const {

    assert(clusters.size() == path_subset_samples.size());

    std::unique_ptr<ScopedPhase> build_phase(new ScopedPhase("nested: EM problem list"));

    // one EM problem per retained subset: its distinct paths
    std::vector<size_t> first_problem(clusters.size() + 1, 0);

    // (the subsets of a cluster sit in a node-based map: counted by the team, then one pass of additions)
    #pragma omp parallel for schedule(static, 1) num_threads(hostThreads())
    for (size_t i = 0; i < clusters.size(); ++i) {

        size_t num_retained = 0;

        for (auto & path_subset: path_subset_samples.at(i)) {

            num_retained += (path_subset.second >= min_hap_prob);
        }

        first_problem.at(i + 1) = num_retained;
    }

    for (size_t i = 0; i < clusters.size(); ++i) {

        first_problem.at(i + 1) += first_problem.at(i);
    }

    std::vector<EMProblem> problems(first_problem.back());

    #pragma omp parallel for schedule(static, 1) num_threads(hostThreads())
    for (size_t i = 0; i < clusters.size(); ++i) {

        size_t problem_idx = first_problem.at(i);

        for (auto & path_subset: path_subset_samples.at(i)) {

            if (path_subset.second < min_hap_prob) {

                continue;
            }

            assert(!path_subset.first.empty());

            auto & problem = problems.at(problem_idx);
            ++problem_idx;

            problem.cluster = clusters.at(i);
            problem.path_ids.reserve(path_subset.first.size());  // (one allocation per problem instead of one per doubling)
            std::unique_copy(path_subset.first.begin(), path_subset.first.end(), std::back_inserter(problem.path_ids));
        }
    }

    build_phase.reset();

    std::vector<EMSolution> solutions;
    EMAbundanceEstimator(&solutions, cluster_batch, problems, true);  // (:668)

    std::vector<CountSamples> count_samples;

    if (num_gibbs_samples > 0) {

        // the -n samples of a cluster are split over its subsets by sequential binomials on the subset
        // weights (src/path_abundance_estimator.cpp:675-697)
        std::vector<uint32_t> num_samples(problems.size(), 0);
        std::vector<uint64_t> seeds(problems.size(), 0);

        for (size_t i = 0; i < clusters.size(); ++i) {

            std::mt19937 * mt_rng = &rngs->at(clusters.at(i));

            uint32_t subset_gibbs_samples = num_gibbs_samples;
            double subset_gibbs_prob = 1;

            size_t problem_idx = first_problem.at(i);

            for (auto & path_subset: path_subset_samples.at(i)) {

                if (path_subset.second < min_hap_prob) {

                    continue;
                }

                if (subset_gibbs_samples > 0) {

                    assert(subset_gibbs_prob > 0);

                    std::binomial_distribution<uint32_t> path_read_count_sampler(subset_gibbs_samples, std::min(1.0, path_subset.second / subset_gibbs_prob));
                    const uint32_t cur_subset_gibbs_samples = path_read_count_sampler(*mt_rng);

                    subset_gibbs_samples -= cur_subset_gibbs_samples;
                    subset_gibbs_prob -= path_subset.second;

                    num_samples.at(problem_idx) = cur_subset_gibbs_samples;
                    seeds.at(problem_idx) = drawSeed(mt_rng);
                }

                ++problem_idx;
            }
        }

        gibbsReadCountSampler(&count_samples, cluster_batch, problems, solutions, num_samples, seeds);
    }

    std::unique_ptr<ScopedPhase> merge_phase(new ScopedPhase("nested: weighted merge"));

    #pragma omp parallel for schedule(dynamic, clusterChunk()) num_threads(hostThreads())
    for (size_t i = 0; i < clusters.size(); ++i) {

        auto & estimates = path_cluster_estimates->at(clusters.at(i));

        if (reset_first) {

            estimates.resetEstimates(0, 0);
        }

        assert(estimates.noise_count == 0);
        estimates.total_count = cluster_batch.totalReadCount(clusters.at(i));

        // (paths of one transcript inside a subset) -> (probability, abundance per path)
        std::map<std::vector<uint32_t>, std::pair<double, std::vector<double> > > path_group_estimates;

        double sum_hap_prob = 0;
        size_t problem_idx = first_problem.at(i);

        for (auto & path_subset: path_subset_samples.at(i)) {

            if (path_subset.second < min_hap_prob) {

                continue;
            }

            sum_hap_prob += path_subset.second;

            const auto & problem = problems.at(problem_idx);
            const auto & solution = solutions.at(problem_idx);
            ++problem_idx;

            assert(solution.total_count == estimates.total_count);

            estimates.em_iterations.emplace_back(solution.iterations);
            estimates.em_problem_paths.emplace_back(problem.path_ids);

            if (!count_samples.empty() && !count_samples.at(problem_idx - 1).noise_samples.empty()) {

                estimates.gibbs_read_count_samples.emplace_back(std::move(count_samples.at(problem_idx - 1)));
            }

            estimates.noise_count += solution.noise_count * path_subset.second;

            std::map<uint32_t, std::vector<uint32_t> > subset_path_group_index;

            for (auto & path: path_subset.first) {

                subset_path_group_index[estimates.paths.at(path).group_id].emplace_back(path);
            }

            for (auto & path_group: subset_path_group_index) {

                assert(path_group.second.size() <= group_size);

                auto path_group_estimates_it = path_group_estimates.emplace(path_group.second, std::make_pair(0.0, std::vector<double>(path_group.second.size(), 0)));
                path_group_estimates_it.first->second.first += path_subset.second;

                for (size_t j = 0; j < path_group.second.size(); ++j) {

                    const uint32_t path = path_group.second.at(j);

                    const auto column_it = std::lower_bound(problem.path_ids.begin(), problem.path_ids.end(), path);
                    assert(column_it != problem.path_ids.end() && *column_it == path);

                    const uint32_t multiplicity = std::count(path_subset.first.begin(), path_subset.first.end(), path);

                    path_group_estimates_it.first->second.second.at(j) += (solution.abundances.at(column_it - problem.path_ids.begin()) * path_subset.second / multiplicity);
                }
            }
        }

        assert(problem_idx == first_problem.at(i + 1));

        estimates.path_group_sets.reserve(path_group_estimates.size());
        estimates.posteriors.reserve(path_group_estimates.size());

        for (auto & group_estimates: path_group_estimates) {

            estimates.path_group_sets.emplace_back(group_estimates.first);
            estimates.posteriors.emplace_back(group_estimates.second.first);
            estimates.abundances.insert(estimates.abundances.end(), group_estimates.second.second.begin(), group_estimates.second.second.end());
        }

        // The reference asserts sum_hap_prob <= 1 up to 100 ulp here (src/path_abundance_estimator.cpp:748).  The weights are
        // posteriors normalised by a log-sum-exp over thousands of sets: they carry an ulp each, and with min_hap_prob 1e-5
        // their sum passes that tolerance on the rounding of whoever adds them up (fuzz seeds 6185, 20207, 21123 of
        // tests/fuzz_parity.py).  A library does not end its host process over that: the arithmetic is the reference's, the check keeps
        // the tolerance of an actual error.
        assert(sum_hap_prob < 1 + 1e-9);
        estimates.noise_count += (1 - sum_hap_prob) * estimates.total_count;  // (src/path_abundance_estimator.cpp:749: unclamped, as the reference adds it)
    }

    merge_phase.reset();

    // tens of thousands of small vectors: released by the team instead of one by one on return
    ScopedPhase teardown_phase("nested: teardown EM problems");

    auto old_problems = std::make_shared<std::vector<EMProblem> >(std::move(problems));
    auto old_solutions = std::make_shared<std::vector<EMSolution> >(std::move(solutions));

    dropNowOrLater([old_problems, old_solutions](const int threads) {

        #pragma omp parallel for schedule(static) num_threads(threads)
        for (size_t i = 0; i < old_problems->size(); ++i) {

            std::vector<uint32_t>().swap(old_problems->at(i).path_ids);
            std::vector<double>().swap(old_solutions->at(i).abundances);
        }
    });
}

}
