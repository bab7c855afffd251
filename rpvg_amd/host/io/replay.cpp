#include "replay.hpp"

#include <algorithm>
#include <numeric>
#include <stdexcept>

#include "../estimator_factory.hpp"

namespace rpvg_amd {

// The order of the reference's cluster loop (src/main.cpp:811-827): descending (number of alignment-path lists, cluster
// index) when every block of the dump carries that key; otherwise descending read count, dump order among equals (the
// reference's dump holds the merged rows, not the lists).
void rankClusters(std::vector<ProbabilityCluster> * clusters) {

    const bool by_rank_key = !clusters->empty() && std::all_of(clusters->begin(), clusters->end(), [](const ProbabilityCluster & cluster) { return cluster.has_rank_key; });

    std::vector<std::pair<std::pair<uint64_t, uint64_t>, size_t> > keys;

    for (size_t i = 0; i < clusters->size(); ++i) {

        if (by_rank_key) {

            keys.emplace_back(std::make_pair(clusters->at(i).num_align_lists, clusters->at(i).cluster_index), i);
            continue;
        }

        uint64_t read_count = 0;

        for (auto & probs: clusters->at(i).cluster_probs) {

            read_count += probs.readCount();
        }

        keys.emplace_back(std::make_pair(read_count, uint64_t(0)), i);
    }

    std::stable_sort(keys.begin(), keys.end(), [](const std::pair<std::pair<uint64_t, uint64_t>, size_t> & lhs, const std::pair<std::pair<uint64_t, uint64_t>, size_t> & rhs) { return lhs.first > rhs.first; });

    std::vector<ProbabilityCluster> ranked;
    ranked.reserve(clusters->size());

    for (auto & key: keys) {

        ranked.emplace_back(std::move(clusters->at(key.second)));
    }

    clusters->swap(ranked);
}

void writeEstimates(const std::string & inference_model, const rpvg_params & params, const ClusterEstimatesList & path_cluster_estimates, const std::string & output_prefix, const uint32_t unaligned_read_count) {

    if (params.num_gibbs_samples > 0 && inference_model != "haplotypes") {

        ReadCountGibbsSamplesWriter read_count_samples_writer(output_prefix + "_gibbs", params.num_gibbs_samples);

        for (auto & cur_estimates: path_cluster_estimates) {

            read_count_samples_writer.addSamples(cur_estimates);
        }

        read_count_samples_writer.addNoiseTranscript(unaligned_read_count);
        read_count_samples_writer.close();
    }

    if (inference_model == "haplotypes") {

        JointHaplotypeEstimatesWriter joint_haplotype_estimates_writer(output_prefix, params.ploidy, params.prob_precision);
        joint_haplotype_estimates_writer.addEstimates(path_cluster_estimates);
        joint_haplotype_estimates_writer.close();
        return;
    }

    const double total_transcript_count = totalTranscriptCount(path_cluster_estimates);

    if (inference_model == "haplotype-transcripts") {

        HaplotypeAbundanceEstimatesWriter haplotype_abundance_estimates_writer(output_prefix, params.ploidy, total_transcript_count);
        JointHaplotypeAbundanceEstimatesWriter joint_haplotype_abundance_estimates_writer(output_prefix + "_joint", params.ploidy, params.prob_precision, total_transcript_count);

        haplotype_abundance_estimates_writer.addEstimates(path_cluster_estimates);
        joint_haplotype_abundance_estimates_writer.addEstimates(path_cluster_estimates);

        haplotype_abundance_estimates_writer.addNoiseTranscript(unaligned_read_count);
        haplotype_abundance_estimates_writer.close();

        joint_haplotype_abundance_estimates_writer.addNoiseTranscript(unaligned_read_count);
        joint_haplotype_abundance_estimates_writer.close();

    } else {

        AbundanceEstimatesWriter abundance_estimates_writer(output_prefix, total_transcript_count);
        abundance_estimates_writer.addEstimates(path_cluster_estimates);
        abundance_estimates_writer.addNoiseTranscript(unaligned_read_count);
        abundance_estimates_writer.close();
    }
}

size_t replayInference(const std::string & probs_filename, const std::string & path_info_filename, const std::string & inference_model, const rpvg_params & params, const std::string & output_prefix, const int device, const uint32_t unaligned_read_count) {

    auto clusters = readProbabilityClusters(probs_filename, params.prob_precision);

    if (!path_info_filename.empty()) {

        const auto haplotype_transcript_info = parseHaplotypeTranscriptInfo(path_info_filename, inference_model == "haplotype-transcripts", false);
        applyHaplotypeTranscriptInfo(&clusters, haplotype_transcript_info);

    } else if (inference_model == "haplotype-transcripts") {

        throw std::runtime_error("haplotype-transcripts inference needs the path info file (-f)");
    }

    rankClusters(&clusters);

    auto engine = std::make_shared<HipEngine>(device);
    auto path_estimator = makePathEstimator(inference_model, params, engine);

    FlatClusterRows rows;

    for (auto & cluster: clusters) {

        rows.addCluster(cluster.cluster_probs, cluster.paths.size());
    }

    const DeviceClusterBatch cluster_batch(engine, rows.view());

    std::vector<PathClusterEstimates> estimates(clusters.size());

    for (size_t i = 0; i < clusters.size(); ++i) {

        estimates.at(i).paths = clusters.at(i).paths;
    }

    path_estimator->estimateBatchSeeded(&estimates, cluster_batch, params.rng_seed);

    ClusterEstimatesList path_cluster_estimates;
    path_cluster_estimates.reserve(estimates.size());

    for (size_t i = 0; i < estimates.size(); ++i) {

        path_cluster_estimates.emplace_back(i + 1, std::move(estimates.at(i)));
    }

    writeEstimates(inference_model, params, path_cluster_estimates, output_prefix, unaligned_read_count);

    return clusters.size();
}

}
