// C entry points over the file formats and the replay (harness: tests bind them with ctypes).

#include <algorithm>
#include <sstream>
#include <string>

#include "../estimator_factory.hpp"
#include "../flat_batch.hpp"
#include "replay.hpp"

using namespace rpvg_amd;

namespace {

thread_local std::string io_last_error;

// Clusters of a flat batch with generated names: path j of cluster k is "c<k>_p<j>", its transcript
// "c<k>_t<group>", its haplotypes "c<k>_h<source id>".
std::vector<ProbabilityCluster> clustersFromBatch(const rpvg_cluster_batch & batch, const double prob_precision) {

    std::vector<ProbabilityCluster> clusters(batch.num_clusters);

    for (uint32_t k = 0; k < batch.num_clusters; ++k) {

        uint32_t local = 0;

        for (uint64_t p = batch.cluster_path_off[k]; p < batch.cluster_path_off[k + 1]; ++p, ++local) {

            PathInfo path("c" + std::to_string(k) + "_p" + std::to_string(local));
            path.group_id = batch.path_group_id[p];
            path.source_count = batch.path_source_count[p];
            path.source_ids.insert(batch.source_id + batch.path_source_off[p], batch.source_id + batch.path_source_off[p + 1]);
            path.effective_length = batch.path_effective_length ? batch.path_effective_length[p] : 0;
            path.length = static_cast<uint32_t>(path.effective_length) + 50;
            clusters[k].paths.emplace_back(std::move(path));
        }

        for (uint64_t r = batch.cluster_row_off[k]; r < batch.cluster_row_off[k + 1]; ++r) {

            ReadPathProbabilities::PathProbs path_probs;

            for (uint64_t g = rpvg_batch_row_group_offset(&batch, r); g < rpvg_batch_row_group_offset(&batch, r + 1); ++g) {

                path_probs.emplace_back(batch.grp_prob[g], std::vector<uint32_t>(batch.path_idx + rpvg_batch_group_entry_offset(&batch, g), batch.path_idx + rpvg_batch_group_entry_offset(&batch, g + 1)));
            }

            clusters[k].cluster_probs.emplace_back(batch.row_count[r], batch.row_noise[r], path_probs, prob_precision);
        }
    }

    return clusters;
}

}

extern "C" {

const char * rpvg_amd_io_last_error(void) {

    return io_last_error.c_str();
}

// What parseHaplotypeTranscriptInfo (the `-f` parser, src/main.cpp:239-353) makes of a path info file, one line per
// path in file-independent (name) order: name <tab> group_id <tab> source_count <tab> source ids ascending, comma
// separated.  Written to out_filename.  Returns the number of paths, -1 on failure.
int64_t rpvg_amd_info_table(const char * info_filename, int parse_haplotype_ids, int use_transcript_names, const char * out_filename) {

    try {

        const auto info = parseHaplotypeTranscriptInfo(info_filename, parse_haplotype_ids != 0, use_transcript_names != 0);

        std::vector<const std::pair<const std::string, PathInfo> *> rows;

        for (auto & entry: info) {

            rows.emplace_back(&entry);
        }

        std::sort(rows.begin(), rows.end(), [](const auto * lhs, const auto * rhs) { return lhs->first < rhs->first; });

        std::stringstream out;

        for (auto & row: rows) {

            std::vector<uint32_t> ids(row->second.source_ids.begin(), row->second.source_ids.end());
            std::sort(ids.begin(), ids.end());

            out << row->first << "\t" << row->second.name << "\t" << row->second.group_id << "\t" << row->second.source_count << "\t";

            for (size_t i = 0; i < ids.size(); ++i) {

                out << (i ? "," : "") << ids[i];
            }

            out << "\n";
        }

        writeTextFile(out_filename, out.str());
        return rows.size();

    } catch (const std::exception & e) {

        io_last_error = e.what();
        return -1;
    }
}

int rpvg_amd_batch_write_files_ranked(const rpvg_cluster_batch * batch, const char * probs_filename, const char * path_info_filename, double prob_precision, const uint64_t * num_align_lists, const uint64_t * cluster_index);

// Writes the batch as a `--write-probs` dump and a matching `-f` path info file.
int rpvg_amd_batch_write_files(const rpvg_cluster_batch * batch, const char * probs_filename, const char * path_info_filename, double prob_precision) {

    return rpvg_amd_batch_write_files_ranked(batch, probs_filename, path_info_filename, prob_precision, nullptr, nullptr);
}

// The same with the rank key of the reference's cluster loop in every block's marker line ("# <lists> <index>",
// cluster_io.hpp): num_align_lists[k], cluster_index[k] of cluster k of the batch (both NULL: the reference's bare "#").
int rpvg_amd_batch_write_files_ranked(const rpvg_cluster_batch * batch, const char * probs_filename, const char * path_info_filename, double prob_precision, const uint64_t * num_align_lists, const uint64_t * cluster_index) {

    try {

        auto clusters = clustersFromBatch(*batch, prob_precision);

        if (num_align_lists && cluster_index) {

            for (size_t k = 0; k < clusters.size(); ++k) {

                clusters[k].has_rank_key = true;
                clusters[k].num_align_lists = num_align_lists[k];
                clusters[k].cluster_index = cluster_index[k];
            }
        }

        writeProbabilityClusters(probs_filename, clusters, prob_precision);

        std::stringstream info;
        info << "Name\tLength\tTranscript\tHaplotypes" << std::endl;

        for (uint32_t k = 0; k < clusters.size(); ++k) {

            for (auto & path: clusters[k].paths) {

                info << path.name << "\t" << path.length << "\tc" << k << "_t" << path.group_id << "\t";

                bool is_first = true;

                for (auto & id: path.source_ids) {

                    info << (is_first ? "" : ",") << "c" << k << "_h" << id;
                    is_first = false;
                }

                if (path.source_ids.empty()) {

                    for (uint32_t i = 0; i < path.source_count; ++i) {

                        info << (i ? "," : "") << "c" << k << "_x" << i;
                    }
                }

                info << std::endl;
            }
        }

        writeTextFile(path_info_filename, info.str());
        return 0;

    } catch (const std::exception & e) {

        io_last_error = e.what();
        return -1;
    }
}

// Reads a dump (+ optional path info) back into a flat batch, clusters ranked as the replay ranks them.
// Free with rpvg_amd_synth_free(); view with rpvg_amd_synth_view()/rpvg_amd_synth_sizes().
void * rpvg_amd_batch_read_files(const char * probs_filename, const char * path_info_filename, int parse_haplotype_ids, double prob_precision) {

    try {

        auto clusters = readProbabilityClusters(probs_filename, prob_precision);

        if (path_info_filename && path_info_filename[0]) {

            applyHaplotypeTranscriptInfo(&clusters, parseHaplotypeTranscriptInfo(path_info_filename, parse_haplotype_ids != 0, false));
        }

        rankClusters(&clusters);

        FlatBatchStorage * batch = new FlatBatchStorage();

        for (auto & cluster: clusters) {

            batch->addCluster(cluster.paths, cluster.cluster_probs);
        }

        return batch;

    } catch (const std::exception & e) {

        io_last_error = e.what();
        return nullptr;
    }
}

// GPU replay: dump -> estimators -> result files.  Returns the number of clusters, or -1.
int64_t rpvg_amd_replay(const char * probs_filename, const char * path_info_filename, const char * inference_model, const rpvg_params * params, const char * output_prefix, int device, uint32_t unaligned_read_count) {

    try {

        return replayInference(probs_filename, path_info_filename ? path_info_filename : "", inference_model, *params, output_prefix, device, unaligned_read_count);

    } catch (const std::exception & e) {

        io_last_error = e.what();
        return -1;
    }
}

// Writes the reference's result files for estimates given as a flat view (e.g. computed elsewhere) over
// the clusters of a dump: the writers without the GPU.  Cluster i of the view is cluster i of the ranked dump.
int rpvg_amd_write_estimates(const char * probs_filename, const char * path_info_filename, const char * inference_model, const rpvg_params * params, const rpvg_estimates_view * view, const char * output_prefix, uint32_t unaligned_read_count) {

    try {

        auto clusters = readProbabilityClusters(probs_filename, params->prob_precision);

        if (path_info_filename && path_info_filename[0]) {

            applyHaplotypeTranscriptInfo(&clusters, parseHaplotypeTranscriptInfo(path_info_filename, std::string(inference_model) == "haplotype-transcripts", false));
        }

        rankClusters(&clusters);

        if (view->num_clusters != clusters.size()) {

            throw std::runtime_error("estimates view and dump disagree on the number of clusters");
        }

        ClusterEstimatesList path_cluster_estimates;

        for (uint32_t k = 0; k < view->num_clusters; ++k) {

            PathClusterEstimates estimates;
            estimates.paths = clusters[k].paths;

            for (uint64_t s = view->set_off[k]; s < view->set_off[k + 1]; ++s) {

                estimates.path_group_sets.emplace_back(view->members + view->member_off[s], view->members + view->member_off[s + 1]);
                estimates.posteriors.emplace_back(view->posteriors[s]);
            }

            estimates.abundances.assign(view->abundances + view->abund_off[k], view->abundances + view->abund_off[k + 1]);
            estimates.noise_count = view->noise_count[k];
            estimates.total_count = view->total_count[k];

            for (uint64_t g = view->gibbs_off[k]; g < view->gibbs_off[k + 1]; ++g) {

                CountSamples count_samples;
                count_samples.path_ids.assign(view->gibbs_path + view->gibbs_path_off[g], view->gibbs_path + view->gibbs_path_off[g + 1]);
                count_samples.noise_samples.assign(view->gibbs_noise + view->gibbs_noise_off[g], view->gibbs_noise + view->gibbs_noise_off[g + 1]);
                count_samples.abundance_samples.assign(view->gibbs_abund + view->gibbs_abund_off[g], view->gibbs_abund + view->gibbs_abund_off[g + 1]);
                estimates.gibbs_read_count_samples.emplace_back(std::move(count_samples));
            }

            path_cluster_estimates.emplace_back(k + 1, std::move(estimates));
        }

        writeEstimates(inference_model, *params, path_cluster_estimates, output_prefix, unaligned_read_count);
        return 0;

    } catch (const std::exception & e) {

        io_last_error = e.what();
        return -1;
    }
}

}
