// On-disk formats on the input side of the inference hot path:
//   <prefix>_probs.txt[.gz]  the dump of the hot-path input written by `--write-probs`
//                            (ProbabilityClusterWriter, src/threaded_output_writer.cpp:42-95)
//   -f path info TSV         Name / Length / Transcript / [Reference] / Haplotypes
//                            (parseHaplotypeTranscriptInfo, src/main.cpp:239-353)
// so that a real rpvg run can be replayed through the GPU engine.  Plain or gzip (zlib).
#ifndef RPVG_AMD_CLUSTER_IO_HPP
#define RPVG_AMD_CLUSTER_IO_HPP

#include <string>
#include <unordered_map>
#include <vector>

#include "../path_cluster_estimates.hpp"
#include "../read_path_probabilities.hpp"

namespace rpvg_amd {

// One block of the dump: the cluster's paths (name, length, effective length; local index = position)
// and its merged read rows.
struct ProbabilityCluster {

    std::vector<PathInfo> paths;
    std::vector<ReadPathProbabilities> cluster_probs;

    // What the reference ranks clusters by (src/main.cpp:811-827: descending (number of alignment-path lists, index of the
    // cluster in PathClusters); the rank is the output ClusterID and the offset of the cluster's random seed, :849,976).
    // The reference's dump does not hold them; a producer that knows them writes "# <lists> <index>" as the marker line of
    // the block instead of "#" (has_rank_key).  Without them a replay ranks by read count.
    bool has_rank_key = false;
    uint64_t num_align_lists = 0;
    uint64_t cluster_index = 0;
};

// Text file helpers (zlib): reading passes plain files through; writing gzips when the name ends in ".gz".
std::string readTextFile(const std::string & filename);
void writeTextFile(const std::string & filename, const std::string & text);

std::vector<ProbabilityCluster> readProbabilityClusters(const std::string & filename, const double prob_precision);

// Same format as the reference's writer (a block whose cluster has a rank key gets the extended marker line); gzip when the
// name ends in ".gz".
void writeProbabilityClusters(const std::string & filename, const std::vector<ProbabilityCluster> & clusters, const double prob_precision);

// Path name -> PathInfo with group_id (dense transcript ids in first-seen order), source_count and — when
// parse_haplotype_ids — source_ids (dense haplotype ids in first-seen order); use_transcript_names
// replaces the name by the transcript's (the reference does that when collapsing haplotypes).
std::unordered_map<std::string, PathInfo> parseHaplotypeTranscriptInfo(const std::string & filename, const bool parse_haplotype_ids, const bool use_transcript_names);

// Fills group_id / source_count / source_ids of the clusters' paths from the info by path name
// (src/main.cpp:866-872).  Throws std::runtime_error when a path is missing from the info.
void applyHaplotypeTranscriptInfo(std::vector<ProbabilityCluster> * clusters, const std::unordered_map<std::string, PathInfo> & haplotype_transcript_info);

}

#endif
