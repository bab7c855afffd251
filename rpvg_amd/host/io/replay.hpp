// Replays a dumped rpvg inference input (`--write-probs` file, optional `-f` path info) through the GPU
// estimators and writes the reference's result files — the stage of src/main.cpp:766-1088 that sits
// around the hot path, without the alignment search in front of it.
#ifndef RPVG_AMD_REPLAY_HPP
#define RPVG_AMD_REPLAY_HPP

#include <string>

#include "../../../include/rpvg_batch.h"
#include "cluster_io.hpp"
#include "estimates_writers.hpp"

namespace rpvg_amd {

// Orders clusters by descending read count (the reference ranks them by their number of alignment lists,
// which the dump does not keep: src/main.cpp:811-827) and numbers them from 1.
void rankClusters(std::vector<ProbabilityCluster> * clusters);

// Writes every result file the reference writes for `inference_model` (src/main.cpp:1016-1088).
void writeEstimates(const std::string & inference_model, const rpvg_params & params, const ClusterEstimatesList & path_cluster_estimates, const std::string & output_prefix, const uint32_t unaligned_read_count);

// The whole replay on GPU `device`.  Returns the number of clusters processed.
size_t replayInference(const std::string & probs_filename, const std::string & path_info_filename, const std::string & inference_model, const rpvg_params & params, const std::string & output_prefix, const int device, const uint32_t unaligned_read_count);

}

#endif
