// Row construction — the step right before the inference hot path — behind the
// reference's types: FragmentLengthDist (src/fragment_length_dist.hpp:13-48, the
// parametric constructors and logProb), AlignmentPath (src/alignment_path.hpp:22-39,
// the fields addPathProbs reads) and a batch builder that takes what the caller's
// loop hands to ReadPathProbabilities::addPathProbs (src/main.cpp:889-905) for
// every cluster and returns the merged rows of all of them as a device-resident
// batch (src/read_path_probabilities.cpp:74-221, src/main.cpp:953-973 on the GPU,
// include/rpvg_rows.h).
#ifndef RPVG_AMD_READ_ROWS_HPP
#define RPVG_AMD_READ_ROWS_HPP

#include <cstdint>
#include <memory>
#include <utility>
#include <vector>

#include "../../include/rpvg_rows.h"
#include "hip_engine.hpp"
#include "path_cluster_estimates.hpp"

namespace rpvg_amd {

class FragmentLengthDist {

    public:

        FragmentLengthDist();
        FragmentLengthDist(const double mean_in, const double sd_in, const uint32_t sd_max_multi);
        FragmentLengthDist(const double loc_in, const double scale_in, const double shape_in, const uint32_t sd_max_multi);

        double loc() const { return loc_; }
        double scale() const { return scale_; }
        double shape() const { return shape_; }

        bool isValid() const;
        uint32_t maxLength() const { return max_length_; }
        double logProb(const uint32_t value) const;

        // logProb(v) for every uint16_t fragment length: the table the GPU path reads
        std::vector<double> logProbTable() const;

    private:

        double loc_;
        double scale_;
        double shape_;
        double max_length_;

        std::vector<double> log_prob_buffer;
};

// The AlignmentPath fields addPathProbs reads; the gbwt search state is replaced by the cluster-local
// indices of the paths it locates (align_paths_ids mapped through clustered_path_index).
struct AlignmentPath {

    uint8_t min_mapq = 0;
    int32_t score_sum = 0;
    uint16_t align_length = 0;
    uint16_t frag_length = 0;

    std::vector<uint32_t> path_idx;

    AlignmentPath() {}
    AlignmentPath(const uint8_t min_mapq_in, const int32_t score_sum_in, const uint16_t align_length_in, const uint16_t frag_length_in, const std::vector<uint32_t> & path_idx_in) : min_mapq(min_mapq_in), score_sum(score_sum_in), align_length(align_length_in), frag_length(frag_length_in), path_idx(path_idx_in) {}
};

// Collects the distinct alignment-path lists of the clusters of a batch (the reference's
// align_paths_index entries: list + multiplicity) in the flat layout of include/rpvg_rows.h.
class AlignmentBatchBuilder {

    public:

        AlignmentBatchBuilder();

        // Opens the next cluster.  group_name_index (one group per path, 0 .. num_groups-1) turns on the
        // collapsing of `-i transcripts` with --path-info; pass an empty vector otherwise.
        void beginCluster(const std::vector<PathInfo> & cluster_paths, const std::vector<uint32_t> & group_name_index = std::vector<uint32_t>(), const uint32_t num_groups = 0);

        // One alignment-path list with its multiplicity; align_paths.back() is the noise entry
        // (src/read_path_probabilities.cpp:43-46).
        void addAlignmentPaths(const std::vector<AlignmentPath> & align_paths, const uint32_t read_count);

        uint32_t numClusters() const { return cluster_read_off.size() - 1; }
        uint64_t totalReadCount(const uint32_t cluster) const { return cluster_total_reads.at(cluster); }

        rpvg_alignment_batch view() const;

    private:

        bool collapse;

        std::vector<uint64_t> cluster_read_off, cluster_path_off, cluster_group_off, read_align_off, align_path_off;
        std::vector<double> path_effective_length;
        std::vector<uint32_t> path_source_count, path_group, read_count, align_path_idx;
        std::vector<uint8_t> read_min_mapq;
        std::vector<int32_t> read_noise_score, align_score_sum;
        std::vector<uint16_t> align_length, align_frag_length;
        std::vector<uint64_t> cluster_total_reads;
};

// The alignment-path lists of a batch resident on the GPU (validated copy; rpvg_hip_alignments_upload).
class DeviceAlignmentBatch {

    public:

        DeviceAlignmentBatch(std::shared_ptr<HipEngine> engine_in, const AlignmentBatchBuilder & alignments);
        ~DeviceAlignmentBatch();

        DeviceAlignmentBatch(const DeviceAlignmentBatch &) = delete;
        DeviceAlignmentBatch & operator=(const DeviceAlignmentBatch &) = delete;

        const rpvg_hip_alignments * handle() const { return device_alignments; }
        const std::shared_ptr<HipEngine> & engine() const { return hip_engine; }
        const std::vector<double> & totalReadCounts() const { return total_read_count; }

    private:

        std::shared_ptr<HipEngine> hip_engine;
        rpvg_hip_alignments * device_alignments;
        std::vector<double> total_read_count;
};

// addPathProbs for every list of the batch + the caller's sort and merge, on the GPU; the rows stay on
// the device as the batch the estimators take.
std::unique_ptr<DeviceClusterBatch> constructReadPathProbabilities(const DeviceAlignmentBatch & alignments, const FragmentLengthDist & fragment_length_dist, const bool is_single_end, const double min_noise_prob, const double prob_precision);

// The same from host memory (uploads the lists first).
std::unique_ptr<DeviceClusterBatch> constructReadPathProbabilities(std::shared_ptr<HipEngine> engine, const AlignmentBatchBuilder & alignments, const FragmentLengthDist & fragment_length_dist, const bool is_single_end, const double min_noise_prob, const double prob_precision);

}

#endif
