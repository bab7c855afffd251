// RAII handles over the C ABI of the GPU engine (include/rpvg_hip.h) for the
// host-side estimator classes.  Nothing here computes: it flattens the
// reference's row containers into the ABI's ragged arrays and forwards.
#ifndef RPVG_AMD_HIP_ENGINE_HPP
#define RPVG_AMD_HIP_ENGINE_HPP

#include <array>
#include <atomic>
#include <cstdint>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/rpvg_hip.h"
#include "path_cluster_estimates.hpp"
#include "pipeline_lanes.hpp"
#include "read_path_probabilities.hpp"

namespace rpvg_amd {

// Thrown when a C-ABI call fails (the reference would have hit an assert).
class EngineError : public std::runtime_error {

    public:

        explicit EngineError(const std::string & what_in) : std::runtime_error(what_in) {}
};

// One GPU + one HIP stream.  Shared by every estimator bound to that GPU.
class HipEngine {

    public:

        // uploader: an engine whose only job is DeviceClusterBatch construction next to an estimating engine on the same GPU
        // (rpvg_hip_create_uploader: a stream, and a hardware queue, of its own)
        // host_lanes: host lanes a batch is cut into (PathEstimator::runInLanes); 0 = the default (RPVG_AMD_LANES, else 2) —
        // the engines of a BatchPipeline run whole batches side by side instead and take 1
        explicit HipEngine(const int device, const bool uploader = false, const int host_lanes = 0);
        ~HipEngine();

        HipEngine(const HipEngine &) = delete;
        HipEngine & operator=(const HipEngine &) = delete;

        // The context of the calling host thread's lane: lane 0 (the engine's own context) unless the thread is
        // the second lane of a pipelined batch, which has a context — streams, scratch ordering — of its own so
        // that the kernels of the two lanes run side by side on the GPU.
        rpvg_hip_ctx * ctx() const {

            // (no lock: the slots of lane_contexts never move and a slot is published when its context exists — a combiner leader
            // creates its context while another leader's batch runs on the one beside it)
            const int lane = currentLane();

            if (lane > 0 && static_cast<size_t>(lane) <= lane_contexts.size()) {

                if (rpvg_hip_ctx * lane_context = lane_contexts[lane - 1].load(std::memory_order_acquire)) {

                    return lane_context;
                }
            }

            return context;
        }
        int device() const { return device_id; }
        int hostLanes() const { return host_lanes; }

        // Lane of the calling thread (thread local; 0 by default).
        static int & currentLane();

        // Kernel statistics of both lanes together (include/rpvg_hip.h, rpvg_hip_kernel_stats).
        void stats(rpvg_hip_kernel_stats * stats_out) const;
        void resetStats() const;

        // The same over several engines of one GPU (the workers of a BatchPipeline): sums, and the union of their busy spans.
        static void stats(const std::vector<const HipEngine *> & engines, rpvg_hip_kernel_stats * stats_out);

        static int deviceCount();

        // The engine estimators are bound to when their constructor is called with the reference's own parameter list
        // (src/path_abundance_estimator.hpp:22,55, src/path_posterior_estimator.hpp:22,33): one per process, on GPU
        // RPVG_AMD_DEVICE (default 0), created on first use.  Throws EngineError without a usable GPU.
        static std::shared_ptr<HipEngine> processDefault();

        // Throws EngineError carrying rpvg_hip_last_error() when status != 0.
        static void check(const int status, const char * what);

        // Host lane `lane` >= 1 of the engine (pipeline_lanes.hpp): a persistent worker thread and a device
        // context of its own, started on first use.  Lane 0 is the calling thread with the engine's context.
        PipelineWorker & lane(const int lane);

        static constexpr int max_lanes = 4;

        // Slot `slot` (0 .. max_combiner_slots - 1) of PathEstimator::estimate()'s call combiner: a device context of its own, made on
        // first use — lean like the engines of a BatchPipeline (three side streams, the main stream on a hardware queue of its own:
        // chains of short dependent kernels, several of them on the GPU at once).  Returns the lane id a leader sets as its
        // currentLane() while its batch runs (ctx() then hands out that context).
        int combinerLane(const int slot);

        static constexpr int max_combiner_slots = 6;

    private:

        rpvg_hip_ctx * context;
        int device_id;
        int host_lanes;

        std::mutex lane_mutex;
        // (lanes 1 .. max_lanes - 1: the host lanes of a batch; behind them the call combiner's slots; a slot is set once, for good)
        std::array<std::atomic<rpvg_hip_ctx *>, max_lanes - 1 + max_combiner_slots> lane_contexts;
        std::vector<std::unique_ptr<PipelineWorker> > lane_workers;  // (under lane_mutex)

        std::vector<rpvg_hip_ctx *> laneContexts() const;
};

// Flat host copy of the rows of K clusters (the arrays rpvg_cluster_batch
// points into), built from the reference's row containers.
class FlatClusterRows {

    public:

        FlatClusterRows();

        void addCluster(const std::vector<ReadPathProbabilities> & cluster_probs, const uint32_t num_paths);

        // The same with the PathInfo fields the device reads: PathInfo::group_id and PathInfo::source_ids (the batch then
        // carries its haplotype columns, rpvg_hip_batch_upload).  All clusters of a batch with them, or none.
        void addCluster(const std::vector<ReadPathProbabilities> & cluster_probs, const std::vector<PathInfo> & paths);

        // The clusters of `other` behind this one's (the call combiner of PathEstimator::estimate() joins the clusters its
        // callers flattened).  Both with paths, or both without.
        void append(const FlatClusterRows & other);

        uint32_t numClusters() const { return cluster_row_off.size() - 1; }
        rpvg_cluster_batch view() const;

    private:

        // (the two long offset arrays in 32 bits: rpvg_cluster_batch::row_grp_off32 / grp_idx_off32 — and as counts of one byte,
        // row_grp_count8 / grp_idx_count8, which is what the copy to the GPU takes while they fit: fewer bytes to copy)
        std::vector<uint64_t> cluster_row_off, cluster_path_off, path_source_off;
        std::vector<uint32_t> row_grp_off, grp_idx_off;
        std::vector<uint8_t> row_grp_count, grp_idx_count;
        bool counts_fit;
        std::vector<uint32_t> row_count, path_idx, path_group_id, source_id;
        std::vector<double> row_noise, grp_prob;

        // the narrow forms of include/rpvg_batch.h, written next to the 32-bit arrays while they fit (the copy to the GPU takes them
        // instead): cluster-local path indices and source ids in 16 bits, read counts in one byte with the list of the rows that
        // need more, noise probabilities as indices into the table of their distinct values
        std::vector<uint16_t> path_idx16, source_id16, row_noise16;
        std::vector<uint8_t> row_count8;
        std::vector<uint32_t> row_count_escape_row, row_count_escape_count;
        std::vector<double> row_noise_table;
        std::unordered_map<uint64_t, uint16_t> noise_index;  // bits of a noise probability -> its place in the table
        bool paths_fit16, sources_fit16, noise_fits16;

        uint16_t noiseIndex(const double noise);
};

// One cluster flattened into page-locked memory that the GPU reads where it lies (include/rpvg_batch.h, rpvg_cluster_segment): what
// a thread that calls PathEstimator::estimate() makes of its cluster before it parks the call (PathEstimator::CallCombiner).  The
// block is kept and reused from call to call (a thread's next cluster is rarely larger than its last; it grows by size class).
class ClusterSegment {

    public:

        ClusterSegment();
        ~ClusterSegment();

        ClusterSegment(const ClusterSegment &) = delete;
        ClusterSegment & operator=(const ClusterSegment &) = delete;

        // The haplotype columns of a cluster as its caller formed them (findPathSourceGroups): multiplicities, list offsets
        // [num + 1], the lists.
        struct Columns {

            uint32_t num;
            const uint32_t * counts;
            const uint32_t * path_off;
            const uint32_t * paths;
        };

        // Two passes over the rows: sizes, then the arrays at their final places.  with_sources: the segment carries
        // PathInfo::group_id and PathInfo::source_ids (the device forms the haplotype columns behind the upload) — or, with
        // `columns`, PathInfo::group_id and the columns themselves (the upload then forms nothing and waits for nothing but its kernel).
        void flatten(const std::vector<ReadPathProbabilities> & cluster_probs, const std::vector<PathInfo> & paths, const bool with_sources, const Columns * columns = nullptr);

        const rpvg_cluster_segment & view() const { return segment; }

    private:

        void * block;
        uint64_t capacity;
        rpvg_cluster_segment segment;
};

// K clusters resident on the GPU.
class DeviceClusterBatch {

    public:

        // From the segments of K callers (rpvg_hip_batch_upload_segments): no joined host copy, no copy commands.
        DeviceClusterBatch(std::shared_ptr<HipEngine> engine_in, const std::vector<rpvg_cluster_segment> & segments);

        // finish_later: only the copies are made here, on engine_in's context (rpvg_hip_batch_upload_begin) — finish() runs the
        // kernels behind them on another engine of the GPU, which owns the batch from then on; host_batch stays valid until then
        DeviceClusterBatch(std::shared_ptr<HipEngine> engine_in, const rpvg_cluster_batch & host_batch, const bool finish_later = false);
        void finish(std::shared_ptr<HipEngine> engine_in);

        // (between the two) the kernels of the second half queued on the uploading engine's context, behind the copies: finish()
        // then only waits for them (rpvg_hip_batch_upload_finish_queue / _wait)
        void queueFinish();

        // Adopts a batch that was built on the device (row construction, read_rows.hpp).  `offsets` carries
        // cluster_row_off / cluster_path_off only; total_read_count_in the read count of every cluster.
        DeviceClusterBatch(std::shared_ptr<HipEngine> engine_in, rpvg_hip_batch * device_batch, const rpvg_cluster_batch & offsets, const std::vector<double> & total_read_count_in);

        ~DeviceClusterBatch();

        DeviceClusterBatch(const DeviceClusterBatch &) = delete;
        DeviceClusterBatch & operator=(const DeviceClusterBatch &) = delete;

        // Hands the batch to another engine of the same GPU (a batch is made by an uploader's context and used, and freed, by an
        // estimator's: the free would otherwise wait for whatever the uploader is copying).
        void rehome(std::shared_ptr<HipEngine> engine_in) { hip_engine = engine_in; }

        const rpvg_hip_batch * handle() const { return batch; }
        const std::shared_ptr<HipEngine> & engine() const { return hip_engine; }

        uint32_t numClusters() const { return num_rows.size(); }
        uint64_t numRows(const uint32_t cluster) const { return num_rows.at(cluster); }
        uint32_t numPaths(const uint32_t cluster) const { return num_paths.at(cluster); }

        // Sum of the read counts of all rows of the cluster (exact: integers).
        double totalReadCount(const uint32_t cluster) const { return total_read_count.at(cluster); }

        // Whether the batch holds the haplotype columns of its clusters (findPathSourceGroups on the device: the batch was
        // uploaded with PathInfo::group_id and PathInfo::source_ids) — rpvg_hip_groups_build_from_sources takes it then.
        bool hasSourceColumns() const { return rpvg_hip_batch_has_source_columns(batch) != 0; }

    private:

        std::shared_ptr<HipEngine> hip_engine;
        rpvg_hip_batch * batch;

        std::vector<uint64_t> num_rows;
        std::vector<uint32_t> num_paths;
        std::vector<double> total_read_count;

        rpvg_cluster_batch unfinished_host_batch;
        bool unfinished;
        bool finish_queued;
};

}

#endif
