// The GPUs of one node behind one call: what the reference's `#pragma omp parallel for schedule(dynamic, 1)` over
// path clusters (src/main.cpp:829) becomes when the workers are GPUs.
//
// One HipEngine per GPU, one host thread per GPU in this process.  Clusters are independent units of inference
// (the reference orders them by size and hands them out dynamically, src/main.cpp:811-829): they are bin-packed onto
// the GPUs by a cost proxy, longest first, every GPU receives its shard once and runs the estimator's
// estimateBatch() on it.  No collective on the data path; the per-path abundances are gathered once at the end over
// the group's communicator (RCCL over xGMI, rpvg_hip_gather), and the TPM denominator
// (total_transcript_count, src/main.cpp:1029-1057) is one double all-reduced the same way.
#ifndef RPVG_AMD_DEVICE_GROUP_HPP
#define RPVG_AMD_DEVICE_GROUP_HPP

#include <memory>
#include <string>
#include <vector>

#include "../../include/rpvg_batch.h"
#include "hip_engine.hpp"
#include "path_cluster_estimates.hpp"

namespace rpvg_amd {

class DeviceGroup {

    public:

        // One engine per entry of `devices`.  With more than one GPU the engines' contexts form a communicator
        // (rpvg_hip_comm_init_all).  The same GPU may be listed more than once (two shards side by side on one
        // GPU, as the tests on a one-GPU box do): such a group has no communicator — RCCL wants one rank per GPU —
        // and gathers through the host memory its threads share.
        explicit DeviceGroup(const std::vector<int> & devices);
        ~DeviceGroup();

        DeviceGroup(const DeviceGroup &) = delete;
        DeviceGroup & operator=(const DeviceGroup &) = delete;

        size_t size() const { return engines.size(); }
        const std::shared_ptr<HipEngine> & engine(const size_t idx) const { return engines.at(idx); }
        bool hasCommunicator() const { return communicator; }

        // Cost proxy of a cluster: its sparse entries plus its dense matrix, rows x (paths + 1).
        static std::vector<double> clusterCosts(const rpvg_cluster_batch & batch);

        // Longest-processing-time bin packing: clusters by descending cost, each to the least loaded part (ties: the
        // lower part); every part lists its clusters in ascending order.  Deterministic.
        static std::vector<std::vector<uint32_t> > partitionClusters(const std::vector<double> & costs, const size_t num_parts);

        // Estimates of every cluster of `batch` (estimates->at(k).paths filled by the caller, as for
        // PathEstimator::estimateBatch): shards by partitionClusters, one host thread per GPU, cluster k seeded with
        // mt19937(rng_seed + k) whichever GPU it lands on (src/main.cpp:976).
        void estimateBatch(std::vector<PathClusterEstimates> * estimates, const rpvg_cluster_batch & batch, const std::string & model, const rpvg_params & params);

        // The shards of the last estimateBatch().
        const std::vector<std::vector<uint32_t> > & lastPartition() const { return partition; }

        // Final gather: the abundances of every cluster, back to back in cluster order, as every GPU's rank holds
        // them after the collective; *total_transcript_count = sum abundance / effective_length over all clusters
        // (src/main.cpp:1029-1057), all-reduced over the ranks.
        std::vector<double> gatherAbundances(const std::vector<PathClusterEstimates> & estimates, double * total_transcript_count) const;

    private:

        std::vector<std::shared_ptr<HipEngine> > engines;
        bool communicator;
        std::vector<std::vector<uint32_t> > partition;
};

}

#endif
