// Owning flat storage of a cluster batch in the layout of include/rpvg_batch.h (harness side: the
// synthetic generator and the file readers hand batches to non-C++ callers through it).
#ifndef RPVG_AMD_FLAT_BATCH_HPP
#define RPVG_AMD_FLAT_BATCH_HPP

#include <cstdint>
#include <vector>

#include "../../include/rpvg_batch.h"
#include "path_cluster_estimates.hpp"
#include "read_path_probabilities.hpp"

namespace rpvg_amd {

struct FlatBatchStorage {

    std::vector<uint64_t> cluster_row_off, cluster_path_off, row_grp_off, grp_idx_off, path_source_off;
    std::vector<uint32_t> row_count, path_idx, path_group_id, path_source_count, source_id;
    std::vector<double> row_noise, grp_prob, path_effective_length;

    FlatBatchStorage() : cluster_row_off(1, 0), cluster_path_off(1, 0), row_grp_off(1, 0), grp_idx_off(1, 0), path_source_off(1, 0) {}

    // handles of this type cross the harness C API and may hold a derived object
    virtual ~FlatBatchStorage() {}

    void addCluster(const std::vector<PathInfo> & paths, const std::vector<ReadPathProbabilities> & rows) {

        for (auto & path: paths) {

            path_group_id.push_back(path.group_id);
            path_source_count.push_back(path.source_count);
            source_id.insert(source_id.end(), path.source_ids.begin(), path.source_ids.end());
            path_source_off.push_back(source_id.size());
            path_effective_length.push_back(path.effective_length);
        }

        cluster_path_off.push_back(path_group_id.size());

        for (auto & row: rows) {

            row_count.push_back(row.readCount());
            row_noise.push_back(row.noiseProb());

            for (auto & path_probs: row.pathProbs()) {

                grp_prob.push_back(path_probs.first);
                path_idx.insert(path_idx.end(), path_probs.second.begin(), path_probs.second.end());
                grp_idx_off.push_back(path_idx.size());
            }

            row_grp_off.push_back(grp_prob.size());
        }

        cluster_row_off.push_back(row_count.size());
    }

    void view(rpvg_cluster_batch * out) const {

        *out = rpvg_cluster_batch();
        out->num_clusters = cluster_row_off.size() - 1;
        out->cluster_row_off = cluster_row_off.data();
        out->cluster_path_off = cluster_path_off.data();
        out->row_count = row_count.data();
        out->row_noise = row_noise.data();
        out->row_grp_off = row_grp_off.data();
        out->grp_prob = grp_prob.data();
        out->grp_idx_off = grp_idx_off.data();
        out->path_idx = path_idx.data();
        out->path_group_id = path_group_id.data();
        out->path_source_count = path_source_count.data();
        out->path_source_off = path_source_off.data();
        out->source_id = source_id.data();
        out->path_effective_length = path_effective_length.data();
    }
};

}

#endif
