#include "device_group.hpp"

#include <algorithm>
#include <atomic>
#include <cassert>
#include <condition_variable>
#include <exception>
#include <mutex>
#include <numeric>
#include <random>
#include <set>
#include <thread>

#include "estimator_factory.hpp"
#include "trace.hpp"

namespace rpvg_amd {

namespace {

// The clusters `clusters` of `batch` as a batch of their own (arrays owned here).
class BatchShard {

    public:

        BatchShard(const rpvg_cluster_batch & batch, const std::vector<uint32_t> & clusters) {

            cluster_row_off.push_back(0);
            cluster_path_off.push_back(0);
            row_grp_off.push_back(0);
            grp_idx_off.push_back(0);
            path_source_off.push_back(0);

            for (auto & cluster: clusters) {

                for (uint64_t row = batch.cluster_row_off[cluster]; row < batch.cluster_row_off[cluster + 1]; ++row) {

                    row_count.push_back(batch.row_count[row]);
                    row_noise.push_back(batch.row_noise[row]);

                    for (uint64_t grp = rpvg_batch_row_group_offset(&batch, row); grp < rpvg_batch_row_group_offset(&batch, row + 1); ++grp) {

                        grp_prob.push_back(batch.grp_prob[grp]);
                        path_idx.insert(path_idx.end(), batch.path_idx + rpvg_batch_group_entry_offset(&batch, grp), batch.path_idx + rpvg_batch_group_entry_offset(&batch, grp + 1));
                        grp_idx_off.push_back(path_idx.size());
                    }

                    row_grp_off.push_back(grp_prob.size());
                }

                cluster_row_off.push_back(row_count.size());

                for (uint64_t path = batch.cluster_path_off[cluster]; path < batch.cluster_path_off[cluster + 1]; ++path) {

                    path_group_id.push_back(batch.path_group_id ? batch.path_group_id[path] : 0);
                    path_source_count.push_back(batch.path_source_count ? batch.path_source_count[path] : 1);
                    path_effective_length.push_back(batch.path_effective_length ? batch.path_effective_length[path] : 0);

                    if (batch.path_source_off) {

                        source_id.insert(source_id.end(), batch.source_id + batch.path_source_off[path], batch.source_id + batch.path_source_off[path + 1]);
                    }

                    path_source_off.push_back(source_id.size());
                }

                cluster_path_off.push_back(path_group_id.size());
            }
        }

        rpvg_cluster_batch view() const {

            rpvg_cluster_batch out = {};
            out.num_clusters = cluster_row_off.size() - 1;
            out.cluster_row_off = cluster_row_off.data();
            out.cluster_path_off = cluster_path_off.data();
            out.row_count = row_count.data();
            out.row_noise = row_noise.data();
            out.row_grp_off = row_grp_off.data();
            out.grp_prob = grp_prob.data();
            out.grp_idx_off = grp_idx_off.data();
            out.path_idx = path_idx.data();
            out.path_group_id = path_group_id.data();
            out.path_source_count = path_source_count.data();
            out.path_source_off = path_source_off.data();
            out.source_id = source_id.data();
            out.path_effective_length = path_effective_length.data();
            return out;
        }

    private:

        std::vector<uint64_t> cluster_row_off, cluster_path_off, row_grp_off, grp_idx_off, path_source_off;
        std::vector<uint32_t> row_count, path_idx, path_group_id, path_source_count, source_id;
        std::vector<double> row_noise, grp_prob, path_effective_length;
};

// a cluster's part of total_transcript_count (src/main.cpp:1029-1057)
double clusterTranscriptCount(const PathClusterEstimates & estimates) {

    double transcript_count = 0;

    if (estimates.abundances.empty()) {

        return transcript_count;
    }

    auto abundances_it = estimates.abundances.begin();

    for (auto & path_group_set: estimates.path_group_sets) {

        for (auto & path: path_group_set) {

            assert(abundances_it != estimates.abundances.end());

            const double path_effective_length = estimates.paths.at(path).effective_length;

            if (path_effective_length > 0) {

                transcript_count += (*abundances_it / path_effective_length);
            }

            ++abundances_it;
        }
    }

    return transcript_count;
}

// runs work(i) for i in [0, n) on n threads (i = 0 on the calling one) and rethrows the first failure
template <typename Work>
void onEveryDevice(const size_t n, Work work) {

    std::vector<std::exception_ptr> failures(n);
    std::vector<std::thread> threads;

    auto guarded = [&](const size_t idx) {

        try {

            work(idx);

        } catch (...) {

            failures.at(idx) = std::current_exception();
        }
    };

    for (size_t i = 1; i < n; ++i) {

        threads.emplace_back(guarded, i);
    }

    guarded(0);

    for (auto & thread: threads) {

        thread.join();
    }

    for (auto & failure: failures) {

        if (failure) {

            std::rethrow_exception(failure);
        }
    }
}

// The ranks of a collective meet here first: everything that can fail on one rank alone (allocations, copies) happens
// before, and a rank that failed says so — its peers then skip the collective instead of waiting in it for ever.
class Rendezvous {

    public:

        explicit Rendezvous(const size_t parties_in) : parties(parties_in), arrived(0), generation(0), failed(false) {}

        // Returns true when every party arrived without a failure.
        bool arrive(const bool ok) {

            std::unique_lock<std::mutex> lock(mutex);

            failed = failed || !ok;
            const size_t my_generation = generation;

            if (++arrived == parties) {

                arrived = 0;
                ++generation;
                all_here.notify_all();

            } else {

                all_here.wait(lock, [&] { return generation != my_generation; });
            }

            return !failed;
        }

    private:

        const size_t parties;
        size_t arrived;
        size_t generation;
        bool failed;

        std::mutex mutex;
        std::condition_variable all_here;
};

}

DeviceGroup::DeviceGroup(const std::vector<int> & devices) : communicator(false) {

    if (devices.empty()) {

        throw EngineError("DeviceGroup: no devices");
    }

    for (auto & device: devices) {

        engines.emplace_back(std::make_shared<HipEngine>(device));
    }

    const bool distinct = std::set<int>(devices.begin(), devices.end()).size() == devices.size();

    if (engines.size() > 1 && distinct) {

        std::vector<rpvg_hip_ctx *> contexts;

        for (auto & engine: engines) {

            contexts.emplace_back(engine->ctx());
        }

        HipEngine::check(rpvg_hip_comm_init_all(contexts.data(), contexts.size()), "rpvg_hip_comm_init_all");
        communicator = true;
    }
}

DeviceGroup::~DeviceGroup() {

    if (communicator) {

        for (auto & engine: engines) {

            rpvg_hip_comm_destroy(engine->ctx());
        }
    }
}

std::vector<double> DeviceGroup::clusterCosts(const rpvg_cluster_batch & batch) {

    std::vector<double> costs(batch.num_clusters, 0);

    for (uint32_t i = 0; i < batch.num_clusters; ++i) {

        const uint64_t first_row = batch.cluster_row_off[i];
        const uint64_t last_row = batch.cluster_row_off[i + 1];

        const double num_rows = last_row - first_row;
        const double num_paths = batch.cluster_path_off[i + 1] - batch.cluster_path_off[i];
        const double num_entries = rpvg_batch_group_entry_offset(&batch, rpvg_batch_row_group_offset(&batch, last_row)) - rpvg_batch_group_entry_offset(&batch, rpvg_batch_row_group_offset(&batch, first_row));

        costs.at(i) = num_entries + num_rows * (num_paths + 1);
    }

    return costs;
}

std::vector<std::vector<uint32_t> > DeviceGroup::partitionClusters(const std::vector<double> & costs, const size_t num_parts) {

    assert(num_parts > 0);

    std::vector<uint32_t> order(costs.size());
    std::iota(order.begin(), order.end(), 0);

    std::sort(order.begin(), order.end(), [&](const uint32_t lhs, const uint32_t rhs) {

        return costs.at(lhs) != costs.at(rhs) ? costs.at(lhs) > costs.at(rhs) : lhs < rhs;
    });

    std::vector<double> loads(num_parts, 0);
    std::vector<std::vector<uint32_t> > parts(num_parts);

    for (auto & cluster: order) {

        const size_t lightest = std::min_element(loads.begin(), loads.end()) - loads.begin();

        parts.at(lightest).emplace_back(cluster);
        loads.at(lightest) += costs.at(cluster);
    }

    for (auto & part: parts) {

        std::sort(part.begin(), part.end());
    }

    return parts;
}

void DeviceGroup::estimateBatch(std::vector<PathClusterEstimates> * estimates, const rpvg_cluster_batch & batch, const std::string & model, const rpvg_params & params) {

    assert(estimates->size() == batch.num_clusters);

    partition = partitionClusters(clusterCosts(batch), engines.size());

    // the host threads of the process are shared by the group's engines (each of which runs its own host lanes)
    const int engine_threads = std::max(2, hostThreads() / static_cast<int>(engines.size()));

    onEveryDevice(engines.size(), [&](const size_t idx) {

        // (the override is thread local and engine 0 runs on the calling thread: put the caller's value back, or the next
        // batch would divide an already divided team — 32, 16, 8, ... threads per engine, batch after batch)
        struct OverrideGuard {

            const int previous = hostThreadsOverride();
            ~OverrideGuard() { hostThreadsOverride() = previous; }

        } restore_override;

        hostThreadsOverride() = engine_threads;

        const auto & clusters = partition.at(idx);

        if (clusters.empty()) {

            return;
        }

        const BatchShard shard(batch, clusters);
        const DeviceClusterBatch device_batch(engines.at(idx), shard.view());

        std::vector<PathClusterEstimates> shard_estimates(clusters.size());
        std::vector<std::mt19937> rngs;

        for (size_t i = 0; i < clusters.size(); ++i) {

            shard_estimates.at(i).paths = estimates->at(clusters.at(i)).paths;
            rngs.emplace_back(params.rng_seed + clusters.at(i));
        }

        auto estimator = makePathEstimator(model, params, engines.at(idx));
        estimator->estimateBatch(&shard_estimates, device_batch, &rngs);

        for (size_t i = 0; i < clusters.size(); ++i) {

            estimates->at(clusters.at(i)) = std::move(shard_estimates.at(i));
        }
    });
}

std::vector<double> DeviceGroup::gatherAbundances(const std::vector<PathClusterEstimates> & estimates, double * total_transcript_count) const {

    assert(!partition.empty());

    // what every rank brings: the abundances of its clusters, shard order; and its part of the TPM denominator
    std::vector<std::vector<double> > local(engines.size());
    std::vector<uint64_t> counts(engines.size(), 0);
    std::vector<double> local_transcript_count(engines.size(), 0);

    for (size_t idx = 0; idx < engines.size(); ++idx) {

        for (auto & cluster: partition.at(idx)) {

            const auto & cluster_estimates = estimates.at(cluster);
            local.at(idx).insert(local.at(idx).end(), cluster_estimates.abundances.begin(), cluster_estimates.abundances.end());
            local_transcript_count.at(idx) += clusterTranscriptCount(cluster_estimates);
        }

        counts.at(idx) = local.at(idx).size();
    }

    const uint64_t total = std::accumulate(counts.begin(), counts.end(), uint64_t(0));
    std::vector<std::vector<double> > gathered(engines.size(), std::vector<double>(total, 0));
    std::vector<double> transcript_counts(engines.size(), 0);

    if (communicator && total > 0) {

        // (total == 0 — a model without abundances, `-i haplotypes` — has nothing to gather and a TPM denominator of zero)
        Rendezvous before_collectives(engines.size());

        onEveryDevice(engines.size(), [&](const size_t idx) {

            rpvg_hip_ctx * ctx = engines.at(idx)->ctx();
            double * device_sum = nullptr;
            std::exception_ptr failure = nullptr;

            // what can fail on this rank alone comes first
            try {

                HipEngine::check(rpvg_hip_malloc(ctx, sizeof(double), reinterpret_cast<void **>(&device_sum)), "rpvg_hip_malloc");
                HipEngine::check(rpvg_hip_memcpy_h2d(ctx, device_sum, &local_transcript_count.at(idx), sizeof(double)), "rpvg_hip_memcpy_h2d");

            } catch (...) {

                failure = std::current_exception();
            }

            const bool everybody_ready = before_collectives.arrive(!failure);

            if (everybody_ready) {

                // a rank without values still takes part (rpvg_hip_gather accepts an empty contribution)
                static const double nothing = 0;
                const double * mine = counts.at(idx) > 0 ? local.at(idx).data() : &nothing;

                HipEngine::check(rpvg_hip_gather(ctx, mine, counts.at(idx), counts.data(), gathered.at(idx).data()), "rpvg_hip_gather");
                HipEngine::check(rpvg_hip_comm_allreduce_sum_f64(ctx, device_sum, 1), "rpvg_hip_comm_allreduce_sum_f64");
                HipEngine::check(rpvg_hip_memcpy_d2h(ctx, &transcript_counts.at(idx), device_sum, sizeof(double)), "rpvg_hip_memcpy_d2h");
            }

            if (device_sum) {

                rpvg_hip_free(ctx, device_sum);
            }

            if (failure) {

                std::rethrow_exception(failure);
            }

            if (!everybody_ready) {

                throw EngineError("DeviceGroup::gatherAbundances: another rank failed before the collectives; skipped them");
            }
        });

    } else {

        // one GPU (a world of one), or shards side by side on one GPU: the threads share this host memory
        uint64_t offset = 0;

        for (size_t idx = 0; idx < engines.size(); ++idx) {

            std::copy(local.at(idx).begin(), local.at(idx).end(), gathered.at(0).begin() + offset);
            offset += counts.at(idx);
        }

        transcript_counts.at(0) = std::accumulate(local_transcript_count.begin(), local_transcript_count.end(), 0.0);
    }

    // rank order -> cluster order
    std::vector<uint64_t> cluster_offset(estimates.size() + 1, 0);

    for (size_t i = 0; i < estimates.size(); ++i) {

        cluster_offset.at(i + 1) = cluster_offset.at(i) + estimates.at(i).abundances.size();
    }

    std::vector<double> ordered(total, 0);
    uint64_t next = 0;

    for (size_t idx = 0; idx < engines.size(); ++idx) {

        for (auto & cluster: partition.at(idx)) {

            const uint64_t num_values = estimates.at(cluster).abundances.size();
            std::copy(gathered.at(0).begin() + next, gathered.at(0).begin() + next + num_values, ordered.begin() + cluster_offset.at(cluster));
            next += num_values;
        }
    }

    if (total_transcript_count) {

        *total_transcript_count = transcript_counts.at(0);
    }

    return ordered;
}

}
