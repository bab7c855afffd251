#include "path_estimator.hpp"

#include <algorithm>
#include <atomic>
#include <limits>

#include <cstring>
#include <type_traits>
#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <chrono>
#include <thread>
#include <cstdio>
#include <cassert>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <functional>
#include <memory>
#include <numeric>
#include <unordered_map>

#include "numeric_utils.hpp"
#include "trace.hpp"

namespace rpvg_amd {

namespace {

const uint32_t no_member = 0xFFFFFFFFu;

// Largest number of log-likelihood requests sent to the GPU in one call.
const size_t max_requests_per_call = size_t(1) << 22;

// Owning handle of the group matrices of a list of posterior problems.
class GroupMatrices {

    public:

        GroupMatrices(const std::shared_ptr<HipEngine> & engine_in, const DeviceClusterBatch & cluster_batch, const std::vector<GroupPosteriorProblem> & problems, const bool normalise, const double prob_precision) : engine(engine_in), groups(nullptr) {

            ScopedPhase phase("posteriors: group matrices build");

            // every problem a list of single-path columns (the raw path posteriors): no list is flattened or copied
            bool all_single_paths = !problems.empty();

            for (auto & problem: problems) {

                all_single_paths = all_single_paths && problem.single_paths;
            }

            if (all_single_paths) {

                std::vector<uint32_t> clusters(problems.size());

                for (size_t i = 0; i < problems.size(); ++i) {

                    clusters[i] = problems[i].cluster;
                    assert(problems[i].numColumns() == cluster_batch.numPaths(problems[i].cluster));
                }

                HipEngine::check(rpvg_hip_groups_build_single_paths(engine->ctx(), cluster_batch.handle(), clusters.size(), clusters.data(), normalise, normalise ? prob_precision : 0.0, &groups), "rpvg_hip_groups_build_single_paths");
                return;
            }

            // flat spec arrays: offsets per problem first (cheap, serial), then every problem copies its columns (the
            // incidences of a batch are megabytes, and the GPU waits for this on the first lane)
            std::vector<uint32_t> clusters(problems.size());
            std::vector<uint64_t> group_off(problems.size() + 1, 0);
            std::vector<uint64_t> first_path(problems.size() + 1, 0);

            for (size_t i = 0; i < problems.size(); ++i) {

                group_off[i + 1] = group_off[i] + problems[i].numColumns();
                first_path[i + 1] = first_path[i] + (problems[i].single_paths ? problems[i].numColumns() : problems[i].column_path.size());
            }

            // (kept by the calling thread from call to call, as the generator words of the device sampler below)
            // (through references: inside the parallel region below the thread_local names would be the team threads' own)
            thread_local std::vector<uint64_t> kept_group_path_off;
            thread_local std::vector<uint32_t> kept_group_path;
            std::vector<uint64_t> & group_path_off = kept_group_path_off;
            std::vector<uint32_t> & group_path = kept_group_path;
            group_path_off.resize(group_off.back() + 1);
            group_path_off[0] = 0;
            group_path.resize(first_path.back());

            #pragma omp parallel for schedule(static, 1) num_threads(hostThreads())
            for (size_t i = 0; i < problems.size(); ++i) {

                const auto & problem = problems[i];
                clusters[i] = problem.cluster;

                if (problem.single_paths) {  // (among problems with lists: the implied ones written out)

                    for (uint32_t column = 1; column <= problem.numColumns(); ++column) {

                        group_path_off[group_off[i] + column] = first_path[i] + column;
                        group_path[first_path[i] + column - 1] = column - 1;
                    }

                    continue;
                }

                for (uint32_t column = 1; column <= problem.numColumns(); ++column) {

                    group_path_off[group_off[i] + column] = first_path[i] + problem.column_path_off[column];
                }

                std::copy(problem.column_path.begin(), problem.column_path.end(), group_path.begin() + first_path[i]);
            }

            rpvg_hip_group_spec spec;
            spec.num_matrices = problems.size();
            spec.cluster = clusters.data();
            spec.group_off = group_off.data();
            spec.group_path_off = group_path_off.data();
            spec.group_path = group_path.data();
            spec.normalise = normalise;
            // the reference collapses every normalised group matrix (src/path_abundance_estimator.cpp:380,443) and none
            // of the raw ones (src/path_posterior_estimator.cpp:45)
            spec.collapse_precision = normalise ? prob_precision : 0.0;
            // the diploid search on the device evaluates all pairs of a matrix at once from a second, row-major copy

            HipEngine::check(rpvg_hip_groups_build(engine->ctx(), cluster_batch.handle(), &spec, &groups), "rpvg_hip_groups_build");
        }

        // The matrices of NestedPathAbundanceEstimator::inferAbundancesCollapsedGroups for the listed clusters from the haplotype
        // columns the batch holds on the device (findPathSourceGroups ran there with the upload: no list is flattened or copied here).
        GroupMatrices(const std::shared_ptr<HipEngine> & engine_in, const DeviceClusterBatch & cluster_batch, const std::vector<uint32_t> & clusters, const bool normalise, const double prob_precision) : engine(engine_in), groups(nullptr) {

            ScopedPhase phase("posteriors: group matrices build (device columns)");
            HipEngine::check(rpvg_hip_groups_build_from_sources(engine->ctx(), cluster_batch.handle(), clusters.size(), clusters.data(), normalise, normalise ? prob_precision : 0.0, &groups), "rpvg_hip_groups_build_from_sources");
        }

        // Freeing the matrices (a stream wait, a dozen blocks back to the pool) sits between a lane's search and its EM:
        // they are kept with the lane's retired containers (dropped when the first lane's work is done, by any other lane
        // at the start of its next batch) — the holder frees them whichever way it goes.
        ~GroupMatrices() {

            ScopedPhase phase("posteriors: group matrices free");

            static const bool never_later = RPVG_AMD_EXPERIMENT_ENV("RPVG_AMD_NO_DEFERRED_TEARDOWN") != nullptr;
            rpvg_hip_ctx * lane_context = engine->ctx();  // (not the engine: its lane threads own the closures, and it owns them)
            std::shared_ptr<rpvg_hip_groups> holder(groups, [lane_context](rpvg_hip_groups * matrices) { rpvg_hip_groups_free(lane_context, matrices); });

            // (only inside a lane of runInLanes does somebody drop what is kept: a caller outside of it — the posterior
            // estimators' batches, a unit test — would pile the matrices up in device memory batch after batch, and the
            // closure would outlive the engine)
            if (!never_later && LaneScope::active()) {

                RetiredContainers::ofThisThread().keep([holder]() mutable { holder.reset(); });
            }
        }

        GroupMatrices(const GroupMatrices &) = delete;
        GroupMatrices & operator=(const GroupMatrices &) = delete;

        const rpvg_hip_groups * handle() const { return groups; }

        // Evaluates the requests in bounded chunks.
        void logLikelihoods(std::vector<double> * out, const std::vector<uint32_t> & matrix, const std::vector<uint32_t> & members, const uint32_t width, const double divisor, const bool add_rowmax) const {

            ScopedPhase phase("posteriors: loglik device calls");

            assert(members.size() == matrix.size() * width);
            out->assign(matrix.size(), 0);

            std::vector<uint8_t> flags;

            for (size_t first = 0; first < matrix.size(); first += max_requests_per_call) {

                const size_t count = std::min(max_requests_per_call, matrix.size() - first);

                if (add_rowmax) {

                    flags.assign(count, 1);
                }

                HipEngine::check(rpvg_hip_group_loglik(engine->ctx(), groups, count, matrix.data() + first, members.data() + first * width, width, divisor, add_rowmax ? flags.data() : nullptr, out->data() + first), "rpvg_hip_group_loglik");
            }
        }

        // Whole conditionals (every candidate column given the other members), request after request.
        void conditionals(std::vector<double> * out, const std::vector<uint32_t> & matrix, const std::vector<uint32_t> & others, const size_t num_values, const uint32_t width, const double divisor) const {

            ScopedPhase phase("posteriors: loglik device calls");

            assert(others.size() == matrix.size() * (width - 1));
            out->assign(num_values, 0);

            HipEngine::check(rpvg_hip_group_conditionals(engine->ctx(), groups, matrix.size(), matrix.data(), others.data(), width, divisor, out->data()), "rpvg_hip_group_conditionals");
        }

    private:

        const std::shared_ptr<HipEngine> engine;
        rpvg_hip_groups * groups;
};

// src/path_estimator.cpp:3-11
const uint32_t min_gibbs_chains = 10;
const double gibbs_chain_scaling = 0.01;
const uint32_t min_burn_it = 50;
const double burn_it_scaling = 0.025;
const uint32_t min_gibbs_it = 100;
const double gibbs_it_scaling = 0.05;

struct GroupSetHash {

    size_t operator()(const std::vector<uint32_t> & group_set) const {

        size_t seed = 0;

        for (auto & value: group_set) {

            seed ^= std::hash<uint32_t>()(value) + 0x9e3779b9 + (seed << 6) + (seed >> 2);
        }

        return seed;
    }
};

// Open-addressing table from a packed group set (<= 2 members, 32 bits each) to its index in
// order of first appearance; the per-iteration bookkeeping of the Gibbs sampler without a heap
// allocation per lookup.
class PackedSetIndex {

    public:

        // Returns the index of key, inserting it as `next_index` if absent (second = inserted).
        std::pair<uint32_t, bool> emplace(const uint64_t key, const uint32_t next_index) {

            if ((size + 1) * 2 > slots.size()) {

                grow();
            }

            size_t pos = hash(key) & (slots.size() - 1);

            while (slots[pos].second != empty_index) {

                if (slots[pos].first == key) {

                    return std::make_pair(slots[pos].second, false);
                }

                pos = (pos + 1) & (slots.size() - 1);
            }

            slots[pos] = std::make_pair(key, next_index);
            ++size;

            return std::make_pair(next_index, true);
        }

    private:

        static constexpr uint32_t empty_index = std::numeric_limits<uint32_t>::max();

        std::vector<std::pair<uint64_t, uint32_t> > slots;
        size_t size = 0;

        static size_t hash(uint64_t x) {

            x ^= x >> 33;
            x *= 0xff51afd7ed558ccdull;
            x ^= x >> 33;

            return x;
        }

        void grow() {

            std::vector<std::pair<uint64_t, uint32_t> > old_slots;
            old_slots.swap(slots);

            slots.assign(std::max<size_t>(64, old_slots.size() * 2), std::make_pair(0, empty_index));

            for (auto & slot: old_slots) {

                if (slot.second != empty_index) {

                    size_t pos = hash(slot.first) & (slots.size() - 1);

                    while (slots[pos].second != empty_index) {

                        pos = (pos + 1) & (slots.size() - 1);
                    }

                    slots[pos] = slot;
                }
            }
        }
};

// One problem's Gibbs sampler, resumable at the point where it needs a conditional
// distribution that has not been evaluated yet.  The draws follow the reference's use of the
// generator exactly (src/path_estimator.cpp:505-575): uniform_int_distribution for the start of
// every chain, one discrete_distribution draw per slot and iteration; caches only memoise.
struct GibbsSampler {

    uint32_t num_columns = 0;
    uint32_t group_size = 0;

    std::mt19937 * mt_rng = nullptr;
    std::vector<double> log_freqs;

    uint32_t num_chains = 0;
    uint32_t num_burn_its = 0;
    uint32_t num_gibbs_its = 0;

    uint32_t chain = 0;
    uint32_t iteration = 0;
    uint32_t slot = 0;
    bool chain_started = false;
    bool done = false;

    std::vector<uint32_t> cur_sampled_group_paths;

    // conditionals evaluated so far; group sizes 1 and 2 (one "other" member at most) index them
    // directly by the other member, larger groups by the sorted others
    std::vector<std::discrete_distribution<uint32_t> > conditionals;
    std::vector<uint32_t> conditional_of_other;
    std::unordered_map<std::vector<uint32_t>, uint32_t, GroupSetHash> conditional_of_others;

    PackedSetIndex packed_group_set_indices;
    std::unordered_map<std::vector<uint32_t>, uint32_t, GroupSetHash> group_set_indices;

    std::vector<uint32_t> sampled_members;
    std::vector<uint32_t> sample_counts;

    // the conditional the sampler is waiting for: the other slots' paths, in slot order
    bool waiting = false;
    std::vector<uint32_t> pending_key;
    std::vector<uint32_t> pending_others;

    static constexpr uint32_t no_conditional = std::numeric_limits<uint32_t>::max();

    void init() {

        if (group_size <= 2) {

            conditional_of_other.assign((group_size == 2) ? num_columns : 1, no_conditional);
        }
    }

    // Stores the conditional the sampler was waiting for.
    void supply(std::discrete_distribution<uint32_t> && conditional) {

        assert(waiting);

        if (group_size <= 2) {

            conditional_of_other.at((group_size == 2) ? pending_others.front() : 0) = conditionals.size();

        } else {

            conditional_of_others.emplace(pending_key, conditionals.size());
        }

        conditionals.emplace_back(std::move(conditional));
        waiting = false;
    }

    // Index of the cached conditional of `slot` given the other slots, or no_conditional
    // (then pending_* describe it).
    uint32_t findConditional() {

        if (group_size <= 2) {

            const uint32_t other = (group_size == 2) ? cur_sampled_group_paths[1 - slot] : 0;
            const uint32_t idx = conditional_of_other[other];

            if (idx == no_conditional) {

                pending_others.clear();

                if (group_size == 2) {

                    pending_others.emplace_back(other);
                }
            }

            return idx;
        }

        pending_key = cur_sampled_group_paths;
        pending_key.at(slot) = num_columns;
        std::sort(pending_key.begin(), pending_key.end());

        auto conditional_it = conditional_of_others.find(pending_key);

        if (conditional_it != conditional_of_others.end()) {

            return conditional_it->second;
        }

        pending_others.clear();

        for (uint32_t k = 0; k < group_size; ++k) {

            if (k != slot) {

                pending_others.emplace_back(cur_sampled_group_paths.at(k));
            }
        }

        return no_conditional;
    }

    void countSample() {

        if (group_size <= 2) {

            uint32_t first = cur_sampled_group_paths.front();
            uint32_t second = cur_sampled_group_paths.back();

            if (first > second) {

                std::swap(first, second);
            }

            const auto index = packed_group_set_indices.emplace((static_cast<uint64_t>(first) << 32) | second, sample_counts.size());

            if (index.second) {

                sampled_members.emplace_back(first);

                if (group_size == 2) {

                    sampled_members.emplace_back(second);
                }

                sample_counts.emplace_back(1);

            } else {

                sample_counts[index.first]++;
            }

            return;
        }

        std::vector<uint32_t> sorted_group = cur_sampled_group_paths;
        std::sort(sorted_group.begin(), sorted_group.end());

        auto group_set_indices_it = group_set_indices.emplace(sorted_group, sample_counts.size());

        if (group_set_indices_it.second) {

            sampled_members.insert(sampled_members.end(), sorted_group.begin(), sorted_group.end());
            sample_counts.emplace_back(1);

        } else {

            sample_counts.at(group_set_indices_it.first->second)++;
        }
    }

    // Runs until the sampler is done or needs a conditional that is not cached.
    void advance() {

        std::uniform_int_distribution<uint32_t> init_path_sampler(0, num_columns - 1);

        while (!done) {

            if (!chain_started) {

                cur_sampled_group_paths.clear();

                for (uint32_t i = 0; i < group_size; ++i) {

                    cur_sampled_group_paths.emplace_back(init_path_sampler(*mt_rng));
                }

                chain_started = true;
                iteration = 0;
                slot = 0;
            }

            const uint32_t conditional_idx = findConditional();

            if (conditional_idx == no_conditional) {

                waiting = true;
                return;
            }

            cur_sampled_group_paths[slot] = conditionals[conditional_idx](*mt_rng);
            ++slot;

            if (slot < group_size) {

                continue;
            }

            slot = 0;

            if (iteration >= num_burn_its) {

                countSample();
            }

            ++iteration;

            if (iteration == num_burn_its + num_gibbs_its) {

                chain_started = false;
                ++chain;

                if (chain == num_chains) {

                    done = true;
                }
            }
        }
    }
};

// Branch-and-bound state of one diploid posterior problem.
struct BoundedSearch {

    std::vector<double> log_freqs;

    // columns in descending marginal-posterior order (ties: larger index first)
    std::vector<uint32_t> order;

    // optimistic log-likelihood of any pair starting at order[i]
    std::vector<double> optimistic;

    uint32_t cursor = 0;
    double max_log_likelihood = numeric::log_zero;

    std::vector<uint32_t> candidates;

    std::vector<uint32_t> kept_members;
    std::vector<double> kept_log_likelihoods;
};

}

PathEstimator::PathEstimator(const double prob_precision_in, std::shared_ptr<HipEngine> engine_in) : prob_precision(prob_precision_in), engine(engine_in) {

    assert(engine);
}

void PathEstimator::estimateAlone(PathClusterEstimates * path_cluster_estimates, const std::vector<ReadPathProbabilities> & cluster_probs, std::mt19937 * mt_rng) {

    FlatClusterRows rows;

    if (wantsSourceColumns()) {

        rows.addCluster(cluster_probs, path_cluster_estimates->paths);  // (with PathInfo::group_id / source_ids: the device forms the haplotype columns)

    } else {

        rows.addCluster(cluster_probs, path_cluster_estimates->paths.size());
    }

    const DeviceClusterBatch cluster_batch(engine, rows.view());

    std::vector<PathClusterEstimates> batch_estimates(1);
    batch_estimates.front() = std::move(*path_cluster_estimates);

    std::vector<std::mt19937> rngs;

    if (mt_rng) {

        rngs.emplace_back(*mt_rng);
    }

    estimateBatch(&batch_estimates, cluster_batch, mt_rng ? &rngs : nullptr);

    if (mt_rng) {

        *mt_rng = rngs.front();
    }

    *path_cluster_estimates = std::move(batch_estimates.front());
}

// The reference calls estimate() once per cluster from every thread of an OpenMP team (src/main.cpp:829,976-977:
// `schedule(dynamic, 1)`).  One cluster is a poor unit of work for a GPU — a chain of dependent launches whatever its size —
// so the calls that are in flight at the same time are joined.  A caller flattens its cluster on its own thread straight into
// page-locked memory that the GPU reads where it lies (ClusterSegment: a block per thread, kept from call to call), parks the
// call and sleeps; the first to park leads: it waits for one of three device contexts of the engine to be free and for the
// batch to be worth taking — 256 clusters parked, or every thread that is inside estimate() has parked and nobody has arrived
// for 15 us, or nobody has arrived for 50 us, or 500 us have passed since the first — takes what is parked as ONE batch (the
// segments as they lie: no joined copy, no staging, no copy commands — rpvg_hip_batch_upload_segments) through estimateBatch()
// on that context (up to three batches of a large team are on the GPU at once; while all three are busy the next batch
// grows), hands every caller its estimates and its advanced generator, and wakes them.  A lone caller's batch of one leaves at
// once.  Cluster i of a batch is estimated exactly as estimateBatch() estimates it: the results do not depend on who shared
// the batch.  A caller without a generator in a batch of callers with one gets a default-seeded one for the call (models that
// draw nothing never look at it; the reference would have dereferenced its null pointer).
// RPVG_AMD_NO_COMBINER=1: every call a batch of one (estimateAlone).  RPVG_AMD_COMBINE_MAX / _QUIET_US / _LINGER_US: the bounds.
class PathEstimator::CallCombiner {

    public:

        CallCombiner() : leader_present(false), busy_slots(0), inside(0), parked_in_batches(0) {

            auto setting = [](const char * name, const long fallback) {

                const char * env = std::getenv(name);
                return env ? std::max(1L, std::atol(env)) : fallback;
            };

            max_clusters = setting("RPVG_AMD_COMBINE_MAX", 256);
            num_slots = static_cast<int>(std::min<long>(HipEngine::max_combiner_slots, setting("RPVG_AMD_COMBINE_SLOTS", 3)));
            quiet = std::chrono::microseconds(setting("RPVG_AMD_COMBINE_QUIET_US", 50));
            all_parked_quiet = std::chrono::microseconds(setting("RPVG_AMD_COMBINE_ALL_PARKED_US", 15));
            linger = std::chrono::microseconds(setting("RPVG_AMD_COMBINE_LINGER_US", 500));
        }

        void call(PathEstimator * owner, PathClusterEstimates * path_cluster_estimates, const std::vector<ReadPathProbabilities> & cluster_probs, std::mt19937 * mt_rng) {

            struct Inside {

                std::atomic<int> & count;
                explicit Inside(std::atomic<int> & count_in) : count(count_in) { ++count; }
                ~Inside() { --count; }

            } here(inside);

            // (the calling thread's block: a thread has one call in flight)
            thread_local ClusterSegment segment;

            {
                ScopedPhase phase("combiner: flatten the cluster (callers, summed)");

                // (the haplotype columns on this thread, as the reference forms them per cluster: the batch's upload then waits for
                // nothing but its own kernel — RPVG_AMD_DEVICE_SOURCE_COLUMNS=1: the ids travel and the device forms them, A/B)
                static const bool device_columns = std::getenv("RPVG_AMD_DEVICE_SOURCE_COLUMNS") != nullptr;
                thread_local GroupPosteriorProblem columns;

                columns.column_path_off.clear();
                columns.column_path.clear();
                columns.column_counts.clear();

                if (!device_columns && owner->sourceColumnsOf(&columns, path_cluster_estimates->paths)) {

                    const ClusterSegment::Columns formed = {columns.numColumns(), columns.column_counts.data(), columns.column_path_off.data(), columns.column_path.data()};
                    segment.flatten(cluster_probs, path_cluster_estimates->paths, true, &formed);

                } else {

                    segment.flatten(cluster_probs, path_cluster_estimates->paths, owner->wantsSourceColumns());
                }
            }

            Parked me;
            me.segment = &segment.view();
            me.estimates = path_cluster_estimates;
            me.rng = mt_rng;

            std::unique_lock<std::mutex> lock(mutex);

            const auto now = std::chrono::steady_clock::now();

            if (staging.empty()) {

                first_arrival = now;
            }

            last_arrival = now;
            staging.emplace_back(&me);

            if (leader_present) {

                arrived.notify_all();
                lock.unlock();

                // (no mutex on the way out: sixty-odd callers woken at once would queue for it one by one — every caller watches its
                // own flag and sleeps on the combiner's generation word, which a finished batch bumps once for all of them)
                while (!me.done.load(std::memory_order_acquire)) {

                    const uint32_t seen = generation.load(std::memory_order_acquire);

                    if (me.done.load(std::memory_order_acquire)) {

                        break;
                    }

                    syscall(SYS_futex, reinterpret_cast<uint32_t *>(&generation), FUTEX_WAIT_PRIVATE, seen, nullptr, nullptr, 0);
                }

            } else {

                leader_present = true;

                // a batch leaves when a device context is free for it — while all three are busy the parked clusters simply pile up,
                // so the batches grow with the load — and, a context being free, when it is full, when waiting can bring nobody
                // (every thread inside estimate() is parked, here or in a batch that is running: at once if that is this thread
                // alone, after a short pause otherwise — threads that a finished batch has just released are on their way) or
                // when nobody has arrived for a while
                while (true) {

                    const bool slot_free = busy_slots != (1 << num_slots) - 1;
                    const bool all_parked = static_cast<size_t>(inside.load()) <= staging.size() + parked_in_batches;
                    const auto deadline = std::min(first_arrival + linger, last_arrival + ((all_parked && parked_in_batches == 0) ? all_parked_quiet : quiet));

                    if (slot_free && (staging.size() >= static_cast<size_t>(max_clusters) || (all_parked && inside.load() == 1) || std::chrono::steady_clock::now() >= deadline)) {

                        break;
                    }

                    if (slot_free) {

                        arrived.wait_until(lock, deadline);

                    } else {

                        arrived.wait(lock);  // (a freed context notifies too)
                    }
                }

                std::vector<Parked *> batch;
                batch.swap(staging);
                leader_present = false;  // (whoever parks next leads the next batch, while this one runs)
                parked_in_batches += batch.size();

                // one of the combiner's device contexts (HipEngine::combinerLane)
                int slot = 0;

                while (busy_slots & (1 << slot)) {

                    ++slot;
                }

                busy_slots |= 1 << slot;
                lock.unlock();

                std::exception_ptr error = nullptr;

                try {

                    flush(owner, batch, slot);

                } catch (...) {

                    error = std::current_exception();
                }

                lock.lock();
                busy_slots &= ~(1 << slot);
                parked_in_batches -= batch.size();

                lock.unlock();
                arrived.notify_all();

                for (auto & parked: batch) {  // (a caller is gone, and its Parked with it, as soon as it sees its flag)

                    parked->error = error;
                    parked->done.store(true, std::memory_order_release);
                }

                generation.fetch_add(1, std::memory_order_release);
                syscall(SYS_futex, reinterpret_cast<uint32_t *>(&generation), FUTEX_WAKE_PRIVATE, std::numeric_limits<int>::max(), nullptr, nullptr, 0);
            }

            if (me.error) {

                std::rethrow_exception(me.error);
            }
        }

    private:

        struct Parked {

            const rpvg_cluster_segment * segment = nullptr;
            PathClusterEstimates * estimates = nullptr;
            std::mt19937 * rng = nullptr;
            std::atomic<bool> done{false};
            std::exception_ptr error = nullptr;
        };

        static void flush(PathEstimator * owner, const std::vector<Parked *> & batch, const int slot) {

            ScopedPhase flush_phase("combiner: batches (leaders, summed)");
            PhaseTrace::add("combiner: number of batches", 1e-3);
            PhaseTrace::add("combiner: clusters in batches", 1e-3 * batch.size());

            std::unique_ptr<ScopedPhase> phase(new ScopedPhase("combiner: upload"));

            std::vector<rpvg_cluster_segment> segments;
            segments.reserve(batch.size());

            bool any_has_generator = false;

            for (auto & parked: batch) {

                segments.emplace_back(*parked->segment);
                any_has_generator = any_has_generator || parked->rng;
            }

            // the engine's context for this thread while the batch runs: the slot's (made on first use; a thread that is not in lane 0
            // runs its batch on its lane's context, whole: PathEstimator::runInLanes)
            struct LaneGuard {

                int previous;
                explicit LaneGuard(const int lane) : previous(HipEngine::currentLane()) { HipEngine::currentLane() = lane; }
                ~LaneGuard() { HipEngine::currentLane() = previous; }

            } lane_guard(owner->engine->combinerLane(slot));

            // (up to sixty-four callers sleep behind this thread: its waits for the GPU query instead of napping — three leaders at the most)
            struct SpinGuard {

                SpinGuard() { rpvg_hip_thread_wait_spin_us(400); }
                ~SpinGuard() { rpvg_hip_thread_wait_spin_us(20); }

            } spin_guard;

            const DeviceClusterBatch cluster_batch(owner->engine, segments);

            phase.reset(new ScopedPhase("combiner: containers in"));

            std::vector<PathClusterEstimates> batch_estimates(batch.size());
            std::vector<std::mt19937> rngs;

            for (size_t i = 0; i < batch.size(); ++i) {

                batch_estimates[i] = std::move(*batch[i]->estimates);

                if (any_has_generator) {

                    rngs.emplace_back(batch[i]->rng ? *batch[i]->rng : std::mt19937());
                }
            }

            phase.reset(new ScopedPhase("combiner: estimateBatch"));
            std::exception_ptr error = nullptr;

            try {

                owner->estimateBatch(&batch_estimates, cluster_batch, any_has_generator ? &rngs : nullptr);

            } catch (...) {

                error = std::current_exception();
            }

            phase.reset(new ScopedPhase("combiner: containers out"));

            for (size_t i = 0; i < batch.size(); ++i) {  // (the callers' containers go back whatever happened: their paths are in them)

                *batch[i]->estimates = std::move(batch_estimates[i]);

                if (batch[i]->rng && !error) {

                    *batch[i]->rng = rngs[i];
                }
            }

            if (error) {

                std::rethrow_exception(error);
            }
        }

        std::mutex mutex;
        std::condition_variable arrived;
        std::atomic<uint32_t> generation{0};  // bumped by every finished batch: what the parked callers sleep on (futex)

        std::vector<Parked *> staging;
        std::chrono::steady_clock::time_point first_arrival, last_arrival;
        bool leader_present;
        int busy_slots;
        std::atomic<int> inside;   // threads inside call(): flattening, parked, or leading
        size_t parked_in_batches;  // of them: in batches that are running (under `mutex`)

        long max_clusters;
        int num_slots;
        std::chrono::microseconds quiet, all_parked_quiet, linger;
};

void PathEstimator::estimate(PathClusterEstimates * path_cluster_estimates, const std::vector<ReadPathProbabilities> & cluster_probs, std::mt19937 * mt_rng) {

    static const bool alone = std::getenv("RPVG_AMD_NO_COMBINER") != nullptr;

    if (alone) {

        estimateAlone(path_cluster_estimates, cluster_probs, mt_rng);
        return;
    }

    std::shared_ptr<CallCombiner> combiner;

    {
        std::lock_guard<std::mutex> lock(call_combiner_mutex);

        if (!call_combiner) {

            call_combiner = std::make_shared<CallCombiner>();
        }

        combiner = call_combiner;
    }

    combiner->call(this, path_cluster_estimates, cluster_probs, mt_rng);
}

// Host lanes over the GPU (pipeline_lanes.hpp): the clusters arrive ordered by size, so dealing them out round
// robin gives parts of equal cost; one lane's host phases run while the others wait for the device, and the
// lanes' kernels (separate device contexts) overlap each other's tails.
void PathEstimator::runInLanes(const std::vector<uint32_t> & clusters, const std::function<void(const std::vector<uint32_t> &, const std::function<void()> &)> & work) const {

    const int num_lanes = engine->hostLanes();  // (RPVG_AMD_LANES, default 2; the engines of a BatchPipeline: 1)

    if (num_lanes == 1 || clusters.size() < 64 || HipEngine::currentLane() != 0) {

        const bool outermost = !LaneScope::active();

        try {

            LaneScope scope;
            work(clusters, []() {});

        } catch (...) {

            if (outermost) {

                RetiredContainers::ofThisThread().dropAll();
            }

            throw;
        }

        if (outermost) {

            RetiredContainers::ofThisThread().dropAll();  // what the work kept for later
        }

        return;
    }

    LaneStagger stagger(num_lanes);

    std::vector<std::vector<uint32_t> > lane_clusters(num_lanes);

    // shares of the lanes (RPVG_AMD_LANE_SHARES="40,60": A/B knob; default equal): the clusters, ordered by size, go
    // one by one to the lane that is furthest behind its share
    std::vector<double> share(num_lanes, 1.0);

    if (const char * env = RPVG_AMD_EXPERIMENT_ENV("RPVG_AMD_LANE_SHARES")) {

        const char * cursor = env;

        for (int lane = 0; lane < num_lanes && *cursor; ++lane) {

            char * end = nullptr;
            const double value = std::strtod(cursor, &end);

            if (end == cursor) {

                break;
            }

            share[lane] = std::max(1e-3, value);
            cursor = (*end == ',') ? end + 1 : end;
        }
    }

    std::vector<double> dealt(num_lanes, 0.0);

    for (size_t i = 0; i < clusters.size(); ++i) {

        int lane = 0;

        for (int other = 1; other < num_lanes; ++other) {

            if ((dealt[other] + 1) / share[other] < (dealt[lane] + 1) / share[lane]) {

                lane = other;
            }
        }

        dealt[lane] += 1;
        lane_clusters[lane].emplace_back(clusters[i]);
    }

    // Every lane may use the rank's whole team (hostThreads(): the host's threads divided by its ranks).  The lanes are
    // staggered, so their parallel regions mostly alternate; when they do coincide the teams share the cores, which
    // costs what two half-size teams would have cost all the time (idle workers sleep: OMP_WAIT_POLICY=passive).
    const int lane_threads = hostThreads();
    const int outer_threads = hostThreadsOverride();

    for (int lane = 1; lane < num_lanes; ++lane) {

        engine->lane(lane).submit([&, lane]() {

            hostThreadsOverride() = lane_threads;
            HipEngine::currentLane() = lane;
            LaneScope scope;

            // the lane before works through its host prologue: time to free what the previous batch left behind
            RetiredContainers::ofThisThread().dropAll();

            stagger.waitTurn(lane);

#ifdef RPVG_AMD_EXPERIMENTS
            // A/B knob RPVG_AMD_LANE_DELAY_US: the lane starts this much later still
            static const long lane_delay_us = std::getenv("RPVG_AMD_LANE_DELAY_US") ? std::atol(std::getenv("RPVG_AMD_LANE_DELAY_US")) : 0;

            if (lane_delay_us > 0) {

                std::this_thread::sleep_for(std::chrono::microseconds(lane_delay_us));
            }
#endif

            try {

                work(lane_clusters[lane], [&stagger, lane]() { stagger.passBaton(lane); });

            } catch (...) {

                stagger.passBaton(lane);
                throw;
            }

            stagger.passBaton(lane);
        });
    }

    hostThreadsOverride() = lane_threads;
    std::exception_ptr first_error = nullptr;

    try {

        LaneScope scope;
        work(lane_clusters[0], [&stagger]() { stagger.passBaton(0); });

    } catch (...) {

        first_error = std::current_exception();
    }

    stagger.passBaton(0);

    // the first lane is done before the others: time for the containers it kept between its device stages
    RetiredContainers::ofThisThread().dropAll();

    hostThreadsOverride() = outer_threads;

    for (int lane = 1; lane < num_lanes; ++lane) {

        try {

            engine->lane(lane).wait();

        } catch (...) {

            if (!first_error) {

                first_error = std::current_exception();
            }
        }
    }

    if (first_error) {

        std::rethrow_exception(first_error);
    }
}

void PathEstimator::estimateBatchSeeded(std::vector<PathClusterEstimates> * path_cluster_estimates, const DeviceClusterBatch & cluster_batch, const uint32_t rng_seed) {

    if (!usesRandomNumbers()) {

        estimateBatch(path_cluster_estimates, cluster_batch, nullptr);
        return;
    }

    // src/main.cpp:976 — cluster i draws from mt19937(rng_seed + i); seeding 624 words per generator adds up over a batch
    // (copies of one generator, not n default constructions: a default-constructed mt19937 seeds its 624 words too — 5 000 of
    // them one after the other were 10 ms of a configs[4] batch before the team's seeding below even started)
    static const std::mt19937 unseeded;
    std::vector<std::mt19937> rngs(cluster_batch.numClusters(), unseeded);

    #pragma omp parallel for schedule(static) num_threads(hostThreads())
    for (uint32_t i = 0; i < cluster_batch.numClusters(); ++i) {

        rngs[i].seed(rng_seed + i);
    }

    estimateBatch(path_cluster_estimates, cluster_batch, &rngs);
}

std::vector<double> PathEstimator::calcPathLogFrequences(const std::vector<uint32_t> & path_counts) {

    const uint32_t count_sum = std::accumulate(path_counts.begin(), path_counts.end(), 0u);
    assert(count_sum > 0);

    std::vector<double> path_log_freqs(path_counts.size());

    // (most columns of a cluster carry one of a few small counts: the logarithm of a count is taken once — the same
    // division and the same logarithm as for every column by itself)
    constexpr uint32_t memo_size = 64;
    double memo[memo_size];
    uint64_t memo_set = 0;

    for (size_t i = 0; i < path_counts.size(); ++i) {

        const uint32_t count = path_counts[i];
        assert(count > 0);

        if (count < memo_size) {

            if (!((memo_set >> count) & 1u)) {

                memo[count] = std::log(count / static_cast<double>(count_sum));
                memo_set |= uint64_t(1) << count;
            }

            path_log_freqs[i] = memo[count];

        } else {

            path_log_freqs[i] = std::log(count / static_cast<double>(count_sum));
        }
    }

    return path_log_freqs;
}

void PathEstimator::calculatePathGroupPosteriorsFull(std::vector<GroupPosteriors> * group_posteriors, const DeviceClusterBatch & cluster_batch, const std::vector<GroupPosteriorProblem> & problems, const uint32_t group_size, const bool normalise) const {

    assert(group_size > 0);

    if (group_size > 4) {

        throw EngineError("calculatePathGroupPosteriorsFull: group sizes above 4 are not supported by the GPU log-likelihood kernel");
    }

    group_posteriors->assign(problems.size(), GroupPosteriors());

    if (problems.empty()) {

        return;
    }

    const GroupMatrices matrices(engine, cluster_batch, problems, normalise, prob_precision);

    // every multiset of every problem is one request
    std::vector<uint32_t> request_matrix;
    std::vector<uint32_t> request_members;
    std::vector<size_t> first_request(problems.size() + 1, 0);

    for (size_t i = 0; i < problems.size(); ++i) {

        PathClusterEstimates enumerator;
        enumerator.generateGroups(problems.at(i).numColumns(), group_size);

        auto & result = group_posteriors->at(i);
        result.group_size = group_size;
        result.members.reserve(enumerator.path_group_sets.size() * group_size);

        for (auto & group_set: enumerator.path_group_sets) {

            request_matrix.emplace_back(i);
            result.members.insert(result.members.end(), group_set.begin(), group_set.end());
        }

        request_members.insert(request_members.end(), result.members.begin(), result.members.end());
        first_request.at(i + 1) = request_matrix.size();
    }

    std::vector<double> log_likelihoods;
    matrices.logLikelihoods(&log_likelihoods, request_matrix, request_members, group_size, group_size, false);

    #pragma omp parallel for schedule(dynamic, 8) num_threads(hostThreads())
    for (size_t i = 0; i < problems.size(); ++i) {

        const auto path_log_freqs = calcPathLogFrequences(problems.at(i).column_counts);
        assert(path_log_freqs.size() == problems.at(i).numColumns());

        auto & result = group_posteriors->at(i);

        const size_t num_sets = first_request.at(i + 1) - first_request.at(i);
        result.posteriors.assign(num_sets, 0);

        double sum_log_posterior = numeric::log_zero;

        for (size_t j = 0; j < num_sets; ++j) {

            double log_posterior = log_likelihoods.at(first_request.at(i) + j);

            const std::vector<uint32_t> group_set(result.set(j), result.set(j) + group_size);

            for (auto & path_idx: group_set) {

                log_posterior += path_log_freqs.at(path_idx);
            }

            log_posterior += std::log(numeric::numPermutations(group_set));

            result.posteriors.at(j) = log_posterior;
            sum_log_posterior = numeric::add_log(sum_log_posterior, log_posterior);
        }

        for (auto & posterior: result.posteriors) {

            posterior = std::exp(posterior - sum_log_posterior);
        }
    }
}

void PathEstimator::calculatePathGroupPosteriorsBounded(std::vector<GroupPosteriors> * group_posteriors, const DeviceClusterBatch & cluster_batch, const std::vector<GroupPosteriorProblem> & problems, const uint32_t group_size, const double min_rel_likelihood, const bool normalise) const {

    assert(group_size == 2);

    if (std::getenv("RPVG_AMD_HOST_BOUNDED")) {

        calculatePathGroupPosteriorsBoundedHostDriven(group_posteriors, cluster_batch, problems, group_size, min_rel_likelihood, normalise);
        return;
    }

    group_posteriors->assign(problems.size(), GroupPosteriors());

    if (problems.empty()) {

        return;
    }

    ScopedPhase whole_phase("posteriors: bounded total incl. teardown");

    const GroupMatrices matrices(engine, cluster_batch, problems, normalise, prob_precision);

    std::vector<uint32_t> column_counts;

    for (auto & problem: problems) {

        column_counts.insert(column_counts.end(), problem.column_counts.begin(), problem.column_counts.end());
    }

    rpvg_hip_pair_posteriors * pair_posteriors = nullptr;

    {
        ScopedPhase phase("posteriors: on-device bounded search");
        HipEngine::check(rpvg_hip_bounded_pair_posteriors(engine->ctx(), matrices.handle(), column_counts.data(), min_rel_likelihood, &pair_posteriors), "rpvg_hip_bounded_pair_posteriors");
    }

    ScopedPhase phase("posteriors: unpack pairs");

    rpvg_hip_pair_posteriors_view view;
    HipEngine::check(rpvg_hip_pair_posteriors_get(pair_posteriors, &view), "rpvg_hip_pair_posteriors_get");

    #pragma omp parallel for schedule(static, 1) num_threads(hostThreads())
    for (size_t i = 0; i < problems.size(); ++i) {

        auto & result = group_posteriors->at(i);
        const uint64_t num_pairs = view.pair_off[i + 1] - view.pair_off[i];

        result.group_size = 2;
        result.members.reserve(num_pairs * 2);
        result.posteriors.assign(view.posterior + view.pair_off[i], view.posterior + view.pair_off[i + 1]);

        for (uint64_t j = view.pair_off[i]; j < view.pair_off[i + 1]; ++j) {

            result.members.emplace_back(view.first[j]);
            result.members.emplace_back(view.second[j]);
        }
    }

    ScopedPhase free_phase("posteriors: pair result free");
    rpvg_hip_pair_posteriors_free(pair_posteriors);
}

PathEstimator::SubsetEmResult::~SubsetEmResult() {

    rpvg_hip_subset_em_free(result);
}

bool PathEstimator::nestedSubsetAbundances(SubsetEmResult * result, const DeviceClusterBatch & cluster_batch, const std::vector<GroupPosteriorProblem> & problems, const double min_rel_likelihood, const double min_hap_prob, const uint32_t max_em_its, const double max_rel_em_conv) const {

    assert(!result->result);

    if (problems.empty() || std::getenv("RPVG_AMD_HOST_BOUNDED")) {

        return false;
    }

    ScopedPhase whole_phase("nested: search + subsets + EM on the device");

    const GroupMatrices matrices(engine, cluster_batch, problems, true, prob_precision);

    std::vector<uint32_t> column_counts;

    for (auto & problem: problems) {

        column_counts.insert(column_counts.end(), problem.column_counts.begin(), problem.column_counts.end());
    }

    const int status = rpvg_hip_nested_subset_em(engine->ctx(), cluster_batch.handle(), matrices.handle(), column_counts.data(), min_rel_likelihood, min_hap_prob, max_em_its, max_rel_em_conv, prob_precision, &result->result);

    if (status == RPVG_HIP_ERR_UNSUPPORTED) {

        return false;
    }

    HipEngine::check(status, "rpvg_hip_nested_subset_em");
    HipEngine::check(rpvg_hip_subset_em_get(result->result, &result->view), "rpvg_hip_subset_em_get");

    return true;
}

bool PathEstimator::nestedSubsetAbundances(SubsetEmResult * result, const DeviceClusterBatch & cluster_batch, const std::vector<uint32_t> & clusters, const double min_rel_likelihood, const double min_hap_prob, const uint32_t max_em_its, const double max_rel_em_conv) const {

    assert(!result->result);
    assert(cluster_batch.hasSourceColumns());

    if (clusters.empty() || std::getenv("RPVG_AMD_HOST_BOUNDED")) {

        return false;
    }

    ScopedPhase whole_phase("nested: matrices + search + subsets + EM + merge on the device");

    const GroupMatrices matrices(engine, cluster_batch, clusters, true, prob_precision);

    const int status = rpvg_hip_nested_subset_em(engine->ctx(), cluster_batch.handle(), matrices.handle(), nullptr, min_rel_likelihood, min_hap_prob, max_em_its, max_rel_em_conv, prob_precision, &result->result);

    if (status == RPVG_HIP_ERR_UNSUPPORTED) {

        return false;
    }

    HipEngine::check(status, "rpvg_hip_nested_subset_em");
    HipEngine::check(rpvg_hip_subset_em_get(result->result, &result->view), "rpvg_hip_subset_em_get");

    return true;
}

namespace {

// A seed sequence that hands out the 624 words it was given: std::mt19937::seed(sequence) copies them into the
// generator's state and regenerates on the next call ([rand.eng.mers]) — the standard's way of putting a generator
// where rpvg_hip_group_gibbs left it, where discard() would generate every word the chains took once more on the host
// (40 M words per configs[4] batch: 60 ms of CPU time).
struct GeneratorStateWords {

    typedef uint32_t result_type;

    const uint32_t * words;

    explicit GeneratorStateWords(const uint32_t * words_in) : words(words_in) {}

    template <typename Iterator>
    void generate(Iterator first, Iterator last) {

        for (size_t i = 0; first != last; ++first, ++i) {

            *first = words[i];
        }
    }
};

// What rpvg_hip_group_gibbs wants of a generator — its next 624 outputs — and what it hands back — the state to go on from —
// without a copy of the generator, 624 calls and a seed(): straight from and to the engine's state array.  libstdc++ keeps a
// std::mt19937 as its 624 state words (in uint_fast32_t) followed by the position of the next one, and nothing else; that is
// not something the standard promises, so the first use draws the same words both ways from a probe generator and the raw
// route is only taken when they agree (and the sizes match at compile time) — otherwise, and with RPVG_AMD_PORTABLE_GENERATORS=1,
// the portable route below it.  5 000 generators per configs[4] batch: 15 ms of CPU time per batch against 2.
struct GeneratorLayout {

    std::mt19937::result_type x[std::mt19937::state_size];
    size_t p;
};

inline uint32_t temperWord(uint32_t y) {

    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

inline uint32_t twistWords(const uint32_t upper, const uint32_t lower) {

    const uint32_t y = (upper & 0x80000000u) | (lower & 0x7fffffffu);
    return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

// the next 624 outputs of a generator whose state words are x and whose next word is x[p] (p == 624: none left)
inline void nextOutputsFromState(const std::mt19937::result_type * x, const size_t p, uint32_t * out) {

    constexpr size_t n = std::mt19937::state_size, m = 397;

    for (size_t i = p; i < n; ++i) {

        out[i - p] = temperWord(static_cast<uint32_t>(x[i]));
    }

    // ... and the first p words of the next state: word k from the old words k, k + 1 and k + 397 — for k >= 227 the latter is
    // the NEW word k - 227 (the generator refills its array in place)
    uint32_t fresh[n];

    for (size_t k = 0; k < p && k < n - m; ++k) {

        fresh[k] = static_cast<uint32_t>(x[k + m]) ^ twistWords(static_cast<uint32_t>(x[k]), static_cast<uint32_t>(x[k + 1]));
    }

    for (size_t k = n - m; k < p; ++k) {

        const uint32_t lower = (k + 1 < n) ? static_cast<uint32_t>(x[k + 1]) : fresh[0];
        fresh[k] = fresh[k - (n - m)] ^ twistWords(static_cast<uint32_t>(x[k]), lower);
    }

    for (size_t k = 0; k < p; ++k) {

        out[n - p + k] = temperWord(fresh[k]);
    }
}

bool rawGeneratorAccess() {

    static const bool usable = []() {

        if (sizeof(GeneratorLayout) != sizeof(std::mt19937) || !std::is_trivially_copyable<std::mt19937>::value || std::getenv("RPVG_AMD_PORTABLE_GENERATORS")) {

            return false;
        }

        for (const size_t advance: {0u, 1u, 227u, 623u, 624u, 1000u}) {

            std::mt19937 probe(20240229u + advance);
            probe.discard(advance);

            GeneratorLayout layout;
            std::memcpy(&layout, &probe, sizeof(layout));

            if (layout.p > std::mt19937::state_size) {

                return false;
            }

            uint32_t raw[std::mt19937::state_size];
            nextOutputsFromState(layout.x, layout.p, raw);

            for (size_t w = 0; w < std::mt19937::state_size; ++w) {

                if (raw[w] != probe()) {

                    return false;
                }
            }
        }

        return true;
    }();

    return usable;
}

// the next 624 outputs of a generator, which stays where it is
void nextGeneratorOutputs(const std::mt19937 & generator, uint32_t * out) {

    if (rawGeneratorAccess()) {

        GeneratorLayout layout;  // (a copy of the bytes: 5 KB, and nothing the compiler's view of the generator's type could object to)
        std::memcpy(&layout, &generator, sizeof(layout));
        nextOutputsFromState(layout.x, layout.p, out);

    } else {

        std::mt19937 ahead = generator;

        for (size_t w = 0; w < std::mt19937::state_size; ++w) {

            out[w] = ahead();
        }
    }
}

// a generator whose last 624 outputs were the tempered `words`: it goes on behind them
void setGeneratorBehind(std::mt19937 * generator, const uint32_t * words) {

    if (rawGeneratorAccess()) {

        GeneratorLayout layout;

        for (size_t w = 0; w < std::mt19937::state_size; ++w) {

            layout.x[w] = words[w];
        }

        layout.p = std::mt19937::state_size;
        std::memcpy(static_cast<void *>(generator), &layout, sizeof(layout));

    } else {

        GeneratorStateWords state_words(words);
        generator->seed(state_words);
    }
}

// The sampler of src/path_estimator.cpp:475-589 in one device call: the generators' next 624 outputs go in (their
// untempered values are the generators' states), the sampled sets come out in the reference's order, and every generator
// is moved past the words its chains took.  False when the device call does not take the input (the distributions of the
// batch outgrow the memory set aside for them): the host-driven sampler below does.
bool estimatePathGroupPosteriorsGibbsOnDevice(std::vector<GroupPosteriors> * group_posteriors, const std::shared_ptr<HipEngine> & engine, const rpvg_hip_groups * groups, const std::vector<GroupPosteriorProblem> & problems, const uint32_t group_size, const std::vector<std::mt19937 *> & rngs, std::vector<double> (*log_frequencies)(const std::vector<uint32_t> &)) {

    ScopedPhase phase("gibbs: device sampler");

    const size_t num_problems = problems.size();

    // problems by generator, in the order the reference's loop reaches them
    std::vector<std::mt19937 *> generators;
    std::vector<uint32_t> generator_of_problem(num_problems);
    std::unordered_map<std::mt19937 *, uint32_t> generator_index;

    for (size_t i = 0; i < num_problems; ++i) {

        auto generator_index_it = generator_index.emplace(rngs.at(i), generators.size());

        if (generator_index_it.second) {

            generators.emplace_back(rngs.at(i));
        }

        generator_of_problem[i] = generator_index_it.first->second;
    }

    std::vector<uint32_t> generator_problem_off(generators.size() + 1, 0);

    for (size_t i = 0; i < num_problems; ++i) {

        generator_problem_off[generator_of_problem[i] + 1]++;
    }

    std::partial_sum(generator_problem_off.begin(), generator_problem_off.end(), generator_problem_off.begin());

    std::vector<uint32_t> generator_problem(num_problems);
    {
        std::vector<uint32_t> cursor(generator_problem_off.begin(), generator_problem_off.end() - 1);

        for (size_t i = 0; i < num_problems; ++i) {

            generator_problem[cursor[generator_of_problem[i]]++] = i;
        }
    }

    std::vector<uint32_t> matrix(num_problems);
    std::vector<uint32_t> num_chains(num_problems);
    std::vector<uint32_t> num_burn_its(num_problems);
    std::vector<uint32_t> num_gibbs_its(num_problems);
    std::vector<uint64_t> column_off(num_problems + 1, 0);

    for (size_t i = 0; i < num_problems; ++i) {

        column_off[i + 1] = column_off[i] + problems[i].numColumns();
    }

    // (megabytes per call — 12.5 MB of generator words for a configs[4] batch — kept by the calling lane thread from call to call:
    // fresh from the allocator every call they are mapped, faulted in page by page and unmapped again, 5 ms of system time)
    // (through references: inside the parallel regions below the thread_local names would be the team threads' own)
    thread_local std::vector<double> kept_log_freqs;
    thread_local std::vector<uint32_t> kept_generator_words;
    std::vector<double> & log_freqs = kept_log_freqs;
    std::vector<uint32_t> & generator_words = kept_generator_words;
    log_freqs.resize(column_off.back());
    generator_words.resize(generators.size() * std::mt19937::state_size);

    ScopedPhase words_phase("gibbs: chain lengths, log frequencies, generator words");

    #pragma omp parallel num_threads(shortLoopThreads())
    {
        #pragma omp for schedule(dynamic, 16) nowait
        for (size_t i = 0; i < num_problems; ++i) {

            const uint32_t num_columns = problems[i].numColumns();

            matrix[i] = i;

            // src/path_estimator.cpp:501-503
            num_chains[i] = min_gibbs_chains + std::round(gibbs_chain_scaling * group_size * num_columns);
            num_burn_its[i] = min_burn_it + std::round(burn_it_scaling * group_size * num_columns);
            num_gibbs_its[i] = min_gibbs_it + std::round(gibbs_it_scaling * group_size * num_columns);

            const auto problem_log_freqs = log_frequencies(problems[i].column_counts);
            std::copy(problem_log_freqs.begin(), problem_log_freqs.end(), log_freqs.begin() + column_off[i]);
        }

        #pragma omp for schedule(dynamic, 16)
        for (size_t g = 0; g < generators.size(); ++g) {

            nextGeneratorOutputs(*generators[g], generator_words.data() + g * std::mt19937::state_size);
        }
    }

    words_phase.stop();

    rpvg_hip_gibbs_spec spec;
    spec.num_problems = num_problems;
    spec.group_size = group_size;
    spec.matrix = matrix.data();
    spec.num_chains = num_chains.data();
    spec.num_burn_its = num_burn_its.data();
    spec.num_gibbs_its = num_gibbs_its.data();
    spec.log_freq = log_freqs.data();
    spec.num_generators = generators.size();
    spec.generator_problem_off = generator_problem_off.data();
    spec.generator_problem = generator_problem.data();
    spec.generator_words = generator_words.data();

    ScopedPhase call_phase("gibbs: rpvg_hip_group_gibbs");

    rpvg_hip_gibbs_sets * sets = nullptr;
    const int status = rpvg_hip_group_gibbs(engine->ctx(), groups, &spec, &sets);

    call_phase.stop();

    if (status == RPVG_HIP_ERR_UNSUPPORTED) {

        return false;
    }

    HipEngine::check(status, "rpvg_hip_group_gibbs");
    std::shared_ptr<rpvg_hip_gibbs_sets> sets_holder(sets, rpvg_hip_gibbs_sets_free);

    rpvg_hip_gibbs_sets_view view;
    HipEngine::check(rpvg_hip_gibbs_sets_get(sets, &view), "rpvg_hip_gibbs_sets_get");

    if (PhaseTrace::enabled()) {

        std::fprintf(stderr, "[rpvg_amd trace] gibbs on the device: %u rounds, %llu conditionals, %llu sets\n", view.rounds, static_cast<unsigned long long>(view.conditionals), static_cast<unsigned long long>(view.set_off[num_problems]));
    }

    ScopedPhase results_phase("gibbs: generators moved on, posteriors");

    #pragma omp parallel num_threads(shortLoopThreads())
    {
        // the generators end where the reference's sampler leaves them
        #pragma omp for schedule(dynamic, 16) nowait
        for (size_t g = 0; g < generators.size(); ++g) {

            if (view.words_consumed[g] >= std::mt19937::state_size) {

                setGeneratorBehind(generators[g], view.generator_state + g * std::mt19937::state_size);

            } else {

                generators[g]->discard(view.words_consumed[g]);
            }
        }

        #pragma omp for schedule(dynamic, 16)
        for (size_t i = 0; i < num_problems; ++i) {

            auto & result = group_posteriors->at(i);

            const uint64_t first_set = view.set_off[i];
            const uint64_t num_sets = view.set_off[i + 1] - first_set;
            const double num_samples = static_cast<double>(num_chains[i] * num_gibbs_its[i]);

            result.group_size = group_size;
            result.members.reserve(num_sets * group_size);
            result.posteriors.reserve(num_sets);

            for (uint64_t s = first_set; s < first_set + num_sets; ++s) {

                result.members.emplace_back(view.first[s]);

                if (group_size == 2) {

                    result.members.emplace_back(view.second[s]);
                }

                // src/path_estimator.cpp:583-586
                result.posteriors.emplace_back(view.count[s] / num_samples);
            }
        }
    }

    return true;
}

}

int PathEstimator::generatorStateSelfTest(const uint32_t rounds) {

    if (!rawGeneratorAccess()) {

        return 0;
    }

    std::mt19937 positions(7);

    for (uint32_t round = 0; round < rounds; ++round) {

        std::mt19937 generator(1000 + round);
        generator.discard(positions() % 3000);

        std::mt19937 reference = generator;
        GeneratorLayout before, after;
        std::memcpy(&before, &generator, sizeof(before));
        const size_t position = before.p;

        std::vector<uint32_t> next(std::mt19937::state_size), states(std::mt19937::state_size);
        nextGeneratorOutputs(generator, next.data());

        if (generator != reference) {

            return -1;
        }

        for (size_t w = 0; w < std::mt19937::state_size; ++w) {

            if (next[w] != reference()) {

                return -1;
            }
        }

        // the device hands back the untempered values of the last 624 words taken (gibbs_chains.hip), in the order they were
        // taken: the words of the old array from its position on, then the first words of the refilled one
        std::memcpy(&after, &reference, sizeof(after));

        for (size_t w = 0; w < std::mt19937::state_size; ++w) {

            states[w] = static_cast<uint32_t>(position + w < std::mt19937::state_size ? before.x[position + w] : after.x[position + w - std::mt19937::state_size]);
        }

        setGeneratorBehind(&generator, states.data());

        for (size_t w = 0; w < 2 * std::mt19937::state_size; ++w) {

            if (generator() != reference()) {

                return -1;
            }
        }
    }

    return 1;
}

void PathEstimator::estimatePathGroupPosteriorsGibbs(std::vector<GroupPosteriors> * group_posteriors, const DeviceClusterBatch & cluster_batch, const std::vector<GroupPosteriorProblem> & problems, const uint32_t group_size, const bool normalise, const std::vector<std::mt19937 *> & rngs) const {

    assert(group_size > 0);
    assert(rngs.size() == problems.size());

    if (group_size > 4) {

        throw EngineError("estimatePathGroupPosteriorsGibbs: group sizes above 4 are not supported by the GPU log-likelihood kernel");
    }

    group_posteriors->assign(problems.size(), GroupPosteriors());

    if (problems.empty()) {

        return;
    }

    const GroupMatrices matrices(engine, cluster_batch, problems, normalise, prob_precision);

    // group sizes 1 and 2: the chains themselves run on the device (rpvg_hip_group_gibbs), draw for draw
    static const bool host_sampler = std::getenv("RPVG_AMD_HOST_GIBBS") != nullptr;

    if (group_size <= 2 && !host_sampler && estimatePathGroupPosteriorsGibbsOnDevice(group_posteriors, engine, matrices.handle(), problems, group_size, rngs, &PathEstimator::calcPathLogFrequences)) {

        return;
    }

    std::vector<GibbsSampler> samplers(problems.size());

    std::vector<std::vector<size_t> > generator_problems;
    std::unordered_map<std::mt19937 *, size_t> generator_index;

    for (size_t i = 0; i < problems.size(); ++i) {

        auto generator_index_it = generator_index.emplace(rngs.at(i), generator_problems.size());

        if (generator_index_it.second) {

            generator_problems.emplace_back();
        }

        generator_problems.at(generator_index_it.first->second).emplace_back(i);
    }

    #pragma omp parallel for schedule(dynamic, 16) num_threads(hostThreads())
    for (size_t i = 0; i < problems.size(); ++i) {

        auto & sampler = samplers.at(i);

        sampler.num_columns = problems.at(i).numColumns();
        sampler.group_size = group_size;
        sampler.mt_rng = rngs.at(i);
        sampler.log_freqs = calcPathLogFrequences(problems.at(i).column_counts);

        // src/path_estimator.cpp:501-503
        sampler.num_chains = min_gibbs_chains + std::round(gibbs_chain_scaling * group_size * sampler.num_columns);
        sampler.num_burn_its = min_burn_it + std::round(burn_it_scaling * group_size * sampler.num_columns);
        sampler.num_gibbs_its = min_gibbs_it + std::round(gibbs_it_scaling * group_size * sampler.num_columns);

        sampler.init();
    }

    // samplers that share a generator (transcripts of one cluster) are advanced one after the other
    {
    ScopedPhase phase("gibbs: first advance");
    #pragma omp parallel for schedule(dynamic, 4) num_threads(hostThreads())
    for (size_t generator_idx = 0; generator_idx < generator_problems.size(); ++generator_idx) {

        for (auto & i: generator_problems.at(generator_idx)) {

            samplers.at(i).advance();
        }
    }
    }

    // rounds: one device call evaluates the conditional every waiting sampler asked for
    size_t num_rounds = 0;
    size_t num_conditionals = 0;
    size_t num_tail_rounds = 0;

    while (true) {

        std::vector<uint32_t> request_matrix;
        std::vector<uint32_t> request_others;
        std::vector<size_t> first_value(problems.size() + 1, 0);

        ScopedPhase request_phase("gibbs: request build");

        for (size_t i = 0; i < problems.size(); ++i) {

            auto & sampler = samplers.at(i);
            first_value.at(i + 1) = first_value.at(i);

            if (sampler.waiting) {

                request_matrix.emplace_back(i);
                request_others.insert(request_others.end(), sampler.pending_others.begin(), sampler.pending_others.end());
                first_value.at(i + 1) += sampler.num_columns;
            }
        }

        if (request_matrix.empty()) {

            break;
        }

        ++num_rounds;
        num_conditionals += request_matrix.size();
        num_tail_rounds += (request_matrix.size() <= 4);

        request_phase.stop();

        std::vector<double> log_likelihoods;
        matrices.conditionals(&log_likelihoods, request_matrix, request_others, first_value.back(), group_size, group_size);

        ScopedPhase advance_phase("gibbs: supply + advance");

        // problems that share a generator (transcripts of one cluster) are advanced one after the other;
        // the last rounds serve a handful of large problems: no thread team for those
        std::vector<size_t> active_generators;

        for (auto & i: request_matrix) {

            active_generators.emplace_back(generator_index.at(rngs.at(i)));
        }

        std::sort(active_generators.begin(), active_generators.end());
        active_generators.erase(std::unique(active_generators.begin(), active_generators.end()), active_generators.end());

        #pragma omp parallel for schedule(dynamic, 1) num_threads(hostThreads()) if(active_generators.size() > 2)
        for (size_t active_idx = 0; active_idx < active_generators.size(); ++active_idx) {
        for (auto & i: generator_problems.at(active_generators[active_idx])) {

            auto & sampler = samplers.at(i);

            if (!sampler.waiting) {

                continue;
            }

            // src/path_estimator.cpp:537-555
            std::vector<double> group_probs(sampler.num_columns);
            double sum_log_group_probs = numeric::log_zero;

            for (uint32_t k = 0; k < sampler.num_columns; ++k) {

                group_probs.at(k) = log_likelihoods.at(first_value.at(i) + k) + sampler.log_freqs.at(k);
                sum_log_group_probs = numeric::add_log(sum_log_group_probs, group_probs.at(k));
            }

            for (auto & prob: group_probs) {

                prob = std::exp(prob - sum_log_group_probs);
            }

            sampler.supply(std::discrete_distribution<uint32_t>(group_probs.begin(), group_probs.end()));
            sampler.advance();
        }
        }
    }

    if (PhaseTrace::enabled()) {

        std::fprintf(stderr, "[rpvg_amd trace] gibbs: %zu rounds (%zu with <= 4 waiting samplers), %zu conditionals\n", num_rounds, num_tail_rounds, num_conditionals);
    }

    ScopedPhase results_phase("gibbs: results + teardown");

    // the samplers hold one pair of vectors per evaluated conditional: tear them down in parallel too (one at a time:
    // the problems come ordered by size, and the largest sampler alone takes milliseconds)
    #pragma omp parallel for schedule(dynamic, 1) num_threads(hostThreads())
    for (size_t i = 0; i < problems.size(); ++i) {

        auto & sampler = samplers.at(i);
        auto & result = group_posteriors->at(i);

        assert(sampler.done);

        result.group_size = group_size;
        result.members = std::move(sampler.sampled_members);
        result.posteriors.reserve(sampler.sample_counts.size());

        for (auto & sample_count: sampler.sample_counts) {

            result.posteriors.emplace_back(sample_count / static_cast<double>(sampler.num_chains * sampler.num_gibbs_its));
        }

        sampler = GibbsSampler();
    }
}

void PathEstimator::calculatePathGroupPosteriorsBoundedHostDriven(std::vector<GroupPosteriors> * group_posteriors, const DeviceClusterBatch & cluster_batch, const std::vector<GroupPosteriorProblem> & problems, const uint32_t group_size, const double min_rel_likelihood, const bool normalise) const {

    assert(group_size == 2);

    group_posteriors->assign(problems.size(), GroupPosteriors());

    if (problems.empty()) {

        return;
    }

    const double min_log_likelihood_diff = std::log(min_rel_likelihood);

    const GroupMatrices matrices(engine, cluster_batch, problems, normalise, prob_precision);

    std::vector<BoundedSearch> searches(problems.size());

    // Marginal posteriors (the reference's nested Full call with group size 1,
    // src/path_estimator.cpp:397-412) and the optimistic bound of every first
    // path (:424-428): two requests per column.
    {
        std::vector<uint32_t> request_matrix;
        std::vector<uint32_t> marginal_members;
        std::vector<uint32_t> optimistic_members;
        std::vector<size_t> first_request(problems.size() + 1, 0);

        for (size_t i = 0; i < problems.size(); ++i) {

            for (uint32_t j = 0; j < problems.at(i).numColumns(); ++j) {

                request_matrix.emplace_back(i);
                marginal_members.emplace_back(j);
                optimistic_members.emplace_back(j);
                optimistic_members.emplace_back(no_member);
            }

            first_request.at(i + 1) = request_matrix.size();
        }

        std::vector<double> marginal_log_likelihoods;
        matrices.logLikelihoods(&marginal_log_likelihoods, request_matrix, marginal_members, 1, 1, false);

        std::vector<double> optimistic_log_likelihoods;
        matrices.logLikelihoods(&optimistic_log_likelihoods, request_matrix, optimistic_members, 2, 2, true);

        #pragma omp parallel for schedule(dynamic, 8) num_threads(hostThreads())
        for (size_t i = 0; i < problems.size(); ++i) {

            auto & search = searches.at(i);
            const uint32_t num_columns = problems.at(i).numColumns();

            search.log_freqs = calcPathLogFrequences(problems.at(i).column_counts);
            assert(search.log_freqs.size() == num_columns);

            std::vector<std::pair<double, uint32_t> > marginal_posteriors(num_columns);
            double sum_log_posterior = numeric::log_zero;

            for (uint32_t j = 0; j < num_columns; ++j) {

                // + log(numPermutations({j})) = log(1)
                marginal_posteriors.at(j) = std::make_pair(marginal_log_likelihoods.at(first_request.at(i) + j) + search.log_freqs.at(j) + std::log(1), j);
                sum_log_posterior = numeric::add_log(sum_log_posterior, marginal_posteriors.at(j).first);
            }

            for (auto & marginal: marginal_posteriors) {

                marginal.first = std::exp(marginal.first - sum_log_posterior);
            }

            std::sort(marginal_posteriors.begin(), marginal_posteriors.end(), std::greater<std::pair<double, uint32_t> >());

            search.order.reserve(num_columns);
            search.optimistic.reserve(num_columns);

            for (auto & marginal: marginal_posteriors) {

                search.order.emplace_back(marginal.second);

                double optimal_log_likelihood = optimistic_log_likelihoods.at(first_request.at(i) + marginal.second);
                optimal_log_likelihood += search.log_freqs.at(marginal.second) + std::log(2);

                search.optimistic.emplace_back(optimal_log_likelihood);
            }
        }
    }

    // Rounds: every unfinished problem asks for the pair rows of its next
    // first paths that still pass the bound; the sequential pruning loop of
    // the reference (src/path_estimator.cpp:418-451) is then replayed on them.
    // The running maximum only grows, so a first path failing the bound now
    // fails it when the reference reaches it; one passing now may fail at
    // replay time, in which case its fetched row is simply not used.
    uint32_t block_size = 1;

    while (true) {

        std::vector<uint32_t> request_matrix;
        std::vector<uint32_t> request_members;
        std::vector<size_t> first_request(problems.size() + 1, 0);

        bool any_candidates = false;

        std::unique_ptr<ScopedPhase> request_phase(new ScopedPhase("posteriors: bounded request build"));

        for (size_t i = 0; i < problems.size(); ++i) {

            auto & search = searches.at(i);
            const uint32_t num_columns = search.order.size();

            search.candidates.clear();

            for (uint32_t pos = search.cursor; pos < num_columns && search.candidates.size() < block_size; ++pos) {

                if (search.optimistic.at(pos) - search.max_log_likelihood < min_log_likelihood_diff) {

                    continue;
                }

                search.candidates.emplace_back(pos);

                for (uint32_t pos2 = pos; pos2 < num_columns; ++pos2) {

                    request_matrix.emplace_back(i);
                    request_members.emplace_back(search.order.at(pos));
                    request_members.emplace_back(search.order.at(pos2));
                }
            }

            if (search.candidates.empty()) {

                search.cursor = num_columns;

            } else {

                any_candidates = true;
            }

            first_request.at(i + 1) = request_matrix.size();
        }

        request_phase.reset();

        if (!any_candidates) {

            break;
        }

        std::vector<double> pair_log_likelihoods;
        matrices.logLikelihoods(&pair_log_likelihoods, request_matrix, request_members, 2, 2, false);

        ScopedPhase replay_phase("posteriors: bounded replay");

        #pragma omp parallel for schedule(dynamic, 8) num_threads(hostThreads())
        for (size_t i = 0; i < problems.size(); ++i) {

            auto & search = searches.at(i);

            if (search.candidates.empty()) {

                continue;
            }

            const uint32_t num_columns = search.order.size();
            size_t request = first_request.at(i);

            for (auto & pos: search.candidates) {

                const size_t row_request = request;
                request += num_columns - pos;

                if (search.optimistic.at(pos) - search.max_log_likelihood < min_log_likelihood_diff) {

                    continue;
                }

                const uint32_t first_path_idx = search.order.at(pos);

                for (uint32_t pos2 = pos; pos2 < num_columns; ++pos2) {

                    const uint32_t second_path_idx = search.order.at(pos2);

                    double log_likelihood = pair_log_likelihoods.at(row_request + (pos2 - pos));
                    log_likelihood += search.log_freqs.at(first_path_idx) + search.log_freqs.at(second_path_idx) + std::log(numeric::numPermutations(std::vector<uint32_t>({first_path_idx, second_path_idx})));

                    if (log_likelihood - search.max_log_likelihood < min_log_likelihood_diff) {

                        continue;
                    }

                    search.max_log_likelihood = std::max(search.max_log_likelihood, log_likelihood);

                    search.kept_log_likelihoods.emplace_back(log_likelihood);
                    search.kept_members.emplace_back(first_path_idx);
                    search.kept_members.emplace_back(second_path_idx);
                }
            }

            search.cursor = search.candidates.back() + 1;
        }

        block_size = std::min<uint32_t>(block_size * 4, 1024);
    }

    // src/path_estimator.cpp:453-470
    #pragma omp parallel for schedule(dynamic, 8) num_threads(hostThreads())
    for (size_t i = 0; i < problems.size(); ++i) {

        auto & search = searches.at(i);
        auto & result = group_posteriors->at(i);

        double sum_log_posterior = numeric::log_zero;

        for (auto & log_likelihood: search.kept_log_likelihoods) {

            if (log_likelihood - search.max_log_likelihood < min_log_likelihood_diff) {

                log_likelihood = numeric::log_zero;
            }

            sum_log_posterior = numeric::add_log(sum_log_posterior, log_likelihood);
        }

        result.group_size = 2;
        result.members = std::move(search.kept_members);
        result.posteriors.reserve(search.kept_log_likelihoods.size());

        for (auto & log_likelihood: search.kept_log_likelihoods) {

            result.posteriors.emplace_back(std::exp(log_likelihood - sum_log_posterior));
        }
    }
}

}
