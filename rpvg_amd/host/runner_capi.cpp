// C entry points over the host-side estimator classes, for harnesses that are
// not C++ (tests and bench.py bind these with ctypes).  A run goes through the
// same classes a C++ caller uses: PathEstimator::estimateBatch() (mode 0) or
// the reference-shaped per-cluster PathEstimator::estimate() (mode 1).

#include <cassert>
#include <chrono>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/rpvg_batch.h"
#include "batch_pipeline.hpp"
#include "device_group.hpp"
#include "estimator_factory.hpp"
#include "read_rows.hpp"
#include "trace.hpp"

using namespace rpvg_amd;

namespace {

thread_local std::string last_error;

struct Engine {

    std::shared_ptr<HipEngine> hip;
};

// A batch resident on the GPU together with the host-side PathInfo of its clusters.
struct PreparedBatch {

    std::unique_ptr<DeviceClusterBatch> device;

    // batches that start from alignment-path lists keep the lists resident: rows can be rebuilt on the device
    std::unique_ptr<DeviceAlignmentBatch> alignments;
    std::unique_ptr<FragmentLengthDist> fragment_length_dist;
    bool is_single_end = false;
    double min_noise_prob = 0;
    double prob_precision = 1e-8;

    std::vector<std::vector<PathInfo> > paths;

    // kept for the per-cluster estimate() mode
    std::vector<std::vector<ReadPathProbabilities> > rows;

    std::vector<PathClusterEstimates> estimates;
};

struct Result {

    std::vector<uint64_t> set_off, member_off, abund_off, em_off, em_col_off;
    std::vector<uint32_t> members, em_iters, em_cols;
    std::vector<double> posteriors, abundances, noise_count, total_count;

    std::vector<uint64_t> gibbs_off, gibbs_path_off, gibbs_noise_off, gibbs_abund_off;
    std::vector<uint32_t> gibbs_path;
    std::vector<double> gibbs_noise, gibbs_abund;
};

std::vector<std::vector<PathInfo> > unpackPaths(const rpvg_cluster_batch & batch) {

    std::vector<std::vector<PathInfo> > paths(batch.num_clusters);

    for (uint32_t i = 0; i < batch.num_clusters; ++i) {

        for (uint64_t j = batch.cluster_path_off[i]; j < batch.cluster_path_off[i + 1]; ++j) {

            PathInfo info;
            info.group_id = batch.path_group_id ? batch.path_group_id[j] : 0;
            info.source_count = batch.path_source_count ? batch.path_source_count[j] : 1;

            if (batch.path_source_off) {

                info.source_ids.insert(batch.source_id + batch.path_source_off[j], batch.source_id + batch.path_source_off[j + 1]);
            }

            info.effective_length = batch.path_effective_length ? batch.path_effective_length[j] : 0;
            paths.at(i).emplace_back(std::move(info));
        }
    }

    return paths;
}

std::vector<ReadPathProbabilities> unpackRows(const rpvg_cluster_batch & batch, const uint32_t cluster, const double prob_precision) {

    std::vector<ReadPathProbabilities> rows;

    for (uint64_t i = batch.cluster_row_off[cluster]; i < batch.cluster_row_off[cluster + 1]; ++i) {

        ReadPathProbabilities::PathProbs path_probs;

        for (uint64_t j = rpvg_batch_row_group_offset(&batch, i); j < rpvg_batch_row_group_offset(&batch, i + 1); ++j) {

            path_probs.emplace_back(batch.grp_prob[j], std::vector<uint32_t>(batch.path_idx + rpvg_batch_group_entry_offset(&batch, j), batch.path_idx + rpvg_batch_group_entry_offset(&batch, j + 1)));
        }

        rows.emplace_back(batch.row_count[i], batch.row_noise[i], path_probs, prob_precision);
    }

    return rows;
}

Result * packResult(const std::vector<PathClusterEstimates> & estimates) {

    Result * result = new Result();

    result->set_off.push_back(0);
    result->member_off.push_back(0);
    result->abund_off.push_back(0);
    result->em_off.push_back(0);
    result->em_col_off.push_back(0);

    result->gibbs_off.push_back(0);
    result->gibbs_path_off.push_back(0);
    result->gibbs_noise_off.push_back(0);
    result->gibbs_abund_off.push_back(0);

    for (auto & cluster_estimates: estimates) {

        for (auto & count_samples: cluster_estimates.gibbs_read_count_samples) {

            result->gibbs_path.insert(result->gibbs_path.end(), count_samples.path_ids.begin(), count_samples.path_ids.end());
            result->gibbs_path_off.push_back(result->gibbs_path.size());
            result->gibbs_noise.insert(result->gibbs_noise.end(), count_samples.noise_samples.begin(), count_samples.noise_samples.end());
            result->gibbs_noise_off.push_back(result->gibbs_noise.size());
            result->gibbs_abund.insert(result->gibbs_abund.end(), count_samples.abundance_samples.begin(), count_samples.abundance_samples.end());
            result->gibbs_abund_off.push_back(result->gibbs_abund.size());
        }

        result->gibbs_off.push_back(result->gibbs_path_off.size() - 1);

        for (size_t i = 0; i < cluster_estimates.path_group_sets.size(); ++i) {

            result->members.insert(result->members.end(), cluster_estimates.path_group_sets.at(i).begin(), cluster_estimates.path_group_sets.at(i).end());
            result->member_off.push_back(result->members.size());
            result->posteriors.push_back(cluster_estimates.posteriors.at(i));
        }

        result->set_off.push_back(result->posteriors.size());

        result->abundances.insert(result->abundances.end(), cluster_estimates.abundances.begin(), cluster_estimates.abundances.end());
        result->abund_off.push_back(result->abundances.size());

        result->noise_count.push_back(cluster_estimates.noise_count);
        result->total_count.push_back(cluster_estimates.total_count);

        for (size_t i = 0; i < cluster_estimates.em_iterations.size(); ++i) {

            result->em_iters.push_back(cluster_estimates.em_iterations.at(i));
            result->em_cols.insert(result->em_cols.end(), cluster_estimates.em_problem_paths.at(i).begin(), cluster_estimates.em_problem_paths.at(i).end());
            result->em_col_off.push_back(result->em_cols.size());
        }

        result->em_off.push_back(result->em_iters.size());
    }

    return result;
}

}

extern "C" {

const char * rpvg_amd_last_error(void) {

    return last_error.c_str();
}

void * rpvg_amd_engine_create(int device) {

    try {

        Engine * engine = new Engine();
        engine->hip = std::make_shared<HipEngine>(device);
        return engine;

    } catch (const std::exception & e) {

        last_error = e.what();
        return nullptr;
    }
}

// an engine for rpvg_amd_batch_reupload next to the engine that estimates (HipEngine(device, uploader = true))
void * rpvg_amd_engine_create_uploader(int device) {

    try {

        Engine * engine = new Engine();
        engine->hip = std::make_shared<HipEngine>(device, true);
        return engine;

    } catch (const std::exception & e) {

        last_error = e.what();
        return nullptr;
    }
}

void rpvg_amd_engine_destroy(void * engine) {

    delete static_cast<Engine *>(engine);
}

// The rpvg_hip_ctx of the engine (to read kernel statistics through include/rpvg_hip.h).
void * rpvg_amd_engine_ctx(void * engine) {

    return static_cast<Engine *>(engine)->hip->ctx();
}

// Kernel statistics of the engine, both host lanes together.
int rpvg_amd_engine_stats_get(void * engine, rpvg_hip_kernel_stats * stats_out) {

    try {

        static_cast<Engine *>(engine)->hip->stats(stats_out);
        return 0;

    } catch (const std::exception & e) {

        last_error = e.what();
        return -1;
    }
}

int rpvg_amd_engine_stats_reset(void * engine) {

    try {

        static_cast<Engine *>(engine)->hip->resetStats();
        return 0;

    } catch (const std::exception & e) {

        last_error = e.what();
        return -1;
    }
}

// The OpenMP team of one host lane of this process (trace.hpp, hostThreads()): what bench.py prints as host_threads_per_lane.
int rpvg_amd_host_threads() {

    return rpvg_amd::hostThreads();
}

// PathEstimator::generatorStateSelfTest (path_estimator.hpp): how the host side of the device sampler reaches the generators' states
int rpvg_amd_generator_state_check(uint32_t rounds) {

    return rpvg_amd::PathEstimator::generatorStateSelfTest(rounds);
}

// Uploads the batch to the GPU.  keep_rows != 0 also keeps ReadPathProbabilities
// objects of every cluster for the per-cluster estimate() mode.
void * rpvg_amd_batch_prepare(void * engine, const rpvg_cluster_batch * batch, int keep_rows) {

    try {

        PreparedBatch * prepared = new PreparedBatch();
        std::unique_ptr<PreparedBatch> guard(prepared);

        prepared->paths = unpackPaths(*batch);

        if (keep_rows) {

            for (uint32_t i = 0; i < batch->num_clusters; ++i) {

                prepared->rows.emplace_back(unpackRows(*batch, i, 1e-8));
            }

        } else {

            prepared->device.reset(new DeviceClusterBatch(static_cast<Engine *>(engine)->hip, *batch));
        }

        return guard.release();

    } catch (const std::exception & e) {

        last_error = e.what();
        return nullptr;
    }
}

// The same, starting one step earlier: the batch's reads arrive as alignment-path lists (include/rpvg_rows.h) and
// the rows are constructed on the GPU (read_rows.hpp); `path_info` supplies the path arrays of rpvg_cluster_batch
// (the PathInfo of every cluster; its row arrays are ignored).  seconds_out = wall time of the row construction
// with the alignment lists in host memory (upload included).
void * rpvg_amd_batch_prepare_from_alignments(void * engine, const rpvg_alignment_batch * alignments, const rpvg_cluster_batch * path_info, double frag_loc, double frag_scale, double frag_shape, uint32_t frag_sd_max_multi, int is_single_end, double min_noise_prob, double prob_precision, double * seconds_out) {

    try {

        PreparedBatch * prepared = new PreparedBatch();
        std::unique_ptr<PreparedBatch> guard(prepared);

        prepared->paths = unpackPaths(*path_info);
        assert(prepared->paths.size() == alignments->num_clusters);

        AlignmentBatchBuilder builder;
        const bool collapse = alignments->path_group != nullptr;

        for (uint32_t i = 0; i < alignments->num_clusters; ++i) {

            std::vector<uint32_t> group_name_index;
            uint32_t num_groups = 0;

            if (collapse) {

                group_name_index.assign(alignments->path_group + alignments->cluster_path_off[i], alignments->path_group + alignments->cluster_path_off[i + 1]);
                num_groups = alignments->cluster_group_off[i + 1] - alignments->cluster_group_off[i];
            }

            builder.beginCluster(prepared->paths.at(i), group_name_index, num_groups);

            for (uint64_t r = alignments->cluster_read_off[i]; r < alignments->cluster_read_off[i + 1]; ++r) {

                std::vector<AlignmentPath> align_paths;

                for (uint64_t a = alignments->read_align_off[r]; a < alignments->read_align_off[r + 1]; ++a) {

                    align_paths.emplace_back(alignments->read_min_mapq[r], alignments->align_score_sum[a], alignments->align_length[a], alignments->align_frag_length[a], std::vector<uint32_t>(alignments->align_path_idx + alignments->align_path_off[a], alignments->align_path_idx + alignments->align_path_off[a + 1]));
                }

                align_paths.emplace_back(alignments->read_min_mapq[r], alignments->read_noise_score[r], 0, 0, std::vector<uint32_t>());
                builder.addAlignmentPaths(align_paths, alignments->read_count[r]);
            }
        }

        prepared->fragment_length_dist.reset(is_single_end ? new FragmentLengthDist() : new FragmentLengthDist(frag_loc, frag_scale, frag_shape, frag_sd_max_multi));
        prepared->is_single_end = is_single_end != 0;
        prepared->min_noise_prob = min_noise_prob;
        prepared->prob_precision = prob_precision;

        const auto start = std::chrono::steady_clock::now();
        prepared->alignments.reset(new DeviceAlignmentBatch(static_cast<Engine *>(engine)->hip, builder));
        prepared->device = constructReadPathProbabilities(*prepared->alignments, *prepared->fragment_length_dist, prepared->is_single_end, min_noise_prob, prob_precision);

        if (seconds_out) {

            *seconds_out = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
        }

        return guard.release();

    } catch (const std::exception & e) {

        last_error = e.what();
        return nullptr;
    }
}

// BASELINE.json configs[1] for the estimator classes: one cluster of `num_rows` rows that each touch all `num_paths` paths,
// generated on the device (rpvg_hip_synth_dense_cluster_batch: the host form of 10^6 x 2 000 rows would be 40 GB) and
// adopted as a resident batch — the caller runs `transcripts` on it like on any other batch.
void * rpvg_amd_batch_prepare_synth_dense(void * engine, uint64_t seed, uint64_t num_rows, uint32_t num_paths) {

    try {

        PreparedBatch * prepared = new PreparedBatch();
        std::unique_ptr<PreparedBatch> guard(prepared);

        prepared->paths.emplace_back();

        for (uint32_t j = 0; j < num_paths; ++j) {

            PathInfo info;
            info.group_id = 0;
            info.source_count = 1;
            info.effective_length = 1000;
            prepared->paths.back().emplace_back(std::move(info));
        }

        auto hip = static_cast<Engine *>(engine)->hip;

        rpvg_hip_batch * device_batch = nullptr;
        HipEngine::check(rpvg_hip_synth_dense_cluster_batch(hip->ctx(), seed, num_rows, num_paths, &device_batch), "rpvg_hip_synth_dense_cluster_batch");

        const uint64_t cluster_row_off[2] = {0, num_rows};
        const uint64_t cluster_path_off[2] = {0, num_paths};

        rpvg_cluster_batch offsets;
        std::memset(&offsets, 0, sizeof(offsets));
        offsets.num_clusters = 1;
        offsets.cluster_row_off = cluster_row_off;
        offsets.cluster_path_off = cluster_path_off;

        // (every row of the synthetic cluster is one read pair; the multi-gigabyte batch is freed here until the handle has adopted it)
        struct BatchGuard {

            rpvg_hip_ctx * ctx;
            rpvg_hip_batch * batch;
            ~BatchGuard() { if (batch) rpvg_hip_batch_free(ctx, batch); }

        } batch_guard{hip->ctx(), device_batch};

        prepared->device.reset(new DeviceClusterBatch(hip, device_batch, offsets, std::vector<double>(1, static_cast<double>(num_rows))));
        batch_guard.batch = nullptr;

        return guard.release();

    } catch (const std::exception & e) {

        last_error = e.what();
        return nullptr;
    }
}

// A new resident copy of the rows of a prepared batch from the same host arrays (what arrives per batch in a
// running pipeline: the rows; the PathInfo side stays).  `engine` may be another engine on the same GPU than the one
// that estimates — an uploader with a context and stream of its own, so that the copy of batch n + 1 runs under
// the kernels of batch n.  seconds_out = wall time of validation + H2D + expansion on the device.
int rpvg_amd_batch_reupload(void * engine, void * prepared_batch, const rpvg_cluster_batch * batch, double * seconds_out) {

    try {

        PreparedBatch * prepared = static_cast<PreparedBatch *>(prepared_batch);
        const auto start = std::chrono::steady_clock::now();

        std::unique_ptr<DeviceClusterBatch> fresh(new DeviceClusterBatch(static_cast<Engine *>(engine)->hip, *batch));
        prepared->device = std::move(fresh);

        if (seconds_out) {

            *seconds_out = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
        }

        return 0;

    } catch (const std::exception & e) {

        last_error = e.what();
        return -1;
    }
}

void rpvg_amd_batch_free(void * prepared) {

    delete static_cast<PreparedBatch *>(prepared);
}

// Runs `model` on a prepared batch.  seconds_out = wall time of the estimator
// call(s) only (inputs already resident on the GPU in batch mode).
void * rpvg_amd_run(void * engine, void * prepared_batch, const char * model, const rpvg_params * params, double * seconds_out) {

    try {

        PreparedBatch * prepared = static_cast<PreparedBatch *>(prepared_batch);
        auto estimator = makePathEstimator(model, *params, static_cast<Engine *>(engine)->hip);

        // the estimates containers (with PathInfo filled in, as src/main.cpp:855-887 does
        // before estimate()) are created once per prepared batch and reused by every run
        if (prepared->estimates.size() != prepared->paths.size()) {

            prepared->estimates.assign(prepared->paths.size(), PathClusterEstimates());

            for (size_t i = 0; i < prepared->estimates.size(); ++i) {

                prepared->estimates.at(i).paths = prepared->paths.at(i);
            }
        }

        std::vector<PathClusterEstimates> & estimates = prepared->estimates;

        const auto start = std::chrono::steady_clock::now();

        if (prepared->device) {

            estimator->estimateBatchSeeded(&estimates, *prepared->device, params->rng_seed);

        } else {

            // the reference's loop body: one estimate() per cluster (src/main.cpp:976-977)
            for (size_t i = 0; i < estimates.size(); ++i) {

                std::mt19937 mt_rng(params->rng_seed + i);
                estimator->estimate(&estimates.at(i), prepared->rows.at(i), &mt_rng);
            }
        }

        const auto stop = std::chrono::steady_clock::now();

        if (seconds_out) {

            *seconds_out = std::chrono::duration<double>(stop - start).count();
        }

        PhaseTrace::add("total estimate call", std::chrono::duration<double>(stop - start).count());
        PhaseTrace::report();

        return packResult(estimates);

    } catch (const std::exception & e) {

        last_error = e.what();
        return nullptr;
    }
}

// The reference's cluster loop (src/main.cpp:829,976-977) on a batch prepared with keep_rows: estimate() once per cluster
// from an OpenMP team of `threads` (schedule(dynamic, 1), clusters in the batch's order), cluster i with mt19937(rng_seed + i).
// The estimates stay in the prepared batch's containers (rpvg_amd_run_team_result flattens them).
int rpvg_amd_run_team(void * engine, void * prepared_batch, const char * model, const rpvg_params * params, int threads, double * seconds_out) {

    try {

        PreparedBatch * prepared = static_cast<PreparedBatch *>(prepared_batch);

        if (prepared->rows.size() != prepared->paths.size()) {

            last_error = "rpvg_amd_run_team needs a batch prepared with keep_rows";
            return -1;
        }

        auto estimator = makePathEstimator(model, *params, static_cast<Engine *>(engine)->hip);

        if (prepared->estimates.size() != prepared->paths.size()) {

            prepared->estimates.assign(prepared->paths.size(), PathClusterEstimates());

            for (size_t i = 0; i < prepared->estimates.size(); ++i) {

                prepared->estimates.at(i).paths = prepared->paths.at(i);
            }
        }

        std::vector<PathClusterEstimates> & estimates = prepared->estimates;
        std::string first_failure;

        const auto start = std::chrono::steady_clock::now();
        ScopedPhase pass_phase("team: one pass of estimate() calls");

        #pragma omp parallel for schedule(dynamic, 1) num_threads(std::max(1, threads))
        for (size_t i = 0; i < estimates.size(); ++i) {

            try {

                std::mt19937 mt_rng(params->rng_seed + i);
                estimator->estimate(&estimates.at(i), prepared->rows.at(i), &mt_rng);

            } catch (const std::exception & e) {

                #pragma omp critical
                if (first_failure.empty()) {

                    first_failure = e.what();
                }
            }
        }

        pass_phase.stop();

        if (seconds_out) {

            *seconds_out = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
        }

        PhaseTrace::report();

        if (!first_failure.empty()) {

            last_error = first_failure;
            return -1;
        }

        return 0;

    } catch (const std::exception & e) {

        last_error = e.what();
        return -1;
    }
}

void * rpvg_amd_run_team_result(void * prepared_batch) {

    try {

        return packResult(static_cast<PreparedBatch *>(prepared_batch)->estimates);

    } catch (const std::exception & e) {

        last_error = e.what();
        return nullptr;
    }
}

// Same run, estimates left in the prepared batch's containers and not flattened
// (timing loops).  Returns 0 on success.
int rpvg_amd_run_inplace(void * engine, void * prepared_batch, const char * model, const rpvg_params * params, double * seconds_out) {

    try {

        PreparedBatch * prepared = static_cast<PreparedBatch *>(prepared_batch);

        if (!prepared->device) {

            last_error = "rpvg_amd_run_inplace needs a batch prepared for estimateBatch()";
            return -1;
        }

        auto estimator = makePathEstimator(model, *params, static_cast<Engine *>(engine)->hip);

        if (prepared->estimates.size() != prepared->paths.size()) {

            prepared->estimates.assign(prepared->paths.size(), PathClusterEstimates());

            for (size_t i = 0; i < prepared->estimates.size(); ++i) {

                prepared->estimates.at(i).paths = prepared->paths.at(i);
            }
        }

        const auto start = std::chrono::steady_clock::now();
        estimator->estimateBatchSeeded(&prepared->estimates, *prepared->device, params->rng_seed);
        const auto stop = std::chrono::steady_clock::now();

        if (seconds_out) {

            *seconds_out = std::chrono::duration<double>(stop - start).count();
        }

        PhaseTrace::add("total estimate call", std::chrono::duration<double>(stop - start).count());
        PhaseTrace::report();

        return 0;

    } catch (const std::exception & e) {

        last_error = e.what();
        return -1;
    }
}

// One pass of the widened path with the alignment-path lists resident on the GPU: rows are constructed and merged on
// the device (read_rows.hpp), become the estimators' batch without leaving it, and `model` runs on them.
// rows_seconds_out / estimate_seconds_out = wall time of the two stages.
int rpvg_amd_run_from_alignments_inplace(void * engine, void * prepared_batch, const char * model, const rpvg_params * params, double * rows_seconds_out, double * estimate_seconds_out) {

    try {

        PreparedBatch * prepared = static_cast<PreparedBatch *>(prepared_batch);

        if (!prepared->alignments) {

            last_error = "rpvg_amd_run_from_alignments_inplace needs a batch prepared from alignment-path lists";
            return -1;
        }

        const auto start = std::chrono::steady_clock::now();

        prepared->device.reset();
        prepared->device = constructReadPathProbabilities(*prepared->alignments, *prepared->fragment_length_dist, prepared->is_single_end, prepared->min_noise_prob, prepared->prob_precision);

        const auto rows_done = std::chrono::steady_clock::now();

        if (rows_seconds_out) {

            *rows_seconds_out = std::chrono::duration<double>(rows_done - start).count();
        }

        return rpvg_amd_run_inplace(engine, prepared_batch, model, params, estimate_seconds_out);

    } catch (const std::exception & e) {

        last_error = e.what();
        return -1;
    }
}

// ---- the GPUs of a node behind one call (device_group.hpp) -------------------------------------------------------

struct Group {

    std::unique_ptr<DeviceGroup> devices;
    std::vector<PathClusterEstimates> estimates;
};

void * rpvg_amd_group_create(const int * devices, int num_devices) {

    try {

        Group * group = new Group();
        std::unique_ptr<Group> guard(group);
        group->devices.reset(new DeviceGroup(std::vector<int>(devices, devices + num_devices)));
        return guard.release();

    } catch (const std::exception & e) {

        last_error = e.what();
        return nullptr;
    }
}

void rpvg_amd_group_destroy(void * group) {

    delete static_cast<Group *>(group);
}

int rpvg_amd_group_has_communicator(void * group) {

    return static_cast<Group *>(group)->devices->hasCommunicator() ? 1 : 0;
}

// Estimates of every cluster of a host batch, its clusters sharded over the group's GPUs.
void * rpvg_amd_group_run(void * group_handle, const rpvg_cluster_batch * batch, const char * model, const rpvg_params * params, double * seconds_out) {

    try {

        Group * group = static_cast<Group *>(group_handle);
        const auto paths = unpackPaths(*batch);

        group->estimates.assign(paths.size(), PathClusterEstimates());

        for (size_t i = 0; i < paths.size(); ++i) {

            group->estimates.at(i).paths = paths.at(i);
        }

        const auto start = std::chrono::steady_clock::now();
        group->devices->estimateBatch(&group->estimates, *batch, model, *params);

        if (seconds_out) {

            *seconds_out = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
        }

        return packResult(group->estimates);

    } catch (const std::exception & e) {

        last_error = e.what();
        return nullptr;
    }
}

// GPU of every cluster in the last run.
int rpvg_amd_group_partition(void * group_handle, uint32_t * device_of_cluster, uint64_t num_clusters) {

    Group * group = static_cast<Group *>(group_handle);
    const auto & partition = group->devices->lastPartition();

    for (size_t idx = 0; idx < partition.size(); ++idx) {

        for (auto & cluster: partition.at(idx)) {

            if (cluster >= num_clusters) {

                last_error = "rpvg_amd_group_partition: output too short";
                return -1;
            }

            device_of_cluster[cluster] = idx;
        }
    }

    return 0;
}

// The final gather of the last run: abundances of all clusters in cluster order (capacity doubles available), their
// number, and the TPM denominator.
int rpvg_amd_group_gather(void * group_handle, double * abundances_out, uint64_t capacity, uint64_t * count_out, double * total_transcript_count_out) {

    try {

        Group * group = static_cast<Group *>(group_handle);
        const auto gathered = group->devices->gatherAbundances(group->estimates, total_transcript_count_out);

        if (gathered.size() > capacity) {

            last_error = "rpvg_amd_group_gather: output too short";
            return -1;
        }

        std::copy(gathered.begin(), gathered.end(), abundances_out);
        *count_out = gathered.size();
        return 0;

    } catch (const std::exception & e) {

        last_error = e.what();
        return -1;
    }
}

// ---- several batches in flight on one GPU (batch_pipeline.hpp) --------------------------------------------------------

struct Pipeline {

    std::unique_ptr<BatchPipeline> pipeline;

    // the containers the estimates of the batches in flight go to: a harness hands the same host batch in again and again,
    // and every batch in flight needs containers of its own
    std::vector<std::vector<PathClusterEstimates> > slots;
};

void * rpvg_amd_pipeline_create(int device, const char * model, const rpvg_params * params, int workers) {

    try {

        Pipeline * pipeline = new Pipeline();
        std::unique_ptr<Pipeline> guard(pipeline);
        pipeline->pipeline.reset(new BatchPipeline(device, model, *params, workers));
        return guard.release();

    } catch (const std::exception & e) {

        last_error = e.what();
        return nullptr;
    }
}

void rpvg_amd_pipeline_destroy(void * pipeline) {

    delete static_cast<Pipeline *>(pipeline);
}

int rpvg_amd_pipeline_workers(void * pipeline) {

    return static_cast<Pipeline *>(pipeline)->pipeline->numWorkers();
}

// `slots` sets of estimates containers with the PathInfo of the batch's clusters filled in (src/main.cpp:855-887).
int rpvg_amd_pipeline_prepare_slots(void * pipeline_handle, const rpvg_cluster_batch * batch, int slots) {

    try {

        Pipeline * pipeline = static_cast<Pipeline *>(pipeline_handle);
        pipeline->pipeline->wait();

        const auto paths = unpackPaths(*batch);
        pipeline->slots.assign(slots, std::vector<PathClusterEstimates>(paths.size()));

        for (auto & slot: pipeline->slots) {

            for (size_t i = 0; i < paths.size(); ++i) {

                slot.at(i).paths = paths.at(i);
            }
        }

        return 0;

    } catch (const std::exception & e) {

        last_error = e.what();
        return -1;
    }
}

// The containers of ONE slot, for the clusters (paths) of `batch` — slots may hold different batches (the parts of one data set,
// submitted one behind the other); the slots in front of it exist afterwards, empty if they were not prepared.
int rpvg_amd_pipeline_prepare_slot(void * pipeline_handle, const rpvg_cluster_batch * batch, int slot) {

    try {

        Pipeline * pipeline = static_cast<Pipeline *>(pipeline_handle);
        pipeline->pipeline->wait();

        const auto paths = unpackPaths(*batch);

        if (pipeline->slots.size() <= static_cast<size_t>(slot)) {

            pipeline->slots.resize(slot + 1);
        }

        auto & containers = pipeline->slots.at(slot);
        containers.assign(paths.size(), PathClusterEstimates());

        for (size_t i = 0; i < paths.size(); ++i) {

            containers.at(i).paths = paths.at(i);
        }

        return 0;

    } catch (const std::exception & e) {

        last_error = e.what();
        return -1;
    }
}

// Queues one batch (its arrays stay the caller's until rpvg_amd_pipeline_wait returns) with the containers of `slot`.
int rpvg_amd_pipeline_submit(void * pipeline_handle, const rpvg_cluster_batch * batch, int slot) {

    try {

        Pipeline * pipeline = static_cast<Pipeline *>(pipeline_handle);
        pipeline->pipeline->submit(*batch, &pipeline->slots.at(slot));
        return 0;

    } catch (const std::exception & e) {

        last_error = e.what();
        return -1;
    }
}

int rpvg_amd_pipeline_wait(void * pipeline_handle) {

    try {

        static_cast<Pipeline *>(pipeline_handle)->pipeline->wait();
        return 0;

    } catch (const std::exception & e) {

        last_error = e.what();
        return -1;
    }
}

// The estimates a finished batch left in the containers of `slot` (call after rpvg_amd_pipeline_wait).
void * rpvg_amd_pipeline_result(void * pipeline_handle, int slot) {

    try {

        return packResult(static_cast<Pipeline *>(pipeline_handle)->slots.at(slot));

    } catch (const std::exception & e) {

        last_error = e.what();
        return nullptr;
    }
}

int rpvg_amd_pipeline_stats_get(void * pipeline_handle, rpvg_hip_kernel_stats * stats_out, double * mean_upload_seconds_out) {

    try {

        Pipeline * pipeline = static_cast<Pipeline *>(pipeline_handle);
        pipeline->pipeline->stats(stats_out);

        if (mean_upload_seconds_out) {

            uint64_t batches = 0;
            mean_upload_seconds_out[0] = pipeline->pipeline->meanUploadSeconds(&batches);

            // [1], [2]: device milliseconds per batch of the uploads' copies and of the kernels behind them
            pipeline->pipeline->uploadDeviceMs(mean_upload_seconds_out + 1, mean_upload_seconds_out + 2);
            mean_upload_seconds_out[1] /= std::max<uint64_t>(1, batches);
            mean_upload_seconds_out[2] /= std::max<uint64_t>(1, batches);

            // [3], [4], [5]: wall seconds per batch of a worker's upload finish, estimate, and wait for a resident batch
            pipeline->pipeline->workerSeconds(mean_upload_seconds_out + 3, mean_upload_seconds_out + 4, mean_upload_seconds_out + 5);
        }

        return 0;

    } catch (const std::exception & e) {

        last_error = e.what();
        return -1;
    }
}

// Completion times of the batches since the last statistics reset (seconds since that reset, in order of completion).
int rpvg_amd_pipeline_completions(void * pipeline_handle, double * seconds_out, uint64_t capacity, uint64_t * count_out) {

    const auto completions = static_cast<Pipeline *>(pipeline_handle)->pipeline->completionSeconds();
    *count_out = completions.size();
    std::copy(completions.begin(), completions.begin() + std::min<size_t>(capacity, completions.size()), seconds_out);
    return 0;
}

int rpvg_amd_pipeline_stats_reset(void * pipeline_handle) {

    try {

        static_cast<Pipeline *>(pipeline_handle)->pipeline->resetStats();
        return 0;

    } catch (const std::exception & e) {

        last_error = e.what();
        return -1;
    }
}

void rpvg_amd_result_view(void * result_handle, rpvg_estimates_view * out) {

    Result * result = static_cast<Result *>(result_handle);

    out->num_clusters = result->noise_count.size();
    out->set_off = result->set_off.data();
    out->member_off = result->member_off.data();
    out->members = result->members.data();
    out->posteriors = result->posteriors.data();
    out->abund_off = result->abund_off.data();
    out->abundances = result->abundances.data();
    out->noise_count = result->noise_count.data();
    out->total_count = result->total_count.data();
    out->em_off = result->em_off.data();
    out->em_iters = result->em_iters.data();
    out->em_col_off = result->em_col_off.data();
    out->em_cols = result->em_cols.data();
    out->gibbs_off = result->gibbs_off.data();
    out->gibbs_path_off = result->gibbs_path_off.data();
    out->gibbs_path = result->gibbs_path.data();
    out->gibbs_noise_off = result->gibbs_noise_off.data();
    out->gibbs_noise = result->gibbs_noise.data();
    out->gibbs_abund_off = result->gibbs_abund_off.data();
    out->gibbs_abund = result->gibbs_abund.data();
}

void rpvg_amd_result_free(void * result_handle) {

    delete static_cast<Result *>(result_handle);
}

}
