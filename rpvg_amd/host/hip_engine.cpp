#include "hip_engine.hpp"

#include <cstdio>
#include <cstring>
#include "trace.hpp"

#include <algorithm>
#include <cassert>
#include <cstdlib>
#include <mutex>

#include <malloc.h>

namespace rpvg_amd {

// Every batch builds and drops tens of megabytes of per-cluster containers on a few dozen threads.  With glibc's
// defaults the top of each heap goes back to the kernel when they are dropped and is faulted in again by the next
// batch: measured on the bench workload, 45 of the 58 CPU-seconds of 60 batches were system time.  A top pad keeps
// that memory with the process (3 CPU-seconds of system time; wall time unchanged, the host's other ranks get the
// cores).  It only takes hold in malloc arenas created afterwards: create the engine before the host threads start
// (the Python harness sets it at import, rpvg_amd/__init__.py).  MALLOC_TOP_PAD_ in the environment wins.
static void keepHeapTop() {

    static std::once_flag once;

    std::call_once(once, []() {

        if (!std::getenv("MALLOC_TOP_PAD_")) {

            mallopt(M_TOP_PAD, 64 << 20);
        }
    });
}

static int defaultHostLanes() {

    if (std::getenv("RPVG_AMD_SINGLE_LANE")) {

        return 1;
    }

    const char * env = std::getenv("RPVG_AMD_LANES");
    return env ? std::max(1, std::min(HipEngine::max_lanes, std::atoi(env))) : 2;
}

HipEngine::HipEngine(const int device, const bool uploader, const int host_lanes_in) : context(nullptr), device_id(device), host_lanes(host_lanes_in > 0 ? std::min(host_lanes_in, max_lanes) : defaultHostLanes()) {

    for (auto & lane_context: lane_contexts) {

        lane_context.store(nullptr);
    }

    lane_workers.reserve(max_lanes - 1);

    keepHeapTop();

    // an engine that runs whole batches (one lane) stands next to others like it on the GPU (BatchPipeline): few side streams,
    // so that all of them together stay within the hardware queues (RPVG_AMD_SIDE_STREAMS: A/B)
    static const int lean_side_streams = []() {

        const char * env = std::getenv("RPVG_AMD_SIDE_STREAMS");
        return env ? std::max(1, std::min(6, std::atoi(env))) : 3;
    }();

    if (uploader) {

        check(rpvg_hip_create_uploader(device, &context), "rpvg_hip_create");

    } else if (host_lanes_in == 1) {

        check(rpvg_hip_create_with_streams(device, lean_side_streams, &context), "rpvg_hip_create");

    } else {

        check(rpvg_hip_create(device, &context), "rpvg_hip_create");
    }
}

HipEngine::~HipEngine() {

    const bool trace = std::getenv("RPVG_AMD_TRACE_EXIT") != nullptr;
    if (trace) std::fprintf(stderr, "[exit] ~HipEngine begins\n");

    lane_workers.clear();
    if (trace) std::fprintf(stderr, "[exit] lane workers joined\n");

    for (auto & lane_context: laneContexts()) {

        rpvg_hip_destroy(lane_context);
        if (trace) std::fprintf(stderr, "[exit] a lane context destroyed\n");
    }

    rpvg_hip_destroy(context);
    if (trace) std::fprintf(stderr, "[exit] ~HipEngine done\n");
}

std::shared_ptr<HipEngine> HipEngine::processDefault() {

    static std::mutex default_mutex;
    static std::shared_ptr<HipEngine> default_engine;

    std::lock_guard<std::mutex> lock(default_mutex);

    if (!default_engine) {

        const char * device = std::getenv("RPVG_AMD_DEVICE");
        default_engine = std::make_shared<HipEngine>(device ? std::atoi(device) : 0);
    }

    return default_engine;
}

int & HipEngine::currentLane() {

    thread_local int lane = 0;
    return lane;
}

void HipEngine::stats(rpvg_hip_kernel_stats * stats_out) const {

    stats(std::vector<const HipEngine *>(1, this), stats_out);
}

void HipEngine::stats(const std::vector<const HipEngine *> & engines, rpvg_hip_kernel_stats * stats_out) {

    std::vector<rpvg_hip_ctx *> contexts;

    for (auto & engine: engines) {

        contexts.emplace_back(engine->context);
        const auto lane_contexts = engine->laneContexts();
        contexts.insert(contexts.end(), lane_contexts.begin(), lane_contexts.end());
    }

    assert(!contexts.empty());
    check(rpvg_hip_stats_get(contexts.front(), stats_out), "rpvg_hip_stats_get");

    for (size_t c = 1; c < contexts.size(); ++c) {

        rpvg_hip_kernel_stats lane_stats;
        check(rpvg_hip_stats_get(contexts[c], &lane_stats), "rpvg_hip_stats_get");

        stats_out->em_sparse_ms += lane_stats.em_sparse_ms;
        stats_out->em_sparse_launches += lane_stats.em_sparse_launches;
        stats_out->em_sparse_alg_bytes += lane_stats.em_sparse_alg_bytes;
        stats_out->em_dense_ms += lane_stats.em_dense_ms;
        stats_out->em_dense_launches += lane_stats.em_dense_launches;
        stats_out->em_dense_alg_bytes += lane_stats.em_dense_alg_bytes;
        stats_out->loglik_ms += lane_stats.loglik_ms;
        stats_out->loglik_launches += lane_stats.loglik_launches;
        stats_out->loglik_evals += lane_stats.loglik_evals;
        stats_out->build_ms += lane_stats.build_ms;
        stats_out->build_launches += lane_stats.build_launches;
        stats_out->h2d_ms += lane_stats.h2d_ms;
        stats_out->h2d_bytes += lane_stats.h2d_bytes;
        stats_out->em_iterations_total += lane_stats.em_iterations_total;
        stats_out->search_pairs_possible += lane_stats.search_pairs_possible;
        stats_out->search_pairs_table += lane_stats.search_pairs_table;
        stats_out->search_pairs_kept += lane_stats.search_pairs_kept;
        stats_out->collapse_ms += lane_stats.collapse_ms;
        stats_out->gibbs_ms += lane_stats.gibbs_ms;
        stats_out->search_tile_ms += lane_stats.search_tile_ms;
        stats_out->search_tile_launches += lane_stats.search_tile_launches;

        for (int i = 0; i < RPVG_HIP_EM_KERNELS; ++i) {

            stats_out->em_kernel[i].ms += lane_stats.em_kernel[i].ms;
            stats_out->em_kernel[i].launches += lane_stats.em_kernel[i].launches;
            stats_out->em_kernel[i].problems += lane_stats.em_kernel[i].problems;
            stats_out->em_kernel[i].iterations += lane_stats.em_kernel[i].iterations;
            stats_out->em_kernel[i].max_iterations += lane_stats.em_kernel[i].max_iterations;
            stats_out->em_kernel[i].alg_bytes += lane_stats.em_kernel[i].alg_bytes;
        }
    }

    // busy time: the union of the timed spans of all contexts (the contexts of a GPU share its clock)
    std::vector<std::pair<double, double> > spans;

    for (auto & ctx: contexts) {

        uint64_t count = 0;
        check(rpvg_hip_stats_intervals(ctx, 0, nullptr, nullptr, nullptr, &count), "rpvg_hip_stats_intervals");

        std::vector<double> start(count), stop(count);
        check(rpvg_hip_stats_intervals(ctx, count, start.data(), stop.data(), nullptr, &count), "rpvg_hip_stats_intervals");

        for (size_t i = 0; i < std::min<size_t>(count, start.size()); ++i) {

            spans.emplace_back(start[i], stop[i]);
        }
    }

    std::sort(spans.begin(), spans.end());

    double busy = 0;
    double begin = 0;
    double end = -1;

    for (auto & span: spans) {

        if (end < begin || span.first > end) {

            if (end >= begin) {

                busy += end - begin;
            }

            begin = span.first;
            end = span.second;

        } else {

            end = std::max(end, span.second);
        }
    }

    if (end >= begin) {

        busy += end - begin;
    }

    stats_out->busy_ms = busy;
}

void HipEngine::resetStats() const {

    check(rpvg_hip_stats_reset(context), "rpvg_hip_stats_reset");

    for (auto & lane_context: laneContexts()) {

        check(rpvg_hip_stats_reset(lane_context), "rpvg_hip_stats_reset");
    }
}

int HipEngine::deviceCount() {

    int count = 0;

    if (rpvg_hip_device_count(&count) != RPVG_HIP_OK) {

        return 0;
    }

    return count;
}

void HipEngine::check(const int status, const char * what) {

    if (status != RPVG_HIP_OK) {

        throw EngineError(std::string(what) + " failed (" + std::to_string(status) + "): " + rpvg_hip_last_error());
    }
}

FlatClusterRows::FlatClusterRows() : cluster_row_off(1, 0), cluster_path_off(1, 0), path_source_off(1, 0), row_grp_off(1, 0), grp_idx_off(1, 0), counts_fit(true), paths_fit16(true), sources_fit16(true), noise_fits16(true) {}

uint16_t FlatClusterRows::noiseIndex(const double noise) {

    uint64_t bits;
    static_assert(sizeof(bits) == sizeof(noise), "a double is 64 bits");
    std::memcpy(&bits, &noise, sizeof(bits));

    auto noise_index_it = noise_index.find(bits);

    if (noise_index_it != noise_index.end()) {

        return noise_index_it->second;
    }

    if (row_noise_table.size() >= 65536) {

        noise_fits16 = false;
        return 0;
    }

    const uint16_t index = row_noise_table.size();
    row_noise_table.emplace_back(noise);
    noise_index.emplace(bits, index);

    return index;
}

void FlatClusterRows::addCluster(const std::vector<ReadPathProbabilities> & cluster_probs, const std::vector<PathInfo> & paths) {

    assert(path_group_id.size() == cluster_path_off.back());

    for (auto & path: paths) {

        path_group_id.emplace_back(path.group_id);
        source_id.insert(source_id.end(), path.source_ids.begin(), path.source_ids.end());
        path_source_off.emplace_back(source_id.size());

        for (auto & id: path.source_ids) {

            sources_fit16 = sources_fit16 && id < 65536;
            source_id16.emplace_back(static_cast<uint16_t>(id));
        }
    }

    addCluster(cluster_probs, paths.size());
}

void FlatClusterRows::addCluster(const std::vector<ReadPathProbabilities> & cluster_probs, const uint32_t num_paths) {

    for (auto & probs: cluster_probs) {

        if (path_idx.size() > 0xF0000000ull) {

            throw EngineError("FlatClusterRows: a batch of more than 2^32 - 1 entries");
        }

        if (probs.readCount() >= 255) {

            row_count_escape_row.emplace_back(row_count.size());
            row_count_escape_count.emplace_back(probs.readCount());
        }

        row_count8.emplace_back(static_cast<uint8_t>(std::min<uint32_t>(probs.readCount(), 255)));
        row_count.emplace_back(probs.readCount());
        row_noise.emplace_back(probs.noiseProb());

        if (noise_fits16) {

            row_noise16.emplace_back(noiseIndex(probs.noiseProb()));
        }

        for (auto & path_probs: probs.pathProbs()) {

            grp_prob.emplace_back(path_probs.first);
            path_idx.insert(path_idx.end(), path_probs.second.begin(), path_probs.second.end());

            for (auto & path: path_probs.second) {

                path_idx16.emplace_back(static_cast<uint16_t>(path));
            }

            grp_idx_off.emplace_back(path_idx.size());
            grp_idx_count.emplace_back(static_cast<uint8_t>(path_probs.second.size()));
            counts_fit = counts_fit && path_probs.second.size() <= 255;
        }

        row_grp_off.emplace_back(grp_prob.size());
        row_grp_count.emplace_back(static_cast<uint8_t>(probs.pathProbs().size()));
        counts_fit = counts_fit && probs.pathProbs().size() <= 255;
    }

    paths_fit16 = paths_fit16 && num_paths < 65536;

    cluster_row_off.emplace_back(row_count.size());
    cluster_path_off.emplace_back(cluster_path_off.back() + num_paths);
}

void FlatClusterRows::append(const FlatClusterRows & other) {

    if (static_cast<uint64_t>(grp_prob.size()) + other.grp_prob.size() > 0xFFFFFFFFull || static_cast<uint64_t>(path_idx.size()) + other.path_idx.size() > 0xFFFFFFFFull) {

        throw EngineError("FlatClusterRows: a batch of more than 2^32 - 1 (probability, path list) groups or entries");
    }

    auto appendOffsets = [](auto & to, const auto & from) {

        const auto base = to.back();

        for (size_t i = 1; i < from.size(); ++i) {

            to.emplace_back(base + from[i]);
        }
    };

    appendOffsets(cluster_row_off, other.cluster_row_off);
    appendOffsets(cluster_path_off, other.cluster_path_off);
    appendOffsets(row_grp_off, other.row_grp_off);
    appendOffsets(grp_idx_off, other.grp_idx_off);
    appendOffsets(path_source_off, other.path_source_off);

    row_grp_count.insert(row_grp_count.end(), other.row_grp_count.begin(), other.row_grp_count.end());
    grp_idx_count.insert(grp_idx_count.end(), other.grp_idx_count.begin(), other.grp_idx_count.end());
    counts_fit = counts_fit && other.counts_fit;

    {
        const uint32_t first_row = row_count.size();

        for (size_t i = 0; i < other.row_count_escape_row.size(); ++i) {

            row_count_escape_row.emplace_back(first_row + other.row_count_escape_row[i]);
            row_count_escape_count.emplace_back(other.row_count_escape_count[i]);
        }

        row_count8.insert(row_count8.end(), other.row_count8.begin(), other.row_count8.end());
        path_idx16.insert(path_idx16.end(), other.path_idx16.begin(), other.path_idx16.end());
        source_id16.insert(source_id16.end(), other.source_id16.begin(), other.source_id16.end());
        paths_fit16 = paths_fit16 && other.paths_fit16;
        sources_fit16 = sources_fit16 && other.sources_fit16;

        for (size_t r = 0; noise_fits16 && r < other.row_noise.size(); ++r) {  // (the other's indices point into the other's table)

            row_noise16.emplace_back(noiseIndex(other.row_noise[r]));
        }
    }

    row_count.insert(row_count.end(), other.row_count.begin(), other.row_count.end());
    row_noise.insert(row_noise.end(), other.row_noise.begin(), other.row_noise.end());
    grp_prob.insert(grp_prob.end(), other.grp_prob.begin(), other.grp_prob.end());
    path_idx.insert(path_idx.end(), other.path_idx.begin(), other.path_idx.end());
    path_group_id.insert(path_group_id.end(), other.path_group_id.begin(), other.path_group_id.end());
    source_id.insert(source_id.end(), other.source_id.begin(), other.source_id.end());
}

rpvg_cluster_batch FlatClusterRows::view() const {

    rpvg_cluster_batch batch = {};

    batch.num_clusters = numClusters();
    batch.cluster_row_off = cluster_row_off.data();
    batch.cluster_path_off = cluster_path_off.data();

    batch.row_count = row_count.data();
    batch.row_noise = row_noise.data();
    batch.row_grp_off = nullptr;
    batch.row_grp_off32 = row_grp_off.data();
    batch.grp_prob = grp_prob.data();
    batch.grp_idx_off = nullptr;
    batch.grp_idx_off32 = grp_idx_off.data();
    batch.path_idx = path_idx.data();

    // what the copy to the GPU takes instead of the offsets while no row has more than 255 groups and no group more than 255 paths
    if (counts_fit && !grp_prob.empty()) {

        batch.row_grp_count8 = row_grp_count.data();
        batch.grp_idx_count8 = grp_idx_count.data();
        batch.num_groups = grp_prob.size();
        batch.num_entries = path_idx.size();
    }

    // the narrow forms, for the copy to the GPU, while they fit (the 32-bit arrays stay: whoever walks the batch on the host reads them)
    if (paths_fit16 && !path_idx.empty()) {

        batch.path_idx16 = path_idx16.data();
    }

    if (!row_count.empty() && row_count.size() < 0xFFFFFFFFull) {

        batch.row_count8 = row_count8.data();
        batch.row_count_escape_row = row_count_escape_row.data();
        batch.row_count_escape_count = row_count_escape_count.data();
        batch.num_row_count_escapes = row_count_escape_row.size();
    }

    if (noise_fits16 && row_noise16.size() == row_noise.size() && !row_noise.empty()) {

        batch.row_noise16 = row_noise16.data();
        batch.row_noise_table = row_noise_table.data();
        batch.num_row_noise_values = row_noise_table.size();
    }

    // the PathInfo fields the device reads, when the clusters were added with their paths; the rest of PathInfo stays on the
    // host side of the ABI (PathClusterEstimates::paths)
    const bool with_paths = !path_group_id.empty() && path_group_id.size() == cluster_path_off.back();

    batch.path_group_id = with_paths ? path_group_id.data() : nullptr;
    batch.path_source_count = nullptr;
    batch.path_source_off = with_paths ? path_source_off.data() : nullptr;
    batch.source_id = with_paths ? source_id.data() : nullptr;
    batch.source_id16 = (with_paths && sources_fit16 && !source_id.empty()) ? source_id16.data() : nullptr;
    batch.path_effective_length = nullptr;

    return batch;
}

std::vector<rpvg_hip_ctx *> HipEngine::laneContexts() const {

    std::vector<rpvg_hip_ctx *> contexts;

    for (auto & lane_context: lane_contexts) {

        if (rpvg_hip_ctx * made = lane_context.load(std::memory_order_acquire)) {

            contexts.emplace_back(made);
        }
    }

    return contexts;
}

PipelineWorker & HipEngine::lane(const int lane) {

    assert(lane >= 1 && lane < max_lanes);

    std::lock_guard<std::mutex> lock(lane_mutex);

    while (lane_workers.size() < static_cast<size_t>(lane)) {

        rpvg_hip_ctx * lane_context = nullptr;
        check(rpvg_hip_create(device_id, &lane_context), "rpvg_hip_create");

        lane_workers.emplace_back(new PipelineWorker([]() { RetiredContainers::ofThisThread().dropAll(); }));
        lane_contexts[lane_workers.size() - 1].store(lane_context, std::memory_order_release);
    }

    return *lane_workers.at(lane - 1);
}

int HipEngine::combinerLane(const int slot) {

    assert(slot >= 0 && slot < max_combiner_slots);

    const int lane = max_lanes + slot;

    if (!lane_contexts[lane - 1].load(std::memory_order_acquire)) {

        std::lock_guard<std::mutex> lock(lane_mutex);

        if (!lane_contexts[lane - 1].load(std::memory_order_acquire)) {

            rpvg_hip_ctx * slot_context = nullptr;
            check(rpvg_hip_create_with_streams(device_id, 3, &slot_context), "rpvg_hip_create_with_streams");
            lane_contexts[lane - 1].store(slot_context, std::memory_order_release);
        }
    }

    return lane;
}

DeviceClusterBatch::DeviceClusterBatch(std::shared_ptr<HipEngine> engine_in, const rpvg_cluster_batch & host_batch, const bool finish_later) : hip_engine(engine_in), batch(nullptr), unfinished_host_batch(host_batch), unfinished(finish_later), finish_queued(false) {

    assert(hip_engine);

    num_rows.resize(host_batch.num_clusters);
    num_paths.resize(host_batch.num_clusters);

    for (uint32_t i = 0; i < host_batch.num_clusters; ++i) {

        num_rows[i] = host_batch.cluster_row_off[i + 1] - host_batch.cluster_row_off[i];
        num_paths[i] = host_batch.cluster_path_off[i + 1] - host_batch.cluster_path_off[i];
    }

    if (finish_later) {

        ScopedPhase upload_phase("device batch: rpvg_hip_batch_upload_begin");
        HipEngine::check(rpvg_hip_batch_upload_begin(hip_engine->ctx(), &host_batch, &batch), "rpvg_hip_batch_upload_begin");
        return;
    }

    {
        ScopedPhase upload_phase("device batch: rpvg_hip_batch_upload");
        HipEngine::check(rpvg_hip_batch_upload(hip_engine->ctx(), &host_batch, &batch), "rpvg_hip_batch_upload");
    }

    // (the read count of every cluster comes back from the device with the upload: the sum over three million rows per batch was
    // a team of its own on the uploading thread)
    total_read_count.resize(host_batch.num_clusters);
    HipEngine::check(rpvg_hip_batch_cluster_totals(batch, total_read_count.data(), host_batch.num_clusters), "rpvg_hip_batch_cluster_totals");
}

ClusterSegment::ClusterSegment() : block(nullptr), capacity(0), segment() {}

ClusterSegment::~ClusterSegment() {

    rpvg_hip_pinned_free(block);
}

void ClusterSegment::flatten(const std::vector<ReadPathProbabilities> & cluster_probs, const std::vector<PathInfo> & paths, const bool with_sources_in, const Columns * columns) {

    const bool with_sources = with_sources_in && !columns;  // (the columns stand for the ids)

    uint64_t num_groups = 0, num_entries = 0, num_sources = 0, total_read_count = 0;

    for (auto & probs: cluster_probs) {

        num_groups += probs.pathProbs().size();
        total_read_count += probs.readCount();

        for (auto & path_probs: probs.pathProbs()) {

            num_entries += path_probs.second.size();
        }
    }

    for (size_t p = 0; with_sources && p < paths.size(); ++p) {

        num_sources += paths[p].source_ids.size();
    }

    if (cluster_probs.size() >= 0xFFFFFFFFull || num_groups >= 0xFFFFFFFFull || num_entries >= 0xFFFFFFFFull || num_sources >= 0xFFFFFFFFull) {

        throw EngineError("ClusterSegment: a cluster of 2^32 - 1 rows, groups, entries or source ids, or more");
    }

    const uint64_t R = cluster_probs.size(), G = num_groups, NNZ = num_entries, P = paths.size(), S = num_sources;

    // the arrays one behind the other, each from a multiple of 8 bytes
    uint64_t at = 0;

    auto place = [&at](const uint64_t count, const uint64_t width) {

        const uint64_t here = at;
        at += (count * width + 7) & ~7ull;
        return here;
    };

    segment = rpvg_cluster_segment();
    segment.row_noise_at = place(R, 8);
    segment.grp_prob_at = place(G, 8);
    segment.row_count_at = place(R, 4);
    segment.row_grp_off_at = place(R + 1, 4);
    segment.grp_idx_off_at = place(G + 1, 4);
    segment.path_idx_at = place(NNZ, 4);
    const uint64_t num_columns = columns ? columns->num : 0, num_column_paths = columns ? columns->path_off[columns->num] : 0;

    segment.path_group_id_at = place((with_sources || columns) ? P : 0, 4);
    segment.path_source_off_at = place(with_sources ? P + 1 : 0, 4);
    segment.source_id_at = place(S, 4);
    segment.col_count_at = place(num_columns, 4);
    segment.col_end_at = place(num_columns, 4);
    segment.col_path_at = place(num_column_paths, 4);
    segment.bytes = std::max<uint64_t>(at, 8);

    if (segment.bytes > capacity) {

        rpvg_hip_pinned_free(block);
        block = nullptr;
        capacity = 0;

        const uint64_t wanted = std::max<uint64_t>(segment.bytes + segment.bytes / 4, 1 << 16);
        HipEngine::check(rpvg_hip_pinned_alloc(wanted, &block), "rpvg_hip_pinned_alloc");
        capacity = wanted;
    }

    unsigned char * base = static_cast<unsigned char *>(block);

    double * row_noise = reinterpret_cast<double *>(base + segment.row_noise_at);
    double * grp_prob = reinterpret_cast<double *>(base + segment.grp_prob_at);
    uint32_t * row_count = reinterpret_cast<uint32_t *>(base + segment.row_count_at);
    uint32_t * row_grp_off = reinterpret_cast<uint32_t *>(base + segment.row_grp_off_at);
    uint32_t * grp_idx_off = reinterpret_cast<uint32_t *>(base + segment.grp_idx_off_at);
    uint32_t * path_idx = reinterpret_cast<uint32_t *>(base + segment.path_idx_at);

    uint32_t group = 0, entry = 0;

    for (size_t r = 0; r < R; ++r) {

        const ReadPathProbabilities & probs = cluster_probs[r];

        row_count[r] = probs.readCount();
        row_noise[r] = probs.noiseProb();
        row_grp_off[r] = group;

        for (auto & path_probs: probs.pathProbs()) {

            grp_prob[group] = path_probs.first;
            grp_idx_off[group] = entry;

            std::copy(path_probs.second.begin(), path_probs.second.end(), path_idx + entry);
            entry += path_probs.second.size();
            ++group;
        }
    }

    row_grp_off[R] = group;
    grp_idx_off[G] = entry;

    uint32_t * path_group_id = reinterpret_cast<uint32_t *>(base + segment.path_group_id_at);
    uint32_t * path_source_off = reinterpret_cast<uint32_t *>(base + segment.path_source_off_at);
    uint32_t * source_id = reinterpret_cast<uint32_t *>(base + segment.source_id_at);

    uint32_t source = 0;

    for (size_t p = 0; with_sources && p < P; ++p) {

        path_group_id[p] = paths[p].group_id;
        path_source_off[p] = source;

        std::copy(paths[p].source_ids.begin(), paths[p].source_ids.end(), source_id + source);
        source += paths[p].source_ids.size();
    }

    if (with_sources) {

        path_source_off[P] = source;
    }

    if (columns) {

        for (size_t p = 0; p < P; ++p) {

            path_group_id[p] = paths[p].group_id;
        }

        uint32_t * col_count = reinterpret_cast<uint32_t *>(base + segment.col_count_at);
        uint32_t * col_end = reinterpret_cast<uint32_t *>(base + segment.col_end_at);
        uint32_t longest = 0;

        for (uint32_t c = 0; c < columns->num; ++c) {

            col_count[c] = columns->counts[c];
            col_end[c] = columns->path_off[c + 1];
            longest = std::max(longest, columns->path_off[c + 1] - columns->path_off[c]);
        }

        std::copy(columns->paths, columns->paths + num_column_paths, reinterpret_cast<uint32_t *>(base + segment.col_path_at));

        segment.has_columns = 1;
        segment.num_columns = columns->num;
        segment.num_column_paths = num_column_paths;
        segment.max_column_paths = longest;
    }

    segment.base = block;
    segment.num_rows = R;
    segment.num_groups = G;
    segment.num_entries = NNZ;
    segment.num_paths = P;
    segment.num_sources = S;
    segment.has_paths = with_sources ? 1 : 0;
    segment.total_read_count = total_read_count;
}

DeviceClusterBatch::DeviceClusterBatch(std::shared_ptr<HipEngine> engine_in, const std::vector<rpvg_cluster_segment> & segments) : hip_engine(engine_in), batch(nullptr), unfinished_host_batch(), unfinished(false), finish_queued(false) {

    assert(hip_engine);

    ScopedPhase upload_phase("device batch: rpvg_hip_batch_upload_segments");
    HipEngine::check(rpvg_hip_batch_upload_segments(hip_engine->ctx(), segments.data(), segments.size(), &batch), "rpvg_hip_batch_upload_segments");

    num_rows.reserve(segments.size());
    num_paths.reserve(segments.size());
    total_read_count.reserve(segments.size());

    for (auto & segment: segments) {

        num_rows.emplace_back(segment.num_rows);
        num_paths.emplace_back(segment.num_paths);
        total_read_count.emplace_back(static_cast<double>(segment.total_read_count));
    }
}

void DeviceClusterBatch::queueFinish() {

    assert(unfinished && batch && !finish_queued);

    ScopedPhase queue_phase("device batch: rpvg_hip_batch_upload_finish_queue");

    rpvg_hip_batch * unfinished_batch = batch;
    batch = nullptr;  // (a batch that fails is freed by the call)

    HipEngine::check(rpvg_hip_batch_upload_finish_queue(hip_engine->ctx(), unfinished_batch, &unfinished_host_batch), "rpvg_hip_batch_upload_finish_queue");
    batch = unfinished_batch;
    finish_queued = true;
}

void DeviceClusterBatch::finish(std::shared_ptr<HipEngine> engine_in) {

    assert(unfinished && batch);
    hip_engine = engine_in;

    ScopedPhase finish_phase("device batch: rpvg_hip_batch_upload_finish");

    rpvg_hip_batch * unfinished_batch = batch;
    batch = nullptr;  // (a batch that fails its second half is freed by the call)
    unfinished = false;

    if (finish_queued) {

        HipEngine::check(rpvg_hip_batch_upload_finish_wait(unfinished_batch, &unfinished_host_batch), "rpvg_hip_batch_upload_finish_wait");

    } else {

        HipEngine::check(rpvg_hip_batch_upload_finish(hip_engine->ctx(), unfinished_batch, &unfinished_host_batch), "rpvg_hip_batch_upload_finish");
    }

    batch = unfinished_batch;

    total_read_count.resize(num_rows.size());
    HipEngine::check(rpvg_hip_batch_cluster_totals(batch, total_read_count.data(), num_rows.size()), "rpvg_hip_batch_cluster_totals");
}

DeviceClusterBatch::DeviceClusterBatch(std::shared_ptr<HipEngine> engine_in, rpvg_hip_batch * device_batch, const rpvg_cluster_batch & offsets, const std::vector<double> & total_read_count_in) : hip_engine(engine_in), batch(device_batch), total_read_count(total_read_count_in), unfinished_host_batch(offsets), unfinished(false), finish_queued(false) {

    assert(hip_engine);
    assert(batch);
    assert(total_read_count.size() == offsets.num_clusters);

    for (uint32_t i = 0; i < offsets.num_clusters; ++i) {

        num_rows.emplace_back(offsets.cluster_row_off[i + 1] - offsets.cluster_row_off[i]);
        num_paths.emplace_back(offsets.cluster_path_off[i + 1] - offsets.cluster_path_off[i]);
    }
}

DeviceClusterBatch::~DeviceClusterBatch() {

    if (batch) {

        rpvg_hip_batch_free(hip_engine->ctx(), batch);
    }
}

}
