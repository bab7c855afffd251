#include "read_rows.hpp"

#include <algorithm>
#include <cassert>
#include <cmath>
#include <limits>

#include "numeric_utils.hpp"

namespace rpvg_amd {

namespace {

const double pi = 3.141592653589793238462643383279;

// src/utils.hpp:143-161
double standardNormalCdf(const double z) {

    static const double root_1_2 = std::sqrt(0.5);

    const double x = z * root_1_2;
    const double a = std::fabs(x);

    if (a < root_1_2) {

        return 0.5 + 0.5 * std::erf(x);
    }

    const double y = 0.5 * std::erfc(a);
    return (x > 0) ? 1.0 - y : y;
}

// src/utils.hpp:165-194
double logStandardNormalCdf(const double z) {

    if (z > 6.0) {

        return -standardNormalCdf(-z);
    }

    if (z > -20.0) {

        return std::log(standardNormalCdf(z));
    }

    const double log_lhs = -0.5 * z * z - std::log(-z) - 0.5 * std::log(2 * pi);

    double last_total = 0;
    double right_hand_side = 1;
    double numerator = 1;
    double denom_factor = 1;

    const double denom_cons = 1.0 / (z * z);

    long sign = 1;
    long i = 0;

    while (std::fabs(last_total - right_hand_side) > std::numeric_limits<double>::epsilon()) {

        i += 1;
        last_total = right_hand_side;
        sign = -sign;
        denom_factor *= denom_cons;
        numerator *= 2 * i - 1;
        right_hand_side += sign * numerator * denom_factor;
    }

    return log_lhs + std::log(right_hand_side);
}

// src/utils.hpp:206-220
double logDensity(const double x, const double loc, const double scale, const double shape) {

    const double z = (x - loc) / scale;

    if (numeric::doubleCompare(shape, 0.0)) {

        static const double inv_sqrt_2pi = 0.3989422804014327;
        return std::log(inv_sqrt_2pi) - std::log(scale) - 0.5 * z * z;
    }

    static const double log_const = std::log(2.0 / std::sqrt(2.0 * pi));
    return log_const + logStandardNormalCdf(shape * z) - std::log(scale) - 0.5 * z * z;
}

}

FragmentLengthDist::FragmentLengthDist() : loc_(0), scale_(0), shape_(0), max_length_(0) {}

FragmentLengthDist::FragmentLengthDist(const double mean_in, const double sd_in, const uint32_t sd_max_multi) : FragmentLengthDist(mean_in, sd_in, 0.0, sd_max_multi) {}

// src/fragment_length_dist.cpp:21-27,396-427
FragmentLengthDist::FragmentLengthDist(const double loc_in, const double scale_in, const double shape_in, const uint32_t sd_max_multi) : loc_(loc_in), scale_(scale_in), shape_(shape_in) {

    assert(isValid());

    const double delta = shape_ / std::sqrt(1.0 + shape_ * shape_);
    const double sd = scale_ * (1.0 - 2.0 * delta * delta / pi);

    max_length_ = std::ceil(loc_ + sd * sd_max_multi);
    assert(max_length_ > 0);

    log_prob_buffer.resize(static_cast<size_t>(max_length_) + 1);

    for (size_t i = 0; i < log_prob_buffer.size(); ++i) {

        log_prob_buffer[i] = logDensity(i, loc_, scale_, shape_);
    }
}

bool FragmentLengthDist::isValid() const {

    return (loc_ >= 0 && scale_ > 0);
}

// src/fragment_length_dist.cpp:385-394
double FragmentLengthDist::logProb(const uint32_t value) const {

    if (value < log_prob_buffer.size()) {

        return log_prob_buffer[value];
    }

    return logDensity(value, loc_, scale_, shape_);
}

std::vector<double> FragmentLengthDist::logProbTable() const {

    std::vector<double> table(RPVG_FRAG_LENGTH_TABLE_SIZE);

    for (uint32_t value = 0; value < RPVG_FRAG_LENGTH_TABLE_SIZE; ++value) {

        table[value] = logProb(value);
    }

    return table;
}

AlignmentBatchBuilder::AlignmentBatchBuilder() : collapse(false), cluster_read_off(1, 0), cluster_path_off(1, 0), cluster_group_off(1, 0), read_align_off(1, 0), align_path_off(1, 0) {}

void AlignmentBatchBuilder::beginCluster(const std::vector<PathInfo> & cluster_paths, const std::vector<uint32_t> & group_name_index, const uint32_t num_groups) {

    if (!group_name_index.empty()) {

        assert(group_name_index.size() == cluster_paths.size());
        assert(collapse || cluster_path_off.back() == 0);
        collapse = true;

    } else {

        assert(!collapse);
    }

    for (size_t i = 0; i < cluster_paths.size(); ++i) {

        path_effective_length.emplace_back(cluster_paths[i].effective_length);
        path_source_count.emplace_back(cluster_paths[i].source_count);

        if (collapse) {

            assert(group_name_index[i] < num_groups);
            path_group.emplace_back(group_name_index[i]);
        }
    }

    cluster_path_off.emplace_back(path_effective_length.size());
    cluster_group_off.emplace_back(cluster_group_off.back() + num_groups);
    cluster_read_off.emplace_back(cluster_read_off.back());
    cluster_total_reads.emplace_back(0);
}

void AlignmentBatchBuilder::addAlignmentPaths(const std::vector<AlignmentPath> & align_paths, const uint32_t read_count_in) {

    assert(cluster_read_off.size() > 1);

    // src/read_path_probabilities.cpp:41-46
    assert(align_paths.size() > 1);
    assert(align_paths.back().path_idx.empty());
    assert(align_paths.back().frag_length == 0);
    assert(align_paths.back().align_length == 0);
    assert(align_paths.back().score_sum <= 0);

    read_count.emplace_back(read_count_in);
    read_min_mapq.emplace_back(align_paths.front().min_mapq);
    read_noise_score.emplace_back(align_paths.back().score_sum);

    for (size_t i = 0; i + 1 < align_paths.size(); ++i) {

        const auto & align_path = align_paths[i];
        assert(align_paths.front().min_mapq == align_path.min_mapq);

        align_score_sum.emplace_back(align_path.score_sum);
        align_length.emplace_back(align_path.align_length);
        align_frag_length.emplace_back(align_path.frag_length);

        // the result does not depend on the order of the ids of one alignment; the GPU path wants them ascending
        const size_t first = align_path_idx.size();
        align_path_idx.insert(align_path_idx.end(), align_path.path_idx.begin(), align_path.path_idx.end());
        std::sort(align_path_idx.begin() + first, align_path_idx.end());

        align_path_off.emplace_back(align_path_idx.size());
    }

    read_align_off.emplace_back(align_score_sum.size());
    cluster_read_off.back()++;
    cluster_total_reads.back() += read_count_in;
}

rpvg_alignment_batch AlignmentBatchBuilder::view() const {

    rpvg_alignment_batch batch;

    batch.num_clusters = numClusters();
    batch.cluster_read_off = cluster_read_off.data();
    batch.cluster_path_off = cluster_path_off.data();
    batch.path_effective_length = path_effective_length.data();
    batch.path_source_count = path_source_count.data();
    batch.path_group = collapse ? path_group.data() : nullptr;
    batch.cluster_group_off = collapse ? cluster_group_off.data() : nullptr;
    batch.read_count = read_count.data();
    batch.read_min_mapq = read_min_mapq.data();
    batch.read_noise_score = read_noise_score.data();
    batch.read_align_off = read_align_off.data();
    batch.align_score_sum = align_score_sum.data();
    batch.align_length = align_length.data();
    batch.align_frag_length = align_frag_length.data();
    batch.align_path_off = align_path_off.data();
    batch.align_path_idx = align_path_idx.data();

    return batch;
}

DeviceAlignmentBatch::DeviceAlignmentBatch(std::shared_ptr<HipEngine> engine_in, const AlignmentBatchBuilder & alignments) : hip_engine(engine_in), device_alignments(nullptr) {

    assert(hip_engine);

    const auto alignment_batch = alignments.view();
    HipEngine::check(rpvg_hip_alignments_upload(hip_engine->ctx(), &alignment_batch, &device_alignments), "rpvg_hip_alignments_upload");

    for (uint32_t i = 0; i < alignments.numClusters(); ++i) {

        total_read_count.emplace_back(alignments.totalReadCount(i));
    }
}

DeviceAlignmentBatch::~DeviceAlignmentBatch() {

    rpvg_hip_alignments_free(hip_engine->ctx(), device_alignments);
}

std::unique_ptr<DeviceClusterBatch> constructReadPathProbabilities(const DeviceAlignmentBatch & alignments, const FragmentLengthDist & fragment_length_dist, const bool is_single_end, const double min_noise_prob, const double prob_precision) {

    const auto & engine = alignments.engine();

    std::vector<double> frag_length_table;

    rpvg_row_params params;
    params.prob_precision = prob_precision;
    params.min_noise_prob = min_noise_prob;
    params.is_single_end = is_single_end;
    params.frag_length_log_prob = nullptr;

    if (!is_single_end) {

        frag_length_table = fragment_length_dist.logProbTable();
        params.frag_length_log_prob = frag_length_table.data();
    }

    rpvg_hip_read_rows * rows = nullptr;
    HipEngine::check(rpvg_hip_read_rows_build(engine->ctx(), alignments.handle(), &params, 1, &rows), "rpvg_hip_read_rows_build");

    rpvg_hip_batch * batch = nullptr;
    const int status = rpvg_hip_read_rows_to_batch(engine->ctx(), rows, &batch);

    rpvg_cluster_batch rows_view = {};
    const int view_status = (status == 0) ? rpvg_hip_read_rows_sizes(engine->ctx(), rows, &rows_view) : 0;

    std::unique_ptr<DeviceClusterBatch> cluster_batch;

    if (status == 0 && view_status == 0) {

        cluster_batch.reset(new DeviceClusterBatch(engine, batch, rows_view, alignments.totalReadCounts()));
    }

    rpvg_hip_read_rows_free(engine->ctx(), rows);

    HipEngine::check(status, "rpvg_hip_read_rows_to_batch");
    HipEngine::check(view_status, "rpvg_hip_read_rows_sizes");

    return cluster_batch;
}

std::unique_ptr<DeviceClusterBatch> constructReadPathProbabilities(std::shared_ptr<HipEngine> engine, const AlignmentBatchBuilder & alignments, const FragmentLengthDist & fragment_length_dist, const bool is_single_end, const double min_noise_prob, const double prob_precision) {

    const DeviceAlignmentBatch device_alignments(engine, alignments);
    return constructReadPathProbabilities(device_alignments, fragment_length_dist, is_single_end, min_noise_prob, prob_precision);
}

}
