// Synthetic pantranscriptome workload (bench / test harness, not part of the
// estimators): K independent path clusters shaped like the "10M read pairs x
// 200k paths in ~5k clusters" configuration of BASELINE.json (SURVEY.md §8d,
// S3).  Produces exactly what the reference's per-cluster loop hands to
// estimate() (src/main.cpp:846-973): PathInfo of the cluster's paths and the
// sorted + merged ReadPathProbabilities rows.
//
// Model of one cluster
//   - N_k paths (haplotype-specific transcripts, HSTs) in T_k transcripts
//     (PathInfo::group_id), at most `num_haplotypes` HSTs per transcript;
//   - `num_haplotypes` haplotype ids; every haplotype carries exactly one HST of
//     every transcript (PathInfo::source_ids / source_count), allele
//     frequencies skewed towards the first HSTs of a transcript;
//   - the sample is one diplotype (two haplotypes); reads pick a transcript by
//     expression, one of the two alleles, and are compatible with the true HST
//     plus Geometric(1/2) sibling HSTs of the transcript, each either score-tied
//     (probability tie_prob: the read does not cover a distinguishing variant)
//     or at a deficit of 1 + Poisson(3) alignment-score units (capped at 20);
//     likelihood exp(-score_log_base * deficit) / effective_length;
//   - mapping quality in {60: 70 %, 30: 15 %, 10: 10 %, 3: 5 %} gives the noise
//     probability max(1e-4, 10^(-mapq/10)) (src/read_path_probabilities.cpp:91);
//   - rows are finished like addPathProbs and merged like the caller does.
// Every cluster has its own counter-seeded generator, so the output does not
// depend on the number of threads.  Clusters are emitted in descending order
// of read count, the order the reference processes them in (src/main.cpp:811-827).

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstring>
#include <map>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/rpvg_batch.h"
#include "path_cluster_estimates.hpp"
#include "read_path_probabilities.hpp"
#include "flat_batch.hpp"

#include "../../include/rpvg_rows.h"

using namespace rpvg_amd;

extern "C" {

typedef struct rpvg_synth_config {
    uint64_t seed;
    uint32_t num_clusters;        /* 5000 */
    uint64_t total_paths;         /* 200000 */
    uint64_t total_reads;         /* 10000000 */
    uint32_t num_haplotypes;      /* 64 */
    uint32_t max_cluster_paths;   /* 4000 */
    double cluster_paths_sigma;   /* 1.0  log-normal sigma of paths per cluster */
    double read_mass_sigma;       /* 1.5  log-normal sigma of reads per cluster */
    double tie_prob;              /* 0.3 */
    double pathless_read_frac;    /* 0.005 reads without any compatible path (noise = 1) */
    uint32_t keep_alignments;     /* 0; 1 = also keep the reads as alignment-path lists (include/rpvg_rows.h) */
} rpvg_synth_config;

}

namespace {

const double score_log_base = 1.383325268738;  // Utils::score_log_base, src/utils.hpp:83

// xoshiro256** seeded through splitmix64
struct Rng {

    uint64_t s[4];

    static uint64_t splitmix(uint64_t & x) {

        uint64_t z = (x += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }

    explicit Rng(uint64_t seed) {

        for (auto & v: s) {

            v = splitmix(seed);
        }
    }

    static uint64_t rotl(const uint64_t x, const int k) { return (x << k) | (x >> (64 - k)); }

    uint64_t next() {

        const uint64_t result = rotl(s[1] * 5, 7) * 9;
        const uint64_t t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
        s[2] ^= t;
        s[3] = rotl(s[3], 45);
        return result;
    }

    double uniform() { return ((next() >> 11) + 0.5) * (1.0 / 9007199254740992.0); }
    uint32_t below(const uint32_t n) { return static_cast<uint32_t>(uniform() * n) % n; }
    double normal() { return std::sqrt(-2.0 * std::log(uniform())) * std::cos(6.283185307179586 * uniform()); }
    double logNormal(const double sigma) { return std::exp(sigma * normal()); }

    uint32_t poisson3() {

        // inverse CDF, lambda = 3
        const double u = uniform();
        double p = std::exp(-3.0), cdf = p;
        uint32_t k = 0;

        while (u > cdf && k < 40) {

            ++k;
            p *= 3.0 / k;
            cdf += p;
        }

        return k;
    }

    uint32_t geometricHalf() {  // number of failures before the first success, p = 1/2

        uint32_t k = 0;

        while (uniform() < 0.5 && k < 64) {

            ++k;
        }

        return k;
    }
};

// The distinct reads of a cluster as the alignment-path lists row construction starts from (include/rpvg_rows.h):
// one alignment per distinct score deficit of the read, holding the paths with that deficit.
struct SynthAlignments {

    std::vector<uint32_t> read_count;
    std::vector<uint8_t> read_min_mapq;
    std::vector<uint64_t> read_align_off = std::vector<uint64_t>(1, 0);
    std::vector<int32_t> align_score_sum;
    std::vector<uint64_t> align_path_off = std::vector<uint64_t>(1, 0);
    std::vector<uint32_t> align_path_idx;
};

struct SynthCluster {

    std::vector<PathInfo> paths;
    std::vector<ReadPathProbabilities> rows;
    SynthAlignments alignments;
    uint64_t num_reads;
};

const int32_t synth_best_score = 100;
const uint16_t synth_align_length = 100;
const uint16_t synth_frag_length = 300;

// Splits `total` into weights-proportional non-negative integers summing to total.
std::vector<uint64_t> apportion(const std::vector<double> & weights, const uint64_t total, const uint64_t minimum) {

    const double weight_sum = std::accumulate(weights.begin(), weights.end(), 0.0);
    std::vector<uint64_t> parts(weights.size());
    std::vector<std::pair<double, size_t> > remainders(weights.size());

    const uint64_t spread = total - minimum * weights.size();
    uint64_t assigned = 0;

    for (size_t i = 0; i < weights.size(); ++i) {

        const double share = weights[i] / weight_sum * spread;
        parts[i] = static_cast<uint64_t>(std::floor(share));
        remainders[i] = std::make_pair(share - parts[i], i);
        assigned += parts[i];
    }

    std::sort(remainders.begin(), remainders.end(), [](const std::pair<double, size_t> & a, const std::pair<double, size_t> & b) { return a.first != b.first ? a.first > b.first : a.second < b.second; });

    for (uint64_t i = 0; i < spread - assigned; ++i) {

        parts[remainders[i % remainders.size()].second]++;
    }

    for (auto & part: parts) {

        part += minimum;
    }

    return parts;
}

void generateCluster(SynthCluster * cluster, const rpvg_synth_config & config, const uint32_t cluster_idx, const uint32_t num_paths, const uint64_t num_reads) {

    Rng rng(config.seed * 0x9E3779B97F4A7C15ull + cluster_idx * 0xD1B54A32D192ED03ull + 1);

    const uint32_t num_haplotypes = config.num_haplotypes;

    // transcripts: 1-8 for ordinary clusters, more only to keep HSTs per transcript <= haplotypes
    uint32_t num_transcripts = std::max<uint32_t>(1, std::min<uint32_t>(8, std::lround(num_paths / (3.0 + 9.0 * rng.uniform()))));
    num_transcripts = std::max<uint32_t>(num_transcripts, (num_paths + num_haplotypes - 1) / num_haplotypes);
    num_transcripts = std::min(num_transcripts, num_paths);

    std::vector<double> split_weights(num_transcripts);

    for (auto & weight: split_weights) {

        weight = 0.5 + rng.uniform();
    }

    auto hst_counts = apportion(split_weights, num_paths, 1);

    // repair the rare transcript that ends above the haplotype count
    for (size_t t = 0; t < hst_counts.size(); ++t) {

        while (hst_counts[t] > num_haplotypes) {

            const size_t smallest = std::min_element(hst_counts.begin(), hst_counts.end()) - hst_counts.begin();
            hst_counts[t]--;
            hst_counts[smallest]++;
        }
    }

    cluster->paths.assign(num_paths, PathInfo());
    std::vector<uint32_t> transcript_first(num_transcripts + 1, 0);
    std::vector<std::vector<uint32_t> > haplotype_hst(num_transcripts, std::vector<uint32_t>(num_haplotypes));

    for (uint32_t t = 0; t < num_transcripts; ++t) {

        transcript_first[t + 1] = transcript_first[t] + hst_counts[t];
        const uint32_t n_hst = hst_counts[t];

        std::vector<uint32_t> haplotype_order(num_haplotypes);
        std::iota(haplotype_order.begin(), haplotype_order.end(), 0);

        for (uint32_t i = num_haplotypes - 1; i > 0; --i) {

            std::swap(haplotype_order[i], haplotype_order[rng.below(i + 1)]);
        }

        // harmonic allele frequencies over the transcript's HSTs
        std::vector<double> hst_cdf(n_hst);
        double acc = 0;

        for (uint32_t j = 0; j < n_hst; ++j) {

            acc += 1.0 / (j + 1);
            hst_cdf[j] = acc;
        }

        for (uint32_t i = 0; i < num_haplotypes; ++i) {

            uint32_t hst = i;  // the first n_hst haplotypes make sure every HST is carried

            if (i >= n_hst) {

                const double u = rng.uniform() * acc;
                hst = std::lower_bound(hst_cdf.begin(), hst_cdf.end(), u) - hst_cdf.begin();
                hst = std::min(hst, n_hst - 1);
            }

            haplotype_hst[t][haplotype_order[i]] = transcript_first[t] + hst;
        }

        for (uint32_t j = transcript_first[t]; j < transcript_first[t + 1]; ++j) {

            cluster->paths[j].group_id = t;
            cluster->paths[j].length = 200 + rng.below(4800);
            cluster->paths[j].effective_length = cluster->paths[j].length;
        }

        for (uint32_t h = 0; h < num_haplotypes; ++h) {

            cluster->paths[haplotype_hst[t][h]].source_ids.insert(h);
        }
    }

    for (auto & path: cluster->paths) {

        path.source_count = std::max<size_t>(1, path.source_ids.size());
    }

    // the sample's diplotype, expression and allelic ratio
    const uint32_t hap_1 = rng.below(num_haplotypes);
    const uint32_t hap_2 = rng.below(num_haplotypes);

    std::vector<double> expression_cdf(num_transcripts);
    double expression_sum = 0;

    for (uint32_t t = 0; t < num_transcripts; ++t) {

        expression_sum += rng.logNormal(1.0);
        expression_cdf[t] = expression_sum;
    }

    const double allele_ratio = 0.3 + 0.4 * rng.uniform();

    static const double mapq_noise[4] = {1e-4, 1e-3, 0.1, 0.50118723362727224};

    // distinct read signatures -> multiplicity; signature = mapq class, then (path, deficit) pairs sorted by path
    std::map<std::vector<uint32_t>, uint32_t> signatures;
    std::vector<uint32_t> signature;
    std::vector<std::pair<uint32_t, uint32_t> > compat;

    cluster->num_reads = num_reads;

    for (uint64_t r = 0; r < num_reads; ++r) {

        signature.clear();

        if (rng.uniform() < config.pathless_read_frac) {

            signature.push_back(0xFFFFFFFFu);
            signatures[signature]++;
            continue;
        }

        const double ue = rng.uniform() * expression_sum;
        const uint32_t t = std::min<uint32_t>(num_transcripts - 1, std::lower_bound(expression_cdf.begin(), expression_cdf.end(), ue) - expression_cdf.begin());
        const uint32_t true_path = haplotype_hst[t][rng.uniform() < allele_ratio ? hap_1 : hap_2];

        const double um = rng.uniform();
        const uint32_t mapq_class = (um < 0.70) ? 0 : (um < 0.85) ? 1 : (um < 0.95) ? 2 : 3;

        const uint32_t n_hst = transcript_first[t + 1] - transcript_first[t];
        const uint32_t num_siblings = std::min<uint32_t>(n_hst - 1, rng.geometricHalf());

        compat.clear();
        compat.emplace_back(true_path, 0);

        while (compat.size() < num_siblings + 1) {

            const uint32_t sibling = transcript_first[t] + rng.below(n_hst);
            bool seen = false;

            for (auto & c: compat) {

                seen = seen || (c.first == sibling);
            }

            if (!seen) {

                const uint32_t deficit = (rng.uniform() < config.tie_prob) ? 0 : std::min<uint32_t>(20, 1 + rng.poisson3());
                compat.emplace_back(sibling, deficit);
            }
        }

        std::sort(compat.begin(), compat.end());
        signature.push_back(mapq_class);

        for (auto & c: compat) {

            signature.push_back(c.first);
            signature.push_back(c.second);
        }

        signatures[signature]++;
    }

    if (config.keep_alignments) {

        // mapq of the four noise classes: phred_to_prob(mapq) = mapq_noise[class] (src/utils.hpp:131-133)
        static const uint8_t class_mapq[4] = {40, 30, 10, 3};

        auto & alignments = cluster->alignments;
        std::vector<uint32_t> deficits;

        for (auto & sig: signatures) {

            alignments.read_count.emplace_back(sig.second);

            if (sig.first.front() == 0xFFFFFFFFu) {

                // no compatible path: a read whose best alignment has mapq 0 contributes noise only
                // (src/read_path_probabilities.cpp:89)
                alignments.read_min_mapq.emplace_back(0);
                alignments.align_score_sum.emplace_back(synth_best_score);
                alignments.align_path_idx.emplace_back(0);
                alignments.align_path_off.emplace_back(alignments.align_path_idx.size());
                alignments.read_align_off.emplace_back(alignments.align_score_sum.size());
                continue;
            }

            alignments.read_min_mapq.emplace_back(class_mapq[sig.first.front()]);
            deficits.clear();

            for (size_t i = 1; i < sig.first.size(); i += 2) {

                deficits.emplace_back(sig.first[i + 1]);
            }

            std::sort(deficits.begin(), deficits.end());
            deficits.erase(std::unique(deficits.begin(), deficits.end()), deficits.end());

            for (auto & deficit: deficits) {

                alignments.align_score_sum.emplace_back(synth_best_score - static_cast<int32_t>(deficit));

                for (size_t i = 1; i < sig.first.size(); i += 2) {

                    if (sig.first[i + 1] == deficit) {

                        alignments.align_path_idx.emplace_back(sig.first[i]);
                    }
                }

                alignments.align_path_off.emplace_back(alignments.align_path_idx.size());
            }

            alignments.read_align_off.emplace_back(alignments.align_score_sum.size());
        }
    }

    cluster->rows.reserve(signatures.size());
    std::vector<std::pair<uint32_t, double> > likelihoods;

    for (auto & sig: signatures) {

        if (sig.first.front() == 0xFFFFFFFFu) {

            cluster->rows.emplace_back(sig.second, 1.0, ReadPathProbabilities::PathProbs(), 1e-8);
            continue;
        }

        likelihoods.clear();

        for (size_t i = 1; i < sig.first.size(); i += 2) {

            likelihoods.emplace_back(sig.first[i], std::exp(-score_log_base * sig.first[i + 1]) / cluster->paths[sig.first[i]].effective_length);
        }

        cluster->rows.emplace_back(ReadPathProbabilities::fromPathLikelihoods(sig.second, mapq_noise[sig.first.front()], likelihoods, 1e-8));
    }

    sortAndMergeReadPathProbabilities(&cluster->rows);
}

struct SynthBatch : public FlatBatchStorage {

    // keep_alignments: the same reads as alignment-path lists, all clusters back to back
    std::vector<uint64_t> cluster_read_off = std::vector<uint64_t>(1, 0);
    SynthAlignments alignments;
    std::vector<int32_t> read_noise_score;
    std::vector<uint16_t> align_length, align_frag_length;
};

}

extern "C" {

rpvg_synth_config rpvg_amd_synth_default_config(void) {

    rpvg_synth_config config;
    config.seed = 3;
    config.num_clusters = 5000;
    config.total_paths = 200000;
    config.total_reads = 10000000;
    config.num_haplotypes = 64;
    config.max_cluster_paths = 4000;
    config.cluster_paths_sigma = 1.0;
    config.read_mass_sigma = 1.5;
    config.tie_prob = 0.3;
    config.pathless_read_frac = 0.005;
    config.keep_alignments = 0;
    return config;
}

void * rpvg_amd_synth_generate(const rpvg_synth_config * config_in) {

    rpvg_synth_config config = *config_in;
    const uint32_t K = config.num_clusters;

    assert(K > 0 && config.total_paths >= K && config.num_haplotypes >= 2);

    // the clip must leave room for all paths
    config.max_cluster_paths = std::max<uint64_t>(config.max_cluster_paths, (config.total_paths + K - 1) / K);

    // sizes: paths per cluster and reads per cluster, from one global stream
    Rng sizes_rng(config.seed ^ 0x5151515151515151ull);

    std::vector<double> path_weights(K), read_weights(K);

    for (uint32_t k = 0; k < K; ++k) {

        path_weights[k] = sizes_rng.logNormal(config.cluster_paths_sigma);
        read_weights[k] = sizes_rng.logNormal(config.read_mass_sigma);
    }

    // clip the path counts to [1, max_cluster_paths] while keeping the total
    auto num_paths = apportion(path_weights, config.total_paths, 1);

    for (int round = 0; round < 8; ++round) {

        uint64_t excess = 0;

        for (auto & n: num_paths) {

            if (n > config.max_cluster_paths) {

                excess += n - config.max_cluster_paths;
                n = config.max_cluster_paths;
            }
        }

        if (excess == 0) {

            break;
        }

        std::vector<double> room(K);

        for (uint32_t k = 0; k < K; ++k) {

            room[k] = (num_paths[k] < config.max_cluster_paths) ? path_weights[k] : 0.0;
        }

        const auto extra = apportion(room, excess, 0);

        for (uint32_t k = 0; k < K; ++k) {

            num_paths[k] += extra[k];
        }
    }

    const auto num_reads = apportion(read_weights, config.total_reads, 0);

    // the reference processes clusters in descending read-count order (src/main.cpp:811-827)
    std::vector<uint32_t> order(K);
    std::iota(order.begin(), order.end(), 0);
    std::sort(order.begin(), order.end(), [&](const uint32_t a, const uint32_t b) { return num_reads[a] != num_reads[b] ? num_reads[a] > num_reads[b] : a > b; });

    std::vector<SynthCluster> clusters(K);

    #pragma omp parallel for schedule(dynamic, 1)
    for (uint32_t i = 0; i < K; ++i) {

        generateCluster(&clusters[i], config, order[i], num_paths[order[i]], num_reads[order[i]]);
    }

    SynthBatch * batch = new SynthBatch();

    for (auto & cluster: clusters) {

        batch->addCluster(cluster.paths, cluster.rows);
        std::vector<ReadPathProbabilities>().swap(cluster.rows);

        if (config.keep_alignments) {

            auto & all = batch->alignments;
            const auto & own = cluster.alignments;

            const uint64_t first_align = all.align_score_sum.size();
            const uint64_t first_entry = all.align_path_idx.size();

            all.read_count.insert(all.read_count.end(), own.read_count.begin(), own.read_count.end());
            all.read_min_mapq.insert(all.read_min_mapq.end(), own.read_min_mapq.begin(), own.read_min_mapq.end());
            all.align_score_sum.insert(all.align_score_sum.end(), own.align_score_sum.begin(), own.align_score_sum.end());
            all.align_path_idx.insert(all.align_path_idx.end(), own.align_path_idx.begin(), own.align_path_idx.end());

            for (size_t i = 1; i < own.read_align_off.size(); ++i) {

                all.read_align_off.emplace_back(first_align + own.read_align_off[i]);
            }

            for (size_t i = 1; i < own.align_path_off.size(); ++i) {

                all.align_path_off.emplace_back(first_entry + own.align_path_off[i]);
            }

            batch->cluster_read_off.emplace_back(all.read_count.size());
            cluster.alignments = SynthAlignments();
        }
    }

    if (config.keep_alignments) {

        batch->read_noise_score.assign(batch->alignments.read_count.size(), std::numeric_limits<int32_t>::lowest());
        batch->align_length.assign(batch->alignments.align_score_sum.size(), synth_align_length);
        batch->align_frag_length.assign(batch->alignments.align_score_sum.size(), synth_frag_length);
    }

    return static_cast<FlatBatchStorage *>(batch);
}

// Builds ONE cluster from raw reads: read r has count read_count[r], noise read_noise[r] and the per-path
// likelihoods (lik_path, lik_value)[lik_off[r] .. lik_off[r+1]); each read is finished like
// ReadPathProbabilities::addPathProbs finishes a row, then the rows are sorted and merged like the
// caller does (src/main.cpp:953-973).  Harness hook for testing the row type; same handle type as
// rpvg_amd_synth_generate().
void * rpvg_amd_rows_from_likelihoods(uint32_t num_paths, uint32_t num_reads, const uint32_t * read_count, const double * read_noise, const uint64_t * lik_off, const uint32_t * lik_path, const double * lik_value, double prob_precision) {

    std::vector<ReadPathProbabilities> rows;
    std::vector<std::pair<uint32_t, double> > likelihoods;

    for (uint32_t r = 0; r < num_reads; ++r) {

        likelihoods.clear();

        for (uint64_t i = lik_off[r]; i < lik_off[r + 1]; ++i) {

            likelihoods.emplace_back(lik_path[i], lik_value[i]);
        }

        if (likelihoods.empty()) {

            rows.emplace_back(read_count[r], 1.0, ReadPathProbabilities::PathProbs(), prob_precision);

        } else {

            rows.emplace_back(ReadPathProbabilities::fromPathLikelihoods(read_count[r], read_noise[r], likelihoods, prob_precision));
        }
    }

    sortAndMergeReadPathProbabilities(&rows);

    SynthBatch * batch = new SynthBatch();
    batch->addCluster(std::vector<PathInfo>(num_paths, PathInfo()), rows);

    return static_cast<FlatBatchStorage *>(batch);
}

void rpvg_amd_synth_view(void * handle, rpvg_cluster_batch * out) {

    static_cast<FlatBatchStorage *>(handle)->view(out);
}

void rpvg_amd_synth_sizes(void * handle, uint64_t * rows, uint64_t * groups, uint64_t * entries, uint64_t * paths, uint64_t * sources) {

    FlatBatchStorage * batch = static_cast<FlatBatchStorage *>(handle);

    *rows = batch->row_count.size();
    *groups = batch->grp_prob.size();
    *entries = batch->path_idx.size();
    *paths = batch->path_group_id.size();
    *sources = batch->source_id.size();
}

// The reads of a batch generated with keep_alignments as the input of row construction.  The path arrays alias
// the batch's own (cluster_path_off, effective lengths, source counts); no name-group collapsing.
int rpvg_amd_synth_alignments_view(void * handle, rpvg_alignment_batch * out) {

    SynthBatch * batch = dynamic_cast<SynthBatch *>(static_cast<FlatBatchStorage *>(handle));

    if (!batch || batch->cluster_read_off.size() != batch->cluster_path_off.size()) {

        return -1;
    }

    std::memset(out, 0, sizeof(*out));

    out->num_clusters = batch->cluster_path_off.size() - 1;
    out->cluster_read_off = batch->cluster_read_off.data();
    out->cluster_path_off = batch->cluster_path_off.data();
    out->path_effective_length = batch->path_effective_length.data();
    out->path_source_count = batch->path_source_count.data();
    out->read_count = batch->alignments.read_count.data();
    out->read_min_mapq = batch->alignments.read_min_mapq.data();
    out->read_noise_score = batch->read_noise_score.data();
    out->read_align_off = batch->alignments.read_align_off.data();
    out->align_score_sum = batch->alignments.align_score_sum.data();
    out->align_length = batch->align_length.data();
    out->align_frag_length = batch->align_frag_length.data();
    out->align_path_off = batch->alignments.align_path_off.data();
    out->align_path_idx = batch->alignments.align_path_idx.data();

    return 0;
}

void rpvg_amd_synth_free(void * handle) {

    delete static_cast<FlatBatchStorage *>(handle);
}

}
