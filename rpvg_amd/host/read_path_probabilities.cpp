#include "read_path_probabilities.hpp"

#include <algorithm>
#include <cassert>
#include <cmath>
#include <limits>
#include <ostream>

#include "numeric_utils.hpp"

namespace rpvg_amd {

ReadPathProbabilities::ReadPathProbabilities() : read_count(0), noise_prob(1), prob_precision(1e-8) {}

ReadPathProbabilities::ReadPathProbabilities(const uint32_t read_count_in, const double prob_precision_in) : read_count(read_count_in), noise_prob(1), prob_precision(prob_precision_in) {}

ReadPathProbabilities::ReadPathProbabilities(const uint32_t read_count_in, const double noise_prob_in, const PathProbs & path_probs_in, const double prob_precision_in) : read_count(read_count_in), noise_prob(noise_prob_in), path_probs(path_probs_in), prob_precision(prob_precision_in) {

    assert(noise_prob > 0 && noise_prob <= 1);
    std::sort(path_probs.begin(), path_probs.end());
}

ReadPathProbabilities ReadPathProbabilities::fromPathLikelihoods(const uint32_t read_count_in, const double noise_prob_in, const std::vector<std::pair<uint32_t, double> > & path_likelihoods, const double prob_precision_in) {

    ReadPathProbabilities row(read_count_in, prob_precision_in);
    row.noise_prob = noise_prob_in;

    double likelihood_sum = 0;

    for (auto & path_likelihood: path_likelihoods) {

        likelihood_sum += path_likelihood.second;
    }

    if (!(likelihood_sum > 0)) {

        row.noise_prob = 1;
        return row;
    }

    double low_prob_sum = 0;

    for (auto & path_likelihood: path_likelihoods) {

        const double prob = path_likelihood.second / likelihood_sum;

        if (prob < prob_precision_in) {

            low_prob_sum += prob;
            continue;
        }

        bool placed = false;

        for (auto & bucket: row.path_probs) {

            if (std::abs(bucket.first - prob) < prob_precision_in) {

                bucket.first = (bucket.first * bucket.second.size() + prob) / (bucket.second.size() + 1);
                bucket.second.emplace_back(path_likelihood.first);

                placed = true;
                break;
            }
        }

        if (!placed) {

            row.path_probs.emplace_back(prob, std::vector<uint32_t>(1, path_likelihood.first));
        }
    }

    for (auto & bucket: row.path_probs) {

        bucket.first *= (1 - row.noise_prob);
    }

    row.noise_prob += low_prob_sum * (1 - row.noise_prob);

    std::sort(row.path_probs.begin(), row.path_probs.end());

    return row;
}

uint32_t ReadPathProbabilities::readCount() const {

    return read_count;
}

double ReadPathProbabilities::noiseProb() const {

    return noise_prob;
}

const ReadPathProbabilities::PathProbs & ReadPathProbabilities::pathProbs() const {

    return path_probs;
}

void ReadPathProbabilities::addReadCount(const uint32_t read_count_in) {

    read_count += read_count_in;
}

bool ReadPathProbabilities::quickMergeIdentical(const ReadPathProbabilities & probs_2) {

    if (std::abs(noise_prob - probs_2.noise_prob) >= prob_precision || path_probs.size() != probs_2.path_probs.size()) {

        return false;
    }

    for (size_t i = 0; i < path_probs.size(); ++i) {

        if (std::abs(path_probs[i].first - probs_2.path_probs[i].first) >= prob_precision || path_probs[i].second != probs_2.path_probs[i].second) {

            return false;
        }
    }

    read_count += probs_2.read_count;
    return true;
}

bool operator==(const ReadPathProbabilities & lhs, const ReadPathProbabilities & rhs) {

    if (lhs.readCount() != rhs.readCount() || !numeric::doubleCompare(lhs.noiseProb(), rhs.noiseProb()) || lhs.pathProbs().size() != rhs.pathProbs().size()) {

        return false;
    }

    for (size_t i = 0; i < lhs.pathProbs().size(); ++i) {

        if (!numeric::doubleCompare(lhs.pathProbs()[i].first, rhs.pathProbs()[i].first) || lhs.pathProbs()[i].second != rhs.pathProbs()[i].second) {

            return false;
        }
    }

    return true;
}

bool operator!=(const ReadPathProbabilities & lhs, const ReadPathProbabilities & rhs) {

    return !(lhs == rhs);
}

// Ordering of src/read_path_probabilities.cpp:283-322: noise (tolerant), number
// of probability groups, then per group probability (tolerant), group size and
// path indices; read count last.
bool operator<(const ReadPathProbabilities & lhs, const ReadPathProbabilities & rhs) {

    if (!numeric::doubleCompare(lhs.noiseProb(), rhs.noiseProb())) {

        return lhs.noiseProb() < rhs.noiseProb();
    }

    const auto & lp = lhs.pathProbs();
    const auto & rp = rhs.pathProbs();

    if (lp.size() != rp.size()) {

        return lp.size() < rp.size();
    }

    for (size_t i = 0; i < lp.size(); ++i) {

        if (!numeric::doubleCompare(lp[i].first, rp[i].first)) {

            return lp[i].first < rp[i].first;
        }

        if (lp[i].second.size() != rp[i].second.size()) {

            return lp[i].second.size() < rp[i].second.size();
        }

        auto mismatch = std::mismatch(lp[i].second.begin(), lp[i].second.end(), rp[i].second.begin());

        if (mismatch.first != lp[i].second.end()) {

            return *mismatch.first < *mismatch.second;
        }
    }

    return lhs.readCount() < rhs.readCount();
}

std::ostream & operator<<(std::ostream & os, const ReadPathProbabilities & read_path_probs) {

    os << read_path_probs.readCount() << " | " << read_path_probs.noiseProb() << " |";

    for (auto & path_probs: read_path_probs.pathProbs()) {

        os << " " << path_probs.first << ":";

        for (size_t i = 0; i < path_probs.second.size(); ++i) {

            os << (i ? "," : " ") << path_probs.second[i];
        }
    }

    return os;
}

void sortAndMergeReadPathProbabilities(std::vector<ReadPathProbabilities> * cluster_probs) {

    std::sort(cluster_probs->begin(), cluster_probs->end());

    if (cluster_probs->empty()) {

        return;
    }

    size_t last_unique = 0;

    for (size_t i = 1; i < cluster_probs->size(); ++i) {

        if (!cluster_probs->at(last_unique).quickMergeIdentical(cluster_probs->at(i))) {

            ++last_unique;

            if (last_unique < i) {

                cluster_probs->at(last_unique) = std::move(cluster_probs->at(i));
            }
        }
    }

    cluster_probs->resize(last_unique + 1);
}

}
