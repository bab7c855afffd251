#include "estimates_writers.hpp"

#include <cassert>
#include <cmath>
#include <iomanip>
#include <limits>

#include "cluster_io.hpp"

namespace rpvg_amd {

// src/threaded_output_writer.cpp:6
static const uint32_t out_precision_digits = 8;

double totalTranscriptCount(const ClusterEstimatesList & path_cluster_estimates) {

    double total_transcript_count = 0;

    for (auto & cur_estimates: path_cluster_estimates) {

        auto abundances_it = cur_estimates.second.abundances.begin();

        for (auto & path_group_set: cur_estimates.second.path_group_sets) {

            for (auto & path: path_group_set) {

                assert(abundances_it != cur_estimates.second.abundances.end());

                const double path_effective_length = cur_estimates.second.paths.at(path).effective_length;

                if (path_effective_length > 0) {

                    total_transcript_count += (*abundances_it / path_effective_length);
                }

                ++abundances_it;
            }
        }
    }

    return total_transcript_count;
}

EstimatesWriter::EstimatesWriter(const std::string & filename_in) : filename(filename_in) {}

void EstimatesWriter::close() {

    writeTextFile(filename, out.str());
}

AbundanceEstimatesWriter::AbundanceEstimatesWriter(const std::string filename_prefix, const double total_transcript_count_in) : EstimatesWriter(filename_prefix + ".txt"), total_transcript_count(total_transcript_count_in), noise_count(0) {

    out << "Name\tClusterID\tLength\tEffectiveLength\tReadCount\tTPM" << std::endl;
}

void AbundanceEstimatesWriter::addEstimates(const ClusterEstimatesList & path_cluster_estimates) {

    out << std::setprecision(out_precision_digits);

    for (auto & cur_estimates: path_cluster_estimates) {

        const auto & estimates = cur_estimates.second;

        assert(estimates.paths.size() == estimates.path_group_sets.size());
        assert(estimates.paths.size() == estimates.abundances.size());

        for (size_t i = 0; i < estimates.path_group_sets.size(); ++i) {

            assert(estimates.path_group_sets.at(i).size() == 1);
            const auto & path = estimates.paths.at(estimates.path_group_sets.at(i).front());

            const double transcript_count = (path.effective_length > 0) ? estimates.abundances.at(i) / path.effective_length : 0;

            out << path.name << "\t" << cur_estimates.first << "\t" << path.length << "\t" << path.effective_length;
            out << "\t" << estimates.abundances.at(i) << "\t" << transcript_count / total_transcript_count * std::pow(10, 6) << std::endl;
        }

        noise_count += estimates.noise_count;
    }
}

void AbundanceEstimatesWriter::addNoiseTranscript(const uint32_t unaligned_read_count) {

    out << std::setprecision(out_precision_digits);
    out << "Unknown\t0\t0\t0\t" << noise_count + unaligned_read_count << "\t0" << std::endl;
}

HaplotypeAbundanceEstimatesWriter::HaplotypeAbundanceEstimatesWriter(const std::string filename_prefix, const uint32_t ploidy_in, const double total_transcript_count_in) : EstimatesWriter(filename_prefix + ".txt"), ploidy(ploidy_in), total_transcript_count(total_transcript_count_in), noise_count(0) {

    out << "Name\tClusterID\tLength\tEffectiveLength\tHaplotypeProbability\tReadCount\tTPM" << std::endl;
}

void HaplotypeAbundanceEstimatesWriter::addEstimates(const ClusterEstimatesList & path_cluster_estimates) {

    out << std::setprecision(out_precision_digits);

    for (auto & cur_estimates: path_cluster_estimates) {

        const auto & estimates = cur_estimates.second;
        assert(estimates.path_group_sets.size() == estimates.posteriors.size());

        // per path: probability of carrying it (a homozygous set counts once) and its read count
        std::vector<double> haplotype_probs(estimates.paths.size(), 0);
        std::vector<double> read_counts(estimates.paths.size(), 0);

        auto abundances_it = estimates.abundances.begin();

        for (size_t i = 0; i < estimates.path_group_sets.size(); ++i) {

            const auto & path_group_set = estimates.path_group_sets.at(i);
            assert(!path_group_set.empty() && path_group_set.size() <= ploidy);

            for (size_t j = 0; j < path_group_set.size(); ++j) {

                if (j == 0 || path_group_set.at(j) != path_group_set.at(j - 1)) {

                    haplotype_probs.at(path_group_set.at(j)) += estimates.posteriors.at(i);
                }

                read_counts.at(path_group_set.at(j)) += *abundances_it;
                ++abundances_it;
            }
        }

        assert(abundances_it == estimates.abundances.end());

        for (size_t i = 0; i < estimates.paths.size(); ++i) {

            const auto & path = estimates.paths.at(i);
            const double transcript_count = (path.effective_length > 0) ? read_counts.at(i) / path.effective_length : 0;

            out << path.name << "\t" << cur_estimates.first << "\t" << path.length << "\t" << path.effective_length;
            out << "\t" << haplotype_probs.at(i) << "\t" << read_counts.at(i) << "\t" << transcript_count / total_transcript_count * std::pow(10, 6) << std::endl;
        }

        noise_count += estimates.noise_count;
    }
}

void HaplotypeAbundanceEstimatesWriter::addNoiseTranscript(const uint32_t unaligned_read_count) {

    out << std::setprecision(out_precision_digits);
    out << "Unknown\t0\t0\t0\t0\t" << noise_count + unaligned_read_count << "\t0" << std::endl;
}

JointHaplotypeAbundanceEstimatesWriter::JointHaplotypeAbundanceEstimatesWriter(const std::string filename_prefix, const uint32_t ploidy_in, const double min_posterior_in, const double total_transcript_count_in) : EstimatesWriter(filename_prefix + ".txt"), ploidy(ploidy_in), min_posterior(min_posterior_in), total_transcript_count(total_transcript_count_in), noise_counts(ploidy_in, 0) {

    for (uint32_t i = 0; i < ploidy; ++i) {

        out << "Name_" << i + 1 << "\t";
    }

    out << "ClusterID\tHaplotypingProbability";

    for (uint32_t i = 0; i < ploidy; ++i) {

        out << "\tReadCount_" << i + 1 << "\tTPM_" << i + 1;
    }

    out << std::endl;
}

void JointHaplotypeAbundanceEstimatesWriter::addEstimates(const ClusterEstimatesList & path_cluster_estimates) {

    out << std::setprecision(out_precision_digits);

    for (auto & cur_estimates: path_cluster_estimates) {

        const auto & estimates = cur_estimates.second;
        assert(estimates.posteriors.size() == estimates.path_group_sets.size());

        auto abundances_it = estimates.abundances.begin();

        for (size_t i = 0; i < estimates.path_group_sets.size(); ++i) {

            const auto & path_group_set = estimates.path_group_sets.at(i);
            assert(!path_group_set.empty() && path_group_set.size() <= ploidy);

            if (estimates.posteriors.at(i) < min_posterior) {

                // not reported; its abundances are skipped (the reference would trip its end-of-abundances
                // assert here, src/threaded_output_writer.cpp:511 — sets this improbable do not occur there)
                abundances_it += path_group_set.size();
                continue;
            }

            for (auto & path: path_group_set) {

                out << estimates.paths.at(path).name << "\t";
            }

            for (size_t j = path_group_set.size(); j < ploidy; ++j) {

                out << ".\t";
            }

            out << cur_estimates.first << "\t" << estimates.posteriors.at(i);

            for (auto & path: path_group_set) {

                const double path_effective_length = estimates.paths.at(path).effective_length;
                const double transcript_count = (path_effective_length > 0) ? *abundances_it / path_effective_length : 0;

                out << "\t" << *abundances_it << "\t" << transcript_count / total_transcript_count * std::pow(10, 6);
                ++abundances_it;
            }

            for (size_t j = path_group_set.size(); j < ploidy; ++j) {

                out << "\t0\t0";
            }

            out << std::endl;
        }

        assert(abundances_it == estimates.abundances.end());

        // the cluster's noise is spread evenly over the ploidy columns of the Unknown row
        for (auto & noise_count: noise_counts) {

            noise_count += estimates.noise_count / noise_counts.size();
        }
    }
}

void JointHaplotypeAbundanceEstimatesWriter::addNoiseTranscript(const uint32_t unaligned_read_count) {

    out << std::setprecision(out_precision_digits);

    for (uint32_t i = 0; i < ploidy; ++i) {

        out << "Unknown\t";
    }

    out << "0\t0";

    for (auto & noise_count: noise_counts) {

        out << "\t" << noise_count + static_cast<float>(unaligned_read_count) / noise_counts.size() << "\t0";
    }

    out << std::endl;
}

JointHaplotypeEstimatesWriter::JointHaplotypeEstimatesWriter(const std::string filename_prefix, const uint32_t ploidy_in, const double min_posterior_in) : EstimatesWriter(filename_prefix + ".txt"), ploidy(ploidy_in), min_posterior(min_posterior_in) {

    for (uint32_t i = 0; i < ploidy; ++i) {

        out << "Name_" << i + 1 << "\t";
    }

    out << "ClusterID\tHaplotypingProbability" << std::endl;
}

void JointHaplotypeEstimatesWriter::addEstimates(const ClusterEstimatesList & path_cluster_estimates) {

    out << std::setprecision(out_precision_digits);

    for (auto & cur_estimates: path_cluster_estimates) {

        const auto & estimates = cur_estimates.second;
        assert(estimates.posteriors.size() == estimates.path_group_sets.size());

        for (size_t i = 0; i < estimates.path_group_sets.size(); ++i) {

            const auto & path_group_set = estimates.path_group_sets.at(i);
            assert(!path_group_set.empty() && path_group_set.size() <= ploidy);

            if (estimates.posteriors.at(i) < min_posterior) {

                continue;
            }

            for (auto & path: path_group_set) {

                out << estimates.paths.at(path).name << "\t";
            }

            for (size_t j = path_group_set.size(); j < ploidy; ++j) {

                out << ".\t";
            }

            out << cur_estimates.first << "\t" << estimates.posteriors.at(i) << std::endl;
        }
    }
}

ReadCountGibbsSamplesWriter::ReadCountGibbsSamplesWriter(const std::string filename_prefix, const uint32_t num_gibbs_samples_in) : EstimatesWriter(filename_prefix + ".txt.gz"), num_gibbs_samples(num_gibbs_samples_in), noise_counts(num_gibbs_samples_in, 0) {

    out << "Name\tClusterID";

    for (uint32_t i = 0; i < num_gibbs_samples; ++i) {

        out << "\tReadCountSample_" << i + 1;
    }

    out << std::endl;
}

void ReadCountGibbsSamplesWriter::addSamples(const std::pair<uint32_t, PathClusterEstimates> & path_cluster_estimate) {

    const auto & estimates = path_cluster_estimate.second;

    if (estimates.gibbs_read_count_samples.empty()) {

        // no samples for the cluster: all of its reads count as noise in every sample
        for (auto & noise_count: noise_counts) {

            noise_count += estimates.total_count;
        }

        return;
    }

    const uint32_t no_column = std::numeric_limits<uint32_t>::max();

    // column of every path inside every CountSamples of the cluster
    std::vector<std::vector<uint32_t> > path_gibbs_sampling_index(estimates.paths.size());
    uint32_t noise_count_idx = 0;

    for (size_t i = 0; i < estimates.gibbs_read_count_samples.size(); ++i) {

        const CountSamples & count_samples = estimates.gibbs_read_count_samples.at(i);

        assert(!count_samples.path_ids.empty());
        assert(count_samples.abundance_samples.size() == count_samples.path_ids.size() * count_samples.noise_samples.size());

        for (auto & noise_sample: count_samples.noise_samples) {

            noise_counts.at(noise_count_idx) += noise_sample;
            ++noise_count_idx;
        }

        for (size_t j = 0; j < count_samples.path_ids.size(); ++j) {

            auto & sampling_indices = path_gibbs_sampling_index.at(count_samples.path_ids.at(j));

            if (sampling_indices.empty()) {

                sampling_indices.assign(estimates.gibbs_read_count_samples.size(), no_column);
            }

            sampling_indices.at(i) = j;
        }
    }

    while (noise_count_idx < num_gibbs_samples) {

        noise_counts.at(noise_count_idx) += estimates.total_count;
        ++noise_count_idx;
    }

    out << std::setprecision(out_precision_digits);

    for (size_t i = 0; i < path_gibbs_sampling_index.size(); ++i) {

        const auto & sampling_indices = path_gibbs_sampling_index.at(i);

        if (sampling_indices.empty()) {

            continue;
        }

        out << estimates.paths.at(i).name << "\t" << path_cluster_estimate.first;

        uint32_t num_samples = 0;

        for (size_t j = 0; j < sampling_indices.size(); ++j) {

            const CountSamples & count_samples = estimates.gibbs_read_count_samples.at(j);
            const size_t num_paths = count_samples.path_ids.size();

            for (size_t k = 0; k < count_samples.noise_samples.size(); ++k) {

                if (sampling_indices.at(j) == no_column) {

                    out << "\t0";

                } else {

                    out << "\t" << count_samples.abundance_samples.at(k * num_paths + sampling_indices.at(j));
                }

                ++num_samples;
            }
        }

        while (num_samples < num_gibbs_samples) {

            out << "\t0";
            ++num_samples;
        }

        out << std::endl;
    }
}

void ReadCountGibbsSamplesWriter::addNoiseTranscript(const uint32_t unaligned_read_count) {

    out << std::setprecision(out_precision_digits);
    out << "Unknown\t0";

    for (auto & noise_count: noise_counts) {

        out << "\t" << noise_count + unaligned_read_count;
    }

    out << std::endl;
}

}
