// Boundary data types of the inference hot path, with the same fields and
// semantics as the reference's (so that a caller written against rpvg's
// structs compiles against these by changing the include):
//   PathInfo              src/path_cluster_estimates.hpp:15-33
//   CountSamples          src/path_cluster_estimates.hpp:35-43
//   PathClusterEstimates  src/path_cluster_estimates.hpp:45-111
// Differences: std containers instead of sparsepp (source_ids is a sorted
// std::set — only membership and iteration are used), no Eigen include.
#ifndef RPVG_AMD_PATH_CLUSTER_ESTIMATES_HPP
#define RPVG_AMD_PATH_CLUSTER_ESTIMATES_HPP

#include <cassert>
#include <cstdint>
#include <set>
#include <string>
#include <vector>

namespace rpvg_amd {

struct PathInfo {

    std::string name;
    uint32_t group_id = 0;

    uint32_t source_count = 1;
    std::set<uint32_t> source_ids;

    uint32_t length = 0;
    double effective_length = 0;

    PathInfo() {}
    explicit PathInfo(const std::string & name_in) : name(name_in) {}
};

struct CountSamples {

    std::vector<uint32_t> path_ids;

    std::vector<double> noise_samples;
    std::vector<double> abundance_samples;
};

struct PathClusterEstimates {

    std::vector<PathInfo> paths;

    std::vector<std::vector<uint32_t> > path_group_sets;

    std::vector<double> posteriors;
    std::vector<double> abundances;

    double noise_count = 0;
    double total_count = 0;

    std::vector<CountSamples> gibbs_read_count_samples;

    // Not in the reference: iteration count and column paths of every EM
    // solve behind this cluster (instrumentation for iteration parity).
    std::vector<uint32_t> em_iterations;
    std::vector<std::vector<uint32_t> > em_problem_paths;

    // All multisets of size group_size over [0, num_components), in
    // lexicographic order (iterative form of generateGroupsRecursive,
    // src/path_cluster_estimates.hpp:65-89).
    void generateGroups(const uint32_t num_components, const uint32_t group_size) {

        if (num_components == 0 || group_size == 0) {

            return;
        }

        std::vector<uint32_t> cur(group_size, 0);

        while (true) {

            path_group_sets.emplace_back(cur);

            int32_t pos = static_cast<int32_t>(group_size) - 1;

            while (pos >= 0 && cur[pos] + 1 == num_components) {

                --pos;
            }

            if (pos < 0) {

                break;
            }

            const uint32_t next = cur[pos] + 1;

            for (uint32_t i = pos; i < group_size; ++i) {

                cur[i] = next;
            }
        }
    }

    // src/path_cluster_estimates.hpp:91-110
    void resetEstimates(const uint32_t num_components, const uint32_t group_size) {

        path_group_sets.clear();

        posteriors.clear();
        abundances.clear();

        noise_count = 0;
        total_count = 0;

        gibbs_read_count_samples.clear();

        em_iterations.clear();
        em_problem_paths.clear();

        if (group_size > 0) {

            generateGroups(num_components, group_size);

            posteriors.assign(path_group_sets.size(), 0);
            abundances.assign(path_group_sets.size() * group_size, 0);
        }
    }
};

}

#endif
