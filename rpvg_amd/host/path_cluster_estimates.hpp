// Boundary data types of the inference hot path, with the same fields and
// semantics as the reference's (so that a caller written against rpvg's
// structs compiles against these by changing the include):
//   PathInfo              src/path_cluster_estimates.hpp:15-33
//   CountSamples          src/path_cluster_estimates.hpp:35-43
//   PathClusterEstimates  src/path_cluster_estimates.hpp:45-111
// Differences: std containers instead of sparsepp, no Eigen include; source_ids is a SourceIdSet (below): the set
// interface the reference's code uses (insert / emplace / count / size / iteration) over ONE sorted array, because
// the estimators walk the ids of every path of every cluster of a batch and a node-based set made that walk most of
// a batch's host time (30 ms per 200 k paths; the GPU step is 13 ms).
#ifndef RPVG_AMD_PATH_CLUSTER_ESTIMATES_HPP
#define RPVG_AMD_PATH_CLUSTER_ESTIMATES_HPP

#include <algorithm>
#include <cassert>
#include <cstdint>
#include <initializer_list>
#include <string>
#include <vector>

namespace rpvg_amd {

// Set of haplotype (source) ids: unique, ascending, contiguous.
class SourceIdSet {

    public:

        typedef std::vector<uint32_t>::const_iterator const_iterator;
        typedef std::vector<uint32_t>::const_reverse_iterator const_reverse_iterator;
        typedef uint32_t value_type;

        SourceIdSet() {}
        SourceIdSet(std::initializer_list<uint32_t> ids_in) { insert(ids_in.begin(), ids_in.end()); }

        std::pair<const_iterator, bool> insert(const uint32_t id) {

            if (ids.empty() || ids.back() < id) {  // ids mostly arrive ascending

                ids.push_back(id);
                return std::make_pair(ids.end() - 1, true);
            }

            auto it = std::lower_bound(ids.begin(), ids.end(), id);

            if (it != ids.end() && *it == id) {

                return std::make_pair(const_iterator(it), false);
            }

            return std::make_pair(const_iterator(ids.insert(it, id)), true);
        }

        std::pair<const_iterator, bool> emplace(const uint32_t id) { return insert(id); }

        template <typename Iterator>
        void insert(Iterator first, Iterator last) {

            const size_t old_size = ids.size();
            ids.insert(ids.end(), first, last);

            bool ascending = true;

            for (size_t i = std::max<size_t>(old_size, 1); ascending && i < ids.size(); ++i) {

                ascending = ids[i - 1] < ids[i];
            }

            if (!ascending) {

                std::sort(ids.begin(), ids.end());
                ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
            }
        }

        size_t count(const uint32_t id) const { return std::binary_search(ids.begin(), ids.end(), id) ? 1 : 0; }

        const_iterator find(const uint32_t id) const {

            auto it = std::lower_bound(ids.begin(), ids.end(), id);
            return (it != ids.end() && *it == id) ? it : ids.end();
        }

        size_t size() const { return ids.size(); }
        bool empty() const { return ids.empty(); }
        void clear() { ids.clear(); }
        void reserve(const size_t capacity) { ids.reserve(capacity); }

        const_iterator begin() const { return ids.begin(); }
        const_iterator end() const { return ids.end(); }
        const_reverse_iterator rbegin() const { return ids.rbegin(); }
        const_reverse_iterator rend() const { return ids.rend(); }

        const uint32_t * data() const { return ids.data(); }

        bool operator==(const SourceIdSet & other) const { return ids == other.ids; }
        bool operator!=(const SourceIdSet & other) const { return ids != other.ids; }

    private:

        std::vector<uint32_t> ids;
};

struct PathInfo {

    std::string name;
    uint32_t group_id = 0;

    uint32_t source_count = 1;
    SourceIdSet source_ids;

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
