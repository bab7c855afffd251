// SourceIdSet (rpvg_amd/host/path_cluster_estimates.hpp): the set interface the reference's code uses on
// PathInfo::source_ids (src/path_cluster_estimates.hpp:21, src/main.cpp:855-887), over one sorted array.
#include "path_cluster_estimates.hpp"

#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>

#define CHECK(condition) do { if (!(condition)) { std::fprintf(stderr, "line %d: %s\n", __LINE__, #condition); return 1; } } while (0)

int main() {
    using rpvg_amd::SourceIdSet;
    SourceIdSet ids;
    CHECK(ids.empty() && ids.size() == 0 && ids.begin() == ids.end());
    CHECK(ids.insert(7).second && ids.insert(3).second && ids.emplace(11).second);
    CHECK(!ids.insert(7).second && *ids.insert(7).first == 7);  // already there
    CHECK(ids.size() == 3 && ids.count(3) == 1 && ids.count(4) == 0 && ids.find(11) != ids.end() && ids.find(12) == ids.end());
    CHECK(std::vector<uint32_t>(ids.begin(), ids.end()) == std::vector<uint32_t>({3, 7, 11}));
    CHECK(*ids.begin() == 3 && *ids.rbegin() == 11);
    const std::vector<uint32_t> more = {11, 2, 9, 2, 40};  // unordered, with duplicates and a member
    ids.insert(more.begin(), more.end());
    CHECK(std::vector<uint32_t>(ids.begin(), ids.end()) == std::vector<uint32_t>({2, 3, 7, 9, 11, 40}));
    const std::vector<uint32_t> ascending = {41, 50, 60};   // the common case: appended
    ids.insert(ascending.begin(), ascending.end());
    CHECK(ids.size() == 9 && *ids.rbegin() == 60);
    // against std::set on a pseudo-random sequence
    std::set<uint32_t> reference;
    SourceIdSet flat = {5, 1, 5};
    reference.insert({5, 1});
    uint32_t x = 12345;
    for (int i = 0; i < 2000; ++i) {
        x = x * 1664525u + 1013904223u;
        const uint32_t id = (x >> 16) % 300;
        CHECK(flat.insert(id).second == reference.insert(id).second);
    }
    CHECK(std::vector<uint32_t>(flat.begin(), flat.end()) == std::vector<uint32_t>(reference.begin(), reference.end()));
    SourceIdSet copy = flat;
    CHECK(copy == flat);
    copy.clear();
    CHECK(copy.empty() && copy != flat);
    std::printf("ok\n");
    return 0;
}
