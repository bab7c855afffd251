// Scalar helpers the estimators share; semantics of the corresponding
// Utils:: functions of the reference (src/utils.hpp:81-117, 300-302).
#ifndef RPVG_AMD_NUMERIC_UTILS_HPP
#define RPVG_AMD_NUMERIC_UTILS_HPP

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

namespace rpvg_amd {
namespace numeric {

// Relative tolerance used when comparing doubles (src/utils.hpp:81).
static const double double_precision = std::numeric_limits<double>::epsilon() * 100;

// Start value of every log-sum-exp fold in the reference (not -inf;
// src/path_estimator.cpp:348,418,453,540).
static const double log_zero = std::numeric_limits<double>::lowest();

inline bool doubleCompare(const double a, const double b) {

    assert(std::isfinite(a));
    assert(std::isfinite(b));

    return (a == b) || (std::abs(a - b) < std::abs(std::min(a, b)) * double_precision);
}

// n! / (n - u + 1)! with u the number of distinct values (src/utils.hpp:95-117).
// Exact multiset permutation count up to ploidy 3; kept as is beyond that.
inline uint32_t numPermutations(std::vector<uint32_t> values) {

    assert(!values.empty());

    std::sort(values.begin(), values.end());
    const uint32_t num_unique = std::unique(values.begin(), values.end()) - values.begin();
    const uint32_t n = values.size();

    if (n == 1) {

        return 1;
    }

    return static_cast<uint32_t>(std::tgamma(n + 1) / std::tgamma(n - num_unique + 2));
}

// log(exp(log_x) + exp(log_y)) (src/utils.hpp:300-302).
inline double add_log(const double log_x, const double log_y) {

    const double hi = std::max(log_x, log_y);
    const double lo = std::min(log_x, log_y);

    // keep the reference's operand choice when the two are equal
    return (log_x > log_y) ? log_x + std::log1p(std::exp(log_y - log_x)) : hi + std::log1p(std::exp(lo - hi));
}

}
}

#endif
