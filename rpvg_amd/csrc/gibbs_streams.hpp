// The random streams of estimatePathGroupPosteriorsGibbs (src/path_estimator.cpp:475-589), restated so that a kernel
// can follow them draw for draw.  The reference draws from a std::mt19937 through libstdc++'s distributions:
//   * std::uniform_int_distribution<uint32_t>(0, n - 1) for the start of every chain (:491,509) — GCC 11's
//     bits/uniform_int_dist.h:246-270,311-317: Lemire's multiply-and-reject on one 32-bit word per attempt;
//   * std::discrete_distribution<uint32_t> for every slot of every iteration (:556) — bits/random.tcc:2656-2713:
//     probabilities divided by their sum, partial sums with the last one set to 1, one generate_canonical<double, 53>
//     (two 32-bit words, :3348-3380) and a lower_bound over the partial sums; a distribution of fewer than two
//     weights returns 0 WITHOUT touching the generator.
// mt19937 itself is a published algorithm (Matsumoto & Nishimura 1998; the parameters of [rand.predef]).
// Everything here compiles for the host too (tests/cpp/gibbs_streams_check.cpp compares it with libstdc++ on the CPU).
#ifndef RPVG_GIBBS_STREAMS_HPP
#define RPVG_GIBBS_STREAMS_HPP

#include <cstdint>

#if defined(__HIPCC__)
#define RPVG_STREAM_FN __host__ __device__ __forceinline__
#else
#define RPVG_STREAM_FN inline
#endif

namespace rpvg_streams {

constexpr uint32_t kMtWords = 624;   // state size n
constexpr uint32_t kMtShift = 397;   // m
constexpr uint32_t kMtTail = kMtWords - kMtShift;  // 227: words whose partner is still the old state

// output function of mt19937 (u = 11, s = 7, b = 0x9d2c5680, t = 15, c = 0xefc60000, l = 18; d is all ones)
RPVG_STREAM_FN uint32_t mtTemper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

// its inverse: the state word behind an output.  x -> x ^ ((x << s) & mask) fixes the low s bits, so iterating
// y ^ ((x << s) & mask) from x = y recovers s more bits per round (likewise from the top for right shifts).
RPVG_STREAM_FN uint32_t mtUntemper(const uint32_t out) {
    uint32_t y = out, x;
    x = y;
    x = y ^ (x >> 18);
    x = y ^ (x >> 18);
    y = x;
    x = y;
    for (int r = 0; r < 3; ++r) x = y ^ ((x << 15) & 0xefc60000u);
    y = x;
    x = y;
    for (int r = 0; r < 5; ++r) x = y ^ ((x << 7) & 0x9d2c5680u);
    y = x;
    x = y;
    for (int r = 0; r < 3; ++r) x = y ^ (x >> 11);
    return x;
}

// x[k + n] from x[k], x[k + 1], x[k + m] (r = 31, a = 0x9908b0df)
RPVG_STREAM_FN uint32_t mtNext(const uint32_t x_k, const uint32_t x_k1, const uint32_t x_km) {
    const uint32_t y = (x_k & 0x80000000u) | (x_k1 & 0x7fffffffu);
    return x_km ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

// generate_canonical<double, 53> over a 32-bit generator: two words, the first one the low part
RPVG_STREAM_FN double canonicalFromWords(const uint32_t first, const uint32_t second) {
    double sum = static_cast<double>(first);
    sum += static_cast<double>(second) * 4294967296.0;
    const double ret = sum * 5.421010862427522e-20;  // / 2^64, exact
    return ret >= 1.0 ? 0.99999999999999988898 : ret;  // nextafter(1, 0)
}

// uniform_int_distribution<uint32_t>(0, range - 1), range >= 1: `next` hands out the generator's words
template <typename NextWord>
RPVG_STREAM_FN uint32_t uniformBelow(const uint32_t range, NextWord next) {
    uint64_t product = static_cast<uint64_t>(next()) * static_cast<uint64_t>(range);
    uint32_t low = static_cast<uint32_t>(product);
    if (low < range) {
        const uint32_t threshold = (0u - range) % range;
        while (low < threshold) {
            product = static_cast<uint64_t>(next()) * static_cast<uint64_t>(range);
            low = static_cast<uint32_t>(product);
        }
    }
    return static_cast<uint32_t>(product >> 32);
}

// std::lower_bound over the partial sums of a discrete_distribution: first index whose partial sum is >= p
template <typename PartialSums>
RPVG_STREAM_FN uint32_t firstNotBelow(PartialSums cp, const uint32_t n, const double p) {
    uint32_t lo = 0, len = n;
    while (len > 0) {
        const uint32_t half = len >> 1;
        if (cp(lo + half) < p) {
            lo += half + 1;
            len -= half + 1;
        } else {
            len = half;
        }
    }
    return lo;
}

}  // namespace rpvg_streams

#endif
