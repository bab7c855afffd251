// estimatePathGroupPosteriorsGibbs (src/path_estimator.cpp:475-589) on the device, draw for draw.
//
// The reference runs, per cluster, num_gibbs_chains chains one after the other on the cluster's std::mt19937; every chain
// starts from uniform_int_distribution draws and then draws each slot of the group from the conditional distribution
// given the other slots (a discrete_distribution, memoised per "other" members).  Two facts make this parallel without
// changing a single draw:
//   * the words a chain consumes are known before it runs — group_size start draws (Lemire's rejection may take an extra
//     word: walked once, sequentially, by the kernel that generates the stream) and exactly two words per conditional
//     draw (none at all for a matrix of one column: libstdc++'s discrete_distribution of fewer than two weights returns
//     0 without touching the generator) — so every chain owns a slice of the generator's output, generated up front;
//   * a conditional depends on the other members only, so the chains of a problem share one memo table and the set of
//     conditionals evaluated is the set the sequential sampler evaluates.
// One thread per chain advances until its chain is done or needs a conditional nobody has evaluated yet; the requests of
// a round are evaluated together — groupConditionalKernel's arithmetic over all candidate columns, then the
// distribution's partial sums — and the chains go on.  Nothing crosses to the host inside the loop: the host queues rounds
// and looks at a counter of unfinished chains every few rounds.  Group sizes 1 and 2 (a conditional is indexed by one
// other member); larger groups stay with the host-driven sampler (rpvg_hip_group_conditionals).
#include <algorithm>
#include <cmath>
#include <numeric>

#include "common.hpp"
#include "gibbs_streams.hpp"

using namespace rpvg_hip_detail;

struct rpvg_hip_gibbs_sets {
    uint32_t num_problems = 0, group_size = 0;
    std::vector<uint64_t> set_off;
    std::vector<uint64_t> words_consumed;
    void * state_block = nullptr;  // pinned: [NG x 624] state words
    void * block = nullptr;  // pinned: first | second | count | sequence
    const uint32_t * first = nullptr, * second = nullptr, * count = nullptr;
    uint32_t rounds = 0;
    uint64_t conditionals = 0;
    ~rpvg_hip_gibbs_sets() {
        if (block) pinnedFree(block);
        if (state_block) pinnedFree(state_block);
    }
};

namespace {

constexpr uint32_t kMaxRounds = 8192;
constexpr uint32_t kPending = 0xffffffffu;
constexpr uint32_t kChainDone = 0x80000000u;
constexpr unsigned long long kEmptyKey = ~0ull;
constexpr uint32_t kErrStream = 1, kErrDistributions = 2;
constexpr uint32_t kCondCands = 64, kCondRows = 64, kCondStride = 70;  // gibbsConditionalTileKernel: doubles per staged row = 4 + 64 + noise + count
constexpr uint32_t kWordWindow = 16;  // generator words a chain keeps in LDS
constexpr uint32_t kRankInLds = 2048;  // sets of a problem ordered by first appearance inside the collect kernel up to this many

struct GibbsHeader {
    unsigned long long num_items;  // turns of the current round's conditionals (itemsPerTurn work items of up to 4 requests x 4 candidate columns)
    unsigned long long used;       // doubles of distribution storage handed out
    unsigned long long total_sets;
    double evals;                  // rows x columns over all requests
    uint32_t num_active;           // problems with a new request, counted by the advancing chains
    uint32_t cur_active;           // ... of the round being evaluated
    uint32_t cur_requests;         // its requests
    uint32_t total_requests;
    uint32_t error;
    uint32_t unsorted;             // problems whose sets left the collect kernel in table order
    uint32_t tiled;                // the current round's conditionals go through gibbsConditionalTileKernel
    uint32_t pad;
};

struct GibbsProblems {  // device arrays over the problems
    const uint32_t * matrix, * chains, * burn, * its;
    const uint64_t * chain_off;  // [P+1]
    const uint64_t * col_off;    // [P+1] columns before the problem: log frequencies, memo
    const uint64_t * tab_off;    // [P+1] slots of the problems' sample tables (powers of two)
    const double * log_freq;
};

struct GibbsChains {  // device arrays over the chains
    uint32_t * problem;
    unsigned long long * pos;  // next word of the chain in `stream`
    uint32_t * cur;            // [2 x chains]
    uint32_t * iter;
    uint32_t * flag;           // slot | kChainDone
};

__device__ __forceinline__ uint64_t mixKey(uint64_t x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33;
    return x;
}

// ---- the generators' words, and where every chain starts in them -------------------------------------
// One workgroup per generator.  The host hands over the generator's next 624 outputs; their untempered values are a
// complete state, from which the recurrence x[k+624] = f(x[k], x[k+1], x[k+397]) goes on a block of 624 at a time: the
// first 227 words of a block depend on the old block only, the next 227 on those, the last 170 on the second part
// (three steps, reads before writes).  Then one thread walks the generator's problems and their chains in the order the
// reference runs them and gives every chain its start (uniform_int_distribution) and its slice.
__global__ __launch_bounds__(256) void gibbsStreamKernel(const uint32_t * __restrict__ gen_words, const uint32_t * __restrict__ gen_prob_off,
                                                         const uint32_t * __restrict__ gen_prob, const uint64_t * __restrict__ stream_off,
                                                         const GibbsProblems pr, const uint32_t * __restrict__ mat_cols, const uint32_t group_size,
                                                         uint32_t * __restrict__ stream, const GibbsChains ch,
                                                         unsigned long long * __restrict__ words_consumed, uint32_t * __restrict__ final_state,
                                                         GibbsHeader * hdr) {
    using namespace rpvg_streams;
    __shared__ uint32_t x[kMtWords];
    __shared__ unsigned long long words_taken;
    const uint32_t g = blockIdx.x;
    const uint32_t tid = threadIdx.x;
    const uint64_t off = stream_off[g];
    const uint64_t capacity = stream_off[g + 1] - off;
    for (uint32_t i = tid; i < kMtWords; i += 256) {
        const uint32_t w = gen_words[static_cast<uint64_t>(g) * kMtWords + i];
        stream[off + i] = w;
        x[i] = mtUntemper(w);
    }
    __syncthreads();
    const uint32_t num_blocks = static_cast<uint32_t>(capacity / kMtWords);
    for (uint32_t b = 1; b < num_blocks; ++b) {
        uint32_t a = 0, a1 = 0, am = 0;
        if (tid < kMtTail) {
            a = x[tid];
            a1 = x[tid + 1];
            am = x[tid + kMtShift];
        }
        __syncthreads();
        if (tid < kMtTail) x[tid] = mtNext(a, a1, am);
        __syncthreads();
        if (tid < kMtTail) {
            a = x[kMtTail + tid];
            a1 = x[kMtTail + tid + 1];
            am = x[tid];
        }
        __syncthreads();
        if (tid < kMtTail) x[kMtTail + tid] = mtNext(a, a1, am);
        __syncthreads();
        const uint32_t i = 2 * kMtTail + tid;
        if (i < kMtWords) {
            a = x[i];
            a1 = x[i + 1 == kMtWords ? 0 : i + 1];
            am = x[i - kMtTail];
        }
        __syncthreads();
        if (i < kMtWords) x[i] = mtNext(a, a1, am);
        __syncthreads();
        for (uint32_t k = tid; k < kMtWords; k += 256) stream[off + static_cast<uint64_t>(b) * kMtWords + k] = mtTemper(x[k]);
    }
    __syncthreads();  // the block's words, as thread 0 reads them below
    if (tid == 0) {
        uint64_t pos = 0;
        bool overrun = false;
        for (uint32_t j = gen_prob_off[g]; j < gen_prob_off[g + 1] && !overrun; ++j) {
            const uint32_t p = gen_prob[j];
            const uint32_t G = mat_cols[pr.matrix[p]];
            const uint64_t words_per_chain = static_cast<uint64_t>(pr.burn[p] + pr.its[p]) * group_size * (G >= 2 ? 2 : 0);
            const uint64_t c0 = pr.chain_off[p];
            const uint32_t chains = pr.chains[p];
            for (uint32_t c = 0; c < chains; ++c) {
                uint32_t start[2] = {0, 0};
                for (uint32_t s = 0; s < group_size; ++s) {
                    start[s] = uniformBelow(G, [&]() {
                        if (pos >= capacity) {
                            overrun = true;
                            return 0xffffffffu;  // never rejected
                        }
                        return stream[off + pos++];
                    });
                }
                ch.problem[c0 + c] = p;
                ch.pos[c0 + c] = off + pos;
                ch.cur[2 * (c0 + c)] = start[0];
                ch.cur[2 * (c0 + c) + 1] = start[1];
                ch.iter[c0 + c] = 0;
                ch.flag[c0 + c] = 0;
                pos += words_per_chain;
                if (pos > capacity) overrun = true;
            }
        }
        words_consumed[g] = pos;
        words_taken = overrun ? 0 : pos;
        if (overrun) atomicOr(&hdr->error, kErrStream);
    }
    __syncthreads();
    // where the generator stands afterwards: the 624 state words BEFORE its next output — a std::mt19937 seeded with them
    // (through a seed sequence that hands them out: [rand.eng.mers] copies them into the state and regenerates on the
    // next call) goes on with output number words_taken.  A generator that gave fewer words than that is moved by discard().
    const unsigned long long taken = words_taken;
    if (taken >= kMtWords) {
        for (uint32_t i = tid; i < kMtWords; i += 256) final_state[static_cast<uint64_t>(g) * kMtWords + i] = mtUntemper(stream[off + taken - kMtWords + i]);
    }
}

// ---- the chains --------------------------------------------------------------------------------------
// src/path_estimator.cpp:505-575, one thread per chain, until the chain is done or needs a conditional that is not there.
// Requests are numbered per problem — request k of problem p is slot col_off[p] + k — so that the new requests of a
// problem sit next to each other and share the reading of the matrix (gibbsConditionalKernel).
// What a chain does most of the time is draw the mode of one of two distributions again: the distribution of each slot
// (its storage, and the partial sums either side of its largest weight) stays in registers while the other member does not
// change, the generator's words are fetched ahead of the draw, and a run of equal samples is counted once.
// Storage of one distribution: the G partial sums, then G bucket records — bucket b covers the draws u in [b / G, (b + 1) / G)
// and holds the first column whose partial sum reaches b / G with the partial sums either side of it: a draw that falls
// between them is settled by that one record, any other walks on from there (a step or two) instead of the log2(G) dependent
// loads of a bisection.
struct DrawBucket {
    double below, upto;  // partial sums before and including the column
    unsigned long long column;
};
__host__ __device__ inline unsigned long long distributionDoubles(const unsigned long long columns) { return 4 * columns; }

struct RequestInfo {  // 32 bytes per (problem, other member): the memo of the conditionals and what a draw needs first
    double mode_below, mode_upto;  // partial sums before and including the largest weight: the mode is drawn iff below < u <= upto
    unsigned long long dist_off;
    uint32_t mode;
    uint32_t state;  // 0: nobody asked; kPending: being numbered; k + 1: request k of the problem
};

template <int GS>
__global__ __launch_bounds__(256) void gibbsAdvanceKernel(const uint32_t num_chains, const uint32_t round, const GibbsProblems pr,
                                                          const uint32_t * __restrict__ mat_cols, const GibbsChains ch,
                                                          const uint32_t * __restrict__ stream, RequestInfo * records, uint32_t * prob_count,
                                                          const uint32_t * __restrict__ prob_done, GibbsHeader * hdr, uint32_t * remaining,
                                                          uint32_t * active_problem, uint32_t * req_other,
                                                          const double * __restrict__ dist, unsigned long long * tab_key, uint32_t * tab_count,
                                                          uint32_t * tab_first, unsigned long long * debug_counts, const uint32_t chains_per_wave) {
    // A wave runs the union of the paths its chains take at every draw, and 60 000 chains in waves of 64 are one wave per SIMD
    // with nothing to hide a latency behind: the first chains_per_wave lanes of a wave carry a chain, the others leave.
    if ((threadIdx.x & 63u) >= chains_per_wave) return;
    const uint32_t ci = ((blockIdx.x * 256 + threadIdx.x) >> 6) * chains_per_wave + (threadIdx.x & 63u);
    if (ci >= num_chains) return;
    uint32_t n_draws = 0, n_lookups = 0, n_off_mode = 0, n_walked = 0, n_keys = 0;  // RPVG_HIP_GIBBS_DEBUG
    uint32_t flag = ch.flag[ci];
    if (flag & kChainDone) return;
    const uint32_t p = ch.problem[ci];
    const uint32_t G = mat_cols[pr.matrix[p]];
    const uint32_t done = prob_done[p];
    const uint32_t burn = pr.burn[p], its = pr.its[p];
    const uint32_t total_its = burn + its;
    unsigned long long pos = ch.pos[ci];
    uint32_t cur0 = ch.cur[2 * ci], cur1 = ch.cur[2 * ci + 1];
    uint32_t iter = ch.iter[ci];
    uint32_t slot = flag & 1u;
    const uint64_t col0 = pr.col_off[p];
    RequestInfo * records_p = records + col0;
    const uint64_t tab = pr.tab_off[p];
    const uint64_t tab_mask = pr.tab_off[p + 1] - tab - 1;
    const uint32_t chain_in_problem = ci - static_cast<uint32_t>(pr.chain_off[p]);
    // the distribution each slot drew from last
    uint32_t held_other[2] = {kPending, kPending};
    RequestInfo held[2];
    // the run of equal samples being counted
    unsigned long long run_key = kEmptyKey;
    uint32_t run_length = 0;
    uint64_t run_at = 0;
    bool waiting = false;
    // The generator's words of the next four draws wait in registers: every draw asks for the pair four draws ahead, the same
    // two loads in every lane (a window of sixteen words in LDS, refilled by whichever lanes ran out, put an 80-instruction
    // block that some lane of the wave needed at nearly every draw into the loop; a pair fetched one draw ahead arrived late).
    constexpr int kAheadDraws = 4;
    uint32_t ahead_first[kAheadDraws], ahead_second[kAheadDraws];
#pragma unroll
    for (int d = 0; d < kAheadDraws; ++d) {
        ahead_first[d] = (G >= 2) ? stream[pos + 2 * d] : 0u;  // (past a slice: the next chain's words, or the slack behind the last)
        ahead_second[d] = (G >= 2) ? stream[pos + 2 * d + 1] : 0u;
    }
    while (true) {
        uint32_t drawn = 0;
        if (G >= 2) {
            const uint32_t other = (GS == 2) ? (slot == 0 ? cur1 : cur0) : 0;
            const uint32_t s = (GS == 2) ? slot : 0;
            if (held_other[s] != other) {
                const RequestInfo record = records_p[other];  // (fields behind a state below `done` were written by earlier launches)
                const uint32_t m = record.state;
                if (m == 0) {
                    if (atomicCAS(&records_p[other].state, 0u, kPending) == 0u) {
                        const uint32_t k = atomicAdd(prob_count + p, 1u);
                        req_other[col0 + k] = other;
                        atomicExch(&records_p[other].state, k + 1);
                        if (k == done) active_problem[atomicAdd(&hdr->num_active, 1u)] = p;  // the problem's first request of this round
                    }
                    waiting = true;
                    break;
                }
                if (m == kPending || m - 1 >= done) {
                    waiting = true;
                    break;
                }
                held[s] = record;
                held_other[s] = other;
                ++n_lookups;
            }
            ++n_draws;
            const double u = rpvg_streams::canonicalFromWords(ahead_first[0], ahead_second[0]);
#pragma unroll
            for (int d = 0; d + 1 < kAheadDraws; ++d) {
                ahead_first[d] = ahead_first[d + 1];
                ahead_second[d] = ahead_second[d + 1];
            }
            ahead_first[kAheadDraws - 1] = stream[pos + 2 * kAheadDraws];
            ahead_second[kAheadDraws - 1] = stream[pos + 2 * kAheadDraws + 1];
            pos += 2;
            if (held[s].mode_below < u && u <= held[s].mode_upto) {
                drawn = held[s].mode;
            } else {
                // std::lower_bound over the partial sums: the bucket's column if u lies between its partial sums, else from
                // there back while the column before reaches u too, on while this one does not (the last partial sum is 1 > u)
                const double * cp = dist + held[s].dist_off;
                const DrawBucket bucket = reinterpret_cast<const DrawBucket *>(cp + G)[min(static_cast<uint32_t>(u * static_cast<double>(G)), G - 1)];
                uint32_t k = static_cast<uint32_t>(bucket.column);
                ++n_off_mode;
                if (!(bucket.below < u && u <= bucket.upto)) {
                    ++n_walked;
                    while (k > 0 && cp[k - 1] >= u) --k;
                    while (cp[k] < u) ++k;
                }
                drawn = k;
            }
        }
        if (GS == 1 || slot == 0) cur0 = drawn; else cur1 = drawn;
        ++slot;
        if (slot < GS) continue;
        slot = 0;
        if (iter >= burn) {
            const uint32_t lo = (GS == 2) ? min(cur0, cur1) : cur0;
            const uint32_t hi = (GS == 2) ? max(cur0, cur1) : cur0;
            const unsigned long long key = (static_cast<unsigned long long>(lo) << 32) | hi;
            if (key == run_key) {
                ++run_length;
            } else {
                if (run_length) atomicAdd(tab_count + run_at, run_length);
                uint64_t h = mixKey(key) & tab_mask;
                while (true) {
                    const unsigned long long old = atomicCAS(tab_key + tab + h, kEmptyKey, key);
                    if (old == kEmptyKey || old == key) break;
                    h = (h + 1) & tab_mask;
                }
                run_key = key;
                run_at = tab + h;
                run_length = 1;
                ++n_keys;
                atomicMin(tab_first + run_at, chain_in_problem * its + (iter - burn));  // later samples of the run come later
            }
        }
        ++iter;
        if (iter == total_its) {
            flag = kChainDone;
            break;
        }
    }
    if (run_length) atomicAdd(tab_count + run_at, run_length);
    ch.pos[ci] = pos;
    ch.cur[2 * ci] = cur0;
    ch.cur[2 * ci + 1] = cur1;
    ch.iter[ci] = iter;
    ch.flag[ci] = (flag & kChainDone) | slot;
    if (waiting) atomicAdd(remaining + round, 1u);
    if (debug_counts) {  // per round: chains that ran, draws, most draws of a chain, record lookups, draws off the mode, walks, key changes
        unsigned long long * c = debug_counts + 8ull * round;
        atomicAdd(c + 0, 1ull);
        atomicAdd(c + 1, static_cast<unsigned long long>(n_draws));
        atomicMax(c + 2, static_cast<unsigned long long>(n_draws));
        atomicAdd(c + 3, static_cast<unsigned long long>(n_lookups));
        atomicAdd(c + 4, static_cast<unsigned long long>(n_off_mode));
        atomicAdd(c + 5, static_cast<unsigned long long>(n_walked));
        atomicAdd(c + 6, static_cast<unsigned long long>(n_keys));
    }
}

// ---- the requests of a round: work items and storage -----------------------------------------------------
// One workgroup: prefix sums over the problems with new requests — work items (up to four requests x four candidate
// columns each), doubles of storage, requests — then the requests' own records.
struct ActiveEntry {
    unsigned long long item_off;  // turns of work before this problem's
    unsigned long long dist_off;  // storage of its first new request
    uint32_t problem, first, count, req_off;  // new requests [first, first + count) of the problem; position in the round's list
};

// A wave's turn = up to this many consecutive work items of ONE problem: finding the problem of an item is a bisection of
// eleven dependent loads for the 2 500 problems of a lane's first round, and the matrix's description a few more — as long as
// the rows of a small matrix take.  Matrices with many rows get one item per turn (eight of them in one wave were the
// round's tail: 4.0 against 3.2 ms, and 1.9 against 0.6 ms for the later rounds' few items).
__device__ __forceinline__ uint32_t itemsPerTurn(const uint64_t rows) { return rows <= 256 ? 8u : rows <= 512 ? 4u : rows <= 1024 ? 2u : 1u; }

__global__ __launch_bounds__(1024) void gibbsRequestOffsetsKernel(const GibbsProblems pr, const uint32_t * __restrict__ mat_cols,
                                                                  const uint64_t * __restrict__ mat_rows, GibbsHeader * hdr,
                                                                  const uint32_t * __restrict__ active_problem, const uint32_t * __restrict__ prob_count,
                                                                  uint32_t * prob_done, ActiveEntry * entries, uint32_t * new_req,
                                                                  const uint32_t * __restrict__ req_other, RequestInfo * records,
                                                                  const unsigned long long dist_capacity, const uint32_t tiled) {
    __shared__ unsigned long long wave_items[16], wave_cols[16], wave_reqs[16];
    __shared__ unsigned long long carry_items, carry_cols, carry_reqs;
    const uint32_t tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const uint32_t num_active = hdr->num_active;
    // A round of many requests (the first: every chain's start) goes through the workgroup-per-tile kernel, which reads
    // less per evaluation; a round of few through the wave-per-item kernel, which makes sixteen times the work items of them
    // (a workgroup walks ALL rows of its matrix: 1.2-1.7 against 0.6-1.0 ms for the rounds behind the first).
    __shared__ unsigned int round_requests;
    if (tid == 0) {
        carry_items = 0;
        carry_cols = hdr->used;
        carry_reqs = 0;
        round_requests = 0;
    }
    __syncthreads();
    {
        unsigned int mine = 0;
        for (uint32_t a = tid; a < num_active; a += 1024) mine += prob_count[active_problem[a]] - prob_done[active_problem[a]];
        if (mine) atomicAdd(&round_requests, mine);
    }
    __syncthreads();
    const bool tiled_round = tiled != 0;
    double evals = 0.0;
    for (uint32_t base = 0; base < num_active; base += 1024) {
        const uint32_t a = base + tid;
        unsigned long long items = 0, cols = 0, reqs = 0;
        uint32_t p = 0, first = 0, G = 0;
        if (a < num_active) {
            p = active_problem[a];
            const uint32_t m = pr.matrix[p];
            G = mat_cols[m];
            first = prob_done[p];
            reqs = prob_count[p] - first;
            cols = reqs * distributionDoubles(G);
            if (tiled_round) {  // gibbsConditionalTileKernel: (four requests) x (64 candidate columns) per workgroup
                items = ((reqs + 3) / 4) * ((G + kCondCands - 1) / kCondCands);
            } else {
                const unsigned long long work_items = ((reqs + 3) / 4) * ((G + 3) / 4);
                const uint32_t per_turn = itemsPerTurn(mat_rows[m]);
                items = (work_items + per_turn - 1) / per_turn;
            }
            evals += static_cast<double>(mat_rows[m]) * static_cast<double>(reqs * G);
        }
        unsigned long long scan_items = items, scan_cols = cols, scan_reqs = reqs;
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long ui = __shfl_up(scan_items, d), uc = __shfl_up(scan_cols, d), ur = __shfl_up(scan_reqs, d);
            if (lane >= d) {
                scan_items += ui;
                scan_cols += uc;
                scan_reqs += ur;
            }
        }
        if (lane == 63) {
            wave_items[wave] = scan_items;
            wave_cols[wave] = scan_cols;
            wave_reqs[wave] = scan_reqs;
        }
        __syncthreads();
        unsigned long long before_items = carry_items, before_cols = carry_cols, before_reqs = carry_reqs;
        for (int w = 0; w < wave; ++w) {
            before_items += wave_items[w];
            before_cols += wave_cols[w];
            before_reqs += wave_reqs[w];
        }
        if (a < num_active) {
            ActiveEntry e;
            e.item_off = before_items + scan_items - items;
            e.dist_off = before_cols + scan_cols - cols;
            e.problem = p;
            e.first = first;
            e.count = static_cast<uint32_t>(reqs);
            e.req_off = static_cast<uint32_t>(before_reqs + scan_reqs - reqs);
            // past the capacity nothing of this round is evaluated (the host reads the error): the records stay harmless
            const bool fits = e.dist_off + cols <= dist_capacity;
            entries[a] = e;
            const uint64_t col0 = pr.col_off[p];
            for (uint32_t j = 0; j < e.count; ++j) {
                const uint32_t id = static_cast<uint32_t>(col0) + first + j;
                const uint32_t record = static_cast<uint32_t>(col0) + req_other[id];
                new_req[e.req_off + j] = record;
                records[record].dist_off = fits ? e.dist_off + static_cast<unsigned long long>(j) * distributionDoubles(G) : 0;
                records[record].mode = p;  // (the distribution kernel reads the problem here and writes the mode over it)
            }
        }
        __syncthreads();
        if (tid == 1023) {
            carry_items = before_items + scan_items;
            carry_cols = before_cols + scan_cols;
            carry_reqs = before_reqs + scan_reqs;
        }
        __syncthreads();
    }
    if (evals != 0.0) atomicAdd(&hdr->evals, evals);
    // the round's requests are complete by the time the next advance starts (stream order) — unless the round does not fit
    // the storage: then nothing of it is evaluated, its chains keep waiting and the host reads the error
    if (carry_cols <= dist_capacity) {
        for (uint32_t a = tid; a < num_active; a += 1024) prob_done[entries[a].problem] = entries[a].first + entries[a].count;
    }
    if (tid == 0) {
        const bool fits = carry_cols <= dist_capacity;
        if (!fits) atomicOr(&hdr->error, kErrDistributions);
        hdr->cur_active = fits ? num_active : 0;
        hdr->cur_requests = fits ? static_cast<uint32_t>(carry_reqs) : 0;
        hdr->num_items = fits ? carry_items : 0;
        hdr->used = fits ? carry_cols : hdr->used;
        hdr->total_requests += static_cast<uint32_t>(carry_reqs);
        hdr->num_active = 0;
        hdr->tiled = tiled_round ? 1u : 0u;
    }
}

// ---- conditionals: src/path_estimator.cpp:527-545 -----------------------------------------------------------
// groupConditionalKernel's arithmetic (loglik.hip) on the requests the chains wrote.  One wave per work item = up to FOUR
// requests of one problem x four candidate columns: per row one read of the noise, of the four other members' columns and
// of the four candidates for sixteen log arguments (a request at a time read the matrix once per request: 12 bytes per
// evaluation, 80 GB per configs[4] batch).  Waves take items in turn; the column's log frequency is added on the way out (:547).
template <int GS, int OTHERS>
__device__ __forceinline__ void conditionalItem(const LogTableEntry * lt, const int lane, const double * __restrict__ M, const uint64_t R,
                                                const uint32_t G, const double * __restrict__ cnt, const double * __restrict__ nz,
                                                const uint64_t fast_end, const uint64_t mid_end, const uint32_t * other_col, const uint32_t k0,
                                                const uint32_t num_others, const double * __restrict__ lf, double * const * out) {
    constexpr int kCand = 4;
    constexpr double divisor = static_cast<double>(GS);
    const double * other[OTHERS];
#pragma unroll
    for (int o = 0; o < OTHERS; ++o) other[o] = (GS == 2) ? M + static_cast<uint64_t>(other_col[o]) * R : nullptr;
    const double * cand[kCand];
#pragma unroll
    for (int c = 0; c < kCand; ++c) cand[c] = M + static_cast<uint64_t>(min(k0 + c, G - 1)) * R;
    constexpr int kOut = OTHERS * kCand;
    double acc[kOut];
    LogProduct prod[kOut];
#pragma unroll
    for (int t = 0; t < kOut; ++t) acc[t] = 0.0;
    struct RowValues {
        double noise, other[OTHERS], cand[kCand];
    };
    auto load = [&](const uint64_t i, RowValues & v) {
        v.noise = nz[i];
#pragma unroll
        for (int o = 0; o < OTHERS; ++o) v.other[o] = (GS == 2) ? other[o][i] : 0.0;
#pragma unroll
        for (int c = 0; c < kCand; ++c) v.cand[c] = cand[c][i];
    };
    // the log arguments of a row, in the reference's order of additions: (noise + other / g) + candidate / g
    auto arguments = [&](const RowValues & v, double (&xs)[kOut]) {
#pragma unroll
        for (int o = 0; o < OTHERS; ++o) {
            double base = v.noise;
            if (GS == 2) base += v.other[o] / divisor;
#pragma unroll
            for (int c = 0; c < kCand; ++c) xs[o * kCand + c] = base + v.cand[c] / divisor;
        }
    };
    // rows of read count 1 (LogProduct, common.hpp): a lane multiplies kFoldFactors factors between folds.  Three rows per
    // step (then two, then one; four take 256 registers and the SIMD's second wave with them), their loads issued together: with 180 registers two waves share a SIMD, and a wave that waits for the
    // nine loads of one row before it asks for the next is all latency (1.0 T evaluations/s: 8 waves per CU x 1 024
    // evaluations per 2 us)
    for (uint64_t seg = 0; seg < fast_end; seg += 64 * kFoldFactors) {
        const uint64_t seg_end = (fast_end - seg) < 64 * kFoldFactors ? fast_end : seg + 64 * kFoldFactors;
        uint64_t i = seg + lane;
        for (; i + 128 < seg_end; i += 192) {
            RowValues a, b, c;
            load(i, a);
            load(i + 64, b);
            load(i + 128, c);
            double xs[kOut];
            arguments(a, xs);
#pragma unroll
            for (int t = 0; t < kOut; ++t) prod[t].mul(xs[t]);
            arguments(b, xs);
#pragma unroll
            for (int t = 0; t < kOut; ++t) prod[t].mul(xs[t]);
            arguments(c, xs);
#pragma unroll
            for (int t = 0; t < kOut; ++t) prod[t].mul(xs[t]);
        }
        for (; i + 64 < seg_end; i += 128) {
            RowValues a, b;
            load(i, a);
            load(i + 64, b);
            double xs[kOut];
            arguments(a, xs);
#pragma unroll
            for (int t = 0; t < kOut; ++t) prod[t].mul(xs[t]);
            arguments(b, xs);
#pragma unroll
            for (int t = 0; t < kOut; ++t) prod[t].mul(xs[t]);
        }
        if (i < seg_end) {
            RowValues a;
            load(i, a);
            double xs[kOut];
            arguments(a, xs);
#pragma unroll
            for (int t = 0; t < kOut; ++t) prod[t].mul(xs[t]);
        }
#pragma unroll
        for (int t = 0; t < kOut; ++t) prod[t].fold();
    }
    // read counts 2 .. kMidMaxCount: the factor that many times; the rest: a logarithm per row
    for (uint64_t i = fast_end + lane; i < mid_end; i += 64) {
        RowValues a;
        load(i, a);
        double xs[kOut];
        arguments(a, xs);
        const int c = static_cast<int>(cnt[i]);
        for (int k = 0; k < c; ++k) {
#pragma unroll
            for (int t = 0; t < kOut; ++t) prod[t].mul(xs[t]);
        }
#pragma unroll
        for (int t = 0; t < kOut; ++t) prod[t].fold();
    }
    for (uint64_t i = mid_end + lane; i < R; i += 64) {
        RowValues a;
        load(i, a);
        double xs[kOut];
        arguments(a, xs);
        const double c = cnt[i];
#pragma unroll
        for (int t = 0; t < kOut; ++t) acc[t] = fma(c, logPositive(xs[t], lt), acc[t]);
    }
    if (mid_end) {
#pragma unroll
        for (int t = 0; t < kOut; ++t) acc[t] += prod[t].value(lt);
    }
#pragma unroll
    for (int o = 0; o < OTHERS; ++o) {
#pragma unroll
        for (int c = 0; c < kCand; ++c) {
            const double total = waveSumF64(acc[o * kCand + c]);
            if (lane == 0 && static_cast<uint32_t>(o) < num_others && k0 + c < G) out[o][k0 + c] = total + lf[k0 + c];
        }
    }
}

template <int GS>
__global__ __launch_bounds__(256) void gibbsConditionalKernel(const GibbsProblems pr, const GibbsHeader * __restrict__ hdr,
                                                              const ActiveEntry * __restrict__ entries, const uint32_t * __restrict__ req_other,
                                                              const uint64_t * __restrict__ mat_val_off, const uint64_t * __restrict__ mat_row_off,
                                                              const uint32_t * __restrict__ mat_fast, const uint32_t * __restrict__ mat_mid,
                                                              const uint64_t * __restrict__ mat_rows, const uint32_t * __restrict__ mat_cols,
                                                              const double * __restrict__ values, const double * __restrict__ row_count,
                                                              const double * __restrict__ row_noise, double * __restrict__ dist) {
    const unsigned long long num_items = hdr->num_items;
    if (num_items == 0 || hdr->tiled) return;
    __shared__ LogTableEntry lt[kLogTableSize];
    loadLogTable(lt);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint32_t num_active = hdr->cur_active;
    const unsigned long long num_waves = static_cast<unsigned long long>(gridDim.x) * 4;
    for (unsigned long long turn = static_cast<unsigned long long>(blockIdx.x) * 4 + (threadIdx.x >> 6); turn < num_items; turn += num_waves) {
        uint32_t lo = 0, hi = num_active - 1;  // last entry with item_off <= turn
        while (lo < hi) {
            const uint32_t mid = lo + ((hi - lo + 1) >> 1);
            if (entries[mid].item_off <= turn) lo = mid; else hi = mid - 1;
        }
        const ActiveEntry e = entries[lo];
        const uint32_t p = e.problem;
        const uint32_t m = pr.matrix[p];
        const uint64_t R = mat_rows[m];
        const uint32_t G = mat_cols[m];
        const double * M = values + mat_val_off[m];
        const double * cnt = row_count + mat_row_off[m];
        const double * nz = row_noise + mat_row_off[m];
        const uint64_t col0 = pr.col_off[p];
        const double * lf = pr.log_freq + col0;
        const uint64_t fast_end = mat_fast[m], mid_end = mat_mid[m];
        // consecutive items = the same four candidate columns under the next four requests (the matrix's columns are
        // read about once per round, the requests' own columns are a dozen and stay in L2)
        const uint32_t other_groups = (e.count + 3) / 4;
        const uint32_t work_items = other_groups * ((G + 3) / 4);
        const uint32_t per_turn = itemsPerTurn(R);
        const uint32_t local_begin = static_cast<uint32_t>(turn - e.item_off) * per_turn;
        const uint32_t local_end = min(work_items, local_begin + per_turn);
        for (uint32_t local = local_begin; local < local_end; ++local) {
            const uint32_t j0 = (local % other_groups) * 4;  // first request of the item among the problem's new ones
            const uint32_t k0 = (local / other_groups) * 4;
            const uint32_t num_others = min(4u, e.count - j0);
            uint32_t other_col[4];
            double * out[4];
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                const uint32_t j = j0 + min(static_cast<uint32_t>(o), num_others - 1);
                other_col[o] = (GS == 2) ? req_other[col0 + e.first + j] : 0;
                out[o] = dist + e.dist_off + static_cast<unsigned long long>(j) * distributionDoubles(G);
            }
            if (num_others == 1) {
                conditionalItem<GS, 1>(lt, lane, M, R, G, cnt, nz, fast_end, mid_end, other_col, k0, 1, lf, out);
            } else {
                conditionalItem<GS, 4>(lt, lane, M, R, G, cnt, nz, fast_end, mid_end, other_col, k0, num_others, lf, out);
            }
        }
    }
}

// ---- conditionals, the rows staged through LDS ---------------------------------------------------------------
// A workgroup per (problem, up to four of its new requests, block of 64 candidate columns).  64 rows at a time go through
// LDS — row-major, [4 others | 64 candidates | noise | read count] — loaded by lane = row (512-byte requests, seventeen in
// flight per thread) where the wave-per-item kernel read nine columns per row for sixteen log arguments at two waves per
// SIMD: 2.2 instead of 4.5 bytes per evaluation, and twice the waves.  Thread = (tile of four candidates, one of sixteen row
// slices): per row two 16-byte LDS reads for each side and sixteen running products, as in the diploid search's tile
// kernel; the slices are multiplied together inside the wave, the four waves' logarithms added through LDS.

template <int GS>
__global__ __launch_bounds__(256) void gibbsConditionalTileKernel(const GibbsProblems pr, const GibbsHeader * __restrict__ hdr,
                                                                  const ActiveEntry * __restrict__ entries, const uint32_t * __restrict__ req_other,
                                                                  const uint64_t * __restrict__ mat_val_off, const uint64_t * __restrict__ mat_row_off,
                                                                  const uint32_t * __restrict__ mat_fast, const uint32_t * __restrict__ mat_mid,
                                                                  const uint64_t * __restrict__ mat_rows, const uint32_t * __restrict__ mat_cols,
                                                                  const double * __restrict__ values, const double * __restrict__ row_count,
                                                                  const double * __restrict__ row_noise, double * __restrict__ dist) {
    constexpr double divisor = static_cast<double>(GS);
    const unsigned long long num_items = hdr->num_items;
    if (num_items == 0 || !hdr->tiled) return;
    __shared__ LogTableEntry lt[kLogTableSize];
    __shared__ __attribute__((aligned(16))) double staged[kCondRows * kCondStride];
    __shared__ double wave_logs[4 * 16 * 16];
    loadLogTable(lt);
    const uint32_t tid = threadIdx.x;
    const uint32_t tile = tid & 15;   // candidates 4 tile .. 4 tile + 3 of the block
    const uint32_t slice = tid >> 4;  // rows slice, slice + 16, ... of a staged chunk
    const uint32_t num_active = hdr->cur_active;
    for (unsigned long long item = blockIdx.x; item < num_items; item += gridDim.x) {
        uint32_t lo = 0, hi = num_active - 1;  // last entry with item_off <= item
        while (lo < hi) {
            const uint32_t mid = lo + ((hi - lo + 1) >> 1);
            if (entries[mid].item_off <= item) lo = mid; else hi = mid - 1;
        }
        const ActiveEntry e = entries[lo];
        const uint32_t p = e.problem;
        const uint32_t m = pr.matrix[p];
        const uint64_t R = mat_rows[m];
        const uint32_t G = mat_cols[m];
        const double * M = values + mat_val_off[m];
        const double * cnt = row_count + mat_row_off[m];
        const double * nz = row_noise + mat_row_off[m];
        const uint64_t col0 = pr.col_off[p];
        const uint64_t fast_end = mat_fast[m], mid_end = mat_mid[m];
        const uint32_t other_groups = (e.count + 3) / 4;
        const uint32_t local = static_cast<uint32_t>(item - e.item_off);
        const uint32_t j0 = (local % other_groups) * 4;
        const uint32_t k0 = (local / other_groups) * kCondCands;
        const uint32_t num_others = min(4u, e.count - j0);
        // the column a thread stages: 0-3 others, 4-67 candidates, 68 noise, 69 read count; columns wave, wave + 4, ...
        LogProduct prod[16];
        double acc[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[t] = 0.0;
        uint32_t factors = 0;  // fast factors since the last fold
        const uint32_t stage_row = tid & 63, stage_col0 = tid >> 6;
        double fetched[18];
        auto fetch = [&](const uint64_t r0) {
            const uint64_t i_stage = r0 + stage_row;
            const bool row_there = i_stage < R;
#pragma unroll
            for (uint32_t q = 0; q < 18; ++q) {
                const uint32_t col = stage_col0 + 4 * q;
                double v = (col == 68) ? 1.0 : 0.0;  // a row past the end: argument 1, a factor that changes nothing
                if (row_there && col < kCondStride) {
                    if (col < 4) {
                        if (GS == 2) v = M[static_cast<uint64_t>(req_other[col0 + e.first + j0 + min(col, num_others - 1)]) * R + i_stage];
                    } else if (col < 68) {
                        if (k0 + col - 4 < G) v = M[static_cast<uint64_t>(k0 + col - 4) * R + i_stage];
                    } else if (col == 68) {
                        v = nz[i_stage];
                    } else {
                        v = cnt[i_stage];
                    }
                }
                fetched[q] = v;
            }
        };
        for (uint64_t r0 = 0; r0 < R; r0 += kCondRows) {
            fetch(r0);  // (fetching the chunk after this one under its arithmetic takes 254 registers: one wave per SIMD)
            __syncthreads();  // the chunk before has been read
#pragma unroll
            for (uint32_t q = 0; q < 18; ++q) {
                const uint32_t col = stage_col0 + 4 * q;
                if (col < kCondStride) staged[stage_row * kCondStride + col] = fetched[q];
            }
            __syncthreads();
            if (factors + 4 > kFoldFactors) {
#pragma unroll
                for (int t = 0; t < 16; ++t) prod[t].fold();
                factors = 0;
            }
            if (k0 + 4 * tile < G) {  // (a block of fewer than 64 candidates: tiles without one)
#pragma unroll 1
            for (uint32_t step = 0; step < kCondRows / 16; ++step) {
                const uint32_t row = slice + 16 * step;
                const uint64_t i = r0 + row;
                const double * at = staged + row * kCondStride;
                const double2 o01 = *reinterpret_cast<const double2 *>(at), o23 = *reinterpret_cast<const double2 *>(at + 2);
                const double2 c01 = *reinterpret_cast<const double2 *>(at + 4 + 4 * tile), c23 = *reinterpret_cast<const double2 *>(at + 6 + 4 * tile);
                const double noise = at[68];
                const double others[4] = {o01.x, o01.y, o23.x, o23.y};
                const double half[4] = {c01.x / divisor, c01.y / divisor, c23.x / divisor, c23.y / divisor};
                double xs[16];
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    double base = noise;
                    if (GS == 2) base += others[o] / divisor;
#pragma unroll
                    for (int c = 0; c < 4; ++c) xs[o * 4 + c] = base + half[c];
                }
                if (i < fast_end || i >= R) {  // (past the end: factors of 1)
#pragma unroll
                    for (int t = 0; t < 16; ++t) prod[t].mul(xs[t]);
                    ++factors;
                } else if (i < mid_end) {
                    const int count = static_cast<int>(at[69]);
#pragma unroll
                    for (int t = 0; t < 16; ++t) prod[t].fold();
                    for (int k = 0; k < count; ++k) {
#pragma unroll
                        for (int t = 0; t < 16; ++t) prod[t].mul(xs[t]);
                    }
#pragma unroll
                    for (int t = 0; t < 16; ++t) prod[t].fold();
                    factors = 0;
                } else {
                    const double count = at[69];
#pragma unroll
                    for (int t = 0; t < 16; ++t) acc[t] = fma(count, logPositive(xs[t], lt), acc[t]);
                }
            }
            }
        }
        // the sixteen slices: four inside every wave (lanes 16 apart), then the waves' logarithms through LDS
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            prod[t].fold();
            for (int d = 16; d < 64; d <<= 1) {
                LogProduct partner;
                partner.p = __shfl_xor(prod[t].p, d);
                partner.e = __shfl_xor(prod[t].e, d);
                prod[t].join(partner);
                prod[t].fold();
                acc[t] += __shfl_xor(acc[t], d);
            }
        }
        __syncthreads();  // (wave_logs of the item before have been read)
        if ((tid & 63) < 16) {
#pragma unroll
            for (int t = 0; t < 16; ++t) wave_logs[((tid >> 6) * 16 + tile) * 16 + t] = acc[t] + prod[t].value(lt);
        }
        __syncthreads();
        {
            const uint32_t out_tile = tid >> 4, out_value = tid & 15;  // 16 tiles x (4 others x 4 candidates)
            const uint32_t o = out_value >> 2, c = out_value & 3;
            const uint32_t k = k0 + 4 * out_tile + c;
            if (o < num_others && k < G) {
                const double total = (wave_logs[(0 * 16 + out_tile) * 16 + out_value] + wave_logs[(1 * 16 + out_tile) * 16 + out_value]) +
                                     (wave_logs[(2 * 16 + out_tile) * 16 + out_value] + wave_logs[(3 * 16 + out_tile) * 16 + out_value]);
                dist[e.dist_off + static_cast<unsigned long long>(j0 + o) * distributionDoubles(G) + k] = total + pr.log_freq[col0 + k];
            }
        }
    }
}

__device__ __forceinline__ double waveMaxF64(double v) {
    for (int d = 32; d > 0; d >>= 1) v = fmax(v, __shfl_xor(v, d));
    return v;
}

// ---- the distribution of a request: src/path_estimator.cpp:547-555 and discrete_distribution's own set-up ----
// log-sum-exp over the columns (the reference adds them up one by one with add_log), exp(value - sum), then what
// libstdc++ does with the weights (bits/random.tcc:2665-2676): divided by their sum, partial sums, the last one 1.
// One wave per request; a lane owns a contiguous stretch of columns for the partial sums.  The partial sums either side
// of the largest weight go into the request's record (what the chains look at first).
__global__ __launch_bounds__(256) void gibbsDistributionKernel(const GibbsProblems pr, const GibbsHeader * __restrict__ hdr,
                                                               const uint32_t * __restrict__ new_req,
                                                               const uint32_t * __restrict__ mat_cols, RequestInfo * records,
                                                               double * __restrict__ dist, const double ask_ahead, uint32_t * prob_count,
                                                               const uint32_t * __restrict__ prob_done, uint32_t * active_problem,
                                                               uint32_t * req_other, GibbsHeader * counters) {
    const int lane = threadIdx.x & 63;
    const uint32_t num_new = hdr->cur_requests;
    const uint32_t num_waves = gridDim.x * 4;
    for (uint32_t q = blockIdx.x * 4 + (threadIdx.x >> 6); q < num_new; q += num_waves) {
        const uint32_t id = new_req[q];  // the request's record
        const uint32_t p = records[id].mode;
        const uint32_t G = mat_cols[pr.matrix[p]];
        const uint64_t col0 = pr.col_off[p];
        const uint32_t done = prob_done[p];
        double * v = dist + records[id].dist_off;
        double largest = -INFINITY;
        uint32_t largest_at = 0;
        for (uint32_t k = lane; k < G; k += 64) {
            const double value = v[k];
            if (value > largest) {
                largest = value;
                largest_at = k;
            }
        }
        const double wave_largest = waveMaxF64(largest);
        // the first column that holds the largest value
        uint32_t mode = (largest == wave_largest) ? largest_at : 0xffffffffu;
        for (int d = 32; d > 0; d >>= 1) mode = min(mode, static_cast<uint32_t>(__shfl_xor(static_cast<int>(mode), d)));
        if (mode >= G) mode = 0;  // nothing but NaNs
        largest = wave_largest;
        double sum = 0.0;
        for (uint32_t k = lane; k < G; k += 64) sum += exp(v[k] - largest);
        sum = waveSumF64(sum);
        const double log_sum = largest + log(sum);
        double weight_sum = 0.0;
        for (uint32_t k = lane; k < G; k += 64) {
            const double w = exp(v[k] - log_sum);
            v[k] = w;
            weight_sum += w;
        }
        weight_sum = waveSumF64(weight_sum);
        const uint32_t stretch = (G + 63) / 64;
        const uint32_t k_begin = min(G, lane * stretch), k_end = min(G, (lane + 1) * stretch);
        // partial sums = (sum of the stretches before the lane's, added up one after the other) + (sum inside the stretch):
        // the last partial sum of a stretch IS the next stretch's start, so the sums never decrease across a boundary
        double mine = 0.0;
        for (uint32_t k = k_begin; k < k_end; ++k) mine += v[k] / weight_sum;
        double before = 0.0, through = 0.0;
        for (int l = 0; l < 64; ++l) {
            if (lane == l) before = through;
            through += readLaneF64(mine, l);
        }
        DrawBucket * buckets_of = reinterpret_cast<DrawBucket *>(v + G);
        const double buckets = static_cast<double>(G);
        double inside = 0.0;
        double below = -1.0, upto = 2.0;
        bool has_mode = false;
        uint32_t bucket = (k_begin == 0 || k_begin >= G) ? 0 : min(G, static_cast<uint32_t>(ceil(before * buckets)));  // where the stretch before ends
        for (uint32_t k = k_begin; k < k_end; ++k) {
            const double previous = (k == 0) ? -1.0 : before + inside;
            if (k == mode) {
                below = previous;
                has_mode = true;
            }
            const double share = v[k] / weight_sum;
            inside += share;
            const double partial = (k + 1 == G) ? 1.0 : before + inside;
            v[k] = partial;
            if (k == mode) upto = partial;
            // RPVG_HIP_GIBBS_ASK_AHEAD=share (A/B): a column a chain is likely to draw here is the other member of its next
            // draw — its conditional is asked for now (next round's requests) instead of when a chain gets there.  The
            // sampler's draws do not change: the memo only fills earlier (and with conditionals nobody may ever draw from).
            if (share >= ask_ahead) {
                RequestInfo * ahead = records + col0 + k;
                if (ahead->state == 0 && atomicCAS(&ahead->state, 0u, kPending) == 0u) {
                    const uint32_t number = atomicAdd(prob_count + p, 1u);
                    req_other[col0 + number] = k;
                    atomicExch(&ahead->state, number + 1);
                    if (number == done) active_problem[atomicAdd(&counters->num_active, 1u)] = p;
                }
            }
            const uint32_t bucket_end = (k + 1 == G) ? G : min(G, static_cast<uint32_t>(ceil(partial * buckets)));
            for (; bucket < bucket_end; ++bucket) buckets_of[bucket] = DrawBucket{previous, partial, k};
        }
        // (every lane has read the problem above: the wave runs in step)
        if (has_mode) {
            records[id].mode_below = below;
            records[id].mode_upto = upto;
            records[id].mode = mode;
        }
    }
}

// ---- the sampled sets, in the order the reference meets them -----------------------------------------------
__global__ __launch_bounds__(256) void gibbsCountSetsKernel(const GibbsProblems pr, const unsigned long long * __restrict__ tab_key,
                                                            unsigned long long * __restrict__ set_count) {
    __shared__ uint32_t total;
    const uint32_t p = blockIdx.x;
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    const uint64_t tab = pr.tab_off[p], size = pr.tab_off[p + 1] - tab;
    uint32_t n = 0;
    for (uint64_t s = threadIdx.x; s < size; s += 256) n += tab_key[tab + s] != kEmptyKey;
    if (n) atomicAdd(&total, n);
    __syncthreads();
    if (threadIdx.x == 0) set_count[p] = total;
}

// exclusive prefix in place over [P] (+ the total behind it)
__global__ __launch_bounds__(1024) void gibbsSetOffsetsKernel(const uint32_t num_problems, unsigned long long * set_off, GibbsHeader * hdr) {
    __shared__ unsigned long long wave_sum[16];
    __shared__ unsigned long long carry;
    const uint32_t tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < num_problems; base += 1024) {
        const uint32_t p = base + tid;
        const unsigned long long mine = p < num_problems ? set_off[p] : 0;
        unsigned long long scan = mine;
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long up = __shfl_up(scan, d);
            if (lane >= d) scan += up;
        }
        if (lane == 63) wave_sum[wave] = scan;
        __syncthreads();
        unsigned long long before = carry;
        for (int w = 0; w < wave; ++w) before += wave_sum[w];
        if (p < num_problems) set_off[p] = before + scan - mine;
        __syncthreads();
        if (tid == 1023) carry = before + scan;
        __syncthreads();
    }
    if (tid == 0) {
        set_off[num_problems] = carry;
        hdr->total_sets = carry;
    }
}

// One workgroup per problem: its sets ordered by the sample that produced them first (the reference appends a set to
// path_group_sets when it first meets it, src/path_estimator.cpp:566-568) — rank by counting in LDS; a problem with more
// sets than fit leaves in table order with the sequence numbers, and the host orders it.
__global__ __launch_bounds__(256) void gibbsCollectKernel(const GibbsProblems pr, const unsigned long long * __restrict__ tab_key,
                                                          const uint32_t * __restrict__ tab_count, const uint32_t * __restrict__ tab_first,
                                                          const unsigned long long * __restrict__ set_off, uint32_t * __restrict__ out_first,
                                                          uint32_t * __restrict__ out_second, uint32_t * __restrict__ out_count,
                                                          uint32_t * __restrict__ out_seq, GibbsHeader * hdr) {
    __shared__ uint32_t seqs[kRankInLds];
    __shared__ uint32_t slots[kRankInLds];
    __shared__ uint32_t cursor;
    const uint32_t p = blockIdx.x;
    const unsigned long long out0 = set_off[p];
    const uint32_t n = static_cast<uint32_t>(set_off[p + 1] - out0);
    if (n == 0) return;
    if (threadIdx.x == 0) cursor = 0;
    __syncthreads();
    const uint64_t tab = pr.tab_off[p], size = pr.tab_off[p + 1] - tab;
    const bool in_lds = n <= kRankInLds;
    for (uint64_t s = threadIdx.x; s < size; s += 256) {
        const unsigned long long key = tab_key[tab + s];
        if (key == kEmptyKey) continue;
        const uint32_t at = atomicAdd(&cursor, 1u);
        if (in_lds) {
            seqs[at] = tab_first[tab + s];
            slots[at] = static_cast<uint32_t>(s);
        } else {
            out_first[out0 + at] = static_cast<uint32_t>(key >> 32);
            out_second[out0 + at] = static_cast<uint32_t>(key);
            out_count[out0 + at] = tab_count[tab + s];
            out_seq[out0 + at] = tab_first[tab + s];
        }
    }
    __syncthreads();
    if (!in_lds) {
        if (threadIdx.x == 0) atomicAdd(&hdr->unsorted, 1u);
        return;
    }
    for (uint32_t e = threadIdx.x; e < n; e += 256) {
        const uint32_t mine = seqs[e];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < n; ++j) rank += seqs[j] < mine;  // sequence numbers are distinct: one sample each
        const uint64_t s = slots[e];
        const unsigned long long key = tab_key[tab + s];
        out_first[out0 + rank] = static_cast<uint32_t>(key >> 32);
        out_second[out0 + rank] = static_cast<uint32_t>(key);
        out_count[out0 + rank] = tab_count[tab + s];
        out_seq[out0 + rank] = mine;
    }
}

uint64_t nextPowerOfTwo(uint64_t v) {
    uint64_t p = 16;
    while (p < v) p <<= 1;
    return p;
}

}  // namespace

extern "C" int rpvg_hip_group_gibbs(rpvg_hip_ctx * ctx, const rpvg_hip_groups * groups, const rpvg_hip_gibbs_spec * spec,
                                    rpvg_hip_gibbs_sets ** result_out) {
    RPVG_REQUIRE(ctx && groups && spec && result_out, "rpvg_hip_group_gibbs: NULL argument");
    *result_out = nullptr;
    const uint32_t P = spec->num_problems, NG = spec->num_generators, GS = spec->group_size;
    if (GS < 1 || GS > 2) {
        setError("rpvg_hip_group_gibbs: group size %u (the device sampler takes 1 and 2)", GS);
        return RPVG_HIP_ERR_UNSUPPORTED;
    }
    RPVG_REQUIRE(P == 0 || (spec->matrix && spec->num_chains && spec->num_burn_its && spec->num_gibbs_its && spec->log_freq &&
                            spec->generator_problem_off && spec->generator_problem && spec->generator_words),
                 "rpvg_hip_group_gibbs: NULL array");
    RPVG_REQUIRE(P == 0 || NG >= 1, "rpvg_hip_group_gibbs: no generators");
    auto result = std::make_unique<rpvg_hip_gibbs_sets>();
    result->num_problems = P;
    result->group_size = GS;
    result->set_off.assign(static_cast<size_t>(P) + 1, 0);
    result->words_consumed.assign(NG, 0);
    if (P == 0) {
        *result_out = result.release();
        return RPVG_HIP_OK;
    }

    std::unique_ptr<HostScope> scope(new HostScope("group_gibbs: host sizes"));
    // sizes: chains, columns, sample tables, stream capacities
    std::vector<uint64_t> chain_off(static_cast<size_t>(P) + 1, 0), col_off(static_cast<size_t>(P) + 1, 0), tab_off(static_cast<size_t>(P) + 1, 0);
    std::vector<uint64_t> words_needed(P, 0);
    long double dist_bound = 0;
    uint64_t out_capacity = 0;
    for (uint32_t p = 0; p < P; ++p) {
        RPVG_REQUIRE(spec->matrix[p] < groups->num_matrices, "rpvg_hip_group_gibbs: problem %u refers to matrix %u of %u", p,
                     spec->matrix[p], groups->num_matrices);
        const uint64_t G = groups->h_num_cols[spec->matrix[p]];
        RPVG_REQUIRE(G >= 1, "rpvg_hip_group_gibbs: problem %u has no columns", p);
        const uint64_t chains = spec->num_chains[p], its = spec->num_gibbs_its[p], all_its = its + spec->num_burn_its[p];
        RPVG_REQUIRE(chains * its < 0x40000000ull, "rpvg_hip_group_gibbs: problem %u draws %llu samples", p,
                     static_cast<unsigned long long>(chains * its));
        chain_off[p + 1] = chain_off[p] + chains;
        col_off[p + 1] = col_off[p] + G;
        const uint64_t sets_bound = std::min<uint64_t>(GS == 2 ? G * (G + 1) / 2 : G, chains * its);
        tab_off[p + 1] = tab_off[p] + nextPowerOfTwo(2 * sets_bound);
        out_capacity += sets_bound;
        const uint64_t draws = chains * all_its * GS;
        words_needed[p] = chains * GS + (G >= 2 ? 2 * draws : 0);
        dist_bound += static_cast<long double>(distributionDoubles(G)) * static_cast<long double>(GS == 2 ? G : 1);  // every column as the other member once
    }
    const uint64_t num_chains = chain_off[P], num_cols = col_off[P], num_slots = tab_off[P];
    RPVG_REQUIRE(num_chains < 0x7fffffffull, "rpvg_hip_group_gibbs: %llu chains exceed one launch", static_cast<unsigned long long>(num_chains));
    RPVG_REQUIRE(num_cols < 0xfffffff0ull, "rpvg_hip_group_gibbs: %llu columns", static_cast<unsigned long long>(num_cols));
    std::vector<uint64_t> stream_off(static_cast<size_t>(NG) + 1, 0);
    std::vector<uint8_t> seen(P, 0);
    for (uint32_t g = 0; g < NG; ++g) {
        RPVG_REQUIRE(spec->generator_problem_off[g] <= spec->generator_problem_off[g + 1] && spec->generator_problem_off[g + 1] <= P,
                     "rpvg_hip_group_gibbs: generator %u has inconsistent offsets", g);
        uint64_t needed = 0;
        for (uint32_t j = spec->generator_problem_off[g]; j < spec->generator_problem_off[g + 1]; ++j) {
            const uint32_t p = spec->generator_problem[j];
            RPVG_REQUIRE(p < P && !seen[p], "rpvg_hip_group_gibbs: generator %u lists problem %u (out of range or listed twice)", g, p);
            seen[p] = 1;
            needed += words_needed[p];
        }
        // room for the start draws' rejections (one more word each, probability columns / 2^32 per draw)
        const uint64_t blocks = std::max<uint64_t>(1, (needed + 64 + rpvg_streams::kMtWords - 1) / rpvg_streams::kMtWords);
        stream_off[g + 1] = stream_off[g] + blocks * rpvg_streams::kMtWords;
    }
    RPVG_REQUIRE(spec->generator_problem_off[0] == 0 && spec->generator_problem_off[NG] == P,
                 "rpvg_hip_group_gibbs: the generators list %u of %u problems", spec->generator_problem_off[NG], P);

    std::lock_guard<std::mutex> lock(ctx->mutex);
    RPVG_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    scope.reset(new HostScope("group_gibbs: uploads, allocations"));

    // storage of the distributions: by the bound when it fits the budget, else the budget (a call that runs out of it
    // reports RPVG_HIP_ERR_UNSUPPORTED: the caller has the host-driven sampler)
    uint64_t dist_capacity = 0;
    {
        const char * env = std::getenv("RPVG_HIP_GIBBS_BYTES");  // (read per call: a test switches it)
        size_t free_bytes = 0, total_bytes = 0;
        RPVG_HIP_CHECK(hipMemGetInfo(&free_bytes, &total_bytes));
        long double budget = env ? std::strtold(env, nullptr) : std::min<long double>(0.25L * free_bytes, 32.0L * (1ull << 30));  // (two host lanes ask at the same time)
        dist_capacity = static_cast<uint64_t>(std::min<long double>(dist_bound, budget / 8));
        dist_capacity = std::max<uint64_t>(dist_capacity, 1);
    }

    DeviceBuffer<uint32_t> d_matrix, d_chains, d_burn, d_its, d_gen_prob_off, d_gen_prob, d_gen_words;
    DeviceBuffer<uint64_t> d_chain_off, d_col_off, d_tab_off, d_stream_off;
    DeviceBuffer<double> d_log_freq;
    DeviceBuffer<uint32_t> d_remaining, d_tab_count, d_prob_count, d_prob_done;
    DeviceBuffer<RequestInfo> d_records;
    DeviceBuffer<GibbsHeader> d_hdr;
    DeviceBuffer<unsigned long long> d_words, d_set_off;
    UploadPack pack;
    pack.add(d_matrix, spec->matrix, P);
    pack.add(d_chains, spec->num_chains, P);
    pack.add(d_burn, spec->num_burn_its, P);
    pack.add(d_its, spec->num_gibbs_its, P);
    pack.add(d_gen_prob_off, spec->generator_problem_off, static_cast<size_t>(NG) + 1);
    pack.add(d_gen_prob, spec->generator_problem, P);
    pack.add(d_gen_words, spec->generator_words, static_cast<size_t>(NG) * rpvg_streams::kMtWords);
    pack.add(d_chain_off, chain_off.data(), chain_off.size());
    pack.add(d_col_off, col_off.data(), col_off.size());
    pack.add(d_tab_off, tab_off.data(), tab_off.size());
    pack.add(d_stream_off, stream_off.data(), stream_off.size());
    pack.add(d_log_freq, spec->log_freq, num_cols);
    pack.addZero(d_records, num_cols);
    pack.addZero(d_prob_count, P);
    pack.addZero(d_prob_done, P);
    pack.addZero(d_remaining, kMaxRounds);
    pack.addZero(d_tab_count, num_slots);
    pack.addZero(d_hdr, 1);
    pack.addZero(d_words, NG);
    pack.addZero(d_set_off, static_cast<size_t>(P) + 1);
    int span = ctx->spanBegin(FAM_H2D);
    RPVG_HIP_CHECK(pack.commit(st));
    ctx->spanEnd(span);
    ctx->stats.h2d_bytes += static_cast<double>(pack.copied_bytes);

    DeviceBuffer<uint32_t> d_stream, d_chain_problem, d_chain_cur, d_chain_iter, d_chain_flag, d_active_problem, d_req_other, d_new_req, d_tab_first;
    DeviceBuffer<unsigned long long> d_chain_pos, d_tab_key;
    DeviceBuffer<ActiveEntry> d_entries;
    DeviceBuffer<double> d_dist;
    DeviceBuffer<uint32_t> d_out;  // first | second | count | sequence, out_capacity each
    DeviceBuffer<uint32_t> d_final_state;
    RPVG_HIP_CHECK(d_final_state.alloc(static_cast<size_t>(NG) * rpvg_streams::kMtWords));
    RPVG_HIP_CHECK(d_stream.alloc(stream_off[NG] + kWordWindow));  // (the chains fetch their words four draws ahead)
    RPVG_HIP_CHECK(d_chain_problem.alloc(num_chains));
    RPVG_HIP_CHECK(d_chain_cur.alloc(2 * num_chains));
    RPVG_HIP_CHECK(d_chain_iter.alloc(num_chains));
    RPVG_HIP_CHECK(d_chain_flag.alloc(num_chains));
    RPVG_HIP_CHECK(d_chain_pos.alloc(num_chains));
    RPVG_HIP_CHECK(d_active_problem.alloc(P));
    RPVG_HIP_CHECK(d_entries.alloc(P));
    RPVG_HIP_CHECK(d_req_other.alloc(num_cols));
    RPVG_HIP_CHECK(d_new_req.alloc(num_cols));
    RPVG_HIP_CHECK(d_tab_key.alloc(num_slots));
    RPVG_HIP_CHECK(d_tab_first.alloc(num_slots));
    RPVG_HIP_CHECK(d_dist.alloc(dist_capacity));
    RPVG_HIP_CHECK(hipMemsetAsync(d_tab_key.ptr, 0xff, num_slots * sizeof(unsigned long long), st));
    RPVG_HIP_CHECK(hipMemsetAsync(d_tab_first.ptr, 0xff, num_slots * sizeof(uint32_t), st));

    const GibbsProblems pr{d_matrix.ptr, d_chains.ptr, d_burn.ptr, d_its.ptr, d_chain_off.ptr, d_col_off.ptr, d_tab_off.ptr, d_log_freq.ptr};
    const GibbsChains ch{d_chain_problem.ptr, d_chain_pos.ptr, d_chain_cur.ptr, d_chain_iter.ptr, d_chain_flag.ptr};
    const int sampler_span = ctx->spanBegin(FAM_GIBBS);
    struct SamplerSpan {  // (closed on every way out)
        rpvg_hip_ctx * ctx;
        int span;
        ~SamplerSpan() { ctx->spanEnd(span); }
    } sampler_span_guard{ctx, sampler_span};
    gibbsStreamKernel<<<dim3(NG), dim3(256), 0, st>>>(d_gen_words.ptr, d_gen_prob_off.ptr, d_gen_prob.ptr, d_stream_off.ptr, pr, groups->mat_cols.ptr, GS,
                                                      d_stream.ptr, ch, d_words.ptr, d_final_state.ptr, d_hdr.ptr);
    RPVG_HIP_CHECK(hipGetLastError());
    RPVG_HIP_CHECK(pinnedAlloc(&result->state_block, static_cast<size_t>(NG) * rpvg_streams::kMtWords * sizeof(uint32_t)));
    RPVG_HIP_CHECK(hipMemcpyAsync(result->state_block, d_final_state.ptr, static_cast<size_t>(NG) * rpvg_streams::kMtWords * sizeof(uint32_t),
                                  hipMemcpyDeviceToHost, st));
    RPVG_HIP_CHECK(groups->waitCollapse(st));

    struct Progress {
        uint32_t remaining;
        uint32_t pad[15];
        GibbsHeader hdr;
    };
    void * pinned = nullptr;
    RPVG_HIP_CHECK(pinnedAlloc(&pinned, sizeof(Progress)));
    std::shared_ptr<void> pinned_guard(pinned, [](void * ptr) { pinnedFree(ptr); });
    Progress * progress = static_cast<Progress *>(pinned);

    // share of a distribution from which a column's own conditional is asked for ahead of the chains (group size 2; above 1: never)
    // (off by default: at 0.02 and 0.002 the fourth round has 750 / 220 chains left instead of 1 330, the dozen chains of a
    // long-tailed posterior that make the last ten rounds are not helped, and the batch takes as long within the noise)
    static const double ask_ahead_env = RPVG_EXPERIMENT_ENV("RPVG_HIP_GIBBS_ASK_AHEAD") ? std::atof(RPVG_EXPERIMENT_ENV("RPVG_HIP_GIBBS_ASK_AHEAD")) : 0.0;
    const double ask_ahead = (GS == 2 && ask_ahead_env > 0) ? ask_ahead_env : 2.0;
    // rounds (from the first) whose conditionals go through the tile kernel (A/B)
    static const uint32_t tiled_rounds = RPVG_EXPERIMENT_ENV("RPVG_HIP_GIBBS_TILED_ROUNDS") ? static_cast<uint32_t>(std::atoi(RPVG_EXPERIMENT_ENV("RPVG_HIP_GIBBS_TILED_ROUNDS"))) : 1u;
    static const uint32_t follow_modes = RPVG_EXPERIMENT_ENV("RPVG_HIP_GIBBS_FOLLOW_MODES") ? static_cast<uint32_t>(std::atoi(RPVG_EXPERIMENT_ENV("RPVG_HIP_GIBBS_FOLLOW_MODES"))) : 2u;
    static const double follow_share = RPVG_EXPERIMENT_ENV("RPVG_HIP_GIBBS_FOLLOW_SHARE") ? std::atof(RPVG_EXPERIMENT_ENV("RPVG_HIP_GIBBS_FOLLOW_SHARE")) : 0.1;
    static const bool debug = RPVG_EXPERIMENT_ENV("RPVG_HIP_GIBBS_DEBUG") != nullptr;
    DeviceBuffer<unsigned long long> d_debug;
    if (debug) {
        RPVG_HIP_CHECK(d_debug.alloc(8ull * kMaxRounds));
        RPVG_HIP_CHECK(hipMemsetAsync(d_debug.ptr, 0, 8ull * kMaxRounds * sizeof(unsigned long long), st));
    }
    scope.reset(new HostScope("group_gibbs: rounds"));
    const uint32_t cus = static_cast<uint32_t>(ctx->props.multiProcessorCount);
    static const uint32_t chains_per_wave = RPVG_EXPERIMENT_ENV("RPVG_HIP_GIBBS_CHAINS_PER_WAVE") ? std::min(64, std::max(1, std::atoi(RPVG_EXPERIMENT_ENV("RPVG_HIP_GIBBS_CHAINS_PER_WAVE")))) : 8u;  // (64 / 16 / 8 chains per wave: 3.7 / 3.2 / 2.6 ms for the five long rounds of a configs[4] lane)
    const uint32_t advance_blocks = static_cast<uint32_t>((num_chains + 4 * chains_per_wave - 1) / (4 * chains_per_wave));
    const uint32_t work_blocks = cus * 8;
    uint32_t round = 0;
    bool finished = false;
    static const uint32_t first_rounds = RPVG_EXPERIMENT_ENV("RPVG_HIP_GIBBS_FIRST_ROUNDS") ? std::max(1, std::atoi(RPVG_EXPERIMENT_ENV("RPVG_HIP_GIBBS_FIRST_ROUNDS"))) : 6;
    uint32_t chunk = first_rounds;
    while (!finished) {
        if (round + chunk >= kMaxRounds) {  // (a round per column of a flat posterior over thousands of columns: the caller's sampler takes it)
            RPVG_HIP_CHECK(waitStream(st));
            setError("rpvg_hip_group_gibbs: the chains are not done after %u rounds", round);
            return RPVG_HIP_ERR_UNSUPPORTED;
        }
        for (uint32_t r = 0; r < chunk; ++r, ++round) {
#define RPVG_GIBBS_ROUND(W)                                                                                                            \
    gibbsAdvanceKernel<W><<<dim3(advance_blocks), dim3(256), 0, st>>>(                                                                 \
        static_cast<uint32_t>(num_chains), round, pr, groups->mat_cols.ptr, ch, d_stream.ptr, d_records.ptr, d_prob_count.ptr,         \
        d_prob_done.ptr, d_hdr.ptr, d_remaining.ptr, d_active_problem.ptr, d_req_other.ptr, d_dist.ptr, d_tab_key.ptr,                 \
        d_tab_count.ptr, d_tab_first.ptr, d_debug.ptr, chains_per_wave);                                                               \
    gibbsRequestOffsetsKernel<<<dim3(1), dim3(1024), 0, st>>>(pr, groups->mat_cols.ptr, groups->mat_rows.ptr, d_hdr.ptr,               \
                                                              d_active_problem.ptr, d_prob_count.ptr, d_prob_done.ptr, d_entries.ptr,  \
                                                              d_new_req.ptr, d_req_other.ptr, d_records.ptr, dist_capacity,            \
                                                              (tiled_rounds > round) ? 1u : 0u);                                       \
    span = ctx->spanBegin(FAM_LOGLIK);                                                                                                 \
    if (tiled_rounds > round) {                                                                                                        \
        gibbsConditionalTileKernel<W><<<dim3(work_blocks), dim3(256), 0, st>>>(                                                        \
            pr, d_hdr.ptr, d_entries.ptr, d_req_other.ptr, groups->mat_val_off.ptr, groups->mat_row_off.ptr, groups->mat_fast.ptr,     \
            groups->mat_mid.ptr, groups->mat_rows.ptr, groups->mat_cols.ptr, groups->values.ptr, groups->row_count.ptr,                \
            groups->row_noise.ptr, d_dist.ptr);                                                                                        \
    } else {                                                                                                                           \
        gibbsConditionalKernel<W><<<dim3(work_blocks), dim3(256), 0, st>>>(                                                            \
            pr, d_hdr.ptr, d_entries.ptr, d_req_other.ptr, groups->mat_val_off.ptr, groups->mat_row_off.ptr, groups->mat_fast.ptr,     \
            groups->mat_mid.ptr, groups->mat_rows.ptr, groups->mat_cols.ptr, groups->values.ptr, groups->row_count.ptr,                \
            groups->row_noise.ptr, d_dist.ptr);                                                                                        \
    }                                                                                                                                  \
    ctx->spanEnd(span)
            if (GS == 1) {
                RPVG_GIBBS_ROUND(1);
            } else {
                RPVG_GIBBS_ROUND(2);
            }
#undef RPVG_GIBBS_ROUND
            const bool follows = round == 0 && GS == 2 && follow_modes > 0;
            gibbsDistributionKernel<<<dim3(work_blocks), dim3(256), 0, st>>>(pr, d_hdr.ptr, d_new_req.ptr, groups->mat_cols.ptr, d_records.ptr, d_dist.ptr,
                                                                            follows ? follow_share : ask_ahead, d_prob_count.ptr, d_prob_done.ptr,
                                                                            d_active_problem.ptr, d_req_other.ptr, d_hdr.ptr);
            // Behind the first round the conditionals of the columns the chains are about to draw — the columns that hold a
            // tenth or more of a distribution just evaluated — are evaluated twice over before the chains move again: a chain
            // from a random start needs the conditional of its first draw, then that of its second, and each was a round of all
            // 60 000 chains of a configs[4] lane (rounds 1-3: 1.3, 6.0 and 12.6 M draws; now 19.6 M draws in the first round
            // behind the starts and 2 500 chains left after it).  The same conditionals as before, to 0.002 % of the
            // evaluations: they are the ones the chains ask for anyway.  (RPVG_HIP_GIBBS_FOLLOW_MODES=n, _FOLLOW_SHARE=x)
            for (uint32_t f = 0; follows && f < follow_modes; ++f) {
                gibbsRequestOffsetsKernel<<<dim3(1), dim3(1024), 0, st>>>(pr, groups->mat_cols.ptr, groups->mat_rows.ptr, d_hdr.ptr, d_active_problem.ptr,
                                                                          d_prob_count.ptr, d_prob_done.ptr, d_entries.ptr, d_new_req.ptr, d_req_other.ptr,
                                                                          d_records.ptr, dist_capacity, 0u);
                span = ctx->spanBegin(FAM_LOGLIK);
                gibbsConditionalKernel<2><<<dim3(work_blocks), dim3(256), 0, st>>>(
                    pr, d_hdr.ptr, d_entries.ptr, d_req_other.ptr, groups->mat_val_off.ptr, groups->mat_row_off.ptr, groups->mat_fast.ptr,
                    groups->mat_mid.ptr, groups->mat_rows.ptr, groups->mat_cols.ptr, groups->values.ptr, groups->row_count.ptr,
                    groups->row_noise.ptr, d_dist.ptr);
                ctx->spanEnd(span);
                gibbsDistributionKernel<<<dim3(work_blocks), dim3(256), 0, st>>>(pr, d_hdr.ptr, d_new_req.ptr, groups->mat_cols.ptr, d_records.ptr, d_dist.ptr,
                                                                                (f + 1 < follow_modes) ? follow_share : ask_ahead, d_prob_count.ptr,
                                                                                d_prob_done.ptr, d_active_problem.ptr, d_req_other.ptr, d_hdr.ptr);
            }
        }
        RPVG_HIP_CHECK(hipGetLastError());
        RPVG_HIP_CHECK(hipMemcpyAsync(&progress->remaining, d_remaining.ptr + (round - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        RPVG_HIP_CHECK(hipMemcpyAsync(&progress->hdr, d_hdr.ptr, sizeof(GibbsHeader), hipMemcpyDeviceToHost, st));
        RPVG_HIP_CHECK(waitStream(st));
        if (progress->hdr.error) {
            const uint32_t err = progress->hdr.error;
            if (err & kErrDistributions) {
                setError("rpvg_hip_group_gibbs: the conditional distributions need more than the %llu bytes reserved for them (RPVG_HIP_GIBBS_BYTES)",
                         static_cast<unsigned long long>(dist_capacity * 8));
                return RPVG_HIP_ERR_UNSUPPORTED;
            }
            setError("rpvg_hip_group_gibbs: a generator's start draws ran past the words generated for it");
            return RPVG_HIP_ERR_RUNTIME;
        }
        finished = progress->remaining == 0;
        chunk = 4;
    }
    if (debug) {
        std::vector<unsigned long long> counts(8ull * round);
        RPVG_HIP_CHECK(hipMemcpy(counts.data(), d_debug.ptr, counts.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        for (uint32_t r = 0; r < round; ++r) {
            const unsigned long long * c = counts.data() + 8ull * r;
            std::fprintf(stderr, "[rpvg_hip gibbs] round %2u: %7llu chains ran, %9llu draws (most of one chain %6llu), %8llu record lookups, %8llu draws off the mode (%llu walked on), %8llu key changes\n",
                         r, c[0], c[1], c[2], c[3], c[4], c[5], c[6]);
        }
    }
    ctx->stats.loglik_launches += round;
    ctx->stats.loglik_evals += progress->hdr.evals;
    result->rounds = round;
    result->conditionals = progress->hdr.total_requests;

    // the sets
    scope.reset(new HostScope("group_gibbs: sets"));
    RPVG_HIP_CHECK(d_out.alloc(4 * out_capacity));
    uint32_t * out_first = d_out.ptr, * out_second = d_out.ptr + out_capacity, * out_count = d_out.ptr + 2 * out_capacity,
             * out_seq = d_out.ptr + 3 * out_capacity;
    gibbsCountSetsKernel<<<dim3(P), dim3(256), 0, st>>>(pr, d_tab_key.ptr, d_set_off.ptr);
    gibbsSetOffsetsKernel<<<dim3(1), dim3(1024), 0, st>>>(P, d_set_off.ptr, d_hdr.ptr);
    gibbsCollectKernel<<<dim3(P), dim3(256), 0, st>>>(pr, d_tab_key.ptr, d_tab_count.ptr, d_tab_first.ptr, d_set_off.ptr, out_first, out_second,
                                                      out_count, out_seq, d_hdr.ptr);
    RPVG_HIP_CHECK(hipGetLastError());
    static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "offsets are copied as they are");
    RPVG_HIP_CHECK(hipMemcpyAsync(result->set_off.data(), d_set_off.ptr, (static_cast<size_t>(P) + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    RPVG_HIP_CHECK(hipMemcpyAsync(result->words_consumed.data(), d_words.ptr, static_cast<size_t>(NG) * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    RPVG_HIP_CHECK(hipMemcpyAsync(&progress->hdr, d_hdr.ptr, sizeof(GibbsHeader), hipMemcpyDeviceToHost, st));
    RPVG_HIP_CHECK(waitStream(st));
    const uint64_t total_sets = progress->hdr.total_sets;
    RPVG_REQUIRE(total_sets <= out_capacity, "rpvg_hip_group_gibbs: %llu sets in room for %llu", static_cast<unsigned long long>(total_sets),
                 static_cast<unsigned long long>(out_capacity));
    if (total_sets > 0) {
        RPVG_HIP_CHECK(pinnedAlloc(&result->block, 4 * total_sets * sizeof(uint32_t)));
        uint32_t * host = static_cast<uint32_t *>(result->block);
        RPVG_HIP_CHECK(hipMemcpyAsync(host, out_first, total_sets * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        RPVG_HIP_CHECK(hipMemcpyAsync(host + total_sets, out_second, total_sets * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        RPVG_HIP_CHECK(hipMemcpyAsync(host + 2 * total_sets, out_count, total_sets * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        if (progress->hdr.unsorted) {
            RPVG_HIP_CHECK(hipMemcpyAsync(host + 3 * total_sets, out_seq, total_sets * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        }
        RPVG_HIP_CHECK(waitStream(st));
        if (progress->hdr.unsorted) {  // the few problems with more sets than the collect kernel ranks in LDS
            uint32_t * seq = host + 3 * total_sets;
            std::vector<uint32_t> order, scratch;
            for (uint32_t p = 0; p < P; ++p) {
                const uint64_t begin = result->set_off[p], n = result->set_off[p + 1] - begin;
                if (n <= kRankInLds) continue;
                order.resize(n);
                std::iota(order.begin(), order.end(), 0u);
                std::sort(order.begin(), order.end(), [&](const uint32_t a, const uint32_t b) { return seq[begin + a] < seq[begin + b]; });
                scratch.resize(n);
                for (uint32_t * column : {host, host + total_sets, host + 2 * total_sets}) {
                    for (uint64_t e = 0; e < n; ++e) scratch[e] = column[begin + order[e]];
                    std::copy(scratch.begin(), scratch.end(), column + begin);
                }
            }
        }
        result->first = host;
        result->second = host + total_sets;
        result->count = host + 2 * total_sets;
    }
    const int build_status = groups->buildError(st);  // the matrices were built without a host sync
    if (build_status != RPVG_HIP_OK) return build_status;
    *result_out = result.release();
    return RPVG_HIP_OK;
}

extern "C" int rpvg_hip_gibbs_sets_get(const rpvg_hip_gibbs_sets * result, rpvg_hip_gibbs_sets_view * view_out) {
    RPVG_REQUIRE(result && view_out, "rpvg_hip_gibbs_sets_get: NULL argument");
    view_out->num_problems = result->num_problems;
    view_out->group_size = result->group_size;
    view_out->set_off = result->set_off.data();
    view_out->first = result->first;
    view_out->second = result->second;
    view_out->count = result->count;
    view_out->words_consumed = result->words_consumed.data();
    view_out->generator_state = static_cast<const uint32_t *>(result->state_block);
    view_out->rounds = result->rounds;
    view_out->conditionals = result->conditionals;
    return RPVG_HIP_OK;
}

extern "C" void rpvg_hip_gibbs_sets_free(rpvg_hip_gibbs_sets * result) { delete result; }
