/* rpvg_hip.h — C ABI of the MI355X (gfx950) engine behind rpvg's inference
 * hot path.  extern "C", plain pointers and sizes only; implemented by
 * rpvg_amd/csrc/librpvg_hip.so (hand-written HIP kernels).
 *
 * The reference (C++17, no FFI today) does all of this arithmetic inside the
 * PathEstimator class hierarchy with Eigen; each entry point below names the
 * reference code whose arithmetic it takes over (paths relative to the rpvg
 * checkout).  The host-side C++ classes in rpvg_amd/host keep the reference's
 * PathEstimator::estimate() interface and call only these functions;
 * INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success, a negative rpvg_hip_status
 *     otherwise; never throws, never aborts; rpvg_hip_last_error() describes
 *     the last failure on the calling thread.
 *   - "host" pointers are ordinary host memory owned by the caller; "device"
 *     pointers are memory of the context's GPU (rpvg_hip_malloc).
 *   - a context is bound to one GPU and owns one HIP stream; calls on one
 *     context are serialised internally, different contexts are independent.
 *   - there is NO CPU fallback: without a usable GPU rpvg_hip_create fails.
 */
#ifndef RPVG_HIP_H
#define RPVG_HIP_H

#include <stdint.h>

#include "rpvg_batch.h"
#include "rpvg_rows.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rpvg_hip_status {
    RPVG_HIP_OK = 0,
    RPVG_HIP_ERR_NO_DEVICE = -1,  /* no gfx950-compatible GPU / HIP runtime unusable */
    RPVG_HIP_ERR_RUNTIME = -2,    /* a HIP call failed (see last_error)              */
    RPVG_HIP_ERR_INVALID = -3,    /* argument validation failed                      */
    RPVG_HIP_ERR_ALLOC = -4,      /* host or device allocation failed                */
    RPVG_HIP_ERR_UNSUPPORTED = -5 /* the call does not take this input; the caller uses the calls it stands for (nothing was changed) */
} rpvg_hip_status;

typedef struct rpvg_hip_ctx rpvg_hip_ctx;
typedef struct rpvg_hip_batch rpvg_hip_batch;   /* device-resident cluster batch          */
typedef struct rpvg_hip_groups rpvg_hip_groups; /* device-resident group (column-set) matrices */

/* ---- context ------------------------------------------------------------ */
int rpvg_hip_device_count(int * count);
int rpvg_hip_create(int device, rpvg_hip_ctx ** ctx_out);
/* A context for rpvg_hip_batch_upload next to contexts that estimate on the same GPU (the rows of batch n + 1 copied
 * under the kernels of batch n; the reference has no analogue: its rows never leave the host, SURVEY.md section 8d puts
 * the copy inside the metric).  Its stream has the device's highest priority, which gives it a hardware queue of its
 * own: the copies and expansion kernels do not queue behind the estimating contexts' kernels (measured: 10.6 -> 7.3 ms
 * per 280 MB batch while a batch is estimated, 14.7 -> 13.7 ms per batch in steady state). */
int rpvg_hip_create_uploader(int device, rpvg_hip_ctx ** ctx_out);
/* A context with fewer side streams than rpvg_hip_create's six (1..6), for engines that run whole batches next to each other
 * on one GPU (rpvg_amd/host/batch_pipeline.hpp): the runtime maps all streams of a process onto its hardware queues
 * (GPU_MAX_HW_QUEUES, 8 here), commands of streams that share a queue run in order, and four engines of nine streams
 * each put one engine's short kernels behind another's long ones.  The launches that would have had streams of their own
 * follow one another on the streams there are; results are the same. */
int rpvg_hip_create_with_streams(int device, int side_streams, rpvg_hip_ctx ** ctx_out);
void rpvg_hip_destroy(rpvg_hip_ctx * ctx);
const char * rpvg_hip_last_error(void);
int rpvg_hip_synchronize(rpvg_hip_ctx * ctx);
/* name of the GPU, number of CUs, bytes of device memory */
int rpvg_hip_device_info(rpvg_hip_ctx * ctx, char * name, uint32_t name_cap, uint32_t * num_cus, uint64_t * mem_bytes);

/* device memory helpers (for callers that keep matrices resident) */
int rpvg_hip_malloc(rpvg_hip_ctx * ctx, uint64_t bytes, void ** device_ptr_out);
int rpvg_hip_free(rpvg_hip_ctx * ctx, void * device_ptr);
int rpvg_hip_memcpy_h2d(rpvg_hip_ctx * ctx, void * device_dst, const void * host_src, uint64_t bytes);
int rpvg_hip_memcpy_d2h(rpvg_hip_ctx * ctx, void * host_dst, const void * device_src, uint64_t bytes);

/* Page-locks a range of the caller's host memory (hipHostRegister): uploads from inside a registered range go to the
 * GPU straight from it; anything else is staged through pinned blocks of the library's own (one more host copy). */
int rpvg_hip_host_register(void * host, uint64_t bytes);
int rpvg_hip_host_unregister(void * host);

/* ---- cluster batch ------------------------------------------------------ */
/* Copies the sparse rows of K clusters to the GPU once; every later call
 * refers to clusters by their index in this batch.  Replaces the per-call
 * walk over `vector<ReadPathProbabilities>` in constructProbabilityMatrix /
 * constructPartialProbabilityMatrix / constructGroupedProbabilityMatrix
 * (src/path_estimator.cpp:55-154): the (probability, path list) groups are
 * expanded to (path, probability) entries on the GPU. */
int rpvg_hip_batch_upload(rpvg_hip_ctx * ctx, const rpvg_cluster_batch * host_batch, rpvg_hip_batch ** batch_out);
void rpvg_hip_batch_free(rpvg_hip_ctx * ctx, rpvg_hip_batch * batch);
/* The same upload in two halves, for a caller that keeps a copy engine busy (rpvg_amd/host/batch_pipeline.hpp): _begin checks
 * the offsets, queues the copies on `ctx` — an uploader's context — and returns when they are done; _finish runs the kernels
 * behind them (expansion, validation, read totals, haplotype columns) on `ctx` — any context of the same GPU, e.g. the one
 * that estimates the batch next — and returns the upload's verdict.  Between the two the batch takes no other call;
 * host_batch is the same caller-owned batch both times (a batch that fails _finish is freed by it). */
int rpvg_hip_batch_upload_begin(rpvg_hip_ctx * ctx, const rpvg_cluster_batch * host_batch, rpvg_hip_batch ** batch_out);
int rpvg_hip_batch_upload_finish(rpvg_hip_ctx * ctx, rpvg_hip_batch * batch, const rpvg_cluster_batch * host_batch);
/* ... or the second half itself in two steps: _finish_queue puts the kernels behind the copies on a stream of `ctx` (an uploader's
 * context: its side stream, next to the copies of the batch after) and returns; _finish_wait, from any thread and without a
 * context, waits for them and does the host's part — what rpvg_hip_batch_upload_finish does in one call.  A pipeline's uploader
 * queues, the estimator that takes the batch waits (rpvg_amd/host/batch_pipeline.hpp).  On failure the batch is freed. */
int rpvg_hip_batch_upload_finish_queue(rpvg_hip_ctx * ctx, rpvg_hip_batch * batch, const rpvg_cluster_batch * host_batch);
int rpvg_hip_batch_upload_finish_wait(rpvg_hip_batch * batch, const rpvg_cluster_batch * host_batch);
/* A batch from the segments its callers flattened (include/rpvg_batch.h, rpvg_cluster_segment): what the reference's per-cluster
 * loop (src/main.cpp:829,976-977) hands to estimate(), one segment per call in flight, joined on the device.  One kernel reads
 * the segments from page-locked host memory, expands the (probability, path list) groups, validates rows and offsets and writes
 * every device array of the batch; the haplotype columns follow as in rpvg_hip_batch_upload.  Returns when the batch is
 * complete; the segments stay untouched until then.  Same results, entry for entry, as rpvg_hip_batch_upload of the joined
 * batch.  Page-locked blocks for the segments: rpvg_hip_pinned_alloc (cached by size class; a block's capacity is at least the
 * bytes asked for) / rpvg_hip_pinned_free. */
int rpvg_hip_batch_upload_segments(rpvg_hip_ctx * ctx, const rpvg_cluster_segment * segments, uint32_t num_segments, rpvg_hip_batch ** batch_out);
int rpvg_hip_pinned_alloc(uint64_t bytes, void ** host_out);
/* How the calling thread waits for the GPU inside the calls of this library from now on: it queries for `microseconds` and then
 * naps between queries (30 us each).  Default 20: a thread that waits for a millisecond-long batch should not hold a core.  A
 * thread whose callers are blocked behind it for a few hundred microseconds (the leaders of PathEstimator::estimate()'s call
 * combiner) asks for more: a nap costs 50-80 us of latency per wait. */
void rpvg_hip_thread_wait_spin_us(uint32_t microseconds);
void rpvg_hip_pinned_free(void * host);
/* Read count of every cluster of an uploaded batch (the sum of its rows' read counts, exact), added up on the device behind
 * the copy (src/path_abundance_estimator.cpp:44,291,690: `read_counts.sum()`). */
int rpvg_hip_batch_cluster_totals(const rpvg_hip_batch * batch, double * totals_out, uint32_t num_clusters);
/* The path side of a batch.  When host_batch carries path_group_id, path_source_off and source_id, rpvg_hip_batch_upload
 * also copies those and runs NestedPathAbundanceEstimator::findPathSourceGroups (src/path_abundance_estimator.cpp:493-546)
 * for every cluster on the device: haplotypes (source ids) that carry the identical list of paths form one column, its
 * multiplicity is their number, columns in ascending order of their smallest haplotype id.  1 when the batch holds such
 * columns (0: no ids were given, or their ranges are too wide for the device scratch — ids are expected to be small
 * consecutive integers — and the caller groups on the host). */
int rpvg_hip_batch_has_source_columns(const rpvg_hip_batch * batch);
/* Inspection (tests): the columns of one cluster — sizes first, then multiplicities [columns], the end of every column's list
 * within column_paths [columns], and the lists back to back (ascending cluster-local paths) [column paths]. */
int rpvg_hip_batch_source_columns_sizes(const rpvg_hip_batch * batch, uint32_t cluster, uint32_t * num_columns_out,
                                        uint32_t * num_column_paths_out);
int rpvg_hip_batch_source_columns_get(rpvg_hip_ctx * ctx, const rpvg_hip_batch * batch, uint32_t cluster,
                                      uint32_t * column_counts_out, uint32_t * column_path_end_out, uint32_t * column_paths_out);

/* ---- EM abundance solves ------------------------------------------------ */
/* One EM problem = one cluster restricted to a strictly ascending list of its
 * paths (all of them for `-i transcripts`, a diplotype's path subset for
 * `-i haplotype-transcripts`), plus the noise component. */
typedef struct rpvg_hip_em_problems {
    uint32_t num_problems;
    const uint32_t * cluster;  /* host [P]   index into the uploaded batch          */
    const uint64_t * col_off;  /* host [P+1] range into col_path                    */
    const uint32_t * col_path; /* host       cluster-local path of each column      */
    /* > 0: readCollapseProbabilityMatrix (src/path_estimator.cpp:219-259) on the normalised rows of every problem, with this
     * prob_precision, as MinimumPathAbundanceEstimator and NestedPathAbundanceEstimator do before their EM
     * (src/path_abundance_estimator.cpp:266,668; PathAbundanceEstimator::estimate, :18-45, does not: 0). */
    double collapse_precision;
} rpvg_hip_em_problems;

typedef struct rpvg_hip_em_results {
    double * abundances;   /* host [col_off[P]]  expected read count per column, laid out like col_path */
    double * noise_count;  /* host [P] */
    double * total_count;  /* host [P] */
    uint32_t * iterations; /* host [P] EM iterations executed */
} rpvg_hip_em_results;

/* For every problem: build the row-normalised probability matrix of the
 * column subset with the noise column appended, and run the EM fixed point
 * to the reference's stop rule; everything on the GPU.  Takes over
 *   constructPartialProbabilityMatrix       src/path_estimator.cpp:79-113
 *   addNoiseAndNormalizeProbabilityMatrix   src/path_estimator.cpp:156-166
 *   EMAbundanceEstimator                    src/path_abundance_estimator.cpp:47-114
 * (start value 1/float(C), convergence over components >= 1e-8 for 10
 * consecutive iterations, sub-1e-8 abundances moved to noise_count) and, with
 * problems->collapse_precision > 0,
 *   readCollapseProbabilityMatrix           src/path_estimator.cpp:219-259
 * between the normalisation and the EM: the rows of every problem are put through the reference's tolerant
 * sort and compare-with-run-head merge (the machinery of the group matrices' collapse on the sparse rows); a
 * merged row's read count moves to its run head, and the EM kernels read the merged counts of the problems in
 * which rows were merged.  Rows whose columns all fall into the same multiple of 2^-44 stand for each other
 * there (and merges of rows equal to 1e-13 relative are not replayed: they move nothing; docs/design/kernels-collapse.md).
 * Rows without any selected path are folded into one exact scalar (docs/design/kernels-em.md). */
int rpvg_hip_em_solve(rpvg_hip_ctx * ctx, const rpvg_hip_batch * batch, uint32_t max_em_its, double max_rel_em_conv,
                      const rpvg_hip_em_problems * problems, rpvg_hip_em_results * results);

/* gibbsReadCountSampler (src/path_abundance_estimator.cpp:116-212) for a batch of EM problems, on the GPU.
 * Problem p starts from its EM estimate (init_abundances laid out like col_path, in expected read counts,
 * plus init_noise_count[p]) and records num_samples[p] states, one every gibbs_thin_its iterations.
 * Random numbers come from the counter-based Philox4x32-10 generator keyed by seeds[p]: the reference's
 * mt19937 / libstdc++ distribution streams cannot be reproduced on a GPU, parity is statistical.
 * noise_samples: [sum num_samples]; abundance_samples: per problem num_samples[p] x columns, sample-major
 * (the layout of CountSamples::abundance_samples, src/path_cluster_estimates.hpp:35-43). */
int rpvg_hip_gibbs_read_counts(rpvg_hip_ctx * ctx, const rpvg_hip_batch * batch, const rpvg_hip_em_problems * problems,
                               const double * init_abundances, const double * init_noise_count,
                               const uint32_t * num_samples, const uint64_t * seeds, uint32_t gibbs_thin_its,
                               double gamma, double * noise_samples, double * abundance_samples);

/* Same EM on a dense matrix that is already resident on the GPU: row-major,
 * `ld` doubles between rows (ld even), columns [0, C-1) = paths and column
 * C-1 = noise, already normalised; counts = read count per row.  Used for
 * clusters large enough to stream from HBM (one 1M x 2k cluster = 16 GB).
 * abundances (host) receives C-1 values. */
int rpvg_hip_em_dense(rpvg_hip_ctx * ctx, const double * device_matrix, uint64_t num_rows, uint32_t num_cols,
                      uint64_t ld, const double * device_counts, double total_count, uint32_t max_em_its,
                      double max_rel_em_conv, double * abundances, double * noise_count, uint32_t * iterations);

/* Row-sharded form of rpvg_hip_em_dense for ONE cluster spread over the ranks of the context's
 * communicator (rpvg_hip_comm_init): this rank holds num_rows of the cluster's rows; total_count is
 * the read count of the WHOLE cluster.  Every EM iteration all-reduces the C partial column sums
 * t_j = sum_i (c_i / s_i) P_ij over the ranks (RCCL, queued on the context's stream between the
 * streaming pass and the update), then every rank applies the identical update and convergence test
 * (src/path_abundance_estimator.cpp:58-97), so all ranks stop at the same iteration and return the
 * same abundances.  With a one-rank communicator it equals rpvg_hip_em_dense. */
int rpvg_hip_em_dense_sharded(rpvg_hip_ctx * ctx, const double * device_matrix, uint64_t num_rows, uint32_t num_cols,
                              uint64_t ld, const double * device_counts, double total_count, uint32_t max_em_its,
                              double max_rel_em_conv, double * abundances, double * noise_count, uint32_t * iterations);

/* Builds the dense normalised matrix (layout above) of one cluster of an
 * uploaded batch, all paths + noise (constructProbabilityMatrix +
 * addNoiseAndNormalizeProbabilityMatrix, src/path_estimator.cpp:55-77,156-166).
 * device_matrix needs R*ld doubles, device_counts R doubles. */
int rpvg_hip_dense_from_cluster(rpvg_hip_ctx * ctx, const rpvg_hip_batch * batch, uint32_t cluster,
                                double * device_matrix, uint64_t ld, double * device_counts, double * total_count);

/* ---- group (haplotype / diplotype) log-likelihoods ---------------------- */
/* A group matrix has one column per path group: M[i][g] = sum of the row's
 * probabilities over the paths of group g (constructGroupedProbabilityMatrix,
 * src/path_estimator.cpp:115-154; a group of one path gives
 * constructProbabilityMatrix).  normalise != 0 applies
 * addNoiseAndNormalizeProbabilityMatrix (as NestedPathAbundanceEstimator does,
 * src/path_abundance_estimator.cpp:440-446); 0 keeps raw probabilities (as
 * PathGroupPosteriorEstimator does, src/path_posterior_estimator.cpp:45). */
typedef struct rpvg_hip_group_spec {
    uint32_t num_matrices;
    const uint32_t * cluster;         /* host [M]   */
    const uint64_t * group_off;       /* host [M+1] columns of each matrix, range into group_path_off */
    const uint64_t * group_path_off;  /* host [G+1] */
    const uint32_t * group_path;      /* host       cluster-local paths of each column (a set: list a path once) */
    int32_t normalise;
    /* > 0 (normalised matrices only): replay readCollapseProbabilityMatrix (src/path_estimator.cpp:197-259, called
     * on every normalised group matrix, src/path_abundance_estimator.cpp:380,443) with this prob_precision — every
     * row takes the values of the head of its run in the reference's tolerant row order; 0: rows stay as built.
     * With the collapse a call takes at most 2^20 - 1 matrices and 2^31 - 1 rows (RPVG_HIP_ERR_INVALID beyond: the
     * matrices of a larger batch are built in several calls). */
    double collapse_precision;
} rpvg_hip_group_spec;

/* Returns once the build is queued on the context's stream (the spec arrays have been consumed by then); a group that
 * refers to a path outside its cluster (or, with RPVG_HIP_BUILD_MASKS=1, lists a path twice: the columns of
 * constructGroupedProbabilityMatrix are sets of paths, src/path_abundance_estimator.cpp:493-546) is reported by the first call that uses the matrices
 * (RPVG_HIP_ERR_INVALID from rpvg_hip_group_loglik / _conditionals / rpvg_hip_bounded_pair_posteriors). */
int rpvg_hip_groups_build(rpvg_hip_ctx * ctx, const rpvg_hip_batch * batch, const rpvg_hip_group_spec * spec,
                          rpvg_hip_groups ** groups_out);
/* The matrices of NestedPathAbundanceEstimator::inferAbundancesCollapsedGroups (src/path_abundance_estimator.cpp:428-446) for
 * the listed clusters, their columns being the batch's own haplotype columns: findPathSourceGroups
 * (src/path_abundance_estimator.cpp:493-546) ran on the device when the batch was uploaded with PathInfo::source_ids
 * (rpvg_hip_batch_upload; rpvg_amd/csrc/path_sources.hip), so no column list crosses the ABI.  Column c of a matrix is the
 * c-th distinct path list among its cluster's haplotypes in ascending order of the smallest haplotype id that carries it; the
 * columns' multiplicities (path_counts) stay on the device with the matrices: rpvg_hip_bounded_pair_posteriors and
 * rpvg_hip_nested_subset_em take column_counts = NULL for them.  Returns RPVG_HIP_ERR_UNSUPPORTED, having changed nothing,
 * for a batch without device-resident columns (the caller groups on the host and calls rpvg_hip_groups_build). */
int rpvg_hip_groups_build_from_sources(rpvg_hip_ctx * ctx, const rpvg_hip_batch * batch, uint32_t num_matrices,
                                       const uint32_t * clusters, int32_t normalise, double collapse_precision,
                                       rpvg_hip_groups ** groups_out);
/* The matrices whose column c is path c of the cluster alone — the raw path posteriors of PathPosteriorEstimator /
 * PathGroupPosteriorEstimator (src/path_posterior_estimator.cpp:9-31, 35-71: one group per path) — for the listed clusters: the
 * same matrices as rpvg_hip_groups_build with one single-entry list per path, without the lists crossing the ABI (configs[4]:
 * 500 000 of them per batch, flattened, copied and uploaded every call). */
int rpvg_hip_groups_build_single_paths(rpvg_hip_ctx * ctx, const rpvg_hip_batch * batch, uint32_t num_matrices,
                                       const uint32_t * clusters, int32_t normalise, double collapse_precision,
                                       rpvg_hip_groups ** groups_out);
void rpvg_hip_groups_free(rpvg_hip_ctx * ctx, rpvg_hip_groups * groups);
/* What the row collapse of these matrices did (waits for their build; any output may be NULL): matrices that held
 * rows within collapse_precision of each other but not equal up to rounding, whose runs were therefore replayed as
 * the reference forms them; rows that took the values of their run head; matrices that were sorted as a whole
 * (otherwise only the rows close to such a pair are); rows that took part in a replay. */
int rpvg_hip_groups_collapse_info(rpvg_hip_ctx * ctx, const rpvg_hip_groups * groups, uint32_t * matrices_replayed,
                                  uint32_t * rows_replaced, uint32_t * matrices_sorted_whole, uint32_t * active_rows);

/* Evaluates, for every request q,
 *   out[q] = sum_i count_i * log( noise_i + ( sum_{m < width, members[q*width+m] != UINT32_MAX}
 *                                             M[i][members[q*width+m]]  +  (add_rowmax[q] ? max_g M[i][g] : 0) ) / divisor )
 * on matrix[q].  This is the `read_counts * (...).array().log().matrix()`
 * contraction of calculatePathGroupPosteriorsFull (src/path_estimator.cpp:354-361),
 * calculatePathGroupPosteriorsBounded (:424-427 with add_rowmax, :439) and the
 * conditional of estimatePathGroupPosteriorsGibbs (:527-545); adding log
 * frequencies, permutation counts, pruning and log-sum-exp stay on the host. */
int rpvg_hip_group_loglik(rpvg_hip_ctx * ctx, const rpvg_hip_groups * groups, uint32_t num_requests,
                          const uint32_t * matrix, const uint32_t * members, uint32_t width, double divisor,
                          const uint8_t * add_rowmax, double * out);

/* A whole conditional of estimatePathGroupPosteriorsGibbs (src/path_estimator.cpp:527-545) per request: request q
 * fixes the other width-1 members of a group (others[q*(width-1) ..], slot order) on matrix[q] and receives, for
 * EVERY column k of that matrix,
 *   sum_i count_i * log( noise_i + ( sum_o M[i][others] + M[i][k] ) / divisor ).
 * out holds the requests back to back, one value per column of the request's matrix.  Same arithmetic as
 * rpvg_hip_group_loglik with members = (others..., k); the request list is O(1) per conditional instead of O(columns). */
int rpvg_hip_group_conditionals(rpvg_hip_ctx * ctx, const rpvg_hip_groups * groups, uint32_t num_requests,
                                const uint32_t * matrix, const uint32_t * others, uint32_t width, double divisor,
                                double * out);

/* ---- the Gibbs sampler of the group posteriors, on the device ---------------------- */
/* estimatePathGroupPosteriorsGibbs (src/path_estimator.cpp:475-589) for group sizes 1 and 2, draw for draw up to the
 * floating-point reduction order of a distribution's sums (the conditional's log-sum-exp, the weight sum and the partial sums
 * are added wave-parallel here, one after the other in std::discrete_distribution: a last-ulp difference in a partial sum can
 * move a draw that lands exactly on a boundary — never seen in 1 300 swept configurations, not excluded): the chains
 * of every problem (:505), their starts (uniform_int_distribution, :491,509), the conditional of a slot given the other
 * member (:527-555: the contraction of rpvg_hip_group_conditionals, + log frequency, log-sum-exp, exp, then
 * std::discrete_distribution's own normalisation and partial sums), one draw per slot and iteration (:556) and the
 * counting of the sorted sets after the burn-in (:559-574).  The generators are the caller's std::mt19937s: the caller
 * passes the NEXT 624 outputs of each (drawn from a copy) — their untempered values are the generator's state — and
 * moves the original past the words_consumed[g] words the chains took (discard(), or the state words of the view), which
 * leaves it where the reference's sampler would.
 * Problems that share a generator (the transcripts of one cluster, src/path_abundance_estimator.cpp:371-412) are listed
 * under it in the order the reference runs them.  libstdc++'s streams (GCC 11: Lemire's rejection for the starts, two
 * words per generate_canonical<double, 53>, no draw at all from a distribution of fewer than two weights) are restated in
 * rpvg_amd/csrc/gibbs_streams.hpp and checked against libstdc++ on the CPU (tests/cpp/gibbs_streams_check.cpp).
 * The chain counts and lengths are the caller's (src/path_estimator.cpp:501-503).
 * The sets of a problem come in the order the reference appends them to path_group_sets (first appearance).
 * Returns RPVG_HIP_ERR_UNSUPPORTED, having changed nothing, for other group sizes, when the distributions outgrow the
 * memory reserved for them (RPVG_HIP_GIBBS_BYTES; default: room for every column of every problem as the other member,
 * at most a quarter of the free device memory and 32 GiB) and when the chains are not done after 8 192 rounds of requests:
 * the caller then drives the sampler itself through rpvg_hip_group_conditionals. */
typedef struct rpvg_hip_gibbs_spec {
    uint32_t num_problems;
    uint32_t group_size;                    /* 1 or 2                                                                  */
    const uint32_t * matrix;                /* [P]   matrix of `groups` the problem samples on                         */
    const uint32_t * num_chains;            /* [P]   :501                                                              */
    const uint32_t * num_burn_its;          /* [P]   :502                                                              */
    const uint32_t * num_gibbs_its;         /* [P]   :503                                                              */
    const double * log_freq;                /*       calcPathLogFrequences of every column, problems back to back      */
    uint32_t num_generators;
    const uint32_t * generator_problem_off; /* [NG+1]                                                                  */
    const uint32_t * generator_problem;     /* [P]   the problems of each generator, in the order it serves them       */
    const uint32_t * generator_words;       /* [NG x 624] the next 624 outputs of each generator                       */
} rpvg_hip_gibbs_spec;

typedef struct rpvg_hip_gibbs_sets rpvg_hip_gibbs_sets;
typedef struct rpvg_hip_gibbs_sets_view {
    uint32_t num_problems;
    uint32_t group_size;
    const uint64_t * set_off;        /* [P+1] sampled sets of each problem                                             */
    const uint32_t * first;          /* [sets] smaller member (the member, for group size 1)                           */
    const uint32_t * second;         /* [sets] larger member                                                           */
    const uint32_t * count;          /* [sets] samples that produced the set (:573); posterior = count / (chains x its) */
    const uint64_t * words_consumed; /* [NG]  32-bit words the sampler took from each generator                        */
    const uint32_t * generator_state;/* [NG x 624] for a generator that gave at least 624 words: the state words before its
                                      * next output — seeding the std::mt19937 with a seed sequence that hands these out
                                      * ([rand.eng.mers]: copied into the state, regenerated on the next call) continues
                                      * the stream at output words_consumed[g]; others are moved by discard()          */
    uint32_t rounds;                 /* rounds of (advance, conditionals) the call queued                              */
    uint64_t conditionals;           /* conditional distributions evaluated                                            */
} rpvg_hip_gibbs_sets_view;

int rpvg_hip_group_gibbs(rpvg_hip_ctx * ctx, const rpvg_hip_groups * groups, const rpvg_hip_gibbs_spec * spec,
                         rpvg_hip_gibbs_sets ** result_out);
int rpvg_hip_gibbs_sets_get(const rpvg_hip_gibbs_sets * result, rpvg_hip_gibbs_sets_view * view_out);
void rpvg_hip_gibbs_sets_free(rpvg_hip_gibbs_sets * result);

/* The whole diploid branch-and-bound of calculatePathGroupPosteriorsBounded
 * (src/path_estimator.cpp:379-473) on the GPU, one workgroup per matrix:
 * marginal posteriors (the nested group-size-1 Full call, :397-412), the
 * descending (posterior, index) order (:412), the optimistic bound of each
 * first column (:414,424-433), the pair loop with its running-maximum pruning
 * in the reference's exact sequence (:435-450), the late-loser zeroing and the
 * log-sum-exp normalisation (:453-470).  column_counts holds `path_counts` of
 * every column of every matrix, concatenated in matrix order.  The result
 * lists, per matrix, the kept pairs in the order the reference keeps them. */
typedef struct rpvg_hip_pair_posteriors rpvg_hip_pair_posteriors;

typedef struct rpvg_hip_pair_posteriors_view {
    uint32_t num_matrices;
    const uint64_t * pair_off;   /* [M+1] kept pairs of each matrix            */
    const uint32_t * first;      /* [pairs] column index (first path)          */
    const uint32_t * second;     /* [pairs] column index (second path)         */
    const double * posterior;    /* [pairs]                                    */
} rpvg_hip_pair_posteriors_view;

int rpvg_hip_bounded_pair_posteriors(rpvg_hip_ctx * ctx, const rpvg_hip_groups * groups, const uint32_t * column_counts,
                                     double min_rel_likelihood, rpvg_hip_pair_posteriors ** result_out);
int rpvg_hip_pair_posteriors_get(const rpvg_hip_pair_posteriors * result, rpvg_hip_pair_posteriors_view * view_out);
void rpvg_hip_pair_posteriors_free(rpvg_hip_pair_posteriors * result);

/* ---- diploid search -> retained path subsets -> EM, in one call ------------------- */
/* NestedPathAbundanceEstimator::inferAbundancesCollapsedGroups (src/path_abundance_estimator.cpp:428-471) from the group
 * matrices on: rpvg_hip_bounded_pair_posteriors, then selectPathSubsetIndices (:569-606: diplotypes with posterior >=
 * min_hap_prob, each expanded to the sorted list of the paths of its two haplotype columns, identical lists merged,
 * weights renormalised), then for every subset that keeps a weight >= min_hap_prob (:627-630) the EM of rpvg_hip_em_solve
 * on its distinct paths (:637-671; collapse_precision as in rpvg_hip_em_problems: the subset's rows are collapsed first, :668) —
 * with nothing but a 64-byte header crossing to the host in between (round 2 made two
 * round trips there: 1.2 ms of a lane's critical path at 32 host threads, 8 ms at 4).  The posterior-weighted merge of
 * the solutions (:702-749) stays with the caller.
 * Subsets of a matrix come in lexicographic order of their path lists.  Returns RPVG_HIP_ERR_UNSUPPORTED, having changed
 * nothing, when it does not take the input (min_hap_prob below 1/1024; a cluster too wide for LDS-resident EM vectors;
 * more subsets than the capacity it reserved — it then reserves by what it saw on the next call): the caller makes the
 * three calls instead. */
typedef struct rpvg_hip_subset_em rpvg_hip_subset_em;
typedef struct rpvg_hip_subset_em_view {
    uint32_t num_matrices;
    const uint64_t * subset_off;   /* [M+1] retained subsets of each matrix                                        */
    const double * weight;         /* [S]   normalised posterior mass of the subset (:602-605)                        */
    const uint64_t * path_off;     /* [S+1]                                                                          */
    const uint32_t * path;         /*       sorted cluster-local paths of the subset, a homozygous path twice (:583-593) */
    const uint64_t * col_off;      /* [S+1]                                                                          */
    const uint32_t * col_path;     /*       its distinct paths = the columns of its EM problem (:637-656)            */
    const double * abundances;     /*       expected read counts, laid out like col_path                             */
    const double * noise_count;    /* [S]                                                                            */
    const double * total_count;    /* [S]                                                                            */
    const uint32_t * iterations;   /* [S]   EM iterations executed                                                   */
    /* The posterior-weighted merge of the solutions (src/path_abundance_estimator.cpp:702-749), done on the device when the
     * batch was uploaded with PathInfo::group_id (rpvg_cluster_batch::path_group_id); all NULL otherwise.  The path group
     * sets of matrix m — one per (transcript, paths of that transcript in a retained subset), in lexicographic order of their
     * path lists — sit in the slots [path_off[subset_off[m]], + set_count[m]) of the set_* arrays: one path (set_second =
     * UINT32_MAX) or two, the sum of the weights of the subsets that hold the set, and per path the sum over those subsets of
     * weight x abundance / multiplicity, added in subset order with the host's roundings (bit-equal to the host merge).
     * cluster_noise_count[m] = sum of weight x noise_count over the subsets + (1 - sum of weights) x total_count (:712,749);
     * a matrix without retained subsets has set_count 0 and leaves its noise count (= its total count) to the caller. */
    const uint32_t * set_count;           /* [M] */
    const double * cluster_noise_count;   /* [M] */
    const uint32_t * set_first;
    const uint32_t * set_second;
    const double * set_posterior;
    const double * set_abundance;         /* two per set slot */
} rpvg_hip_subset_em_view;
int rpvg_hip_nested_subset_em(rpvg_hip_ctx * ctx, const rpvg_hip_batch * batch, const rpvg_hip_groups * groups,
                              const uint32_t * column_counts, double min_rel_likelihood, double min_hap_prob,
                              uint32_t max_em_its, double max_rel_em_conv, double collapse_precision,
                              rpvg_hip_subset_em ** result_out);
int rpvg_hip_subset_em_get(const rpvg_hip_subset_em * result, rpvg_hip_subset_em_view * view_out);
void rpvg_hip_subset_em_free(rpvg_hip_subset_em * result);

/* ---- minimum path cover (`-i strains`) ------------------------------------ */
/* For every listed cluster: the read-path cover and path weights of MinimumPathAbundanceEstimator::estimate
 * (src/path_abundance_estimator.cpp:233-257) and the greedy weightedMinimumPathCover (:297-340) on the GPU.
 * cover[cover_off[i] .. cover_off[i] + cover_size[i]) receives the ascending cover of clusters[i];
 * the range cover_off[i+1] - cover_off[i] must hold the cluster's number of paths. */
int rpvg_hip_min_path_cover(rpvg_hip_ctx * ctx, const rpvg_hip_batch * batch, uint32_t num_clusters,
                            const uint32_t * clusters, const uint64_t * cover_off, uint32_t * cover, uint32_t * cover_size);

/* ---- path clustering ------------------------------------------------------- */
/* PathClusters (src/path_clusters.cpp:12-86 constructor, :163-207 createPathClusters, :88-262 addNodeClusters +
 * mergeClusters): the paths of every id set — the paths one alignment-path list locates, or the paths through one
 * node — end up in one cluster.  Clusters are numbered by ascending smallest path id and list their members in
 * ascending order, as the reference does (test src/tests/path_clusters_test.cpp:82-87,130-135).
 * set s = set_path[set_off[s] .. set_off[s+1]) (non-empty; ids < num_paths).  Outputs (host): path_to_cluster
 * [num_paths]; *num_clusters_out; cluster_off[num_clusters + 1] (capacity num_paths + 1); cluster_paths [num_paths]. */
int rpvg_hip_path_clusters(rpvg_hip_ctx * ctx, uint32_t num_paths, uint64_t num_sets, const uint64_t * set_off,
                           const uint32_t * set_path, uint32_t * path_to_cluster, uint32_t * num_clusters_out,
                           uint64_t * cluster_off, uint32_t * cluster_paths);

/* ---- row construction: the step before the path (include/rpvg_rows.h) ------- */
/* ReadPathProbabilities::addPathProbs for every read of every cluster of the batch (src/read_path_probabilities.cpp:
 * 39-221) and, when merge != 0, the caller's sort + quickMergeIdentical of adjacent rows (src/main.cpp:953-973,
 * src/read_path_probabilities.cpp:223-322), on the GPU.
 *   rpvg_hip_alignments_upload   validates the alignment-path lists and makes them resident in HBM;
 *   rpvg_hip_read_rows_build     resident alignments -> resident rows (grouped layout of rpvg_cluster_batch; with
 *                                name-group collapsing the "paths" of a cluster are its groups);
 *   rpvg_hip_read_rows_to_batch  resident rows -> the rpvg_hip_batch the estimators take, without leaving the GPU;
 *   rpvg_hip_read_rows_sizes     the cluster offsets of the rows only (cluster_row_off, cluster_path_off; no copy);
 *   rpvg_hip_read_rows_view      host copy of the rows (row fields of rpvg_cluster_batch; valid until the rows are
 *                                freed) and the device time of the row kernels / of sort + merge + pack. */
typedef struct rpvg_hip_alignments rpvg_hip_alignments;
typedef struct rpvg_hip_read_rows rpvg_hip_read_rows;
int rpvg_hip_alignments_upload(rpvg_hip_ctx * ctx, const rpvg_alignment_batch * alignments, rpvg_hip_alignments ** out);
void rpvg_hip_alignments_free(rpvg_hip_ctx * ctx, rpvg_hip_alignments * alignments);
int rpvg_hip_read_rows_build(rpvg_hip_ctx * ctx, const rpvg_hip_alignments * alignments, const rpvg_row_params * params,
                             int32_t merge, rpvg_hip_read_rows ** rows_out);
int rpvg_hip_read_rows_to_batch(rpvg_hip_ctx * ctx, const rpvg_hip_read_rows * rows, rpvg_hip_batch ** batch_out);
int rpvg_hip_read_rows_sizes(rpvg_hip_ctx * ctx, const rpvg_hip_read_rows * rows, rpvg_cluster_batch * view);
int rpvg_hip_read_rows_view(rpvg_hip_ctx * ctx, rpvg_hip_read_rows * rows, rpvg_cluster_batch * view, double * build_ms,
                            double * merge_ms);
void rpvg_hip_read_rows_free(rpvg_hip_ctx * ctx, rpvg_hip_read_rows * rows);

/* ---- communicator (RCCL over xGMI; one process per GPU) --------------------- */
/* The reference is one process with OpenMP threads (src/main.cpp:829) and has no exchange step; the
 * only collectives of this engine are the per-iteration all-reduce of rpvg_hip_em_dense_sharded and
 * the sum of one double behind the TPM denominator (total_transcript_count, src/main.cpp:1029-1057)
 * when clusters are sharded over ranks.  RCCL is opened at run time (librccl.so.1) by the first of
 * these calls, so the rest of the library has no dependency on it.
 * rpvg_hip_comm_unique_id: rank 0 fills id (RPVG_HIP_COMM_ID_BYTES) and hands it to the other ranks by
 * any means (the harness broadcasts it with torch.distributed); every rank then calls
 * rpvg_hip_comm_init(ctx, id, world, rank) — collective over the ranks. */
#define RPVG_HIP_COMM_ID_BYTES 128
int rpvg_hip_comm_unique_id(uint8_t * id_out);
int rpvg_hip_comm_init(rpvg_hip_ctx * ctx, const uint8_t * id, int world_size, int rank);
int rpvg_hip_comm_destroy(rpvg_hip_ctx * ctx);
/* The same for the contexts of ONE process — one host thread per GPU instead of one process per GPU, the shape of
 * the reference's own parallelism (one process, `#pragma omp parallel for` over clusters, src/main.cpp:829):
 * context i becomes rank i of num_contexts; the contexts must sit on different GPUs. */
int rpvg_hip_comm_init_all(rpvg_hip_ctx * const * ctxs, int num_contexts);
/* In-place sum over ranks of n doubles in device memory (stream-ordered; synchronises before returning). */
int rpvg_hip_comm_allreduce_sum_f64(rpvg_hip_ctx * ctx, double * device_buf, uint64_t n);
/* The one collective of a run whose clusters are sharded over GPUs: the final gather of per-path abundances
 * (what the writers need in one place: src/main.cpp:1020-1083).  Collective over the ranks of the context's
 * communicator (every rank calls it, from its own host thread or process): rank r brings counts[r] host values,
 * every rank receives all of them in rank order (all_values: sum of counts; ncclAllGather over xGMI, ragged
 * lengths padded to the longest).  A context without a communicator is a world of one. */
int rpvg_hip_gather(rpvg_hip_ctx * ctx, const double * local_values, uint64_t local_count, const uint64_t * counts, double * all_values);

/* ---- synthetic workload (bench / tests only) ---------------------------- */
/* Fills a dense normalised R x C matrix (layout of rpvg_hip_em_dense) and unit
 * counts on the GPU from a counter-based generator: the "1M read pairs x 2k
 * paths single dense cluster" configuration. */
int rpvg_hip_synth_dense_cluster(rpvg_hip_ctx * ctx, uint64_t seed, uint64_t num_rows, uint32_t num_paths,
                                 double * device_matrix, uint64_t ld, double * device_counts);
/* Rows [row_begin, row_begin + num_rows) of the same cluster (a rank's shard of it). */
int rpvg_hip_synth_dense_rows(rpvg_hip_ctx * ctx, uint64_t seed, uint64_t row_begin, uint64_t num_rows, uint32_t num_paths,
                              double * device_matrix, uint64_t ld, double * device_counts);

/* The same cluster as a cluster batch resident on the GPU (one cluster, every row holding all num_paths paths) — what
 * the estimator classes take: BASELINE.json configs[1] behind `-i transcripts` without 40 GB of host arrays (SURVEY.md
 * section 8d S2 generates it on the device for the same reason).  Freed with rpvg_hip_batch_free. */
int rpvg_hip_synth_dense_cluster_batch(rpvg_hip_ctx * ctx, uint64_t seed, uint64_t num_rows, uint32_t num_paths,
                                       rpvg_hip_batch ** batch_out);

/* Test hook: the FP64 logarithm of the log-likelihood kernels (positive normal x) evaluated on the device,
 * with (use_table != 0) or without the LDS table — tests/test_hip_kernels.py measures its error. */
int rpvg_hip_debug_log(rpvg_hip_ctx * ctx, uint64_t n, const double * x, double * out, int32_t use_table);

/* ---- instrumentation ----------------------------------------------------- */
/* The EM kernels of rpvg_hip_em_solve (rpvg_amd/csrc/em_sparse.hip): one variant per size bin of the problems; each
 * carries its own device time (HIP events on the stream it is launched on), launches, problems, EM iterations and
 * algorithmic bytes (per iteration of a problem 12 B/entry + 20 B/row + 16 B/column, DESIGN.md section 3, docs/design/kernels-em.md). */
#define RPVG_HIP_EM_KERNELS 12
typedef struct rpvg_hip_em_kernel_stats {
    double ms;               /* sum over launches of the HIP-event span around the launch on its own stream */
    uint64_t launches;
    uint64_t problems;
    uint64_t iterations;     /* EM iterations of all problems of all launches */
    uint64_t max_iterations; /* sum over launches of the iteration count of the launch's slowest problem */
    double alg_bytes;
} rpvg_hip_em_kernel_stats;
/* "emRegisterKernel<1,16>", "emSparseKernel<64,true>", ...; NULL for an index outside [0, RPVG_HIP_EM_KERNELS) */
const char * rpvg_hip_em_kernel_name(int index);

/* Device time (HIP events on the stream the kernels are launched on) and launch count of the
 * kernels of each family since the last reset, plus the algorithmic bytes
 * they processed (DESIGN.md defines the per-launch figure). */
typedef struct rpvg_hip_kernel_stats {
    double em_sparse_ms;      uint64_t em_sparse_launches;   double em_sparse_alg_bytes;
    double em_dense_ms;       uint64_t em_dense_launches;    double em_dense_alg_bytes;
    double loglik_ms;         uint64_t loglik_launches;      double loglik_evals;   /* evals = FP64 log evaluations */
    double build_ms;          uint64_t build_launches;
    double h2d_ms;            double h2d_bytes;
    uint64_t em_iterations_total;
    /* diploid branch-and-bound (rpvg_hip_bounded_pair_posteriors): pairs of columns the searched matrices have
     * (G (G + 1) / 2 each), pairs among them that belong to matrices on the pair-table path (all of them evaluated),
     * and pairs kept — the reference evaluates the kept pairs plus the ones it prunes one by one. */
    double search_pairs_possible; double search_pairs_table; double search_pairs_kept;
    /* em_sparse_* split by kernel variant (em_sparse_ms is the span of a whole rpvg_hip_em_solve's kernels, whose bins
     * run side by side on several streams; the per-kernel spans overlap each other) */
    rpvg_hip_em_kernel_stats em_kernel[RPVG_HIP_EM_KERNELS];
    double collapse_ms;       /* row collapse of the group matrices (row_collapse.hip), on the collapse stream */
    /* length of the union of all timed spans of this context (kernels and copies; a span runs from the first command of
     * a stage to its last on the stage's stream, so this is an upper bound of the time the GPU worked for the context) */
    double busy_ms;
    /* rpvg_hip_group_gibbs: the span from the sampler's first kernel to its last (chains, request bookkeeping, distributions AND the
     * conditionals, whose own spans are in loglik_ms: gibbs_ms - loglik_ms is what the chains and their bookkeeping take) */
    double gibbs_ms;
    /* pairTile2Kernel alone (round 6): its own HIP-event spans on the stream it runs on and its launches — the kernel the row-pair
     * evaluations of loglik_evals belong to on the diploid search's table path (loglik_ms also holds the kernels behind it, and the
     * spans of batches in flight overlap); timed by contexts that time every kernel family (RPVG_HIP_SPANS=2) */
    double search_tile_ms;    uint64_t search_tile_launches;
} rpvg_hip_kernel_stats;

int rpvg_hip_stats_get(rpvg_hip_ctx * ctx, rpvg_hip_kernel_stats * stats_out);
int rpvg_hip_stats_reset(rpvg_hip_ctx * ctx);
/* The timed spans behind the statistics since the last reset, as intervals in milliseconds on a clock shared by all
 * contexts of the GPU (so that a caller with several contexts on one GPU — the host lanes of rpvg_amd::HipEngine — can
 * take the union over them).  Writes at most `capacity` intervals (any array may be NULL) and the number there are. */
int rpvg_hip_stats_intervals(rpvg_hip_ctx * ctx, uint64_t capacity, double * start_ms, double * stop_ms, int32_t * family,
                             uint64_t * count_out);

#ifdef __cplusplus
}
#endif

#endif /* RPVG_HIP_H */
