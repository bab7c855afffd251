/* rpvg_batch.h — plain-C description of the data that crosses the inference
 * hot path of rpvg, flattened so that it can cross a C ABI.
 *
 * What it replaces in the reference (paths relative to the rpvg checkout):
 *   - input : `const vector<ReadPathProbabilities> & cluster_probs` and
 *             `PathClusterEstimates::paths` as handed to
 *             `PathEstimator::estimate()`            src/path_estimator.hpp:23,
 *             filled by the caller                   src/main.cpp:846-973;
 *             row type                               src/read_path_probabilities.hpp:39-43
 *             path type                              src/path_cluster_estimates.hpp:15-33
 *   - output: the remaining fields of `PathClusterEstimates`
 *                                                    src/path_cluster_estimates.hpp:49-57
 *   - knobs : the command-line options of this path  src/main.cpp:402-418
 *
 * A batch is K clusters back to back.  Everything is caller-owned host memory,
 * read-only for the callee.  Cluster k owns rows
 * [cluster_row_off[k], cluster_row_off[k+1]) and paths
 * [cluster_path_off[k], cluster_path_off[k+1]); path indices inside rows are
 * cluster-local (0 .. N_k-1), exactly as in ReadPathProbabilities::pathProbs().
 */
#ifndef RPVG_BATCH_H
#define RPVG_BATCH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rpvg_cluster_batch {
    uint32_t num_clusters;             /* K */
    const uint64_t * cluster_row_off;  /* [K+1] */
    const uint64_t * cluster_path_off; /* [K+1] */

    /* rows: one per merged ReadPathProbabilities */
    const uint32_t * row_count;        /* [R]   readCount()                         */
    const double * row_noise;          /* [R]   noiseProb()                         */
    const uint64_t * row_grp_off;      /* [R+1] range of (prob, path list) groups   */
    const double * grp_prob;           /* [G]   pathProbs()[g].first, ascending     */
    const uint64_t * grp_idx_off;      /* [G+1] range into path_idx                 */
    const uint32_t * path_idx;         /* [NNZ] pathProbs()[g].second, cluster-local */

    /* paths: the PathInfo fields the estimators read */
    const uint32_t * path_group_id;     /* [P]   PathInfo::group_id                 */
    const uint32_t * path_source_count; /* [P]   PathInfo::source_count             */
    const uint64_t * path_source_off;   /* [P+1] range into source_id               */
    const uint32_t * source_id;         /* [S]   PathInfo::source_ids (any order)   */
    const double * path_effective_length; /* [P] PathInfo::effective_length (TPM only; may be NULL) */

    /* The two long offset arrays in 32 bits, for a batch of fewer than 2^32 groups and entries: used INSTEAD of row_grp_off /
     * grp_idx_off when not NULL (those may then be NULL).  A third of a batch's bytes are offsets, and the copy to the GPU is
     * what paces a pipeline of batches (rpvg_amd/host/batch_pipeline.hpp): whoever flattens rows writes these. */
    const uint32_t * row_grp_off32;    /* [R+1] */
    const uint32_t * grp_idx_off32;    /* [G+1] */

    /* ... or, for the copy to the GPU, as one byte each: the groups of every row and the paths of every group, for a batch
     * none of whose rows has more than 255 groups and none of whose groups more than 255 paths.  When BOTH are given
     * rpvg_hip_batch_upload copies them instead of the offset arrays (which it then does not read: they may be NULL) and sums them
     * up on the device; num_groups / num_entries are their totals (G and NNZ), which the counts must add up to.  Another
     * 40 MB less of the 230 MB of a configs[2] batch.  Whoever walks a batch on the host reads the offsets: a batch that is
     * to be sharded, written out or handed to a host estimator keeps them (the accessors below do not look at the counts). */
    const uint8_t * row_grp_count8;    /* [R] */
    const uint8_t * grp_idx_count8;    /* [G] */
    uint64_t num_groups;               /* G   (read with the counts only) */
    uint64_t num_entries;              /* NNZ (read with the counts only) */

    /* Narrower forms of three more arrays, again for the copy to the GPU only (rpvg_hip_batch_upload takes each INSTEAD of its
     * 32-bit array when it is not NULL — that one may then be NULL — and widens it on the device behind the copy; the device batch
     * is the same either way, to the bit).  Cluster-local path indices in 16 bits: no cluster of the batch has 65 536 paths or
     * more (src/main.cpp's clusters have up to a few thousand).  Source (haplotype) ids in 16 bits: all ids below 65 536.  Read
     * counts in one byte: min(count, 255), the rows with a count of 255 or more (ascending) and their counts listed beside.
     * With the noise table below, configs[2]: 190 -> 136 MB per batch. */
    const uint16_t * path_idx16;               /* [NNZ] */
    const uint16_t * source_id16;              /* [S]   */
    const uint8_t * row_count8;                /* [R]   */
    const uint32_t * row_count_escape_row;     /* [E]   rows whose row_count8 is 255 */
    const uint32_t * row_count_escape_count;   /* [E]   their read counts           */
    uint64_t num_row_count_escapes;            /* E     */
    /* ... and the noise probabilities as 16-bit indices into the table of their distinct values (they come from mapping qualities:
     * a few hundred values for millions of rows; at most 65 536): row r has noise row_noise_table[row_noise16[r]], the very double. */
    const uint16_t * row_noise16;              /* [R]   instead of row_noise */
    const double * row_noise_table;            /* [T]   */
    uint64_t num_row_noise_values;             /* T     */
} rpvg_cluster_batch;

/* One cluster as the thread that calls PathEstimator::estimate() for it (src/main.cpp:977) flattens it: the arrays of
 * rpvg_cluster_batch for a single cluster, offsets local and in 32 bits, all of them inside ONE block of page-locked memory
 * from rpvg_hip_pinned_alloc (include/rpvg_hip.h) that the GPU reads where it lies.  rpvg_hip_batch_upload_segments takes the
 * segments of the calls that are in flight at once as one batch: no joined copy on the host, no staging, no copy commands —
 * one kernel pulls the segments over PCIe and writes the device batch.  Every `*_at` is a byte offset from `base`, a multiple
 * of 8; the arrays must lie inside [base, base + bytes).  Without paths (has_paths = 0: the batch then carries no haplotype
 * columns) the three path arrays are not read; all segments of a batch with them, or none. */
typedef struct rpvg_cluster_segment {
    const void * base;           /* a block from rpvg_hip_pinned_alloc */
    uint64_t bytes;              /* of the block that the segment uses */
    uint32_t num_rows;           /* R   */
    uint32_t num_groups;         /* G   */
    uint32_t num_entries;        /* NNZ */
    uint32_t num_paths;          /* P   */
    uint32_t num_sources;        /* S   */
    uint32_t has_paths;
    uint64_t total_read_count;   /* sum of row_count: the flattening thread adds them up on the way */
    uint64_t row_count_at;       /* uint32 [R]    readCount()                              */
    uint64_t row_noise_at;       /* double [R]    noiseProb()                              */
    uint64_t row_grp_off_at;     /* uint32 [R+1]  groups of every row, from 0              */
    uint64_t grp_idx_off_at;     /* uint32 [G+1]  paths of every group, from 0             */
    uint64_t grp_prob_at;        /* double [G]    pathProbs()[g].first                     */
    uint64_t path_idx_at;        /* uint32 [NNZ]  pathProbs()[g].second, cluster-local     */
    uint64_t path_group_id_at;   /* uint32 [P]    PathInfo::group_id                       */
    uint64_t path_source_off_at; /* uint32 [P+1]  source ids of every path, from 0         */
    uint64_t source_id_at;       /* uint32 [S]    PathInfo::source_ids                     */
    /* The haplotype columns of the cluster, formed by the caller (findPathSourceGroups, src/path_abundance_estimator.cpp:493-546:
     * the reference does this per cluster on the calling thread too): haplotypes with the identical path list are one column, its
     * multiplicity their number, columns in ascending order of their smallest haplotype id, a column's paths ascending.  With
     * them (has_columns = 1: all segments of a batch, or none) the upload forms no columns on the device and waits for nothing
     * but its own kernel; path_source_off / source_id are not read then (path_group_id is). */
    uint32_t has_columns;
    uint32_t num_columns;        /* C                                                       */
    uint32_t num_column_paths;   /* L = sum of the columns' list lengths (>= C)             */
    uint32_t max_column_paths;   /* the longest list                                        */
    uint64_t col_count_at;       /* uint32 [C]    haplotypes that carry the column's list   */
    uint64_t col_end_at;         /* uint32 [C]    end of the column's list in col_path      */
    uint64_t col_path_at;        /* uint32 [L]    the lists, one behind the other           */
} rpvg_cluster_segment;

/* The two long offset arrays of a batch in whichever width its owner wrote them. */
static inline uint64_t rpvg_batch_row_group_offset(const rpvg_cluster_batch * batch, uint64_t row) {
    return batch->row_grp_off32 ? batch->row_grp_off32[row] : batch->row_grp_off[row];
}
static inline uint64_t rpvg_batch_group_entry_offset(const rpvg_cluster_batch * batch, uint64_t group) {
    return batch->grp_idx_off32 ? batch->grp_idx_off32[group] : batch->grp_idx_off[group];
}

/* Every option of src/main.cpp that reaches the estimators, same defaults
 * (see rpvg_params_default()). */
typedef struct rpvg_params {
    uint32_t max_em_its;        /* --max-em-its        10000   main.cpp:416 */
    double max_rel_em_conv;     /* --max-rel-em-conv   0.001   main.cpp:417 */
    uint32_t num_gibbs_samples; /* -n                  0       main.cpp:415 */
    uint32_t gibbs_thin_its;    /* --gibbs-thin-its    25      main.cpp:418 */
    double prob_precision;      /* --prob-precision    1e-8    main.cpp:402 */
    uint32_t ploidy;            /* -y                  2       main.cpp:407 */
    double min_hap_prob;        /* --min-hap-prob      0.001   main.cpp:409 */
    int32_t ind_hap_inference;  /* --ind-hap-inference 0       main.cpp:410 */
    int32_t use_hap_gibbs;      /* --use-hap-gibbs     0       main.cpp:411 */
    uint32_t rng_seed;          /* -r; cluster i uses mt19937(rng_seed + i)  main.cpp:976 */
} rpvg_params;

static inline rpvg_params rpvg_params_default(void) {
    rpvg_params p;
    p.max_em_its = 10000;
    p.max_rel_em_conv = 0.001;
    p.num_gibbs_samples = 0;
    p.gibbs_thin_its = 25;
    p.prob_precision = 1e-8;
    p.ploidy = 2;
    p.min_hap_prob = 0.001;
    p.ind_hap_inference = 0;
    p.use_hap_gibbs = 0;
    p.rng_seed = 0;
    return p;
}

/* Read-only view of the estimates of a batch (callee-owned until the result
 * handle is freed).  Cluster k owns group sets [set_off[k], set_off[k+1]);
 * set s owns members [member_off[s], member_off[s+1]) — cluster-local path
 * indices, PathClusterEstimates::path_group_sets — and posteriors[s];
 * cluster k owns abundances [abund_off[k], abund_off[k+1]) laid out as in
 * PathClusterEstimates::abundances (one per set for `transcripts`/`strains`,
 * one per set member for `haplotype-transcripts`, none for `haplotypes`).
 * em_* is instrumentation the reference does not have: the iteration count
 * of every EM solve and the cluster-local paths that formed its columns. */
typedef struct rpvg_estimates_view {
    uint32_t num_clusters;
    const uint64_t * set_off;     /* [K+1] */
    const uint64_t * member_off;  /* [S+1] */
    const uint32_t * members;     /* [M]   */
    const double * posteriors;    /* [S]   */
    const uint64_t * abund_off;   /* [K+1] */
    const double * abundances;    /* [A]   */
    const double * noise_count;   /* [K]   */
    const double * total_count;   /* [K]   */
    const uint64_t * em_off;      /* [K+1] EM solves per cluster */
    const uint32_t * em_iters;    /* [E]   */
    const uint64_t * em_col_off;  /* [E+1] */
    const uint32_t * em_cols;     /* [..]  */
    /* PathClusterEstimates::gibbs_read_count_samples (-n > 0): cluster k owns CountSamples
     * [gibbs_off[k], gibbs_off[k+1]); CountSamples g has path_ids [gibbs_path_off[g], ..+1),
     * noise_samples [gibbs_noise_off[g], ..+1) and abundance_samples (sample-major)
     * [gibbs_abund_off[g], ..+1)  (src/path_cluster_estimates.hpp:35-43). */
    const uint64_t * gibbs_off;        /* [K+1] */
    const uint64_t * gibbs_path_off;   /* [Gs+1] */
    const uint32_t * gibbs_path;
    const uint64_t * gibbs_noise_off;  /* [Gs+1] */
    const double * gibbs_noise;
    const uint64_t * gibbs_abund_off;  /* [Gs+1] */
    const double * gibbs_abund;
} rpvg_estimates_view;

#ifdef __cplusplus
}
#endif

#endif /* RPVG_BATCH_H */
