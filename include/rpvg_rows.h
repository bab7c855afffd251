/* rpvg_rows.h — plain-C description of the step immediately BEFORE the inference hot path: turning the
 * alignment paths of every read (pair) of a path cluster into the cluster's merged ReadPathProbabilities rows
 * (SURVEY.md §8f rank 2).
 *
 * What it replaces in the reference (paths relative to the rpvg checkout):
 *   - ReadPathProbabilities::addPathProbs / calcAlignPathLogProbs   src/read_path_probabilities.cpp:39-221
 *   - the caller's loop over the distinct alignment-path lists of a cluster, followed by sort +
 *     quickMergeIdentical of adjacent rows                          src/main.cpp:889-973
 *     (ReadPathProbabilities::operator<, quickMergeIdentical        src/read_path_probabilities.cpp:223-322)
 *   - FragmentLengthDist::logProb as a table                        src/fragment_length_dist.cpp:385-427
 *
 * A "read" here is one distinct vector<AlignmentPath> of the reference's align_paths_index with its multiplicity
 * (src/main.cpp:893-905).  Its last AlignmentPath is the noise entry (empty path list, lengths 0, score <= 0,
 * src/read_path_probabilities.cpp:43-46); it is carried as read_noise_score and NOT listed among the alignments.
 * gbwt path ids are already mapped to cluster-local path indices (clustered_path_index, src/main.cpp:846-887);
 * the indices of one alignment are distinct and ascending (the result does not depend on their order).
 * Everything is caller-owned host memory, read-only for the callee.
 */
#ifndef RPVG_ROWS_H
#define RPVG_ROWS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RPVG_FRAG_LENGTH_TABLE_SIZE 65536 /* AlignmentPath::frag_length is uint16_t (src/alignment_path.hpp:35) */

typedef struct rpvg_alignment_batch {
    uint32_t num_clusters;             /* K */
    const uint64_t * cluster_read_off; /* [K+1] reads of each cluster */
    const uint64_t * cluster_path_off; /* [K+1] paths of each cluster */

    /* paths: the PathInfo fields addPathProbs reads */
    const double * path_effective_length; /* [P] PathInfo::effective_length; 0 = path cannot emit the read */
    const uint32_t * path_source_count;   /* [P] PathInfo::source_count (collapse only; may be NULL) */
    /* --collapse-haps (collapse_groups): paths sharing a name form one output column.  path_group[p] = the
     * cluster-local group index of path p (group_name_index), cluster_group_off = groups of each cluster.
     * Both NULL = no collapsing. */
    const uint32_t * path_group;          /* [P] or NULL */
    const uint64_t * cluster_group_off;   /* [K+1] or NULL */

    /* reads */
    const uint32_t * read_count;       /* [N] multiplicity of the alignment-path list (align_paths->second) */
    const uint8_t * read_min_mapq;     /* [N] align_paths.front().min_mapq (shared by the read's alignments) */
    const int32_t * read_noise_score;  /* [N] align_paths.back().score_sum  (<= 0) */
    const uint64_t * read_align_off;   /* [N+1] alignments of each read (noise entry excluded; >= 1 each) */

    /* alignments */
    const int32_t * align_score_sum;    /* [A] AlignmentPath::score_sum    */
    const uint16_t * align_length;      /* [A] AlignmentPath::align_length (> 0) */
    const uint16_t * align_frag_length; /* [A] AlignmentPath::frag_length  */
    const uint64_t * align_path_off;    /* [A+1] */
    const uint32_t * align_path_idx;    /* [E] cluster-local path indices, ascending within an alignment */
} rpvg_alignment_batch;

typedef struct rpvg_row_params {
    double prob_precision;  /* --prob-precision  1e-8  src/main.cpp:402 */
    double min_noise_prob;  /* --min-noise-prob  1e-4  src/main.cpp:404 */
    int32_t is_single_end;  /* -e: the fragment length term is skipped  src/read_path_probabilities.cpp:58-61 */
    /* log density of every fragment length 0 .. 65535: FragmentLengthDist::logProb(v).  Ignored if is_single_end. */
    const double * frag_length_log_prob; /* [RPVG_FRAG_LENGTH_TABLE_SIZE] */
} rpvg_row_params;

#ifdef __cplusplus
}
#endif

#endif /* RPVG_ROWS_H */
