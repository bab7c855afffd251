"""Row construction (ReadPathProbabilities::addPathProbs, quickMergeIdentical, the caller's sort + merge) —
the CPU oracle against the reference's OWN test vectors: every case of
src/tests/read_path_probabilities_test.cpp:9-205 (inputs and expected values transcribed as data below; the
expectations use the reference's comparison, Utils::doubleCompare).  This row of the oracle is therefore pinned.

AlignmentPath(search, is_simple, min_mapq, score_sum, align_length, frag_length)  (src/alignment_path.hpp:26)
is written here as (score_sum, align_length, frag_length, [cluster-local path indices]).
"""
import copy

import numpy as np
import pytest

from oracle import pyoracle
from rpvg_amd.rows import INT32_LOWEST, AlignmentBatch, RowParams

NOISE_SCORE_LOG_BASE = 1e-6  # Utils::noise_score_log_base, src/utils.hpp:84


def frag_table():
    return pyoracle.frag_length_table(10, 2, 0.0, 10)  # FragmentLengthDist(10, 2, 10), test :12


def base_case():
    """test :11-26 — clustered_path_index {100: 0, 200: 1}; one alignment on both paths; mapq 10."""
    return dict(paths=[dict(effective_length=3.0), dict(effective_length=3.0)],
                reads=[dict(count=1, min_mapq=10, noise_score=INT32_LOWEST, aligns=[(3, 5, 10, [0, 1])])])


def multi_case():
    """test :51-66 — second alignment (score 5, length 8, fragment 15) on path id 50 -> index 3; four paths."""
    c = base_case()
    c["paths"] += [dict(effective_length=3.0), dict(effective_length=3.0)]
    c["reads"][0]["aligns"].append((5, 8, 15, [3]))
    return c


def build(cluster, precision=1e-8, min_noise=0.0, merge=False):
    params = RowParams(prob_precision=precision, min_noise_prob=min_noise, is_single_end=False, frag_length_log_prob=frag_table())
    rows, _ = pyoracle.build_rows(AlignmentBatch.from_clusters([cluster]), params, merge=merge)
    return rows.cluster(0)["rows"]


def same(a, b):
    return pyoracle.lib().rpvg_oracle_double_compare(float(a), float(b)) == 1


def check_row(row, count, noise, groups):
    assert row[0] == count
    assert same(row[1], noise), (row[1], noise)
    assert len(row[2]) == len(groups)
    for (p, idx), (wp, widx) in zip(row[2], groups):
        assert same(p, wp), (p, wp)
        assert idx == widx


def test_single_alignment_two_paths():  # :28-33
    (row,) = build(base_case())
    check_row(row, 1, 0.1, [(0.45, [0, 1])])


def test_improbable_alignment_path_returns_finite_probabilities():  # :35-47
    c = base_case()
    c["reads"][0]["aligns"][0] = (3, 5, 10000, [0, 1])
    (row,) = build(c)
    (ref,) = build(base_case())
    assert row[0] == ref[0] and same(row[1], ref[1])
    assert len(row[2]) == 1 and abs(row[2][0][0] - ref[2][0][0]) < 1e-8 and row[2][0][1] == ref[2][0][1]


def test_multiple_alignment_paths():  # :49-75
    (row,) = build(multi_case())
    check_row(row, 1, 0.1, [(0.233044027062125, [3]), (0.333477986468937, [0, 1])])


def test_probability_precision_affects_number_of_unique_probabilities():  # :77-90
    c = multi_case()
    c["paths"][-1]["effective_length"] = 2.0
    (row,) = build(c, precision=0.1)
    check_row(row, 1, 0.1, [(0.3, [0, 1, 3])])


def test_longest_alignment_path_is_always_chosen():  # :92-108
    c = multi_case()
    c["reads"][0]["aligns"].append((3, 10, 10, [3]))
    (row,) = build(c, precision=0.1)
    check_row(row, 1, 0.1, [(0.3, [0, 1, 3])])


def test_highest_scoring_alignment_path_is_chosen_if_identical():  # :110-128
    c = multi_case()
    c["reads"][0]["aligns"].append((3, 8, 15, [3]))
    (row,) = build(c, precision=0.1)
    (ref,) = build(multi_case())
    assert row[0] == ref[0] and same(row[1], ref[1]) and len(row[2]) == 2
    for (p, idx), (wp, widx) in zip(row[2], ref[2]):
        assert abs(p - wp) < 1e-8 and idx == widx


def test_noise_alignment_path_affects_noise_probability():  # :131-154
    c = base_case()
    c["reads"][0]["noise_score"] = int(-2.302585 / NOISE_SCORE_LOG_BASE)  # int32_t score_sum = double, truncated
    (row,) = build(c)
    check_row(row, 1, 0.190000008369464, [(0.404999995815267, [0, 1])])
    c["reads"][0]["noise_score"] = 0
    (row,) = build(c)
    check_row(row, 1, 1.0, [])


def test_effective_path_lengths_affect_path_probabilities():  # :156-171
    c = base_case()
    c["paths"][-1]["effective_length"] = 2.0
    (row,) = build(c)
    check_row(row, 1, 0.1, [(0.36, [0]), (0.54, [1])])


def test_base_noise_probability_affects_path_probabilities():  # :173-186
    c = base_case()
    c["reads"][0]["noise_score"] = int(-5.0 / NOISE_SCORE_LOG_BASE)
    (row,) = build(c, min_noise=0.3)
    check_row(row, 1, 0.304716562899359, [(0.347641718550320, [0, 1])])


def test_identical_read_path_probabilities_can_be_merged():  # :189-204
    c = base_case()
    c["reads"].append(copy.deepcopy(c["reads"][0]))
    (row,) = build(c, merge=True)
    check_row(row, 2, 0.1, [(0.45, [0, 1])])


# ---- beyond the reference's cases: properties of the restatement ---------------------------------------------

def test_zero_mapq_read_is_pure_noise():  # src/read_path_probabilities.cpp:89 — nothing happens unless min_mapq > 0
    c = base_case()
    c["reads"][0]["min_mapq"] = 0
    (row,) = build(c)
    check_row(row, 1, 1.0, [])


def test_zero_effective_length_path_is_skipped():  # :120-123
    c = base_case()
    c["paths"][1]["effective_length"] = 0.0
    (row,) = build(c)
    check_row(row, 1, 0.1, [(0.9, [0])])


def test_sub_precision_probabilities_move_to_noise():  # :208-217
    c = base_case()
    c["paths"].append(dict(effective_length=3.0))
    c["reads"][0]["aligns"].append((-20, 5, 10, [2]))  # exp(-23 * 1.38) ~ 1.5e-14 relative
    (row,) = build(c)
    assert row[2][0][1] == [0, 1] and len(row[2]) == 1
    total = row[1] + sum(p * len(idx) for p, idx in row[2])
    assert abs(total - 1) < 1e-12


def test_collapsed_groups_sum_member_paths():  # :151-168 — group log prob = add_log over members of lp + log(source_count)
    c = dict(paths=[dict(effective_length=3.0, source_count=2, group=0), dict(effective_length=3.0, source_count=1, group=0),
                    dict(effective_length=3.0, source_count=1, group=1)],
             reads=[dict(count=4, min_mapq=10, noise_score=INT32_LOWEST, aligns=[(3, 5, 10, [0, 1, 2])])])
    (row,) = build(c)
    check_row(row, 4, 0.1, [(0.25 * 0.9, [1]), (0.75 * 0.9, [0])])


def test_rows_always_sum_to_one_and_merge_preserves_counts():
    rng = np.random.default_rng(11)
    paths = [dict(effective_length=float(rng.integers(50, 3000))) for _ in range(12)]
    reads = []
    for _ in range(300):
        aligns = []
        for _ in range(int(rng.integers(1, 4))):
            idx = sorted(set(int(x) for x in rng.integers(0, 12, size=int(rng.integers(1, 5)))))
            aligns.append((int(rng.integers(-10, 60)), int(rng.integers(50, 150)), int(rng.integers(1, 40)), idx))
        reads.append(dict(count=int(rng.integers(1, 4)), min_mapq=int(rng.choice([0, 3, 10, 30, 60])),
                          noise_score=int(rng.choice([INT32_LOWEST, -3000000, -500000])), aligns=aligns))
    c = dict(paths=paths, reads=reads)
    raw = build(c, min_noise=1e-4)
    assert len(raw) == 300
    for cnt, noise, groups in raw:
        assert 0 < noise <= 1
        assert abs(noise + sum(p * len(idx) for p, idx in groups) - 1) < 1e-12
        assert all(groups[i][0] <= groups[i + 1][0] for i in range(len(groups) - 1))
    merged = build(c, min_noise=1e-4, merge=True)
    assert sum(r[0] for r in merged) == sum(r["count"] for r in reads)
    assert len(merged) <= len(raw)
