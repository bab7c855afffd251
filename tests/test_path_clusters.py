"""Path clustering (PathClusters, src/path_clusters.cpp) — the oracle against the reference's own test
(src/tests/path_clusters_test.cpp:11-135), and the GPU union-find against both.

The reference test builds a GBWT with four paths over nodes 1..7 — thread 1 = (1+, 2+, 4+), thread 2 = (1-, 6-),
thread 3 = (3+), thread 4 = (6-, 7-) — and clusters paths that share an oriented node (addNodeClusters); with a
bidirectional index the orientation is ignored.  Transcribed here as the id sets "paths through a node" that
addNodeClusters obtains from locatePathIds."""
import numpy as np
import pytest

from oracle import pyoracle

UNIDIRECTIONAL_NODE_SETS = [[0], [1], [0], [2], [0], [1, 3], [3]]   # 1+, 1-, 2+, 3+, 4+, 6-, 7-
BIDIRECTIONAL_NODE_SETS = [[0, 1], [0], [2], [0], [1, 3], [3]]      # 1, 2, 3, 4, 6, 7


def test_oracle_reference_case_unidirectional():  # :82-87
    p2c, members, _ = pyoracle.path_clusters(4, UNIDIRECTIONAL_NODE_SETS)
    assert p2c.tolist() == [0, 1, 2, 1]
    assert members == [[0], [1, 3], [2]]


def test_oracle_reference_case_bidirectional():  # :130-135
    p2c, members, _ = pyoracle.path_clusters(4, BIDIRECTIONAL_NODE_SETS)
    assert p2c.tolist() == [0, 0, 1, 0]
    assert members == [[0, 1, 3], [2]]


def test_oracle_no_sets_gives_singletons():
    p2c, members, _ = pyoracle.path_clusters(5, [])
    assert p2c.tolist() == [0, 1, 2, 3, 4] and members == [[0], [1], [2], [3], [4]]


def random_sets(seed, num_paths, num_sets, max_size):
    rng = np.random.default_rng(seed)
    return [[int(x) for x in rng.choice(num_paths, size=int(rng.integers(1, max_size + 1)), replace=False)] for _ in range(num_sets)]


def test_oracle_clusters_are_the_connected_components():
    sets = random_sets(5, 300, 120, 4)
    p2c, members, _ = pyoracle.path_clusters(300, sets)
    assert sorted(p for m in members for p in m) == list(range(300))
    assert [m[0] for m in members] == sorted(m[0] for m in members)  # numbered by ascending smallest path id
    assert all(m == sorted(m) for m in members)
    for ids in sets:
        assert len({int(p2c[i]) for i in ids}) == 1


@pytest.mark.gpu
def test_gpu_reference_cases(hip_ctx):
    p2c, members = hip_ctx.path_clusters(4, UNIDIRECTIONAL_NODE_SETS)
    assert p2c.tolist() == [0, 1, 2, 1] and members == [[0], [1, 3], [2]]
    p2c, members = hip_ctx.path_clusters(4, BIDIRECTIONAL_NODE_SETS)
    assert p2c.tolist() == [0, 0, 1, 0] and members == [[0, 1, 3], [2]]
    p2c, members = hip_ctx.path_clusters(3, [])
    assert p2c.tolist() == [0, 1, 2] and members == [[0], [1], [2]]


@pytest.mark.gpu
@pytest.mark.parametrize("seed,num_paths,num_sets,max_size", [(1, 50, 30, 3), (2, 5000, 4000, 5), (3, 200000, 150000, 8),
                                                             (4, 1000, 3000, 40)])
def test_gpu_matches_oracle(hip_ctx, seed, num_paths, num_sets, max_size):
    sets = random_sets(seed, num_paths, num_sets, max_size)
    p2c_o, members_o, _ = pyoracle.path_clusters(num_paths, sets)
    p2c, members = hip_ctx.path_clusters(num_paths, sets)
    assert np.array_equal(p2c, p2c_o)
    assert members == members_o


@pytest.mark.gpu
def test_gpu_long_chain_and_one_giant_set(hip_ctx):
    n = 100000
    chain = [[i, i + 1] for i in range(n - 1)]  # worst case for pointer chasing
    p2c, members = hip_ctx.path_clusters(n, chain)
    assert not p2c.any() and members == [list(range(n))]
    p2c, members = hip_ctx.path_clusters(n, [list(range(n - 1, -1, -1))])
    assert not p2c.any() and len(members) == 1
