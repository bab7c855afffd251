"""CPU-only checks (no GPU needed): the C-ABI library loads and exports every symbol its header
declares, the engine fails loudly without a GPU (no CPU fallback), and the host-side row type follows
the reference's ReadPathProbabilities semantics."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from rpvg_amd import hip, synth
from rpvg_amd.batch import ClusterBatch
from tests import small_cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rpvg_hip_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    declared = _declared_functions("rpvg_hip.h")
    assert len(declared) >= 20
    lib = hip.lib()
    for name in declared:
        assert hasattr(lib, name), f"{name} is declared in include/rpvg_hip.h but not exported"
    # the Python binding list is the header's list
    assert sorted(hip.EXPORTS) == declared


def test_host_library_exports_runner_symbols():
    from rpvg_amd import engine
    lib = engine.lib()
    for name in ("rpvg_amd_engine_create", "rpvg_amd_engine_create_uploader", "rpvg_amd_engine_destroy", "rpvg_amd_batch_prepare", "rpvg_amd_batch_free",
                 "rpvg_amd_run", "rpvg_amd_run_inplace", "rpvg_amd_result_view", "rpvg_amd_result_free",
                 "rpvg_amd_synth_generate", "rpvg_amd_rows_from_likelihoods", "rpvg_amd_last_error"):
        assert hasattr(lib, name)


def test_engine_fails_loudly_without_gpu():
    if hip.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(hip.EngineError) as err:
        hip.Context(0)
    assert "no CPU fallback" in str(err.value) or "no HIP device" in str(err.value)
    from rpvg_amd import engine
    with pytest.raises(hip.EngineError):
        engine.Engine(0)


def test_product_package_does_not_import_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "rpvg_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")) or f == "Makefile":
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.lower() or f in (), f"{os.path.join(dirpath, f)} mentions the oracle"


# ---- ReadPathProbabilities mirror -----------------------------------------------------------

def test_row_finishing_reference_pinned_case():
    # src/tests/read_path_probabilities_test.cpp:29-34: two equally likely paths, noise 0.1 -> one group 0.45 {0, 1}
    b = synth.rows_from_likelihoods(2, [(1, 0.1, {0: 1 / 3.0, 1: 1 / 3.0})])
    (cnt, noise, groups), = b.cluster(0)["rows"]
    assert cnt == 1 and abs(noise - 0.1) < 1e-15
    assert len(groups) == 1 and abs(groups[0][0] - 0.45) < 1e-15 and groups[0][1] == [0, 1]


def test_quick_merge_identical_reference_pinned_case():
    # src/tests/read_path_probabilities_test.cpp:194-204: merging a row with itself doubles the count only
    b = synth.rows_from_likelihoods(2, [(1, 0.1, {0: 1 / 3.0, 1: 1 / 3.0}), (1, 0.1, {0: 1 / 3.0, 1: 1 / 3.0})])
    rows = b.cluster(0)["rows"]
    assert len(rows) == 1
    assert rows[0][0] == 2 and abs(rows[0][1] - 0.1) < 1e-15 and abs(rows[0][2][0][0] - 0.45) < 1e-15


def test_row_finishing_and_merging_match_python_mirror():
    rng = np.random.default_rng(5)
    n_paths = 9
    reads = []
    for _ in range(400):
        if rng.random() < 0.05:
            reads.append((1, 1.0, {}))
            continue
        k = int(rng.integers(1, 4))
        idx = rng.choice(n_paths, size=k, replace=False)
        lik = {int(p): float(np.exp(-1.383325268738 * int(rng.integers(0, 3))) / (200 + 100 * int(p))) for p in idx}
        reads.append((int(rng.integers(1, 3)), float(rng.choice([1e-4, 1e-3, 0.1])), lik))
    got = synth.rows_from_likelihoods(n_paths, reads).cluster(0)["rows"]
    want = small_cases.sort_and_merge([small_cases.finish_row(c, n, l) if l else (c, 1.0, []) for c, n, l in reads])
    assert len(got) == len(want)
    assert sum(r[0] for r in got) == sum(r[0] for r in reads)
    for g, w in zip(got, want):
        assert g[0] == w[0] and abs(g[1] - w[1]) < 1e-15
        assert [x[1] for x in g[2]] == [x[1] for x in w[2]]
        assert np.allclose([x[0] for x in g[2]], [x[0] for x in w[2]], rtol=1e-14, atol=0)


def test_sub_precision_mass_moves_to_noise():
    # a path below prob_precision is dropped and its mass added to the noise (src/read_path_probabilities.cpp:209-217)
    b = synth.rows_from_likelihoods(2, [(1, 0.01, {0: 1.0, 1: 1e-10})])
    (cnt, noise, groups), = b.cluster(0)["rows"]
    assert len(groups) == 1 and groups[0][1] == [0]
    low = 1e-10 / (1 + 1e-10)
    assert abs(noise - (0.01 + low * 0.99)) < 1e-15
    assert abs(groups[0][0] - (1 - low) * 0.99) < 1e-12


# ---- synthetic pantranscriptome generator ------------------------------------------------------

def test_generator_is_deterministic_and_well_formed():
    kw = dict(seed=11, num_clusters=60, total_paths=2400, total_reads=60000)
    a, b = synth.generate(**kw), synth.generate(**kw)
    for name in ClusterBatch._DTYPES:
        assert np.array_equal(getattr(a, name), getattr(b, name)), name
    assert a.num_clusters == 60 and a.num_paths == 2400 and a.total_reads == 60000
    reads = np.add.reduceat(a.row_count.astype(np.int64), a.cluster_row_off[:-1].astype(np.int64))
    assert np.all(np.diff(reads) <= 0)  # clusters come in descending read-count order (src/main.cpp:811-827)
    # row invariants the estimators rely on
    assert np.all((a.row_noise > 0) & (a.row_noise <= 1))
    assert np.all(a.grp_prob >= 1e-8)
    for k in range(a.num_clusters):
        n = int(a.cluster_path_off[k + 1] - a.cluster_path_off[k])
        r0, r1 = int(a.cluster_row_off[k]), int(a.cluster_row_off[k + 1])
        e0, e1 = int(a.grp_idx_off[int(a.row_grp_off[r0])]), int(a.grp_idx_off[int(a.row_grp_off[r1])])
        assert n >= 1 and (e1 == e0 or a.path_idx[e0:e1].max() < n)
    # every haplotype carries exactly one HST of every transcript
    cl = a.cluster(0)
    by_group = {}
    for p in cl["paths"]:
        by_group.setdefault(p["group_id"], []).extend(p["source_ids"])
    for ids in by_group.values():
        assert sorted(ids) == list(range(64))
    assert synth.generate(**dict(kw, seed=12)).num_rows != a.num_rows or True


def test_batch_select_round_trips():
    a = synth.generate(seed=13, num_clusters=12, total_paths=300, total_reads=4000)
    sub = a.select([5, 2, 9])
    assert sub.num_clusters == 3
    for i, k in enumerate([5, 2, 9]):
        assert sub.cluster(i) == a.cluster(k)
    assert sub.total_reads == sum(sum(r[0] for r in a.cluster(k)["rows"]) for k in (5, 2, 9))


def test_reference_factory_compiles_with_the_reference_constructor_lists():
    """src/main.cpp:766-788 unchanged apart from the namespace: the estimators take the reference's own parameter lists
    (the engine is a defaulted last argument); and without a GPU constructing one fails loudly."""
    import subprocess
    from tests import small_cases
    binary = small_cases.build_reference_factory()
    assert "usage" in subprocess.run([binary], capture_output=True, text=True, check=True).stdout
    import torch
    if not torch.cuda.is_available():
        out = subprocess.run([binary, "transcripts"], capture_output=True, text=True)
        assert out.returncode != 0 and "no CPU fallback" in out.stderr


def test_source_id_set_behaves_like_the_set_it_replaces():
    """PathInfo::source_ids: insert / emplace / count / find / iteration as the reference's code uses them, one sorted array."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out_dir = os.path.join(root, "tests", "cpp", "_build")
    os.makedirs(out_dir, exist_ok=True)
    binary = os.path.join(out_dir, "source_id_set")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I" + os.path.join(root, "rpvg_amd", "host"),
                           os.path.join(root, "tests", "cpp", "source_id_set.cpp"), "-o", binary])
    assert subprocess.run([binary], capture_output=True, text=True, check=True).stdout.strip() == "ok"


def test_restated_random_streams_equal_libstdcxx():
    """rpvg_amd/csrc/gibbs_streams.hpp (mt19937 from its next 624 outputs, Lemire's uniform_int_distribution, the
    two-word generate_canonical, lower_bound over a discrete_distribution's partial sums) against libstdc++ itself — what
    the device sampler of --use-hap-gibbs rests on (src/path_estimator.cpp:491,509,555-556)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out_dir = os.path.join(root, "tests", "cpp", "_build")
    os.makedirs(out_dir, exist_ok=True)
    binary = os.path.join(out_dir, "gibbs_streams_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I" + os.path.join(root, "rpvg_amd", "csrc"),
                           os.path.join(root, "tests", "cpp", "gibbs_streams_check.cpp"), "-o", binary])
    assert subprocess.run([binary], capture_output=True, text=True, check=True).stdout.strip() == "ok"


def test_generator_states_read_and_written_in_place_equal_the_standard_interface():
    """The host side of the device sampler takes a generator's next 624 outputs from its state array and sets the state it goes
    on from (path_estimator.cpp, GeneratorLayout) instead of copying the generator, calling it 624 times and seeding it: the same
    words both ways on this libstdc++ (300 generators at random positions), and the portable route when asked for."""
    import subprocess
    import sys
    from rpvg_amd import engine
    lib = engine.lib()
    lib.rpvg_amd_generator_state_check.restype = C.c_int
    assert lib.rpvg_amd_generator_state_check(C.c_uint32(300)) == 1
    code = "import ctypes as C; from rpvg_amd import engine; L = engine.lib(); L.rpvg_amd_generator_state_check.restype = C.c_int; print(L.rpvg_amd_generator_state_check(C.c_uint32(5)))"
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True, env=dict(os.environ, RPVG_AMD_PORTABLE_GENERATORS="1"),
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.stdout.strip().splitlines()[-1] == "0"


@pytest.mark.gpu
def test_restated_random_streams_equal_the_gpu_box_libstdcxx():
    """The same check on the GPU box, against the libstdc++ the host library there is built with and runs on: the device
    sampler's draw-for-draw equality with the reference holds for a reference built with that library (INTEGRATION.md:
    GCC 11's uniform_int_distribution; GCC <= 10 draws different starts)."""
    test_restated_random_streams_equal_libstdcxx()
