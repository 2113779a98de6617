"""CPU: the host-side builder of the set-resident solver's records (emsar_amd/csrc/sets.hpp) through the C ABI's
host-only self-check -- connected sets of the rows with reads, closed-form transcripts, merging of identical rows,
the LDS classes, and what is left to the streaming passes."""
import numpy as np
import pytest

from emsar_amd.hip import EmsarHipError, sets_selfcheck
from emsar_amd.synth import family_matrix


def test_toy_split():
    # rows: {0,1} R=3 | {1,2} R=0 (does not couple) | {2} R=5 | {3,3} R=2 | {4,5} R=1 | {5,4} R=1 (same set of tids)
    rp = np.array([0, 2, 4, 5, 7, 9, 11], dtype=np.uint64)
    ci = np.array([0, 1, 1, 2, 2, 3, 3, 4, 5, 5, 4], dtype=np.int32)
    w = np.array([3, 0, 5, 2, 1, 1], dtype=np.int32)
    d = sets_selfcheck(7, rp, ci, w)
    assert d["n_components"] == 2                      # {0,1} and {4,5}; tid 2 is not linked through the R=0 row
    assert d["sets_resident"] == [2, 0, 0] and d["sets_streamed"] == 0
    assert d["tids_resident"] == 4 and d["tids_closed"] == 3 and d["tids_streamed"] == 0
    assert d["rows_in"] == 3 and d["rows_stored"] == 2   # {4,5} and {5,4} are one stored row of weight 2
    d = sets_selfcheck(7, rp, ci, None)                  # every row counts 1: {1,2} now links 0-1-2
    assert d["n_components"] == 2 and d["tids_resident"] == 5


def test_classes_clusters_and_streamed_sets(monkeypatch):
    n_tx, rp, ci, w = family_matrix([2, 3, 5, 8, 40, 200, 900, 2500, 6000], seed=3)
    monkeypatch.setenv("EMSAR_HIP_CLUSTER", "1")                             # the cluster solver is opt-in (slower per pass than streaming)
    d = sets_selfcheck(n_tx, rp, ci, w)
    assert d["tids_closed"] + d["tids_resident"] + d["tids_cluster"] + d["tids_streamed"] == n_tx
    # 8 vectors of 2500 or 6000 doubles do not fit one workgroup's 156 KiB: those two families go to clusters of workgroups
    assert d["sets_cluster"] == 2 and d["tids_cluster"] >= 8000 and d["sets_streamed"] == 0
    assert 0 < d["max_lds_cluster"] <= 156 * 1024
    assert d["sets_resident"][0] >= 3 and d["sets_resident"][1] >= 1 and d["sets_resident"][2] >= 1
    assert d["max_lds_bytes"][0] <= 6 * 1024 and d["max_lds_bytes"][1] <= 48 * 1024 and d["max_lds_bytes"][2] <= 156 * 1024
    assert d["rows_stored"] < d["rows_in"]                                   # duplicates merged
    monkeypatch.delenv("EMSAR_HIP_CLUSTER")                                  # default: they are streamed, as in round 1
    d = sets_selfcheck(n_tx, rp, ci, w)
    assert d["sets_cluster"] == 0 and d["sets_streamed"] == 2 and d["tids_streamed"] >= 8000
    # a component beyond a cluster's reach (20 000 transcripts: the whole point no longer fits a workgroup's LDS) is still streamed
    monkeypatch.setenv("EMSAR_HIP_CLUSTER", "1")
    n_tx, rp, ci, w = family_matrix([20000, 3000], seed=4)
    d = sets_selfcheck(n_tx, rp, ci, w)
    assert d["sets_streamed"] == 1 and d["tids_streamed"] >= 19000 and d["sets_cluster"] == 1


def test_no_reads_and_empty():
    n_tx, rp, ci, w = family_matrix([4, 4], seed=1)
    d = sets_selfcheck(n_tx, rp, ci, np.zeros_like(w))
    assert d["n_components"] == 0 and d["tids_closed"] == n_tx
    d = sets_selfcheck(3, np.zeros(1, dtype=np.uint64), np.zeros(0, dtype=np.int32), None)
    assert d["tids_closed"] == 3
    with pytest.raises(EmsarHipError):
        sets_selfcheck(n_tx, rp, ci, -np.ones_like(w))


@pytest.mark.parametrize("seed", range(6))
def test_random_block_matrices(seed):
    rng = np.random.default_rng(100 + seed)
    sizes = list(rng.integers(1, 60, size=200)) + [int(rng.integers(300, 1500))]
    n_tx, rp, ci, w = family_matrix(sizes, rows_per_tid=int(rng.integers(1, 6)), seed=seed)
    d = sets_selfcheck(n_tx, rp, ci, w if seed % 2 else None)
    assert d["tids_closed"] + d["tids_resident"] + d["tids_cluster"] + d["tids_streamed"] == n_tx
    assert d["n_components"] == sum(d["sets_resident"]) + d["sets_cluster"] + d["sets_streamed"]


def test_giant_component_is_detected_early():
    # a read-level matrix whose cross-family reads tie everything together: the builder stops after ~1M rows and
    # leaves the whole problem to the streaming passes
    from emsar_amd import synth
    s = synth.make_matrix(n_tx=6000, n_reads=1200000, law="human", xfam=0.02, seed=3)
    d = sets_selfcheck(s["n_tx"], s["row_ptr"], s["col_idx"])
    assert d["sets_streamed"] == 1 and d["tids_streamed"] == s["n_tx"] and sum(d["sets_resident"]) == 0
