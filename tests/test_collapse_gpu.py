"""GPU: read -> segment collapse on the device (emsar_hip_collapse_rows) against the oracle's restatement of
update_ReadCounts (oracle_collapse_rows).  Integer work: every output array must be IDENTICAL."""
import numpy as np
import pytest

import oracle as O
from emsar_amd import EmsarHip, EmsarHipError, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    ctx = EmsarHip(0)
    yield ctx
    ctx.close()


def _same(dev, n_tx, rp, ci, w=None):
    want = O.collapse_rows(rp, ci, w)
    got = dev.collapse_rows(n_tx, rp, ci, w)
    np.testing.assert_array_equal(got[0], want[0])
    np.testing.assert_array_equal(got[1], want[1])
    np.testing.assert_array_equal(got[2].astype(np.int64), want[2])
    np.testing.assert_array_equal(got[3], want[3])
    return got




def test_toy_semantics(dev):
    # rows: {3,1,2} {2,3,1} {1,2} {} {1,1,2} {2,1,1} {1,2,2} {7} {7} ; weights with a zero
    rows = [[3, 1, 2], [2, 3, 1], [1, 2], [], [1, 1, 2], [2, 1, 1], [1, 2, 2], [7], [7]]
    rp = np.zeros(len(rows) + 1, dtype=np.uint64)
    rp[1:] = np.cumsum([len(r) for r in rows])
    ci = np.array([t for r in rows for t in r], dtype=np.int32)
    got = _same(dev, 8, rp, ci)
    assert [list(got[1][int(got[0][i]):int(got[0][i + 1])]) for i in range(len(got[2]))] == [[1, 2, 3], [1, 2], [1, 1, 2], [1, 2, 2], [7]]
    assert list(got[2]) == [2, 1, 2, 1, 2] and list(got[3]) == [0, 0, 1, -1, 2, 2, 3, 4, 4]
    w = np.array([5, 0, 2, 9, 1, 1, 0, 3, 4], dtype=np.int32)
    got = _same(dev, 8, rp, ci, w)
    assert list(got[2]) == [5, 2, 2, 7] and list(got[3]) == [0, -1, 1, -1, 2, 2, -1, 3, 3]


@pytest.mark.parametrize("name,scale", [("cfg2", 0.02), ("cfg3", 0.01), ("cfg5", 0.0005)])
def test_synthetic_configs(dev, name, scale):
    s = synth.make_config(name, scale)
    got = _same(dev, s["n_tx"], s["row_ptr"], s["col_idx"])
    assert got[2].sum() == s["n_reads"] and got[4].n_unique == len(got[2])
    rng = np.random.default_rng(1)
    w = rng.integers(0, 4, size=s["n_reads"]).astype(np.int32)
    got = _same(dev, s["n_tx"], s["row_ptr"], s["col_idx"], w)
    assert got[2].sum() == w.sum()


def test_collapsed_matrix_gives_the_same_em(dev):
    s = synth.make_matrix(n_tx=3000, n_reads=200000, law="human", xfam=0.02, seed=21)
    rp, ci, w, _, st = dev.collapse_rows(s["n_tx"], s["row_ptr"], s["col_idx"])
    assert st.n_unique < 0.5 * s["n_reads"]
    dev.upload_structure(s["n_tx"], s["row_ptr"], s["col_idx"])
    dev.upload_sample(None, None, s["den"])
    dev.run_passes(20)
    a = dev.get_theta()
    dev.upload_structure(s["n_tx"], rp, ci)
    dev.upload_sample(w, None, s["den"])
    dev.run_passes(20)
    b = dev.get_theta()
    assert np.all(np.abs(a - b) <= 1e-10 * np.abs(a) + 1e-300)


def test_edge_cases_and_errors(dev):
    e = np.zeros(1, dtype=np.uint64)
    got = dev.collapse_rows(4, e, np.zeros(0, dtype=np.int32))
    assert len(got[2]) == 0 and list(got[0]) == [0]
    rp = np.array([0, 0, 0], dtype=np.uint64)                      # only empty rows
    got = dev.collapse_rows(4, rp, np.zeros(0, dtype=np.int32))
    assert len(got[2]) == 0 and list(got[3]) == [-1, -1]
    # one long row next to many copies of it in another order, and 64-bit sums that do not fit the output
    long_row = np.arange(90, dtype=np.int32)
    rows = [long_row, long_row[::-1], np.roll(long_row, 7)]
    rp = np.arange(0, 91 * 3, 90, dtype=np.uint64)[:4]
    got = _same(dev, 100, rp, np.concatenate(rows).astype(np.int32))
    assert list(got[2]) == [3]
    with pytest.raises(EmsarHipError):
        dev.collapse_rows(100, rp, np.concatenate(rows).astype(np.int32), np.array([2 ** 31 - 1, 5, 0], dtype=np.int32))
    with pytest.raises(EmsarHipError):
        dev.collapse_rows(100, rp, np.concatenate(rows).astype(np.int32), np.array([-1, 5, 0], dtype=np.int32))
    with pytest.raises(EmsarHipError):
        dev.collapse_rows(50, rp, np.concatenate(rows).astype(np.int32))       # tid out of range


def test_mostly_unique_rows_outgrow_the_optimistic_table(dev):
    """The table is sized for one segment per four reads first; when (nearly) every row is a segment of its own the probe
    chains overrun, the call starts over with the worst-case table, and the result is the same."""
    rng = np.random.default_rng(5)
    n = 60000
    ci = rng.integers(0, 2 ** 20, size=3 * n).astype(np.int32)           # 3 random ids of a million per row: no two rows alike
    rp = np.arange(0, 3 * n + 1, 3, dtype=np.uint64)
    got = _same(dev, 2 ** 20, rp, ci)
    assert got[4].n_unique > 0.99 * n and got[4].table_slots >= n


def test_full_hash_collisions_are_resolved_by_comparison(monkeypatch):
    """Test hook EMSAR_HIP_COLLAPSE_WEAK_HASH: every row of one length gets the same two hashes and the same table tag, so
    each insert walks a probe chain of unrelated rows and must tell them apart by comparing the id multisets."""
    monkeypatch.setenv("EMSAR_HIP_COLLAPSE_WEAK_HASH", "1")
    s = synth.make_matrix(n_tx=300, n_reads=6000, law="human", xfam=0.05, seed=11)
    with EmsarHip(0) as ctx:
        got = ctx.collapse_rows(s["n_tx"], s["row_ptr"], s["col_idx"])
    want = O.collapse_rows(s["row_ptr"], s["col_idx"])
    for a, b in zip((got[0], got[1], got[2].astype(np.int64), got[3]), want):
        np.testing.assert_array_equal(a, b)
    assert len(got[2]) > 200                                       # many distinct multisets of equal length shared one chain


def test_table_forced_too_small_with_long_rows_restarts(monkeypatch):
    """The first table is sized by EMSAR_HIP_COLLAPSE_SHIFT (one slot per 2^shift rows; the knob is clamped to 0..8 and the table to
    1024 slots: an unclamped value of 16 once made a table of 0 slots, i.e. a probe mask of all ones and a wild address -- DESIGN.md,
    'the 14:08 memory fault of round 2').  At the largest shift on a matrix whose distinct segments outnumber the slots many times over,
    with long rows (> 8 ids, the listed path) and short ones mixed, probe chains overrun in both insert kernels, the call starts over
    with the worst-case table, and every output array is still the oracle's."""
    monkeypatch.setenv("EMSAR_HIP_COLLAPSE_SHIFT", "8")
    s = synth.make_config("cfg5", 0.002)                          # 400k reads, 20 ids per read on average, rows of 50-100 ids among them
    with EmsarHip(0) as ctx:
        got = ctx.collapse_rows(s["n_tx"], s["row_ptr"], s["col_idx"])
        assert got[4].table_slots >= s["n_reads"]                 # the restart happened: the worst-case table went through
        want = O.collapse_rows(s["row_ptr"], s["col_idx"])
        for a, b in zip((got[0], got[1], got[2].astype(np.int64), got[3]), want):
            np.testing.assert_array_equal(a, b)
        monkeypatch.setenv("EMSAR_HIP_COLLAPSE_SHIFT", "16")       # out of range: ignored (the default applies), never a table of 0 slots
        got = ctx.collapse_rows(s["n_tx"], s["row_ptr"], s["col_idx"])
        for a, b in zip((got[0], got[1], got[2].astype(np.int64), got[3]), want):
            np.testing.assert_array_equal(a, b)
