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
    """(Nearly) every row is a segment of its own: the partition tables are sized by the RECORDS, so they hold them all."""
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


def test_partitions_larger_than_their_table_overflow_into_further_rounds(monkeypatch):
    """One workgroup counts one hash-prefix partition in an LDS table of 2048 slots; partitions are sized for ~1024 records, so the
    table never fills.  EMSAR_HIP_COLLAPSE_PART_ROWS (test hook, clamped to [16, 2^20]: the partition count can never be 0 -- an
    unclamped table-size knob once produced a table of 0 slots and a wild probe address, DESIGN.md 'the 14:08 memory fault of round
    2') makes partitions of ~8192 records: keys that find no place within 64 probes of their home go to the overflow list and are
    hashed again with the next seed, next to long rows (> 8 ids, the listed path), short rows and singles.  Every output array is
    still the oracle's."""
    monkeypatch.setenv("EMSAR_HIP_COLLAPSE_PART_ROWS", "8192")
    rng = np.random.default_rng(7)
    s = synth.make_config("cfg5", 0.0005)                         # 100k reads of 20 ids on average, rows of 50-100 ids among them, many duplicates
    n_extra = 60000                                               # + rows that are nearly all segments of their own: 3 and 12 random ids of 2^20
    e3 = rng.integers(0, 2 ** 20, size=(n_extra // 2, 3)).astype(np.int32)
    e12 = rng.integers(0, 2 ** 20, size=(n_extra // 2, 12)).astype(np.int32)
    ci = np.concatenate([s["col_idx"], e3.ravel(), e12.ravel()]).astype(np.int32)
    lens = np.concatenate([np.diff(s["row_ptr"].astype(np.int64)), np.full(n_extra // 2, 3), np.full(n_extra // 2, 12)])
    perm = rng.permutation(len(lens))                             # interleave the three kinds of rows
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
    rp = np.zeros(len(lens) + 1, dtype=np.uint64)
    rp[1:] = np.cumsum(lens[perm])
    ci = np.concatenate([ci[starts[r]:starts[r] + lens[r]] for r in perm]).astype(np.int32)
    want = O.collapse_rows(rp, ci)
    with EmsarHip(0) as ctx:
        got = ctx.collapse_rows(2 ** 20, rp, ci)
        assert got[4].rounds > 2                                    # the overflow rounds happened
        for a, b in zip((got[0], got[1], got[2].astype(np.int64), got[3]), want):
            np.testing.assert_array_equal(a, b)
        monkeypatch.setenv("EMSAR_HIP_COLLAPSE_PART_ROWS", "0")        # out of range: ignored (the default applies)
        got = ctx.collapse_rows(2 ** 20, rp, ci)
        assert got[4].rounds == 1
        for a, b in zip((got[0], got[1], got[2].astype(np.int64), got[3]), want):
            np.testing.assert_array_equal(a, b)
