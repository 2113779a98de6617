"""CPU-only checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/emsar_hip.h declares, refuses to run without a device, and its host-side layout builder stores
exactly the rows it was given.  No compute call is made here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import emsar_amd
from emsar_amd import hip as H
from emsar_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from emsar_amd import _build
    _build.build_hip()
    return emsar_amd.load_library()


def test_header_symbols_are_exported(lib):
    text = open(os.path.join(ROOT, "include", "emsar_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = sorted(set(re.findall(r"\b(emsar_hip_[a-z_]+)\s*\(", text)))
    assert declared == sorted(H.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name


def test_struct_sizes_match_header(lib):
    # plain C layout, no padding surprises between the header and the ctypes mirror
    assert C.sizeof(H.EmParams) == 4 + 4 + 8 + 8 + 4 + 4 + 8 + 8 + 8 + 4 + 4
    assert C.sizeof(H.EmStats) == 4 + 4 + 8 * 4 + 8 * 2 + 4 * 4 + 8 * 3 + 4 + 4 + 8
    assert C.sizeof(H.SetsInfo) == 8 * 16
    assert C.sizeof(H.Info) == 8 * 2 + 4 * 2 + 8 * 4 + 4 * 2 + 8 * 2 + 8 * 3 + 4 * 2


def test_strerror(lib):
    assert lib.emsar_hip_strerror(0) == b"ok"
    for s in range(-6, 0):
        assert lib.emsar_hip_strerror(s) not in (b"ok", b"unknown status")


def test_no_silent_fallback_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(emsar_amd.EmsarHipError) as e:
        emsar_amd.EmsarHip(0)
    assert e.value.status == -2      # EMSAR_HIP_ERR_NO_DEVICE: the product path fails loudly, no CPU path


def test_null_and_malformed_arguments(lib):
    assert lib.emsar_hip_create(None, 0) == -1
    assert lib.emsar_hip_upload_structure(None, 0, 1, None, None, 0) == -1
    rp = np.array([0, 2, 1], dtype=np.uint64)           # not monotone
    ci = np.array([0, 1], dtype=np.int32)
    with pytest.raises(emsar_amd.EmsarHipError):
        emsar_amd.layout_selfcheck_tiled(4, rp, ci)
    rp = np.array([0, 2], dtype=np.uint64)
    for bad in ([0, 4], [-1, 0]):                        # tid outside [0, n_tx)
        with pytest.raises(emsar_amd.EmsarHipError):
            emsar_amd.layout_selfcheck_tiled(4, rp, np.array(bad, dtype=np.int32))


@pytest.mark.parametrize("block,frag,unit", [(None, None, None), ("128", None, "1"), (None, "3072", "4"), ("64", None, "3")])
def test_layout_roundtrip_synthetic(lib, block, frag, unit, monkeypatch):
    """The TILED builder on a matrix with cross-family reads: descriptors inside their arrays, decoded rows = input rows,
    for the default sort block, a small one, and with the rows cut into many independently tiled fragments."""
    if block:
        monkeypatch.setenv("EMSAR_HIP_TILE_BLOCK", block)
    if unit:
        monkeypatch.setenv("EMSAR_HIP_UNIT_TILES", unit)                 # tiles that share one dictionary (1 .. 4)
    if frag:
        monkeypatch.setenv("EMSAR_HIP_FRAG_ROWS", frag)
        monkeypatch.setenv("EMSAR_HOST_THREADS", "3")
    m = synth.make_matrix(n_tx=6000, n_reads=40000, law="human", xfam=0.05, seed=9)
    info = emsar_amd.layout_selfcheck_tiled(m["n_tx"], m["row_ptr"], m["col_idx"])
    nnz = len(m["col_idx"])
    assert info["nnz"] == nnz
    # an operand names a block of three neighbouring transcripts and a subset of it: fewer operands than hits where reads hit neighbours
    assert 0 < info["padded_entries"] and info["padded_entries"] % 768 == 0
    assert 0 < info["n_chunks"] <= info["n_slices"] <= 4 * info["n_chunks"]
    assert info["far_entries"] > 0                          # cross-family hits fall outside their tile's window


def test_layout_roundtrip_edge_cases(lib, monkeypatch):
    monkeypatch.setenv("EMSAR_HIP_RENUMBER", "0")               # the caller's numbering: the far-slot paths below are the point
    # empty matrix, empty rows, one row, one very long row, duplicate tids inside a row
    chk = emsar_amd.layout_selfcheck_tiled
    chk(5, np.array([0], dtype=np.uint64), np.array([], dtype=np.int32))
    chk(5, np.array([0, 0, 0], dtype=np.uint64), np.array([], dtype=np.int32))
    chk(5, np.array([0, 1], dtype=np.uint64), np.array([4], dtype=np.int32))
    chk(5, np.array([0, 0, 3, 3, 4], dtype=np.uint64), np.array([2, 4, 4, 0], dtype=np.int32))
    rng = np.random.default_rng(0)
    long_row = rng.integers(0, 3000, 700).astype(np.int32)      # 700 scattered tids: more than a tile's dictionary holds (240) -> leftover CSR
    rp = np.array([0, 700, 701], dtype=np.uint64)
    info = chk(3000, rp, np.append(long_row, 5).astype(np.int32))
    assert info["n_slices"] == 0 and info["folded_single_rows"] == 1
    wide_row = rng.integers(0, 3000, 200).astype(np.int32)      # 200 scattered tids: one tile whose dictionary is mostly far slots
    rp = np.array([0, 200, 201], dtype=np.uint64)
    info = chk(3000, rp, np.append(wide_row, 5).astype(np.int32))
    assert info["n_slices"] == 1 and info["far_entries"] > 100
    near_row = (1000 + rng.integers(0, 120, 200)).astype(np.int32)  # 200 hits inside a window of 120 tids: neighbours share an entry
    info = chk(3000, rp, np.append(near_row, 5).astype(np.int32))    # (an entry names a block of four tids and a subset of it)
    assert info["n_slices"] == 1 and info["padded_entries"] % 768 == 0 and info["padded_entries"] < 768 * 200


@pytest.mark.parametrize("merge", [False, True])
def test_tiled_layout_roundtrip(lib, golden, merge):
    m = golden.model
    info = emsar_amd.layout_selfcheck_tiled(m.n_tx, m.row_ptr, m.col_idx, merge)
    assert info["folded_single_rows"] >= m.n_tx                    # every transcript has a single-tid row in an rsh


def test_tiled_layout_roundtrip_synthetic(lib):
    for law, xfam in (("human", 0.05), ("repeats", 0.02), ("poisson2", 0.0)):
        m = synth.make_matrix(n_tx=6000, n_reads=60000, law=law, xfam=xfam, seed=3)
        info = emsar_amd.layout_selfcheck_tiled(m["n_tx"], m["row_ptr"], m["col_idx"])
        assert info["n_chunks"] >= 1
        merged = emsar_amd.layout_selfcheck_tiled(m["n_tx"], m["row_ptr"], m["col_idx"], True)
        assert merged["n_slices"] <= info["n_slices"] and merged["layout"] == 3 | 0x100
    # rows longer than a tile can hold, duplicates inside rows, empty rows
    rng = np.random.default_rng(1)
    rows = [rng.integers(0, 9000, size=3000), np.array([], dtype=np.int64), rng.integers(0, 9000, size=1025),
            np.array([7, 7, 7]), np.array([8999]), np.array([0, 8999])]
    rp = np.zeros(len(rows) + 1, dtype=np.uint64)
    rp[1:] = np.cumsum([len(r) for r in rows])
    info = emsar_amd.layout_selfcheck_tiled(9000, rp, np.concatenate(rows).astype(np.int32))
    assert info["folded_single_rows"] == 1
    # merging: the same multiset in any order is one stored row; repeated tids are part of the multiset
    rows = [np.array([3, 1, 2]), np.array([2, 3, 1]), np.array([1, 2, 3]), np.array([1, 2]), np.array([5]), np.array([5]),
            np.array([1, 1, 2]), np.array([2, 1, 1]), np.array([1, 2, 2])]
    rp = np.zeros(len(rows) + 1, dtype=np.uint64)
    rp[1:] = np.cumsum([len(r) for r in rows])
    emsar_amd.layout_selfcheck_tiled(8, rp, np.concatenate(rows).astype(np.int32), True)


@pytest.mark.parametrize("structure", synth.STRUCTURES)
@pytest.mark.parametrize("mode", ["0", "1", "2"])      # never / decided by the sampled rows / always
def test_renumbered_layout_roundtrip(lib, monkeypatch, structure, mode):
    # the library's own transcript numbering (csrc/renumber.hpp): whatever it decides, the stored layout decodes to the caller's rows
    # mapped through a permutation (check_tiled), with and without merged rows and with several builder fragments
    monkeypatch.setenv("EMSAR_HIP_RENUMBER", mode)
    monkeypatch.setenv("EMSAR_HIP_FRAG_ROWS", "6144")
    monkeypatch.setenv("EMSAR_HOST_THREADS", "3")
    m = synth.make_matrix(n_tx=6000, n_reads=120000, law="human", xfam=0.03, seed=17, structure=structure)
    info = emsar_amd.layout_selfcheck_tiled(m["n_tx"], m["row_ptr"], m["col_idx"])
    merged = emsar_amd.layout_selfcheck_tiled(m["n_tx"], m["row_ptr"], m["col_idx"], True)
    assert info["renumbered"] == merged["renumbered"] == {"0": 0, "2": 1}.get(mode, info["renumbered"])
    assert info["tiled_ids"] >= info["tiled_entries"] > 0


def test_renumbering_recovers_shuffled_families(lib, monkeypatch):
    # SURVEY 8d's family law with the transcripts numbered at random: the caller's tid order carries no information, a row needs one
    # operand per hit; numbered by co-occurrence the same matrix packs as well as the unshuffled one
    a = synth.make_matrix(n_tx=8000, n_reads=400000, law="human", xfam=0.02, seed=5, structure="family")
    b = synth.make_matrix(n_tx=8000, n_reads=400000, law="human", xfam=0.02, seed=5, structure="family_shuffled")
    per = {}
    for name, m, mode in (("family", a, "1"), ("shuffled_off", b, "0"), ("shuffled", b, "1")):
        monkeypatch.setenv("EMSAR_HIP_RENUMBER", mode)
        i = emsar_amd.layout_selfcheck_tiled(m["n_tx"], m["row_ptr"], m["col_idx"])
        per[name] = (i["tiled_ids"] / i["tiled_entries"], i["renumbered"], i["far_entries"])
    assert per["shuffled_off"][0] < 1.05 and per["shuffled_off"][1] == 0
    assert per["shuffled"][1] == 1 and per["shuffled"][0] > 0.95 * per["family"][0]
    assert per["shuffled"][2] < 2 * per["family"][2] + 1000          # and the cross-family hits are the only far ones again


def test_numbering_does_not_depend_on_the_thread_count(lib, monkeypatch):
    # the pair counts of csrc/renumber.hpp are made by several host threads (keys split by hash class): same clusters, same layout
    m = synth.make_matrix(n_tx=8000, n_reads=300000, law="human", xfam=0.02, seed=23, structure="family_shuffled")
    monkeypatch.setenv("EMSAR_HIP_RENUMBER", "2")
    seen = set()
    for threads in ("1", "2", "5", "16"):
        monkeypatch.setenv("EMSAR_HOST_THREADS", threads)
        i = emsar_amd.layout_selfcheck_tiled(m["n_tx"], m["row_ptr"], m["col_idx"])
        seen.add((i["tiled_entries"], i["far_entries"], i["n_units"], i["padded_entries"], i["stored_bytes_per_pass"]))
    assert len(seen) == 1, seen
