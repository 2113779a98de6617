"""CPU: the C host (emsar_amd/csrc/host) against the compiled reference's files in tests/golden/.

Covers the drop-in boundary on the input and output side: rsh reader, bowtie / SAM text
readers for SE and PE with the per-read filters, fragment-length weights, segment effective lengths, connected
sets, and the three writers.  Deterministic outputs must be BYTE-identical to the reference's.
"""
import glob
import os

import numpy as np
import pytest

import oracle as O
from emsar_amd import _build, hostlib as HL
from tests.conftest import aln_path

HALF_QUANTUM = 5.01e-7


@pytest.fixture(scope="module", autouse=True)
def _built():
    _build.build_host()


def _count(golden, r, **kw):
    opts = golden.meta["opts"]
    k = int(opts[opts.index("-k") + 1]) if "-k" in opts else 100
    aln, fmt = aln_path(golden.dir)
    assert fmt == (2 if "-B" in opts else 1 if "-S" in opts else 0)
    return r.count(aln, pe=int("-P" in opts), fmt=fmt, max_repeat=k, **kw)


def _host(golden):
    r = HL.HostRsh(os.path.join(golden.dir, "index.rsh"))
    c = _count(golden, r)
    return r, c, r.model(c)


def test_rsh_reader_matches_python_restatement(golden):
    r = HL.HostRsh(os.path.join(golden.dir, "index.rsh"))
    pr = golden.rsh
    assert r.names == pr["names"]
    assert r.n_rows == len(pr["rows"])
    m = golden.model
    assert (r.row_ptr == m.row_ptr).all() and (r.col_idx == m.col_idx).all()
    for c, (tids, euma) in enumerate(pr["rows"]):
        if euma is None:
            assert r.has_node[c] == 0 and (r.euma[c] == 0).all()
        else:
            assert r.has_node[c] == 1 and list(r.euma[c]) == euma[: r.nfl]
    assert r.tid_of(pr["names"][-1]) == len(pr["names"]) - 1 and r.tid_of("no_such_transcript") == -1
    # segment lookup: multisets, order-free, repeated tids kept
    for c in range(r.n_tx, r.n_rows, max(1, (r.n_rows - r.n_tx) // 50)):
        tids = list(m.col_idx[int(m.row_ptr[c]):int(m.row_ptr[c + 1])])
        assert r.row_of(tids[::-1]) == c
    assert r.row_of([0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0]) == -1


def test_read_collapse_matches_reference_counts(golden):
    r, c, m = _host(golden)
    assert (c.R == golden.seg.R).all()                                    # column 6 of .segments
    assert c.total_reads == golden.N
    assert (c.frag_counts[golden.frag_lens] == golden.frag_counts).all()    # column 2 of .fraglength_effect
    if golden.meta["total_read_count"] is not None:
        assert c.total_reads == golden.meta["total_read_count"]


def test_model_matches_reference(golden):
    r, c, m = _host(golden)
    assert np.abs(m.L - golden.seg.L).max() <= HALF_QUANTUM               # eff.length column of .segments
    np.testing.assert_array_equal(m.L, golden.model.L)                     # bit-identical to the python restatement
    np.testing.assert_array_equal(m.E, golden.model.E)
    assert (m.CS == golden.cs_ref).all() and m.eumacut == 0.0
    assert (m.E_solver == m.E).all()                                        # nothing dropped when EUMAcut stays 0


def test_fraglength_file_is_byte_identical(golden, tmp_path):
    r, c, m = _host(golden)
    out = str(tmp_path / "x.fraglength_effect")
    r.write_fraglength(out, c, m)
    assert open(out).read() == open(os.path.join(golden.dir, "ref.run0.fraglength_effect")).read()


def test_fpkm_file_is_byte_identical_for_the_seeded_run(golden, tmp_path):
    """Feed the writer the rounds of the seeded pattern search (the oracle reproduces the reference's
    rand() stream): the resulting .fpkm must equal the reference's seeded output byte for byte."""
    r, c, m = _host(golden)
    om = golden.model
    n, cs, _, _ = om.components()
    rounds = []
    for k in range(4):
        th, _ = om.mle_pattern_search(cs, n, seed=golden.meta["seed"] if k == 0 else 0, n_threads=1)
        rounds.append(th)
    mean, sd = HL.mean_sd(np.array(rounds))
    ie = om.ieuma()
    _, _, ir, iri, tpm = O.fpkm_table(np.array(rounds), ie, golden.N)
    out = str(tmp_path / "x.fpkm")
    r.write_fpkm(out, mean, sd, ie, ir, iri, tpm)
    assert open(out).read() == open(os.path.join(golden.dir, "ref.seed%d.fpkm" % golden.meta["seed"])).read()


def test_segments_file_matches_reference(golden, tmp_path):
    r, c, m = _host(golden)
    out = str(tmp_path / "x.segments")
    r.write_segments(out, c, m, golden.runs[0]["fpkm"])
    got = open(out).read().splitlines()
    ref = open(os.path.join(golden.dir, "ref.run0.segments")).read().splitlines()
    assert len(got) == len(ref) and got[0] == ref[0]
    for a, b in zip(got[1:], ref[1:]):
        fa, fb = a.split("\t"), b.split("\t")
        assert fa[:6] == fb[:6]                                             # ids, names, eff.length, Readcount: exact
        # expected count is computed from the 6-decimal FPKM we read back, the reference used its unrounded mean
        assert abs(float(fa[6]) - float(fb[6])) <= 1e-5 * abs(float(fb[6])) + 2e-3


def test_eumacut_loop_drops_weak_links_of_oversized_sets(tmp_path):
    """A set with more than 5000 transcripts raises EUMAcut by 2 until it splits (emsar_main.c:417-423);
    multi-tid rows below the cut leave the likelihood (E_solver 0) but keep their effective length."""
    n = 5003
    lines = ["#%d,2,30,30,-1" % (n - 1)] + ["@%d\tt%d" % (i, i) for i in range(n)]
    lines.append("cid\tno.tids\tfirst.tid\tother.tids\tsegment.length")
    for i in range(n):
        lines.append("%d\t1\t%d\t\t100," % (i, i))
    for i in range(n - 1):                       # a chain: one weak link (EUMA 1) in the middle, strong links elsewhere
        lines.append("%d\t2\t%d\t%d,\t%d," % (n + i, i, i + 1, 1 if i == 2500 else 50))
    p = tmp_path / "chain.rsh"
    p.write_text("\n".join(lines) + "\n")
    aln = tmp_path / "r.bowtie"
    aln.write_text("".join("r%d\t+\tt%d\t1\t%s\t%s\t0\t\n" % (i, i % n, "A" * 30, "I" * 30) for i in range(200)))
    r = HL.HostRsh(str(p))
    c = r.count(str(aln))
    m = r.model(c)
    assert m.eumacut == 2.0 and m.n_sets == 2
    weak = n + 2500
    assert m.CS[weak] == -1 and m.E_solver[weak] == 0.0 and m.E[weak] > 0 and m.L[weak] == 1.0
    assert (np.delete(m.CS, weak) >= 0).all()
    om = O.Csr(n, r.row_ptr, r.col_idx, L=m.L)
    n_sets, cs, ts, cut = om.components()
    assert cut == 2.0 and n_sets == 2 and (cs == m.CS).all() and (ts == m.TS).all()


def test_reader_error_paths(tmp_path):
    with pytest.raises(HL.HostError):
        HL.HostRsh(str(tmp_path / "missing.rsh"))
    bad = tmp_path / "bad.rsh"
    bad.write_text("@0\tt0\n")
    with pytest.raises(HL.HostError):
        HL.HostRsh(str(bad))
    ok = tmp_path / "ok.rsh"
    ok.write_text("#0,1,30,30,-1\n@0\tt0\ncid\tx\n0\t1\t0\t\t10,\n")
    r = HL.HostRsh(str(ok))
    empty = tmp_path / "empty.bowtie"
    empty.write_text("")
    with pytest.raises(HL.HostError):                      # the reference aborts on a NULL alignment list
        r.count(str(empty))
    unk = tmp_path / "unk.bowtie"
    unk.write_text("r0\t+\tnope\t1\t%s\t%s\t0\t\n" % ("A" * 30, "I" * 30))
    with pytest.raises(HL.HostError):                      # unknown transcript name
        r.count(str(unk))
    with pytest.raises(HL.HostError):
        r.count(str(unk), strand="bogus")
    notbam = tmp_path / "x.bam"
    notbam.write_bytes(b"this is not a BAM file")
    with pytest.raises(HL.HostError):
        r.count(str(notbam), fmt=2)


def test_model_with_precomputed_L_is_the_same_model(golden):
    """The split the CLI uses: Wf on the host, L = EUMA . Wf elsewhere (emsar_hip_adj_euma), the rest from L."""
    r, c, m = _host(golden)
    wf = r.wf(c)
    np.testing.assert_array_equal(wf, m.Wf)
    L = np.zeros(r.n_rows)
    for cid in range(r.n_rows):                                   # the reference's loop: i ascending, multiply then add
        a = 0.0
        for i in range(r.nfl):
            a += wf[i] * float(r.euma[cid, i])
        L[cid] = a
    np.testing.assert_array_equal(L, m.L)
    m2 = r.model(c, L=L)
    for k in ("L", "E", "E_solver", "CS", "TS", "Wf"):
        np.testing.assert_array_equal(getattr(m2, k), getattr(m, k))
    assert m2.n_sets == m.n_sets


def test_rsh_binary_cache_round_trip(golden, tmp_path):
    src = os.path.join(golden.dir, "index.rsh")
    a = HL.HostRsh(src)
    cache = str(tmp_path / "index.rsh.bin")
    a.write_cache(cache)
    b = HL.HostRsh(src, cache=cache)
    assert (a.n_tx, a.n_rows, a.nfl, a.frag_min, a.frag_max) == (b.n_tx, b.n_rows, b.nfl, b.frag_min, b.frag_max)
    assert a.names == b.names
    for k in ("row_ptr", "col_idx", "euma", "has_node"):
        np.testing.assert_array_equal(getattr(a, k), getattr(b, k))
    # the rebuilt lookup tables answer like the parsed ones, and the sample pipeline gives the same model
    for t in (0, a.n_tx // 2, a.n_tx - 1):
        assert b.tid_of(a.names[t]) == t
    opts = golden.meta["opts"]
    aln, fmt = aln_path(golden.dir)
    k = int(opts[opts.index("-k") + 1]) if "-k" in opts else 100
    ca, cb = a.count(aln, pe=int("-P" in opts), fmt=fmt, max_repeat=k), b.count(aln, pe=int("-P" in opts), fmt=fmt, max_repeat=k)
    np.testing.assert_array_equal(ca.R, cb.R)
    assert ca.total_reads == cb.total_reads


def test_rsh_binary_cache_refuses_bad_files(tmp_path):
    import shutil
    case = os.path.join(os.path.dirname(__file__), "golden", "toy5_se50")
    src = str(tmp_path / "index.rsh")
    shutil.copy(os.path.join(case, "index.rsh"), src)
    a = HL.HostRsh(src)
    cache = str(tmp_path / "c.bin")
    a.write_cache(cache)
    HL.HostRsh(src, cache=cache)
    raw = open(cache, "rb").read()
    # truncated
    open(str(tmp_path / "t.bin"), "wb").write(raw[: len(raw) - 5])
    with pytest.raises(HL.HostError):
        HL.HostRsh(src, cache=str(tmp_path / "t.bin"))
    # foreign
    open(str(tmp_path / "f.bin"), "wb").write(b"not a cache at all" * 10)
    with pytest.raises(HL.HostError):
        HL.HostRsh(src, cache=str(tmp_path / "f.bin"))
    # a tid out of range inside an otherwise well-formed file
    import struct
    hdr = 8 + 4 + 4 + 8 * 4 + 5 * 8
    names_bytes = struct.unpack_from("<q", raw, 8 + 8 + 32 + 16)[0]
    off = hdr + names_bytes + 8 * (a.n_rows + 1)
    bad = bytearray(raw)
    struct.pack_into("<i", bad, off, 10 ** 6)
    open(str(tmp_path / "o.bin"), "wb").write(bytes(bad))
    with pytest.raises(HL.HostError):
        HL.HostRsh(src, cache=str(tmp_path / "o.bin"))
    # stale: the text changed after the cache was written
    with open(src, "a") as f:
        f.write("\n")
    with pytest.raises(HL.HostError):
        HL.HostRsh(src, cache=cache)
    HL.HostRsh(src, cache=cache, check_source=False)             # explicit opt-out of the staleness check


def test_line_reader_edge_cases(tmp_path):
    """The chunked line reader behind every text input: empty lines, CRLF, a line longer than its 4 MiB buffer, a last
    line without newline, plain and gzip."""
    import ctypes as C
    import gzip
    lib = HL.lib()
    lib.emsar_lr_open.restype = C.c_void_p
    lib.emsar_lr_open.argtypes = [C.c_char_p]
    lib.emsar_lr_next.restype = C.c_char_p
    lib.emsar_lr_next.argtypes = [C.c_void_p]
    lib.emsar_lr_close.argtypes = [C.c_void_p]
    lines = ["first", "", "crlf\r", "x" * (9 << 20), "tab\tsep", "", "last without newline"]
    raw = ("\n".join(lines)).encode()
    want = [ln.rstrip("\r").encode() for ln in lines]
    for name, opener in (("plain.txt", open), ("packed.gz", gzip.open)):
        p = str(tmp_path / name)
        with opener(p, "wb") as f:
            f.write(raw)
        h = lib.emsar_lr_open(p.encode())
        assert h
        got = []
        while True:
            ln = lib.emsar_lr_next(h)
            if ln is None:
                break
            got.append(ln)
        lib.emsar_lr_close(h)
        assert got == want
    p = str(tmp_path / "empty.txt")
    open(p, "wb").close()
    h = lib.emsar_lr_open(p.encode())
    assert lib.emsar_lr_next(h) is None
    lib.emsar_lr_close(h)


@pytest.mark.parametrize("batch_rows", [0, 7])
def test_collapse_hook_gives_the_per_read_counts(golden, batch_rows):
    """emsar_aln_opts.collapse: the kept reads with two or more transcripts are gathered as read-level rows and counted through
    a collapse function (on the GPU: emsar_hip_collapse_rows, emsar-hip --device-collapse) instead of one rsh lookup per read.
    Here the function is the oracle's restatement of update_ReadCounts' merge; batches of 7 rows exercise the flushes in the
    middle of a file.  Every count must equal the per-read path's -- on every fixture, all input formats, threads on."""
    import oracle as O
    r, c, m = _host(golden)
    calls = []

    def collapse(rp, ci):
        calls.append(len(rp) - 1)
        a, b, w, _ = O.collapse_rows(rp, ci)
        return a, b, w

    c2 = _count(golden, r, collapse=collapse, collapse_batch_rows=batch_rows)
    assert (c2.R == c.R).all() and c2.total_reads == c.total_reads
    assert (c2.frag_counts == c.frag_counts).all()
    assert c2.stats == c.stats
    if (c.R[r.n_tx:] > 0).any():
        assert calls and (batch_rows == 0 or max(calls) <= batch_rows)
