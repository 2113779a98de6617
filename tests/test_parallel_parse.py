"""CPU: alignment text counted in byte ranges by several threads (emsar_count_alignments) must give exactly the counts
of one sequential pass -- for every fixture in a text format and for files built to hit the seams: a read group much
longer than a range, filtered records sitting on the boundaries, pairs cut between their mates."""
import gzip
import os
import random
import sys

import numpy as np
import pytest

from emsar_amd import _build, hostlib as HL
from tests.conftest import CASES, aln_path, get_fixture

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_golden as G


@pytest.fixture(scope="module", autouse=True)
def _built():
    _build.build_host()


def _count(r, path, monkeypatch, threads, range_bytes, **kw):
    monkeypatch.setenv("EMSAR_HOST_THREADS", str(threads))
    monkeypatch.setenv("EMSAR_HOST_RANGE_BYTES", str(range_bytes))
    return r.count(path, **kw)


def _same(a, b):
    np.testing.assert_array_equal(a.R, b.R)
    np.testing.assert_array_equal(a.frag_counts, b.frag_counts)
    assert a.total_reads == b.total_reads and a.stats == b.stats


@pytest.mark.parametrize("case", [c for c in CASES if "bam" not in c])
def test_fixtures_in_ranges(case, tmp_path, monkeypatch):
    fx = get_fixture(case)
    src, fmt = aln_path(fx.dir)
    plain = str(tmp_path / "reads.txt")
    with gzip.open(src, "rb") as f, open(plain, "wb") as g:
        g.write(f.read())
    opts = fx.meta["opts"]
    kw = dict(pe=int("-P" in opts), fmt=fmt, max_repeat=int(opts[opts.index("-k") + 1]) if "-k" in opts else 100)
    r = HL.HostRsh(os.path.join(fx.dir, "index.rsh"))
    want = _count(r, plain, monkeypatch, 1, 1 << 40, **kw)
    size = os.path.getsize(plain)
    for threads, rb in ((2, size // 2), (3, size // 3), (7, size // 7), (16, max(64, size // 16)), (64, 97)):
        _same(_count(r, plain, monkeypatch, threads, max(rb, 1), **kw), want)
    _same(_count(r, src, monkeypatch, 8, 64, **kw), want)                 # gzip: one range whatever is asked


def test_seams(tmp_path, monkeypatch):
    fx = get_fixture("syn300_se")
    r = HL.HostRsh(os.path.join(fx.dir, "index.rsh"))
    names = r.names
    rng = random.Random(5)
    lines = []
    rid = 0
    for _ in range(400):
        n = rng.choice([1, 1, 2, 3, 5, 9])
        if rng.random() < 0.02:
            n = 700                                                    # one read with hundreds of alignments: spans many ranges
        t0 = rng.randrange(len(names))
        for j in range(n):
            strand = "+" if rng.random() < 0.6 else "-"               # "-" records are filtered under ssf and must not break a group
            lines.append(G.bowtie_line("q%d" % rid, strand, names[(t0 + j) % len(names)], 3 + j, 50, "" if rng.random() < 0.7 else "10:A>C"))
        rid += 1
    p = str(tmp_path / "seams.bowtie")
    with open(p, "w") as f:
        f.writelines(lines)
    size = os.path.getsize(p)
    for strand in ("ns", "ssf", "ssr"):
        want = _count(r, p, monkeypatch, 1, 1 << 40, strand=strand, max_repeat=1000)
        for threads, rb in ((2, size // 2), (5, size // 5), (13, size // 13), (64, 211), (64, 64)):
            _same(_count(r, p, monkeypatch, threads, rb, strand=strand, max_repeat=1000), want)


def test_pairs_cut_between_mates(tmp_path, monkeypatch):
    fx = get_fixture("toy5_pe")
    r = HL.HostRsh(os.path.join(fx.dir, "index.rsh"))
    src, fmt = aln_path(fx.dir)
    plain = str(tmp_path / "pe.txt")
    with gzip.open(src, "rb") as f:
        raw = f.read()
    open(plain, "wb").write(raw)
    want = _count(r, plain, monkeypatch, 1, 1 << 40, pe=1)
    # every byte offset inside the first kilobytes as a range boundary: mate 1 | mate 2, mid-line, line start
    line_starts = [i + 1 for i, ch in enumerate(raw[:4000]) if ch == 10]
    assert len(line_starts) > 10
    for threads in range(2, 64):                                       # boundaries at every multiple of size / threads
        _same(_count(r, plain, monkeypatch, threads, 1, pe=1), want)


def test_ranges_are_really_used(tmp_path, monkeypatch, capfd):
    fx = get_fixture("syn2k_se")
    r = HL.HostRsh(os.path.join(fx.dir, "index.rsh"))
    plain = str(tmp_path / "reads.txt")
    with gzip.open(aln_path(fx.dir)[0], "rb") as f, open(plain, "wb") as g:
        g.write(f.read())
    monkeypatch.setenv("EMSAR_HOST_DEBUG", "1")
    _count(r, plain, monkeypatch, 6, 1000)
    assert "6 range(s)" in capfd.readouterr().err
    _count(r, aln_path(fx.dir)[0], monkeypatch, 6, 1000)               # gzip input: sequential
    assert "1 range(s)" in capfd.readouterr().err


def _rsh_arrays(r):
    return (r.names, r.row_ptr.tobytes(), r.col_idx.tobytes(), r.euma.tobytes(), r.has_node.tobytes())


@pytest.mark.parametrize("case", ["toy5_se50", "toy5_pe", "syn300_se", "syn2k_se"])
def test_rsh_body_parsed_in_ranges(case, tmp_path, monkeypatch, capfd):
    """emsar_rsh_read parses the body of a plain rsh in byte ranges on several threads: names, row order (size, first
    tid, file order), tids, EUMA vectors and 'last single-tid line wins' must be those of the one-thread read."""
    fx = get_fixture(case)
    src = os.path.join(fx.dir, "index.rsh")
    monkeypatch.setenv("EMSAR_HOST_THREADS", "1")
    want = _rsh_arrays(HL.HostRsh(src))
    size = os.path.getsize(src)
    monkeypatch.setenv("EMSAR_HOST_DEBUG", "1")
    for threads, rb in ((2, size // 5), (5, size // 11), (16, max(20, size // 40)), (64, 17)):
        monkeypatch.setenv("EMSAR_HOST_THREADS", str(threads))
        monkeypatch.setenv("EMSAR_HOST_RANGE_BYTES", str(max(rb, 1)))
        capfd.readouterr()
        got = _rsh_arrays(HL.HostRsh(src))
        assert "emsar_rsh_read: 1 part(s)" not in capfd.readouterr().err
        assert got == want
    gz = str(tmp_path / "index.rsh.gz")
    with open(src, "rb") as f, gzip.open(gz, "wb") as g:
        g.write(f.read())
    capfd.readouterr()
    assert _rsh_arrays(HL.HostRsh(gz)) == want                    # gzip: one part whatever is asked
    assert "emsar_rsh_read: 1 part(s)" in capfd.readouterr().err


def test_rsh_ranges_keep_file_order_semantics(tmp_path, monkeypatch):
    """Lines whose order matters, spread over many ranges: a transcript with several single-tid lines (the last one
    wins), several @ lines for one tid (the last name wins), equal (size, first tid) nodes (file order), and the
    errors of the one-thread read (first bad line in file order, a second header line)."""
    fx = get_fixture("syn300_se")
    text = open(os.path.join(fx.dir, "index.rsh")).read().split("\n")
    hdr = [l for l in text if l.startswith("#")][0]
    n_tx = int(hdr[1:].split(",")[0]) + 1
    body = [l for l in text if l and not l.startswith("#")]
    rng = random.Random(3)
    extra = []
    for _ in range(60):
        t = rng.randrange(n_tx)
        extra.append("0\t1\t%d\t\t%d,%d,%d," % (t, rng.randrange(900), rng.randrange(900), rng.randrange(900)))
        extra.append("@%d\tRENAMED%d_%d" % (t, t, rng.randrange(1000)))
        a = rng.randrange(n_tx - 3)
        extra.append("0\t2\t%d\t%d,\t%d,%d,%d," % (a, a + 1 + rng.randrange(2), rng.randrange(90), rng.randrange(90), rng.randrange(90)))
    mains = [l for l in body if not l.startswith("@")] + [e for e in extra if not e.startswith("@")]
    rng.shuffle(mains)
    ats = [l for l in body if l.startswith("@")] + [e for e in extra if e.startswith("@")]
    # renamed @ lines must stay unique as names: the name index maps a name to one tid
    path = str(tmp_path / "order.rsh")
    open(path, "w").write("\n".join([hdr] + ats + mains) + "\n")
    monkeypatch.setenv("EMSAR_HOST_THREADS", "1")
    want = _rsh_arrays(HL.HostRsh(path))
    for threads, rb in ((3, 4000), (16, 500), (64, 61)):
        monkeypatch.setenv("EMSAR_HOST_THREADS", str(threads))
        monkeypatch.setenv("EMSAR_HOST_RANGE_BYTES", str(rb))
        assert _rsh_arrays(HL.HostRsh(path)) == want
    # errors: the first bad line in file order names the error, with any number of ranges
    bad = [hdr] + ats + mains[:200] + ["7\t2\t%d\t5,\t1,2,3," % (n_tx + 5)] + mains[200:400] + ["9\t3\t1\t2,\t1,2,3,"] + mains[400:] + [hdr]
    path2 = str(tmp_path / "bad.rsh")
    open(path2, "w").write("\n".join(bad) + "\n")
    msgs = []
    for threads, rb in ((1, 1 << 30), (7, 300)):
        monkeypatch.setenv("EMSAR_HOST_THREADS", str(threads))
        monkeypatch.setenv("EMSAR_HOST_RANGE_BYTES", str(rb))
        with pytest.raises(HL.HostError) as e:
            HL.HostRsh(path2)
        msgs.append(str(e.value))
    assert msgs[0] == msgs[1] and "bad segment line" in msgs[0]


def test_collapse_callback_is_never_entered_by_two_threads(tmp_path, monkeypatch):
    """emsar_aln_opts.collapse is flushed by the parse workers from their own threads (a batch per worker).  The library serialises
    the calls: a callback with one device context / one stream behind it (emsar_hip_collapse_rows through ctypes releases the GIL) is
    entered by one thread at a time.  Here the callback sleeps with the GIL released and checks that it was alone; counts equal the
    one-thread per-read path's."""
    import gzip
    import threading
    import time
    import oracle as O
    case = get_fixture("syn2k_se").dir
    r = HL.HostRsh(os.path.join(case, "index.rsh"))
    src = os.path.join(case, "reads.bowtie.gz")
    plain = str(tmp_path / "reads.bowtie")
    with gzip.open(src, "rb") as f, open(plain, "wb") as g:
        g.write(f.read())
    monkeypatch.setenv("EMSAR_HOST_THREADS", "1")
    want = r.count(plain, fmt=0)
    inside = threading.Lock()
    seen = {"calls": 0, "overlap": 0, "threads": set()}

    def collapse(rp, ci):
        if not inside.acquire(blocking=False):
            seen["overlap"] += 1
            inside.acquire()
        try:
            seen["calls"] += 1
            seen["threads"].add(threading.get_ident())
            time.sleep(0.002)                                             # GIL released: another worker could enter now
            a, b, w, _ = O.collapse_rows(rp, ci)
            return a, b, w
        finally:
            inside.release()

    monkeypatch.setenv("EMSAR_HOST_THREADS", "6")
    monkeypatch.setenv("EMSAR_HOST_RANGE_BYTES", "4096")
    got = r.count(plain, fmt=0, collapse=collapse, collapse_batch_rows=50)
    assert (got.R == want.R).all() and got.total_reads == want.total_reads and got.stats == want.stats
    assert seen["calls"] > 6 and seen["overlap"] == 0
