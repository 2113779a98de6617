"""CPU: text alignments that cannot be cut into byte ranges (gzip, paired-end SAM) are read by one thread and counted by a pool
(emsar_amd/csrc/host/align.c, count_text_parallel).  Whatever the number of threads and the batch size, the counts are those of the
one-thread loop, errors included."""
import gzip
import os
import random
import sys

import numpy as np
import pytest

from emsar_amd import _build, hostlib as HL
from tests.conftest import CASES, get_fixture

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_golden as G


def _outcome(r, path, monkeypatch, threads, batch, **kw):
    monkeypatch.setenv("EMSAR_HOST_THREADS", threads)
    monkeypatch.setenv("EMSAR_HOST_RANGE_BYTES", batch)
    monkeypatch.setenv("EMSAR_HOST_TEXT_POOL", "gz")                 # gzip goes through the pool only on request (it is slower there)
    try:
        c = r.count(path, **kw)
    except HL.HostError as e:
        return str(e).replace(path, "<the file>")
    return (c.R.tobytes(), c.frag_counts.tobytes(), c.total_reads, c.stats)


@pytest.mark.parametrize("case", [c for c in CASES if not c.endswith("bam")])
def test_gzip_fixtures_through_the_pool(case, monkeypatch, capfd):
    """every golden fixture whose alignments are gzipped text (bowtie and SAM, single- and paired-end): gzip cannot be read in ranges"""
    from tests.conftest import aln_path
    _build.build_host()
    fx = get_fixture(case)
    r = HL.HostRsh(os.path.join(fx.dir, "index.rsh"))
    path, fmt = aln_path(fx.dir)
    kw = dict(fmt=fmt, pe=1 if "-P" in fx.meta["opts"] else 0)
    want = _outcome(r, path, monkeypatch, "1", "1", **kw)
    assert not isinstance(want, str)
    monkeypatch.setenv("EMSAR_HOST_DEBUG", "1")
    for threads, batch in (("2", "1"), ("4", "3000"), ("16", "100000")):
        capfd.readouterr()
        assert _outcome(r, path, monkeypatch, threads, batch, **kw) == want, (threads, batch)
        assert "text, " in capfd.readouterr().err                    # the pool really ran


def test_sam_text_single_end_gzip_and_plain(tmp_path, monkeypatch):
    _build.build_host()
    fx = get_fixture("syn2k_se")
    r = HL.HostRsh(os.path.join(fx.dir, "index.rsh"))
    rng = random.Random(3)
    lines = ["@SQ\tSN:%s\tLN:100000\n" % n for n in r.names]
    for line in gzip.open(os.path.join(fx.dir, "reads.bowtie.gz"), "rt"):
        q = line.rstrip("\n").split("\t")
        lines.append(G.sam_line(q[0], 0 if q[1] == "+" else 16, q[2], int(q[3]), len(q[4]), str(len(q[4]))))
        if rng.random() < 0.05:
            lines.append("u\t4\t*\t0\t0\t*\t*\t0\t0\t%s\t%s\n" % ("A" * 30, "I" * 30))
    sam = str(tmp_path / "r.sam")
    open(sam, "w").write("".join(lines))
    gz = str(tmp_path / "r.sam.gz")
    with gzip.open(gz, "wt") as f:
        f.write("".join(lines))
    for strand in ("ns", "ssf", "ssr"):
        want = _outcome(r, sam, monkeypatch, "1", "1", fmt=1, strand=strand)
        assert not isinstance(want, str) or ("no usable" in want and strand != "ns"), (strand, want)
        for path in (sam, gz):
            for threads, batch in (("3", "1"), ("8", "50000")):
                assert _outcome(r, path, monkeypatch, threads, batch, fmt=1, strand=strand) == want, (strand, path, threads, batch)


def _pe_sam(names, seed, n_groups=1500):
    """the records of tests/test_pbgzf.py::test_paired_end_bam_seams as SAM text: groups of several pairs, pairs that fail the orientation
    filter inside and between groups, unaligned records skipped singly between pairs, mates in either order, an id that comes back"""
    rng = random.Random(seed)
    L = 50
    lines = ["@SQ\tSN:%s\tLN:100000\n" % n for n in names]
    rid = 0
    last_kept = None
    for _ in range(n_groups):
        rid += 1
        name = "q%d" % (rid if rng.random() > 0.15 or last_kept is None else last_kept)
        for _k in range(rng.choice([1, 1, 2, 3, 6])):
            t = rng.choice(names)
            a = rng.randrange(0, 300)
            b = a + rng.randrange(98, 113)
            kind = rng.random()
            if kind < 0.7:      f1, f2 = 0x1 | 0x2 | 0x20 | 0x40, 0x1 | 0x2 | 0x10 | 0x80
            elif kind < 0.85:   f1, f2 = 0x1 | 0x2 | 0x10 | 0x40, 0x1 | 0x2 | 0x20 | 0x80
            else:               f1, f2 = 0x1 | 0x2 | 0x10 | 0x20 | 0x40, 0x1 | 0x2 | 0x10 | 0x80
            m1 = G.sam_line(name, f1, t, a, L, str(L), "=")
            m2 = G.sam_line(name, f2, t, b, L, str(L), "=")
            lines += [m1, m2] if rng.random() < 0.6 else [m2, m1]
            if rng.random() < 0.2:
                lines.append("u%d\t4\t*\t0\t0\t*\t*\t0\t0\t%s\t%s\n" % (rid, "A" * L, "I" * L))
        last_kept = rid
    return lines


def test_paired_end_sam_text_seams(tmp_path, monkeypatch):
    _build.build_host()
    fx = get_fixture("toy5_pe")
    r = HL.HostRsh(os.path.join(fx.dir, "index.rsh"))
    lines = _pe_sam(r.names, 9)
    sam = str(tmp_path / "pe.sam")
    open(sam, "w").write("".join(lines))
    for strand in ("ns", "ssfr", "ssrf"):
        want = _outcome(r, sam, monkeypatch, "1", "1", pe=1, fmt=1, strand=strand)
        assert not isinstance(want, str) or "no usable" in want
        for threads, batch in (("2", "1"), ("5", "700"), ("16", "20000")):
            assert _outcome(r, sam, monkeypatch, threads, batch, pe=1, fmt=1, strand=strand) == want, (strand, threads, batch)
    assert not isinstance(_outcome(r, sam, monkeypatch, "1", "1", pe=1, fmt=1, strand="ns"), str)


def test_errors_are_those_of_the_one_thread_loop(tmp_path, monkeypatch):
    """a malformed record, an unknown transcript, ungrouped mates, a header line where a mate should be: same message or same counts"""
    _build.build_host()
    fx = get_fixture("toy5_pe")
    r = HL.HostRsh(os.path.join(fx.dir, "index.rsh"))
    base = _pe_sam(r.names, 4, n_groups=400)
    n_hdr = len(r.names)
    variants = {}
    v = list(base); v.insert(n_hdr + 801, "broken\tline\n"); variants["malformed"] = v
    v = list(base); v[n_hdr + 600] = v[n_hdr + 600].replace("\t" + v[n_hdr + 600].split("\t")[2] + "\t", "\tNOT_A_TRANSCRIPT\t", 1); variants["unknown transcript"] = v
    v = list(base); v.insert(n_hdr + 1001, "@CO\ta comment in the middle\n"); variants["header in the middle"] = v
    k = next(i for i in range(n_hdr + 300, len(base)) if base[i].split("\t")[1] != "4" and base[i + 1].split("\t")[1] != "4"
             and base[i].split("\t")[0] == base[i + 1].split("\t")[0])
    v = list(base); q = v[k].split("\t"); q[1] = str(int(q[1]) & ~0xC0); v[k] = "\t".join(q); variants["ungrouped mates"] = v
    for what, lines in variants.items():
        p = str(tmp_path / (what.replace(" ", "_") + ".sam"))
        open(p, "w").write("".join(lines))
        want = _outcome(r, p, monkeypatch, "1", "1", pe=1, fmt=1)
        for threads, batch in (("4", "1"), ("7", "5000")):
            assert _outcome(r, p, monkeypatch, threads, batch, pe=1, fmt=1) == want, (what, threads, batch)
