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
