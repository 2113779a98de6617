"""CPU: parallel BGZF inflate (emsar_amd/csrc/host/pbgzf.c) behind the BAM reader -- a multi-block BAM made from a
golden bowtie fixture must give the read counts of the text, with any number of threads; damaged blocks are errors."""
import gzip
import os
import sys

import numpy as np
import pytest

from emsar_amd import _build, hostlib as HL
from tests.conftest import get_fixture

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_golden as G


@pytest.fixture(scope="module")
def bam_case(tmp_path_factory):
    _build.build_host()
    fx = get_fixture("syn2k_se")
    d = tmp_path_factory.mktemp("pbgzf")
    r = HL.HostRsh(os.path.join(fx.dir, "index.rsh"))
    sam = str(d / "reads.sam")
    with open(sam, "w") as f:
        for n in r.names:
            f.write("@SQ\tSN:%s\tLN:100000\n" % n)
        for line in gzip.open(os.path.join(fx.dir, "reads.bowtie.gz"), "rt"):
            q = line.rstrip("\n").split("\t")                     # bowtie: name strand ref pos seq qual n mm
            f.write(G.sam_line(q[0], 0 if q[1] == "+" else 16, q[2], int(q[3]), len(q[4]), str(len(q[4]))))
    bam = str(d / "reads.bam")
    G.sam_to_bam(sam, bam)
    assert os.path.getsize(bam) > 3 * 20000                       # several BGZF blocks
    want = r.count(os.path.join(fx.dir, "reads.bowtie.gz"))
    return r, bam, want, d


@pytest.mark.parametrize("threads", ["1", "3", "16"])
def test_bam_counts_equal_text_counts(bam_case, threads, monkeypatch):
    r, bam, want, _ = bam_case
    monkeypatch.setenv("EMSAR_HOST_THREADS", threads)
    got = r.count(bam, fmt=2)
    np.testing.assert_array_equal(got.R, want.R)
    np.testing.assert_array_equal(got.frag_counts, want.frag_counts)
    assert got.total_reads == want.total_reads and got.stats == want.stats


@pytest.mark.parametrize("threads,batch", [("2", "1"), ("4", "3000"), ("16", "40000"), ("3", "100000000")])
def test_bam_records_counted_in_parallel_batches(bam_case, threads, batch, monkeypatch, capfd):
    """Single-end BAM: the calling thread cuts the inflated stream into batches of whole records at read-group
    boundaries, a pool counts them (align.c, count_bam_parallel).  Any batch size -- one group per batch included --
    gives the counts of the one-thread loop, with either strand filter."""
    r, bam, want, d = bam_case
    monkeypatch.setenv("EMSAR_HOST_THREADS", threads)
    monkeypatch.setenv("EMSAR_HOST_RANGE_BYTES", batch)
    monkeypatch.setenv("EMSAR_HOST_DEBUG", "1")
    got = r.count(bam, fmt=2)
    dbg = capfd.readouterr().err
    assert "BAM, " in dbg and "on %s thread(s)" % threads in dbg
    n_batches = int(dbg.split("BAM, ")[1].split(" batch")[0])
    assert n_batches > 50 if batch == "1" else n_batches >= 1
    np.testing.assert_array_equal(got.R, want.R)
    np.testing.assert_array_equal(got.frag_counts, want.frag_counts)
    assert got.total_reads == want.total_reads and got.stats == want.stats
    def outcome(n_threads, strand):
        monkeypatch.setenv("EMSAR_HOST_THREADS", n_threads)
        try:
            c = r.count(bam, fmt=2, strand=strand)
        except HL.HostError as e:                                  # a strand nobody maps to: 'NULL alignment list' either way
            return str(e)
        return (c.R.tobytes(), c.frag_counts.tobytes(), c.total_reads, c.stats)

    for strand in ("ssf", "ssr"):
        assert outcome("1", strand) == outcome(threads, strand)


@pytest.mark.parametrize("case", ["toy5_bam", "toy5_pe_bam"])
def test_golden_bam_fixtures_in_parallel_batches(case, monkeypatch, capfd):
    """The reference's own BAM inputs (both strands, unaligned records between the kept ones; mates in either order in the
    paired-end file): one read group per batch gives the one-thread counts."""
    _build.build_host()
    fx = get_fixture(case)
    r = HL.HostRsh(os.path.join(fx.dir, "index.rsh"))
    pe = 1 if "pe" in case else 0
    path = os.path.join(fx.dir, "reads.bam")
    monkeypatch.setenv("EMSAR_HOST_DEBUG", "1")
    for strand in ("ns", "ssf", "ssr") if not pe else ("ns", "ssfr", "ssrf"):
        res = []
        for threads, batch in (("1", "1"), ("4", "1"), ("3", "200")):
            monkeypatch.setenv("EMSAR_HOST_THREADS", threads)
            monkeypatch.setenv("EMSAR_HOST_RANGE_BYTES", batch)
            capfd.readouterr()
            try:
                c = r.count(path, pe=pe, strand=strand, fmt=2)
                res.append((c.R.tobytes(), c.frag_counts.tobytes(), c.total_reads, c.stats))
            except HL.HostError as e:
                res.append(str(e))
            assert ("BAM, " in capfd.readouterr().err) == (threads != "1")
        assert res[0] == res[1] == res[2]


def test_paired_end_bam_seams(tmp_path, monkeypatch):
    """Paired-end BAM built to hit the seams of the batch cutter: read groups of several pairs, pairs that fail the
    orientation filter inside and between groups (they must not split a group), unaligned records sprinkled singly
    between pairs (the reader skips them one by one, which shifts the pairing), mates in either order."""
    import random
    _build.build_host()
    fx = get_fixture("toy5_pe")
    r = HL.HostRsh(os.path.join(fx.dir, "index.rsh"))
    names = r.names
    rng = random.Random(9)
    L = 50
    lines = ["@SQ\tSN:%s\tLN:100000\n" % n for n in names]
    rid = 0
    last_kept = None
    for _ in range(1500):
        rid += 1
        name = "q%d" % (rid if rng.random() > 0.15 or last_kept is None else last_kept)   # now and then the previous id again
        for _k in range(rng.choice([1, 1, 2, 3, 6])):
            t = rng.choice(names)
            a = rng.randrange(0, 300)
            b = a + rng.randrange(98, 113)                         # fragment 148-162 against the index range 150-160
            kind = rng.random()
            if kind < 0.7:      f1, f2 = 0x1 | 0x2 | 0x20 | 0x40, 0x1 | 0x2 | 0x10 | 0x80      # mate 1 forward, mate 2 reverse: kept (ns)
            elif kind < 0.85:   f1, f2 = 0x1 | 0x2 | 0x10 | 0x40, 0x1 | 0x2 | 0x20 | 0x80      # reversed orientation at these positions: filtered
            else:               f1, f2 = 0x1 | 0x2 | 0x10 | 0x20 | 0x40, 0x1 | 0x2 | 0x10 | 0x80  # both reverse: filtered
            m1 = G.sam_line(name, f1, t, a, L, str(L), "=")
            m2 = G.sam_line(name, f2, t, b, L, str(L), "=")
            pair = [m1, m2] if rng.random() < 0.6 else [m2, m1]
            lines += pair
            if rng.random() < 0.2:
                lines.append("u%d\t4\t*\t0\t0\t*\t*\t0\t0\t%s\t%s\n" % (rid, "A" * L, "I" * L))
        last_kept = rid
    sam = str(tmp_path / "pe.sam")
    open(sam, "w").write("".join(lines))
    bam = str(tmp_path / "pe.bam")
    G.sam_to_bam(sam, bam)

    def outcome(threads, batch, strand):
        monkeypatch.setenv("EMSAR_HOST_THREADS", threads)
        monkeypatch.setenv("EMSAR_HOST_RANGE_BYTES", batch)
        try:
            c = r.count(bam, pe=1, strand=strand, fmt=2)
        except HL.HostError as e:
            return str(e)
        return (c.R.tobytes(), c.frag_counts.tobytes(), c.total_reads, c.stats)      # stats carries readlength

    for strand in ("ns", "ssfr", "ssrf"):
        want = outcome("1", "1", strand)
        assert not isinstance(want, str) or "no usable" in want
        for threads, batch in (("2", "1"), ("5", "700"), ("16", "20000")):
            assert outcome(threads, batch, strand) == want, (strand, threads, batch)
    assert not isinstance(outcome("1", "1", "ns"), str)
    # the same records as SAM text go through the one-thread reader: same counts
    monkeypatch.setenv("EMSAR_HOST_THREADS", "1")
    c_sam = r.count(sam, pe=1, strand="ns", fmt=1)
    assert outcome("4", "1", "ns")[:3] == (c_sam.R.tobytes(), c_sam.frag_counts.tobytes(), c_sam.total_reads)


def test_parallel_bam_errors_match_the_one_thread_loop(bam_case, monkeypatch):
    r, bam, _, d = bam_case
    raw = open(bam, "rb").read()
    p = str(d / "trunc2.bam")
    open(p, "wb").write(raw[: len(raw) // 2])
    msgs = []
    for threads in ("1", "4"):
        monkeypatch.setenv("EMSAR_HOST_THREADS", threads)
        monkeypatch.setenv("EMSAR_HOST_RANGE_BYTES", "5000")
        with pytest.raises(HL.HostError) as e:
            r.count(p, fmt=2)
        msgs.append(str(e.value))
    assert msgs[0] == msgs[1]


def test_damaged_blocks_are_errors(bam_case):
    r, bam, _, d = bam_case
    raw = bytearray(open(bam, "rb").read())
    # flip a byte inside the deflate data of the second block: CRC / inflate failure
    bsize = raw[16] | (raw[17] << 8)
    bad = bytearray(raw)
    bad[bsize + 1 + 40] ^= 0xFF
    p = str(d / "crc.bam")
    open(p, "wb").write(bytes(bad))
    with pytest.raises(HL.HostError):
        r.count(p, fmt=2)
    # truncated in the middle of a block
    p = str(d / "trunc.bam")
    open(p, "wb").write(bytes(raw[: len(raw) // 2]))
    with pytest.raises(HL.HostError):
        r.count(p, fmt=2)


def test_many_batches(tmp_path, monkeypatch):
    """More than one batch of 512 blocks: the prefetched batch takes over where the first one ends."""
    import struct
    import zlib
    _build.build_host()
    lib = HL.lib()
    import ctypes as C
    lib.emsar_pbgzf_open.restype = C.c_void_p
    lib.emsar_pbgzf_open.argtypes = [C.c_char_p]
    lib.emsar_pbgzf_read.restype = C.c_long
    lib.emsar_pbgzf_read.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.emsar_pbgzf_close.argtypes = [C.c_void_p]
    rng = np.random.default_rng(0)
    payload = rng.integers(0, 7, size=1300 * 3000, dtype=np.uint8).tobytes()      # 1300 small blocks
    path = str(tmp_path / "x.bgzf")
    with open(path, "wb") as fo:
        for i in range(0, len(payload), 3000):
            chunk = payload[i:i + 3000]
            c = zlib.compressobj(1, zlib.DEFLATED, -15)
            data = c.compress(chunk) + c.flush()
            fo.write(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(data) + 25) + data +
                     struct.pack("<II", zlib.crc32(chunk), len(chunk)))
        fo.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
    for threads in ("1", "5"):
        monkeypatch.setenv("EMSAR_HOST_THREADS", threads)
        h = lib.emsar_pbgzf_open(path.encode())
        assert h
        out = bytearray()
        buf = C.create_string_buffer(70001)
        while True:
            n = lib.emsar_pbgzf_read(h, buf, 70001)
            assert n >= 0
            out += buf.raw[:n]
            if n < 70001:
                break
        lib.emsar_pbgzf_close(h)
        assert bytes(out) == payload
    assert lib.emsar_pbgzf_open(os.path.join(os.path.dirname(__file__), "golden", "syn2k_se", "reads.bowtie.gz").encode()) is None   # plain gzip


def test_both_inflate_engines_read_the_same_bytes(bam_case):
    """pbgzf.c inflates with libdeflate when the system has its shared library and with zlib otherwise (or when
    EMSAR_HOST_INFLATE=zlib says so).  The engine is chosen once per process: two child processes, same file, same counts;
    a damaged block is an error under both."""
    import json
    import subprocess
    _, bam, want, d = bam_case
    fx = get_fixture("syn2k_se")
    raw = bytearray(open(bam, "rb").read())
    bsize = raw[16] | (raw[17] << 8)
    raw[bsize + 1 + 40] ^= 0xFF
    bad = str(d / "crc2.bam")
    open(bad, "wb").write(bytes(raw))
    code = ("import sys, json, ctypes, zlib\n"
            "sys.path.insert(0, %r)\n"
            "from emsar_amd import hostlib as HL\n"
            "lib = HL.lib(); lib.emsar_pbgzf_engine.restype = ctypes.c_char_p\n"
            "r = HL.HostRsh(%r)\n"
            "c = r.count(%r, fmt=2)\n"
            "try:\n"
            "    r.count(%r, fmt=2); err = False\n"
            "except HL.HostError:\n"
            "    err = True\n"
            "print(json.dumps({'engine': lib.emsar_pbgzf_engine().decode(), 'crc': zlib.crc32(c.R.tobytes()), 'total': int(c.total_reads), 'damaged_is_error': err}))\n"
            % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(fx.dir, "index.rsh"), bam, bad))
    out = {}
    for eng in ("", "zlib"):
        env = dict(os.environ, EMSAR_HOST_INFLATE=eng, EMSAR_HOST_THREADS="3")
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr
        out[eng] = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["zlib"]["engine"] == "zlib"
    assert out[""]["engine"] in ("libdeflate", "zlib")          # libdeflate where libdeflate.so.0 exists (this image), zlib elsewhere
    import zlib
    for o in out.values():
        assert o["crc"] == zlib.crc32(want.R.tobytes()) and o["total"] == want.total_reads and o["damaged_is_error"]
