#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the compiled reference.

Runs only in the build container (needs oracle/_ref/emsar and emsar-build, which
oracle/Makefile compiles from /root/reference/src).  The fixtures are DATA: our own
synthetic inputs (FASTA / rsh text / default-bowtie text) plus the files the reference
wrote for them (.fpkm, .segments, .fraglength_effect).  Nothing of the reference's
source is stored.

Every case is run RUNS times because the reference seeds rand() with time(NULL)
(emsar_main.c:441) and disagrees with itself in the last digits;
run 0 keeps all three outputs, later runs keep the .fpkm only (noise mask).

    python tests/golden/make_golden.py [case ...]
"""
import gzip
import json
import os
import random
import shutil
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_EMSAR = os.path.join(ROOT, "oracle", "_ref", "emsar")
REF_BUILD = os.path.join(ROOT, "oracle", "_ref", "emsar-build")
FIXED_TIME = os.path.join(ROOT, "oracle", "_ref", "libfixedtime.so")
RUNS = 3
SEED = 12345


def rand_seq(rng, n):
    return "".join(rng.choice("ACGT") for _ in range(n))


def bowtie_line(rid, strand, tname, pos, fraglen, mm=""):
    # default bowtie output: name, strand, ref, 0-based offset, seq, qual, other-count, mismatches.
    # The reference reads fields 1-5 and 8 only; the LENGTH of field 5 is the fragment length
    # (emsar_functions.c:565-573).
    return "%s\t%s\t%s\t%d\t%s\t%s\t0\t%s\n" % (rid, strand, tname, pos, "A" * fraglen, "I" * fraglen, mm)


def gzip_inplace(path):
    # the reference reads plain text; fixtures are stored gzipped (mtime=0 for reproducible bytes)
    with open(path, "rb") as fi, open(path + ".gz", "wb") as raw:
        with gzip.GzipFile(filename="", mode="wb", compresslevel=9, fileobj=raw, mtime=0) as fo:
            shutil.copyfileobj(fi, fo)
    os.remove(path)


def sam_line(rid, flag, tname, pos0, length, md, mate=""):
    # SAM text: QNAME FLAG RNAME POS(1-based) MAPQ CIGAR RNEXT PNEXT TLEN SEQ QUAL tags
    return "%s\t%d\t%s\t%d\t255\t%dM\t%s\t0\t0\t%s\t%s\tXA:i:0\tMD:Z:%s\tNM:i:0\n" % (
        rid, flag, tname, pos0 + 1, length, mate or "*", "A" * length, "I" * length, md)


def sam_to_bam(sam_path, bam_path):
    """Minimal SAM text -> BAM (BGZF) writer, enough for the records this script emits (single-op CIGAR, Z/i tags).
    Layout: SAM/BAM specification section 4; BGZF blocks carry the 'BC' extra field the reference's bgzf.c reads."""
    import struct
    import zlib
    text, refs, recs = "", [], []
    for line in open(sam_path):
        if line.startswith("@"):
            text += line
            if line.startswith("@SQ"):
                f = dict(x.split(":", 1) for x in line.rstrip("\n").split("\t")[1:])
                refs.append((f["SN"], int(f["LN"])))
        else:
            recs.append(line.rstrip("\n").split("\t"))
    rid = {n: i for i, (n, _) in enumerate(refs)}
    out = bytearray(b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(refs)))
    for n, ln in refs:
        out += struct.pack("<i", len(n) + 1) + n.encode() + b"\0" + struct.pack("<i", ln)
    code = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}
    for f in recs:
        name, flag, rname, pos, mapq, cigar, rnext, pnext, tlen, seq, qual = f[:11]
        flag, pos, mapq, pnext, tlen = int(flag), int(pos), int(mapq), int(pnext), int(tlen)
        ref = -1 if rname == "*" else rid[rname]
        nref = -1 if rnext == "*" else (ref if rnext == "=" else rid[rnext])
        cig = b"" if cigar == "*" else struct.pack("<I", (int(cigar[:-1]) << 4) | "MIDNSHP=X".index(cigar[-1]))
        l_seq = len(seq)
        packed = bytearray((l_seq + 1) // 2)
        for i, c in enumerate(seq):
            packed[i // 2] |= code[c] << (4 if i % 2 == 0 else 0)
        q = bytes(ord(c) - 33 for c in qual) if qual != "*" else b"\xff" * l_seq
        aux = b""
        for tag in f[11:]:
            t, ty, v = tag.split(":", 2)
            aux += t.encode() + (b"Z" + v.encode() + b"\0" if ty == "Z" else b"i" + struct.pack("<i", int(v)))
        body = struct.pack("<iiBBHHHiiii", ref, pos - 1, len(name) + 1, mapq, 4680, 0 if cigar == "*" else 1, flag, l_seq,
                           nref, pnext - 1, tlen) + name.encode() + b"\0" + cig + bytes(packed) + q + aux
        out += struct.pack("<i", len(body)) + body
    with open(bam_path, "wb") as fo:
        for i in range(0, len(out), 60000):
            chunk = bytes(out[i:i + 60000])
            c = zlib.compressobj(6, zlib.DEFLATED, -15)
            data = c.compress(chunk) + c.flush()
            fo.write(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(data) + 25) + data +
                     struct.pack("<II", zlib.crc32(chunk), len(chunk)))
        fo.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))   # BGZF EOF block


def run_reference(case_dir, rsh, aln, extra_opts, runs=RUNS):
    out = os.path.join(case_dir, "_out")
    for r in range(runs):
        shutil.rmtree(out, ignore_errors=True)
        cmd = [REF_EMSAR, "-q", "-g"] + extra_opts + ["-I", rsh, out, "ref", aln]
        t0 = time.time()
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        dt = time.time() - t0
        if r == 0:
            for ext in ("fpkm", "segments", "fraglength_effect"):
                shutil.copy(os.path.join(out, "ref.0." + ext), os.path.join(case_dir, "ref.run0." + ext))
        else:
            shutil.copy(os.path.join(out, "ref.0.fpkm"), os.path.join(case_dir, "ref.run%d.fpkm" % r))
        print("  run %d: %.1fs" % (r, dt))
        time.sleep(1.1)  # new time(NULL) seed for the next run
    # seeded run: time() pinned by oracle/_ref/libfixedtime.so, one thread -> rand() stream reproducible.
    # Pins oracle_mle_pattern_search bit-for-bit (tests/test_oracle_golden.py).
    shutil.rmtree(out, ignore_errors=True)
    env = dict(os.environ, LD_PRELOAD=FIXED_TIME, EMSAR_FIXED_TIME=str(SEED))
    scmd = [REF_EMSAR, "-q"] + extra_opts + ["-p", "1", "-I", rsh, out, "ref", aln]
    subprocess.run(scmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=env)
    shutil.copy(os.path.join(out, "ref.0.fpkm"), os.path.join(case_dir, "ref.seed%d.fpkm" % SEED))
    shutil.rmtree(out, ignore_errors=True)
    return cmd


# ----------------------------------------------------------------------------------------------
# case 1: 5 transcripts, real emsar-build index (exercises internal repeats "2,4,4")
# ----------------------------------------------------------------------------------------------
def case_toy5(case_dir):
    rng = random.Random(7)
    ex = [rand_seq(rng, 120) for _ in range(6)]
    tx = [("tA", ex[0] + ex[1] + ex[2]), ("tB", ex[0] + ex[2]), ("tC", ex[3] + ex[1] + ex[4]),
          ("tD", ex[5]), ("tE", ex[3] + ex[4] + ex[3])]
    L = 50
    with open(os.path.join(case_dir, "tx.fa"), "w") as f:
        for n, s in tx:
            f.write(">%s\n%s\n" % (n, s))
    subprocess.run([REF_BUILD, "-q", os.path.join(case_dir, "tx.fa"), str(L), case_dir, "index"],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    # exact alignments: every (tid,pos) where the 50-mer occurs on the forward strand
    occ = {}
    for n, s in tx:
        for p in range(len(s) - L + 1):
            occ.setdefault(s[p:p + L], []).append((n, p))
    abund = {"tA": 30.0, "tB": 12.0, "tC": 7.0, "tD": 0.0, "tE": 3.0}
    starts = [(n, p) for n, s in tx for p in range(len(s) - L + 1) for _ in range(1)]
    weights = [abund[n] for n, p in starts]
    n_reads = 3000
    lines = []
    seqs = dict(tx)
    for i in range(n_reads):
        n, p = rng.choices(starts, weights)[0]
        kmer = seqs[n][p:p + L]
        hits = occ[kmer]
        for (hn, hp) in hits:
            lines.append(bowtie_line("r%d" % i, "+", hn, hp, L))
        if i % 500 == 0:  # consecutive duplicate record: dropped by alignment.c:37-41
            lines.append(bowtie_line("r%d" % i, "+", hits[0][0], hits[0][1], L))
        if i % 700 == 1:  # a worse (1-mismatch) extra hit on tD: dropped by the best-mm filter
            lines.append(bowtie_line("r%d" % i, "+", "tD", 3, L, "10:A>C"))
    aln = os.path.join(case_dir, "reads.bowtie")
    with open(aln, "w") as f:
        f.writelines(lines)
    cmd = run_reference(case_dir, os.path.join(case_dir, "index.rsh"), aln, [])
    gzip_inplace(aln)
    return {"n_reads_emitted": n_reads, "total_read_count": n_reads, "opts": [], "cmd": " ".join(cmd)}


# ----------------------------------------------------------------------------------------------
# synthetic rsh written directly (format: emsar_functions.c:2085-2127)
# ----------------------------------------------------------------------------------------------
def synth_rsh_case(case_dir, seed, n_tx, minfrag, maxfrag, n_reads, opts, fam_max=6,
                   n_orphans=20, n_badlen=15, with_quirks=True):
    rng = random.Random(seed)
    nfl = maxfrag - minfrag + 1
    names = ["ENST%07d" % (1000 + i) for i in range(n_tx)]
    # families of consecutive tids
    fams, t = [], 0
    while t < n_tx:
        k = min(n_tx - t, rng.choice([1, 1, 1, 2, 2, 3, 4, fam_max]))
        fams.append(list(range(t, t + k)))
        t += k
    singles = {}       # tid -> EUMA list (or None: no unique region)
    multis = {}        # tuple(sorted tids with repeats) -> EUMA list

    def euma_vec(base):
        # effective count per fragment length: shrinks by one position per extra base
        return [max(0, base - i) for i in range(nfl)]

    for fam in fams:
        for tid in fam:
            if with_quirks and len(fam) > 1 and rng.random() < 0.15:
                singles[tid] = None                      # no unique region -> empty EUMA row
            else:
                singles[tid] = euma_vec(rng.randint(30, 1500))
        if len(fam) > 1:
            n_sub = rng.randint(1, min(6, 2 ** len(fam) - len(fam) - 1))
            for _ in range(n_sub):
                k = rng.randint(2, len(fam))
                sub = sorted(rng.sample(fam, k))
                if with_quirks and rng.random() < 0.1:  # internal repeat: a tid listed twice
                    sub = sorted(sub + [rng.choice(sub)])
                multis[tuple(sub)] = euma_vec(rng.randint(20, 900))
    if with_quirks:
        # cross-family segments
        for _ in range(max(1, n_tx // 40)):
            a, b = rng.sample(range(n_tx), 2)
            multis[tuple(sorted((a, b)))] = euma_vec(rng.randint(20, 200))
        # a tied pair: two transcripts that only ever occur together (identical columns)
        a = fams[-1][0]
        if len(fams[-1]) >= 2:
            b = fams[-1][1]
            for key in [k for k in multis if (a in k) != (b in k)]:
                del multis[key]
            singles[a] = None
            singles[b] = None
            multis[(a, b)] = euma_vec(400)
        # a multi segment whose EUMA is zero at every length -> E_c = 0 row
        f0 = next(f for f in fams if len(f) >= 2)
        multis[tuple(f0[:2])] = [0] * nfl
    max_t = max([len(k) for k in multis] + [1])

    rsh = os.path.join(case_dir, "index.rsh")
    with open(rsh, "w") as f:
        f.write("#%d,%d,%d,%d,%d\n" % (n_tx - 1, max_t, minfrag, maxfrag, -1))
        for i, n in enumerate(names):
            f.write("@%d\t%s\n" % (i, n))
        f.write("cid\tno.tids\tfirst.tid\tother.tids\tsegment.length\n")
        cid = 0
        for tid in range(n_tx):
            e = singles[tid]
            if e is None:
                f.write("%d\t1\t%d\t\t\t\n" % (cid, tid))
            else:
                f.write("%d\t1\t%d\t\t%s\n" % (cid, tid, "".join("%d," % x for x in e)))
            cid += 1
        for size in range(2, max_t + 1):
            keys = sorted(k for k in multis if len(k) == size)
            for k in keys:
                f.write("%d\t%d\t%d\t%s\t%s\n" % (cid, size, k[0], "".join("%d," % x for x in k[1:]),
                                                  "".join("%d," % x for x in multis[k])))
                cid += 1

    # true abundances: log-normal with 30 % zeros; some whole families silent (zero-count sets)
    theta = [0.0 if rng.random() < 0.3 else rng.lognormvariate(0, 2) for _ in range(n_tx)]
    for fam in fams[::7]:
        for tid in fam:
            theta[tid] = 0.0
    flw = [rng.random() + 0.2 for _ in range(nfl)]
    segs = [((tid,), e) for tid, e in singles.items() if e is not None] + list(multis.items())
    choices, weights = [], []
    for key, e in segs:
        s = sum(theta[t] for t in key)
        for i in range(nfl):
            w = e[i] * flw[i] * s
            if w > 0:
                choices.append((key, i))
                weights.append(w)
    picks = rng.choices(choices, weights, k=n_reads)
    lines = []
    n_total = 0
    kmax = int(opts[opts.index("-k") + 1]) if "-k" in opts else 100
    for r, (key, i) in enumerate(picks):
        fl = minfrag + i
        order = list(key)
        rng.shuffle(order)                      # alignment order is arbitrary; the collapse sorts
        for j, tid in enumerate(order):
            lines.append(bowtie_line("r%d" % r, "+", names[tid], 10 + 7 * j, fl))
        n_total += len(order) <= kmax           # reads over -k are discarded, not counted (emsar_functions.c:752)
    # reads whose tid-set has no rsh node: counted in N, not in any R_c
    for r in range(n_orphans):
        a, b, c = rng.sample(range(n_tx), 3)
        for j, tid in enumerate((a, b, c)):
            lines.append(bowtie_line("orph%d" % r, "+", names[tid], 5 + j, minfrag))
        n_total += 3 <= kmax
    # reads with a fragment length outside [min,max]: dropped entirely (emsar_functions.c:849)
    for r in range(n_badlen):
        lines.append(bowtie_line("short%d" % r, "+", names[r % n_tx], 1, minfrag - 1))
    aln = os.path.join(case_dir, "reads.bowtie")
    with open(aln, "w") as f:
        f.writelines(lines)
    cmd = run_reference(case_dir, rsh, aln, opts)
    gzip_inplace(aln)
    return {"n_reads_emitted": n_reads + n_orphans + n_badlen, "total_read_count": n_total,
            "opts": opts, "cmd": " ".join(cmd), "true_theta_nonzero": sum(1 for x in theta if x > 0)}


def toy5_transcripts():
    rng = random.Random(7)
    ex = [rand_seq(rng, 120) for _ in range(6)]
    return [("tA", ex[0] + ex[1] + ex[2]), ("tB", ex[0] + ex[2]), ("tC", ex[3] + ex[1] + ex[4]),
            ("tD", ex[5]), ("tE", ex[3] + ex[4] + ex[3])]


def case_toy5_sam(case_dir, bam=False):
    """Same model as toy5_se50 but the reads come as SAM text (-S): exercises the MD:Z mismatch count,
    the 0x10 strand bit, unaligned records and 1-based POS."""
    rng = random.Random(17)
    tx = toy5_transcripts()
    L = 50
    with open(os.path.join(case_dir, "tx.fa"), "w") as f:
        for n, s_ in tx:
            f.write(">%s\n%s\n" % (n, s_))
    subprocess.run([REF_BUILD, "-q", os.path.join(case_dir, "tx.fa"), str(L), case_dir, "index"],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    occ = {}
    for n, s_ in tx:
        for p_ in range(len(s_) - L + 1):
            occ.setdefault(s_[p_:p_ + L], []).append((n, p_))
    abund = {"tA": 5.0, "tB": 40.0, "tC": 9.0, "tD": 2.0, "tE": 11.0}
    starts = [(n, p_) for n, s_ in tx for p_ in range(len(s_) - L + 1)]
    weights = [abund[n] for n, _ in starts]
    seqs = dict(tx)
    lines = ["@HD\tVN:1.0\tSO:unsorted\n"] + ["@SQ\tSN:%s\tLN:%d\n" % (n, len(s_)) for n, s_ in tx]
    n_reads = 2500
    for i in range(n_reads):
        n, p_ = rng.choices(starts, weights)[0]
        hits = occ[seqs[n][p_:p_ + L]]
        strand_flag = 16 if i % 3 == 0 else 0                     # unstranded library: both strands count
        for (hn, hp) in hits:
            lines.append(sam_line("r%d" % i, strand_flag, hn, hp, L, str(L)))
        if i % 400 == 2:                                            # worse hit: MD with one mismatch -> filtered
            lines.append(sam_line("r%d" % i, 0, "tD", 7, L, "10A39"))
        if i % 450 == 3:                                            # an unaligned record in between (skipped)
            lines.append("u%d\t4\t*\t0\t0\t*\t*\t0\t0\t%s\t%s\n" % (i, "A" * L, "I" * L))
    aln = os.path.join(case_dir, "reads.sam")
    with open(aln, "w") as f:
        f.writelines(lines)
    if bam:
        sam_to_bam(aln, os.path.join(case_dir, "reads.bam"))
        os.remove(aln)
        aln = os.path.join(case_dir, "reads.bam")
    opts = ["-B"] if bam else ["-S"]
    cmd = run_reference(case_dir, os.path.join(case_dir, "index.rsh"), aln, opts)
    if not bam:
        gzip_inplace(aln)
    return {"n_reads_emitted": n_reads, "total_read_count": n_reads, "opts": opts, "cmd": " ".join(cmd)}


def revcomp(s_):
    return s_[::-1].translate(str.maketrans("ACGT", "TGCA"))


def case_toy5_pe(case_dir, sam=False, bam=False):
    """Paired-end: real emsar-build --PE index (fragment lengths 150-160), pairs as default-bowtie text or SAM."""
    rng = random.Random(27 + sam)
    tx = toy5_transcripts()
    L, fmin, fmax = 50, 150, 160
    with open(os.path.join(case_dir, "tx.fa"), "w") as f:
        for n, s_ in tx:
            f.write(">%s\n%s\n" % (n, s_))
    subprocess.run([REF_BUILD, "-q", "--PE", "-f", str(fmin), "-F", str(fmax), os.path.join(case_dir, "tx.fa"), str(L),
                    case_dir, "index"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    seqs = dict(tx)
    occ = {}
    for n, s_ in tx:
        for p_ in range(len(s_) - L + 1):
            occ.setdefault(s_[p_:p_ + L], []).append((n, p_))
    abund = {"tA": 20.0, "tB": 6.0, "tC": 9.0, "tD": 3.0, "tE": 14.0}
    frags = [(n, p_, fl) for n, s_ in tx for fl in range(fmin, fmax + 1) for p_ in range(len(s_) - fl + 1)]
    weights = [abund[n] * (1.0 + 0.1 * (fl - fmin)) for n, p_, fl in frags]
    lines = ["@HD\tVN:1.0\tSO:unsorted\n"] + ["@SQ\tSN:%s\tLN:%d\n" % (n, len(s_)) for n, s_ in tx] if sam else []
    n_reads = 2500
    for i in range(n_reads):
        n, p_, fl = rng.choices(frags, weights)[0]
        m1, m2 = seqs[n][p_:p_ + L], seqs[n][p_ + fl - L:p_ + fl]
        # every transcript where both mates occur at a distance inside the aligner's insert window
        pairs = []
        for (h1, q1) in occ[m1]:
            for (h2, q2) in occ[m2]:
                if h1 == h2 and q2 >= q1 and fmin <= q2 - q1 + L <= fmax:
                    pairs.append((h1, q1, q2))
        flip = (i % 2 == 1)   # the fragment came from the other strand: mate1 is the reverse read
        for (h, q1, q2) in pairs:
            if sam:
                if not flip:
                    lines.append(sam_line("p%d" % i, 0x1 | 0x2 | 0x20 | 0x40, h, q1, L, str(L), "="))
                    lines.append(sam_line("p%d" % i, 0x1 | 0x2 | 0x10 | 0x80, h, q2, L, str(L), "="))
                else:
                    lines.append(sam_line("p%d" % i, 0x1 | 0x2 | 0x10 | 0x40, h, q2, L, str(L), "="))
                    lines.append(sam_line("p%d" % i, 0x1 | 0x2 | 0x20 | 0x80, h, q1, L, str(L), "="))
            else:
                a = ("+", q1)
                b = ("-", q2)
                first, second = (a, b) if not flip else (b, a)
                lines.append("p%d/1\t%s\t%s\t%d\t%s\t%s\t0\t\n" % (i, first[0], h, first[1], "A" * L, "I" * L))
                lines.append("p%d/2\t%s\t%s\t%d\t%s\t%s\t0\t\n" % (i, second[0], h, second[1], "A" * L, "I" * L))
    aln = os.path.join(case_dir, "reads.sam" if sam else "reads.bowtie")
    with open(aln, "w") as f:
        f.writelines(lines)
    if bam:
        sam_to_bam(aln, os.path.join(case_dir, "reads.bam"))
        os.remove(aln)
        aln = os.path.join(case_dir, "reads.bam")
    opts = ["-P"] + (["-B"] if bam else ["-S"] if sam else [])
    cmd = run_reference(case_dir, os.path.join(case_dir, "index.rsh"), aln, opts)
    if not bam:
        gzip_inplace(aln)
    return {"n_reads_emitted": n_reads, "total_read_count": None, "opts": opts, "cmd": " ".join(cmd)}


# ----------------------------------------------------------------------------------------------
# BASELINE config 1 stand-in: the bundled Vicugna BAM + rsh are absent from the checkout, so a Vicugna-SHAPED case:
# the 12 704 transcript names and the gene column of samples/Vicugna_pacos.vicPac1.72.cdna.all.g2t, random sequences in which the
# transcripts of a gene (and ~200 made-up paralog groups) share an exon block, paired-end L = 101, fragments 290-300, 100 000 pairs,
# through the real `emsar-build --PE -f 290 -F 300` and the real `emsar -P`.
# ----------------------------------------------------------------------------------------------
VICUGNA_G2T = "/root/reference/samples/Vicugna_pacos.vicPac1.72.cdna.all.g2t"


def case_vicugna_pe(case_dir, n_pairs=100000):
    rng = random.Random(72)
    L, fmin, fmax = 101, 290, 300
    names, genes = [], {}
    for line in open(VICUGNA_G2T):
        g, t = line.split()
        genes.setdefault(g, []).append(len(names))
        names.append(t)
    n_tx = len(names)
    groups = [m for m in genes.values() if len(m) > 1]                      # 37 genes with 2-3 transcripts
    lone = [m[0] for m in genes.values() if len(m) == 1]
    rng.shuffle(lone)
    k = 0
    for _ in range(200):                                                    # paralog groups among the single-transcript genes
        n = rng.choice([2, 2, 2, 3, 3, 4])
        groups.append(lone[k:k + n])
        k += n
    seq = [None] * n_tx
    share = {}                # tid -> (group id, offset of the block in the transcript, length of the block the transcript carries)
    blocks = []
    for gid, mem in enumerate(groups):
        blen = rng.randint(330, 460)
        block = rand_seq(rng, blen)
        blocks.append(block)
        for j, t in enumerate(mem):
            have = blen if (j == 0 or rng.random() < 0.7) else rng.randint(300, blen)      # some members carry a prefix of the block only
            left, right = rand_seq(rng, rng.randint(0, 160)), rand_seq(rng, rng.randint(0, 160))
            seq[t] = left + block[:have] + right
            share[t] = (gid, len(left), have)
    for t in range(n_tx):
        if seq[t] is None:
            seq[t] = rand_seq(rng, rng.randint(320, 620))
    fa = os.path.join(case_dir, "tx.fa")
    with open(fa, "w") as f:
        for n, s_ in zip(names, seq):
            f.write(">%s\n%s\n" % (n, s_))
    t0 = time.time()
    subprocess.run([REF_BUILD, "-q", "--PE", "-f", str(fmin), "-F", str(fmax), fa, str(L), case_dir, "index"], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    print("  emsar-build --PE: %.1fs" % (time.time() - t0))
    os.remove(fa)                                                            # 6 MB of random sequence: reproducible from the seed, not stored
    for f in os.listdir(case_dir):
        if f.startswith("index.") and f != "index.rsh":
            os.remove(os.path.join(case_dir, f))
    # abundances: log-normal, 30 % silent; a fragment = (transcript, start, length)
    theta = [0.0 if rng.random() < 0.3 else rng.lognormvariate(0, 2) for _ in range(n_tx)]
    tw = [theta[t] * max(0, len(seq[t]) - fmax + 1) for t in range(n_tx)]
    picks = rng.choices(range(n_tx), tw, k=n_pairs)
    by_gid = {}
    for t, (gid, off, have) in share.items():
        by_gid.setdefault(gid, []).append((t, off, have))
    lines = []
    for i, t in enumerate(picks):
        fl = rng.randint(fmin, fmax)
        p0 = rng.randint(0, len(seq[t]) - fl)
        hits = [(t, p0)]
        if t in share:                        # both mates inside the shared block: the pair maps to every member that carries that stretch
            gid, off, have = share[t]
            x = p0 - off                      # block coordinate of the fragment start
            if x >= 0 and x + fl <= have:
                hits = [(u, uoff + x) for (u, uoff, uhave) in sorted(by_gid[gid]) if x + fl <= uhave]
        flip = (i % 2 == 1)                   # the fragment came from the other strand: mate 1 is the reverse read
        for (h, q1) in hits:
            a, b = ("+", q1), ("-", q1 + fl - L)
            first, second = (a, b) if not flip else (b, a)
            lines.append("p%d/1\t%s\t%s\t%d\t%s\t%s\t0\t\n" % (i, first[0], names[h], first[1], "A" * L, "I" * L))
            lines.append("p%d/2\t%s\t%s\t%d\t%s\t%s\t0\t\n" % (i, second[0], names[h], second[1], "A" * L, "I" * L))
    aln = os.path.join(case_dir, "reads.bowtie")
    with open(aln, "w") as f:
        f.writelines(lines)
    opts = ["-P", "-p", "4"]
    cmd = run_reference(case_dir, os.path.join(case_dir, "index.rsh"), aln, opts)
    gzip_inplace(aln)
    return {"n_reads_emitted": n_pairs, "total_read_count": None, "opts": opts, "cmd": " ".join(cmd),
            "note": "Vicugna-shaped stand-in for BASELINE config 1 (bundled BAM / rsh absent): names and genes of the .g2t, synthetic sequences"}


CASES = {
    "vicugna_pe": case_vicugna_pe,
    "toy5_se50": case_toy5,
    "toy5_sam": case_toy5_sam,
    "toy5_pe": case_toy5_pe,
    "toy5_pe_sam": lambda d: case_toy5_pe(d, sam=True),
    "toy5_bam": lambda d: case_toy5_sam(d, bam=True),
    "toy5_pe_bam": lambda d: case_toy5_pe(d, sam=True, bam=True),
    "syn300_se": lambda d: synth_rsh_case(d, seed=11, n_tx=300, minfrag=40, maxfrag=44, n_reads=6000, opts=[]),
    "syn300_k2": lambda d: synth_rsh_case(d, seed=12, n_tx=300, minfrag=36, maxfrag=36, n_reads=4000,
                                          opts=["-k", "2"]),
    "syn2k_se": lambda d: synth_rsh_case(d, seed=13, n_tx=2000, minfrag=50, maxfrag=52, n_reads=40000,
                                         opts=["-p", "4"]),
}


def main():
    if not (os.path.exists(REF_EMSAR) and os.path.exists(REF_BUILD)):
        sys.exit("build the reference first: make -C oracle ref")
    todo = sys.argv[1:] or list(CASES)
    for name in todo:
        d = os.path.join(HERE, name)
        shutil.rmtree(d, ignore_errors=True)
        os.makedirs(d)
        print("case", name)
        meta = CASES[name](d)
        meta["case"] = name
        meta["runs"] = RUNS
        meta["seed"] = SEED
        meta["cmd"] = meta["cmd"].replace(ROOT + "/", "")
        with open(os.path.join(d, "meta.json"), "w") as f:
            json.dump(meta, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
