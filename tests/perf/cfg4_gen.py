"""Test infrastructure: inputs of BASELINE config 4 (-M: several alignment files of ONE index, one per GPU) at any scale.

The index is a synthetic rsh in SURVEY.md 8d's family law (gene families ~ Zipf(1.6) capped at 60 isoforms; every family brings
its single-transcript segments and up to three multi-transcript segments per isoform: random isoform subsets); a sample draws every
read from the index's segments with probability E_c * sum theta (theta ~ LogNormal(0, 2), 30 % zeros, seed 40 + i) and is written
as a single-end BAM by tests/perf/synth_bam.c -- one record per transcript of the read's segment, what an aligner run with -k 100
emits.  Used by tests/perf/cfg4_m.py (full size: 200 000 transcripts, 8 x 20 M reads) and tests/test_cli_gpu.py (1/100 of it)."""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from emsar_amd import synth  # noqa: E402

READ_LEN = 50


def build_tool(out_dir):
    exe = os.path.join(out_dir, "synth_bam")
    if not os.path.exists(exe):
        subprocess.run(["gcc", "-O2", "-o", exe, os.path.join(HERE, "synth_bam.c"), "-lz", "-lpthread"], check=True)
    return exe


def make_index(out_dir, n_tx, seed=40):
    """Writes index.rsh and segments.bin under out_dir; returns dict(row_ptr, col_idx, E, n_tx)."""
    rng = np.random.default_rng(seed)
    fam_start, _ = synth.make_families(n_tx, seed)
    rows = [(t,) for t in range(n_tx)]
    for f in range(len(fam_start) - 1):
        a, n = int(fam_start[f]), int(fam_start[f + 1] - fam_start[f])
        if n < 2:
            continue
        want = min(3 * n, 2 ** n - n - 1)
        seen = set()
        for _ in range(4 * want):
            if len(seen) >= want:
                break
            k = min(n, 2 + int(rng.geometric(0.2)) - 1)
            seen.add(tuple(sorted(int(x) for x in a + rng.choice(n, size=k, replace=False))))
        rows.extend(sorted(seen, key=lambda s: (len(s), s)))
    singles, multis = rows[:n_tx], sorted(rows[n_tx:], key=lambda s: (len(s), s))
    rows = singles + multis
    E = np.concatenate([12 * rng.integers(30, 1500, size=n_tx), rng.integers(20, 900, size=len(multis))]).astype(np.int64)
    row_ptr = np.zeros(len(rows) + 1, dtype=np.uint64)
    row_ptr[1:] = np.cumsum([len(r) for r in rows])
    col_idx = np.fromiter((t for r in rows for t in r), dtype=np.int32, count=int(row_ptr[-1]))
    max_t = max(len(r) for r in rows)
    with open(os.path.join(out_dir, "index.rsh"), "w") as f:
        f.write("#%d,%d,%d,%d,%d\n" % (n_tx - 1, max_t, READ_LEN, READ_LEN, -1))
        f.write("".join("@%d\tENST%07d\n" % (i, 1000 + i) for i in range(n_tx)))
        f.write("cid\tno.tids\tfirst.tid\tother.tids\tsegment.length\n")
        out = []
        for cid, r in enumerate(rows):
            out.append("%d\t%d\t%d\t%s\t%d,\n" % (cid, len(r), r[0], "".join("%d," % x for x in r[1:]), E[cid]))
        f.write("".join(out))
    with open(os.path.join(out_dir, "segments.bin"), "wb") as f:
        np.array([n_tx, len(rows)], dtype=np.int64).tofile(f)
        row_ptr.tofile(f)
        col_idx.tofile(f)
    return {"row_ptr": row_ptr, "col_idx": col_idx, "E": E.astype(np.float64), "n_tx": n_tx, "dir": out_dir}


def make_sample(idx, n_reads, seed, bam_path, threads=16):
    """One sample of the index: reads drawn from its segments, written as BAM.  Returns the number of alignment records."""
    rng = np.random.default_rng(seed)
    theta = rng.lognormal(0.0, 2.0, idx["n_tx"])
    theta[rng.random(idx["n_tx"]) < 0.3] = 0.0
    S = np.add.reduceat(theta[idx["col_idx"]], idx["row_ptr"][:-1].astype(np.int64))
    cdf = np.cumsum(idx["E"] * S)
    cdf /= cdf[-1]
    seg = np.minimum(np.searchsorted(cdf, rng.random(n_reads), side="right"), len(cdf) - 1).astype(np.int32)
    reads_bin = bam_path + ".reads.bin"
    with open(reads_bin, "wb") as f:
        np.array([n_reads], dtype=np.int64).tofile(f)
        seg.tofile(f)
    exe = build_tool(idx["dir"])
    subprocess.run([exe, os.path.join(idx["dir"], "segments.bin"), reads_bin, bam_path, str(READ_LEN), str(threads)], check=True)
    os.remove(reads_bin)
    return int(np.diff(idx["row_ptr"].astype(np.int64))[seg].sum())
