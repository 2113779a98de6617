#!/usr/bin/env python3
"""CPU: gzipped bowtie text and paired-end SAM text through count_text_parallel (one reader + a pool of counters) against the
one-thread loop.   python tests/perf/text_pool_bench.py [n_reads] [threads]"""
import gzip
import os
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
import cfg4_gen as G
from emsar_amd import hostlib as HL

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
threads = sys.argv[2] if len(sys.argv) > 2 else str(min(len(os.sched_getaffinity(0)), 16))
work = tempfile.mkdtemp(prefix="textpool_")
idx = G.make_index(work, 20000)
rng = np.random.default_rng(5)
rp, ci = idx["row_ptr"].astype(np.int64), idx["col_idx"]
seg = rng.integers(0, len(rp) - 1, n_reads)
names = ["ENST%07d" % (1000 + i) for i in range(idx["n_tx"])]
se = os.path.join(work, "se.bowtie.gz")
pe = os.path.join(work, "pe.sam")
L = G.READ_LEN
with gzip.open(se, "wt", compresslevel=1) as f, open(pe, "w") as g:
    g.write("".join("@SQ\tSN:%s\tLN:100000\n" % n for n in names))
    seq, q = "A" * L, "I" * L
    for i, s in enumerate(seg):
        for t in ci[rp[s]:rp[s + 1]]:
            f.write("r%d\t+\t%s\t%d\t%s\t%s\t0\t\n" % (i, names[t], 100 + i % 50, seq, q))
            g.write("r%d\t99\t%s\t%d\t255\t%dM\t=\t%d\t0\t%s\t%s\tMD:Z:%d\n" % (i, names[t], 101, L, 201, seq, q, L))
            g.write("r%d\t147\t%s\t%d\t255\t%dM\t=\t%d\t0\t%s\t%s\tMD:Z:%d\n" % (i, names[t], 201, L, 101, seq, q, L))
r = HL.HostRsh(os.path.join(work, "index.rsh"))
for what, path, kw in (("gzipped bowtie, single-end", se, {}), ("SAM text, paired-end", pe, dict(pe=1, fmt=1))):
    res = {}
    for th in ("1", threads):
        os.environ["EMSAR_HOST_THREADS"] = th
        t0 = time.time()
        try:
            c = r.count(path, **kw)
            res[th] = (time.time() - t0, int(c.total_reads), c.R.tobytes())
        except HL.HostError as e:
            res[th] = (time.time() - t0, str(e), None)
    a, b = res["1"], res[threads]
    print("%-28s %.1f MB: 1 thread %.2f s, %s threads %.2f s (x%.2f); same counts: %s (%s reads)" %
          (what, os.path.getsize(path) / 1e6, a[0], threads, b[0], a[0] / b[0], a[1:] == b[1:], a[1]), flush=True)
import shutil
shutil.rmtree(work, ignore_errors=True)
