#!/usr/bin/env python3
"""BAM ingestion with the parallel BGZF reader: the same synthetic reads as default-bowtie text and as BAM, counted by
emsar_count_alignments with 1 and N host threads (BGZF inflate pool + record-counting pool).  CPU only.

    python tools/bam_parse_bench.py [n_tx] [n_reads] [largest_family]
"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np

import make_golden as G
from emsar_amd import hostlib as H

n_tx = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 150000
fam_max = int(sys.argv[3]) if len(sys.argv) > 3 else 24

with tempfile.TemporaryDirectory() as d:
    G.run_reference = lambda *a, **k: ["-"]
    G.gzip_inplace = lambda p: None
    G.synth_rsh_case(d, seed=5, n_tx=n_tx, minfrag=50, maxfrag=52, n_reads=n_reads, opts=[], fam_max=fam_max, with_quirks=False)
    rsh = H.HostRsh(os.path.join(d, "index.rsh"))
    txt = os.path.join(d, "reads.bowtie")
    sam, bam = os.path.join(d, "r.sam"), os.path.join(d, "r.bam")
    with open(sam, "w") as f:
        for n in rsh.names:
            f.write("@SQ\tSN:%s\tLN:100000\n" % n)
        for line in open(txt):
            q = line.rstrip("\n").split("\t")
            f.write(G.sam_line(q[0], 0 if q[1] == "+" else 16, q[2], int(q[3]), len(q[4]), str(len(q[4]))))
    G.sam_to_bam(sam, bam)
    print("reads %d, alignments text %.1f MB, BAM %.1f MB" % (n_reads, os.path.getsize(txt) / 1e6, os.path.getsize(bam) / 1e6))
    t0 = time.perf_counter(); a = rsh.count(txt); t_txt = time.perf_counter() - t0
    res = {}
    for th in ("1", "4", str(min(16, os.cpu_count() or 1))):
        os.environ["EMSAR_HOST_THREADS"] = th
        t0 = time.perf_counter(); b = rsh.count(bam, fmt=2); res[th] = time.perf_counter() - t0
        assert np.array_equal(a.R, b.R) and a.total_reads == b.total_reads
    print("count_alignments: bowtie text %.2f s; BAM " % t_txt + ", ".join("%s thread(s) %.2f s" % kv for kv in res.items()))
