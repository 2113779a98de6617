#!/usr/bin/env python3
"""rsh text parsing (host/rsh.c, emsar_rsh_read) on a paired-end-shaped index: 100 k transcripts, 600 k segments, 151
fragment lengths per line (~390 MB of text), read with 1 / 4 / 16 host threads.  CPU only.

    python tools/rsh_parse_bench.py [n_tx] [n_multi] [nfl]
"""
import ctypes as C
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from emsar_amd import hostlib as H

n_tx = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
n_multi = int(sys.argv[2]) if len(sys.argv) > 2 else 500000
nfl = int(sys.argv[3]) if len(sys.argv) > 3 else 151

with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "big.rsh")
    rng = np.random.default_rng(1)
    base = np.arange(nfl)
    with open(path, "w") as f:
        f.write("#%d,%d,%d,%d,%d\n" % (n_tx - 1, 8, 200, 200 + nfl - 1, 101))
        f.write("".join("@%d\tENST%011d\n" % (t, t) for t in range(n_tx)))
        f.write("cid\tno.tids\tfirst.tid\tother.tids\tsegment.length\n")
        cid = 0
        for t in range(n_tx):
            f.write("%d\t1\t%d\t\t%s,\n" % (cid, t, ",".join(map(str, (int(rng.integers(300, 5000)) - base).tolist()))))
            cid += 1
        for _ in range(n_multi):
            t0 = int(rng.integers(0, n_tx - 10))
            tids = sorted(set((t0 + rng.integers(0, 10, size=int(rng.integers(2, 8)))).tolist()))
            if len(tids) < 2:
                tids = [t0, t0 + 1]
            eu = np.maximum(int(rng.integers(160, 900)) - base, 0).tolist()
            f.write("%d\t%d\t%d\t%s,\t%s,\n" % (cid, len(tids), tids[0], ",".join(map(str, tids[1:])), ",".join(map(str, eu))))
            cid += 1
    print("rsh text %.1f MB, %d rows x %d fragment lengths, %d host cores" % (os.path.getsize(path) / 1e6, n_tx + n_multi, nfl, os.cpu_count()))
    sums = set()
    for th in ("1", "4", "16"):
        os.environ["EMSAR_HOST_THREADS"] = th
        t0 = time.perf_counter()
        r = H.HostRsh(path)
        dt = time.perf_counter() - t0
        sums.add((int(r.euma.sum()), int(r.col_idx.sum()), r.euma.shape))
        print("emsar_rsh_read %2s thread(s): %.2f s  (%.0f MB/s)" % (th, dt, os.path.getsize(path) / 1e6 / dt))
        del r
    assert len(sums) == 1
