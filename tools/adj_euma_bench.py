#!/usr/bin/env python3
"""compute_adjEUMA on the device at human paired-end size (10^6 segments x 401 fragment lengths = 1.6 GB of EUMA):
host loop vs emsar_hip_adj_euma.  Run under `rocprofv3 --kernel-trace --stats` for the kernel's own time.

    python tools/adj_euma_bench.py [n_rows] [nfl]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from emsar_amd import EmsarHip

n_rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
nfl = int(sys.argv[2]) if len(sys.argv) > 2 else 401
rng = np.random.default_rng(0)
euma = rng.integers(0, 3000, size=(n_rows, nfl), dtype=np.int32)
wf = rng.random(nfl)
wf /= wf.sum()
dev = EmsarHip(0)
dev.upload_structure(4, np.arange(n_rows + 1, dtype=np.uint64), np.zeros(n_rows, dtype=np.int32), 1)
t0 = time.perf_counter()
dev.upload_euma(euma)
t_up = time.perf_counter() - t0
dev.adj_euma(wf)
t0 = time.perf_counter()
for _ in range(10):
    L = dev.adj_euma(wf)
t_dev = (time.perf_counter() - t0) / 10
t0 = time.perf_counter()
want = np.zeros(n_rows)
for i in range(nfl):
    want = want + wf[i] * euma[:, i].astype(np.float64)
t_host = time.perf_counter() - t0
print("EUMA %d x %d = %.2f GB; upload + transpose %.3f s (once per rsh)" % (n_rows, nfl, euma.nbytes / 1e9, t_up))
print("adj_euma per sample: device call %.3f ms incl. the %.1f MB copy back (%.0f GB/s of EUMA); numpy on the host %.2f s"
      % (t_dev * 1e3, n_rows * 8 / 1e6, euma.nbytes / t_dev / 1e9, t_host))
print("bit-identical to the host's order of operations:", bool(np.array_equal(L, want)))
dev.close()
