#!/usr/bin/env python3
"""Read -> segment collapse of a BASELINE-shaped read-level matrix: device (emsar_hip_collapse_rows) against the oracle's
restatement of update_ReadCounts on one host core.  Prints the kernel time, the algorithmic bytes and the HBM rate.

    python tests/perf/collapse_bench.py [config] [scale] [structure: window|family|family_shuffled]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

import oracle as O
from emsar_amd import EmsarHip, synth

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 0.2
structure = sys.argv[3] if len(sys.argv) > 3 else "window"
s = synth.make_config(cfg, scale, structure)
print("%s x %.2f (%s): %d reads, %d transcripts, nnz %d" % (cfg, scale, structure, s["n_reads"], s["n_tx"], len(s["col_idx"])), flush=True)
dev = EmsarHip(0)
dev.collapse_rows(s["n_tx"], s["row_ptr"][:1001], s["col_idx"][:int(s["row_ptr"][1000])])       # warm up
t0 = time.perf_counter()
rp, ci, w, m, st = dev.collapse_rows(s["n_tx"], s["row_ptr"], s["col_idx"])
t_dev = time.perf_counter() - t0
_, _, _, _, st_nomap = dev.collapse_rows(s["n_tx"], s["row_ptr"], s["col_idx"], want_map=False)
print("without the row map (what the CLI asks for): kernels %.2f ms; rounds %d, LDS table slots %d" % (st_nomap.kernel_ms, st_nomap.rounds, st_nomap.table_slots), flush=True)
print("device: %d segments (nnz %d); kernels %.2f ms = %.0f GB/s of algorithmic bytes (%.2f GB); call incl. PCIe both ways %.2f s"
      % (st.n_unique, st.nnz_unique, st.kernel_ms, st.algorithmic_bytes / st.kernel_ms / 1e6, st.algorithmic_bytes / 1e9, t_dev), flush=True)
t0 = time.perf_counter()
want = O.collapse_rows(s["row_ptr"], s["col_idx"])
t_cpu = time.perf_counter() - t0
same = all(np.array_equal(a, b) for a, b in zip((rp, ci, w.astype(np.int64), m), want))
print("oracle (update_ReadCounts restated, 1 host core): %.2f s; identical output: %s; kernel speed-up %.0fx, whole call %.1fx"
      % (t_cpu, same, t_cpu * 1e3 / st.kernel_ms, t_cpu / t_dev))
dev.close()
