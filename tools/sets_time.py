"""Time of the set-resident solve on bench.py's segment-level problem (time_to_mle), without the CPU leg."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emsar_amd import EmsarHip, synth  # noqa: E402

rng = np.random.default_rng(11)
sizes = np.minimum(rng.zipf(1.6, size=40000), 60)
sizes = sizes[np.cumsum(sizes) <= 100000]
n_tx, rp, ci, _ = synth.family_matrix([int(x) for x in sizes], rows_per_tid=3, seed=11, dup=0.0)
E = rng.uniform(0.5, 2.0, size=len(rp) - 1)
theta_true = np.where(rng.random(n_tx) < 0.3, 0.0, rng.lognormal(0.0, 2.0, size=n_tx))
R = rng.poisson(E * np.add.reduceat(theta_true[ci], rp[:-1].astype(np.int64))).astype(np.int32)
with EmsarHip(0) as dev:
    dev.upload_structure(n_tx, rp, ci)
    dev.upload_sample(R, E, None)
    dev.solve(max_iter=200000, tol=1e-10)
    for kw in ({}, dict(zero_cut=2.5e-7, abs_step=1e-13)):
        t0 = time.perf_counter()
        th, st = dev.solve(max_iter=200000, tol=1e-10, **kw)
        print("%-40s %.4f s wall, sets kernel %.2f ms, slowest set %d passes, %d set-passes, loglik %.9e, converged %d"
              % (kw or "strict", time.perf_counter() - t0, st.sets_kernel_ms, st.set_passes_max, st.set_passes_sum, st.loglik, st.converged))
