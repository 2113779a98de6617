"""Latency of the streaming pass on small problems: us per EM pass for the tile knobs given in the environment.

    EMSAR_HIP_TILED_MULTI=0 EMSAR_HIP_TILE_ROWS=768 python tools/small_prof.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emsar_amd import EmsarHip, synth  # noqa: E402

probs = []
for n_tx, n_reads in ((2000, 40000), (20000, 400000), (100000, 2000000)):
    probs.append(("reads %dk x %dk" % (n_reads // 1000, n_tx // 1000), synth.make_matrix(n_tx=n_tx, n_reads=n_reads, law="human", xfam=0.02, seed=5)))
n_tx, rp, ci, R = synth.family_matrix([2, 3, 5, 8, 13, 40, 200] * 300, rows_per_tid=3, seed=3)
E = np.random.default_rng(3).uniform(0.5, 2.0, size=len(R))
probs.append(("segments %dk x %dk" % (len(R) // 1000, n_tx // 1000), dict(n_tx=n_tx, row_ptr=rp, col_idx=ci, R=R, E=E)))
n_tx, rp, ci, R = synth.family_matrix([3000, 1500, 800], rows_per_tid=3, seed=4)
E = np.random.default_rng(4).uniform(0.5, 2.0, size=len(R))
probs.append(("3 big sets %dk x %dk" % (len(R) // 1000, n_tx // 1000), dict(n_tx=n_tx, row_ptr=rp, col_idx=ci, R=R, E=E)))
knobs = " ".join("%s=%s" % (k[10:], v) for k, v in sorted(os.environ.items()) if k.startswith("EMSAR_HIP_"))
for name, m in probs:
    with EmsarHip(0) as dev:
        dev.upload_structure(m["n_tx"], m["row_ptr"], m["col_idx"])
        dev.upload_sample(m.get("R"), m.get("E"), m.get("den"))
        info = dev.info()
        dev.solve(max_iter=48, accel=1, tol=1e-12, set_mode=1)
        th, st = dev.solve(max_iter=1200, accel=1, tol=1e-12, set_mode=1)
        ms = dev.run_passes(200)
    print("[%s] %-24s tiles %5d  solve %7.2f us/pass  plain pass %7.2f us  F %.10e" % (knobs, name, info["n_chunks"], 1e3 * st.kernel_ms / st.iters, 1e3 * ms / 200, st.loglik))
    sys.stdout.flush()
