"""Diagnostic: convergence of the device solver vs the CPU oracle on a scaled BASELINE config."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from emsar_amd import EmsarHip, synth
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.01
s = synth.make_config("cfg3", scale)
dev = EmsarHip(0)
for layout in (1, 2, 3):
    dev.upload_structure(s["n_tx"], s["row_ptr"], s["col_idx"], layout)
    dev.upload_sample(None, None, s["den"])
    for accel in (0, 1):
        for tol in (1e-6, 1e-9):
            t0 = time.time()
            th, st = dev.solve(max_iter=20000, accel=accel, tol=tol, check_every=4)
            print("layout", layout, "accel", accel, "tol", tol, "passes", st.iters, "conv", st.converged, "delta %.3g" % st.final_delta,
                  "F %.6f" % st.loglik, "%.2fs" % (time.time() - t0), flush=True)
