"""Diagnostic: delta trajectory of plain EM on the device at full size; which transcript dominates the max-norm."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from emsar_amd import EmsarHip, synth
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
s = synth.make_config("cfg3", scale)
den = s["den"]
dev = EmsarHip(0)
dev.upload_structure(s["n_tx"], s["row_ptr"], s["col_idx"], 3)
dev.upload_sample(None, None, den)
done = 0
for n in (100, 400, 500, 1000, 2000, 4000, 8000):
    dev.run_passes(n - 1); a = dev.get_theta(); dev.run_passes(1); b = dev.get_theta(); done += n
    d = np.abs(b - a) / (np.abs(b) + 1e-6)
    t = int(np.argmax(d))
    print("passes", done, "delta %.3g" % d.max(), "argmax t", t, "theta %.6g -> %.6g" % (a[t], b[t]), "den %.4g" % den[t],
          "n(d>1e-6)", int((d > 1e-6).sum()), "n(d>1e-8)", int((d > 1e-8).sum()), flush=True)
th, st = dev.solve(max_iter=3000, accel=1, tol=1e-6, check_every=1)
print("squarem 3000:", st.iters, st.converged, st.final_delta, flush=True)
