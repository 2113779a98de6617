"""Diagnostic: per-phase cycle shares of k_pass_tiled (stamped instance) on a BASELINE config."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from emsar_amd import EmsarHip, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
s = synth.make_config(cfg, scale)
dev = EmsarHip(0)
dev.upload_structure(s["n_tx"], s["row_ptr"], s["col_idx"], 3)
dev.upload_sample(None, None, s["den"])
dev.run_passes(3)
out = (C.c_double * 8)()
dev._L.emsar_hip_debug_tiled_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
rc = dev._L.emsar_hip_debug_tiled_stamps(dev._h, out)
names = ["issue loads + dictionary", "barrier 1", "E-step", "M-step", "barrier 2"]
tot = sum(out[i] for i in range(5))
print("rc", rc, "tiles", out[7], "mean cycles per wave", tot)
for i, n in enumerate(names):
    print("%-26s %10.0f  %5.1f%%" % (n, out[i], 100 * out[i] / tot))
print("ms/pass (unstamped)", dev.run_passes(20) / 20)
out = (C.c_double * 8)()
dev._L.emsar_hip_debug_unit_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
rc = dev._L.emsar_hip_debug_unit_stamps(dev._h, out)
names = ["descriptor + dictionary + loads", "barrier 1", "E-steps", "M-steps", "barrier 2", "flush"]
tot = sum(out[i] for i in range(6))
print("unit kernel: rc", rc, "units", out[7], "tiles per unit %.2f" % out[6], "mean cycles per wave", tot)
for i, n in enumerate(names):
    print("%-34s %10.0f  %5.1f%%" % (n, out[i], 100 * out[i] / tot))
