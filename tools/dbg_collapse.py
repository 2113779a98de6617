import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from emsar_amd import EmsarHip, synth
s = synth.make_config("cfg2", 0.02)
print("rows", s["n_reads"], flush=True)
with EmsarHip(0) as dev:
    got = dev.collapse_rows(s["n_tx"], s["row_ptr"], s["col_idx"])
    print("ok", got[4].n_unique, got[4].rounds, flush=True)
