#!/usr/bin/env python3
"""Copy one tools/prof_round.sh result into profiles/ and derive its profiles/traffic.json entry.

    python tools/pmc_to_traffic.py gpurun_out/prof_<tag> <key, e.g. cfg3/family/tiled> <name under profiles/, e.g. r03a_family>

HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KB counters; FETCH_SIZE doubled: gfx950 tallies wide streaming reads at half
their bytes, MI355X_MICROARCH.md HBM section), mean over the dispatches of the pass kernel."""
import glob
import json
import os
import re
import shutil
import sys

src, key, name = sys.argv[1], sys.argv[2], sys.argv[3]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(root, "profiles")
txt = open(os.path.join(src, "pmc_summary.txt")).read()
val = {m.group(1): float(m.group(2)) for m in re.finditer(r"\s(\w+)\s+n=\d+ mean=([0-9.e+-]+)", txt)}
shutil.copy(os.path.join(src, "pmc_summary.txt"), os.path.join(prof, name + "_pmc.txt"))
shutil.copy(os.path.join(src, "bench.json"), os.path.join(prof, name + "_bench.json"))
ks = glob.glob(os.path.join(src, "kt", "**", "*kernel_stats.csv"), recursive=True)
if ks:
    shutil.copy(ks[0], os.path.join(prof, name + "_kernel_stats.csv"))
tr_path = os.path.join(prof, "traffic.json")
tr = json.load(open(tr_path))
tr[key] = {"fetch_size_kb": val["FETCH_SIZE"], "write_size_kb": val["WRITE_SIZE"], "tcc_ea0_atomic": val.get("TCC_EA0_ATOMIC_sum"),
           "hbm_bytes_per_launch": int((2 * val["FETCH_SIZE"] + val["WRITE_SIZE"]) * 1024),
           "profile": "profiles/%s_pmc.txt" % name,
           "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE TCC_EA0_ATOMIC_sum in separate passes (tools/prof_round.sh), mean over the "
                   "dispatches of the pass kernel; FETCH_SIZE doubled (gfx950 correction, MI355X_MICROARCH.md HBM section)"}
json.dump(tr, open(tr_path, "w"), indent=1)
print(key, tr[key]["hbm_bytes_per_launch"], "SQ_WAIT_ANY/SQ_WAVE_CYCLES %.2f" % (val.get("SQ_WAIT_ANY", 0) / max(val.get("SQ_WAVE_CYCLES", 1), 1)),
      "LDS conflict share %.2f" % (val.get("SQ_LDS_BANK_CONFLICT", 0) / max(val.get("SQ_LDS_IDX_ACTIVE", 1), 1)))
