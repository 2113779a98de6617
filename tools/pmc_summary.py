#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc output: mean counter value per dispatch of the kernels whose name contains a pattern.

    python tools/pmc_summary.py <dir> [pattern]      # walks <dir> for *counter_collection.csv
"""
import csv
import os
import sys
from collections import defaultdict

root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "k_pass_tiled"
for dp, _, files in sorted(os.walk(root)):
    for f in sorted(files):
        if not f.endswith("counter_collection.csv"):
            continue
        vals = defaultdict(lambda: defaultdict(float))      # counter -> dispatch -> value (summed over dimensions)
        names = set()
        for row in csv.DictReader(open(os.path.join(dp, f))):
            if pat not in row["Kernel_Name"]:
                continue
            names.add(row["Kernel_Name"].split("(")[0][:60])
            vals[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
        tag = os.path.relpath(dp, root).split(os.sep)[0]
        for c in sorted(vals):
            v = list(vals[c].values())
            print("%-12s %-24s n=%d mean=%.4e   %s" % (tag, c, len(v), sum(v) / len(v), ",".join(sorted(names))))
