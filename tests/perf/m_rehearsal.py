#!/usr/bin/env python3
"""One-GPU rehearsal of the -M multi-sample path (BASELINE config 4 in shape: several alignment files of one index, one worker
per GPU): `emsar-hip -M --devices 0,0,...` with W workers sharing ONE card, so that what is measured is the HOST side per sample
(parse, model, output) against the device side (upload + solve) -- on 8 GPUs the device side runs 8-wide, the host side must
keep up with it.

    python tests/perf/m_rehearsal.py [n_tx] [n_reads_per_sample] [n_samples] [workers]

Prints, per sample, the times `--stats-json` reports, and the wall time of the sequential run (-M on one worker) for comparison.
"""
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden as G

n_tx = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 2000000
n_samples = int(sys.argv[3]) if len(sys.argv) > 3 else 8
workers = int(sys.argv[4]) if len(sys.argv) > 4 else 4
HIP = os.path.join(ROOT, "emsar_amd", "emsar-hip")

with tempfile.TemporaryDirectory() as d:
    G.run_reference = lambda *a, **k: ["-"]
    G.gzip_inplace = lambda p: None
    # one generated sample, listed n_samples times (a timing rehearsal: the work per sample is what matters, not its content)
    G.synth_rsh_case(d, seed=77, n_tx=n_tx, minfrag=50, maxfrag=52, n_reads=n_reads, opts=[], fam_max=24, with_quirks=False)
    rsh = os.path.join(d, "index.rsh")
    alns = [os.path.join(d, "reads.bowtie")] * n_samples
    lst = os.path.join(d, "list.txt")
    open(lst, "w").write("\n".join(alns) + "\n")
    print("input: %d transcripts, %d samples x %d reads (%.1f MB of bowtie text each)" % (n_tx, n_samples, n_reads, os.path.getsize(alns[0]) / 1e6), flush=True)
    for label, dev in (("1 worker", "0"), ("%d workers on one GPU" % workers, ",".join(["0"] * workers))):
        out = os.path.join(d, "out_" + label.split()[0])
        t0 = time.time()
        subprocess.run([HIP, "-q", "-M", "--devices", dev, "--stats-json", os.path.join(d, "st.json"), "-I", rsh, out, "o", lst],
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        wall = time.time() - t0
        st = json.load(open(os.path.join(d, "st.json")))
        ps = st["per_sample"]
        print("%-24s wall %.2f s for %d samples = %.2f s per sample" % (label, wall, n_samples, wall / n_samples))
        for i, q in enumerate(ps):
            print("   sample %d: parse %.2f s (on a thread of its own, overlapped with the sample before), model %.3f s, other host work of the sample %.3f s | solve %.3f s (%d passes, %d sets)"
                  % (i, q.get("parse_s", 0.0), q.get("model_s", 0.0), q.get("host_s", 0.0) - q.get("model_s", 0.0), q.get("solve_ms", 0.0) / 1e3, q.get("em_passes", 0), q.get("sets_resident", 0)))
        shutil.rmtree(out, ignore_errors=True)
