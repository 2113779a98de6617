#!/usr/bin/env python3
"""BASELINE config 4 at its own shape on ONE GPU: the -M multi-sample path over 8 single-end BAM samples of 20 M reads each of one
200 000-transcript index (`emsar_main.c:380-488`: samples strictly one after the other on the reference).

    python tests/perf/cfg4_m.py [n_tx] [n_reads_per_sample] [n_samples] [workers] [work_dir]

Writes the index and the BAMs (tests/perf/cfg4_gen.py + synth_bam.c), then runs `emsar-hip -M -B` with one worker and with
`workers` workers sharing the card (`--devices 0,0,...`) and prints, per sample, what `--stats-json` reports: parse / model / other
host work against the device's solve, the share of the wall time the GPU spends solving, and the host cores one GPU needs to be
kept busy at this sample size.  On an 8-GPU node the same command with `--devices 0,1,...,7` gives one worker per card."""
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import cfg4_gen as G

n_tx = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 20000000
n_samples = int(sys.argv[3]) if len(sys.argv) > 3 else 8
workers = int(sys.argv[4]) if len(sys.argv) > 4 else 4
work = sys.argv[5] if len(sys.argv) > 5 else tempfile.mkdtemp(prefix="cfg4_")
HIP = os.path.join(ROOT, "emsar_amd", "emsar-hip")
cores = len(os.sched_getaffinity(0))
threads = min(cores, 16)

os.makedirs(work, exist_ok=True)
t0 = time.time()
idx = G.make_index(work, n_tx)
print("index: %d transcripts, %d segments (%d multi-transcript), written in %.1f s" % (n_tx, len(idx["E"]), len(idx["E"]) - n_tx, time.time() - t0), flush=True)
bams = []
for i in range(n_samples):
    t0 = time.time()
    p = os.path.join(work, "sample%d.bam" % i)
    n_rec = G.make_sample(idx, n_reads, 40 + i, p, threads=threads)
    bams.append(p)
    print("sample %d: %d reads, %d alignment records (%.2f per read), BAM %.2f GB, written in %.1f s" % (i, n_reads, n_rec, n_rec / n_reads, os.path.getsize(p) / 1e9, time.time() - t0), flush=True)
lst = os.path.join(work, "list.txt")
open(lst, "w").write("\n".join(bams) + "\n")
rsh = os.path.join(work, "index.rsh")
print("host: %d cores visible, EMSAR_HOST_THREADS default min(cores, 16) = %d" % (cores, threads), flush=True)
extra = os.environ.get("CFG4_EXTRA", "").split()          # e.g. CFG4_EXTRA=--device-collapse
for label, dev in (("1 worker", "0"), ("%d workers on one GPU" % workers, ",".join(["0"] * workers))):
    out = os.path.join(work, "out_" + label.split()[0])
    t0 = time.time()
    subprocess.run([HIP, "-q", "-M", "-B"] + extra + ["--devices", dev, "--stats-json", os.path.join(work, "st.json"), "-I", rsh, out, "o", lst], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    wall = time.time() - t0
    st = json.load(open(os.path.join(work, "st.json")))
    ps = st["per_sample"]
    solve = sum(q.get("solve_ms", 0.0) for q in ps) / 1e3
    parse = sum(q.get("parse_s", 0.0) for q in ps)
    print("%-24s wall %.2f s for %d samples = %.2f s per sample; GPU solving %.2f s in all = %.1f %% of the wall time; parse %.2f s summed over the samples"
          % (label, wall, n_samples, wall / n_samples, solve, 100.0 * solve / wall, parse), flush=True)
    for i, q in enumerate(ps):
        print("   sample %d: parse %.2f s, model %.3f s, other host work %.3f s | solve %.3f s (%d passes, %d resident sets, %d streamed)"
              % (i, q.get("parse_s", 0.0), q.get("model_s", 0.0), q.get("host_s", 0.0) - q.get("model_s", 0.0), q.get("solve_ms", 0.0) / 1e3,
                 q.get("em_passes", 0), q.get("sets_resident", 0), q.get("sets_streamed", 0)))
    if dev == "0":
        per_sample_host = (parse + sum(q.get("host_s", 0.0) for q in ps)) / n_samples
        per_sample_dev = solve / n_samples
        print("   => per sample the host needs %.2f s on %d threads, the device %.3f s: one GPU is kept busy by about %.0f host cores at this sample size"
              % (per_sample_host, threads, per_sample_dev, threads * per_sample_host / max(per_sample_dev, 1e-9)), flush=True)
    shutil.rmtree(out, ignore_errors=True)
if len(sys.argv) <= 5:
    shutil.rmtree(work, ignore_errors=True)
