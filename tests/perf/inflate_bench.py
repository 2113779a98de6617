#!/usr/bin/env python3
"""CPU: what the BGZF inflate engine (emsar_amd/csrc/host/pbgzf.c: libdeflate when the system has it, else zlib) costs on a BAM of
BASELINE config 4's shape, and what the whole record count (inflate + record walk + per-read filters + collapse) takes with either.

    python tests/perf/inflate_bench.py [n_tx] [n_reads] [threads]

Writes a synthetic index and one single-end BAM (tests/perf/cfg4_gen.py), then in child processes (the engine is chosen once per
process) reads the file through emsar_pbgzf_read alone and through HostRsh.count."""
import json
import os
import subprocess
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import cfg4_gen as G

n_tx = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 2000000
threads = sys.argv[3] if len(sys.argv) > 3 else str(min(len(os.sched_getaffinity(0)), 16))
work = tempfile.mkdtemp(prefix="inflate_")
idx = G.make_index(work, n_tx)
bam = os.path.join(work, "s.bam")
n_rec = G.make_sample(idx, n_reads, 40, bam, threads=int(threads))
print("BAM: %d reads, %d records, %.1f MB" % (n_reads, n_rec, os.path.getsize(bam) / 1e6), flush=True)
code = r'''
import sys, time, json, ctypes as C
sys.path.insert(0, %r)
from emsar_amd import hostlib as HL
lib = HL.lib()
lib.emsar_pbgzf_engine.restype = C.c_char_p
lib.emsar_pbgzf_open.restype = C.c_void_p; lib.emsar_pbgzf_open.argtypes = [C.c_char_p]
lib.emsar_pbgzf_read.restype = C.c_long; lib.emsar_pbgzf_read.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
lib.emsar_pbgzf_close.argtypes = [C.c_void_p]
buf = C.create_string_buffer(32 << 20)
best = 1e9
for _ in range(3):
    t0 = time.time(); p = lib.emsar_pbgzf_open(%r.encode()); n = 0
    while True:
        g = lib.emsar_pbgzf_read(p, buf, len(buf))
        if g <= 0: break
        n += g
    lib.emsar_pbgzf_close(p); best = min(best, time.time() - t0)
r = HL.HostRsh(%r)
t0 = time.time(); c = r.count(%r, fmt=2); tc = time.time() - t0
print(json.dumps({"engine": lib.emsar_pbgzf_engine().decode(), "inflated_bytes": n, "inflate_s": best, "count_s": tc, "total_reads": int(c.total_reads)}))
''' % (ROOT, bam, os.path.join(work, "index.rsh"), bam)
for eng in ("zlib", ""):
    env = dict(os.environ, EMSAR_HOST_INFLATE=eng, EMSAR_HOST_THREADS=threads)
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    if p.returncode != 0:
        print(p.stderr); sys.exit(1)
    o = json.loads(p.stdout.strip().splitlines()[-1])
    print("%-10s %s threads: stream inflated in %.3f s = %.0f MB/s (%.0f ns per record); whole count %.3f s = %.0f ns per record"
          % (o["engine"], threads, o["inflate_s"], o["inflated_bytes"] / o["inflate_s"] / 1e6, 1e9 * o["inflate_s"] / n_rec, o["count_s"], 1e9 * o["count_s"] / n_rec), flush=True)
import shutil
shutil.rmtree(work, ignore_errors=True)
