#!/usr/bin/env python3
"""Raw throughput of the parallel BGZF reader (emsar_amd/csrc/host/pbgzf.c) against zlib on one thread.  CPU only.

    python tools/pbgzf_bench.py [MB]
"""
import ctypes as C
import gzip
import os
import struct
import sys
import tempfile
import time
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from emsar_amd import hostlib as HL

mb = int(sys.argv[1]) if len(sys.argv) > 1 else 200
lib = HL.lib()
lib.emsar_pbgzf_open.restype = C.c_void_p
lib.emsar_pbgzf_open.argtypes = [C.c_char_p]
lib.emsar_pbgzf_read.restype = C.c_long
lib.emsar_pbgzf_read.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
lib.emsar_pbgzf_close.argtypes = [C.c_void_p]
rng = np.random.default_rng(0)
words = [bytes(rng.integers(65, 90, size=8, dtype=np.uint8)) for _ in range(4096)]        # ~2:1 like real BAM payloads
unit = b"".join(words[i] for i in rng.integers(0, 4096, size=4_000_000 // 8))
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "x.bgzf")
    blocks = []
    for i in range(0, len(unit), 60000):
        chunk = unit[i:i + 60000]
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        data = c.compress(chunk) + c.flush()
        blocks.append(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(data) + 25) + data +
                      struct.pack("<II", zlib.crc32(chunk), len(chunk)))
    with open(path, "wb") as fo:
        for _ in range(mb // 4):
            fo.write(b"".join(blocks))
        fo.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
    total = len(unit) * (mb // 4)
    print("%.0f MB inflated, %.0f MB on disk, %d host cores" % (total / 1e6, os.path.getsize(path) / 1e6, os.cpu_count()))
    buf = C.create_string_buffer(1 << 20)
    for th in ("1", "2", "4", "8", "16"):
        os.environ["EMSAR_HOST_THREADS"] = th
        t = time.perf_counter()
        h = lib.emsar_pbgzf_open(path.encode())
        tot = 0
        while True:
            n = lib.emsar_pbgzf_read(h, buf, 1 << 20)
            tot += n
            if n < (1 << 20):
                break
        lib.emsar_pbgzf_close(h)
        dt = time.perf_counter() - t
        assert tot == total
        print("pbgzf %2s thread(s): %.3f s  %6.0f MB/s inflated" % (th, dt, tot / dt / 1e6))
    t = time.perf_counter()
    with gzip.open(path, "rb") as g:
        while g.read(1 << 20):
            pass
    print("zlib via gzip module, one thread: %.3f s" % (time.perf_counter() - t))
