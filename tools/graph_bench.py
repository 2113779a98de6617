"""Streaming solve with and without the hipGraph replay of the SQUAREM cycle (EMSAR_HIP_GRAPH), on problems small enough
that the cycle is launch-bound.  Prints microseconds per EM pass for both and checks that the two solves agree.

    python tools/graph_bench.py            # on the GPU box
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emsar_amd import EmsarHip, synth  # noqa: E402


def run(graph, prob, accel, tol, max_iter):
    os.environ["EMSAR_HIP_GRAPH"] = "1" if graph else "0"
    with EmsarHip(0) as dev:
        dev.upload_structure(prob["n_tx"], prob["row_ptr"], prob["col_idx"])
        dev.upload_sample(prob.get("R"), prob.get("E"), prob.get("den"))
        dev.solve(max_iter=48, accel=accel, tol=tol, set_mode=1)           # warm-up: code objects, allocations
        t0 = time.perf_counter()
        th, st = dev.solve(max_iter=max_iter, accel=accel, tol=tol, set_mode=1)
        wall = time.perf_counter() - t0
    return th, st, wall


def main():
    probs = []
    for n_tx, n_reads in ((2000, 40000), (20000, 400000), (100000, 2000000)):
        m = synth.make_matrix(n_tx=n_tx, n_reads=n_reads, law="human", xfam=0.02, seed=5)
        probs.append(("reads %dk x %dk" % (n_reads // 1000, n_tx // 1000), m))
    n_tx, rp, ci, R = synth.family_matrix([2, 3, 5, 8, 13, 40, 200] * 300, rows_per_tid=3, seed=3)
    E = np.random.default_rng(3).uniform(0.5, 2.0, size=len(R))
    probs.append(("segments %dk x %dk" % (len(R) // 1000, n_tx // 1000), dict(n_tx=n_tx, row_ptr=rp, col_idx=ci, R=R, E=E)))
    print("%-28s %5s %9s %12s %12s %8s" % ("problem", "accel", "passes", "launch us/p", "graph us/p", "ratio"))
    for name, m in probs:
        for accel in (1, 0):
            th0, st0, w0 = run(False, m, accel, 1e-7, 6000)
            th1, st1, w1 = run(True, m, accel, 1e-7, 6000)
            # same kernels in the same order; the acc atomics reorder sums, so compare to rounding, not bit for bit
            scale = np.maximum(np.abs(th0), 1e-6)
            print("%-28s %5d %9d %12.2f %12.2f %8.2f   F %.9e / %.9e  conv %d/%d  max rel dtheta %.1e" % (
                name, accel, st1.iters, 1e3 * st0.kernel_ms / st0.iters, 1e3 * st1.kernel_ms / st1.iters,
                st0.kernel_ms / st0.iters / (st1.kernel_ms / st1.iters), st0.loglik, st1.loglik, st0.converged, st1.converged,
                np.max(np.abs(th1 - th0) / scale)))
            sys.stdout.flush()


if __name__ == "__main__":
    main()
