#!/usr/bin/env python3
"""CPU prototype: pass counts of global SQUAREM (likelihood safeguard) vs per-set SQUAREM (one step length per
connected set, residual safeguard) on a synthetic segment-level problem.  numpy/scipy only; no GPU.

    python tools/sqblock_proto.py [n_tx] [n_reads] [largest_family] [tol]
"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import scipy.sparse as sp

import make_golden as G
from emsar_amd import hostlib as H

n_tx = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 400000
fam_max = int(sys.argv[3]) if len(sys.argv) > 3 else 40
tol = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-10
floor = 1e-6

with tempfile.TemporaryDirectory() as d:
    G.run_reference = lambda *a, **k: ["-"]
    G.gzip_inplace = lambda p: None
    G.synth_rsh_case(d, seed=77, n_tx=n_tx, minfrag=50, maxfrag=52, n_reads=n_reads, opts=[], fam_max=fam_max, with_quirks=False)
    rsh = H.HostRsh(os.path.join(d, "index.rsh"))
    cnt = rsh.count(os.path.join(d, "reads.bowtie"))
    mdl = rsh.model(cnt)
    rp, ci = rsh.row_ptr.astype(np.int64), rsh.col_idx.copy()
    R = cnt.R.astype(np.float64).copy()
    E = mdl.E_solver.copy()
    TS = mdl.TS.copy()
    n_sets = mdl.n_sets

rows = np.repeat(np.arange(len(rp) - 1), np.diff(rp))
A = sp.csr_matrix((np.ones(len(ci)), (rows, ci)), shape=(len(rp) - 1, n_tx))
A = A[E > 0]
Rl, El = R[E > 0], E[E > 0]
AT = A.T.tocsr()
den = AT @ El
live = den > 0
print("rows %d  nnz %d  sets %d  largest set %d" % (A.shape[0], A.nnz, n_sets, np.bincount(TS[TS >= 0]).max()))


def em(th):
    S = A @ th
    w = np.where(S > 0, Rl / np.where(S > 0, S, 1), 0.0)
    acc = AT @ w
    return np.where(live, th * acc / np.where(live, den, 1), 0.0), S


def F(th, S=None):
    if S is None:
        S = A @ th
    m = (S > 0) & (Rl > 0)
    return float((Rl[m] * np.log(S[m])).sum() - (th * den).sum())


def delta(a, b):
    return float((np.abs(b - a) / (np.abs(b) + floor)).max())


def run_global():
    th0 = live.astype(float)
    sm, passes = 1.0, 0
    while passes < 400000:
        th1, S0 = em(th0)
        if delta(th0, th1) < tol:
            return passes + 1, th1
        th2, S1 = em(th1)
        r, v = th1 - th0, (th2 - th1) - (th1 - th0)
        s = np.sqrt((r * r).sum() / (v * v).sum()) if (v * v).sum() > 0 else 1.0
        s = min(max(s, 1.0), sm)
        ex = s > 1.01
        thx = th2
        if ex:
            y = th0 + 2 * s * r + s * s * v
            thx = np.where((y > 0) & (th2 > 0), y, th2)
        thn, Sx = em(thx)
        ok = (not ex) or F(thx, Sx) >= F(th1, S1)
        passes += 3
        if not ok and s >= sm:
            sm = max(1.0, sm / 4)
        if (s if ok else 1.0) >= sm:
            sm *= 4
        th0 = thn if ok else th2
    return passes, th0


def run_block(kres=1.0):
    key = np.where(TS >= 0, TS, n_sets)          # transcripts in no set share a dummy block
    nb = n_sets + 1
    th0 = live.astype(float)
    sm = np.ones(nb)
    passes = 0
    seg = lambda x: np.bincount(key, weights=x, minlength=nb)
    while passes < 400000:
        th1, _ = em(th0)
        if delta(th0, th1) < tol:
            return passes + 1, th1
        th2, _ = em(th1)
        r, v = th1 - th0, (th2 - th1) - (th1 - th0)
        sr2, sv2 = seg(r * r), seg(v * v)
        s = np.where(sv2 > 0, np.sqrt(sr2 / np.where(sv2 > 0, sv2, 1)), 1.0)
        s = np.minimum(np.maximum(s, 1.0), sm)
        ex = s > 1.01
        st = s[key]
        y = th0 + 2 * st * r + st * st * v
        thx = np.where(ex[key] & (y > 0) & (th2 > 0), y, th2)
        thn, _ = em(thx)
        d = thn - thx
        res = seg(d * d)
        ok = (~ex) | (res <= (1 + kres) ** 2 * sr2)
        passes += 3
        shrink = (~ok) & (s >= sm)
        sm = np.where(shrink, np.maximum(1.0, sm / 4), sm)
        grow = np.where(ok, s, 1.0) >= sm
        sm = np.where(grow, sm * 4, sm)
        th0 = np.where(ok[key], thn, th2)
    return passes, th0


pg, tg = run_global()
print("global SQUAREM : %6d passes  F = %.9f" % (pg, F(tg)))
for k in (1.0, 0.0):
    pb, tb = run_block(k)
    print("per-set (kres=%g): %6d passes  F = %.9f  max|dtheta| vs global %.3e (rel+1.5e-6 fails: %d)"
          % (k, pb, F(tb), np.abs(tb - tg).max(), int((np.abs(tb - tg) > 1e-5 * np.abs(tg) + 1.5e-6).sum())))
