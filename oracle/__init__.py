"""CPU ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes front-end for oracle/liboracle.so (em_oracle.c: our plain-C restatement of the reference's
abundance path, /root/reference/src/emsar_functions.c:2946-3232) plus readers for the files the compiled
reference writes (.fpkm / .segments), used to pin the restatement against tests/golden/.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
Nothing under emsar_amd/ imports it; the product fails loudly without its HIP library instead.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    """Compile liboracle.so (and oracle/_ref when /root/reference is present)."""
    so = os.path.join(_HERE, "liboracle.so")
    src = [os.path.join(_HERE, f) for f in ("em_oracle.c", "em_oracle.h")]
    stale = force or not os.path.exists(so) or any(
        os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(so) for s in src)
    if stale:
        subprocess.run(["make", "-C", _HERE, "liboracle.so"] + (["-B"] if force else []), check=True,
                       stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/src") and not os.path.exists(os.path.join(_HERE, "_ref", "emsar")):
        subprocess.run(["make", "-C", _HERE, "ref"], check=True, stdout=subprocess.DEVNULL)
    return so


class EmParams(C.Structure):
    _fields_ = [("max_iter", C.c_int32), ("accel", C.c_int32), ("tol", C.c_double),
                ("abs_floor", C.c_double), ("n_threads", C.c_int32)]


class EmStats(C.Structure):
    _fields_ = [("iters", C.c_int32), ("converged", C.c_int32), ("final_delta", C.c_double),
                ("loglik", C.c_double), ("seconds", C.c_double)]


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        u64p, i32p, f64p = C.POINTER(C.c_uint64), C.POINTER(C.c_int32), C.POINTER(C.c_double)
        csr = [C.c_int64, C.c_int32, u64p, i32p]
        L.oracle_loglik.restype = C.c_double
        L.oracle_loglik.argtypes = [C.c_int64, u64p, i32p, i32p, f64p, f64p]
        L.oracle_den.argtypes = csr + [f64p, f64p]
        L.oracle_ieuma.argtypes = csr + [f64p, f64p]
        L.oracle_em_step.restype = C.c_double
        L.oracle_em_step.argtypes = csr + [i32p, f64p, f64p, f64p, f64p, C.c_int]
        L.oracle_em_solve.restype = C.c_int
        L.oracle_em_solve.argtypes = csr + [i32p, f64p, C.POINTER(EmParams), f64p, C.POINTER(EmStats)]
        L.oracle_components.restype = C.c_int32
        L.oracle_components.argtypes = csr + [f64p, f64p, C.c_int32, i32p, i32p]
        L.oracle_mle_pattern_search.restype = C.c_int64
        L.oracle_mle_pattern_search.argtypes = csr + [i32p, f64p, i32p, C.c_int32, C.c_double, C.c_double,
                                                      C.c_int32, C.c_int32, C.c_uint32, C.c_int32, f64p]
        L.oracle_collapse_rows.restype = C.c_int64
        L.oracle_collapse_rows.argtypes = [C.c_int64, u64p, i32p, i32p, u64p, i32p, C.POINTER(C.c_int64), i32p]
        L.oracle_fpkm_table.argtypes = [C.c_int32, C.c_int32, f64p, f64p, C.c_int64, f64p, f64p, f64p, i32p, f64p]
        _LIB = L
    return _LIB


def _p(a, ct):
    return None if a is None else a.ctypes.data_as(C.POINTER(ct))


class Csr:
    """Flat segment->transcript (or read->transcript) incidence with per-row R, L/E."""

    def __init__(self, n_tx, row_ptr, col_idx, R=None, E=None, L=None):
        self.n_tx = int(n_tx)
        self.row_ptr = np.ascontiguousarray(row_ptr, dtype=np.uint64)
        self.col_idx = np.ascontiguousarray(col_idx, dtype=np.int32)
        self.n_rows = len(self.row_ptr) - 1
        self.R = None if R is None else np.ascontiguousarray(R, dtype=np.int32)
        self.E = np.ones(self.n_rows) if E is None else np.ascontiguousarray(E, dtype=np.float64)
        self.L = None if L is None else np.ascontiguousarray(L, dtype=np.float64)

    def _csr(self):
        return (self.n_rows, self.n_tx, _p(self.row_ptr, C.c_uint64), _p(self.col_idx, C.c_int32))

    def loglik(self, theta):
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        return lib().oracle_loglik(self.n_rows, _p(self.row_ptr, C.c_uint64), _p(self.col_idx, C.c_int32),
                                   _p(self.R, C.c_int32), _p(self.E, C.c_double), _p(theta, C.c_double))

    def den(self):
        d = np.zeros(self.n_tx)
        lib().oracle_den(*self._csr(), _p(self.E, C.c_double), _p(d, C.c_double))
        return d

    def ieuma(self):
        d = np.zeros(self.n_tx)
        lib().oracle_ieuma(*self._csr(), _p(self.L, C.c_double), _p(d, C.c_double))
        return d

    def em_step(self, theta, den=None, n_threads=1):
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        den = self.den() if den is None else den
        out = np.zeros(self.n_tx)
        ll = lib().oracle_em_step(*self._csr(), _p(self.R, C.c_int32), _p(self.E, C.c_double), _p(den, C.c_double),
                                  _p(theta, C.c_double), _p(out, C.c_double), n_threads)
        return out, ll

    def em_solve(self, max_iter=100000, accel=1, tol=1e-10, abs_floor=1e-6, n_threads=1):
        p = EmParams(max_iter, accel, tol, abs_floor, n_threads)
        st = EmStats()
        out = np.zeros(self.n_tx)
        rc = lib().oracle_em_solve(*self._csr(), _p(self.R, C.c_int32), _p(self.E, C.c_double), C.byref(p),
                                   _p(out, C.c_double), C.byref(st))
        if rc != 0:
            raise MemoryError("oracle_em_solve")
        return out, st

    def components(self, eumacut=0.0, max_ntid=5000):
        cs = np.zeros(self.n_rows, dtype=np.int32)
        ts = np.zeros(self.n_tx, dtype=np.int32)
        cut = C.c_double(eumacut)
        n = lib().oracle_components(*self._csr(), _p(self.L, C.c_double), C.byref(cut), max_ntid,
                                    _p(cs, C.c_int32), _p(ts, C.c_int32))
        return n, cs, ts, cut.value

    def mle_pattern_search(self, cs, n_sets, seed, n_threads=1, eps=1e-9, eps_step=1e-15, max_niter=200000,
                           max_nloop=100):
        out = np.zeros(self.n_tx)
        sweeps = lib().oracle_mle_pattern_search(*self._csr(), _p(self.R, C.c_int32), _p(self.E, C.c_double),
                                                 _p(cs, C.c_int32), n_sets, eps, eps_step, max_niter, max_nloop,
                                                 seed, n_threads, _p(out, C.c_double))
        return out, sweeps


def collapse_rows(row_ptr, col_idx, row_weight=None):
    """Read -> segment collapse as update_ReadCounts does it: (row_ptr, col_idx, weight[int64], row_map)."""
    row_ptr = np.ascontiguousarray(row_ptr, dtype=np.uint64)
    col_idx = np.ascontiguousarray(col_idx, dtype=np.int32)
    n = len(row_ptr) - 1
    w = None if row_weight is None else np.ascontiguousarray(row_weight, dtype=np.int32)
    rp_o = np.zeros(n + 1, dtype=np.uint64)
    ci_o = np.zeros(max(len(col_idx), 1), dtype=np.int32)
    w_o = np.zeros(max(n, 1), dtype=np.int64)
    m_o = np.zeros(max(n, 1), dtype=np.int32)
    u = lib().oracle_collapse_rows(n, _p(row_ptr, C.c_uint64), _p(col_idx, C.c_int32), None if w is None else _p(w, C.c_int32),
                                   _p(rp_o, C.c_uint64), _p(ci_o, C.c_int32), _p(w_o, C.c_int64), _p(m_o, C.c_int32))
    if u < 0:
        raise MemoryError("oracle_collapse_rows")
    return rp_o[:u + 1], ci_o[:int(rp_o[u])], w_o[:u], m_o[:n]


def fpkm_table(rounds, ieuma, total_read_count):
    rounds = np.ascontiguousarray(np.atleast_2d(rounds), dtype=np.float64)
    n_round, n_tx = rounds.shape
    mean, sd, ir, tpm = (np.zeros(n_tx) for _ in range(4))
    iri = np.zeros(n_tx, dtype=np.int32)
    ieuma = np.ascontiguousarray(ieuma, dtype=np.float64)
    lib().oracle_fpkm_table(n_tx, n_round, _p(rounds, C.c_double), _p(ieuma, C.c_double), int(total_read_count),
                            _p(mean, C.c_double), _p(sd, C.c_double), _p(ir, C.c_double), _p(iri, C.c_int32),
                            _p(tpm, C.c_double))
    return mean, sd, ir, iri, tpm


# ----------------------------------------------------------------------------------------------------------
# readers for the reference's output files (formats: emsar_functions.c:3184-3207 and 2274-2297)
# ----------------------------------------------------------------------------------------------------------
def read_fpkm(path):
    names, cols = [], []
    with open(path) as f:
        header = f.readline().rstrip("\n").split("\t")
        assert header == ["transcriptID", "FPKM", "sd.of.FPKM", "eff.length", "iReadcount", "iReadcount.int", "TPM"]
        for line in f:
            p = line.rstrip("\n").split("\t")
            names.append(p[0])
            cols.append([float(x) for x in p[1:]])
    a = np.array(cols)
    return {"names": names, "fpkm": a[:, 0], "sd": a[:, 1], "efflen": a[:, 2], "ireadcount": a[:, 3],
            "ireadcount_int": a[:, 4].astype(np.int64), "tpm": a[:, 5]}


def read_segments(path, n_tx):
    """.segments (-g) -> Csr with L (eff.length) and R, plus the set ids and expected counts."""
    row_ptr, col, L, R, cs, expc = [0], [], [], [], [], []
    with open(path) as f:
        f.readline()
        for i, line in enumerate(f):
            p = line.rstrip("\n").split("\t")
            assert p[0] == "c%d" % i
            cs.append(int(p[1][1:]))
            col.extend(int(x[1:]) for x in p[2].split(","))
            row_ptr.append(len(col))
            L.append(float(p[4]))
            R.append(int(p[5]))
            expc.append(float(p[6]))
    m = Csr(n_tx, row_ptr, col, R=R, L=L)
    return m, np.array(cs, dtype=np.int32), np.array(expc)


def read_fraglength_effect(path):
    """.fraglength_effect (emsar_functions.c:2489-2490) -> (lengths, observed counts)."""
    lens, cnt = [], []
    with open(path) as f:
        f.readline()
        for line in f:
            p = line.split("\t")
            lens.append(int(p[0]))
            cnt.append(int(p[1]))
    return np.array(lens), np.array(cnt, dtype=np.int64)


def read_rsh(path):
    """rsh text (written by emsar_functions.c:2085-2127, read by 1351-1510) -> names, rows in cid order.

    cid order at scan time (emsar_functions.c:2149-2191): one single-tid row per tid (EUMA None when the
    transcript has no unique region), then the multi-tid rows in file order (the file is written by size,
    first tid, list order)."""
    names, single, multi = {}, {}, []
    hdr = None
    with open(path) as f:
        for line in f:
            line = line.rstrip("\n")
            if line.startswith("#"):
                hdr = [int(x) for x in line[1:].split(",")]
            elif line.startswith("@"):
                tid, name = line[1:].split("\t")
                names[int(tid)] = name
            elif line.startswith("c"):
                continue
            elif line:
                p = line.split("\t")
                size, tid0 = int(p[1]), int(p[2])
                euma = [int(x) for x in p[4].split(",") if x != ""] if len(p) > 4 else []
                if size == 1:
                    single[tid0] = euma if euma else None
                elif euma:
                    multi.append(([tid0] + [int(x) for x in p[3].split(",") if x != ""], euma))
    max_tid, max_t, minfrag, maxfrag, readlen = hdr
    rows = [([t], single.get(t)) for t in range(max_tid + 1)] + multi
    return {"names": [names[t] for t in range(max_tid + 1)], "rows": rows, "minfrag": minfrag,
            "maxfrag": maxfrag, "readlength": readlen}


def model_from_fixture(rsh, frag_counts, R, total_read_count, delta=0):
    """Exact restatement of transfer_fraglendist_to_Wf + compute_adjEUMA + construct_EUMAps
    (emsar_functions.c:2503-2523, 3148-3154) on parsed fixture data: returns Csr with L and E in full
    double precision (the .segments file prints L with 6 decimals only)."""
    nfl = rsh["maxfrag"] - rsh["minfrag"] + 1
    wf = [float(c) for c in frag_counts[:nfl]]
    s = 0.0
    for x in wf:
        s += x
    wf = [x / s for x in wf]
    row_ptr, col, L = [0], [], []
    for tids, euma in rsh["rows"]:
        col.extend(tids)
        row_ptr.append(len(col))
        a = 0.0
        if euma is not None:
            for i in range(nfl):
                a += wf[i] * float(euma[i])
        L.append(a)
    L = np.array(L)
    E = L / 1E3 * (float(total_read_count) / 1E6) * (10.0 ** delta)
    return Csr(len(rsh["names"]), row_ptr, col, R=R, E=E, L=L)
