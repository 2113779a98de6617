"""ctypes binding of include/emsar_hip.h (the C ABI that replaces run_MLE_threads(),
/root/reference/src/emsar_main.c:446).  Thin: argument marshalling and error mapping only."""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
# EMSAR_HIP_LIB: experiment hook -- another build of the same library (timing-only ablations, compile-time variants)
_LIB_PATH = os.environ.get("EMSAR_HIP_LIB") or os.path.join(_PKG, "libemsar_hip.so")
_lib = None

LAYOUT_AUTO, LAYOUT_CSR, LAYOUT_TILED = 0, 1, 3
FLAG_MERGE_ROWS = 0x100

# every symbol include/emsar_hip.h declares (tests check that the library exports exactly these)
SYMBOLS = [
    "emsar_hip_create", "emsar_hip_destroy", "emsar_hip_strerror", "emsar_hip_last_error",
    "emsar_hip_upload_structure", "emsar_hip_upload_sample", "emsar_hip_solve",
    "emsar_hip_reset_theta", "emsar_hip_set_theta", "emsar_hip_get_theta", "emsar_hip_run_passes",
    "emsar_hip_ieuma", "emsar_hip_normalise", "emsar_hip_get_info",
    "emsar_hip_layout_selfcheck_tiled", "emsar_hip_sets_selfcheck", "emsar_hip_upload_euma", "emsar_hip_adj_euma", "emsar_hip_collapse_rows",
    "emsar_hip_set_deterministic",
]


class EmsarHipError(RuntimeError):
    def __init__(self, status, what, detail=""):
        self.status = status
        super().__init__("%s: %s (status %d)%s" % (what, _strerror(status), status, (" -- " + detail) if detail else ""))


class EmParams(C.Structure):
    _fields_ = [("max_iter", C.c_int32), ("accel", C.c_int32), ("tol", C.c_double), ("abs_floor", C.c_double),
                ("check_every", C.c_int32), ("set_mode", C.c_int32), ("count_floor", C.c_double), ("zero_cut", C.c_double), ("abs_step", C.c_double),
                ("newton_after", C.c_int32), ("reserved0", C.c_int32)]


class EmStats(C.Structure):
    _fields_ = [("iters", C.c_int32), ("converged", C.c_int32), ("final_delta", C.c_double), ("loglik", C.c_double),
                ("solve_ms", C.c_double), ("kernel_ms", C.c_double), ("bytes_per_pass", C.c_int64),
                ("stored_bytes_per_pass", C.c_int64),
                ("sets_resident", C.c_int32), ("sets_streamed", C.c_int32), ("set_passes_max", C.c_int32),
                ("sets_unconverged", C.c_int32), ("set_passes_sum", C.c_int64), ("sets_build_ms", C.c_double),
                ("sets_kernel_ms", C.c_double), ("sets_cluster", C.c_int32), ("cluster_passes_max", C.c_int32), ("cluster_kernel_ms", C.c_double)]


class CollapseStats(C.Structure):
    _fields_ = [("kernel_ms", C.c_double), ("total_ms", C.c_double), ("n_rows", C.c_int64), ("nnz", C.c_int64),
                ("n_unique", C.c_int64), ("nnz_unique", C.c_int64), ("table_slots", C.c_int64), ("algorithmic_bytes", C.c_int64), ("rounds", C.c_int64)]


class SetsInfo(C.Structure):
    _fields_ = [("n_components", C.c_int64), ("sets_resident", C.c_int64 * 3), ("max_lds_bytes", C.c_int64 * 3),
                ("sets_streamed", C.c_int64), ("tids_closed", C.c_int64), ("tids_resident", C.c_int64),
                ("tids_streamed", C.c_int64), ("rows_in", C.c_int64), ("rows_stored", C.c_int64),
                ("sets_cluster", C.c_int64), ("tids_cluster", C.c_int64), ("max_lds_cluster", C.c_int64)]


class Info(C.Structure):
    _fields_ = [("n_rows", C.c_int64), ("nnz", C.c_int64), ("n_tx", C.c_int32), ("layout", C.c_int32),
                ("n_chunks", C.c_int64), ("n_slices", C.c_int64), ("padded_entries", C.c_int64),
                ("far_entries", C.c_int64), ("window", C.c_int32), ("device_id", C.c_int32),
                ("bytes_per_pass", C.c_int64), ("stored_bytes_per_pass", C.c_int64),
                ("tiled_entries", C.c_int64), ("tiled_ids", C.c_int64), ("n_units", C.c_int64), ("renumbered", C.c_int32), ("reserved0", C.c_int32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def load_library():
    """dlopen the in-tree HIP library; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise EmsarHipError(-4, "load_library", "%s missing: run python -m emsar_amd._build" % _LIB_PATH)
    L = C.CDLL(_LIB_PATH)
    vp, u64p, i32p, f64p = C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_int32), C.POINTER(C.c_double)
    L.emsar_hip_create.argtypes = [C.POINTER(vp), C.c_int]
    L.emsar_hip_destroy.argtypes = [vp]
    L.emsar_hip_destroy.restype = None
    L.emsar_hip_strerror.argtypes = [C.c_int]
    L.emsar_hip_strerror.restype = C.c_char_p
    L.emsar_hip_last_error.argtypes = [vp]
    L.emsar_hip_last_error.restype = C.c_char_p
    L.emsar_hip_upload_structure.argtypes = [vp, C.c_int64, C.c_int32, u64p, i32p, C.c_int]
    L.emsar_hip_upload_sample.argtypes = [vp, i32p, f64p, f64p]
    L.emsar_hip_solve.argtypes = [vp, C.POINTER(EmParams), f64p, C.POINTER(EmStats)]
    L.emsar_hip_reset_theta.argtypes = [vp]
    L.emsar_hip_set_theta.argtypes = [vp, f64p]
    L.emsar_hip_get_theta.argtypes = [vp, f64p]
    L.emsar_hip_run_passes.argtypes = [vp, C.c_int32, C.POINTER(C.c_float), f64p]
    L.emsar_hip_ieuma.argtypes = [vp, f64p, f64p]
    L.emsar_hip_normalise.argtypes = [vp, f64p, f64p, C.c_int64, f64p, f64p, i32p]
    L.emsar_hip_get_info.argtypes = [vp, C.POINTER(Info)]
    L.emsar_hip_layout_selfcheck_tiled.argtypes = [C.c_int64, C.c_int32, u64p, i32p, C.c_int, C.POINTER(Info)]
    L.emsar_hip_set_deterministic.argtypes = [vp, C.c_int]
    L.emsar_hip_set_deterministic.restype = C.c_int
    L.emsar_hip_collapse_rows.argtypes = [vp, C.c_int64, C.c_int32, u64p, i32p, i32p, C.POINTER(C.c_int64), u64p, i32p, i32p, i32p,
                                          C.POINTER(CollapseStats)]
    L.emsar_hip_upload_euma.argtypes = [vp, i32p, C.c_int32]
    L.emsar_hip_adj_euma.argtypes = [vp, f64p, f64p]
    L.emsar_hip_sets_selfcheck.argtypes = [C.c_int64, C.c_int32, u64p, i32p, i32p, C.POINTER(SetsInfo)]
    _lib = L
    return L


def _strerror(status):
    try:
        return load_library().emsar_hip_strerror(status).decode()
    except Exception:
        return "?"


def _p(a, ct):
    return None if a is None else a.ctypes.data_as(C.POINTER(ct))


def _arr(a, dt):
    return None if a is None else np.ascontiguousarray(a, dtype=dt)


def sets_selfcheck(n_tx, row_ptr, col_idx, row_weight=None):
    """Host-only: find + pack the connected sets for the set-resident solver and check the records (no GPU needed)."""
    L = load_library()
    row_ptr, col_idx = _arr(row_ptr, np.uint64), _arr(col_idx, np.int32)
    w = None if row_weight is None else _arr(row_weight, np.int32)
    info = SetsInfo()
    rc = L.emsar_hip_sets_selfcheck(len(row_ptr) - 1, n_tx, _p(row_ptr, C.c_uint64), _p(col_idx, C.c_int32),
                                    None if w is None else _p(w, C.c_int32), C.byref(info))
    if rc != 0:
        raise EmsarHipError(rc, "sets_selfcheck")
    d = {k: getattr(info, k) for k, _ in SetsInfo._fields_}
    d["sets_resident"] = list(d["sets_resident"])
    d["max_lds_bytes"] = list(d["max_lds_bytes"])
    return d


def layout_selfcheck_tiled(n_tx, row_ptr, col_idx, merge_rows=False):
    """Host-only: build the TILED layout, check its descriptors and decode it again (no GPU needed).  Returns its statistics."""
    L = load_library()
    row_ptr, col_idx = _arr(row_ptr, np.uint64), _arr(col_idx, np.int32)
    info = Info()
    rc = L.emsar_hip_layout_selfcheck_tiled(len(row_ptr) - 1, n_tx, _p(row_ptr, C.c_uint64), _p(col_idx, C.c_int32),
                                            int(merge_rows), C.byref(info))
    if rc != 0:
        raise EmsarHipError(rc, "layout_selfcheck_tiled")
    d = info.as_dict()
    d["folded_single_rows"] = d.pop("bytes_per_pass")
    return d


class EmsarHip:
    """One context = one GPU.  Mirrors the call sequence of the reference's per-sample loop
    (emsar_main.c:380-488): upload_structure once per rsh, upload_sample + solve per alignment file."""

    def __init__(self, device_id=0):
        self._L = load_library()
        h = C.c_void_p()
        rc = self._L.emsar_hip_create(C.byref(h), device_id)
        if rc != 0:
            raise EmsarHipError(rc, "emsar_hip_create")
        self._h = h
        self.n_tx = 0
        self.n_rows = 0

    def close(self):
        if getattr(self, "_h", None):
            self._L.emsar_hip_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc, what):
        if rc != 0:
            raise EmsarHipError(rc, what, self._L.emsar_hip_last_error(self._h).decode())

    def upload_structure(self, n_tx, row_ptr, col_idx, layout=LAYOUT_AUTO, merge_rows=False):
        row_ptr, col_idx = _arr(row_ptr, np.uint64), _arr(col_idx, np.int32)
        if merge_rows:
            layout |= FLAG_MERGE_ROWS
        self._chk(self._L.emsar_hip_upload_structure(self._h, len(row_ptr) - 1, n_tx, _p(row_ptr, C.c_uint64),
                                                     _p(col_idx, C.c_int32), layout), "upload_structure")
        self.n_tx = int(n_tx)
        self.n_rows = len(row_ptr) - 1

    def upload_sample(self, row_weight=None, row_E=None, den=None):
        w, e, d = _arr(row_weight, np.int32), _arr(row_E, np.float64), _arr(den, np.float64)
        for a, n in ((w, self.n_rows), (e, self.n_rows), (d, self.n_tx)):
            if a is not None and a.shape != (n,):
                raise ValueError("array of length %d expected" % n)
        self._chk(self._L.emsar_hip_upload_sample(self._h, _p(w, C.c_int32), _p(e, C.c_double), _p(d, C.c_double)),
                  "upload_sample")

    def solve(self, max_iter=100000, accel=1, tol=1e-10, abs_floor=1e-6, check_every=8, count_floor=0.0, set_mode=0, zero_cut=0.0, abs_step=0.0,
              newton_after=0):
        """set_mode 0: connected sets that fit a CU's LDS are solved by one workgroup each; 1: streaming passes only.
        newton_after: resident sets get projected-Newton steps once they have used this many passes (0 = 60, < 0 = never)."""
        p = EmParams(max_iter, accel, tol, abs_floor, check_every, set_mode, count_floor, zero_cut, abs_step, newton_after, 0)
        st = EmStats()
        out = np.zeros(self.n_tx)
        self._chk(self._L.emsar_hip_solve(self._h, C.byref(p), _p(out, C.c_double), C.byref(st)), "solve")
        return out, st

    def collapse_rows(self, n_tx, row_ptr, col_idx, row_weight=None, want_map=True):
        """Read -> segment collapse on the device: rows with the same multiset of ids become one weighted row.
        Returns (row_ptr, col_idx, weight, row_map, stats); unique rows in order of first occurrence, ids sorted."""
        row_ptr, col_idx = _arr(row_ptr, np.uint64), _arr(col_idx, np.int32)
        n_rows = len(row_ptr) - 1
        w = None if row_weight is None else _arr(row_weight, np.int32)
        rp_o = np.zeros(n_rows + 1, dtype=np.uint64)
        ci_o = np.zeros(max(len(col_idx), 1), dtype=np.int32)
        w_o = np.zeros(max(n_rows, 1), dtype=np.int32)
        m_o = np.zeros(max(n_rows, 1), dtype=np.int32) if want_map else None
        nu = C.c_int64(0)
        st = CollapseStats()
        self._chk(self._L.emsar_hip_collapse_rows(self._h, n_rows, n_tx, _p(row_ptr, C.c_uint64), _p(col_idx, C.c_int32),
                                                  None if w is None else _p(w, C.c_int32), C.byref(nu), _p(rp_o, C.c_uint64),
                                                  _p(ci_o, C.c_int32), _p(w_o, C.c_int32), None if m_o is None else _p(m_o, C.c_int32),
                                                  C.byref(st)), "collapse_rows")
        u = nu.value
        return rp_o[:u + 1], ci_o[:int(rp_o[u])], w_o[:u], (None if m_o is None else m_o[:n_rows]), st

    def upload_euma(self, euma):
        """EUMA[n_rows][nfl] (int32, 0 where absent): once per rsh, after upload_structure."""
        euma = _arr(euma, np.int32)
        if euma.ndim != 2 or euma.shape[0] != self.n_rows:
            raise ValueError("euma must be [n_rows][nfl]")
        self.nfl = int(euma.shape[1])
        self._chk(self._L.emsar_hip_upload_euma(self._h, _p(euma, C.c_int32), self.nfl), "upload_euma")

    def adj_euma(self, wf):
        """L_c = sum_i Wf[i] * EUMA_c[i] (compute_adjEUMA), bit-identical to the host loop."""
        wf = _arr(wf, np.float64)
        if wf.shape != (getattr(self, "nfl", -1),):
            raise ValueError("wf must have nfl entries")
        out = np.zeros(self.n_rows)
        self._chk(self._L.emsar_hip_adj_euma(self._h, _p(wf, C.c_double), _p(out, C.c_double)), "adj_euma")
        return out

    def reset_theta(self):
        self._chk(self._L.emsar_hip_reset_theta(self._h), "reset_theta")

    def set_theta(self, theta):
        theta = _arr(theta, np.float64)
        if theta.shape != (self.n_tx,):
            raise ValueError("theta must have n_tx entries")
        self._chk(self._L.emsar_hip_set_theta(self._h, _p(theta, C.c_double)), "set_theta")

    def get_theta(self):
        out = np.zeros(self.n_tx)
        self._chk(self._L.emsar_hip_get_theta(self._h, _p(out, C.c_double)), "get_theta")
        return out

    def set_deterministic(self, on=True):
        """Fixed-point sums in the streaming passes: two solves of the same input are bit-identical (include/emsar_hip.h)."""
        self._chk(self._L.emsar_hip_set_deterministic(self._h, 1 if on else 0), "set_deterministic")

    def run_passes(self, n, want_loglik=False):
        ms = C.c_float(0)
        ll = C.c_double(0)
        self._chk(self._L.emsar_hip_run_passes(self._h, n, C.byref(ms), C.byref(ll) if want_loglik else None),
                  "run_passes")
        return (ms.value, ll.value) if want_loglik else ms.value

    def ieuma(self, row_L):
        row_L = _arr(row_L, np.float64)
        out = np.zeros(self.n_tx)
        self._chk(self._L.emsar_hip_ieuma(self._h, _p(row_L, C.c_double), _p(out, C.c_double)), "ieuma")
        return out

    def normalise(self, mean_fpkm, ieuma, total_read_count):
        m, ie = _arr(mean_fpkm, np.float64), _arr(ieuma, np.float64)
        tpm, ir = np.zeros(self.n_tx), np.zeros(self.n_tx)
        iri = np.zeros(self.n_tx, dtype=np.int32)
        self._chk(self._L.emsar_hip_normalise(self._h, _p(m, C.c_double), _p(ie, C.c_double), int(total_read_count),
                                              _p(tpm, C.c_double), _p(ir, C.c_double), _p(iri, C.c_int32)), "normalise")
        return tpm, ir, iri

    def info(self):
        i = Info()
        self._chk(self._L.emsar_hip_get_info(self._h, C.byref(i)), "get_info")
        return i.as_dict()
