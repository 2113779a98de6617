"""ctypes binding of libemsar_host.so (emsar_amd/csrc/host): the C host side -- rsh reader, alignment readers +
per-read collapse, model preparation and the output writers.  Used by the tests; the CLI emsar-hip links the same
code directly."""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_PKG, "libemsar_host.so")
_lib = None


class Rsh(C.Structure):
    _fields_ = [("n_tx", C.c_int32), ("names", C.POINTER(C.c_char_p)),
                ("hdr_minfrag", C.c_int32), ("hdr_maxfrag", C.c_int32), ("hdr_readlength", C.c_int32), ("max_t_size", C.c_int32),
                ("frag_min", C.c_int32), ("frag_max", C.c_int32), ("nfl", C.c_int32),
                ("n_rows", C.c_int64), ("row_ptr", C.POINTER(C.c_uint64)), ("col_idx", C.POINTER(C.c_int32)),
                ("euma", C.POINTER(C.c_int32)), ("has_node", C.POINTER(C.c_uint8)),
                ("name_index", C.c_void_p), ("set_index", C.c_void_p)]


COLLAPSE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_int32, C.POINTER(C.c_uint64), C.POINTER(C.c_int32),
                          C.POINTER(C.c_int64), C.POINTER(C.c_uint64), C.POINTER(C.c_int32), C.POINTER(C.c_int32))


class AlnOpts(C.Structure):
    _fields_ = [("pe", C.c_int), ("strand", C.c_char), ("max_repeat", C.c_int), ("format", C.c_int),
                ("collapse", C.c_void_p), ("collapse_user", C.c_void_p), ("collapse_batch_rows", C.c_int64)]


class Counts(C.Structure):
    _fields_ = [("n_rows", C.c_int64), ("R", C.POINTER(C.c_int32)), ("n_frag", C.c_int32),
                ("frag_counts", C.POINTER(C.c_int32)), ("total_reads", C.c_int64),
                ("reads_seen", C.c_int64), ("reads_over_k", C.c_int64), ("reads_bad_fraglen", C.c_int64),
                ("reads_discrepant", C.c_int64), ("reads_no_segment", C.c_int64), ("readlength", C.c_int32), ("batch", C.c_void_p)]


class Model(C.Structure):
    _fields_ = [("n_rows", C.c_int64), ("n_tx", C.c_int32), ("nfl", C.c_int32), ("Wf", C.POINTER(C.c_double)),
                ("L", C.POINTER(C.c_double)), ("E", C.POINTER(C.c_double)), ("E_solver", C.POINTER(C.c_double)),
                ("CS", C.POINTER(C.c_int32)), ("TS", C.POINTER(C.c_int32)), ("n_sets", C.c_int32), ("eumacut", C.c_double)]


class HostError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            raise HostError("%s missing: run python -m emsar_amd._build" % _PATH)
        L = C.CDLL(_PATH)
        L.emsar_rsh_read.argtypes = [C.c_char_p, C.POINTER(C.POINTER(Rsh)), C.c_char_p, C.c_size_t]
        L.emsar_rsh_free.argtypes = [C.POINTER(Rsh)]
        L.emsar_rsh_free.restype = None
        L.emsar_rsh_tid_of.argtypes = [C.POINTER(Rsh), C.c_char_p]
        L.emsar_rsh_tid_of.restype = C.c_int32
        L.emsar_rsh_row_of.argtypes = [C.POINTER(Rsh), C.POINTER(C.c_int32), C.c_int]
        L.emsar_rsh_row_of.restype = C.c_int64
        L.emsar_set_strand.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_char)]
        L.emsar_count_alignments.argtypes = [C.POINTER(Rsh), C.c_char_p, C.POINTER(AlnOpts), C.POINTER(C.POINTER(Counts)),
                                             C.c_char_p, C.c_size_t]
        L.emsar_counts_free.argtypes = [C.POINTER(Counts)]
        L.emsar_counts_free.restype = None
        L.emsar_model_build.argtypes = [C.POINTER(Rsh), C.POINTER(Counts), C.c_int, C.POINTER(C.c_double),
                                        C.POINTER(C.POINTER(Model)), C.c_char_p, C.c_size_t]
        L.emsar_rsh_write_cache.argtypes = [C.POINTER(Rsh), C.c_char_p, C.c_char_p]
        L.emsar_rsh_read_cache.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.POINTER(Rsh)), C.c_char_p, C.c_size_t]
        L.emsar_model_wf.argtypes = [C.POINTER(Rsh), C.POINTER(Counts), C.POINTER(C.c_double)]
        L.emsar_model_build_L.argtypes = [C.POINTER(Rsh), C.POINTER(Counts), C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                          C.POINTER(C.POINTER(Model)), C.c_char_p, C.c_size_t]
        L.emsar_model_free.argtypes = [C.POINTER(Model)]
        L.emsar_model_free.restype = None
        f64p, i32p = C.POINTER(C.c_double), C.POINTER(C.c_int32)
        L.emsar_mean_sd.argtypes = [C.c_int32, C.c_int32, f64p, f64p, f64p]
        L.emsar_mean_sd.restype = None
        L.emsar_write_fpkm.argtypes = [C.c_char_p, C.POINTER(Rsh), f64p, f64p, f64p, f64p, i32p, f64p, C.POINTER(C.c_int64)]
        L.emsar_write_fraglength.argtypes = [C.c_char_p, C.POINTER(Rsh), C.POINTER(Counts), C.POINTER(Model)]
        L.emsar_write_segments.argtypes = [C.c_char_p, C.POINTER(Rsh), C.POINTER(Counts), C.POINTER(Model), f64p]
        _lib = L
    return _lib


def _np(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class HostRsh:
    def __init__(self, path, cache=None, check_source=True):
        """cache = path of a binary cache to read INSTEAD of the text (HostError if it is stale, foreign or damaged)."""
        p = C.POINTER(Rsh)()
        err = C.create_string_buffer(512)
        if cache is None:
            rc = lib().emsar_rsh_read(path.encode(), C.byref(p), err, 512)
        else:
            rc = lib().emsar_rsh_read_cache(path.encode() if check_source else None, cache.encode(), C.byref(p), err, 512)
        if rc != 0:
            raise HostError("rsh_read rc=%d: %s" % (rc, err.value.decode()))
        self.path = path
        self._p = p
        r = p.contents
        self.n_tx, self.n_rows, self.nfl = r.n_tx, r.n_rows, r.nfl
        self.frag_min, self.frag_max = r.frag_min, r.frag_max
        self.names = [r.names[i].decode() for i in range(r.n_tx)]
        self.row_ptr = _np(r.row_ptr, r.n_rows + 1, np.uint64)
        self.col_idx = _np(r.col_idx, int(self.row_ptr[-1]), np.int32)
        self.euma = _np(r.euma, r.n_rows * r.nfl, np.int32).reshape(r.n_rows, r.nfl)
        self.has_node = _np(r.has_node, r.n_rows, np.uint8)

    def write_cache(self, cache_path):
        rc = lib().emsar_rsh_write_cache(self._p, self.path.encode(), cache_path.encode())
        if rc != 0:
            raise HostError("rsh_write_cache rc=%d" % rc)

    def tid_of(self, name):
        return lib().emsar_rsh_tid_of(self._p, name.encode())

    def row_of(self, tids):
        a = np.ascontiguousarray(sorted(tids), dtype=np.int32)
        return lib().emsar_rsh_row_of(self._p, a.ctypes.data_as(C.POINTER(C.c_int32)), len(a))

    def count(self, aln_path, pe=0, strand="ns", max_repeat=100, fmt=0, collapse=None, collapse_batch_rows=0):
        """collapse: optional callable (row_ptr u64[n+1], col_idx i32[nnz]) -> (row_ptr, col_idx, weight) of the unique rows --
        the kept reads with two or more transcripts then reach the counts through it, in batches (emsar_aln_opts.collapse)."""
        s = C.c_char()
        if lib().emsar_set_strand(strand.encode(), pe, C.byref(s)) != 0:
            raise HostError("invalid strand type")
        o = AlnOpts(pe, s.value, max_repeat, fmt, None, None, collapse_batch_rows)
        if collapse is not None:
            def _cb(user, n_rows, n_tx, rp, ci, nu, rp_o, ci_o, w_o):
                try:
                    rp_a = np.ctypeslib.as_array(rp, shape=(n_rows + 1,)).copy()
                    ci_a = np.ctypeslib.as_array(ci, shape=(max(int(rp_a[-1]), 1),))[:int(rp_a[-1])].copy()
                    a, b, w = collapse(rp_a, ci_a)
                    nu[0] = len(w)
                    np.ctypeslib.as_array(rp_o, shape=(n_rows + 1,))[:len(a)] = a
                    if len(b):
                        np.ctypeslib.as_array(ci_o, shape=(max(int(rp_a[-1]), 1),))[:len(b)] = b
                    np.ctypeslib.as_array(w_o, shape=(n_rows,))[:len(w)] = w
                    return 0
                except Exception:
                    return -1
            self._collapse_cb = COLLAPSE_FN(_cb)            # kept alive for the duration of the call
            o.collapse = C.cast(self._collapse_cb, C.c_void_p)
        p = C.POINTER(Counts)()
        err = C.create_string_buffer(512)
        rc = lib().emsar_count_alignments(self._p, aln_path.encode(), C.byref(o), C.byref(p), err, 512)
        if rc != 0:
            raise HostError("count_alignments rc=%d: %s" % (rc, err.value.decode()))
        return HostCounts(p)

    def wf(self, counts):
        """Normalised fragment-length histogram of the sample (transfer_fraglendist_to_Wf)."""
        out = np.zeros(self.nfl)
        if lib().emsar_model_wf(self._p, counts._p, _dp(out)) != 0:
            raise HostError("no read inside the fragment-length range")
        return out

    def model(self, counts, delta=0, eumacut=0.0, L=None):
        """L = per-row adjEUMA computed elsewhere (EmsarHip.adj_euma); None = the host loop."""
        cut = C.c_double(eumacut)
        p = C.POINTER(Model)()
        err = C.create_string_buffer(512)
        if L is not None:
            L = np.ascontiguousarray(L, dtype=np.float64)
            if L.shape != (self.n_rows,):
                raise ValueError("L must have n_rows entries")
        rc = lib().emsar_model_build_L(self._p, counts._p, delta, C.byref(cut), None if L is None else _dp(L), C.byref(p), err, 512)
        if rc != 0:
            raise HostError("model_build rc=%d: %s" % (rc, err.value.decode()))
        return HostModel(p, cut.value)

    def write_fpkm(self, path, mean, sd, ieuma, ir, iri, tpm):
        a = [np.ascontiguousarray(x, dtype=np.float64) for x in (mean, sd, ieuma, ir)]
        iri = np.ascontiguousarray(iri, dtype=np.int32)
        tpm = np.ascontiguousarray(tpm, dtype=np.float64)
        tot = C.c_int64()
        rc = lib().emsar_write_fpkm(path.encode(), self._p, _dp(a[0]), _dp(a[1]), _dp(a[2]), _dp(a[3]),
                                    iri.ctypes.data_as(C.POINTER(C.c_int32)), _dp(tpm), C.byref(tot))
        if rc != 0:
            raise HostError("write_fpkm rc=%d" % rc)
        return tot.value

    def write_fraglength(self, path, counts, model):
        if lib().emsar_write_fraglength(path.encode(), self._p, counts._p, model._p) != 0:
            raise HostError("write_fraglength")

    def write_segments(self, path, counts, model, mean):
        mean = np.ascontiguousarray(mean, dtype=np.float64)
        if lib().emsar_write_segments(path.encode(), self._p, counts._p, model._p, _dp(mean)) != 0:
            raise HostError("write_segments")

    def __del__(self):
        if getattr(self, "_p", None):
            lib().emsar_rsh_free(self._p)
            self._p = None


class HostCounts:
    def __init__(self, p):
        self._p = p
        c = p.contents
        self.R = _np(c.R, c.n_rows, np.int32)
        self.frag_counts = _np(c.frag_counts, c.n_frag, np.int32)
        self.total_reads = c.total_reads
        self.stats = {k: getattr(c, k) for k in ("reads_seen", "reads_over_k", "reads_bad_fraglen", "reads_discrepant",
                                                 "reads_no_segment", "readlength")}

    def __del__(self):
        if getattr(self, "_p", None):
            lib().emsar_counts_free(self._p)
            self._p = None


class HostModel:
    def __init__(self, p, eumacut):
        self._p = p
        m = p.contents
        self.Wf = _np(m.Wf, m.nfl, np.float64)
        self.L = _np(m.L, m.n_rows, np.float64)
        self.E = _np(m.E, m.n_rows, np.float64)
        self.E_solver = _np(m.E_solver, m.n_rows, np.float64)
        self.CS = _np(m.CS, m.n_rows, np.int32)
        self.TS = _np(m.TS, m.n_tx, np.int32)
        self.n_sets = m.n_sets
        self.eumacut = eumacut

    def __del__(self):
        if getattr(self, "_p", None):
            lib().emsar_model_free(self._p)
            self._p = None


def mean_sd(rounds):
    rounds = np.ascontiguousarray(np.atleast_2d(rounds), dtype=np.float64)
    n_round, n_tx = rounds.shape
    mean, sd = np.zeros(n_tx), np.zeros(n_tx)
    lib().emsar_mean_sd(n_tx, n_round, _dp(rounds), _dp(mean), _dp(sd))
    return mean, sd
