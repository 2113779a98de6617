"""Diagnostic: how long every workgroup (chunk) and wave of k_pass_tiled runs in one stamped pass, and how much of a wave's life is
spent inside slices (the rest: dictionary phases, the barriers around them, waiting for the other waves at the end of a group)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from emsar_amd import EmsarHip, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
s = synth.make_config(cfg, scale)
dev = EmsarHip(0)
dev.upload_structure(s["n_tx"], s["row_ptr"], s["col_idx"], 3)
dev.upload_sample(None, None, s["den"])
dev.run_passes(30)
info = dev.info()
n = info["n_chunks"] * 32 + 5 * info["n_slices"]
out = (C.c_ulonglong * n)()
dev._L.emsar_hip_debug_chunk_times.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int64]
rc = dev._L.emsar_hip_debug_chunk_times(dev._h, out, n)
raw = np.frombuffer(out, dtype=np.uint64).astype(np.float64)
a = raw[:info["n_chunks"] * 32].reshape(info["n_chunks"], 4, 8)
sl_ = raw[info["n_chunks"] * 32:].reshape(info["n_slices"], 5).copy()
chunk_of = (sl_[:, 4].astype(np.int64) >> 8)
sl_[:, 4] = (sl_[:, 4].astype(np.int64) & 255)
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed("gpurun_out/chunk_times_%s.npz" % os.environ.get("EMSAR_TAG", "run"), waves=a, slices=sl_, chunk_of=chunk_of)
t0 = a[:, :, 0].min()
start, end = (a[:, :, 0] - t0) / 100.0, (a[:, :, 1] - t0) / 100.0          # microseconds
wg_end = end.max(axis=1); wg_start = start.min(axis=1)
dur = wg_end - wg_start
print("rc", rc, info)
print("kernel span %.1f us; workgroup start spread %.1f us" % (wg_end.max(), wg_start.max()))
print("workgroup duration us: min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f" % (dur.min(), np.percentile(dur, 10), np.median(dur), np.percentile(dur, 90), dur.max()))
print("workgroup end us:      min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f" % (wg_end.min(), np.percentile(wg_end, 10), np.median(wg_end), np.percentile(wg_end, 90), wg_end.max()))
sl = a[:, :, 3]
ph = a[:, :, 4:8].sum(axis=(0, 1)); tot_ph = ph.sum()
print("share of in-slice cycles: loads issued %.1f %%, E-step %.1f %%, weights + next slice's loads %.1f %%, M-step %.1f %%" % tuple(100 * ph / tot_ph))
print("slices per wave: mean %.2f  min %d  max %d" % (sl.mean(), sl.min(), sl.max()))
cyc = a[:, :, 2]
life_us = (end - start)
clk = cyc.sum() / (life_us.sum() * 1e-6) / 1e9 if life_us.sum() > 0 else 0
print("cycles inside slices / wave life (at an assumed 2.1 GHz): %.1f %%;   mean cycles per slice %.0f" % (100 * cyc.sum() / (life_us.sum() * 2100.0), cyc.sum() / max(sl.sum(), 1)))
# cost model of a slice: cycles ~ c0 + c1 k + c2 m + c3 coo_n + c4 nf (least squares over all slices of this pass)
X = np.column_stack([np.ones(len(sl_)), sl_[:, 1], sl_[:, 2], sl_[:, 3], sl_[:, 4]])
coef, *_ = np.linalg.lstsq(X, sl_[:, 0], rcond=None)
res = sl_[:, 0] - X @ coef
print("slice cycles ~ %.0f + %.0f k + %.0f m + %.2f coo_n + %.0f nf   (rms residual %.0f of mean %.0f; k mean %.1f, m mean %.1f)"
      % (coef[0], coef[1], coef[2], coef[3], coef[4], np.sqrt((res ** 2).mean()), sl_[:, 0].mean(), sl_[:, 1].mean(), sl_[:, 2].mean()))
for kk in (2, 3, 4, 6, 8, 9, 12, 16, 17, 24, 32, 48, 64, 100):
    sel = sl_[:, 1] == kk
    if sel.any(): print("  k = %3d: %5d slices, mean cycles %.0f, mean m %.1f" % (kk, sel.sum(), sl_[sel, 0].mean(), sl_[sel, 2].mean()))
print("ms/pass (unstamped)", dev.run_passes(50) / 50)
