"""Diagnostic: per-phase cycle shares of k_pass_tiled (stamped instance) on a BASELINE config."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from emsar_amd import EmsarHip, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
s = synth.make_config(cfg, scale, sys.argv[3] if len(sys.argv) > 3 else "family")
dev = EmsarHip(0)
dev.upload_structure(s["n_tx"], s["row_ptr"], s["col_idx"], 3)
dev.upload_sample(None, None, s["den"])
dev.run_passes(3)
out = (C.c_double * 8)()
dev._L.emsar_hip_debug_tiled_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
rc = dev._L.emsar_hip_debug_tiled_stamps(dev._h, out)
names = ["issue loads + dictionary", "barrier 1", "E-step", "M-step", "barrier 2"]
tot = sum(out[i] for i in range(5))
print("rc", rc, "tiles", out[7], "mean cycles per wave", tot)
for i, n in enumerate(names):
    print("%-26s %10.0f  %5.1f%%" % (n, out[i], 100 * out[i] / tot))
print("ms/pass (unstamped)", dev.run_passes(20) / 20)
out = (C.c_double * 8)()
n_units = int(dev.info()["n_chunks"])            # an upper bound of the units
tl = np.zeros(n_units * 4, dtype=np.uint64)
dev._L.emsar_hip_debug_unit_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_void_p]
for rep in range(2):                              # the second launch: code and descriptors warm
    rc = dev._L.emsar_hip_debug_unit_stamps(dev._h, out, tl.ctypes.data)
names = ["descriptor + dictionary + loads", "barrier 1", "E-steps", "M-steps", "barrier 2", "flush"]
tot = sum(out[i] for i in range(6))
print("unit kernel: rc", rc, "units", out[7], "tiles per unit %.2f" % out[6], "mean cycles per wave", tot)
for i, n in enumerate(names):
    print("%-34s %10.0f  %5.1f%%" % (n, out[i], 100 * out[i] / tot))
# timeline of the workgroups (100 MHz clock: 10 ns ticks)
nu = int(out[7])
t = tl[: nu * 4].reshape(nu, 4)
st, en, place = t[:, 0].astype(np.int64), t[:, 1].astype(np.int64), t[:, 2].astype(np.int64)
t0 = st.min()
st, en = (st - t0) * 0.01, (en - t0) * 0.01        # us
span = en.max()
life = en - st
print("timeline: span %.1f us, workgroup life mean %.2f us (p10 %.2f, p50 %.2f, p90 %.2f, max %.2f); resident workgroups on average %.0f of 1024"
      % (span, life.mean(), *np.percentile(life, [10, 50, 90]), life.max(), life.sum() / span))
print("          last workgroup starts at %.1f us; first 1024 started by %.1f us" % (st.max(), np.sort(st)[min(1023, nu - 1)]))
grid = np.arange(0, span, 4.0)
res = [(int(((st <= g) & (en > g)).sum())) for g in grid]
print("          resident at t = 0, 4, 8 ... us:", res)
cu = (place >> 16) * 4096 + ((place >> 13) & 7) * 64 + ((place >> 12) & 1) * 16 + ((place >> 8) & 15)    # xcc, se, sh, cu
inv = np.unique(cu, return_inverse=True)[1]
cnt = np.bincount(inv)
print("          distinct CUs seen", cnt.size, "workgroups per CU min/mean/max", cnt.min(), nu / cnt.size, cnt.max())
occ = np.zeros(cnt.size)
np.add.at(occ, inv, life)
occ /= span
print("          resident workgroups per CU over the span: mean %.2f min %.2f max %.2f (capacity 4)" % (occ.mean(), occ.min(), occ.max()))
xcc = place >> 16
for x in np.unique(xcc):
    m = xcc == x
    print("          XCD %d: %d workgroups, last end %.1f us, mean life %.2f us" % (x, m.sum(), en[m].max(), life[m].mean()))
# slot turnover: on every CU, the time its four workgroup slots were empty between the first start and the last start there,
# per workgroup that started after the first four (= what a finished workgroup's successor waited to begin)
idle = []
for c in range(cnt.size):
    m = inv == c
    s_c, e_c = np.sort(st[m]), en[m]
    t_a, t_b = s_c[min(3, s_c.size - 1)], s_c[-1]          # window: all four slots taken ... last start
    if t_b <= t_a or s_c.size <= 4:
        continue
    busy = (np.minimum(e_c, t_b) - np.maximum(st[m], t_a)).clip(min=0).sum()
    idle.append((4 * (t_b - t_a) - busy) / (s_c.size - 4))
idle = np.array(idle)
print("          empty slot time per successor workgroup: mean %.2f us (p10 %.2f, p90 %.2f)" % (idle.mean(), *np.percentile(idle, [10, 90])))
