"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the reference's golden outputs.

Tolerances (FP64 throughout; the only difference between device and oracle is summation order):
  * one EM pass from the same theta:      |dtheta| <= 1e-12 * theta + 1e-300
  * solve vs the reference's .fpkm:       SURVEY.md 8c criterion (likelihood parity, 1e-5 rel + 1.5e-6 abs off
                                          the reference's own noise mask, segment expected counts)
  * iEUMA / TPM / iReadcount:             1e-12 relative (same arithmetic, different summation order)
"""
import numpy as np
import pytest

import oracle as O
from emsar_amd import EmsarHip, EmsarHipError, synth
from emsar_amd.hip import LAYOUT_CSR, LAYOUT_TILED

pytestmark = pytest.mark.gpu
from emsar_amd.hip import FLAG_MERGE_ROWS

LAYOUT_TILED_MERGED = LAYOUT_TILED | FLAG_MERGE_ROWS      # identical rows stored once (read -> segment collapse)
LAYOUTS = [LAYOUT_CSR, LAYOUT_TILED, LAYOUT_TILED_MERGED]


@pytest.fixture(scope="module")
def dev():
    ctx = EmsarHip(0)
    yield ctx
    ctx.close()


def _upload(dev, m, layout):
    dev.upload_structure(m.n_tx, m.row_ptr, m.col_idx, layout)
    dev.upload_sample(m.R, m.E, None)


@pytest.mark.parametrize("layout", LAYOUTS)
def test_single_pass_matches_oracle(dev, golden, layout):
    m = golden.model
    _upload(dev, m, layout)
    den = m.den()
    th = np.where(den > 0, 1.0, 0.0)
    np.testing.assert_array_equal(dev.get_theta(), th)          # uniform start, zeros outside F
    for _ in range(5):
        th, ll = m.em_step(th, den)
        _, ll_dev = dev.run_passes(1, want_loglik=True)
        got = dev.get_theta()
        assert np.all(np.abs(got - th) <= 1e-12 * np.abs(th) + 1e-300)
        assert abs(ll_dev - ll) <= 1e-11 * abs(ll) + 1e-9
        dev.set_theta(th)                                         # keep both on the identical trajectory


@pytest.mark.parametrize("layout", LAYOUTS)
@pytest.mark.parametrize("accel", [0, 1])
def test_solve_meets_reference_parity(dev, golden, layout, accel):
    m = golden.model
    _upload(dev, m, layout)
    th, st = dev.solve(max_iter=600000, accel=accel, tol=1e-10, check_every=16)
    assert st.converged == 1
    golden.check_fpkm_parity(th, "hip layout=%d accel=%d" % (layout, accel))
    F = m.loglik(th)
    assert abs(st.loglik - F) <= 1e-9 * abs(F)                    # device-side F equals Fp's definition
    th_o, st_o = m.em_solve(max_iter=600000, accel=accel, tol=1e-10)
    assert np.all(np.abs(th - th_o) <= 1e-6 * np.abs(th_o) + 1.5e-6)


@pytest.mark.parametrize("layout", LAYOUTS)
def test_postprocessing_matches_oracle_and_reference(dev, golden, layout):
    m = golden.model
    _upload(dev, m, layout)
    ie = dev.ieuma(m.L)
    np.testing.assert_allclose(ie, m.ieuma(), rtol=1e-12, atol=0)
    assert np.abs(ie - golden.runs[0]["efflen"]).max() <= 5.01e-7   # column 4 of the reference's .fpkm
    ref = golden.runs[0]
    tpm, ir, iri = dev.normalise(ref["fpkm"], ie, golden.N)
    mean, sd, ir_o, iri_o, tpm_o = O.fpkm_table(np.array([ref["fpkm"]] * 4), ie, golden.N)
    np.testing.assert_allclose(tpm, tpm_o, rtol=1e-12)
    np.testing.assert_allclose(ir, ir_o, rtol=1e-12)
    assert (iri == iri_o).all()
    # against the reference's own columns (printed with 6 decimals from its unrounded mean)
    assert np.all(np.abs(tpm - ref["tpm"]) <= 1e-6 * np.abs(ref["tpm"]) + 2e-6)
    assert np.all(np.abs(ir - ref["ireadcount"]) <= 1e-6 * np.abs(ref["ireadcount"]) + 2e-6)


def test_known_answers_and_edge_cases(dev):
    for layout in LAYOUTS:
        # closed form R/E, all-zero set, E==0 row, duplicate tid
        dev.upload_structure(4, [0, 1, 2, 4, 6, 6], [0, 1, 1, 2, 3, 3], layout)   # last row empty
        dev.upload_sample([7, 0, 0, 8, 5], [2.0, 3.0, 1.5, 2.0, 1.0], None)
        th, st = dev.solve(max_iter=2000, accel=0, tol=1e-14)
        assert th[0] == 3.5 and th[1] == 0 and th[2] == 0
        assert abs(th[3] - 2.0) < 1e-14
        dev.upload_structure(2, [0, 1, 2], [0, 1], layout)
        dev.upload_sample([5, 9], [1.0, 0.0], None)
        th, _ = dev.solve(max_iter=100, accel=1, tol=1e-14)
        assert th[0] == 5.0 and th[1] == 0.0
        # empty matrix
        dev.upload_structure(3, [0], [], layout)
        dev.upload_sample(None, None, None)
        th, _ = dev.solve(max_iter=10)
        assert (th == 0).all()


def test_tiled_long_rows_and_folded_singles(dev):
    """TILED: rows longer than 1024 tids go to the leftover CSR, single-tid rows are folded analytically."""
    rng = np.random.default_rng(5)
    n_tx = 5000
    rows = [rng.choice(n_tx, size=1500, replace=False), rng.choice(n_tx, size=1100, replace=False)]
    rows += [rng.integers(0, n_tx, size=int(k)) for k in rng.integers(1, 40, size=3000)]      # duplicates allowed (A2)
    rows += [np.array([t]) for t in rng.integers(0, n_tx, size=2000)]
    rp = np.zeros(len(rows) + 1, dtype=np.uint64)
    rp[1:] = np.cumsum([len(r) for r in rows])
    ci = np.concatenate(rows).astype(np.int32)
    R = rng.integers(0, 9, size=len(rows)).astype(np.int32)
    E = rng.random(len(rows)) + 0.1
    E[::17] = 0.0
    m = O.Csr(n_tx, rp, ci, R=R, E=E)
    den = m.den()
    want = np.where(den > 0, 1.0, 0.0)
    for _ in range(4):
        want, ll = m.em_step(want, den)
    got = {}
    for layout in LAYOUTS:
        dev.upload_structure(n_tx, rp, ci, layout)
        dev.upload_sample(R, E, None)
        _, ll_dev = dev.run_passes(4, want_loglik=True)
        got[layout] = dev.get_theta()
        assert np.all(np.abs(got[layout] - want) <= 1e-11 * np.abs(want) + 1e-300), layout
        assert abs(ll_dev - ll) <= 1e-10 * abs(ll)
    if True:
        dev.upload_structure(n_tx, rp, ci, LAYOUT_TILED)
        np.testing.assert_allclose(dev.ieuma(E), O.Csr(n_tx, rp, ci, L=E).ieuma(), rtol=1e-12)


def test_error_paths(dev):
    fresh = EmsarHip(0)
    with pytest.raises(EmsarHipError) as e:
        fresh.upload_sample(None, None, None)                      # no structure yet
    assert e.value.status == -5
    with pytest.raises(EmsarHipError) as e:
        fresh.upload_structure(2, [0, 1], [2])                     # tid out of range never reaches a kernel
    assert e.value.status == -1
    fresh.upload_structure(2, [0, 1], [1])
    with pytest.raises(EmsarHipError):
        fresh.upload_sample(np.array([-1], dtype=np.int32), None, None)
    fresh.close()
    with pytest.raises(EmsarHipError) as e:
        EmsarHip(999)
    assert e.value.status == -2


@pytest.mark.parametrize("set_mode", [0, 1])
def test_non_finite_theta_is_reported_not_returned(set_mode):
    """EMSAR_HIP_ERR_NUMERIC (-6): effective lengths so small that theta = reads / den leaves the double range.  The solve must
    say so instead of handing back inf / NaN (the reference would print them)."""
    s = synth.make_matrix(n_tx=300, n_reads=20000, law="human", xfam=0.02, seed=4)
    den = np.full(s["n_tx"], 1e-307)                               # reads / den > 1.8e308 for every expressed transcript: the first M-step overflows
    with EmsarHip(0) as ctx:
        ctx.upload_structure(s["n_tx"], s["row_ptr"], s["col_idx"], LAYOUT_TILED)
        ctx.upload_sample(None, None, den)
        with pytest.raises(EmsarHipError) as e:
            ctx.solve(set_mode=set_mode, max_iter=200, tol=1e-9)
        assert e.value.status == -6
        ctx.upload_sample(None, None, s["den"])                     # the context is still usable
        th, st = ctx.solve(set_mode=set_mode, max_iter=5000, tol=1e-8)
        assert np.isfinite(th).all() and st.converged == 1


# the three row laws of emsar_amd/synth.py: consecutive-tid windows, SURVEY 8d's family subsets, and the same with the transcripts
# numbered at random; renumber 2 = the library's own numbering (csrc/renumber.hpp) forced on, 0 = off, 1 = decided by the data
@pytest.mark.parametrize("name,scale,structure,renumber", [
    ("cfg2", 0.05, "window", "1"), ("cfg3", 0.004, "window", "1"), ("cfg5", 0.0005, "window", "1"),
    ("cfg3", 0.004, "family", "1"), ("cfg3", 0.004, "family", "2"), ("cfg3", 0.004, "family_shuffled", "0"),
    ("cfg3", 0.004, "family_shuffled", "2"), ("cfg3", 0.02, "family_shuffled", "1"), ("cfg3", 0.004, "window", "2"),
    ("cfg5", 0.0005, "family_shuffled", "2"), ("cfg2", 0.05, "family", "2")])
def test_synthetic_read_level_matches_oracle(dev, monkeypatch, name, scale, structure, renumber):
    monkeypatch.setenv("EMSAR_HIP_RENUMBER", renumber)
    s = synth.make_config(name, scale, structure)               # cfg5: heavy repeats, rows of 50-100 tids
    m = O.Csr(s["n_tx"], s["row_ptr"], s["col_idx"])
    den = s["den"]
    th0 = np.ones(s["n_tx"])
    want, ll = m.em_step(th0, den, n_threads=4)
    want2, _ = m.em_step(want, den, n_threads=4)
    for layout in LAYOUTS:
        dev.upload_structure(s["n_tx"], s["row_ptr"], s["col_idx"], layout)
        dev.upload_sample(None, None, den)
        dev.run_passes(2)
        got = dev.get_theta()
        assert np.all(np.abs(got - want2) <= 1e-11 * np.abs(want2) + 1e-300)
        if layout != LAYOUT_CSR:
            assert dev.info()["renumbered"] == {"0": 0, "2": 1}.get(renumber, dev.info()["renumbered"])
    # the ABI speaks the caller's numbering whatever the library chose inside: set / get round trip, den, the per-transcript outputs
    dev.upload_structure(s["n_tx"], s["row_ptr"], s["col_idx"], LAYOUT_TILED)
    dev.upload_sample(None, None, None)                         # den computed on the device: E = 1 per row
    dev.set_theta(want)
    np.testing.assert_array_equal(dev.get_theta(), want)
    dev.run_passes(1)
    want_e1, _ = m.em_step(want, np.bincount(s["col_idx"], minlength=s["n_tx"]).astype(np.float64), n_threads=4)
    got = dev.get_theta()
    assert np.all(np.abs(got - want_e1) <= 1e-11 * np.abs(want_e1) + 1e-300)
    L = np.random.default_rng(1).uniform(0.5, 2.0, size=s["n_reads"])
    ie = dev.ieuma(L)
    ie_want = np.bincount(s["col_idx"], weights=np.repeat(L, np.diff(s["row_ptr"].astype(np.int64))), minlength=s["n_tx"])
    assert np.all(np.abs(ie - ie_want) <= 1e-10 * ie_want + 1e-300)


def test_collapsed_and_read_level_agree(dev):
    # the reference solves the collapsed (segment) form; the read-level matrix must give the same EM
    s = synth.make_matrix(n_tx=800, n_reads=30000, law="human", xfam=0.02, seed=4)
    rp, ci, cnt = synth.collapse(s["row_ptr"], s["col_idx"])
    for layout in (LAYOUT_TILED, LAYOUT_TILED_MERGED):
        dev.upload_structure(s["n_tx"], s["row_ptr"], s["col_idx"], layout)
        dev.upload_sample(None, None, s["den"])
        dev.run_passes(25)
        a = dev.get_theta()
        dev.upload_structure(s["n_tx"], rp, ci, layout)
        dev.upload_sample(cnt, None, s["den"])
        dev.run_passes(25)
        b = dev.get_theta()
        assert np.all(np.abs(a - b) <= 1e-10 * np.abs(a) + 1e-300)


def test_count_floor_stops_early_without_moving_expressed_transcripts(dev):
    """Stopping-rule floor in reads: boundary components (optimum theta = 0, zero gradient) decay like 1/k and keep
    the plain rule busy; with a floor of 1e-3 inferred reads the solve ends much earlier and every transcript that
    holds reads is where the tight solve puts it."""
    s = synth.make_matrix(n_tx=3000, n_reads=300000, law="human", xfam=0.02, seed=12)
    dev.upload_structure(s["n_tx"], s["row_ptr"], s["col_idx"], LAYOUT_TILED)
    dev.upload_sample(None, None, s["den"])
    tight, st_t = dev.solve(max_iter=60000, accel=1, tol=1e-8, abs_floor=1e-6)
    loose, st_l = dev.solve(max_iter=60000, accel=1, tol=1e-8, abs_floor=1e-6, count_floor=1e-3)
    assert st_l.converged == 1 and st_t.converged == 1      # (pass counts vary run to run: atomics order steers SQUAREM)
    reads = tight * s["den"]
    big = reads > 1.0
    assert np.all(np.abs(loose - tight)[big] <= 1e-4 * tight[big])
    assert np.abs((loose - tight) * s["den"]).max() < 0.5             # nobody moved by half a read
    with pytest.raises(EmsarHipError):
        dev.solve(count_floor=-1.0)


def test_merge_rows_reduces_the_stored_matrix(dev):
    s = synth.make_matrix(n_tx=2000, n_reads=200000, law="human", xfam=0.02, seed=8)
    dev.upload_structure(s["n_tx"], s["row_ptr"], s["col_idx"], LAYOUT_TILED)
    a = dev.info()
    dev.upload_structure(s["n_tx"], s["row_ptr"], s["col_idx"], LAYOUT_TILED, merge_rows=True)
    b = dev.info()
    assert b["layout"] == LAYOUT_TILED_MERGED and a["layout"] == LAYOUT_TILED
    assert b["n_slices"] * 2 < a["n_slices"]                         # reads of one compatibility class collapse
    with pytest.raises(EmsarHipError):                                # the flag belongs to the TILED layout only
        dev.upload_structure(s["n_tx"], s["row_ptr"], s["col_idx"], LAYOUT_CSR | FLAG_MERGE_ROWS)


def test_full_size_properties_cfg2(dev):
    """BASELINE config 2 at full size (5M reads x 80k transcripts): properties that need no oracle."""
    s = synth.make_config("cfg2", 1.0)
    n_reads, den = s["n_reads"], s["den"]
    dev.upload_structure(s["n_tx"], s["row_ptr"], s["col_idx"], LAYOUT_TILED)
    dev.upload_sample(None, None, den)
    info = dev.info()
    assert info["padded_entries"] <= 1.15 * info["nnz"], info
    prev_ll = -np.inf
    for it in range(6):
        _, ll = dev.run_passes(1, want_loglik=True)                # ll is evaluated at the pass input
        th = dev.get_theta()
        assert np.isfinite(th).all() and (th >= 0).all(), it
        # mass conservation: sum_t theta_t den_t = number of reads, after every M-step
        assert abs((th * den).sum() - n_reads) <= 1e-9 * n_reads, (it, (th * den).sum(), n_reads)
        # EM monotonicity: F = sum log S - sum theta*den, the second term is constant after a pass
        assert ll >= prev_ll - 1e-9 * abs(ll), (it, ll, prev_ll)
        prev_ll = ll
    # layout independence at full size: the CSR kernel walks the rows in the caller's order
    dev.reset_theta()
    dev.run_passes(3)
    a = dev.get_theta()
    for layout in (LAYOUT_CSR,):
        dev.upload_structure(s["n_tx"], s["row_ptr"], s["col_idx"], layout)
        dev.upload_sample(None, None, den)
        dev.run_passes(3)
        b = dev.get_theta()
        bad = np.flatnonzero(~(np.abs(a - b) <= 1e-10 * np.abs(a) + 1e-300))
        assert bad.size == 0, (bad.size, bad[:5], a[bad[:5]], b[bad[:5]])


def _avail_gib():
    for line in open("/proc/meminfo"):
        if line.startswith("MemAvailable:"):
            return int(line.split()[1]) / 2 ** 20
    return 0.0


def _full_size_properties(dev, s, passes, csr_passes, solve_tol=0.0):
    """Properties that need no oracle, at BASELINE's full size: mass conservation after every M-step, monotone F, finite
    non-negative theta, layout independence against the CSR kernel (other row order, other number of adds) and -- BASELINE's metric
    says 'to convergence', config 5 'convergence to 1e-6' -- the whole solve to solve_tol (bench.py's solve_to_convergence)."""
    n_reads, den = s["n_reads"], s["den"]
    dev.upload_structure(s["n_tx"], s["row_ptr"], s["col_idx"], LAYOUT_TILED)
    dev.upload_sample(None, None, den)
    prev_ll = -np.inf
    for _ in range(passes):
        _, ll = dev.run_passes(1, want_loglik=True)                # ll is evaluated at the pass input
        th = dev.get_theta()
        assert np.isfinite(th).all() and (th >= 0).all()
        assert abs((th * den).sum() - n_reads) <= 1e-9 * n_reads
        assert ll >= prev_ll - 1e-9 * abs(ll)
        prev_ll = ll
    info = dev.info()
    solved = None
    if solve_tol > 0:
        th, st = dev.solve(set_mode=1, max_iter=20000, accel=1, tol=solve_tol, abs_floor=0.01, check_every=4)
        assert st.converged == 1, (st.iters, st.final_delta)
        assert np.isfinite(th).all() and (th >= 0).all()
        assert abs((th * den).sum() - n_reads) <= 1e-9 * n_reads
        assert np.isfinite(st.loglik)
        solved = (info, st.iters, st.kernel_ms)
        print("solve to %g: %d passes, %.3f s of device time" % (solve_tol, st.iters, st.kernel_ms / 1e3))
    dev.reset_theta()
    dev.run_passes(csr_passes)
    a = dev.get_theta()
    dev.upload_structure(s["n_tx"], s["row_ptr"], s["col_idx"], LAYOUT_CSR)
    dev.upload_sample(None, None, den)
    dev.run_passes(csr_passes)
    b = dev.get_theta()
    assert np.all(np.abs(a - b) <= 1e-10 * np.abs(a) + 1e-300)
    return solved


@pytest.mark.parametrize("structure", synth.STRUCTURES)
def test_full_size_properties_cfg3(dev, structure):
    """BASELINE config 3 at full size (50M reads x 200k transcripts, nnz 251M): about 6 GiB of host arrays.  On all three row laws:
    the block-entry encoder's non-contiguous paths (one entry per hit, subsets with gaps, many far entries) at a size where tiles,
    units and far lists fill, and the library's own numbering on the shuffled one."""
    if _avail_gib() < 24:
        pytest.skip("needs 24 GiB of free host memory")
    s = synth.make_config("cfg3", 1.0, structure)
    info, _, _ = _full_size_properties(dev, s, passes=4 if structure == "window" else 2, csr_passes=3 if structure == "window" else 2, solve_tol=1e-6)
    if structure == "family_shuffled":                               # numbered by co-occurrence, the shuffled matrix packs like the unshuffled one
        assert info["renumbered"] == 1 and info["tiled_ids"] > 1.4 * info["tiled_entries"], info


def test_full_size_properties_cfg5(dev):
    """BASELINE config 5 at full size (200M reads x 250k transcripts, 20 alignments per read: nnz 4.0e9 -> 16 GB of column ids alone;
    generating it and laying it out needs about 100 GiB of host memory): gated on the memory the box has."""
    if _avail_gib() < 120:
        pytest.skip("needs 120 GiB of free host memory")
    s = synth.make_config("cfg5", 1.0)
    _full_size_properties(dev, s, passes=2, csr_passes=2, solve_tol=1e-6)          # config 5: 'convergence to 1e-6' (3600-4400 passes, ~4 s of device time)


@pytest.mark.parametrize("multi", ["0", "2", "5"])
def test_loglik_of_unweighted_rows_at_extreme_theta(monkeypatch, multi):
    """The unweighted TILED kernels take the log of a product of six row sums instead of six logs; a product that
    leaves the double range falls back to the per-row logs.  theta of 1e-70 / 1e+70 everywhere (products 1e-420 / 1e+420),
    and a mixture in which single transcripts sit at 1e-300, must give the oracle's likelihood."""
    s = synth.make_config("cfg3", 0.002)
    m = O.Csr(s["n_tx"], s["row_ptr"], s["col_idx"])
    rng = np.random.default_rng(8)
    base = rng.uniform(0.5, 2.0, size=s["n_tx"])
    mixed = base.copy()
    mixed[rng.random(s["n_tx"]) < 0.3] = 1e-300
    monkeypatch.setenv("EMSAR_HIP_TILED_MULTI", multi)
    with EmsarHip(0) as ctx:
        ctx.upload_structure(s["n_tx"], s["row_ptr"], s["col_idx"], LAYOUT_TILED)
        ctx.upload_sample(None, None, s["den"])
        for th in (base, base * 1e-70, base * 1e70, mixed):
            th = np.where(s["den"] > 0, th, 0.0)
            _, ll = m.em_step(th, s["den"], n_threads=4)
            ctx.set_theta(th)
            _, ll_dev = ctx.run_passes(1, want_loglik=True)
            assert np.isfinite(ll_dev) and abs(ll_dev - ll) <= 1e-11 * abs(ll) + 1e-9, (th[:3], ll_dev, ll)


def test_tiled_pair_and_single_kernels_match_oracle(monkeypatch):
    """Unweighted TILED passes run a unit of up to two tiles per workgroup (k_pass_tiled_unit) only when the tiles outnumber the
    chip's workgroup slots, one tile (k_pass_tiled) below that; the knob forces either kernel, or the two-independent-tiles
    kernel (k_pass_tiled_multi), on a matrix of a few dozen tiles.  The
    solve is long enough for the hipGraph replay of the SQUAREM cycle to start (4 x check_every cycles in)."""
    s = synth.make_config("cfg3", 0.004)
    m = O.Csr(s["n_tx"], s["row_ptr"], s["col_idx"])
    want = np.ones(s["n_tx"])
    for _ in range(3):
        want, _ = m.em_step(want, s["den"], n_threads=4)
    F = {}
    for multi, graph in (("0", "1"), ("2", "1"), ("2", "0"), ("5", "1")):       # one tile / two tiles / a unit (shared dictionary) per workgroup
        monkeypatch.setenv("EMSAR_HIP_TILED_MULTI", multi)
        monkeypatch.setenv("EMSAR_HIP_GRAPH", graph)
        with EmsarHip(0) as ctx:
            ctx.upload_structure(s["n_tx"], s["row_ptr"], s["col_idx"], LAYOUT_TILED)
            ctx.upload_sample(None, None, s["den"])
            assert ctx.info()["n_chunks"] > 3
            ctx.run_passes(3)
            got = ctx.get_theta()
            assert np.all(np.abs(got - want) <= 1e-11 * np.abs(want) + 1e-300), (multi, graph)
            th, st = ctx.solve(max_iter=3000, accel=1, tol=1e-9, set_mode=1)
            assert st.iters > 4 * 8 * 3 and (st.converged == 1 or st.iters >= 2990)
            assert abs((th * s["den"]).sum() - s["n_reads"]) < 1e-9 * s["n_reads"]       # the EM map conserves the read mass
            F[(multi, graph)] = st.loglik
    ref = F[("0", "1")]
    assert all(abs(v - ref) <= 1e-8 * abs(ref) for v in F.values()), F


def test_csr_with_64bit_row_pointers(monkeypatch):
    """Config 5 at full size has nnz = 4e9, just under 2^32; above it the CSR kernels read 64-bit row pointers.  The
    test hook forces that code path on a small matrix: same pass, same solve."""
    s = synth.make_config("cfg5", 0.0005)
    m = O.Csr(s["n_tx"], s["row_ptr"], s["col_idx"])
    want, _ = m.em_step(np.ones(s["n_tx"]), s["den"], n_threads=4)
    monkeypatch.setenv("EMSAR_HIP_FORCE_PTR64", "1")
    ctx = EmsarHip(0)
    try:
        ctx.upload_structure(s["n_tx"], s["row_ptr"], s["col_idx"], LAYOUT_CSR)
        ctx.upload_sample(None, None, s["den"])
        ctx.run_passes(1)
        got = ctx.get_theta()
        assert np.all(np.abs(got - want) <= 1e-11 * np.abs(want) + 1e-300)
        info = ctx.info()
        assert info["bytes_per_pass"] == 4 * len(s["col_idx"]) + 8 * (s["n_reads"] + 1) + 32 * s["n_tx"]      # P = 8
    finally:
        ctx.close()
