"""GPU: deterministic mode of the streaming passes (emsar_hip_set_deterministic, include/emsar_hip.h).

Every sum that workgroups share is a 64-bit fixed-point integer, so the arrival order of the adds no longer shows: two solves of
the same input must be BIT-identical, and so must the tile kernel and the CSR kernel (different row order, different number of
adds per transcript).  The values stay within the resolution stated in the header of the oracle's."""
import numpy as np
import pytest

import oracle as O
from emsar_amd import EmsarHip, synth
from emsar_amd.hip import LAYOUT_CSR, LAYOUT_TILED

pytestmark = pytest.mark.gpu


def _solve(s, layout, det, **kw):
    with EmsarHip(0) as ctx:
        ctx.set_deterministic(det)
        ctx.upload_structure(s["n_tx"], s["row_ptr"], s["col_idx"], layout)
        ctx.upload_sample(None, None, s["den"])
        return ctx.solve(set_mode=1, **kw)


@pytest.mark.parametrize("structure", synth.STRUCTURES)
def test_two_solves_are_bit_identical(structure, monkeypatch):
    if structure == "family_shuffled":
        monkeypatch.setenv("EMSAR_HIP_RENUMBER", "2")             # the library's own transcript numbering on
    s = synth.make_config("cfg3", 0.004, structure)
    runs = [_solve(s, LAYOUT_TILED, True, max_iter=3000, accel=1, tol=1e-8) for _ in range(3)]
    th0, st0 = runs[0]
    assert st0.iters > 100 and (st0.converged == 1 or st0.iters >= 2990)
    for th, st in runs[1:]:
        np.testing.assert_array_equal(th, th0)
        assert st.iters == st0.iters and st.loglik == st0.loglik and st.final_delta == st0.final_delta
    # plain EM too (no SQUAREM scalars involved), and the graph replay does not change the bits
    a, sa = _solve(s, LAYOUT_TILED, True, max_iter=400, accel=0, tol=0.0)
    b, sb = _solve(s, LAYOUT_TILED, True, max_iter=400, accel=0, tol=0.0)
    np.testing.assert_array_equal(a, b)
    assert sa.loglik == sb.loglik


def test_passes_agree_across_kernels_to_the_fixed_point_resolution(monkeypatch):
    """The one-tile kernel, the two-tile kernel and the CSR kernel add the same masses in different groupings (sums of 6 or 12 rows
    inside a lane, or one row at a time), each rounded once to the fixed-point grid: across kernels the inferred reads agree to a
    small multiple of the grid, N * 2^-61 reads; within one kernel the bits are the same (test above)."""
    s = synth.make_config("cfg3", 0.004)
    out = {}
    for name, layout, multi in (("tiled1", LAYOUT_TILED, "0"), ("tiled2", LAYOUT_TILED, "2"), ("unit", LAYOUT_TILED, "5"), ("csr", LAYOUT_CSR, "0")):
        monkeypatch.setenv("EMSAR_HIP_TILED_MULTI", multi)
        with EmsarHip(0) as ctx:
            ctx.set_deterministic(True)
            ctx.upload_structure(s["n_tx"], s["row_ptr"], s["col_idx"], layout)
            ctx.upload_sample(None, None, s["den"])
            _, ll = ctx.run_passes(5, want_loglik=True)
            out[name] = (ctx.get_theta(), ll)
    # every contribution is rounded to the grid once (N * 2^-61 reads = 8.7e-14 here): a transcript with k contributions is within
    # k / 2 grid steps of the exact sum, and the CSR kernel makes one contribution per row where the tile kernel makes one per 11
    res = s["n_reads"] * 2.0 ** -61 * 20000
    for other in ("tiled2", "unit", "csr"):
        assert np.all(np.abs(out["tiled1"][0] - out[other][0]) * s["den"] <= res + 1e-11 * out[other][0] * s["den"]), other
        assert abs(out["tiled1"][1] - out[other][1]) <= 1e-11 * abs(out[other][1])


@pytest.mark.parametrize("name,scale,structure", [("cfg2", 0.05, "window"), ("cfg3", 0.004, "window"), ("cfg5", 0.0005, "window"),
                                                  ("cfg3", 0.004, "family"), ("cfg3", 0.004, "family_shuffled")])
def test_deterministic_passes_match_oracle(name, scale, structure, monkeypatch):
    """Same EM map as the oracle's: after two passes every transcript's inferred reads agree to the fixed-point resolution, and
    the log-likelihood to 1e-10."""
    if structure == "family_shuffled":
        monkeypatch.setenv("EMSAR_HIP_RENUMBER", "2")
    s = synth.make_config(name, scale, structure)
    m = O.Csr(s["n_tx"], s["row_ptr"], s["col_idx"])
    den = s["den"]
    want, _ = m.em_step(np.ones(s["n_tx"]), den, n_threads=4)
    want2, ll = m.em_step(want, den, n_threads=4)
    for layout in (LAYOUT_TILED, LAYOUT_TILED | 0x100, LAYOUT_CSR):
        with EmsarHip(0) as ctx:
            ctx.set_deterministic(True)
            ctx.upload_structure(s["n_tx"], s["row_ptr"], s["col_idx"], layout)
            ctx.upload_sample(None, None, den)
            ctx.run_passes(1)
            _, ll_dev = ctx.run_passes(1, want_loglik=True)
            got = ctx.get_theta()
        assert abs(ll_dev - ll) <= 1e-10 * abs(ll)
        assert np.all(np.abs(got - want2) * den <= s["n_reads"] * 2.0 ** -61 * 20000 + 1e-11 * want2 * den), layout
        assert abs((got * den).sum() - s["n_reads"]) <= 1e-9 * s["n_reads"]


def test_weighted_rows_and_solve_quality():
    """Segment-level sample (weights R, lengths E): deterministic and default solves reach the same optimum."""
    s = synth.make_matrix(n_tx=3000, n_reads=200000, law="human", xfam=0.02, seed=21)
    with EmsarHip(0) as ctx:
        rp, ci, w, _, _ = ctx.collapse_rows(s["n_tx"], s["row_ptr"], s["col_idx"])
        res = {}
        for det in (False, True, True):
            ctx.set_deterministic(det)
            ctx.upload_structure(s["n_tx"], rp, ci, LAYOUT_TILED)
            ctx.upload_sample(w, None, s["den"])
            res.setdefault(det, []).append(ctx.solve(set_mode=1, max_iter=5000, accel=1, tol=1e-9))
        ctx.set_deterministic(False)
    (th_a, st_a), = res[False]
    (th_b, st_b), (th_c, st_c) = res[True]
    np.testing.assert_array_equal(th_b, th_c)
    assert st_b.iters == st_c.iters
    assert abs(st_a.loglik - st_b.loglik) <= 1e-9 * abs(st_a.loglik)
    big = th_a * s["den"] > 1.0
    assert np.all(np.abs(th_a[big] - th_b[big]) <= 1e-5 * th_a[big])
