"""Pin the CPU oracle (oracle/em_oracle.c) against the compiled reference's outputs in tests/golden/.

CPU only.  These tests are what allows the oracle to stand in for the reference on the GPU box:
  * the connected sets equal column 2 of the reference's .segments file,
  * eff.length (iEUMA) equals column 4 of .fpkm,
  * the clean-room pattern-search MLE reproduces the SEEDED reference run (time() pinned, -p 1) to the
    print quantum on every column of .fpkm -- i.e. the restatement is the reference's algorithm,
  * the EM fixed point satisfies the parity criterion of SURVEY.md 8c against the unseeded runs.
"""
import numpy as np
import pytest

import oracle as O

HALF_QUANTUM = 5.01e-7  # the reference prints %lf (6 decimals)


def test_model_matches_segments_file(golden):
    m, seg = golden.model, golden.seg
    assert (m.col_idx == seg.col_idx).all() and (m.row_ptr == seg.row_ptr).all()
    assert np.abs(m.L - seg.L).max() <= HALF_QUANTUM            # adjEUMA, printed with 6 decimals
    if golden.meta["total_read_count"] is not None:
        assert golden.N == golden.meta["total_read_count"]


def test_components_match_reference(golden):
    n, cs, ts, cut = golden.model.components()
    assert cut == 0.0
    assert (cs == golden.cs_ref).all()
    assert n == golden.cs_ref.max() + 1


def test_ieuma_matches_fpkm_column4(golden):
    ie = golden.model.ieuma()
    assert np.abs(ie - golden.runs[0]["efflen"]).max() <= HALF_QUANTUM


def test_pattern_search_reproduces_seeded_reference(golden):
    m = golden.model
    n, cs, _, _ = m.components()
    rounds = []
    for r in range(4):                                           # NUM_ROUND=4, one srand before the rounds
        th, _ = m.mle_pattern_search(cs, n, seed=golden.meta["seed"] if r == 0 else 0, n_threads=1)
        rounds.append(th)
    mean, sd, ir, iri, tpm = O.fpkm_table(np.array(rounds), m.ieuma(), golden.N)
    s = golden.seeded
    assert np.abs(mean - s["fpkm"]).max() <= HALF_QUANTUM
    assert np.abs(sd - s["sd"]).max() <= HALF_QUANTUM
    assert np.abs(ir - s["ireadcount"]).max() <= HALF_QUANTUM
    assert (iri == s["ireadcount_int"]).all()
    assert np.abs(tpm - s["tpm"]).max() <= HALF_QUANTUM


def test_pattern_search_threads_same_answer_within_noise(golden):
    m = golden.model
    n, cs, _, _ = m.components()
    th, sweeps = m.mle_pattern_search(cs, n, seed=7, n_threads=4)
    assert sweeps >= 0
    golden.check_fpkm_parity(th, "pattern-search -p 4")


@pytest.mark.parametrize("accel", [0, 1])
def test_em_fixed_point_meets_parity_criterion(golden, accel):
    th, st = golden.model.em_solve(max_iter=400000, accel=accel, tol=1e-10)
    assert st.converged
    golden.check_fpkm_parity(th, "oracle EM accel=%d" % accel)


def test_em_invariants(golden):
    m = golden.model
    den = m.den()
    th = np.where(den > 0, 1.0, 0.0)
    in_lik = (m.E != 0)
    total_R = m.R[in_lik].sum()
    prev = -np.inf
    for _ in range(20):
        th, ll = m.em_step(th, den)
        # mass conservation: sum_t theta_t den_t = sum_c R_c after every M-step
        assert abs((th * den).sum() - total_R) <= 1e-9 * total_R
        F = m.loglik(th)
        assert F >= prev - 1e-9 * abs(F)                         # EM is monotone in F
        prev = F
    assert (th >= 0).all()


def test_em_known_answers():
    # single-segment / single-tid set: theta = R/E  (emsar_functions.c:3062-3066)
    m = O.Csr(3, [0, 1, 2, 4], [0, 1, 1, 2], R=[7, 0, 0], E=[2.0, 3.0, 1.5])
    th, st = m.em_solve(max_iter=1000, accel=0, tol=1e-14)
    assert th[0] == 7 / 2.0
    # all-zero-count set -> 0  (emsar_functions.c:3054-3059)
    assert th[1] == 0 and th[2] == 0
    # row with E == 0 is outside the likelihood; a tid seen only there is defined as 0
    m = O.Csr(2, [0, 1, 2], [0, 1], R=[5, 9], E=[1.0, 0.0])
    th, _ = m.em_solve(max_iter=100, accel=1, tol=1e-14)
    assert th[0] == 5.0 and th[1] == 0.0
    # duplicate tid in a row counts twice (A2): lambda = E*(2 theta) -> theta = R/(2E)
    m = O.Csr(1, [0, 2], [0, 0], R=[8], E=[2.0])
    th, _ = m.em_solve(max_iter=100, accel=0, tol=1e-14)
    assert abs(th[0] - 2.0) < 1e-14


def test_fpkm_table_properties(golden):
    th, _ = golden.model.em_solve(max_iter=400000, accel=1, tol=1e-10)
    mean, sd, ir, iri, tpm = O.fpkm_table(np.array([th] * 4), golden.model.ieuma(), golden.N)
    assert abs(tpm.sum() - 1e6) < 1e-3
    # stationarity: total inferred read count = reads inside the likelihood
    m = golden.model
    assert abs(ir.sum() - m.R[m.E != 0].sum()) <= 1e-6 * golden.N + 1e-6
