"""GPU: randomised differential test -- many small, ugly matrices (empty rows, repeated ids inside a row, rows longer
than a tile holds, cross-family rows, zero weights, rows outside the likelihood, transcripts nobody names) through every
layout against the CPU oracle, pass by pass and to convergence."""
import os

import numpy as np
import pytest

import oracle as O
from emsar_amd import EmsarHip
from emsar_amd.hip import FLAG_MERGE_ROWS, LAYOUT_CSR, LAYOUT_TILED

pytestmark = pytest.mark.gpu
LAYOUTS = [LAYOUT_CSR, LAYOUT_TILED, LAYOUT_TILED | FLAG_MERGE_ROWS]


def random_problem(seed):
    rng = np.random.default_rng(seed)
    n_tx = int(rng.integers(3, 2500))
    n_rows = int(rng.integers(1, 4000))
    fam = int(rng.integers(1, 40))
    rows = []
    for _ in range(n_rows):
        u = rng.random()
        if u < 0.03:
            k = 0
        elif u < 0.45:
            k = 1
        elif u < 0.97:
            k = int(rng.integers(2, 12))
        elif u < 0.995:
            k = int(rng.integers(12, 120))
        else:
            k = int(rng.integers(769, 1500))                      # longer than a tile's dictionary
        base = int(rng.integers(0, n_tx))
        t = (base + rng.integers(0, fam, size=k)) % n_tx           # in-family ids, repeats allowed
        if k >= 2 and rng.random() < 0.1:
            t[rng.integers(0, k)] = rng.integers(0, n_tx)          # a cross-family hit
        rows.append(t)
    rp = np.zeros(n_rows + 1, dtype=np.uint64)
    rp[1:] = np.cumsum([len(r) for r in rows])
    ci = (np.concatenate(rows) if rp[-1] else np.zeros(0)).astype(np.int32)
    R = rng.integers(0, 30, size=n_rows).astype(np.int32)
    R[rng.random(n_rows) < 0.2] = 0
    E = rng.uniform(0.1, 3.0, size=n_rows)
    E[rng.random(n_rows) < 0.1] = 0.0
    E[np.diff(rp.astype(np.int64)) == 0] = 0.0                     # an empty row with reads has likelihood 0 for every theta (Fp clamps
    return n_tx, rp, ci, R, E                                       # it to -9.9e307): in an rsh such rows have no EUMA, i.e. E = 0


@pytest.fixture(scope="module")
def dev():
    ctx = EmsarHip(0)
    yield ctx
    ctx.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("EMSAR_RANDOM_SEEDS", "10"))))     # a longer campaign: EMSAR_RANDOM_SEEDS=60
def test_passes_and_solve_match_the_oracle(dev, seed):
    n_tx, rp, ci, R, E = random_problem(1000 + seed)
    weighted = seed % 3 != 0
    m = O.Csr(n_tx, rp, ci, R=R if weighted else np.ones(len(R), dtype=np.int32), E=E)
    den = m.den()
    want = np.where(den > 0, 1.0, 0.0)
    for _ in range(3):
        want, _ = m.em_step(want, den)
    th_o, _ = m.em_solve(max_iter=3000, tol=1e-8)
    F_o = m.loglik(th_o)
    for layout in LAYOUTS:
        dev.upload_structure(n_tx, rp, ci, layout)
        dev.upload_sample(R if weighted else None, E, None)
        dev.run_passes(3)
        got = dev.get_theta()
        assert np.all(np.abs(got - want) <= 1e-11 * np.abs(want) + 1e-300), (seed, layout)
        for set_mode in (0, 1):
            th, st = dev.solve(max_iter=3000, tol=1e-8, set_mode=set_mode)
            assert np.isfinite(th).all()
            F = m.loglik(th)
            assert abs(st.loglik - F) <= 1e-9 * abs(F) + 1e-9, (seed, layout, set_mode)
            if np.isfinite(F_o) and F_o != 0:
                assert F >= F_o - 1e-6 * abs(F_o) - 1e-6, (seed, layout, set_mode)      # no worse than the oracle's EM at the same budget
