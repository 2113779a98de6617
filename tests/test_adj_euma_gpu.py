"""GPU: compute_adjEUMA on the device (emsar_hip_upload_euma / emsar_hip_adj_euma) against the C host's loop.
Bit-exact: one lane per row, fragment lengths in ascending order, product and sum rounded separately."""
import os

import numpy as np
import pytest

from emsar_amd import EmsarHip, EmsarHipError, hostlib as HL
from tests.conftest import aln_path

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    ctx = EmsarHip(0)
    yield ctx
    ctx.close()


def test_fixtures_bit_identical(dev, golden):
    r = HL.HostRsh(os.path.join(golden.dir, "index.rsh"))
    opts = golden.meta["opts"]
    aln, fmt = aln_path(golden.dir)
    c = r.count(aln, pe=int("-P" in opts), fmt=fmt, max_repeat=int(opts[opts.index("-k") + 1]) if "-k" in opts else 100)
    m = r.model(c)
    dev.upload_structure(r.n_tx, r.row_ptr, r.col_idx)
    dev.upload_euma(r.euma)
    L = dev.adj_euma(r.wf(c))
    np.testing.assert_array_equal(L, m.L)
    m2 = r.model(c, L=L)
    np.testing.assert_array_equal(m2.E, m.E)


@pytest.mark.parametrize("n_rows,nfl", [(1, 1), (63, 7), (1000, 64), (70001, 401), (5, 1000)])
def test_random_shapes_bit_identical(dev, n_rows, nfl):
    rng = np.random.default_rng(n_rows + nfl)
    euma = rng.integers(0, 5000, size=(n_rows, nfl)).astype(np.int32)
    euma[rng.random(n_rows) < 0.2] = 0
    wf = rng.random(nfl)
    wf /= wf.sum()
    rp = np.arange(n_rows + 1, dtype=np.uint64)
    dev.upload_structure(4, rp, np.zeros(n_rows, dtype=np.int32))
    dev.upload_euma(euma)
    got = dev.adj_euma(wf)
    want = np.zeros(n_rows)
    for i in range(nfl):                                          # the reference's order of operations, vectorised over rows
        want = want + wf[i] * euma[:, i].astype(np.float64)
    np.testing.assert_array_equal(got, want)
    got2 = dev.adj_euma(wf[::-1].copy())                          # a second sample re-uses the uploaded EUMA
    want2 = np.zeros(n_rows)
    for i in range(nfl):
        want2 = want2 + wf[nfl - 1 - i] * euma[:, i].astype(np.float64)
    np.testing.assert_array_equal(got2, want2)


def test_call_order(dev):
    fresh = EmsarHip(0)
    with pytest.raises(EmsarHipError):
        fresh._chk(fresh._L.emsar_hip_adj_euma(fresh._h, None, None), "adj_euma")
    fresh.upload_structure(2, [0, 1], [1])
    fresh.n_rows = 1
    fresh.nfl = 3
    with pytest.raises(EmsarHipError) as e:
        fresh.adj_euma(np.ones(3))
    assert e.value.status == -5                                   # no EUMA uploaded yet
    fresh.close()
