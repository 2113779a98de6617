import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
CASES = ["vicugna_pe", "toy5_se50", "toy5_sam", "toy5_bam", "toy5_pe", "toy5_pe_sam", "toy5_pe_bam", "syn300_se", "syn300_k2", "syn2k_se"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver on the GPU box)")


def aln_path(case_dir):
    """The alignment input of a fixture (bowtie/SAM text gzipped, or BAM) and its emsar format code."""
    import glob
    p = (glob.glob(os.path.join(case_dir, "reads.*.gz")) + glob.glob(os.path.join(case_dir, "reads.bam")))[0]
    return p, (2 if p.endswith(".bam") else 1 if ".sam" in p else 0)


class Fixture:
    """One golden case: our synthetic inputs + the files the compiled reference wrote for them."""

    def __init__(self, case):
        import oracle as O
        self.case = case
        self.dir = os.path.join(GOLDEN, case)
        self.meta = json.load(open(os.path.join(self.dir, "meta.json")))
        self.rsh = O.read_rsh(os.path.join(self.dir, "index.rsh"))
        self.n_tx = len(self.rsh["names"])
        self.seg, self.cs_ref, self.expected_counts = O.read_segments(
            os.path.join(self.dir, "ref.run0.segments"), self.n_tx)
        self.frag_lens, self.frag_counts = O.read_fraglength_effect(
            os.path.join(self.dir, "ref.run0.fraglength_effect"))
        # TotalReadCount and FraglengthCounts are incremented together (emsar_functions.c:940-941)
        self.N = int(self.frag_counts.sum())
        self.model = O.model_from_fixture(self.rsh, self.frag_counts, self.seg.R, self.N)
        self.runs = [O.read_fpkm(os.path.join(self.dir, "ref.run%d.fpkm" % r)) for r in range(self.meta["runs"])]
        self.seeded = O.read_fpkm(os.path.join(self.dir, "ref.seed%d.fpkm" % self.meta["seed"]))

    def noise_mask(self):
        """transcripts on which the reference disagrees with itself (random starts / ties)."""
        a = np.array([r["fpkm"] for r in self.runs])
        return (a.max(0) - a.min(0)) > 1e-6 * np.abs(a).max(0) + 1.5e-6

    def ref_fpkm(self):
        return self.runs[0]["fpkm"]

    def check_fpkm_parity(self, theta, what="theta"):
        """The binding criterion of SURVEY.md 8c: likelihood parity + per-transcript parity off the mask +
        group-sum parity per connected set (covers tied groups)."""
        m = self.model
        ref = self.ref_fpkm()
        F_ref = max(m.loglik(r["fpkm"]) for r in self.runs)
        F = m.loglik(theta)
        assert F >= F_ref - 1e-9 * abs(F_ref), (what, F, F_ref)
        mask = self.noise_mask()
        err = np.abs(theta - ref)
        tol = 1e-5 * np.abs(ref) + 1.5e-6
        bad = (err > tol) & ~mask
        assert not bad.any(), (what, np.nonzero(bad)[0][:10], err[bad][:10], ref[bad][:10])
        # tied / weakly determined transcripts: compare what the data does determine, the expected count
        # of every segment, E_c * sum theta (column 7 of .segments is that quantity at the reference's mean)
        lam = m.E * np.add.reduceat(np.append(theta[m.col_idx], 0.0), m.row_ptr[:-1].astype(np.int64))[: m.n_rows]
        lam[np.diff(m.row_ptr.astype(np.int64)) == 0] = 0
        lam_ref = self.expected_counts
        assert np.all(np.abs(lam - lam_ref) <= 1e-5 * np.abs(lam_ref) + 2e-3), what


_cache = {}


@pytest.fixture(params=CASES)
def golden(request):
    if request.param not in _cache:
        _cache[request.param] = Fixture(request.param)
    return _cache[request.param]


def get_fixture(case):
    if case not in _cache:
        _cache[case] = Fixture(case)
    return _cache[case]
