"""GPU: the C command-line driver emsar-hip end to end (C host -> C ABI -> HIP kernels -> .fpkm) against the
reference's own output files, single-sample and -M multi-sample."""
import glob
import os
import subprocess

import numpy as np
import pytest

import oracle as O
from emsar_amd import _build
from tests.conftest import CASES, aln_path, get_fixture

pytestmark = pytest.mark.gpu
CLI = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "emsar_amd", "emsar-hip")


@pytest.fixture(scope="module", autouse=True)
def _built():
    _build.build_all()
    assert os.path.exists(CLI)


def _aln(fx):
    return aln_path(fx.dir)[0]


def _check_fpkm_file(fx, path):
    got = O.read_fpkm(path)
    ref = fx.runs[0]
    assert got["names"] == ref["names"]
    fx.check_fpkm_parity(got["fpkm"], "emsar-hip " + fx.case)
    assert (got["sd"] == 0).all()                                          # documented deviation (deterministic EM)
    assert np.abs(got["efflen"] - ref["efflen"]).max() <= 1.01e-6           # column 4: deterministic, 6 decimals
    mask = fx.noise_mask()
    ok = ~mask
    assert np.all(np.abs(got["ireadcount"] - ref["ireadcount"])[ok] <= 1e-5 * np.abs(ref["ireadcount"])[ok] + 2e-3)
    assert np.all(np.abs(got["ireadcount_int"] - ref["ireadcount_int"])[ok] <= 1)
    assert abs(got["tpm"].sum() - 1e6) < 1.0
    # stationarity: inferred reads = reads inside the likelihood
    inside = fx.model.R[fx.model.E != 0].sum()
    assert abs(got["ireadcount"].sum() - inside) <= 1e-5 * inside + 1e-2


@pytest.mark.parametrize("case", CASES)
def test_single_sample(case, tmp_path):
    fx = get_fixture(case)
    cmd = [CLI, "-q", "-g"] + fx.meta["opts"] + ["-I", os.path.join(fx.dir, "index.rsh"), str(tmp_path), "out", _aln(fx)]
    subprocess.run(cmd, check=True, timeout=300)
    _check_fpkm_file(fx, str(tmp_path / "out.0.fpkm"))
    assert open(tmp_path / "out.0.fraglength_effect").read() == open(os.path.join(fx.dir, "ref.run0.fraglength_effect")).read()
    seg = open(tmp_path / "out.0.segments").read().splitlines()
    ref = open(os.path.join(fx.dir, "ref.run0.segments")).read().splitlines()
    assert len(seg) == len(ref)
    for a, b in zip(seg[1:], ref[1:]):
        assert a.split("\t")[:6] == b.split("\t")[:6]


def test_multisample_list(tmp_path):
    """-M: the three SE cases that share an option set run as one job; file i of the list -> out.i.fpkm."""
    fxs = [get_fixture(c) for c in ("syn300_se", "syn2k_se")]
    # samples of one job share ONE rsh in the reference; give each rsh its own job but several files per job
    for fx in fxs:
        lst = tmp_path / (fx.case + ".list")
        lst.write_text("\n".join([_aln(fx)] * 3) + "\n")
        out = tmp_path / fx.case
        stats = tmp_path / (fx.case + ".json")
        subprocess.run([CLI, "-q", "-M", "--gpus", "1", "--stats-json", str(stats), "-I", os.path.join(fx.dir, "index.rsh"),
                        str(out), "ms", str(lst)], check=True, timeout=600)
        files = sorted(glob.glob(str(out / "ms.*.fpkm")))
        assert [os.path.basename(f) for f in files] == ["ms.0.fpkm", "ms.1.fpkm", "ms.2.fpkm"]
        # independent samples of identical input: equal up to the summation order of the FP64 atomics
        vals = [O.read_fpkm(f)["fpkm"] for f in files]
        for v in vals[1:]:
            assert np.all(np.abs(v - vals[0]) <= 1e-6 * np.abs(vals[0]) + 1.5e-6)
        _check_fpkm_file(fx, files[0])
        import json
        st = json.load(open(stats))
        assert st["samples"] == 3 and st["failed"] == 0 and all(s["converged"] == 1 for s in st["per_sample"])


def test_multisample_list_with_broken_samples(tmp_path):
    """-M with a missing and an empty alignment file between good ones: the good samples are written (the alignments of
    sample i+1 are counted while sample i is solved, so a failure surfaces one step ahead), the job reports two failures
    and exits non-zero like the reference does on its first bad sample -- but after finishing the others."""
    import json
    fx = get_fixture("syn300_se")
    empty = tmp_path / "empty.bowtie"
    empty.write_text("")
    lst = tmp_path / "mixed.list"
    lst.write_text("\n".join([_aln(fx), str(tmp_path / "missing.bowtie"), _aln(fx), str(empty), _aln(fx)]) + "\n")
    out = tmp_path / "mixed"
    stats = tmp_path / "mixed.json"
    r = subprocess.run([CLI, "-q", "-M", "--gpus", "1", "--stats-json", str(stats), "-I", os.path.join(fx.dir, "index.rsh"),
                        str(out), "ms", str(lst)], capture_output=True, timeout=600)
    assert r.returncode != 0
    files = sorted(os.path.basename(f) for f in glob.glob(str(out / "ms.*.fpkm")))
    assert files == ["ms.0.fpkm", "ms.2.fpkm", "ms.4.fpkm"]
    st = json.load(open(stats))
    assert st["samples"] == 5 and st["failed"] == 2
    assert [int(s["status"] != 0) for s in st["per_sample"]] == [0, 1, 0, 1, 0]
    for i in (0, 2, 4):
        _check_fpkm_file(fx, str(out / ("ms.%d.fpkm" % i)))
    assert b"alnfile[1]" in r.stderr and b"alnfile[3]" in r.stderr


def test_cli_errors(tmp_path):
    fx = get_fixture("toy5_se50")
    r = subprocess.run([CLI, "-q", "-I", str(tmp_path / "nope.rsh"), str(tmp_path), "o", _aln(fx)], capture_output=True)
    assert r.returncode != 0 and b"can't open input rsh file" in r.stderr
    r = subprocess.run([CLI, "-q", "-s", "bogus", "-I", os.path.join(fx.dir, "index.rsh"), str(tmp_path), "o", _aln(fx)],
                       capture_output=True)
    assert r.returncode != 0 and b"invalid strand type" in r.stderr
    r = subprocess.run([CLI], capture_output=True)
    assert r.returncode != 0 and b"Usage" in r.stderr


def test_rsh_cache_and_streaming_only_give_the_same_files(tmp_path):
    """--rsh-cache: the first run parses the text and writes the binary cache, the second reads it; --streaming-only
    solves the whole matrix with the streaming passes instead of set by set.  All three write the same numbers."""
    fx = get_fixture("syn2k_se")
    cache = str(tmp_path / "idx.bin")
    outs = []
    for tag, extra in (("a", ["--rsh-cache=" + cache]), ("b", ["--rsh-cache=" + cache]), ("c", ["--streaming-only"])):
        d = tmp_path / tag
        cmd = [CLI, "-g"] + extra + fx.meta["opts"] + ["-I", os.path.join(fx.dir, "index.rsh"), str(d), "out", _aln(fx)]
        r = subprocess.run(cmd, check=True, timeout=300, capture_output=True, text=True)
        outs.append(r.stdout)
        _check_fpkm_file(fx, str(d / "out.0.fpkm"))
    assert os.path.exists(cache) and "binary cache" not in outs[0] and "binary cache" in outs[1]
    a, b = O.read_fpkm(str(tmp_path / "a" / "out.0.fpkm")), O.read_fpkm(str(tmp_path / "b" / "out.0.fpkm"))
    np.testing.assert_array_equal(a["fpkm"], b["fpkm"])                  # resident sets: no atomics, bit-reproducible
    for k in ("efflen", "ireadcount", "tpm"):                              # eff.length is scattered with atomics: last printed digit may flip
        assert np.all(np.abs(a[k] - b[k]) <= 1e-9 * np.abs(a[k]) + 2e-6)
    c = O.read_fpkm(str(tmp_path / "c" / "out.0.fpkm"))
    assert np.abs(a["fpkm"] - c["fpkm"]).max() <= 1e-5 * np.abs(c["fpkm"]).max() + 2e-6
