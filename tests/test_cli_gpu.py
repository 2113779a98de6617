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
    mask = fx.noise_mask()
    for a, b in zip(seg[1:], ref[1:]):
        fa, fb = a.split("\t"), b.split("\t")
        assert fa[:6] == fb[:6]
        # column 7, the expected read count of the segment, from the written file: sum_t FPKM_t * L_c / 1e3 * N / 1e6 -- comparable
        # wherever the reference agrees with itself on all the segment's transcripts
        tids = [int(t[1:]) for t in fa[2].split(",")]
        if not mask[tids].any():
            assert abs(float(fa[6]) - float(fb[6])) <= 1e-5 * abs(float(fb[6])) + 2e-3, (fa[0], fa[6], fb[6])


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


def _chain_rsh(path, n=5003):
    """A chain of n transcripts linked by two-transcript segments, two fragment lengths (30, 31).  Every link has 50 positions
    at both lengths except the middle one: 1 position for 30-nt fragments, 50 for 31-nt ones.  A sample of 30-nt reads sees one
    weak link (EUMAcut 0 -> 2 drops it, two sets remain); a sample of 31-nt reads sees none (EUMAcut climbs to 52 until every
    link is gone).  EUMAcut is never reset between the samples of one -M job (emsar_main.c:95,418)."""
    lines = ["#%d,2,30,31,-1" % (n - 1)] + ["@%d\tt%d" % (i, i) for i in range(n)]
    lines.append("cid\tno.tids\tfirst.tid\tother.tids\tsegment.length")
    for i in range(n):
        lines.append("%d\t1\t%d\t\t100,100," % (i, i))
    for i in range(n - 1):
        lines.append("%d\t2\t%d\t%d,\t%d,50," % (n + i, i, i + 1, 1 if i == 2500 else 50))
    open(path, "w").write("\n".join(lines) + "\n")


def _reads(path, n_tx, length, n=400, seed=0):
    rng = np.random.default_rng(seed)
    with open(path, "w") as f:
        for i in range(n):
            f.write("r%d\t+\tt%d\t1\t%s\t%s\t0\t\n" % (i, int(rng.integers(n_tx)), "A" * length, "I" * length))


@pytest.mark.parametrize("extra", [[], ["--streaming-only"], ["--streaming-only", "--plain"]])
def test_two_runs_write_the_same_bytes(tmp_path, extra):
    """The reference's output changes from run to run (concurrent rand(), emsar_functions.c:3079).  Ours must not: the per-set solver
    has no atomics, the streaming passes add in fixed point (deterministic mode, the CLI default), den and iEUMA are summed on
    the host in row order -- every output file of two runs is the same bytes, also when the whole matrix goes through the
    streaming kernels."""
    fx = get_fixture("syn2k_se")
    blobs = []
    for tag in ("a", "b"):
        d = tmp_path / tag
        cmd = [CLI, "-q", "-g"] + extra + fx.meta["opts"] + ["-I", os.path.join(fx.dir, "index.rsh"), str(d), "out", _aln(fx)]
        subprocess.run(cmd, check=True, timeout=300)
        files = sorted(os.listdir(d))
        assert len(files) >= 3
        blobs.append([(f, open(os.path.join(d, f), "rb").read()) for f in files])
    assert blobs[0] == blobs[1]


def test_two_workers_share_one_card_with_eumacut_handover_and_a_failing_sample(tmp_path):
    """-M with two workers on device 0 (--devices 0,0): five samples, one of them missing.  The workers build their models
    concurrently from the EUMAcut they see and hand the value over in sample order; the 31-nt sample (index 0) raises it to 52,
    so the 30-nt samples that follow lose EVERY link -- unlike the same sample run on its own.  The outputs must equal those
    of the one-worker run file by file, whichever worker took which sample."""
    import json
    n = 5003
    rsh = str(tmp_path / "chain.rsh")
    _chain_rsh(rsh, n)
    a30, a31 = str(tmp_path / "a30.bowtie"), str(tmp_path / "a31.bowtie")
    _reads(a30, n, 30, seed=1)
    _reads(a31, n, 31, seed=2)
    lst = tmp_path / "five.list"
    lst.write_text("\n".join([a31, a30, str(tmp_path / "missing.bowtie"), a30, a31]) + "\n")
    outs = {}
    for tag, extra in (("one", ["--gpus", "1"]), ("two", ["--devices", "0,0"])):
        d = tmp_path / tag
        st = tmp_path / (tag + ".json")
        r = subprocess.run([CLI, "-q", "-g", "-M"] + extra + ["--stats-json", str(st), "-I", rsh, str(d), "ms", str(lst)],
                           capture_output=True, timeout=900)
        assert r.returncode != 0 and b"alnfile[2]" in r.stderr
        js = json.load(open(st))
        assert js["samples"] == 5 and js["failed"] == 1 and js["gpus"] == (1 if tag == "one" else 2)
        assert [int(s["status"] != 0) for s in js["per_sample"]] == [0, 0, 1, 0, 0]
        outs[tag] = d
        assert sorted(os.path.basename(f) for f in glob.glob(str(d / "ms.*.fpkm"))) == ["ms.%d.fpkm" % i for i in (0, 1, 3, 4)]
    for i in (0, 1, 3, 4):
        for ext in ("fraglength_effect", "segments"):
            assert open(outs["one"] / ("ms.%d.%s" % (i, ext))).read() == open(outs["two"] / ("ms.%d.%s" % (i, ext))).read(), (i, ext)
        a, b = O.read_fpkm(str(outs["one"] / ("ms.%d.fpkm" % i))), O.read_fpkm(str(outs["two"] / ("ms.%d.fpkm" % i)))
        np.testing.assert_array_equal(a["fpkm"], b["fpkm"])
    # the hand-over happened: after the 31-nt sample every link is below the cut (set id -1 in column 2 of .segments) ...
    seg1 = open(outs["two"] / "ms.1.segments").read().splitlines()[1:]
    multi = [l.split("\t") for l in seg1 if "," in l.split("\t")[2]]
    assert len(multi) == n - 1 and all(f[1] == "s-1" for f in multi)
    # ... whereas the same 30-nt sample on its own loses only the weak link
    solo = tmp_path / "solo"
    subprocess.run([CLI, "-q", "-g", "-I", rsh, str(solo), "s", a30], check=True, timeout=600)
    multi = [l.split("\t") for l in open(solo / "s.0.segments").read().splitlines()[1:] if "," in l.split("\t")[2]]
    assert sum(f[1] == "s-1" for f in multi) == 1


def test_nested_output_directory_and_unwritable_one(tmp_path):
    """The reference runs `mkdir -p outdir` (emsar_main.c:284-285); an outdir that cannot be created is reported before any work."""
    fx = get_fixture("toy5_se50")
    d = tmp_path / "a" / "b" / "c"
    subprocess.run([CLI, "-q"] + fx.meta["opts"] + ["-I", os.path.join(fx.dir, "index.rsh"), str(d), "out", _aln(fx)], check=True, timeout=300)
    assert os.path.exists(d / "out.0.fpkm")
    blocker = tmp_path / "file"
    blocker.write_text("x")
    r = subprocess.run([CLI, "-q", "-I", os.path.join(fx.dir, "index.rsh"), str(blocker / "sub"), "out", _aln(fx)], capture_output=True, timeout=300)
    assert r.returncode != 0 and b"can't create output directory" in r.stderr


@pytest.mark.parametrize("case", ["syn300_se", "syn300_k2", "toy5_pe_bam", "syn2k_se"])
def test_device_collapse_writes_the_same_files(case, tmp_path):
    """--device-collapse (SURVEY.md 8f N1 on the product path): the filters of alignment.c:29-95 / emsar_functions.c:372,849 stay on
    the host, the merge of update_ReadCounts (emsar_functions.c:838-943) runs in emsar_hip_collapse_rows.  The counts are the
    same integers either way, so every output file is byte-identical to the host-collapse run (the resident-set solver is
    bit-reproducible given equal inputs)."""
    fx = get_fixture(case)
    outs = []
    for tag, extra in (("host", []), ("dev", ["--device-collapse"])):
        d = tmp_path / tag
        cmd = [CLI, "-q", "-g"] + extra + fx.meta["opts"] + ["-I", os.path.join(fx.dir, "index.rsh"), str(d), "out", _aln(fx)]
        subprocess.run(cmd, check=True, timeout=300)
        outs.append(d)
    _check_fpkm_file(fx, str(outs[1] / "out.0.fpkm"))
    assert open(outs[1] / "out.0.fraglength_effect").read() == open(os.path.join(fx.dir, "ref.run0.fraglength_effect")).read()
    for ext in ("fraglength_effect", "segments"):
        assert open(outs[0] / ("out.0." + ext)).read() == open(outs[1] / ("out.0." + ext)).read()
    a, b = O.read_fpkm(str(outs[0] / "out.0.fpkm")), O.read_fpkm(str(outs[1] / "out.0.fpkm"))
    np.testing.assert_array_equal(a["fpkm"], b["fpkm"])


def test_config4_shape_at_one_hundredth(tmp_path):
    """BASELINE config 4 (-M: 8 single-end BAM samples of one index, one per GPU) at 1/100 of its size: 2000 transcripts in Zipf(1.6)
    families, 8 x 200 000 reads written as BAM by tests/perf/synth_bam.c (tests/perf/cfg4_m.py runs the same at full size).  The list
    run by four workers sharing the card (`--devices 0,0,0,0`) and by one worker must write the bytes the eight single-sample runs write
    (`emsar_main.c:380-488`: samples are independent but for EUMAcut, which no set of this index raises), with and without the device
    collapse; every sample conserves its reads; and the first sample agrees with the compiled reference where that is on the box."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "perf"))
    import cfg4_gen as G
    work = str(tmp_path)
    idx = G.make_index(work, 2000)
    bams = []
    for i in range(8):
        p = os.path.join(work, "s%d.bam" % i)
        G.make_sample(idx, 200000, 40 + i, p, threads=4)
        bams.append(p)
    lst = os.path.join(work, "list.txt")
    open(lst, "w").write("\n".join(bams) + "\n")
    rsh = os.path.join(work, "index.rsh")
    for tag, extra in (("w4", ["-M", "--devices", "0,0,0,0"]), ("w1", ["-M", "--devices", "0"]), ("w4c", ["-M", "--devices", "0,0,0,0", "--device-collapse"])):
        subprocess.run([CLI, "-q", "-B"] + extra + ["-I", rsh, os.path.join(work, tag), "o", lst], check=True, timeout=600)
    for i, b in enumerate(bams):
        subprocess.run([CLI, "-q", "-B", "-I", rsh, os.path.join(work, "single%d" % i), "o", b], check=True, timeout=300)
        want = open(os.path.join(work, "single%d" % i, "o.0.fpkm"), "rb").read()
        for tag in ("w4", "w1", "w4c"):
            assert open(os.path.join(work, tag, "o.%d.fpkm" % i), "rb").read() == want, (tag, i)
        got = O.read_fpkm(os.path.join(work, "w4", "o.%d.fpkm" % i))
        assert abs(got["ireadcount"].sum() - 200000) <= 1e-5 * 200000 + 1e-2 and abs(got["tpm"].sum() - 1e6) < 1.0
    ref = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "emsar")
    if os.path.exists(ref):
        runs = []
        for k in range(3):                                              # the mask needs k >= 3 reference runs
            subprocess.run([ref, "-q", "-p", "4", "-B", "-I", rsh, os.path.join(work, "ref%d" % k), "o", bams[0]], check=True, timeout=600,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            runs.append(O.read_fpkm(os.path.join(work, "ref%d" % k, "o.0.fpkm"))["fpkm"])
        a = np.array(runs)
        mask = (a.max(0) - a.min(0)) > 1e-6 * np.abs(a).max(0) + 1.5e-6
        got = O.read_fpkm(os.path.join(work, "w4", "o.0.fpkm"))["fpkm"]
        bad = (np.abs(got - a[0]) > 1e-5 * np.abs(a[0]) + 1.5e-6) & ~mask
        assert bad.sum() <= 2, (int(bad.sum()), np.nonzero(bad)[0][:5], got[bad][:5], a[0][bad][:5])
        assert open(os.path.join(work, "w4", "o.0.fraglength_effect")).read() == open(os.path.join(work, "ref0", "o.0.fraglength_effect")).read()


@pytest.mark.parametrize("case", ["syn2k_se", "vicugna_pe", "toy5_pe_bam"])
def test_library_numbering_does_not_show_in_the_files(case, tmp_path):
    """The TILED layout may number the transcripts by co-occurrence (csrc/renumber.hpp); ids at the ABI stay the caller's, so the files
    emsar-hip writes are the same bytes with the numbering forced on, off, or left to the data."""
    fx = get_fixture(case)
    out = {}
    for mode in ("0", "2", "1"):
        d = tmp_path / ("m" + mode)
        cmd = [CLI, "-q", "-g"] + fx.meta["opts"] + ["-I", os.path.join(fx.dir, "index.rsh"), str(d), "out", _aln(fx)]
        subprocess.run(cmd, check=True, timeout=300, env=dict(os.environ, EMSAR_HIP_RENUMBER=mode))
        out[mode] = {ext: open(d / ("out.0." + ext), "rb").read() for ext in ("fpkm", "segments", "fraglength_effect")}
    assert out["0"] == out["2"] == out["1"]
    _check_fpkm_file(fx, str(tmp_path / "m2" / "out.0.fpkm"))
