#!/usr/bin/env python3
"""bench.py -- EM iterations/s of the HIP abundance core on BASELINE.json's headline workload.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg3] [--scale 1.0]

A "step" is one EM pass (E-step + M-step) over one synthetic read->transcript compatibility matrix that is
already resident in HBM.  N=1: BASELINE config 3 (50M reads x 200k transcripts, mean 5 alignments, -k 100).
N>1 (torchrun, one rank per GPU): the -M multi-sample path -- every rank solves its OWN independent sample
of the same shape (different seed), no data-path collective; weak scaling.  The only torch.distributed use
is the barrier and the max-over-ranks of the timed region.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel of the TILED layout, k_pass_tiled_unit
(+ the small k_update that shares the pass): algorithmic bytes per pass /
mean device time per pass measured with HIP events on the library's own stream (`frac_actual`: the same with the HBM bytes the PMC
counters saw, profiles/traffic.json).  After the timed passes: `solve_to_convergence` (the same matrix solved to --solve) and
`fpkm_delta_vs_oracle` (a down-scaled config 3 solved by both).  `cpu_baseline` times the CPU oracle's OpenMP EM pass on the same
matrix on this box's host cores (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def _cfg(args, rank, structure):
    from emsar_amd import synth
    cfg = dict(synth.CONFIGS[args.config])
    cfg["seed"] = cfg["seed"] + 100 * rank
    if args.xfam is not None:
        cfg["xfam"] = args.xfam
    if args.scale != 1.0:
        cfg["n_reads"] = max(1000, int(cfg["n_reads"] * args.scale))
        cfg["n_tx"] = max(500, int(cfg["n_tx"] * args.scale))
    cfg["structure"] = structure
    return cfg


def _traffic(args, structure, layout_name):
    """HBM bytes per launch of the pass kernel from the PMC counters.  They cannot be read inside this process, so the value is the
    committed rocprofv3 measurement of exactly this kernel + workload (profiles/traffic.json, tools/prof_round.sh), else None."""
    try:
        tr = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except (OSError, ValueError):
        return None
    key = "%s/%s/%s" % (args.config, structure, layout_name)
    if args.scale != 1.0 or args.xfam is not None or key not in tr:
        return None
    return tr[key]


def _kernel_name(info, weighted):
    multi = int(os.environ.get("EMSAR_HIP_TILED_MULTI", "1"))
    if info["layout"] == 1:
        return "k_pass_csr"
    units = (multi == 5) or (multi == 1 and info["n_chunks"] > 2048)
    if units and weighted:                # the plain EM pass of weighted rows runs the unit kernel too (likelihood passes: k_pass_tiled)
        return "k_pass_tiled" if os.environ.get("EMSAR_HIP_WEIGHTED_UNIT", "1") == "0" else "k_pass_tiled_unit<weighted>"
    if units:
        return "k_pass_tiled_unit"
    if not weighted and multi in (2, 3, 4):
        return "k_pass_tiled_multi<%d>" % multi
    return "k_pass_tiled"


def _roofline(args, structure, info, per_pass_s, weighted=False, tag="", live=None, live_why=None):
    """frac = HBM bytes the counters saw per launch / device time per pass / 8 TB/s -- a fraction of the peak the memory system really
    delivered.  The SURVEY 8d formula (what a CSR walk would stream) is reported as csr_equivalent_GBps: the TILED layout stores and
    moves a third of those bytes, so that figure may exceed the HBM peak -- it is not a roofline fraction."""
    layout_name = {1: "csr", 3: "tiled", 259: "tiled+merged-rows"}[info["layout"]] + tag
    tr = _traffic(args, structure, layout_name)
    r = {"bound": "latency", "bound_evidence": "PMC: HBM below its copy ceiling, waves waiting (SQ_WAIT_ANY) 41-49 % of their cycles, LDS pipe < 50 % busy, VALU ~20 % (profiles/*_pmc.txt)",
         "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
         "kernel": _kernel_name(info, weighted) + "+k_update", "device_ms_per_pass": per_pass_s * 1e3,
         "algorithmic_bytes_per_pass": info["bytes_per_pass"], "stored_bytes_per_pass": info["stored_bytes_per_pass"],
         "csr_equivalent_GBps": info["bytes_per_pass"] / per_pass_s / 1e9,
         "stored_GBps": info["stored_bytes_per_pass"] / per_pass_s / 1e9}
    if live is not None:                       # observed by this very run (live_traffic): two rocprofv3 --pmc child processes on this box
        r["traffic"] = live["hbm_bytes_per_launch"]
        r["traffic_source"] = ("observed in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate child processes, 3 passes of "
                               "this matrix each, %d dispatches of the pass kernel): FETCH_SIZE %.0f KB x 2 (gfx950 correction) + WRITE_SIZE %.0f KB; %s s"
                               % (live["FETCH_SIZE_dispatches"], live["FETCH_SIZE"], live["WRITE_SIZE"], live.get("seconds")))
        if tr is not None:
            r["traffic_committed"] = tr["hbm_bytes_per_launch"]      # profiles/traffic.json, for comparison
        r["achieved"] = r["traffic"] / per_pass_s / 1e9
        r["frac"] = r["achieved"] / HBM_PEAK_GBS
    elif tr is not None:
        if live_why and live_why != "not requested":
            r["live_pmc_failed"] = live_why
        r["traffic"] = tr["hbm_bytes_per_launch"]
        r["traffic_source"] = "profiles/traffic.json (rocprofv3 PMC run of this kernel and workload: %s)" % tr.get("profile", "")
        r["achieved"] = tr["hbm_bytes_per_launch"] / per_pass_s / 1e9
        r["frac"] = r["achieved"] / HBM_PEAK_GBS
    else:
        r["note"] = "no PMC measurement committed for this workload: frac is null (stored_GBps / peak = %.3f is a lower bound of it)" % (
            r["stored_GBps"] / HBM_PEAK_GBS)
    return r


def _layout_stats(info):
    d = {k: info[k] for k in ("n_chunks", "n_units", "n_slices", "padded_entries", "far_entries", "window", "tiled_entries", "tiled_ids", "renumbered")}
    d["tids_per_entry"] = info["tiled_ids"] / info["tiled_entries"] if info["tiled_entries"] else None
    return d


def _pmc_child(cache):
    """Internal (bench.py --pmc-child FILE): the profiled process of live_traffic() -- the matrix of the parent from its cache file, a few
    passes of the same kernel, nothing else (no torch, no JSON)."""
    import numpy as np
    from emsar_amd import EmsarHip
    z = np.load(cache)
    with EmsarHip(int(z["device"])) as dev:
        if int(z["weighted"]):
            dev.upload_structure(int(z["n_tx"]), z["row_ptr"], z["col_idx"], int(z["layout"]))
            dev.upload_sample(z["wgt"], None, z["den"])
        else:
            dev.upload_structure(int(z["n_tx"]), z["row_ptr"], z["col_idx"], int(z["layout"]), merge_rows=bool(z["merge_rows"]))
            dev.upload_sample(None, None, z["den"])
        dev.run_passes(1)
        dev.run_passes(3)


LIVE = {"on": False, "device": 0}        # set by main(): rank 0 of a one-GPU run that is not itself being profiled


def _live_for(n_tx, row_ptr, col_idx, den, wgt=None):
    """live_traffic() for one of the other forms of the workload, when the run observes traffic at all"""
    if not LIVE["on"] or len(col_idx) > 1.5e9:
        return None, "not requested"
    t0 = time.time()
    got, why = live_traffic(LIVE["device"], n_tx, row_ptr, col_idx, den, LIVE["layout"], wgt=wgt)
    if got is not None:
        got["seconds"] = round(time.time() - t0, 1)
    return got, why


def live_traffic(device, n_tx, row_ptr, col_idx, den, layout, wgt=None, merge_rows=False, kernel_pattern="k_pass_tiled", timeout_s=240):
    """HBM bytes per launch of the pass kernel, OBSERVED in this run: two child processes under `rocprofv3 --kernel-trace --pmc <counter>`
    (FETCH_SIZE; WRITE_SIZE -- separate passes, as /opt/skills/guides/MI355X_MICROARCH.md's HBM section prescribes) run three passes of
    the same matrix; bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (KB counters; FETCH_SIZE doubled: the guide's gfx950 correction for
    wide streaming reads), mean over the dispatches of the kernel.  None (and the reason) when the profiler is not there or fails:
    the caller then falls back to the committed profiles/traffic.json."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    import numpy as np
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None, "rocprofv3 not found"
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else None
    work = tempfile.mkdtemp(prefix="emsar_pmc_", dir="/tmp")
    cache = os.path.join(tempfile.mkdtemp(prefix="emsar_pmc_", dir=shm), "m.npz") if shm else os.path.join(work, "m.npz")
    got = {}
    try:
        np.savez(cache, device=device, n_tx=n_tx, row_ptr=row_ptr, col_idx=col_idx, den=den, layout=layout, weighted=int(wgt is not None),
                 wgt=wgt if wgt is not None else np.zeros(1, dtype=np.int32), merge_rows=int(merge_rows))
        env = dict(os.environ, TMPDIR="/tmp")
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(work, counter)
            cmd = [prof, "--kernel-trace", "--pmc", counter, "-d", out, "-o", "p", "--output-format", "csv", "--",
                   sys.executable, os.path.abspath(__file__), "--pmc-child", cache]
            # a session of its own: on a timeout the profiler AND the profiled process are ended (the exact group this call started)
            pr = subprocess.Popen(cmd, cwd=work, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, start_new_session=True)
            try:
                log, _ = pr.communicate(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                import signal
                try:
                    os.killpg(pr.pid, signal.SIGKILL)
                except OSError:
                    pass
                pr.communicate()
                return None, "rocprofv3 --pmc %s did not finish in %d s" % (counter, timeout_s)
            if pr.returncode != 0:
                return None, "rocprofv3 --pmc %s exited %d: %s" % (counter, pr.returncode, log[-300:].replace("\n", " | "))
            per = {}
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if kernel_pattern in row["Kernel_Name"] and row["Counter_Name"] == counter:
                        per[row["Dispatch_Id"]] = per.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
            if not per:
                return None, "no %s rows for %s in the profiler's output" % (counter, kernel_pattern)
            got[counter] = sum(per.values()) / len(per)
            got[counter + "_dispatches"] = len(per)
    except Exception as e:                                  # the roofline then quotes the committed measurement
        return None, "%s: %s" % (type(e).__name__, e)
    finally:
        shutil.rmtree(work, ignore_errors=True)
        if shm:
            shutil.rmtree(os.path.dirname(cache), ignore_errors=True)
    got["hbm_bytes_per_launch"] = int((2.0 * got["FETCH_SIZE"] + got["WRITE_SIZE"]) * 1024)
    return got, None


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--pmc-child":
        return _pmc_child(sys.argv[2])
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="cfg3")
    ap.add_argument("--structure", default="family", choices=["window", "family", "family_shuffled"],
                    help="row law of the synthetic matrix (emsar_amd/synth.py): SURVEY 8d's family subsets (default), the same with "
                         "shuffled transcript ids, or consecutive-tid windows (the generator of rounds 1-2)")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink reads and transcripts (debug only)")
    ap.add_argument("--layout", default="auto", choices=["auto", "csr", "tiled"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip workload_variants and the collapsed form (profiling runs)")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="do not run the two rocprofv3 --pmc child processes that observe the pass kernel's HBM traffic in this run; "
                         "roofline.traffic then comes from profiles/traffic.json (N > 1 and profiling runs always do)")
    ap.add_argument("--xfam", type=float, default=None, help="experiment: override the config's share of cross-family reads")
    ap.add_argument("--merge-rows", action="store_true",
                    help="store identical rows once (read -> segment collapse at upload); NOT the headline configuration")
    ap.add_argument("--collapsed", action="store_true",
                    help="time the COLLAPSED form as the main workload: rows = segments with read counts (device collapse first)")
    ap.add_argument("--solve", type=float, default=1e-6, metavar="TOL",
                    help="after the timed passes, run the full solver (SQUAREM) on the same matrix to this tolerance and report passes / time "
                         "(BASELINE metric: 'to convergence'; config 5 names 1e-6); 0 = skip")
    ap.add_argument("--spinup", type=float, default=1.0, metavar="SECONDS",
                    help="un-timed passes before the warm-up so that the clocks are up when the K timed steps start (a fresh box idles)")
    ap.add_argument("--solve-floor", type=float, default=1e-2,
                    help="abs_floor of the stopping rule max|dtheta|/(theta+floor) in FPKM; boundary components decay like 1/k, "
                         "so a floor at the print quantum (1e-6) is only reachable on small problems")
    args = ap.parse_args()

    import numpy as np
    import torch
    from emsar_amd import EmsarHip, synth
    from emsar_amd import dist as D
    from emsar_amd.hip import LAYOUT_AUTO, LAYOUT_CSR

    rank, world, local_rank = D.env_rank()
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # rehearsal knobs (one-GPU box): EMSAR_BENCH_BACKEND=gloo EMSAR_BENCH_DEVICE=0 lets several ranks share a card
    backend = os.environ.get("EMSAR_BENCH_BACKEND", "nccl")
    if "EMSAR_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["EMSAR_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    # RCCL (backend "nccl"); used for the barrier and the max-over-ranks of the timed region only
    group = D.Group(backend, torch.device("cuda", local_rank) if backend == "nccl" else None)

    # ---- workload: one independent sample per rank ------------------------------------------------------
    t0 = time.time()
    s = synth.make_matrix(**_cfg(args, rank, args.structure))
    t_gen = time.time() - t0
    nnz = int(len(s["col_idx"]))
    layout = {"auto": LAYOUT_AUTO, "csr": LAYOUT_CSR, "tiled": 3}[args.layout]
    dev = EmsarHip(local_rank)
    t0 = time.time()
    collapsed_main = None
    if args.collapsed:
        dev.collapse_rows(s["n_tx"], s["row_ptr"][:1001], s["col_idx"][:int(s["row_ptr"][1000])], want_map=False)     # loads the kernels' code objects
        rp_c, ci_c, w_c, _, cst = dev.collapse_rows(s["n_tx"], s["row_ptr"], s["col_idx"], want_map=False)
        collapsed_main = {"rows": int(len(w_c)), "nnz": int(len(ci_c)), "collapse_kernel_ms": cst.kernel_ms}
        dev.upload_structure(s["n_tx"], rp_c, ci_c, layout)
        dev.upload_sample(w_c, None, s["den"])
    else:
        dev.upload_structure(s["n_tx"], s["row_ptr"], s["col_idx"], layout, merge_rows=args.merge_rows)
        dev.upload_sample(None, None, s["den"])
    t_up = time.time() - t0
    info = dev.info()

    def barrier():
        torch.cuda.synchronize()
        group.barrier()
        torch.cuda.synchronize()

    # ---- spin-up (clocks), warmup, then exactly K timed steps --------------------------------------------------
    t_spin = time.perf_counter()
    while args.spinup > 0 and time.perf_counter() - t_spin < args.spinup:
        dev.run_passes(100)
    dev.reset_theta()
    if args.warmup > 0:
        dev.run_passes(args.warmup)
    barrier()
    t0 = time.perf_counter()
    kernel_ms = dev.run_passes(args.steps)   # K passes back to back on the library stream; returns after its sync
    barrier()
    wall = time.perf_counter() - t0
    wall, kernel_ms = group.max([wall, kernel_ms])

    # ---- sanity of what was timed: mass conservation after the last pass --------------------------------
    th = dev.get_theta()
    mass = float((th * s["den"]).sum())
    ok = bool(np.isfinite(th).all() and abs(mass - s["n_reads"]) <= 1e-8 * s["n_reads"])

    solve = None
    if args.solve > 0:
        # the whole solve of the benchmarked matrix: SQUAREM-accelerated EM from the uniform start until the plain EM step moves no
        # component by more than tol relative to (theta + abs_floor); streaming passes (a read-level matrix is one connected set)
        t0 = time.perf_counter()
        th_s, st = dev.solve(max_iter=200000, accel=1, tol=args.solve, abs_floor=args.solve_floor, check_every=4, set_mode=1)
        dt = time.perf_counter() - t0
        solve = {"tol": args.solve, "abs_floor": args.solve_floor, "passes": st.iters, "converged": bool(st.converged), "seconds": dt,
                 "kernel_ms": st.kernel_ms, "iters_per_s_to_convergence": st.iters / dt, "loglik": st.loglik, "final_delta": st.final_delta,
                 "mass_conserved": bool(abs(float((th_s * s["den"]).sum()) - s["n_reads"]) <= 1e-8 * s["n_reads"])}
    live, live_why = None, "not requested"
    under_profiler = any(k.startswith("ROCPROF") for k in os.environ)          # this process is itself being profiled (tools/prof_round.sh)
    if nnz > 1.5e9 and not args.no_live_pmc:
        live_why = "skipped above 1.5e9 nonzeros (the matrix would be written out and uploaded twice more)"
    elif rank == 0 and world == 1 and not args.no_live_pmc and info["layout"] != 1 and not under_profiler:
        t0 = time.time()
        if args.collapsed:
            live, live_why = live_traffic(local_rank, s["n_tx"], rp_c, ci_c, s["den"], layout, wgt=w_c)
        else:
            live, live_why = live_traffic(local_rank, s["n_tx"], s["row_ptr"], s["col_idx"], s["den"], layout, merge_rows=args.merge_rows)
        if live is not None:
            live["seconds"] = round(time.time() - t0, 1)
        LIVE.update(on=live is not None, device=local_rank, layout=layout)       # the other forms below observe theirs too, if this one could
    out = None
    if rank == 0:
        per_pass_s = kernel_ms / 1e3 / args.steps
        form = "collapsed (segments x read counts)" if args.collapsed else "read-level CSR"
        out = {
            "metric": "EM iterations/s", "value": world * args.steps / wall, "unit": "iter/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall * 1e3 / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: %d reads x %d transcripts, nnz %d (mean %.2f aln/read), row law '%s' (emsar_amd/synth.py), %s, one sample per GPU"
                       % (args.config, s["n_reads"], s["n_tx"], nnz, nnz / s["n_reads"], args.structure, form),
                       "structure": args.structure,
                       "layout": {1: "csr", 3: "tiled", 259: "tiled+merged-rows"}[info["layout"]], "parallelism": "1 sample per GPU x %d" % world},
            "read_alignments_per_s": world * nnz * args.steps / wall,
            "roofline": _roofline(args, args.structure, info, per_pass_s, weighted=args.collapsed, tag="+collapsed" if args.collapsed else "",
                                  live=live, live_why=live_why),
            "layout_stats": _layout_stats(info),
            "setup_s": {"generate": round(t_gen, 2), "upload_and_layout": round(t_up, 2)},
            "mass_conserved": ok,
        }
        if collapsed_main:
            out["config"]["collapsed"] = collapsed_main
        if solve is not None:
            out["solve_to_convergence"] = solve
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(s, nnz)
    # ---- the other forms of the same config (rank 0, N = 1): the other row laws, and the collapsed (segment-level) form -----------
    if rank == 0 and world == 1 and not args.no_variants and not args.collapsed and not args.merge_rows:
        variants = {args.structure: {"ms_per_pass": kernel_ms / args.steps, "nnz": nnz, **_variant_stats(info),
                                     "roofline_frac": out["roofline"]["frac"], "hbm_traffic_bytes": out["roofline"]["traffic"],
                                     "traffic_observed_in_this_run": live is not None}}
        out["collapsed_form"] = collapsed_form(args, dev, s, info)
        for st_name in ("window", "family", "family_shuffled"):
            if st_name in variants:
                continue
            if st_name == "family_shuffled" and args.structure == "family":
                v = shuffled_copy(s)                      # the same matrix, transcripts numbered at random
            else:
                v = synth.make_matrix(**_cfg(args, rank, st_name))
            variants[st_name] = time_variant(args, dev, v, st_name)
            del v
        out["workload_variants"] = variants
    dev.close()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["fpkm_delta_vs_oracle"] = fpkm_delta_vs_oracle(local_rank, args.config, args.structure)
        out["time_to_mle"] = time_to_mle(local_rank)
    group.close()
    if rank == 0:
        print(json.dumps(out))


def _variant_stats(info):
    return {"tids_per_entry": info["tiled_ids"] / info["tiled_entries"] if info["tiled_entries"] else None,
            "renumbered": info["renumbered"], "units": info["n_units"], "far_entries": info["far_entries"],
            "stored_bytes_per_pass": info["stored_bytes_per_pass"]}


def shuffled_copy(s):
    import numpy as np
    new_of_old = np.random.default_rng(12345).permutation(s["n_tx"]).astype(np.int32)
    den = np.empty_like(s["den"])
    den[new_of_old] = s["den"]
    return {"n_tx": s["n_tx"], "n_reads": s["n_reads"], "row_ptr": s["row_ptr"], "col_idx": new_of_old[s["col_idx"]], "den": den}


def _spin(dev, seconds):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        dev.run_passes(100)


def time_variant(args, dev, v, name, passes=100):
    """ms per pass of another row law of the same config (same kernel, same library defaults), 20 warm-up + `passes` timed passes."""
    t0 = time.time()
    dev.upload_structure(v["n_tx"], v["row_ptr"], v["col_idx"])
    dev.upload_sample(None, None, v["den"])
    t_up = time.time() - t0
    info = dev.info()
    _spin(dev, 0.5)                       # the card idled while the host built the layout: clocks up again before timing
    ms = dev.run_passes(passes) / passes
    live, why = _live_for(v["n_tx"], v["row_ptr"], v["col_idx"], v["den"]) if info["layout"] != 1 else (None, "not requested")
    r = _roofline(args, name, info, ms / 1e3, live=live, live_why=why)
    return {"ms_per_pass": ms, "nnz": int(len(v["col_idx"])), **_variant_stats(info), "upload_and_layout_s": round(t_up, 2),
            "roofline_frac": r["frac"], "hbm_traffic_bytes": r["traffic"], "traffic_observed_in_this_run": live is not None}


def collapsed_form(args, dev, s, info_read_level, passes=100):
    """'also build the collapsed form (unique tid-sets + counts) and report both -- the collapsed form is what the reference
    actually solves' (update_ReadCounts, emsar_functions.c:838-943; main.c:404).  The read-level matrix is collapsed on the device
    (emsar_hip_collapse_rows), the segments and their read counts are uploaded as a weighted matrix and timed like the main workload."""
    # (a first call on 1000 rows loads the collapse kernels' code objects, so that the call whose kernel time is reported does not)
    dev.collapse_rows(s["n_tx"], s["row_ptr"][:1001], s["col_idx"][:int(s["row_ptr"][1000])], want_map=False)
    rp, ci, w, _, cst = dev.collapse_rows(s["n_tx"], s["row_ptr"], s["col_idx"], want_map=False)
    t0 = time.time()
    dev.upload_structure(s["n_tx"], rp, ci)
    dev.upload_sample(w, None, s["den"])
    t_up = time.time() - t0
    info = dev.info()
    _spin(dev, 0.5)
    dev.reset_theta()
    dev.run_passes(20)
    ms = dev.run_passes(passes) / passes
    th = dev.get_theta()
    mass = float((th * s["den"]).sum())
    live, why = _live_for(s["n_tx"], rp, ci, s["den"], wgt=w) if info["layout"] != 1 else (None, "not requested")
    r = _roofline(args, args.structure, info, ms / 1e3, weighted=True, tag="+collapsed", live=live, live_why=why)
    return {"rows": int(len(w)), "nnz": int(len(ci)), "reads": int(w.sum()), "ms_per_pass": ms, "iters_per_s": 1e3 / ms,
            "collapse_kernel_ms": cst.kernel_ms, "upload_and_layout_s": round(t_up, 2), "kernel": r["kernel"],
            "roofline": {k: r[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_pass",
                                           "stored_bytes_per_pass", "stored_GBps", "csr_equivalent_GBps", "traffic_source") if k in r},
            **_variant_stats(info), "mass_conserved": bool(abs(mass - s["n_reads"]) <= 1e-8 * s["n_reads"])}


def cpu_baseline(s, nnz):
    """The CPU oracle's EM pass (oracle/em_oracle.c, OpenMP) on the same matrix, bounded to ~10-30 s."""
    import numpy as np
    import oracle as O
    # a one-GPU box owns a 16-core share of its host (the pool's rule for worker counts); never more threads
    # than that, whatever os.cpu_count() says about the whole machine
    cores = min(os.cpu_count() or 1, len(os.sched_getaffinity(0)), int(os.environ.get("EMSAR_CPU_THREADS", "16")))
    m = O.Csr(s["n_tx"], s["row_ptr"], s["col_idx"])
    th = np.ones(s["n_tx"])
    th, _ = m.em_step(th, s["den"], n_threads=cores)     # untimed warm pass
    n, t0 = 0, time.perf_counter()
    while True:
        th, _ = m.em_step(th, s["den"], n_threads=cores)
        n += 1
        dt = time.perf_counter() - t0
        if dt > 12.0 or n >= 2000:
            break
    return {"value": n / dt, "unit": "iter/s", "cores": cores, "kind": "port",
            "sample": "%d EM passes of the oracle's OpenMP EM over the same %d-read matrix (nnz %d)" % (n, s["n_reads"], nnz)}


def fpkm_delta_vs_oracle(device, config, structure="family"):
    """BASELINE metric, last clause ('FPKM delta vs ref'): the benchmarked config, down-scaled until the CPU oracle solves it in
    seconds, solved by the HIP path and by the oracle's EM to the same tolerance; relative FPKM differences."""
    import numpy as np
    import oracle as O
    from emsar_amd import EmsarHip, synth
    scale = {"cfg2": 0.02, "cfg3": 0.004, "cfg4": 0.01, "cfg5": 0.0005}.get(config, 0.004)
    s = synth.make_config(config, scale, structure)
    m = O.Csr(s["n_tx"], s["row_ptr"], s["col_idx"])
    cores = min(os.cpu_count() or 1, len(os.sched_getaffinity(0)), int(os.environ.get("EMSAR_CPU_THREADS", "16")))
    th_o, st_o = m.em_solve(max_iter=200000, accel=1, tol=1e-9, n_threads=cores)
    with EmsarHip(device) as dev:
        dev.upload_structure(s["n_tx"], s["row_ptr"], s["col_idx"])
        dev.upload_sample(None, None, None)          # E = 1 per row, den computed on the device: the oracle's own model of this matrix
        th, st = dev.solve(max_iter=200000, accel=1, tol=1e-9, set_mode=1)
    big = th_o > 1e-3
    rel = np.abs(th - th_o)[big] / th_o[big]
    return {"workload": "%s x %g (%s): %d reads x %d transcripts" % (config, scale, structure, s["n_reads"], s["n_tx"]), "tol": 1e-9,
            "max_rel_delta_fpkm_above_1e-3": float(rel.max()) if rel.size else 0.0, "max_abs_delta": float(np.abs(th - th_o).max()),
            "within_1e-5_rel_plus_1.5e-6": bool(np.all(np.abs(th - th_o) <= 1e-5 * np.abs(th_o) + 1.5e-6)),
            "loglik_gpu_minus_oracle": float(m.loglik(th) - m.loglik(th_o)), "gpu_passes": st.iters, "oracle_passes": st_o.iters}


def time_to_mle(device):
    """SURVEY.md 8d, headline reported the second way: wall time to the MLE of a SEGMENT-level problem (what the
    reference actually solves: families of transcripts, thousands of independent sets) -- emsar_hip_solve against the
    oracle's port of the reference's own algorithm (pattern search per set, static split over threads:
    MLE / run_MLE_threads, emsar_functions.c:2977-3126).  Not part of `value`."""
    import numpy as np
    import oracle as O
    from emsar_amd import EmsarHip, synth
    rng = np.random.default_rng(11)
    sizes = np.minimum(rng.zipf(1.6, size=40000), 60)
    sizes = sizes[np.cumsum(sizes) <= 100000]
    n_tx, rp, ci, _ = synth.family_matrix([int(x) for x in sizes], rows_per_tid=3, seed=11, dup=0.0)
    E = rng.uniform(0.5, 2.0, size=len(rp) - 1)
    # counts drawn from the model itself: theta* ~ LogNormal(0, 2) with 30 % zeros, R_c ~ Poisson(E_c * sum theta*)
    theta_true = np.where(rng.random(n_tx) < 0.3, 0.0, rng.lognormal(0.0, 2.0, size=n_tx))
    R = rng.poisson(E * np.add.reduceat(theta_true[ci], rp[:-1].astype(np.int64))).astype(np.int32)
    cores = min(os.cpu_count() or 1, len(os.sched_getaffinity(0)), int(os.environ.get("EMSAR_CPU_THREADS", "16")))
    m = O.Csr(n_tx, rp, ci, R=R, E=E, L=E)
    n_sets, cs, ts, _ = m.components()
    t0 = time.perf_counter()
    th_ref, sweeps = m.mle_pattern_search(cs, n_sets, seed=1, n_threads=cores)
    cpu_s = time.perf_counter() - t0
    dev = EmsarHip(device)
    dev.upload_structure(n_tx, rp, ci)
    dev.upload_sample(R, E, None)
    dev.solve(max_iter=200000, tol=1e-10)                       # first call: finds and packs the sets (host), warms up
    t0 = time.perf_counter()
    th, st = dev.solve(max_iter=200000, tol=1e-10)
    gpu_s = time.perf_counter() - t0
    # the same solve with the stopping rule emsar-hip uses by default: components below a quarter of the .fpkm print
    # quantum that are still falling, or that move by less than 1e-13 FPKM per pass, do not hold the solve up
    t0 = time.perf_counter()
    th_q, st_q = dev.solve(max_iter=200000, tol=1e-10, zero_cut=2.5e-7, abs_step=1e-13)
    gpu_q_s = time.perf_counter() - t0
    # time to EQUAL likelihood: the pattern search stops short of the optimum on large families (loglik_gpu_minus_cpu > 0), so 'time to
    # the MLE' compares two different end points; this is the time emsar_hip_solve needs to reach the likelihood the CPU run ended at
    F_ref = m.loglik(th_ref)
    equal_F = None
    for tol_e in (1e-1, 1e-2, 1e-3, 1e-4, 1e-5, 1e-6, 1e-8, 1e-10):
        t0 = time.perf_counter()
        th_e, st_e = dev.solve(max_iter=200000, tol=tol_e)
        dt_e = time.perf_counter() - t0
        if m.loglik(th_e) >= F_ref:
            equal_F = {"tol": tol_e, "gpu_s": dt_e, "gpu_em_passes_slowest_set": st_e.set_passes_max, "speedup": cpu_s / dt_e,
                       "loglik_gpu_minus_cpu": m.loglik(th_e) - F_ref}
            break
    dev.close()
    F = m.loglik(th)
    return {"workload": "segment-level synthetic: %d transcripts in %d families (Zipf 1.6, <= 60), %d segments, %d reads, %d connected sets"
                        % (n_tx, len(sizes), len(R), int(R.sum()), n_sets),
            "gpu_s": gpu_s, "gpu_kernel_ms": st.kernel_ms, "gpu_em_passes_slowest_set": st.set_passes_max, "gpu_sets_resident": st.sets_resident,
            "gpu_sets_streamed": st.sets_streamed, "gpu_converged": bool(st.converged), "set_packing_host_ms": st.sets_build_ms,
            "cpu_reference_algorithm_s": cpu_s, "cpu_cores": cores, "cpu_kind": "port", "cpu_sweeps": int(sweeps),
            "speedup": cpu_s / gpu_s, "loglik_gpu_minus_cpu": F - F_ref, "loglik": F, "time_to_equal_loglik": equal_F,
            "note": "cpu_kind port = the oracle's restatement of the reference's pattern search; the same-box whole-program comparison against the "
                    "compiled reference is profiles/r03_ref_vs_hip_sets_100k.txt (tests/perf/ref_vs_hip.py)",
            "print_quantum_stop": {"zero_cut": 2.5e-7, "abs_step": 1e-13, "gpu_s": gpu_q_s, "gpu_em_passes_slowest_set": st_q.set_passes_max,
                                   "max_abs_dtheta_vs_strict": float(np.abs(th_q - th).max()),
                                   "printed_differently": int((np.round(th_q, 6) != np.round(th, 6)).sum()), "speedup": cpu_s / gpu_q_s}}


if __name__ == "__main__":
    main()
