# sort block swept for the 4-slot-block build (EMSAR_HIP_LIB) on both laws of config 3, config 5 x 0.25 and config 2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-sweep_blk4}; mkdir -p $O; cd $R
python - <<'PY' > $O/sweep.txt 2>&1
import os, sys, subprocess
sys.path.insert(0, os.getcwd())
code = r'''
import os, sys
sys.path.insert(0, os.getcwd())
from emsar_amd import EmsarHip, synth
cfg, scale, st = sys.argv[1], float(sys.argv[2]), sys.argv[3]
s = synth.make_config(cfg, scale, st)
dev = EmsarHip(0)
for b in sys.argv[4:]:
    os.environ["EMSAR_HIP_TILE_BLOCK"] = b
    dev.upload_structure(s["n_tx"], s["row_ptr"], s["col_idx"])
    dev.upload_sample(None, None, s["den"])
    i = dev.info()
    dev.run_passes(200)
    ms = min(dev.run_passes(200) / 200 for _ in range(3))
    print(os.path.basename(os.environ.get("EMSAR_HIP_LIB", "default")), cfg, scale, st, "block", b, "ms/pass %.4f" % ms, "units", i["n_units"], "far", i["far_entries"], "stored MB %.1f" % (i["stored_bytes_per_pass"] / 1e6), flush=True)
'''
root = os.getcwd()
for lib, cfg, scale, st, blocks in (("blk4", "cfg3", "1.0", "family", ["24", "32", "48", "64", "96"]), ("blk4", "cfg3", "1.0", "window", ["24", "32", "48", "64", "96"]),
                                    ("blk4", "cfg5", "0.25", "window", ["32", "48", "64", "96"]), ("", "cfg5", "0.25", "window", ["48"]),
                                    ("blk4", "cfg2", "1.0", "window", ["32", "48", "64"]), ("", "cfg2", "1.0", "window", ["48"])):
    env = dict(os.environ)
    if lib:
        env["EMSAR_HIP_LIB"] = os.path.join(root, "emsar_amd", "libemsar_hip_blk4.so")
    r = subprocess.run([sys.executable, "-c", code, cfg, scale, st] + blocks, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    print(r.stdout, flush=True)
PY
cat $O/sweep.txt
