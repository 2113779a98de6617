# layout knobs swept on the family law of config 3 (they were tuned on the window law in rounds 1-2); one line per setting
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-sweep}; mkdir -p $O; cd $R
python - <<'PY' > $O/sweep.txt 2>&1
import os, sys, time
sys.path.insert(0, os.getcwd())
from emsar_amd import EmsarHip, synth
s = synth.make_config("cfg3", 1.0, "family")
dev = EmsarHip(0)
def run(env):
    for k in ("EMSAR_HIP_TAIL_SPLIT", "EMSAR_HIP_UNIT_FAR_SOFT", "EMSAR_HIP_UNIT_TILES_MAX", "EMSAR_HIP_TILE_BLOCK", "EMSAR_HIP_UNIT_TILES", "EMSAR_HIP_TILE_ROWS", "EMSAR_HIP_RENUMBER", "EMSAR_HIP_SHORT_ECNT", "EMSAR_HIP_SHORT_BLOCK"):
        os.environ.pop(k, None)
    os.environ.update(env)
    os.environ.setdefault("EMSAR_HIP_RENUMBER", "0")
    dev.upload_structure(s["n_tx"], s["row_ptr"], s["col_idx"])
    dev.upload_sample(None, None, s["den"])
    i = dev.info()
    dev.run_passes(200)
    ms = min(dev.run_passes(200) / 200 for _ in range(3))
    print(env, "ms/pass %.4f" % ms, "units", i["n_units"], "tiles", i["n_chunks"], "padded", i["padded_entries"], "entries", i["tiled_entries"], "far", i["far_entries"],
          "stored MB %.1f" % (i["stored_bytes_per_pass"] / 1e6), flush=True)
for rep in range(2):
    run({})
    for f in ("10", "20", "30"):
        run({"EMSAR_HIP_TAIL_SPLIT": f})
for e in ({"EMSAR_HIP_UNIT_TILES_MAX": "2"}, {"EMSAR_HIP_UNIT_TILES_MAX": "4", "EMSAR_HIP_UNIT_FAR_SOFT": "200"}, {"EMSAR_HIP_UNIT_FAR_SOFT": "120"}, {"EMSAR_HIP_UNIT_FAR_SOFT": "200"}, {}):
    run(e)
s = synth.make_config("cfg3", 1.0, "window")
print("window law", flush=True)
for rep in range(2):
    run({})
    for f in ("10", "20"):
        run({"EMSAR_HIP_TAIL_SPLIT": f})
PY
cat $O/sweep.txt
