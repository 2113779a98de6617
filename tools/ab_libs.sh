# A/B of library builds (EMSAR_HIP_LIB) on config 3, family and window law: ms per pass, units, stored bytes
# usage: tools/ab_libs.sh <tag> <lib> [<lib> ...]      (paths relative to the repo root)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd $R; shift
for S in family window; do for L in "$@"; do
EMSAR_HIP_LIB=$R/$L python - "$S" "$L" <<'PY' >> $O/ab.txt 2>&1
import os, sys
sys.path.insert(0, os.getcwd())
from emsar_amd import EmsarHip, synth
st, lib = sys.argv[1], sys.argv[2]
s = synth.make_config("cfg3", 1.0, st)
dev = EmsarHip(0)
dev.upload_structure(s["n_tx"], s["row_ptr"], s["col_idx"])
dev.upload_sample(None, None, s["den"])
i = dev.info()
dev.run_passes(300)
ms = min(dev.run_passes(200) / 200 for _ in range(4))
th = dev.get_theta()
print(st, lib, "ms/pass %.4f" % ms, "units", i["n_units"], "ids/entry %.3f" % (i["tiled_ids"] / i["tiled_entries"]), "far", i["far_entries"],
      "stored MB %.1f" % (i["stored_bytes_per_pass"] / 1e6), "mass ok", abs((th * s["den"]).sum() - s["n_reads"]) < 1e-8 * s["n_reads"], flush=True)
PY
done; done
cat $O/ab.txt
