# same box: family law vs the same matrix with shuffled ids (library numbering on), alternating; and the host-side layout timings
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-ab_shuf}; mkdir -p $O; cd $R
EMSAR_HIP_DEBUG=1 python - > $O/out.txt 2> $O/err.txt <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from emsar_amd import EmsarHip, synth
a = synth.make_config("cfg3", 1.0, "family")
rng = np.random.default_rng(12345)
perm = rng.permutation(a["n_tx"]).astype(np.int32)
den_b = np.empty_like(a["den"]); den_b[perm] = a["den"]
b = {"n_tx": a["n_tx"], "row_ptr": a["row_ptr"], "col_idx": perm[a["col_idx"]], "den": den_b}
dev = EmsarHip(0)
for rep in range(3):
    for name, m in (("family", a), ("shuffled", b)):
        t0 = time.time()
        dev.upload_structure(m["n_tx"], m["row_ptr"], m["col_idx"])
        dev.upload_sample(None, None, m["den"])
        t_up = time.time() - t0
        i = dev.info()
        dev.run_passes(300)
        ms = min(dev.run_passes(200) / 200 for _ in range(4))
        print(rep, name, "ms/pass %.4f" % ms, "units", i["n_units"], "renumbered", i["renumbered"], "upload+layout %.2f s" % t_up, flush=True)
PY
cat $O/out.txt; grep "build_tiled\|upload_structure" $O/err.txt | tail -8
