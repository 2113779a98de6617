# A/B of the weighted pass kernels on the collapsed (segment-level) form of config 3: unit kernel vs one tile per workgroup
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-abw}; mkdir -p $O; cd $R
for S in family; do for U in 0 1 2; do
  EMSAR_HIP_WEIGHTED_UNIT=$U timeout -k 10 400 python bench.py --collapsed --structure $S --steps 200 --warmup 20 --no-live-pmc --no-cpu-baseline --no-variants > $O/$S.$U.json 2> $O/$S.$U.err || exit 1
  python - <<PY
import json
d=json.loads(open("$O/$S.$U.json").read().strip().splitlines()[-1])
print("$S unit=$U", "ms/pass %.4f"%d["roofline"]["device_ms_per_pass"], d["roofline"]["kernel"], d["config"]["collapsed"], "solve", d["solve_to_convergence"]["passes"], "%.3fs"%d["solve_to_convergence"]["seconds"], "tids/entry %.3f"%d["layout_stats"]["tids_per_entry"])
PY
done; done
