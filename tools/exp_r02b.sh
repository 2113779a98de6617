# round-2 experiment B: conflict-free gather ablations, the pipelined gathers, one tile per workgroup
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02b; mkdir -p $O; cd $R
run() { # name, env...
  n=$1; shift
  env "$@" python bench.py --steps 200 --warmup 50 --no-cpu-baseline > $O/$n.json 2> $O/$n.err || true
  python - <<PY
import json
try:
    d=json.load(open("$O/$n.json")); print("$n", round(d["ms_per_step"],5), round(d["roofline"]["device_ms_per_pass"],5), d["mass_conserved"], d["roofline"]["kernel"])
except Exception as e: print("$n failed", e)
PY
}
V=$R/emsar_amd/_variants
run base A=1
run single EMSAR_HIP_TILED_MULTI=0
run e5 EMSAR_HIP_LIB=$V/libemsar_hip_e5.so
run m5 EMSAR_HIP_LIB=$V/libemsar_hip_m5.so
run em5 EMSAR_HIP_LIB=$V/libemsar_hip_em5.so
run em5_single EMSAR_HIP_LIB=$V/libemsar_hip_em5.so EMSAR_HIP_TILED_MULTI=0
run pipe EMSAR_HIP_LIB=$V/libemsar_hip_pipe.so
run pipe_single EMSAR_HIP_LIB=$V/libemsar_hip_pipe.so EMSAR_HIP_TILED_MULTI=0
