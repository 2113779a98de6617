# round-2 experiment C: the chunk/group/slice kernel with exported far entries -- GPU test suite, then bench at several chunk counts
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02c; mkdir -p $O; cd $R
echo skip tests


run() { # name, env...
  n=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 200 --warmup 50 --no-cpu-baseline > $O/$n.json 2> $O/$n.err || true
  python - <<PY
import json
try:
    d=json.load(open("$O/$n.json")); print("$n", round(d["ms_per_step"],5), round(d["roofline"]["device_ms_per_pass"],5), d["mass_conserved"], d["roofline"]["stored_bytes_per_pass"], d["layout_stats"], d["setup_s"])
except Exception as e: print("$n failed", e)
PY
}
run base A=1
run c2048 EMSAR_HIP_CHUNKS=2048
run c4096 EMSAR_HIP_CHUNKS=4096
run c8192 EMSAR_HIP_CHUNKS=8192
run noexport EMSAR_HIP_FAR_EXPORT=0
