R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02e; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_random_gpu.py -x -q -m gpu > $O/gpu_suite.txt 2>&1; tail -3 $O/gpu_suite.txt
run() { n=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 200 --warmup 50 --no-cpu-baseline > $O/$n.json 2> $O/$n.err || true
  python - <<PY
import json
try:
    d=json.load(open("$O/$n.json")); print("$n", round(d["ms_per_step"],5), round(d["roofline"]["device_ms_per_pass"],5), d["mass_conserved"], d["roofline"]["stored_bytes_per_pass"], d["layout_stats"])
except Exception as e: print("$n failed", e)
PY
}
run base A=1
run c2048 EMSAR_HIP_CHUNKS=2048
python tools/chunk_times.py cfg3 > $O/chunk_times.txt 2>&1; cat $O/chunk_times.txt
EMSAR_HIP_CHUNKS=4096 python tools/chunk_times.py cfg3 > $O/chunk_times_4096.txt 2>&1; cat $O/chunk_times_4096.txt
