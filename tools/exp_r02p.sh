cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02q; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_random_gpu.py tests/test_collapse_gpu.py -x -q -m gpu > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt
prof() { n=$1; shift
  env "$@" rocprofv3 --kernel-trace --stats -d $O/$n -o kt --output-format csv -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --solve 0 --spinup 0 > $O/$n.log 2>&1
  python - <<PY
import csv,glob,json
for f in glob.glob("$O/$n/*kernel_stats.csv"):
    for r in list(csv.reader(open(f)))[1:3]: print("$n", r[0][:50], r[1], "avg_us", round(float(r[3])/1e3,2))
PY
}
prof normal A=1
prof b384 EMSAR_HIP_TILE_BLOCK=384
find $O -name "*.csv" -size +1M -delete
EMSAR_TAG=q python tools/chunk_times.py cfg3 > $O/chunk_times.txt 2>&1; head -9 $O/chunk_times.txt | tail -7; tail -2 $O/chunk_times.txt
