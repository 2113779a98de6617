cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02g; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_random_gpu.py tests/test_set_solver.py -x -q -m gpu > $O/gpu_tests.txt 2>&1; tail -12 $O/gpu_tests.txt
python tools/chunk_times.py cfg3 > $O/chunk_times.txt 2>&1; cat $O/chunk_times.txt
rocprofv3 --kernel-trace --stats -d $O/exp -o kt --output-format csv -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --solve 0 --spinup 0 > $O/exp.log 2>&1
python - <<PY
import csv,glob
for f in glob.glob("$O/exp/*kernel_stats.csv"):
    for r in list(csv.reader(open(f)))[1:4]: print(r[0][:50], r[1], "avg_us", round(float(r[3])/1e3,2))
PY
find $O -name "*.csv" -size +1M -delete
python bench.py --steps 200 --warmup 50 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; cut -c1-1500 $O/bench.json
