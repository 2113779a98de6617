cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02v; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_collapse_gpu.py tests/test_cli_gpu.py -x -q -m gpu > $O/gpu_tests.txt 2>&1; rc=$?; tail -5 $O/gpu_tests.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tests/perf/collapse_bench.py cfg3 0.2 > $O/collapse_02.txt 2>&1 && cat $O/collapse_02.txt &&
rocprofv3 --kernel-trace --stats -d $O/prof -o c --output-format csv -- python3 tests/perf/collapse_bench.py cfg3 1.0 > $O/collapse_10.txt 2>&1 && grep -v "^\[\|^W2\|^E2\|^I2" $O/collapse_10.txt | tail -4
python - <<PY
import csv,glob
for f in glob.glob("$O/prof/*kernel_stats.csv"):
    for r in list(csv.reader(open(f)))[1:12]: print(r[0][:60], r[1], "total_us", round(float(r[2])/1e3,1))
PY
find $O -name "*.csv" -size +1M -delete
