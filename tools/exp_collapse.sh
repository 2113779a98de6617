# one GPU call: collapse tests, then the device collapse of config 3 at full size under rocprofv3 (per-kernel times); output under gpurun_out/$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03c}; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_collapse_gpu.py tests/test_cli_gpu.py -x -q -m gpu > $O/gpu_tests.txt 2>&1; rc=$?; tail -5 $O/gpu_tests.txt
[ $rc -eq 0 ] || exit $rc
for S in window family; do
rocprofv3 --kernel-trace --stats -d $O/prof_$S -o c --output-format csv -- python3 tests/perf/collapse_bench.py cfg3 1.0 $S > $O/collapse_$S.txt 2>&1 || exit 1
grep -v "^\[\|^W2\|^E2\|^I2" $O/collapse_$S.txt | tail -5
python - <<PY
import csv,glob
for f in glob.glob("$O/prof_$S/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.reader(open(f)))[1:14]: print("   ", r[0][:70], "calls", r[1], "total_us", round(float(r[2])/1e3,1))
PY
done
find $O -name "*.csv" -size +1M -delete
