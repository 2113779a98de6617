cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02f; mkdir -p $O; cd $R
python tools/chunk_times.py cfg3 > $O/chunk_times.txt 2>&1; cat $O/chunk_times.txt
rocprofv3 --kernel-trace --stats -d $O/exp -o kt --output-format csv -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --solve 0 --spinup 0 > $O/exp.log 2>&1
f=$(find $O/exp -name "*kernel_stats.csv" | head -1); head -4 $f | cut -c1-60,230-330
find $O -name "*.csv" -size +1M -delete
timeout -k 10 900 python -m pytest tests/test_set_solver.py tests/test_cli_gpu.py -x -q -m gpu > $O/gpu_tests.txt 2>&1; tail -15 $O/gpu_tests.txt
