R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02o; mkdir -p $O; cd $R
EMSAR_TAG=o python tools/chunk_times.py cfg3 > $O/chunk_times.txt 2>&1; head -9 $O/chunk_times.txt; tail -2 $O/chunk_times.txt
python bench.py --steps 200 --warmup 50 --no-cpu-baseline --solve 0 --xfam 0 > $O/xfam0.json 2> $O/xfam0.err; cut -c1-200 $O/xfam0.json
