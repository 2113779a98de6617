cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02i; mkdir -p $O; cd $R
prof() { n=$1; shift
  env "$@" rocprofv3 --kernel-trace --stats -d $O/$n -o kt --output-format csv -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --solve 0 --spinup 0 > $O/$n.log 2>&1
  python - <<PY
import csv,glob
for f in glob.glob("$O/$n/*kernel_stats.csv"):
    for r in list(csv.reader(open(f)))[1:3]: print("$n", r[0][:50], r[1], "avg_us", round(float(r[3])/1e3,2))
PY
}
prof dbg1_noread EMSAR_HIP_DBG_FARSUM=1
prof dbg2_nolds EMSAR_HIP_DBG_FARSUM=2
find $O -name "*.csv" -size +1M -delete
EMSAR_TAG=pipe python tools/chunk_times.py cfg3 > $O/chunk_times.txt 2>&1; tail -3 $O/chunk_times.txt
