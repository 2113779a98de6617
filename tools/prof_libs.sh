# per-kernel averages of several library builds on config 3 family: tools/prof_libs.sh <tag> <lib> ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd $R; shift
i=0
for L in "$@"; do
  i=$((i+1)); export EMSAR_HIP_LIB=$R/$L
  rocprofv3 --kernel-trace --stats -d $O/kt$i -o kt --output-format csv -- python3 bench.py --steps 100 --warmup 10 --no-live-pmc --no-cpu-baseline --solve 0 --no-variants > $O/kt$i.log 2>&1
  f=$(find $O/kt$i -name "*kernel_stats.csv" | head -1); echo "$L"; python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.reader(open(sys.argv[1])))[1:3]: print("   ", r[0][27:52], r[1], r[3])
PY
done
find $O -name "*.csv" -size +2M -delete
