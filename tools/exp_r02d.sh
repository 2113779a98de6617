# which kernel takes the time: rocprofv3 kernel stats with and without the far export
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02d; mkdir -p $O; cd $R
rocprofv3 --kernel-trace --stats -d $O/exp -o kt --output-format csv -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/exp.log 2>&1
EMSAR_HIP_FAR_EXPORT=0 rocprofv3 --kernel-trace --stats -d $O/noexp -o kt --output-format csv -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/noexp.log 2>&1
for d in exp noexp; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); echo $d; head -6 $f | cut -c1-200; done
find $O -name "*.csv" -size +1M -delete
