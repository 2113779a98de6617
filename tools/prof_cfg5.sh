# PMC traffic of the pass kernel on config 5 at full size (two counter passes; ~100 GiB of host memory): tools/prof_cfg5.sh <tag>
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_cfg5_${1:-r02}
mkdir -p $O
cd $R
B="--config cfg5 --structure window --no-live-pmc --no-variants --no-cpu-baseline --solve 0 --steps 3 --warmup 1 --spinup 0"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o p --output-format csv -- python3 bench.py $B > $O/p1.log 2>&1
echo fetch done
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_EA0_ATOMIC_sum -d $O/pmc_write -o p --output-format csv -- python3 bench.py $B > $O/p2.log 2>&1
python tools/pmc_summary.py $O k_pass_tiled > $O/pmc_summary.txt; cat $O/pmc_summary.txt
find $O -name "*.csv" -size +2M -delete
