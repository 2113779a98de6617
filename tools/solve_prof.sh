# kernel mix of the SQUAREM streaming solve on config 3 (run on the GPU box): tools/solve_prof.sh <tag> [ENV=VALUE ...]
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
for kv in "$@"; do export "$kv"; done
O=$R/gpurun_out/solve_prof_$tag
rocprofv3 --kernel-trace --stats -d $O -o sv --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-variants --spinup 0 --solve 1e-6 > $O.txt 2>&1
f=$(find $O -name "*kernel_stats.csv" | head -1)
echo "== $tag $@"; cut -d, -f1-4 $f | sed 's/(anonymous namespace):://g; s/(emsar::Tile const.*)"/"/; s/(int, double const.*)"/"/' | head -9
grep -o "\"solve_to_convergence\": {[^}]*}" $O.txt || true
find $O -name "*.csv" -size +2M -delete
