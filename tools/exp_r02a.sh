# round-2 experiment A: where does the pass stand on this box, and what is the prize of taking far entries out of the tiles
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02a; mkdir -p $O; cd $R
python bench.py --steps 200 --warmup 50 --no-cpu-baseline > $O/base.json 2> $O/base.err; cat $O/base.json | cut -c1-400
python bench.py --steps 200 --warmup 50 --no-cpu-baseline --xfam 0 > $O/xfam0.json 2> $O/xfam0.err; cat $O/xfam0.json | cut -c1-400
EMSAR_HIP_TILED_MULTI=0 python bench.py --steps 200 --warmup 50 --no-cpu-baseline --xfam 0 > $O/xfam0_single.json 2> $O/xfam0_single.err; cat $O/xfam0_single.json | cut -c1-300
python tools/tiled_stamps.py cfg3 > $O/stamps.txt 2>&1; cat $O/stamps.txt
