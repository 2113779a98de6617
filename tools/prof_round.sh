# one GPU call: rocprofv3 kernel stats + PMC passes of the bench command (output under gpurun_out/prof_$1; copy the summaries to profiles/)
# usage: tools/prof_round.sh <tag> [structure: family|window|family_shuffled] [extra bench flags, e.g. --collapsed]
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_${1:-r02}
mkdir -p $O
cd $R
S=${2:-family}; X=${3:-}
B="--no-live-pmc --no-cpu-baseline --solve 0 --no-variants --structure $S $X"
python bench.py --steps 20 --warmup 5 --structure $S $X > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | cut -c1-400
rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 bench.py --steps 100 --warmup 10 $B > $O/kt.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o p --output-format csv -- python3 bench.py --steps 3 --warmup 1 --spinup 0 $B > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_EA0_ATOMIC_sum -d $O/pmc_write -o p --output-format csv -- python3 bench.py --steps 3 --warmup 1 --spinup 0 $B > $O/p2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d $O/pmc_sq -o p --output-format csv -- python3 bench.py --steps 3 --warmup 1 --spinup 0 $B > $O/p3.log 2>&1
python tools/pmc_summary.py $O k_pass_tiled > $O/pmc_summary.txt; cat $O/pmc_summary.txt
f=$(find $O/kt -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-200
find $O -name "*.csv" -size +2M -delete
