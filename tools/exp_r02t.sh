cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02t; mkdir -p $O; cd $R
run() { n=$1; shift
  env "$@" python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --solve 0 --spinup 1 > $O/$n.log 2>&1
  tail -1 $O/$n.log | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('$n', 'ms_per_step', d['ms_per_step'], d['config'].get('layout_stats'))"
}
run base A=1
run noexp EMSAR_HIP_FAR_EXPORT=0
run noexp2k EMSAR_HIP_FAR_EXPORT=0 EMSAR_HIP_CHUNKS=2048
run noexp3k EMSAR_HIP_FAR_EXPORT=0 EMSAR_HIP_CHUNKS=3072
run noexp4k EMSAR_HIP_FAR_EXPORT=0 EMSAR_HIP_CHUNKS=4096
run exp2k EMSAR_HIP_CHUNKS=2048
run exp3k EMSAR_HIP_CHUNKS=3072
