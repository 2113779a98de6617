cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02bm; mkdir -p $O; cd $R
run() { n=$1; shift
  env "$@" python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --solve 0 > $O/$n.json 2>$O/$n.err; tail -1 $O/$n.json | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('$n', d['ms_per_step'], d['roofline']['stored_bytes_per_pass'], d['layout_stats']['n_chunks'], d['layout_stats']['padded_entries'])"
}
run def A=1
run b64 EMSAR_HIP_TILE_BLOCK=64
run b80 EMSAR_HIP_TILE_BLOCK=80
run b112 EMSAR_HIP_TILE_BLOCK=112
