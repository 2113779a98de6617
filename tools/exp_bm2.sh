cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02bm; mkdir -p $O; cd $R
export EMSAR_HIP_LIB=$R/emsar_amd/_variants/libemsar_hip_blk4.so
timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_random_gpu.py tests/test_deterministic_gpu.py -x -q -m gpu -k "not full_size and not merge_rows_reduces" > $O/t4.txt 2>&1; rc=$?; tail -3 $O/t4.txt
[ $rc -eq 0 ] || exit $rc
run() { n=$1; shift
  env "$@" python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --solve 0 > $O/$n.json 2>$O/$n.err; tail -1 $O/$n.json | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('$n', d['ms_per_step'], d['roofline']['stored_bytes_per_pass'], d['layout_stats']['n_chunks'], d['layout_stats']['padded_entries'])"
}
run k4b64 EMSAR_HIP_TILE_BLOCK=64
run k4b80 EMSAR_HIP_TILE_BLOCK=80
run k4b96 EMSAR_HIP_TILE_BLOCK=96
run k4b128 EMSAR_HIP_TILE_BLOCK=128
