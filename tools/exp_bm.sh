cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02bm; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_random_gpu.py tests/test_deterministic_gpu.py -x -q -m gpu -k "not full_size" > $O/t.txt 2>&1; rc=$?; tail -8 $O/t.txt
[ $rc -eq 0 ] || exit $rc
for b in 256 192 320; do
EMSAR_HIP_TILE_BLOCK=$b python bench.py --steps 40 --warmup 10 --no-cpu-baseline --solve 0 > $O/b$b.json 2>$O/b$b.err; tail -1 $O/b$b.json | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('block $b', d['ms_per_step'], d['roofline']['stored_bytes_per_pass'], d['layout_stats'])"
done
