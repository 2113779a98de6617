# one GPU call: the whole -m gpu suite, smoke, then the bench line; output under gpurun_out/$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-round}; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1; rc=$?; tail -5 $O/gpu_tests.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1 && tail -1 $O/smoke.txt &&
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err && tail -1 $O/bench.json | cut -c1-1500
