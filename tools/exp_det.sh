cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02x; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_deterministic_gpu.py -x -q -m gpu > $O/det_tests.txt 2>&1; rc=$?; tail -15 $O/det_tests.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "full_size" --durations=5 > $O/full_tests.txt 2>&1; rc=$?; tail -12 $O/full_tests.txt
[ $rc -eq 0 ] || exit $rc
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --solve 0 > $O/b0.json 2>$O/b0.err && EMSAR_HIP_DETERMINISTIC=1 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --solve 0 > $O/b1.json 2>$O/b1.err
python - <<PY
import json
for f in ("b0","b1"):
    d=json.loads(open("$O/%s.json"%f).read().strip().split("\n")[-1]); print(f, d["ms_per_step"], d["roofline"]["device_ms_per_pass"])
PY
