cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02n; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_random_gpu.py -x -q -m gpu > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt
prof() { n=$1; shift
  env "$@" rocprofv3 --kernel-trace --stats -d $O/$n -o kt --output-format csv -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --solve 0 --spinup 0 > $O/$n.log 2>&1
  python - <<PY
import csv,glob,json
for f in glob.glob("$O/$n/*kernel_stats.csv"):
    for r in list(csv.reader(open(f)))[1:3]: print("$n", r[0][:50], r[1], "avg_us", round(float(r[3])/1e3,2))
PY
}
prof normal A=1
prof prio EMSAR_HIP_LIB=$R/emsar_amd/_variants/libemsar_hip_prio.so
prof skew EMSAR_HIP_AGE_SKEW=1.106,1.034,0.959,0.904
prof skew_b384 EMSAR_HIP_AGE_SKEW=1.106,1.034,0.959,0.904 EMSAR_HIP_TILE_BLOCK=384
find $O -name "*.csv" -size +1M -delete
EMSAR_HIP_LIB=$R/emsar_amd/_variants/libemsar_hip_prio.so EMSAR_TAG=n python tools/chunk_times.py cfg3 > $O/chunk_times_prio.txt 2>&1; head -6 $O/chunk_times_prio.txt | tail -4
# PMC: which unit is busy
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d $O/pmc_sq -o p --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --solve 0 --spinup 0 > $O/p3.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -d $O/pmc_sq2 -o p --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --solve 0 --spinup 0 > $O/p4.log 2>&1
python tools/pmc_summary.py $O k_pass_tiled > $O/pmc_summary.txt 2>&1; cat $O/pmc_summary.txt
find $O -name "*.csv" -size +1M -delete
