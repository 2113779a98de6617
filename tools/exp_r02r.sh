R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02r; mkdir -p $O; cd $R
EMSAR_TAG=r0 python tools/chunk_times.py cfg3 > $O/ct0.txt 2>&1; sed -n 2,6p $O/ct0.txt; tail -1 $O/ct0.txt
EMSAR_HIP_CHUNK_REVERSE=1 EMSAR_TAG=r1 python tools/chunk_times.py cfg3 > $O/ct1.txt 2>&1; sed -n 2,6p $O/ct1.txt; tail -1 $O/ct1.txt
