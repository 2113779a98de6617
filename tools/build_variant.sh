#!/bin/bash
# tools/build_variant.sh NAME [-DFLAG ...]  ->  emsar_amd/_variants/libemsar_hip_NAME.so   (experiment builds; select with EMSAR_HIP_LIB)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
N=$1; shift
mkdir -p $R/emsar_amd/_variants $R/build/variants/$N && cd $R/build/variants/$N
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics -std=c++17 -fPIC -shared -Wall -Wextra -Wno-unused-value "$@" -save-temps \
    -o $R/emsar_amd/_variants/libemsar_hip_$N.so $R/emsar_amd/csrc/emsar_hip.hip $R/emsar_amd/csrc/collapse.hip
echo built $R/emsar_amd/_variants/libemsar_hip_$N.so
