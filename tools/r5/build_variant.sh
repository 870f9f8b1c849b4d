#!/bin/bash
# full-library variant with extra -D flags for csrc/gemm.hip:  tools/r5/build_variant.sh NAME [-DFLAG=..]  ->  uniir_amd/libuniir_var_NAME.so
set -e
cd "$(dirname "$0")/../../uniir_amd/csrc"
name=$1; shift
mkdir -p build/exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result "$@" -c gemm.hip -o build/exp/gemm_var_$name.o 2>/dev/null
objs=$(ls build/*.o | grep -v "build/gemm.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libuniir_var_$name.so build/exp/gemm_var_$name.o $objs
echo built libuniir_var_$name.so
