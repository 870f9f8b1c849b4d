#!/bin/bash
# fast experimental build of the GEMM object only (NT ping-pong kernel), linked with the regular objects:
#   tools/build_exp.sh NAME [-DPP_EXP=..]   ->  uniir_amd/libuniir_exp_NAME.so   (use with UNIIR_HIP_LIB=<path>)
set -e
cd "$(dirname "$0")/../uniir_amd/csrc"
name=$1; shift
mkdir -p build/exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DUNIIR_EXP_BUILD "$@" -c gemm.hip -o build/exp/gemm_exp_$name.o
objs=$(ls build/*.o | grep -v "gemm" )
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libuniir_exp_$name.so build/exp/gemm_exp_$name.o $objs
echo built ../libuniir_exp_$name.so
