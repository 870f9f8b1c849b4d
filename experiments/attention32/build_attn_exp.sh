#!/bin/bash
# experimental build of the attention object only, linked with the regular objects:
#   tools/build_attn_exp.sh NAME [-DATT_...=..]   ->  uniir_amd/libuniir_exp_NAME.so   (use with UNIIR_HIP_LIB=<path>)
set -e
cd "$(dirname "$0")/../uniir_amd/csrc"
name=$1; shift
mkdir -p build/exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result "$@" -c attention.hip -o build/exp/attn_exp_$name.o
objs=$(ls build/*.o | grep -v "attention" )
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libuniir_exp_$name.so build/exp/attn_exp_$name.o $objs
echo built ../libuniir_exp_$name.so
