#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for cfg in "base:" "gnt:UNIIR_HIP_LIB=$R/experiments/build/libuniir_gnt.so" "base2:" "gnt2:UNIIR_HIP_LIB=$R/experiments/build/libuniir_gnt.so"; do
  name=${cfg%%:*}; e1=${cfg#*:}
  echo "== $name"; env $e1 NQS=16,64,128,256 timeout 300 python tools/r3/topk_bench.py 2>&1 | grep topk
done
