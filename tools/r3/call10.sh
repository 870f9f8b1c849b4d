#!/bin/bash
# round-3 GPU call 10: the 256x128 two-workgroups-per-CU GEMM: bitwise check against the 256x256 kernel, then timing
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3c10
mkdir -p $O
cd $R
timeout 300 python tools/r3/pp2_check.py /tmp/pp_ref.pt 2>&1 | tail -2
UNIIR_GEMM_PP2=1 timeout 300 python tools/r3/pp2_check.py /tmp/pp2.pt 2>&1 | tail -2
python tools/r3/pp2_check.py cmp /tmp/pp_ref.pt /tmp/pp2.pt 2>&1 | tail -12
echo "== microbench 256x256"; MB_ITEMS=1024 timeout 300 python tools/microbench.py 2>&1 | grep "gemm" | tee $O/mb_pp.txt
echo "== microbench pp2"; UNIIR_GEMM_PP2=1 MB_ITEMS=1024 timeout 300 python tools/microbench.py 2>&1 | grep "gemm" | tee $O/mb_pp2.txt
