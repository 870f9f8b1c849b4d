#!/bin/bash
# round 4, call 1: where does the attention kernels' time go (knock-outs + timeline), same-box baseline microbench
cd /root/repo
export PYTHONPATH=/root/repo
mkdir -p gpurun_out/r4
MB_ONLY=attn MB_ITEMS=1024 timeout 300 python tools/microbench.py > gpurun_out/r4/c1_microbench.txt 2>&1
UNIIR_HIP_LIB=/root/repo/uniir_amd/libuniir_exp_attdiag.so timeout 600 python tools/r4/attn_diag.py > gpurun_out/r4/c1_diag257.txt 2>&1
T=77 H=12 CAUSAL=1 UNIIR_HIP_LIB=/root/repo/uniir_amd/libuniir_exp_attdiag.so timeout 600 python tools/r4/attn_diag.py > gpurun_out/r4/c1_diag77.txt 2>&1
tail -50 gpurun_out/r4/c1_diag257.txt
