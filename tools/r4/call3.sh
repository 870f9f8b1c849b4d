#!/bin/bash
# round 4, call 3: which heads differ from the general kernel; timeline of the pair kernels
cd /root/repo
export PYTHONPATH=/root/repo
mkdir -p gpurun_out/r4
timeout 300 python tools/r4/attn_pair_check.py > gpurun_out/r4/c3_pair_check.txt 2>&1
UNIIR_HIP_LIB=/root/repo/uniir_amd/libuniir_exp_attdiag.so timeout 300 python tools/r4/attn_pair_stamps.py > gpurun_out/r4/c3_stamps.txt 2>&1
cat gpurun_out/r4/c3_pair_check.txt | grep -v "^T=2.. H=.* b=.*: fwd vs" | head -60
cat gpurun_out/r4/c3_stamps.txt
