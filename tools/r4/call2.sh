#!/bin/bash
# round 4, call 2: pair-tile backward -- parity against the general kernel / torch, timing
cd /root/repo
export PYTHONPATH=/root/repo
mkdir -p gpurun_out/r4
timeout 300 python tools/r4/attn_pair_check.py > gpurun_out/r4/c2_pair_check.txt 2>&1
echo "rc=$?" >> gpurun_out/r4/c2_pair_check.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" > gpurun_out/r4/c2_pytest_attn.txt 2>&1
tail -5 gpurun_out/r4/c2_pytest_attn.txt
cat gpurun_out/r4/c2_pair_check.txt
