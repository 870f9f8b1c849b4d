#!/bin/bash
# forward attention with the deferred per-row maximum: kernel / model tests, then timings at 1024 items
cd /root/repo
export PYTHONPATH=/root/repo
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_clip_model_gpu.py tests/test_blip_gpu.py tests/test_clipff_gpu.py tests/test_parity_exact_gpu.py tests/test_fp32_parity_gpu.py -m gpu -x -q > gpurun_out/r4/fwd_pytest.txt 2>&1
tail -6 gpurun_out/r4/fwd_pytest.txt
timeout 300 python tools/r4/attn_pair_check.py > gpurun_out/r4/fwd_check.txt 2>&1
grep -E "^T=.*b=1024|ALL PAIR" gpurun_out/r4/fwd_check.txt
MB_ITEMS=1024 timeout 300 python tools/microbench.py 2>&1 | grep -E "^attn|cross" 
