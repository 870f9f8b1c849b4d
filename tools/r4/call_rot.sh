#!/bin/bash
# query / key tiles dealt to the waves from a start that rotates with the head: tests, timings
cd /root/repo
export PYTHONPATH=/root/repo
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_clip_model_gpu.py tests/test_blip_gpu.py tests/test_clipff_gpu.py -m gpu -x -q > gpurun_out/r4/rot_pytest.txt 2>&1
tail -3 gpurun_out/r4/rot_pytest.txt
MB_ITEMS=1024 MB_SKIP_GEMM=1 timeout 300 python tools/microbench.py 2>&1 | grep -E "^attn|cross"
timeout 300 python tools/r4/attn_pair_check.py 2>&1 | grep -E "^T=.*b=1024"
