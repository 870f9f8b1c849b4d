#!/bin/bash
# the 8-column DACT copy-out: GEMM tests, microbench (all forms: the other epilogues must not move), tower tests
cd /root/repo
export PYTHONPATH=/root/repo
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gemm" > gpurun_out/r4/dact8_pytest.txt 2>&1
tail -3 gpurun_out/r4/dact8_pytest.txt
MB_ITEMS=1024 timeout 300 python tools/microbench.py 2>&1 | grep -E "^gemm"
