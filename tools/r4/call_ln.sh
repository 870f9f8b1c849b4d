#!/bin/bash
# LayerNorm with paired chunks (16-byte bf16 accesses): tests + timings
cd /root/repo
export PYTHONPATH=/root/repo
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "layernorm or rmsnorm or ln" 2>&1 | tail -2
MB_ITEMS=1024 MB_SKIP_GEMM=1 timeout 300 python tools/microbench.py 2>&1 | grep -E "^ln"
