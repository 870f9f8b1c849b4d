#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 600 python -m pytest tests/test_topk_gpu.py tests/test_bench_paths_gpu.py "tests/test_fullsize_gpu.py::test_topk_full_shard_properties" -x -q 2>&1 | tail -3
UNIIR_TOPK_STREAM5=1 timeout 600 python -m pytest tests/test_topk_gpu.py -x -q -k "oracle or bit_for_bit" 2>&1 | tail -2
for rep in 1 2; do NQS=64,100,128,192,256 python tools/r3/topk_bench.py 2>&1 | grep topk; done
