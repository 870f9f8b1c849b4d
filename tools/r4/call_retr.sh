#!/bin/bash
# retrieval-side checks after the topk.hip clean-up + sub-shard split + bench N>1 blocks
cd /root/repo
export PYTHONPATH=/root/repo
mkdir -p gpurun_out/r4
timeout 1200 python -m pytest tests/test_topk_gpu.py tests/test_fullsize_gpu.py tests/test_bench_paths_gpu.py tests/test_pipeline_gpu.py tests/test_bench_launcher.py tests/test_kernels_gpu.py tests/test_clip_model_gpu.py -m gpu -x -q > gpurun_out/r4/retr_pytest.txt 2>&1
tail -12 gpurun_out/r4/retr_pytest.txt
