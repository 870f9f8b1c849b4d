#!/bin/bash
# text-row packing: kernel + tower tests, then the headline bench (packed value, value_unpacked)
cd /root/repo
export PYTHONPATH=/root/repo
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_clip_model_gpu.py tests/test_parity_exact_gpu.py tests/test_clipff_gpu.py tests/test_pipeline_gpu.py -m gpu -x -q > gpurun_out/r4/pack_pytest.txt 2>&1
tail -15 gpurun_out/r4/pack_pytest.txt
timeout 600 python bench.py --no-secondary > gpurun_out/r4/pack_bench.txt 2>gpurun_out/r4/pack_bench.err
tail -c 3000 gpurun_out/r4/pack_bench.txt; tail -5 gpurun_out/r4/pack_bench.err
