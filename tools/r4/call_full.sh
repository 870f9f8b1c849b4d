#!/bin/bash
# full GPU test suite + headline bench (no secondary blocks)
cd /root/repo
export PYTHONPATH=/root/repo
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4/full_pytest.txt 2>&1
tail -15 gpurun_out/r4/full_pytest.txt
timeout 400 python bench.py --steps 10 --warmup 3 --no-secondary --no-retrieval --no-cpu-baseline > gpurun_out/r4/full_bench.txt 2>&1
tail -3 gpurun_out/r4/full_bench.txt | cut -c1-1500
