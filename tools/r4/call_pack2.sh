#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r4/full_pytest2.txt 2>&1
tail -8 gpurun_out/r4/full_pytest2.txt
