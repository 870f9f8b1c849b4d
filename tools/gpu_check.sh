#!/bin/bash
# One parameterised GPU call for work in progress (replaces the per-experiment tools/r3/call*.sh, tools/r4/call*.sh):
#   tools/gpu_check.sh "<pytest selection, e.g. tests/test_kernels_gpu.py -k gemm>" [bench args ...]
# runs the selection with -m gpu -x -q, then the headline step (no secondary blocks) and prints its summary.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export PYTHONPATH=$R
SEL=$1; shift
if [ -n "$SEL" ]; then timeout 1500 python -m pytest $SEL -m gpu -x -q 2>&1 | tail -4; fi
timeout 600 python bench.py --no-secondary --no-retrieval --no-cpu-baseline --steps 8 --warmup 3 "$@" 2>/dev/null > /tmp/gpu_check_line.json
python tools/bench_summary.py /tmp/gpu_check_line.json
