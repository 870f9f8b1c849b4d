#!/bin/bash
# The round's closing GPU call: what the driver runs at round end, in the driver's order, on the snapshot of the
# commit named on the command line --
#   HEAD=$(git rev-parse HEAD) tools/gpu_final.sh <tag>            (through gpurun; .git does not travel, hence the variable)
# full `pytest tests -x -q -m gpu`, then __graft_entry__.smoke(), then the driver-shaped bench (N = 1, 20 steps after 5).
# Everything lands in gpurun_out/<tag>/; pytest_gpu.log starts with the commit.  After this call only *.md / profiles/ may change.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export PYTHONPATH=$R
TAG=${1:-final}
OUT=gpurun_out/$TAG
mkdir -p $OUT
{
  echo "git rev-parse HEAD: ${HEAD:-unknown}"
  echo "libuniir_hip.so sha256: $(sha256sum uniir_amd/libuniir_hip.so | cut -c1-16)   bench.py sha256: $(sha256sum bench.py | cut -c1-16)"
  timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -15
} > $OUT/pytest_gpu.log 2>&1
tail -3 $OUT/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc $?"; tail -2 $OUT/smoke.log
if [ -z "$NO_BENCH" ]; then
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench.err; echo "bench rc $?"
  python tools/bench_summary.py $OUT/bench_line.json 2>&1 | tail -40
fi
