#!/bin/bash
# GEMM tile rasterisation (row panels x column panels per XCD working set): step time, interleaved, twice
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
ARGS="--steps 8 --warmup 2 --no-cpu-baseline --no-secondary --no-retrieval"
for rep in 1 2; do
for r in "8,4" "16,4" "8,2" "4,4" "16,8" "0"; do
  echo "== raster $r $(UNIIR_GEMM_RASTER=$r timeout 300 python bench.py $ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['achieved'])")"
done
done
