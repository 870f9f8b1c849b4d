#!/bin/bash
# nt loads in the attention kernels / the GEMM epilogues (resid, aux): step time A/B, interleaved, twice
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
ARGS="--steps 8 --warmup 2 --no-cpu-baseline --no-secondary --no-retrieval"
for rep in 1 2; do
for cfg in "base:" "attnt:UNIIR_HIP_LIB=$R/experiments/build/libuniir_attnt.so" "epint:UNIIR_HIP_LIB=$R/experiments/build/libuniir_epint.so"; do
  name=${cfg%%:*}; e1=${cfg#*:}
  echo "== $name $(env $e1 timeout 300 python bench.py $ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['achieved'])")"
done
done
