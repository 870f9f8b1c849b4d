#!/bin/bash
# round-3 evidence run (GPU box): full GPU test suite, the default bench line, rocprofv3 kernel stats + PMC passes of the step,
# the retrieval per-kernel table, the embed profile, the isolated-kernel record -> gpurun_out/r3final/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3final
mkdir -p $O
cd $R
if [ "${SKIP_TESTS:-0}" != 1 ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
fi
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 1500 $O/bench_line.json | head -c 1500; echo
bash tools/profile_bench.sh > $O/profile_bench.out 2>&1; tail -12 $O/profile_bench.out | cut -c1-200
bash tools/topk_table.sh > /dev/null 2>&1; cp gpurun_out/topk_table.txt $O/
SEC_LIST=embed bash tools/profile_secondary.sh > $O/profile_embed.out 2>&1; tail -3 $O/profile_embed.out | cut -c1-200
MB_ITEMS=1024 timeout 600 python tools/microbench.py > $O/microbench.txt 2>&1; tail -5 $O/microbench.txt
