#!/bin/bash
# round-3 GPU call: rolling-register kernel at <= 64 queries (QW = 1) vs stream2; full topk suite on the new defaults
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3c13
mkdir -p $O
cd $R
true 2>&1 | tail -3
timeout 600 python -m pytest tests/test_topk_gpu.py tests/test_bench_paths_gpu.py "tests/test_fullsize_gpu.py::test_topk_full_shard_properties" -x -q 2>&1 | tail -3
for cfg in "default:" "defaultb:"; do
  name=${cfg%%:*}; e1=${cfg#*:}
  env $e1 NQS=1,16,64 timeout 300 python tools/r3/topk_bench.py > $O/tb_$name.txt 2>&1
  echo "== $name"; grep topk $O/tb_$name.txt
done
cd /tmp && export TMPDIR=/tmp
for nq in 64; do
  rm -rf /tmp/tkp_$nq
  NQ=$nq timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/tkp_$nq -o t -- python $R/tools/topk_prof.py > /dev/null 2>&1
  DB=$(find /tmp/tkp_$nq -name "*_results.db" | head -1)
  echo "## nq=$nq"; python $R/tools/rocpd_summary.py $DB | grep "topk_" | cut -c1-60,100-140
done
