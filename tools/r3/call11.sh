#!/bin/bash
# round-3 GPU call 11: the shared-ring streaming scan for 65..256 queries: oracle parity, A/B timing, per-kernel table
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3c11
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_topk_gpu.py "tests/test_fullsize_gpu.py::test_topk_full_shard_properties" -x -q > $O/pytest.log 2>&1; grep -n "^E " $O/pytest.log | head -8; tail -2 $O/pytest.log
for cfg in "new:" "pp:UNIIR_TOPK_STREAM4=0" "qw4:UNIIR_TOPK_STREAM4=4"; do
  name=${cfg%%:*}; e1=${cfg#*:}
  env $e1 NQS=64,65,100,128,192,256 timeout 300 python tools/r3/topk_bench.py > $O/tb_$name.txt 2>&1
  echo "== $name"; grep topk $O/tb_$name.txt
done
cd /tmp && export TMPDIR=/tmp
for nq in 128 256; do
  rm -rf /tmp/tkp_$nq
  UNIIR_TOPK_STREAM4=4 NQ=$nq timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/tkp_$nq -o t -- python $R/tools/topk_prof.py > /dev/null 2>&1
  DB=$(find /tmp/tkp_$nq -name "*_results.db" | head -1)
  echo "## nq=$nq"; python $R/tools/rocpd_summary.py $DB | grep "topk_" | cut -c1-60,100-140
done
