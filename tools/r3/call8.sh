#!/bin/bash
# round-3 GPU call 8: filtered scan (stream3) -- parity, timing
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3c8
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_topk_gpu.py "tests/test_fullsize_gpu.py::test_topk_full_shard_properties" tests/test_bench_paths_gpu.py::test_search_shard_multi_sweep_loop_equals_the_oracle tests/test_pipeline_gpu.py -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log
for cfg in "new:" "nofilter:UNIIR_TOPK_FILTER=0"; do
  name=${cfg%%:*}; e1=${cfg#*:}
  env $e1 NQS=16,64 timeout 300 python tools/r3/topk_bench.py > $O/tb_$name.txt 2>&1
  echo "== $name"; grep topk $O/tb_$name.txt
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tkp_a
NQ=64 timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/tkp_a -o t -- python $R/tools/topk_prof.py > /dev/null 2>&1
DB=$(find /tmp/tkp_a -name "*_results.db" | head -1)
python $R/tools/rocpd_summary.py $DB | grep "topk_\|Memset\|fill"
