#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3c9
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_topk_gpu.py "tests/test_fullsize_gpu.py::test_topk_full_shard_properties" -x -q > $O/pytest.log 2>&1; grep -n "^E " $O/pytest.log | head -8; tail -2 $O/pytest.log
python tools/r3/filt_debug.py 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp
for cfg in "filt:" "dense:UNIIR_TOPK_FILTER=0" "filt2:" "dense2:UNIIR_TOPK_FILTER=0"; do
  name=${cfg%%:*}; rest=${cfg#*:}; e1=${rest%%:*}
  rm -rf /tmp/tkp_$name
  env $e1 NQ=64 timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/tkp_$name -o t -- python $R/tools/topk_prof.py > /dev/null 2>&1
  DB=$(find /tmp/tkp_$name -name "*_results.db" | head -1)
  echo "## $name"; python $R/tools/rocpd_summary.py $DB | grep "topk_\|fillBuffer" | cut -c1-60,100-140
done
cd $R
for cfg in "new:" "nofilter:UNIIR_TOPK_FILTER=0"; do
  name=${cfg%%:*}; e1=${cfg#*:}
  env $e1 NQS=16,64 timeout 300 python tools/r3/topk_bench.py > $O/tb_$name.txt 2>&1
  echo "== $name"; grep topk $O/tb_$name.txt
done
