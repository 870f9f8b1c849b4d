#!/bin/bash
# round-3 GPU call: rolling-register scan (stream5) vs stream4 / ping-pong scan at 65..256 queries
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3c12
mkdir -p $O
cd $R
UNIIR_TOPK_STREAM5=1 UNIIR_TOPK_STREAM4=4 timeout 600 python -m pytest tests/test_topk_gpu.py -x -q -k "oracle or bit_for_bit" 2>&1 | tail -3
for cfg in "s5:UNIIR_TOPK_STREAM5=1 UNIIR_TOPK_STREAM4=4" "s4:UNIIR_TOPK_STREAM4=4" "default:"; do
  name=${cfg%%:*}; e1=${cfg#*:}
  env $e1 NQS=100,128,192,256 timeout 300 python tools/r3/topk_bench.py > $O/tb_$name.txt 2>&1
  echo "== $name"; grep topk $O/tb_$name.txt
done
cd /tmp && export TMPDIR=/tmp
for nq in 128 256; do
  rm -rf /tmp/tkp_$nq
  UNIIR_TOPK_STREAM5=1 UNIIR_TOPK_STREAM4=4 NQ=$nq timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/tkp_$nq -o t -- python $R/tools/topk_prof.py > /dev/null 2>&1
  DB=$(find /tmp/tkp_$nq -name "*_results.db" | head -1)
  echo "## nq=$nq"; python $R/tools/rocpd_summary.py $DB | grep "topk_" | cut -c1-60,100-140
done
