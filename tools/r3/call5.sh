#!/bin/bash
# round-3 GPU call 5: tail breakdown of the fused search (stop modes)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3c5
mkdir -p $O
cd $R
for cfg in "new:" "stop1:UNIIR_TOPK_TAIL_STOP=1" "stop2:UNIIR_TOPK_TAIL_STOP=2"; do
  name=${cfg%%:*}; e1=${cfg#*:}
  env $e1 NQS=16,64,128 timeout 300 python tools/r3/topk_bench.py > $O/tb_$name.txt 2>&1
  echo "== $name"; grep topk $O/tb_$name.txt
done
cd /tmp && export TMPDIR=/tmp
for S in 0 1 2; do
  rm -rf /tmp/tkp_$S
  UNIIR_TOPK_TAIL_STOP=$S NQ=64 timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/tkp_$S -o t -- python $R/tools/topk_prof.py > /dev/null 2>&1
  DB=$(find /tmp/tkp_$S -name "*_results.db" | head -1)
  echo "## stop=$S nq=64"
  python $R/tools/rocpd_summary.py $DB | grep "topk_"
done
