#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for cfg in "base:" "relaxed:UNIIR_TOPK_EXP_STORE=4" "nostore:UNIIR_TOPK_EXP_STORE=3 UNIIR_TOPK_HIER=0"; do
  name=${cfg%%:*}; e1=${cfg#*:}
  for nq in 16 64; do
    rm -rf /tmp/tkp_$nq
    env $e1 NQ=$nq NSEARCH=40 timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/tkp_$nq -o t -- python $R/tools/topk_prof.py > /dev/null 2>&1
    DB=$(find /tmp/tkp_$nq -name "*_results.db" | head -1)
    echo "## $name nq=$nq $(python $R/tools/rocpd_summary.py $DB | grep "topk_stream" | cut -c100-140)"
  done
done
done
