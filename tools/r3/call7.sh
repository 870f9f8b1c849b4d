#!/bin/bash
# round-3 GPU call 7: what the group-max stores cost the 64-query scan (timing experiments, results invalid)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3c7
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for X in 0 1 2 3; do
  rm -rf /tmp/tkp_$X
  UNIIR_TOPK_EXP_STORE=$X UNIIR_TOPK_TAIL_STOP=1 NQ=64 timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/tkp_$X -o t -- python $R/tools/topk_prof.py > /dev/null 2>&1
  DB=$(find /tmp/tkp_$X -name "*_results.db" | head -1)
  echo "## exp_store=$X nq=64 (stop after selection)"
  python $R/tools/rocpd_summary.py $DB | grep "topk_stream2"
done
