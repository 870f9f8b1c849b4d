#!/bin/bash
# HBM-side traffic of the shard search (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, kernel-trace only) at 64 and 256 queries
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/topk_pmc
mkdir -p $O
for NQ in 64 256; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/tp_$C
    NQ=$NQ NSEARCH=8 timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $C -d /tmp/tp_$C -o p -- python $R/tools/topk_prof.py > /dev/null 2> $O/err_${NQ}_$C.txt
    F=$(find /tmp/tp_$C -name "*counter_collection.csv" | head -1)
    echo "## nq = $NQ  $C" >> $O/topk_pmc.txt
    python $R/tools/pmc_summary.py $F | grep -v "randn\|distribution\|vectorized\|arange" >> $O/topk_pmc.txt
  done
done
cat $O/topk_pmc.txt | cut -c1-150
