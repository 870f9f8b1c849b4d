#!/bin/bash
# per-kernel rocprofv3 table of one shard search (700k x 768 fp16, k = 10) at several query counts -> gpurun_out/topk_table.txt
# usage (GPU box): bash tools/topk_table.sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/topk_table.txt
: > $OUT
for NQ in 16 64 128 256 1024; do
  rm -rf /tmp/tkp_$NQ
  NQ=$NQ rocprofv3 --kernel-trace --stats -d /tmp/tkp_$NQ -o t -- python $R/tools/topk_prof.py > /dev/null 2>&1
  DB=$(find /tmp/tkp_$NQ -name "*_results.db" | head -1)
  echo "## nq = $NQ (12 searches; pool 700000 x 768 fp16 = 1.075 GB per sweep)" >> $OUT
  python $R/tools/rocpd_summary.py $DB | grep -v "randn\|distribution\|copyBuffer\|vectorized\|^# rocprofv3" | head -14 >> $OUT
done
# the whole 5.6 M x 768 pool as one resident shard (uniir_topk_ip_multi: 8 scans, one batched tail, one sort, one merge per sweep)
for NQ in 64 256; do
  rm -rf /tmp/tkp_full_$NQ
  ROWS=5600000 NQ=$NQ NSEARCH=6 rocprofv3 --kernel-trace --stats -d /tmp/tkp_full_$NQ -o t -- python $R/tools/topk_prof.py > /dev/null 2>&1
  DB=$(find /tmp/tkp_full_$NQ -name "*_results.db" | head -1)
  echo "## FULL POOL nq = $NQ (6 searches; pool 5600000 x 768 fp16 = 8.6 GB per sweep, 8 logical sub-shards)" >> $OUT
  python $R/tools/rocpd_summary.py $DB | grep -v "randn\|distribution\|copyBuffer\|vectorized\|^# rocprofv3" | head -10 >> $OUT
done
cat $OUT
