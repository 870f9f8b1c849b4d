#!/bin/bash
# kernel tables of the embedding forward with fp16 and with bf16 towers on one box -> gpurun_out/r06_embed/
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export PYTHONPATH=$R
O=$R/gpurun_out/r06_embed
mkdir -p $O
for P in fp16 bf16 fp16 bf16; do
  python $R/tools/bench_embed.py --steps 6 --precision $P 2>/dev/null | tail -1 | cut -c1-200
done
for P in fp16 bf16; do
  rm -rf /tmp/pe_$P
  rocprofv3 --kernel-trace --stats -d /tmp/pe_$P -o s -- python $R/tools/bench_embed.py --steps 3 --precision $P > $O/$P.json 2> $O/$P.err
  DB=$(find /tmp/pe_$P -name "*_results.db" | head -1)
  python $R/tools/rocpd_summary.py $DB > $O/${P}_kernel_stats.txt
  head -24 $O/${P}_kernel_stats.txt | cut -c1-170
done
