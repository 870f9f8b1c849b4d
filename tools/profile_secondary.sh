#!/bin/bash
# rocprofv3 kernel stats of the BLIP_FF large and CLIP_FF train steps (tools/bench_blip.py, tools/bench_clipff.py) -> gpurun_out/prof_sec/
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_sec
mkdir -p $O
for N in ${SEC_LIST:-blip clipff embed}; do
  rm -rf /tmp/ps_$N
  EXTRA="--warmup 1"; [ $N = embed ] && EXTRA=""
  rocprofv3 --kernel-trace --stats -d /tmp/ps_$N -o s -- python $R/tools/bench_$N.py --steps 3 $EXTRA > $O/$N.json 2> $O/$N.err
  DB=$(find /tmp/ps_$N -name "*_results.db" | head -1)
  python $R/tools/rocpd_summary.py $DB > $O/${N}_kernel_stats.txt
  head -32 $O/${N}_kernel_stats.txt | cut -c1-150
  tail -1 $O/$N.json | cut -c1-300
done
